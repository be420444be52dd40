python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "respair_clx" 2>&1 | tail -2
for v in 0 1 0 1; do SBV2_RPX_RRES_LATE=$v python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('late=$v', d['value'], d['ms_per_step'], {k:round(v,2) for k,v in d['roofline']['per_config_ms'].items() if 'respair' in k})
"; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r04g_fetch -- python3 tests/respair_pmc.py 3 > /dev/null 2>&1
python tests/pmc_sum.py gpurun_out/r04g_fetch/.. respair 2>/dev/null | tail -3
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/r04g_fetch/*/*counter_collection.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'respair' in r['Kernel_Name']: agg[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, 'FETCH bytes (x2):', sum(v)/len(v)*1024*2/1e9, 'GB')
PY
