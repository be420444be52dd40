export TMPDIR=/tmp
rm -rf gpurun_out/pmc_iso
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_iso -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_iso/*/*counter_collection.csv')[0]
dur=collections.defaultdict(list); seen=set(); cnt=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:70]
    cnt[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Dispatch_Id'] in seen: continue
    seen.add(r['Dispatch_Id']); dur[k].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
rows=sorted(dur.items(), key=lambda kv:-sum(kv[1]))
for k,v in rows[:26]:
    c=cnt[k]; n=len(v)
    print(f"{k:70s} n {n:4d} avg_us {sum(v)/n/1e3:8.1f} tot_ms/step {sum(v)/1e6/3:7.2f} bankconf {c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1):.3f} mfma {c['SQ_VALU_MFMA_BUSY_CYCLES']/max(c['GRBM_GUI_ACTIVE']/8*1024,1):.3f}")
PY
