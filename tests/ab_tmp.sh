python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "clx or vits_e2e or config2 or decoder" 2>&1 | tail -2
for v in 0 1 0 1; do SBV2_CLX_UPSPLIT=$v python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('upsplit=$v', d['value'], d['ms_per_step'], {k:round(v,2) for k,v in d['roofline']['per_config_ms'].items() if 'conv_cl' in k})
"; done
