SBV2_CLX_CFG=8 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "clx" 2>&1 | tail -2
for v in 7 8 7 8; do SBV2_CLX_CFG=$v python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('clxcfg=$v', d['value'], d['ms_per_step'], {k:round(v,2) for k,v in d['roofline']['per_config_ms'].items() if 'clx' in k})
"; done
