timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "clx or vits_full or config2 or vits_e2e" 2>&1 | tail -4
for m in 1 0 1 0; do echo "=== SBV2_CLX=$m"; SBV2_CLX=$m timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], json.dumps(r['per_config_ms']))
    else: print(l.rstrip())
"; done
