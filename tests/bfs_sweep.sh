for m in f32 bf16x6 bf16x3 f32 bf16x6; do echo "=== SBV2_BERT_GEMM=$m"; SBV2_BERT_GEMM=$m timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], json.dumps(r['per_config_ms']))
    else: print(l.rstrip())
"; done
