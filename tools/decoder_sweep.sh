# SBV2_DECODER sweep at HEAD (same box, same build): step time and per-kernel buckets of the instrumented step
for m in bf16x3 f16 bf16 bf16x3; do echo "=== SBV2_DECODER=$m"; SBV2_DECODER=$m timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(json.dumps({'decoder': '$m', 'audio_s_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'dominant': r['kernel'], 'achieved_tflops': r['achieved'], 'per_config_ms': r['per_config_ms']}))
    else: print(l.rstrip())
"; done
