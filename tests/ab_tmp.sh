for v in 2 3 4 2 3; do SBV2_PIPELINE_DEPTH=$v python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('depth=$v', d['value'], d['ms_per_step'])
"; done
