for v in 1 5 6 7 1 5 6 7; do SBV2_BFS_HALF=$v python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('half=$v', d['value'], d['ms_per_step'], {k:round(v,2) for k,v in d['roofline']['per_config_ms'].items() if 'bfs' in k})
"; done
