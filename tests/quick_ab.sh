# quick A/B on the GPU box: attention parity subset, single-utterance latency, short bench
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$1" 2>&1 | tail -3
python3 tests/b1_latency.py 40 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('b1', d['median_ms_per_call'], d['min_ms_per_call'])"
python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step']); r=d['roofline']['per_config_ms']; print({k:round(v,2) for k,v in r.items() if v>1})"
