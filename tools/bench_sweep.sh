#!/bin/bash
# usage: tools/bench_sweep.sh "<env assignments>" <bench args...>   -> prints ms_per_step, value, dominant achieved, conv ms
env $1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "${@:2}" 2>&1 | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], d["value"], r["kernel"], r["achieved"], r["all_conv_gemm_ms_per_step"])'
