# Builder tool: short bench runs under a list of NAME=VALUE settings ("-" = defaults); prints value, ms per step and the buckets named in $PAT
PAT=$1; shift
for setting in "$@"; do
  if [ "$setting" != "-" ]; then export "$setting"; fi
  python3 bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']['per_config_ms']
print('$setting', d['value'], d['ms_per_step'], {k:round(v,2) for k,v in r.items() if any(p in k for p in '$PAT'.split(','))})"
  if [ "$setting" != "-" ]; then unset "${setting%%=*}"; fi
done
