# Builder tool: single-utterance latency under a list of NAME=VALUE settings ("-" = defaults)
for setting in "$@"; do
  if [ "$setting" != "-" ]; then export "$setting"; fi
  python3 tests/b1_latency.py 40 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$setting', 'b1', d['median_ms_per_call'], d['min_ms_per_call'])"
  if [ "$setting" != "-" ]; then unset "${setting%%=*}"; fi
done
