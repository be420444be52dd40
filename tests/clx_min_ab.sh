for v in 1024 400 100; do
  export SBV2_CLX_MIN_TILES=$v
  echo "== SBV2_CLX_MIN_TILES=$v"
  python3 tests/b1_latency.py 40 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('b1', d['median_ms_per_call'], d['min_ms_per_call'])"
  python3 tests/long_form_check.py 2000 256 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('long', d['whole_sequence_ms'], d['stream_time_to_first_chunk_ms'], d['stream_total_ms'], d['chunked_vs_whole_max_abs'])"
done
