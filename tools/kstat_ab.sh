# Builder tool: isolated kernel durations of bench steps (pipeline depth 1: one execution context, no overlap; bench.py's collect then fails on its first
# ticket, AFTER the traced launches: the ms-per-step line stays empty, the kernel statistics are those of the steps that ran) for a list of
# "NAME=VALUE" environment settings ("-" = defaults); prints the kernels whose name contains $PAT.
#   bash tools/kstat_ab.sh flash - SBV2_FLASH_Q=0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
PAT=$1; shift
export SBV2_PIPELINE_DEPTH=1
i=0
for setting in "$@"; do
  i=$((i+1))
  if [ "$setting" != "-" ]; then export "$setting"; fi
  rm -rf $O/kab_$i
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kab_$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kab_$i.log 2>&1
  echo "== $setting: $(tail -1 $O/kab_$i.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
  python3 - "$O/kab_$i" "$PAT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
tot = 0
for r in csv.DictReader(open(f)):
    tot += float(r["TotalDurationNs"])
    if any(p in r["Name"] for p in sys.argv[2].split(",")):
        print(f"  {r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us  total/step {float(r['TotalDurationNs']) / 4e6:7.3f} ms")
print(f"  all kernels {tot / 4e6:.2f} ms per step")
PY
  if [ "$setting" != "-" ]; then unset "${setting%%=*}"; fi
done
