# Builder tool (GPU box): board power / shader clock while a command runs, sampled from the amdgpu hwmon files every 50 ms.
#   bash tools/power_trace.sh <out.csv> <command ...>
OUT=$1; shift
H=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
echo "hwmon: $H" >&2
ls $H >&2
for f in power1_cap power1_cap_max power1_average power1_input freq1_input; do [ -r $H/$f ] && echo "$f $(cat $H/$f)" >&2; done
( while true; do
    echo "$(date +%s.%N),$(cat $H/power1_average 2>/dev/null),$(cat $H/power1_input 2>/dev/null),$(cat $H/freq1_input 2>/dev/null)"
    sleep 0.05
  done ) > $OUT &
SAMPLER=$!
"$@"
RC=$?
kill $SAMPLER
python3 - $OUT <<'PY'
import sys
rows = [l.strip().split(',') for l in open(sys.argv[1]) if l.strip()]
def col(i):
    v = []
    for r in rows:
        try: v.append(float(r[i]))
        except Exception: pass
    return v
for name, i, scale in (("power1_average W", 1, 1e-6), ("power1_input W", 2, 1e-6), ("sclk MHz", 3, 1e-6)):
    v = sorted(col(i))
    if v: print(f"{name}: n {len(v)} min {v[0]*scale:.0f} median {v[len(v)//2]*scale:.0f} p90 {v[int(len(v)*0.9)]*scale:.0f} max {v[-1]*scale:.0f}")
PY
exit $RC
