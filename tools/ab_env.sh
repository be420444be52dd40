# Builder tool (GPU box): alternating same-box A/B of bench.py under two environment settings.
#   bash tools/ab_env.sh "SBV2_UPX=2" "SBV2_UPX=1" [reps] [bucket substring]
A="$1"; B="$2"; REPS=${3:-3}; KEY=${4:-conv_clx}
run() { env $1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | KEY="$KEY" TAG="$1" python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read()); r=d['roofline']['per_config_ms']; print(os.environ['TAG'], d['ms_per_step'], d['value'], {k:v for k,v in r.items() if os.environ['KEY'] in k})"; }
for rep in $(seq $REPS); do run "$A"; run "$B"; done
