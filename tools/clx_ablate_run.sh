# Builder tool (GPU box): the conv_clx timeline of the product build and of the ablation builds under build/abl_*/ (tools/clx_ablate.sh), alternating.
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for D in $R $R/build/abl_*; do
    ( cd $D && CLX_TL_SHAPES=${AB_SHAPES:-1,2} CLX_TL_KINDS=${AB_KINDS:-1} python3 tools/clx_timeline.py 0 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    print('$(basename $D)', 'C%d k%d %s %.1f us clk %d loop %.1f us = %.1fk cycles epi %.1f inloop %.2f' % (d['C'], d['k'], d['kind'], d['ms_per_launch'] * 1e3, d['loop_clock_mhz'], d['loop_us'], d['loop_us'] * d['loop_clock_mhz'] / 1e3, d['epilogue_issue_us'], d['avg_workgroups_in_loop_per_cu']))
" )
  done
done
