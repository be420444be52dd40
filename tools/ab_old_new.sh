# Builder tool (GPU box): same-box A/B of the committed HEAD (a git worktree built under build/ab_old) against the working tree: the conv_clx timeline of a
# few shapes and the bench, alternating.   bash tools/ab_old_new.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
for rep in 1 2; do
  for side in old new; do
    if [ $side = old ]; then D=$R/build/ab_old; else D=$R; fi
    ( cd $D && CLX_TL_SHAPES=${AB_SHAPES:-1,2,5} python3 tools/clx_timeline.py 0 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    print('$side', 'C%d k%d %s %.1f us clk %d loop %.1f epi %.1f inloop %.2f' % (d['C'], d['k'], d['kind'], d['ms_per_launch'] * 1e3, d['loop_clock_mhz'], d['loop_us'], d['epilogue_issue_us'], d['avg_workgroups_in_loop_per_cu']))
" )
    ( cd $D && python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$side bench', d['ms_per_step'], {k: v for k, v in d['roofline']['per_config_ms'].items() if 'clx' in k or 'respair' in k}, 'err', d.get('cpu_baseline'))
" )
  done
done
