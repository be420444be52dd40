export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/final_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/final_stats.log 2>&1
ls $O/final_stats/*/ | head
