# Re-collects the round's profile artifacts (run on the GPU box from the repo root): kernel stats of the bench command, and the PMC passes
# for HBM traffic and MFMA / LDS utilisation.  Outputs land in gpurun_out/; tools/prof_summarise.py turns them into profiles/<tag>_*.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
python3 bench.py --steps 8 --warmup 2 > $O/final_bench.json 2> $O/final_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/final_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/final_pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/final_pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/final_pmc_mfma -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
# the other BASELINE configs one GPU can run, as JSON lines (tools/prof_summarise.py copies them to profiles/<tag>_*.json)
python3 bench.py --config mixed256 --steps 3 --warmup 1 --no-cpu-baseline > $O/final_mixed256.json 2> $O/final_mixed256.err
python3 tools/b1_latency.py 40 2> $O/final_b1.err | tail -1 > $O/final_b1_latency.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_b1_stats -- python3 tools/b1_latency.py 40 > $O/final_b1_stats.log 2>&1
python3 tools/dropin_latency.py 60 2> $O/final_dropin.err | tail -1 > $O/final_dropin_latency.json
python3 tools/dropin_latency.py 60 --fp32 2>> $O/final_dropin.err | tail -1 > $O/final_dropin_latency_fp32.json
rm -rf /tmp/kt25; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt25 -- python3 tools/dropin_latency.py 40 --tokens=25 > /dev/null 2>&1
python3 tools/trace_gaps.py /tmp/kt25 300 > $O/final_dropin25_gaps.json
python3 tools/long_form_check.py 2000 256 2> $O/final_longform.err | tail -1 > $O/final_longform_c256.json

python3 tools/long_form_check.py 2000 1024 2>> $O/final_longform.err | tail -1 > $O/final_longform_c1024.json
tail -c 600 $O/final_bench.json
