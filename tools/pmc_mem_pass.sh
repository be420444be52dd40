# Builder tool (GPU box): extra PMC passes of the bench command for the memory-side question "why do the HBM-bound launches stop where they do":
# L2 (TCC) hits / misses, the fabric's (EA) read / write requests and its DRAM credit stalls, and the waves' wait reasons.  One or two counters of a block per
# pass (five TCC counters in one pass abort rocprofv3: "exceeds the capabilities of the hardware" - and then it hangs: every pass runs under timeout).
# Outputs: gpurun_out/mem_pmc_<pass>/; tools/pmc_mem_summarise.py makes profiles/<tag>_pmc_memory_side.csv of them.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
pass() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/mem_pmc_$n -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/mem_pmc_$n.log 2>&1; echo "pass $n rc $?"; find $O/mem_pmc_$n -name "*kernel_trace.csv" -delete; }
pass hit TCC_HIT_sum TCC_MISS_sum
pass ea TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
pass stall TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum
pass sq SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
