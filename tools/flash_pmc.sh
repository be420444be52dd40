# Builder tool: PMC passes around the single-utterance call (tools/b1_latency.py), summarised for the flow attention kernels.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/flash_pmc
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/p1 -- python3 tools/b1_latency.py 6 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python3 tools/b1_latency.py 6 > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS --output-format csv -d $O/p3 -- python3 tools/b1_latency.py 6 > $O/p3.log 2>&1
for p in p1 p2 p3; do python3 tools/pmc_kernel_summary.py $(ls $O/$p/*/*counter_collection.csv | head -1) ${1:-flash_x3}; done
