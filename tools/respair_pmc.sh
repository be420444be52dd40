export TMPDIR=/tmp
V=${1:-3}
O=$GRAFT_REPO_ROOT/gpurun_out/r04_rp_pmc_v$V
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/p1 -- python3 tools/respair_pmc.py $V > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python3 tools/respair_pmc.py $V > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $O/p3 -- python3 tools/respair_pmc.py $V > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p4 -- python3 tools/respair_pmc.py $V > $O/p4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p5 -- python3 tools/respair_pmc.py $V > $O/p5.log 2>&1
