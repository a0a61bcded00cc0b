# PMC passes over tools/time_conv_c16.py (the 16-channel 3x3x3 forward / dgrad kernels at 2 x 128^3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
O=gpurun_out/pmc_conv
rm -rf $O && mkdir -p $O
pass() {
  n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d $O/$n -o $n --output-format csv -- python3 tools/time_conv_c16.py > $O/$n.log 2>&1 || { echo pass $n failed; tail -5 $O/$n.log; exit 1; }
  python3 tools/pmc_summary.py $O/$n k_conv3_c16 > gpurun_out/conv_pmc_$n.csv || exit 1
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY &&
pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES &&
pass sq3 SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_MISC
