# SQ counters of the weight-gradient kernels in tools/exp_wgrad.py (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ctr in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
rm -rf /tmp/pp; timeout 300 rocprofv3 --pmc $ctr -d /tmp/pp -o r -- python3 tools/exp_wgrad.py > /dev/null 2>&1
DB=$(find /tmp/pp -name "*.db" | head -1); python tools/pmc_summary.py $DB 2>/dev/null | grep "wgrad128\|wgrad_kernel<4\|^kernel " | cut -c1-200
done
