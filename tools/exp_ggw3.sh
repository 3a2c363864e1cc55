cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ctr in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
rm -rf /tmp/pp; PCD_GGW=3 timeout 300 rocprofv3 --pmc $ctr -d /tmp/pp -o r -- python3 tools/exp_l4.py 128 > /dev/null 2>&1
DB=$(find /tmp/pp -name "*.db" | head -1); python tools/pmc_summary.py $DB 2>/dev/null | grep "ggw_kernel\|^kernel " | cut -c1-220
done
