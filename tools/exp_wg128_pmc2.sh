# L2 counters of the weight-gradient kernels in tools/exp_wgrad.py
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ctr in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
rm -rf /tmp/pp; timeout 300 rocprofv3 --pmc $ctr -d /tmp/pp -o r -- python3 tools/exp_wgrad.py > /dev/null 2>&1
DB=$(find /tmp/pp -name "*.db" | head -1); python tools/pmc_summary.py $DB 2>/dev/null | grep "wgrad128\|wgrad_kernel<4\|^kernel " | cut -c1-200
done
