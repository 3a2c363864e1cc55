# SQ counters of the final code (own --pmc pass, eager, 2 steps; averages per launch) -> gpurun_out/<tag>_pmc_sq.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r02_v7}
rm -rf /tmp/psq; timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d /tmp/psq -o r -- python3 bench.py --mode eager --steps 2 --warmup 2 --light > /dev/null 2>&1
S=$(find /tmp/psq -name "*.db" | head -1)
python tools/pmc_summary.py $S > gpurun_out/${TAG}_pmc_sq.txt
head -30 gpurun_out/${TAG}_pmc_sq.txt | cut -c1-220
