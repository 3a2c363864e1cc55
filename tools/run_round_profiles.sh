# Round profile artefacts of the CURRENT code under gpurun_out/ (copy the ones to keep into profiles/):
#   ${TAG}_kernel_stats_replay.txt  per-REPLAY kernel statistics of the captured step (difference of two traces)
#   ${TAG}_kernel_stats_eager.txt   rocprofv3 --kernel-trace --stats of the eager command (the bench line's frac_rocprof)
#   ${TAG}_pmc_{fetch,write}_size.txt + ${TAG}_pmc_traffic.json   HBM-side traffic per kernel (separate --pmc passes)
#   ${TAG}_pmc_sq.txt               SQ counters (MFMA busy, waits, LDS bank conflicts) per kernel
#   ${TAG}_regime.jsonl             rulebook chain at B = 4 / 32
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
bash tools/run_graph_diff.sh $TAG > /dev/null 2>&1
rm -rf /tmp/pe; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pe -o r -- python3 bench.py --mode eager --steps 5 --warmup 2 --light > gpurun_out/${TAG}_bench_eager.log 2>&1
DB=$(find /tmp/pe -name "*.db" | head -1); python tools/rocprof_summary.py $DB 7 > gpurun_out/${TAG}_kernel_stats_eager.txt
rm -rf /tmp/pf; timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o r -- python3 bench.py --mode eager --steps 2 --warmup 2 --light > /dev/null 2>&1
rm -rf /tmp/pw; timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o r -- python3 bench.py --mode eager --steps 2 --warmup 2 --light > /dev/null 2>&1
F=$(find /tmp/pf -name "*.db" | head -1); W=$(find /tmp/pw -name "*.db" | head -1)
python tools/pmc_summary.py $F > gpurun_out/${TAG}_pmc_fetch_size.txt; python tools/pmc_summary.py $W > gpurun_out/${TAG}_pmc_write_size.txt
python tools/pmc_traffic.py $F $W > gpurun_out/${TAG}_pmc_traffic.json
rm -rf /tmp/psq; timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d /tmp/psq -o r -- python3 bench.py --mode eager --steps 2 --warmup 2 --light > /dev/null 2>&1
S=$(find /tmp/psq -name "*.db" | head -1); python tools/pmc_summary.py $S > gpurun_out/${TAG}_pmc_sq.txt
timeout 300 python tools/regime.py 4 32 > gpurun_out/${TAG}_regime.jsonl 2>&1
head -4 gpurun_out/${TAG}_kernel_stats_replay.txt | cut -c1-120; head -14 gpurun_out/${TAG}_pmc_sq.txt | cut -c1-96,100-104,240-300
