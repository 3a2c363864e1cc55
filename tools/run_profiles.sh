# Produces the round's profile artefacts under gpurun_out/ (copy the ones to keep into profiles/).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r01_v5}
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk -o r -- python3 bench.py --steps 5 --warmup 2 --light > gpurun_out/${TAG}_bench_graph.log 2>&1
DB=$(find /tmp/pk -name "*.db" | head -1); python tools/rocprof_summary.py $DB 11 > gpurun_out/${TAG}_kernel_stats_graph.txt
rm -rf /tmp/pe; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pe -o r -- python3 bench.py --mode eager --steps 5 --warmup 2 --light > gpurun_out/${TAG}_bench_eager.log 2>&1
DB=$(find /tmp/pe -name "*.db" | head -1); python tools/rocprof_summary.py $DB 7 > gpurun_out/${TAG}_kernel_stats_eager.txt
rm -rf /tmp/pf; timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o r -- python3 bench.py --mode eager --steps 2 --warmup 2 --light > /dev/null 2>&1
rm -rf /tmp/pw; timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o r -- python3 bench.py --mode eager --steps 2 --warmup 2 --light > /dev/null 2>&1
F=$(find /tmp/pf -name "*.db" | head -1); W=$(find /tmp/pw -name "*.db" | head -1)
python tools/pmc_summary.py $F > gpurun_out/${TAG}_pmc_fetch_size.txt; python tools/pmc_summary.py $W > gpurun_out/${TAG}_pmc_write_size.txt
python tools/pmc_traffic.py $F $W > gpurun_out/${TAG}_pmc_traffic.json
timeout 300 python tools/regime.py 4 32 > gpurun_out/${TAG}_regime.jsonl 2>&1
tail -1 gpurun_out/${TAG}_bench_graph.log | cut -c1-160
