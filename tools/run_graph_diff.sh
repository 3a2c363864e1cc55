# per-replay kernel statistics of the captured step -> gpurun_out/${TAG}_kernel_stats_replay.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
rm -rf /tmp/pa /tmp/pb
timeout 300 rocprofv3 --kernel-trace -d /tmp/pa -o r -- python3 bench.py --steps 5 --warmup 2 --light > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace -d /tmp/pb -o r -- python3 bench.py --steps 45 --warmup 2 --light > gpurun_out/${TAG}_bench_graph.log 2>&1
A=$(find /tmp/pa -name "*.db" | head -1); B=$(find /tmp/pb -name "*.db" | head -1)
python tools/rocprof_diff.py $A $B 40 > gpurun_out/${TAG}_kernel_stats_replay.txt
head -60 gpurun_out/${TAG}_kernel_stats_replay.txt | cut -c1-170
