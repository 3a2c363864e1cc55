# usage: exp_timeline.sh  -> gpurun_out/timeline.txt (one graph-replayed step: queue, start us, dur us, kernel)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/ptl; timeout 300 rocprofv3 --kernel-trace -d /tmp/ptl -o r -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d > /dev/null 2>&1
DB=$(find /tmp/ptl -name "*.db" | head -1); mkdir -p gpurun_out; python tools/timeline.py $DB gpurun_out/timeline.txt
