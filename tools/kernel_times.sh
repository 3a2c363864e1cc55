#!/bin/bash
# usage: tools/kernel_times.sh <regex>   -- in-graph kernel durations (rocprofv3 --kernel-trace --stats of `bench.py --light`) of the kernels matching <regex>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk -o r -- python3 bench.py --steps 5 --warmup 2 --light > /tmp/kt.log 2>&1
DB=$(find /tmp/pk -name "*.db" | head -1); python tools/rocprof_summary.py $DB 11 | grep -E "$1" | cut -c1-130
tail -1 /tmp/kt.log | cut -c1-120
