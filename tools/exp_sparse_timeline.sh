# per-kernel timeline of ONE replayed step of the hot-path bench (rocprofv3 changes the scheduling: use for WHAT runs, not when)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pst; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pst -o r -- python3 bench.py --steps 5 --warmup 2 --light > gpurun_out/st_bench.log 2>&1
DB=$(find /tmp/pst -name "*.db" | head -1)
python tools/timeline.py $DB gpurun_out/sparse_seq.txt | head -6
grep -n "rocclr\|FillFunctor\|multi_tensor\|elementwise" gpurun_out/sparse_seq.txt | cut -c1-150 | head -60
