# per-kernel durations of the voxeliser chain (eager launches under rocprofv3)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pe; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pe -o r -- python3 bench.py --mode eager --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-h2d > gpurun_out/sv_eager.log 2>&1
DB=$(find /tmp/pe -name "*.db" | head -1); python tools/rocprof_summary.py $DB 7 > gpurun_out/sv_kernel_stats_eager.txt
grep -E "vox_|scan_|pcd_fill|os16|adam|sumsq|pack_weights|fillBuffer|copyBuffer" gpurun_out/sv_kernel_stats_eager.txt | cut -c1-130
