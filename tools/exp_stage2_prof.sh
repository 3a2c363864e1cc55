cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/ps2; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ps2 -o r -- python3 tools/exp_stage2_prof.py > gpurun_out/s2_prof.log 2>&1
DB=$(find /tmp/ps2 -name "*.db" | head -1); python tools/rocprof_summary.py $DB 8 | head -40
tail -2 gpurun_out/s2_prof.log | cut -c1-300
for g in 32 48 64; do PCD_FPS_G=$g python tools/exp_stage2_prof.py 2>/dev/null | tail -1 | cut -c1-120; done
