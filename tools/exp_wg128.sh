# isolated wgrad128 kernel time for debug masks (1 no gathers, 2 no LDS stores, 4 no compute, 8 no MFMA)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in "$@"; do
  export PCD_WG128_NB=$d
  rm -rf /tmp/pkp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pkp -o r -- python3 tools/exp_wgrad.py > /dev/null 2>&1
  DB=$(find /tmp/pkp -name "*.db" | head -1); echo "nb=$d $(python tools/rocprof_summary.py $DB | grep -i "wgrad128" | cut -c1-40)"
done
