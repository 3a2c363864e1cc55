cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=${1:-32}
rm -rf /tmp/pr; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr -o r -- python3 tools/regime.py $B > /dev/null 2>&1
DB=$(find /tmp/pr -name "*.db" | head -1); python tools/rocprof_summary.py $DB | head -30 | cut -c1-150
