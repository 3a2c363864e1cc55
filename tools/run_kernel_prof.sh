# usage: run_kernel_prof.sh <script.py> <grep-pattern> [env assignments...]   -> per-kernel avg durations (rocprofv3)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SCRIPT=$1; PAT=$2; shift 2
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/pkp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pkp -o r -- python3 $SCRIPT > /dev/null 2>&1
DB=$(find /tmp/pkp -name "*.db" | head -1); python tools/rocprof_summary.py $DB | grep -i "$PAT" | cut -c1-140
