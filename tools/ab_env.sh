# A/B of environment switches in the default bench: bash tools/ab_env.sh "VAR=a" "VAR=b" ...   (each spec: space-separated VAR=value list, "-" = none)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for spec in "$@"; do
    ( [ "$spec" != "-" ] && export $spec; timeout 300 python bench.py --light --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', d['value'], d['ms_per_step'])" )
  done
done
