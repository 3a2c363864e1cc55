cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2 3; do
  for v in 1 0; do
    PCD_IMPLICIT_PAIRS=$v timeout 300 python bench.py --light --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('implicit=$v', d['value'], d['ms_per_step'])"
  done
done
