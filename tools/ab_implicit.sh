# A/B of the strided-rulebook forms in the default bench: compact tables on / off (PCD_COMPACT_TABLES), pair lists (PCD_IMPLICIT_PAIRS=0)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2 3; do
  for v in "1 1" "1 0" "0 0"; do
    set -- $v
    PCD_IMPLICIT_PAIRS=$1 PCD_COMPACT_TABLES=$2 timeout 300 python bench.py --light --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('implicit=$1 compact=$2', d['value'], d['ms_per_step'])"
  done
done
