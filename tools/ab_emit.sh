# A/B of the strided builds' emit pass (option cm_emit_coop = 0 / 1 / 2): step time at B = 4 and the as-built rulebook chain at B = 4 / 32
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in 0 1 2; do
    PCD_OPT_CM_EMIT_COOP=$v timeout 300 python bench.py --light --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('coop=$v', d['value'], d['ms_per_step'])"
  done
done
for v in 0 1 2; do
PCD_OPT_CM_EMIT_COOP=$v PCD_REGIME_SKIP_PLAIN=1 timeout 300 python tools/regime.py 4 32 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); print('coop=$v', d['frames'], [(b['kind'],b['level'],b.get('as_built_us')) for b in d['builds'] if b['kind'].startswith('strided')])
"
done
