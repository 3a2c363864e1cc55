cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
LIGHT="--no-cpu-baseline --no-roofline --no-h2d --no-ragged --no-stage2 --no-full-model --no-fp8 --no-regime"
for ord in ${1:-yxz}; do
PCD_STAMPS=1 PCD_ROW_ORDER=$ord timeout 300 python bench.py --steps 40 --warmup 10 $LIGHT 2>&1 | grep -E "stamps|frames" | cut -c1-900
done
