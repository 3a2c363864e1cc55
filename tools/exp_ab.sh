# usage: exp_ab.sh "<pytest -k expr or empty>" ENV1 ENV2 ...   (each ENV like "A=1,B=2" or "-" for none)
cd $GRAFT_REPO_ROOT
K="$1"; shift
if [ -n "$K" ]; then python -m pytest tests -m gpu -q -x -k "$K" 2>&1 | tail -3; fi
for e in "$@"; do
  echo "== $e"
  ( if [ "$e" != "-" ]; then IFS=, ; for kv in $e; do export "$kv"; done; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step']); print({k['kernel']:(k['avg_launch_us'],k['frac']) for k in r['roofline']['kernels'] if k['kernel'].startswith(('gather','wgrad','ggw'))})" )
done
