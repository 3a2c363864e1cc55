# A/B of PCD_CONV2D_WG_BLOCKS (workgroups per dense weight-gradient launch) inside the full CenterPoint step
for nb in 64 96 128 192 256; do
  for rep in 1 2; do
    PCD_CONV2D_WG_BLOCKS=$nb python bench.py --dense-head --com --steps 60 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d --no-ragged --no-stage2 --no-full-model --no-fp8 --no-regime 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('blocks', $nb, 'ms_per_step', r['ms_per_step'])"
  done
done
