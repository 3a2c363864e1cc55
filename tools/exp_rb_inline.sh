# A/B: level-1 rulebook on the main stream (PCD_RB_INLINE0=1) vs on the prefetch stream (0); sparse-only step
run() { python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --no-h2d --no-ragged --no-stage2 --no-full-model --no-fp8 --no-regime 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', r['ms_per_step'])"; }
for rep in 1 2 3; do
PCD_RB_INLINE0=0 run side
PCD_RB_INLINE0=1 run inline
done
PCD_RB_INLINE0=1 PCD_STAMPS=1 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d --no-ragged --no-stage2 --no-full-model --no-fp8 --no-regime 2>&1 | grep stamps | cut -c1-600
