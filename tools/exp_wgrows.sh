# A/B of the sparse weight-gradient granularity in the hot-path step: PCD_WG_ROWS (rows per split), PCD_WG128_NB (chunks)
run() { python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --no-h2d --no-ragged --no-stage2 --no-full-model --no-fp8 --no-regime 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', r['ms_per_step'])"; }
for rep in 1 2; do
run base
PCD_WG_ROWS=12288 run rows12288
PCD_WG_ROWS=24576 run rows24576
PCD_WG_ROWS=49152 run rows49152
PCD_WG128_NB=256 run nb256
PCD_WG128_NB=128 run nb128
done
