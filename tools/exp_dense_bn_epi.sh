# A/B: BatchNorm sums in the dense convs' epilogues (PCD_DENSE_BN_EPI bit 0 forward / bit 1 backward), mid folding
run() { python bench.py --dense-head --com --steps 80 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d --no-ragged --no-stage2 --no-full-model --no-fp8 --no-regime 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', r['ms_per_step'])"; }
for rep in 1 2; do
PCD_DENSE_BN_EPI=0 run epi0
PCD_DENSE_BN_EPI=1 run epi1
PCD_DENSE_BN_EPI=2 run epi2
PCD_DENSE_BN_EPI=3 run epi3
PCD_DENSE_BN_EPI=3 PCD_BN_FUSED_MID=0 run epi3_nomid
done
