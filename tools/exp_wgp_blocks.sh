# A/B inside the full step: plane weight-gradient kernels (modes 4 / 6) at different workgroup counts vs the pair kernels
run() { python bench.py --dense-head --com --steps 80 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d --no-ragged --no-stage2 --no-full-model --no-fp8 --no-regime 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', r['ms_per_step'])"; }
for rep in 1 2; do
PCD_CONV2D_WGP_BLOCKS=1 run pairs_only_equiv
PCD_CONV2D_WGP_BLOCKS=128 run wgp128
PCD_CONV2D_WGP_BLOCKS=256 run wgp256
PCD_CONV2D_WGP_BLOCKS=512 run wgp512
PCD_CONV2D_WGP_BLOCKS=512 PCD_CONV2D_WGP_MODE2=1 run wgp512_mode2
done
