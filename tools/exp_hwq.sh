# does the number of HW queues the HIP runtime may use change the replayed step?
run() { python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --no-h2d --no-ragged --no-stage2 --no-full-model --no-fp8 --no-regime $2 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', r['ms_per_step'])"; }
for rep in 1 2; do
run default
GPU_MAX_HW_QUEUES=2 run q2
GPU_MAX_HW_QUEUES=8 run q8
GPU_MAX_HW_QUEUES=16 run q16
done
GPU_MAX_HW_QUEUES=8 run q8_full "--dense-head --com"
run default_full "--dense-head --com"
