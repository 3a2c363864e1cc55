# does the HW-queue placement of the replayed graph's internal streams matter?  K dummy streams created before the capture
run() { python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --no-h2d --no-ragged --no-stage2 --no-full-model --no-fp8 --no-regime 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', r['ms_per_step'])"; }
for rep in 1 2; do
for k in 0 1 2 3 5; do PCD_GRAPH_STREAM_SHIFT=$k run shift$k; done
done
