# HBM-resident vs H2D-inclusive step time
for cfg in "-" "PCD_H2D_STREAM_WAIT=1" "-" "PCD_H2D_STREAM_WAIT=1"; do
  echo "== $cfg"
  ( if [ "$cfg" != "-" ]; then export $cfg; fi
    python bench.py --no-cpu-baseline --no-roofline --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('resident', d['ms_per_step'], 'h2d', d['h2d_inclusive']['ms_per_step'])" )
done
