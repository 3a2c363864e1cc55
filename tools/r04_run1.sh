# round 4, run 1: row-order parity + bench at both row orders + kernel stats of the yxz order
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_g7_backbone.py tests/test_gpu_static.py -x -q -m gpu > gpurun_out/r04_run1_tests.log 2>&1
tail -5 gpurun_out/r04_run1_tests.log
LIGHT="--no-cpu-baseline --no-roofline --no-h2d --no-ragged --no-stage2 --no-full-model --no-fp8 --no-regime"
for ord in key yxz key yxz; do
  PCD_ROW_ORDER=$ord timeout 300 python bench.py --steps 100 --warmup 10 $LIGHT 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/$ord: /" | tee -a gpurun_out/r04_run1_bench.log
done
for ord in key yxz; do
  rm -rf /tmp/pk; PCD_ROW_ORDER=$ord timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk -o r -- python3 bench.py --steps 5 --warmup 2 $LIGHT > /dev/null 2>&1
  DB=$(find /tmp/pk -name "*.db" | head -1); python tools/rocprof_summary.py $DB 11 > gpurun_out/r04_run1_kernel_stats_graph_$ord.txt
done
head -30 gpurun_out/r04_run1_kernel_stats_graph_yxz.txt | cut -c1-150
