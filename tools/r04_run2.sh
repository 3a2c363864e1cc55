cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_subm_window.py tests/test_gpu_g7_backbone.py tests/test_gpu_static.py tests/test_gpu_bench_forms.py -x -q -m gpu > gpurun_out/r04_run2_tests.log 2>&1
tail -8 gpurun_out/r04_run2_tests.log
LIGHT="--no-cpu-baseline --no-roofline --no-h2d --no-ragged --no-stage2 --no-full-model --no-fp8 --no-regime"
for ord in key yxz key yxz; do
  PCD_ROW_ORDER=$ord timeout 300 python bench.py --steps 100 --warmup 10 $LIGHT 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/$ord: /" | tee -a gpurun_out/r04_run2_bench.log
done
PCD_OPT_SUBM_WINDOW=3 PCD_ROW_ORDER=yxz timeout 300 python bench.py --steps 100 --warmup 10 $LIGHT 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/yxz win64+32: /" | tee -a gpurun_out/r04_run2_bench.log
for ord in yxz; do
  rm -rf /tmp/pk; PCD_ROW_ORDER=$ord timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk -o r -- python3 bench.py --steps 5 --warmup 2 $LIGHT > /dev/null 2>&1
  DB=$(find /tmp/pk -name "*.db" | head -1); python tools/rocprof_summary.py $DB 11 > gpurun_out/r04_run2_kernel_stats_graph_$ord.txt
done
head -24 gpurun_out/r04_run2_kernel_stats_graph_yxz.txt | cut -c1-150
