cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_g7_backbone.py tests/test_gpu_dense2d.py tests/test_gpu_bench_forms.py tests/test_gpu_train_converges.py -x -q -m gpu > gpurun_out/r04_run3_tests.log 2>&1
tail -4 gpurun_out/r04_run3_tests.log
for i in 1 2; do
  timeout 300 python bench.py --steps 100 --warmup 10 --light 2>/dev/null | tail -1 | cut -c1-200 | tee -a gpurun_out/r04_run3_bench.log
done
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk -o r -- python3 bench.py --steps 5 --warmup 2 --light > /dev/null 2>&1
DB=$(find /tmp/pk -name "*.db" | head -1); python tools/rocprof_summary.py $DB 11 > gpurun_out/r04_run3_kernel_stats_graph.txt
grep -E "bn_|total kernel" gpurun_out/r04_run3_kernel_stats_graph.txt | cut -c1-120
for ord in key yxz; do PCD_ROW_ORDER=$ord timeout 300 python tools/regime.py 4 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$ord', d['rulebook_chain_graph_us'], {b['kind']+'_L'+str(b['level']): b['graph_us'] for b in d['builds']})"; done
