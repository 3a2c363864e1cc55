cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_stage2.py tests/test_pointnet2_stack.py -x -q -m gpu > gpurun_out/r04_run4_tests.log 2>&1
tail -4 gpurun_out/r04_run4_tests.log
bash tools/run_profiles.sh r04_v2 > gpurun_out/r04_v2_profiles.log 2>&1
bash tools/run_sq_profile.sh r04_v2 > /dev/null 2>&1
grep -E "subm_win|ggw_kernel|gather_gemm_kernel<4|total kernel" gpurun_out/r04_v2_kernel_stats_graph.txt | cut -c1-130
grep -E "subm_win|ggw_kernel" gpurun_out/r04_v2_pmc_sq.txt | cut -c1-220 | head -4
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_v2_pmc_traffic.json'))
for k,v in list(d.get('kernels',d).items())[:0]: pass
print([k for k in (d.get('kernels') or d)][:12])
PY
