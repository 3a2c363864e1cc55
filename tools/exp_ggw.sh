cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "sparse_conv or basic_block or epilogue or fused_reductions or strided_dgrad" 2>&1 | tail -2
python -m pytest tests/test_gpu_g7_backbone.py tests/test_gpu_static.py -q -x 2>&1 | tail -2
for m in 0 1 5; do echo "PCD_GGW=$m"; PCD_GGW=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step']); print({k['kernel']:(k['avg_launch_us'],k['frac']) for k in r['roofline']['kernels'] if 'gather_gemm' in k['kernel'] or 'ggw' in k['kernel']})"; done
