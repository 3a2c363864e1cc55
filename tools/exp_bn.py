"""Micro-benchmark: fused BatchNorm forward / backward / col_sum at the four backbone levels
(run under tools/run_kernel_prof.sh for per-kernel durations: the Python call overhead exceeds these kernels)."""
import sys, torch
sys.path.insert(0, '.')
from com_amd import ops
dev = 'cuda'
def t(fn, reps=50):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n, c in ((337758, 16), (295015, 32), (114977, 64), (42437, 128)):
    x = torch.randn(n, c, device=dev).bfloat16(); res = torch.randn(n, c, device=dev).bfloat16()
    dy = torch.randn(n, c, device=dev).bfloat16()
    g = torch.rand(c, device=dev) + 0.5; b = torch.randn(c, device=dev) * 0.1
    rm = torch.zeros(c, device=dev); rv = torch.ones(c, device=dev)
    y, sm, si = ops.bn_forward(x, None, g, b, 1e-3, 0.01, True, rm, rv, True)
    tf = t(lambda: ops.bn_forward(x, None, g, b, 1e-3, 0.01, True, rm, rv, True))
    tfr = t(lambda: ops.bn_forward(x, res, g, b, 1e-3, 0.01, True, rm, rv, True))
    tb = t(lambda: ops.bn_backward(dy, x, None, g, sm, si, True, True, False, beta=b))
    tby = t(lambda: ops.bn_backward(dy, x, y, g, sm, si, True, True, True))
    tc = t(lambda: ops.col_sum(dy))
    print(f"n={n} c={c} ({n * c * 2 / 1e6:.1f} MB/tensor): fwd {tf:.1f} us  fwd+res {tfr:.1f} us  bwd(mask from x) {tb:.1f} us  "
          f"bwd(y, dres) {tby:.1f} us  col_sum {tc:.1f} us", flush=True)
