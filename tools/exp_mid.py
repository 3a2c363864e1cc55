"""Cost of the BatchNorm finalisation chain (bn_mid_kernel + apply prologue) vs the number of partial rows:
50 pcd_bn_forward calls on a tiny tensor with external partials, captured in one hipGraph."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from com_amd import ops  # noqa: E402


def run(rows, c, n=512, reps=50):
    x = torch.randn(n, c, device="cuda").bfloat16()
    part = torch.randn(rows, 2, c, device="cuda").abs()
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    def body():
        for _ in range(reps):
            ops.bn_forward(x, None, g, b, 1e-3, 0.01, True, rm, rv, True, partials=(part, rows) if rows else None)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            body()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"rows {rows:5d} c {c:4d}: {e0.elapsed_time(e1) / 5 / reps * 1e3:6.2f} us per bn_forward")


if __name__ == "__main__":
    for c in (16, 128):
        for rows in (0, 16, 64, 512, 2304, 5280):
            run(rows, c)
