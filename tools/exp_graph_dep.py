"""Experiment: how precisely does a replayed hipGraph honour a cross-stream event dependency?
side stream: 6 long kernels, event after the FIRST; main: waits on that event, then one short kernel.
Variant 'late': the main kernel is captured right after the first side kernel (before the other 5)."""
import sys, torch
variant = sys.argv[1] if len(sys.argv) > 1 else "early"
dev = "cuda"
a = torch.randn(4096, 4096, device=dev)
b = torch.zeros(1 << 20, device=dev)
side = torch.cuda.Stream()
def long_kernel():
    for _ in range(1):
        a.mul_(1.0001)       # ~64 MB r/w: tens of us
def short_kernel():
    b.add_(1.0)
cap = torch.cuda.Stream()
with torch.cuda.stream(cap):
    for _ in range(3):
        long_kernel(); short_kernel()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        short_kernel()                       # root
        side.wait_stream(cur)
        ev = torch.cuda.Event()
        with torch.cuda.stream(side):
            long_kernel()
            ev.record(side)
            if variant == "early":
                for _ in range(5):
                    long_kernel()
        if variant == "late":
            cur.wait_event(ev)
            short_kernel()                   # dependent captured BEFORE the rest of the side chain
            with torch.cuda.stream(side):
                for _ in range(5):
                    long_kernel()
        else:
            cur.wait_event(ev)
            short_kernel()
        cur.wait_stream(side)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
print("done", variant)
