"""Host-side cost of enqueueing one 15.4 MB pinned host -> device copy (non_blocking) and of one graph replay."""
import time, torch
n = 15_400_000
h = torch.empty(n, dtype=torch.uint8).pin_memory(); h.fill_(1)
d = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream()
torch.cuda.synchronize()
for tag, size in (("15.4 MB", n), ("1 MB", 1 << 20), ("64 KB", 1 << 16)):
    ts = []
    for i in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(s):
            d[:size].copy_(h[:size], non_blocking=True)
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    ts.sort()
    print(f"{tag}: call returns after median {1e6 * ts[10]:.0f} us (min {1e6 * ts[0]:.0f}, max {1e6 * ts[-1]:.0f})")
# back-to-back (queue not empty)
t0 = time.perf_counter()
with torch.cuda.stream(s):
    for i in range(20):
        d.copy_(h, non_blocking=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"20 back-to-back 15.4 MB copies: enqueue {1e3 * (t1 - t0):.2f} ms, complete {1e3 * (t2 - t0):.2f} ms")
