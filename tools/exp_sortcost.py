"""How long does a device radix sort of the level-1 voxel keys take (torch.sort = rocPRIM underneath)?"""
import torch
dev = 'cuda'
n = 426000
key = torch.randint(0, 1 << 29, (n,), device=dev, dtype=torch.int32)
for dt in (torch.int32, torch.int64):
    k = key.to(dt)
    for _ in range(3): torch.sort(k)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): torch.sort(k)
    e1.record(); torch.cuda.synchronize()
    print(dt, e0.elapsed_time(e1) / 50 * 1e3, "us per sort (eager launches)")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.sort(k)
        with torch.cuda.graph(g):
            for _ in range(10): out = torch.sort(k)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(dt, e0.elapsed_time(e1) / 100 * 1e3, "us per sort (graph)")
