"""Experiment: split the kernel offsets of a 128 -> 128 (or 64 -> 64) SubM gather-GEMM over S concurrent launches
(fp32 partial outputs, summed and rounded by one elementwise kernel) vs the single launch: is the level-4 layer
(42 k rows at B = 4: 332 workgroups, 1.3 per CU) short of parallelism?  argv[1] = channels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from com_amd import hotpath, ops  # noqa: E402
from com_amd.utils import synth  # noqa: E402

dev = 'cuda'
CH = int(sys.argv[1]) if len(sys.argv) > 1 else 128
frames = [synth.synth_cloud(f) for f in range(4)]
pts, offs = hotpath.collate_points(frames, dev)
res = ops.voxelize_hard(pts, offs, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000, feat_offset=1, num_features=5,
                        want_voxels=False)
idx, shape = res['coords'], [41, 1504, 1504]
for geo in ((3, 2, 1), (3, 2, 1), (3, 2, (0, 1, 1)))[:3 if CH == 128 else 2]:
    rb = ops.rulebook_conv(idx, 4, shape, geo[0], geo[1], geo[2])
    idx, shape = rb.out_indices, rb.out_shape
n = idx.shape[0]
rb = ops.rulebook_subm(idx, 4, shape)
x = torch.randn(n, CH, device=dev).bfloat16()
w = torch.randn(CH, 27, CH, device=dev) * 0.05


def graph_time(body, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                body()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 / reps * 1e3


pw = ops.pack_weight(w, 0)
full = graph_time(lambda: ops.gather_gemm(x, pw, None, rb.nbr_out, 27, False, n, CH, torch.bfloat16))
print(f"{CH} ch, {n} rows: single launch {full:.1f} us")
for S in (2, 3):
    cuts = [round(27 * i / S) for i in range(S + 1)]
    packs = [ops.pack_weight(w[:, cuts[i]:cuts[i + 1]].contiguous(), 0) for i in range(S)]
    nbrs = [rb.nbr_out[cuts[i]:cuts[i + 1]].contiguous() for i in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S - 1)]

    def body():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        parts = [None] * S
        for i in range(1, S):
            streams[i - 1].wait_event(ev)
            with torch.cuda.stream(streams[i - 1]):
                parts[i] = ops.gather_gemm(x, packs[i], None, nbrs[i], cuts[i + 1] - cuts[i], False, n, CH,
                                           torch.float32)
        parts[0] = ops.gather_gemm(x, packs[0], None, nbrs[0], cuts[1] - cuts[0], False, n, CH, torch.float32)
        for st in streams:
            cur.wait_stream(st)
        acc = parts[0]
        for p in parts[1:]:
            acc = acc + p
        return acc.bfloat16()

    print(f"  split {S}: {graph_time(body):.1f} us (incl. fp32 partial sum + rounding by torch)")
    one = graph_time(lambda: ops.gather_gemm(x, packs[0], None, nbrs[0], cuts[1] - cuts[0], False, n, CH, torch.float32))
    print(f"  one part of {S} alone: {one:.1f} us")
