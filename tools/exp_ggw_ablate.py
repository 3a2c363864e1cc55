"""ggw_kernel (128 -> 128 SubM, level 4 of the B = 4 batch) with its ablation switches ("ggw_dbg": 1 no gather traffic,
2 no weight streaming, 4 no MFMAs): where the 50 us go."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from com_amd import ops, hotpath, _lib as L
sys.path.insert(0, 'tools')
import env_switches
env_switches.apply()          # PCD_OPT_* -> pcd_set_option
from com_amd.utils import synth

dev = torch.device("cuda")
B = 4
frames = [synth.synth_cloud(f) for f in range(B)]
pts, offs = hotpath.collate_points(frames, dev)
order = sys.argv[1] if len(sys.argv) > 1 else "yxz"
res = ops.voxelize_hard(pts, offs, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000, feat_offset=1, num_features=5,
                        want_voxels=False, row_order=order, key_depth=41)
idx, rank, shape = res["coords"], res["rank"], [41, 1504, 1504]
for k, s, p in [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))]:
    rbc = ops.rulebook_conv(idx, B, shape, k, s, p, want_pairs=False, order=ops.ROW_ORDERS[order])
    idx, rank, shape = rbc.out_indices, rbc.rank, rbc.out_shape
n, ch = idx.shape[0], 128
rb = ops.rulebook_subm(idx, B, shape, rank=rank, want_pairs=False)
w = (torch.randn(ch, 3, 3, 3, ch) * 0.02).to(dev)
x = torch.randn(n, ch).to(dev).to(torch.bfloat16)
pk = ops.pack_weight(w, 0)
pairs = int((rb.nbr_out >= 0).sum().item())


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters * 1e3


print(f"rows {n}, pairs {pairs} ({pairs / n:.1f} per row), order {order}")
for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
    L.set_option("ggw_dbg", dbg)
    t = timeit(lambda: ops.gather_gemm(x, pk, None, rb.nbr_out, 27, False, n, ch, torch.bfloat16))
    print(f"ggw_dbg {dbg}: {t:.1f} us   ({2.0 * pairs * ch * ch / t * 1e-6:.0f} TFLOP/s algorithmic)")
L.set_option("ggw_dbg", 0)
