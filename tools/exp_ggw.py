"""ggw_kernel variants at level 4 of the B = 4 batch (128 -> 128 SubM), forward and data gradient, isolated launches
(HIP events, 50 launches): option ggw_cw (consumer waves per SIMD), ggw_mi (rows per wave / 16), ggw_dbg ablations.
usage: python tools/exp_ggw.py"""
import sys, torch
sys.path.insert(0, '.')
from com_amd import ops, hotpath, _lib as L
from com_amd.utils import synth
dev = torch.device("cuda")
import os
B = int(os.environ.get('B', 4))
pts, offs = hotpath.collate_points([synth.synth_cloud(f) for f in range(B)], dev)
res = ops.voxelize_hard(pts, offs, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000, feat_offset=1, num_features=5,
                        want_voxels=False, row_order="yxz", key_depth=41)
idx, rank, shape = res["coords"], res["rank"], [41, 1504, 1504]
for geo in ((3, 2, 1), (3, 2, 1), (3, 2, (0, 1, 1))):
    rb = ops.rulebook_conv(idx, B, shape, geo[0], geo[1], geo[2], order=ops.ROWS_YXZ, in_rank=rank, want_pairs=False)
    idx, rank, shape = rb.out_indices, rb.rank, rb.out_shape
rb = ops.rulebook_subm(idx, B, shape, rank=rank, want_pairs=False)
n, ch = idx.shape[0], 128
x = torch.randn(n, ch, device=dev).to(torch.bfloat16)
w = torch.randn(ch, 3, 3, 3, ch, device=dev) * 0.02
print(f"level 4: {n} rows, {int((rb.nbr_out >= 0).sum())} pairs")
pk = ops.pack_weight(w, 0)
ref = None
for cw, mi, dbg in ((1, 3, 0), (2, 3, 0), (1, 2, 0), (1, 2, 32), (1, 2, 36), (1, 2, 4), (1, 2, 3)):
    L.set_option("ggw_cw", cw); L.set_option("ggw_mi", mi); L.set_option("ggw_dbg", dbg)
    f = lambda: ops.gather_gemm(x, pk, None, rb.nbr_out, 27, False, n, ch, torch.bfloat16)
    for _ in range(5):
        y = f()
    torch.cuda.synchronize()
    if dbg == 0:
        if ref is None:
            ref = y
        else:
            assert torch.equal(ref, y), "variants differ"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record(); torch.cuda.synchronize()
    print(f"cw={cw} mi={mi} dbg={dbg} (1 no gathers, 2 no weights, 4 no MFMAs, 8 no stages, 16 no rulebook tile, 32 no DMA instructions, 64 loader priority off): {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per launch (incl. ~6 us of launch)")
L.set_option("ggw_dbg", 0)
