"""ggwin_kernel (128 -> 128 SubM over z-fastest rows: x through windows, weights streamed) against ggw_kernel (27 gather
slots per row), level 4 of the B = 4 batch, forward and data gradient, isolated launches (HIP events, 50 launches).
usage: make -C com_amd/csrc EXPERIMENTS=1 && python tools/exp_ggwin.py"""
import sys, torch
sys.path.insert(0, '.')
from com_amd import ops, hotpath, _lib as L
from com_amd import _lib as _L
_L.use_experiments_library()          # (make -C com_amd/csrc EXPERIMENTS=1)
sys.path.insert(0, 'tools')
import env_switches
env_switches.apply()
from com_amd.utils import synth
dev = torch.device("cuda")
B = 4
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
for mode in (0, 1):
    pk = ops.pack_weight(w, mode)
    for opt in (1, 0):
        L.set_option("ggwin", opt)
        f = lambda: ops.gather_gemm(x, pk, None, rb.nbr_out, 27, bool(mode), n, ch, torch.bfloat16, zfast=True)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record(); torch.cuda.synchronize()
        print(f"{'dgrad' if mode else 'forward'} {'ggwin_kernel' if opt else 'ggw_kernel  '}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per launch (incl. ~6 us of launch)")
