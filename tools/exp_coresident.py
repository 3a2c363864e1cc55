"""Can a light kernel on another stream share the CUs with the persistent window conv, or does the conv wait for it?
A spin kernel (256 workgroups, `threads` threads, ~16 VGPRs, 300 us) on stream A; one window conv of level 1 / 2 / 3 (238 /
232 / 251 VGPRs per wave, 2 waves per SIMD) on stream B launched while it runs.  If the conv's workgroups can be placed beside
the spinning waves, B finishes in about the conv's own time; if they need the CUs to themselves, B takes the spin's 300 us.
usage: python tools/exp_coresident.py"""
import sys, ctypes, torch, numpy as np
sys.path.insert(0, '.')
from com_amd import ops, hotpath, _lib as L
from com_amd.utils import synth
lib = L.lib()
dev = torch.device("cuda")
B = 4
pts, offs = hotpath.collate_points([synth.synth_cloud(f, 64, 2500) for f in range(B)], dev)
res = ops.voxelize_hard(pts, offs, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000, feat_offset=1, num_features=5,
                        want_voxels=False, row_order="yxz", key_depth=41)
idx, rank, shape = res["coords"], res["rank"], [41, 1504, 1504]
A, Bs = torch.cuda.Stream(), torch.cuda.Stream()
TICKS = 30000      # 300 us of the 100 MHz clock
for lvl, ch in ((1, 16), (2, 32), (3, 64)):
    if lvl > 1:
        rbs = ops.rulebook_conv(idx, B, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1), want_pairs=False, order=ops.ROWS_YXZ, in_rank=rank)
        idx, rank, shape = rbs.out_indices, rbs.rank, rbs.out_shape
    n = idx.shape[0]
    rb = ops.rulebook_subm(idx, B, shape, rank=rank, want_pairs=False)
    w = (torch.randn(ch, 3, 3, 3, ch) / np.sqrt(27 * ch)).to(dev)
    pw = ops.pack_weight_window(w, 0)
    x = torch.randn(n, ch, device=dev).to(torch.bfloat16)
    conv = lambda: ops.subm_window(x, pw, None, rb, ch)
    for _ in range(3):
        conv()
    torch.cuda.synchronize()
    def timed(spin_shape):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        if spin_shape is not None:
            blocks, threads, lds, nv = spin_shape
            with torch.cuda.stream(A):
                L.check(lib.pcd_debug_spin_shape(blocks, threads, lds, nv, TICKS, ctypes.c_void_p(A.cuda_stream)), "spin")
        with torch.cuda.stream(Bs):
            if spin_shape is not None:
                L.check(lib.pcd_debug_spin_shape(1, 64, 0, 0, 2000, ctypes.c_void_p(Bs.cuda_stream)), "delay")   # 20 us: the spin is resident
            e0.record()
            conv()
            e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3
    alone = min(timed(None) for _ in range(3))
    line = f"level {lvl} ({ch} ch, {n} rows): alone {alone:6.1f} us"
    for shape_ in ((256, 256, 0, 0), (256, 256, 0, 24), (256, 256, 0, 40), (256, 256, 0, 56), (256, 256, 0, 72), (256, 256, 8192, 0), (256, 256, 16384, 0), (256, 256, 32768, 0)):
        t = min(timed(shape_) for _ in range(3))
        line += f" | {shape_[3] or 'few'} VGPRs{', %d K LDS' % (shape_[2] >> 10) if shape_[2] else ''}: {t:6.1f}"
    print(line)
