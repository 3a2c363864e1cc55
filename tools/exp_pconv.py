"""pconv_kernel (pair-driven strided conv) against the gather kernels on the level-1 -> 2 conv of the B = 4 batch: forward
16 -> 32 (gather_gemm_kernel<2,2,1,0>) and data gradient 32 -> 16 (gather_gemm_cls_kernel<2,2>), isolated launches.
usage: make -C com_amd/csrc EXPERIMENTS=1 && python tools/exp_pconv.py"""
import sys, torch
sys.path.insert(0, '.')
from com_amd import ops, hotpath
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
rb = ops.rulebook_conv(idx, B, shape, 3, 2, 1, order=ops.ROWS_YXZ, in_rank=rank)
n_in, n_out = idx.shape[0], rb.n_out
print(f"level 1 -> 2: {n_in} -> {n_out} rows, {int(rb.pair_num.sum())} pairs")
w = torch.randn(32, 3, 3, 3, 16, device=dev) * 0.05
x = torch.randn(n_in, 16, device=dev).to(torch.bfloat16)
dy = torch.randn(n_out, 32, device=dev).to(torch.bfloat16)
pf, pd = ops.pack_weight(w, 0), ops.pack_weight(w, 1)
ops.pair_conv_plan(rb, 0); ops.pair_conv_plan(rb, 1)


def timeit(f, reps=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"forward  pconv_kernel<16,32>         {timeit(lambda: ops.pair_conv(x, pf, None, rb, 0, 32, torch.bfloat16)):7.1f} us")
print(f"forward  gather_gemm_kernel<2,2,1,0> {timeit(lambda: ops.gather_gemm(x, pf, None, rb.nbr_out, 27, False, n_out, 32, torch.bfloat16)):7.1f} us")
print(f"dgrad    pconv_kernel<32,16>         {timeit(lambda: ops.pair_conv(dy, pd, None, rb, 1, 16, torch.bfloat16)):7.1f} us")
print(f"dgrad    gather_gemm_cls_kernel<2,2> {timeit(lambda: ops.dgrad_classes(dy, pd, rb, 16, torch.bfloat16)):7.1f} us")
print("(per launch, incl. ~6 us of launch overhead)")
