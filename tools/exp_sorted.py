"""Experiment: level-1 SubM conv (16->16) + wgrad on first-appearance (random) vs key-sorted row order."""
import sys, torch
sys.path.insert(0, '.')
from com_amd import ops, hotpath
from com_amd.utils import synth
dev = 'cuda'
frames = [synth.synth_cloud(f) for f in range(4)]
pts, offs = hotpath.collate_points(frames, dev)
res = ops.voxelize_hard(pts, offs, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000, feat_offset=1, num_features=5, want_voxels=False)
idx = res['coords']; n = idx.shape[0]
shape = [41, 1504, 1504]
key = ((idx[:, 0].long() * 41 + idx[:, 1]) * 1504 + idx[:, 2]) * 1504 + idx[:, 3]
perm = torch.argsort(key)
def bench(ix, tag):
    rb = ops.rulebook_subm(ix, 4, shape)
    for cin, cout in ((16, 16), (32, 32)):
        x = torch.randn(n, cin, device=dev).bfloat16(); dy = torch.randn(n, cout, device=dev).bfloat16()
        w = torch.randn(cout, 27, cin, device=dev) * 0.1
        pw = ops.pack_weight(w, 0)
        def t(fn, reps=20):
            for _ in range(3): fn()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        tf = t(lambda: ops.gather_gemm(x, pw, None, rb.nbr_out, 27, False, n, cout, torch.bfloat16))
        tw = t(lambda: ops.wgrad(x, cin, dy, rb.pairs, rb.pair_num, 27))
        tr = t(lambda: ops.rulebook_subm(ix, 4, shape))
        print(f"{tag} {cin}->{cout}: gather_gemm {tf:.1f} us  wgrad(+reduce) {tw:.1f} us  rulebook {tr:.1f} us", flush=True)
bench(idx, 'first-appearance')
bench(idx[perm].contiguous(), 'key-sorted      ')
