"""Bucket-pruned FPS (pointnet2.hip: fps_bucket_kernel) on the B = 4 x 160 k-point frames, 4096 samples each: time per call and the
kernel's own per-frame statistics (touched buckets per sample, rounds of 16 waves, shader clocks of its three phases).
usage: python tools/exp_fps.py"""
import sys, torch
sys.path.insert(0, '.')
from com_amd import _lib as L, pointnet2_stack as P
sys.path.insert(0, 'tools')
import env_switches
env_switches.apply()          # PCD_OPT_FPS_G=1000 + ablation bits (1: no loads, 2: no stores, 4: no bucket reduction)
from com_amd.utils import synth
import numpy as np
B = 4
frames = [synth.synth_cloud(f)[:, :3].astype(np.float32) for f in range(B)]
R = synth.WAYMO_RANGE
frames = [f[(f[:, 0] >= R[0]) & (f[:, 0] <= R[3]) & (f[:, 1] >= R[1]) & (f[:, 1] <= R[4])] for f in frames]
cnt = [f.shape[0] for f in frames]
xyz = torch.from_numpy(np.concatenate(frames, 0)).cuda()
c = torch.tensor(cnt, dtype=torch.int32).cuda()
npnt = torch.tensor([4096] * B, dtype=torch.int32).cuda()
lib = L.lib()
total = xyz.shape[0]
ws = torch.zeros((int(lib.pcd_stack_fps_buckets_workspace_bytes(B, total)),), dtype=torch.uint8, device="cuda")
out = torch.empty((4096 * B,), dtype=torch.int32, device="cuda")
def run():
    L.check(lib.pcd_stack_farthest_point_sampling_buckets(B, L.ptr(xyz), L.ptr(c), L.ptr(out), L.ptr(npnt), total, max(cnt), L.ptr(ws),
                                                          ws.numel(), L.stream_ptr()), "fps")
for mode in ("buckets", "coop"):
    P.FPS_LARGE = mode
    f = (lambda: P.stack_farthest_point_sample(xyz, c, [4096] * B)) if mode == "coop" else run
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); f(); f(); e1.record(); torch.cuda.synchronize()
    print(f"{mode}: {e0.elapsed_time(e1) / 3:.2f} ms per call ({B} frames x 4096 samples)")
# statistics live at the tail of the workspace: B x 8 int64 (the last carved piece)
st = ws[-((B * 64 + 255) // 256 * 256):][:B * 64].view(torch.int64).view(B, 8).cpu().numpy()
for b in range(B):
    d, r, t1, t2, t3, nb, ch, m = [int(v) for v in st[b]]
    it = max(m - 1, 1)
    print(f"frame {b}: {cnt[b]} points in {nb} buckets of {ch}; per sample: {d / it:.1f} buckets touched, {r / it:.2f} rounds; "
          f"clocks per sample: test+compact {t1 / it:.0f}, update {t2 / it:.0f}, argmax {t3 / it:.0f}")
