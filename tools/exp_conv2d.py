"""Dense 3x3 conv (com_amd.ops.conv2d_3x3_nhwc) vs torch / MIOpen on the BaseBEVBackbone shapes: error and time."""
import sys, torch
sys.path.insert(0, '.')
from com_amd import ops
dev = 'cuda'
def t(fn, reps=20):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
torch.manual_seed(0)
for (B, H, W, cin, cout) in ((4, 188, 188, 128, 128), (4, 188, 188, 256, 128), (4, 94, 94, 256, 256), (4, 188, 188, 512, 64), (4, 188, 188, 64, 64)):
    x = torch.randn(B, cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5)
    bias = torch.randn(cout, device=dev)
    xn = x.permute(0, 2, 3, 1).contiguous()
    pw = ops.conv2d_pack_weight(w, 0)
    y = ops.conv2d_3x3_nhwc(xn, pw, cout, bias)
    ref = torch.nn.functional.conv2d(x.float(), w.bfloat16().float(), bias, padding=1).permute(0, 2, 3, 1)
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    # data gradient through the same kernel
    dy = torch.randn(B, H, W, cout, device=dev).bfloat16()
    pd = ops.conv2d_pack_weight(w, 1)
    dx = ops.conv2d_3x3_nhwc(dy, pd, cin)
    dref = torch.nn.grad.conv2d_input((B, cin, H, W), w.bfloat16().float(), dy.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    derr = float((dx.float() - dref).abs().max() / dref.abs().max())
    wb = w.bfloat16().contiguous(memory_format=torch.channels_last)
    bb = bias.bfloat16()
    t_mi = t(lambda: torch.nn.functional.conv2d(x, wb, bb, padding=1))
    t_us = t(lambda: ops.conv2d_3x3_nhwc(xn, pw, cout, bias))
    fl = 2 * 9 * B * H * W * cin * cout
    print(f"{H}x{W} {cin}->{cout}: err {err:.2e} dgrad err {derr:.2e}  MIOpen {t_mi:.1f} us ({fl/t_mi*1e-6:.0f} TF)  ours {t_us:.1f} us ({fl/t_us*1e-6:.0f} TF)", flush=True)
