"""nn.Conv2d(k = 3, stride 1, padding 1) on channels-last bf16 maps through the hand-written implicit-GEMM kernel
(com_amd/csrc/conv2d.hip) instead of MIOpen: forward and data gradient are the same MFMA kernel (weights packed
forward / transposed + rotated), the weight gradient is the sparse path's pair-wise X^T dY kernel (spconv.hip) run over
the DENSE pair lists of the 9 taps -- built once per map shape.  `Conv3x3` is a drop-in nn.Conv2d subclass (same
parameters and state-dict keys, base_bev_backbone.py:37-51 / center_head.py:17-24 construct it with the same arguments);
anything the kernel does not cover (other kernel sizes / strides, CPU tensors, fp32 maps outside a bf16 autocast
region, cin % 32 != 0) takes
nn.Conv2d's own path."""
import torch
import torch.nn as nn

from .. import ops

ENABLED = True
_PAIRS = {}


def _dense_pairs(B, H, W, device):
    """indice_pairs of a dense 3x3 / stride 1 / padding 1 conv over [B, H, W] pixels (row = (b * H + y) * W + x):
    pairs [9, 2, n] (input row, output row; ascending), pair_num [9]."""
    key = (B, H, W, str(device))
    hit = _PAIRS.get(key)
    if hit is not None:
        return hit
    n = B * H * W
    p = torch.arange(n, device=device, dtype=torch.int64)
    y, x = (p // W) % H, p % W
    pairs = torch.full((9, 2, n), -1, dtype=torch.int32, device=device)
    num = torch.zeros((9,), dtype=torch.int32, device=device)
    for k in range(9):
        dy, dx = k // 3 - 1, k % 3 - 1
        ok = (y + dy >= 0) & (y + dy < H) & (x + dx >= 0) & (x + dx < W)
        out_rows = p[ok]
        m = out_rows.numel()
        pairs[k, 0, :m] = (out_rows + dy * W + dx).int()
        pairs[k, 1, :m] = out_rows.int()
        num[k] = m
    _PAIRS[key] = (pairs, num)
    return pairs, num


class _Conv3x3Function(torch.autograd.Function):
    """Output channel counts that are not a multiple of 32 (the 1 / 2 / 3-channel final convs of the head towers) run
    with the weights zero-padded to 32 channels: MIOpen spends 0.4 ms on each of those tiny convs."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        # x: [B, C, H, W] bf16, channels_last storage
        xn = x.detach().permute(0, 2, 3, 1)
        assert xn.is_contiguous()
        cout = weight.shape[0]
        cp = (cout + 31) // 32 * 32          # (the data gradient contracts over the output channels in steps of 32)
        w = weight.detach().float()
        b = bias.detach().float() if bias is not None else None
        if cp != cout:
            w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, cp - cout))
            b = torch.nn.functional.pad(b, (0, cp - cout)) if b is not None else None
        y = ops.conv2d_3x3_nhwc(xn, ops.conv2d_pack_weight(w, 0), cp, b)
        if cp != cout:
            y = y[..., :cout].contiguous()
        ctx.save_for_backward(xn, w)
        ctx.has_bias, ctx.cout, ctx.wdtype = bias is not None, cout, weight.dtype
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        xn, w = ctx.saved_tensors                      # w: f32, output channels padded to a multiple of 32
        B, H, W, cin = xn.shape
        cout, cp = ctx.cout, w.shape[0]
        dyn = dy.permute(0, 2, 3, 1)
        if dyn.dtype != torch.bfloat16:
            dyn = dyn.to(torch.bfloat16)
        if cp != cout:
            dyn = torch.nn.functional.pad(dyn, (0, cp - cout))
        dyn = dyn.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_3x3_nhwc(dyn, ops.conv2d_pack_weight(w, 1), cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            pairs, num = _dense_pairs(B, H, W, xn.device)
            dwk = ops.wgrad(xn.reshape(-1, cin), cin, dyn.reshape(-1, cp), pairs, num, 9)      # [cp, 9, cin] f32
            dw = dwk[:cout].permute(0, 2, 1).reshape(cout, cin, 3, 3).to(ctx.wdtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dyn.reshape(-1, cp)[:, :cout].float().sum(0)
        return dx, dw, db


class Conv3x3(nn.Conv2d):
    """nn.Conv2d whose forward takes the HIP kernel when it applies (see the module docstring)."""

    def _fast(self, x):
        bf16 = x.dtype == torch.bfloat16 or (torch.is_autocast_enabled()
                                             and torch.get_autocast_gpu_dtype() == torch.bfloat16)
        return (ENABLED and bf16 and x.is_cuda and x.dim() == 4 and self.kernel_size == (3, 3) and self.stride == (1, 1)
                and self.padding == (1, 1) and self.dilation == (1, 1) and self.groups == 1
                and self.padding_mode == 'zeros' and self.in_channels % 32 == 0
                and x.shape[0] * x.shape[2] * x.shape[3] * max(self.in_channels, self.out_channels) * 2 < 2 ** 32 - 4096)

    def forward(self, x):
        if not self._fast(x):
            return super().forward(x)
        if x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        return _Conv3x3Function.apply(x, self.weight, self.bias)


class BatchNormReLU2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (+ the nn.ReLU behind it when relu=True) on channels-last bf16 maps through the fused BatchNorm
    kernels of the sparse path (fused.hip: a [B, H, W, C] map is the same [rows, C] matrix as a sparse tensor's
    features) -- two streaming passes forward, two backward, ReLU included, instead of MIOpen's BatchNorm kernels plus
    an elementwise ReLU each way.  Same parameters / buffers / state-dict keys as nn.BatchNorm2d
    (base_bev_backbone.py:40-41,49-50, center_head.py:19-20); other inputs take nn.BatchNorm2d's own path."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, relu=False):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine,
                         track_running_stats=track_running_stats)
        self.relu = bool(relu)

    def forward(self, x):
        from ..spconv import functional as Fsp
        if ENABLED and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 \
                and x.is_contiguous(memory_format=torch.channels_last):
            B, C, H, W = x.shape
            rows = x.permute(0, 2, 3, 1).reshape(B * H * W, C)
            if Fsp._fusable(self, rows):
                y = Fsp.batch_norm_act(self, rows, None, self.relu)
                return y.view(B, H, W, C).permute(0, 3, 1, 2)
        y = super().forward(x)
        return torch.relu(y) if self.relu else y
