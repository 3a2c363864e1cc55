"""nn.Conv2d(k = 3, stride 1, padding 1) on channels-last bf16 maps through the hand-written implicit-GEMM kernel
(com_amd/csrc/conv2d.hip) instead of MIOpen: forward and data gradient are the same MFMA kernel (weights packed
forward / transposed + rotated), the weight gradient is the sparse path's pair-wise X^T dY kernel (spconv.hip) run over
the DENSE pair lists of the 9 taps -- built once per map shape.  `Conv3x3` is a drop-in nn.Conv2d subclass (same
parameters and state-dict keys, base_bev_backbone.py:37-51 / center_head.py:17-24 construct it with the same arguments);
anything the kernel does not cover (other kernel sizes / strides, CPU tensors, fp32 maps outside a bf16 autocast
region, cin % 32 != 0) takes
nn.Conv2d's own path."""
import os

import torch
import torch.nn as nn

from .. import ops

ENABLED = True
# BatchNorm sums in the dense convs' epilogues (PcdBnReduce modes 1 / 2 of pcd_conv2d_3x3_nhwc_bn): bit 0 = forward
# statistics, bit 1 = backward reductions on the data-gradient launch.  Measured in the full step
# (tools/exp_dense_bn_epi.sh, 2 x 80 replays): off 7.84, forward 7.79, backward 7.89, both 7.86 ms -- the backward form
# re-reads the BatchNorm's input and output tile in the data-gradient epilogue and loses; only the forward one is on.
DENSE_BN_EPILOGUE = 1
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


def _pad32(c):
    return (c + 31) // 32 * 32


BATCH_BN_COUNTERS = True


def _plane_pairs(mode_f, B, hi, wi, device):
    """Dense pair lists for the WEIGHT gradients of the plane operators (conv2d.hip, pack modes 2 / 4 / 6), in the
    (gathered row, accumulated row) convention of the sparse pair kernels, built once per map shape:
      mode 2  Conv2d(3, stride 2, padding 1): tap k = ky * 3 + kx pairs the fine INPUT pixel (2 oy + ky - 1, 2 ox + kx - 1)
              with the coarse output pixel (oy, ox) -> dW [cout, 9, cin]
      mode 4  ConvTranspose2d(2, stride 2): tap k = a * 2 + b pairs the fine OUTPUT pixel (2 y + a, 2 x + b) with the coarse
              input pixel (y, x); the kernel runs with the roles of x / dy swapped so that dW comes out as [cin, 4, cout],
              i.e. in the parameter's own [cin, cout, 2, 2] layout after the conv2d_layout reduction
      mode 6  ConvTranspose2d(1, stride 1): identity pairs.
    (hi, wi) = size of the layer's INPUT map.  Returns pairs [K, 2, n], pair_num [K]."""
    key = ("planes", mode_f, B, hi, wi, str(device))
    hit = _PAIRS.get(key)
    if hit is not None:
        return hit
    if mode_f == 2:
        ho, wo = (hi - 1) // 2 + 1, (wi - 1) // 2 + 1
        n = B * ho * wo
        p = torch.arange(n, device=device, dtype=torch.int64)
        b, oy, ox = p // (ho * wo), (p // wo) % ho, p % wo
        pairs = torch.full((9, 2, n), -1, dtype=torch.int32, device=device)
        num = torch.zeros((9,), dtype=torch.int32, device=device)
        for k in range(9):
            fy, fx = 2 * oy + k // 3 - 1, 2 * ox + k % 3 - 1
            ok = (fy >= 0) & (fy < hi) & (fx >= 0) & (fx < wi)
            m = int(ok.sum())
            pairs[k, 0, :m] = ((b * hi + fy) * wi + fx)[ok].int()
            pairs[k, 1, :m] = p[ok].int()
            num[k] = m
    elif mode_f == 4:
        n = B * hi * wi
        p = torch.arange(n, device=device, dtype=torch.int64)
        b, y, x = p // (hi * wi), (p // wi) % hi, p % wi
        pairs = torch.empty((4, 2, n), dtype=torch.int32, device=device)
        for k in range(4):
            pairs[k, 0] = ((b * 2 * hi + 2 * y + k // 2) * (2 * wi) + 2 * x + k % 2).int()
            pairs[k, 1] = p.int()
        num = torch.full((4,), n, dtype=torch.int32, device=device)
    else:
        n = B * hi * wi
        p = torch.arange(n, device=device, dtype=torch.int32)
        pairs = torch.stack([p, p]).reshape(1, 2, n).contiguous()
        num = torch.full((1,), n, dtype=torch.int32, device=device)
    _PAIRS[key] = (pairs, num)
    return pairs, num


def _scheduled_backward(need_dx, want_w, want_b, wp, bp, direct_ok, dgrad, wgrad, bsum, keep):
    """The backward schedule shared by the dense convs (as in the sparse convs, spconv/functional.py): the data gradient
    is issued first on the current stream, the weight / bias gradients run on the side stream from an event recorded
    before it; with DIRECT_GRAD they are written straight into .grad (the slab reduction deferred to ONE launch at the
    join, in the parameter's own layout) and the join is lagged -- no copy, no AccumulateGrad add, no per-layer reduce
    launch.  dgrad() -> dx; wgrad(direct) -> dw or None; bsum(direct) -> db or None; keep = tensors alive until the join."""
    from ..spconv import functional as Fsp
    dx = dw = db = None
    cur = torch.cuda.current_stream()
    side = ready = None
    if Fsp.OVERLAP_WGRAD and need_dx and (want_w or want_b):
        ready = torch.cuda.Event()
        ready.record(cur)
    if need_dx:
        dx = dgrad()
    if ready is not None:
        side = Fsp._side_stream(keep[0].device)
        side.wait_event(ready)

    def direct(ps):          # (wp / bp: one parameter or a list of them -- all or none)
        ps = ps if isinstance(ps, (list, tuple)) else [ps]
        return len(ps) > 0 and all(Fsp.DIRECT_GRAD and direct_ok and p is not None and p.grad is not None
                                   and p.grad.dtype == torch.float32 and p.grad.is_contiguous() for p in ps)
    direct_w, direct_b = want_w and direct(wp), want_b and direct(bp)
    deferred = (Fsp.WGRAD_JOIN_LAG > 0 and side is not None and (direct_w or not want_w)
                and (direct_b or not want_b))
    if direct_w and deferred:
        for p in (wp if isinstance(wp, (list, tuple)) else [wp]):
            Fsp._claim_direct(p, "w")
    if direct_b and deferred:
        for p in (bp if isinstance(bp, (list, tuple)) else [bp]):
            Fsp._claim_direct(p, "b")
    with torch.cuda.stream(side) if side is not None else Fsp._NullCtx():
        if ops.STAMPS is not None:
            Fsp._STAMP_SEQ[0] += 1
            ops.stamp(f"d2w{Fsp._STAMP_SEQ[0]}")
        if want_w:
            dw = wgrad(direct_w and deferred)
        if want_b:
            db = bsum(direct_b and deferred)
        if deferred:
            ev = torch.cuda.Event()
            ev.record(side)
    if deferred:
        Fsp._PENDING.append((ev,) + tuple(keep) + (None,))     # inputs stay alive until the lagged join
        if len(Fsp._PENDING) > Fsp.WGRAD_JOIN_LAG:
            cur.wait_event(Fsp._PENDING[-1 - Fsp.WGRAD_JOIN_LAG][0])
            del Fsp._PENDING[:len(Fsp._PENDING) - Fsp.WGRAD_JOIN_LAG]
    elif side is not None:
        cur.wait_stream(side)
        for t in (dw, db):
            for u in (t if isinstance(t, (list, tuple)) else [t]):
                if u is not None:
                    u.record_stream(cur)
    return dx, dw, db


class _Conv3x3Function(torch.autograd.Function):
    """Output channel counts that are not a multiple of 32 (the 1 / 2 / 3-channel final convs of the head towers) run
    padded with zero weights (the packs carry the padding): MIOpen spends 0.4 ms on each of those tiny convs.
    pack_f / pack_d: packs made ahead (Conv3x3Packs), or None: packed here."""

    @staticmethod
    def forward(ctx, x, weight, bias, pack_f, pack_d, bn_follows=False, bn_link=None):
        # x: [B, C, H, W] bf16, channels_last storage
        # bn_follows: a training-mode BatchNorm consumes the output -> its statistics are taken in this launch's epilogue
        # bn_link: x IS the output of a fused BatchNorm -> its backward reductions ride on our data-gradient launch
        from ..spconv import functional as Fsp
        xn = x.detach().permute(0, 2, 3, 1)
        assert xn.is_contiguous()
        cout = weight.shape[0]
        cp = _pad32(cout)
        b = bias.detach().float() if bias is not None else None
        if b is not None and cp != cout:
            b = torch.nn.functional.pad(b, (0, cp - cout))
        if pack_f is None:
            pack_f = ops.conv2d_pack_weight(weight, 0)
        stats = ops.BnReduce(1) if (bn_follows and Fsp.FUSE_BN_REDUCTIONS and cp == cout and (DENSE_BN_EPILOGUE & 1)) else None
        y = ops.conv2d_3x3_nhwc(xn, pack_f, cp, b, bn_reduce=stats)
        if cp != cout:
            y = y[..., :cout].contiguous()
        ctx.save_for_backward(xn, weight, pack_d)
        ctx.has_bias, ctx.cout = bias is not None, cout
        ctx.weight_param = weight if isinstance(weight, nn.Parameter) else None
        ctx.bias_param = bias if isinstance(bias, nn.Parameter) else None
        ctx.bn_link = bn_link if (Fsp.FUSE_BN_REDUCTIONS and (DENSE_BN_EPILOGUE & 2)) else None
        out = y.permute(0, 3, 1, 2)
        if stats is not None and stats.partial is not None:
            out._pcd_stats = stats
        return out

    @staticmethod
    def backward(ctx, dy):
        xn, weight, pack_d = ctx.saved_tensors
        B, H, W, cin = xn.shape
        cout, cp = ctx.cout, _pad32(ctx.cout)
        dyn = dy.permute(0, 2, 3, 1)
        if dyn.dtype != torch.bfloat16:
            dyn = dyn.to(torch.bfloat16)
        if cp != cout:
            dyn = torch.nn.functional.pad(dyn, (0, cp - cout))
        dyn = dyn.contiguous()
        from ..spconv import functional as Fsp
        want_w = ctx.needs_input_grad[1]
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        wp, bp = ctx.weight_param, ctx.bias_param

        def dgrad():
            pd = pack_d if pack_d is not None else ops.conv2d_pack_weight(weight, 1)
            link, red = ctx.bn_link, None
            if link is not None and link.x.shape == (B * H * W, cin):
                red = ops.BnReduce(2, relu=link.relu, x=link.x, y=xn.reshape(-1, cin) if link.relu else None,
                                   mean=link.mean, invstd=link.invstd)
            dxn = ops.conv2d_3x3_nhwc(dyn, pd, cin, bn_reduce=red)
            if red is not None and red.partial is not None:
                link.result = (dxn.view(-1, cin), red.partial, red.rows)
            return dxn.permute(0, 3, 1, 2)

        def wgrad(direct):
            if ops.conv2d_wgrad_splits(B, H, W, cin, cp) > 0:      # the dense kernel (no pair lists)
                if direct:
                    ops.conv2d_wgrad(xn, dyn, cout, out=wp.grad, defer=Fsp._WGRAD_JOBS)
                    return None
                return ops.conv2d_wgrad(xn, dyn, cout).to(weight.dtype)
            pairs, num = _dense_pairs(B, H, W, xn.device)
            if direct:   # (zero-padded output channels: only the real rows of the slabs are reduced into .grad)
                ops.wgrad(xn.reshape(-1, cin), cin, dyn.reshape(-1, cp), pairs, num, 9, out=wp.grad,
                          defer=Fsp._WGRAD_JOBS, conv2d_layout=True, cout_write=cout if cp != cout else 0)
                return None
            dwk = ops.wgrad(xn.reshape(-1, cin), cin, dyn.reshape(-1, cp), pairs, num, 9)  # [cp, 9, cin] f32
            return dwk[:cout].permute(0, 2, 1).reshape(cout, cin, 3, 3).to(weight.dtype)

        def bsum(direct):
            if direct and cp == cout:
                ops.col_sum(dyn.reshape(-1, cp), out=bp.grad)
                return None
            if direct:
                bp.grad.copy_(ops.col_sum(dyn.reshape(-1, cp))[:cout])
                return None
            return ops.col_sum(dyn.reshape(-1, cp))[:cout]   # fp32 column sums in a fixed order

        dx, dw, db = _scheduled_backward(ctx.needs_input_grad[0], want_w, want_b, wp, bp, True, dgrad, wgrad, bsum,
                                         (xn, dyn))
        return dx, dw, db, None, None, None, None


class Conv3x3(nn.Conv2d):
    """nn.Conv2d whose forward takes the HIP kernel when it applies (see the module docstring)."""

    def _fast(self, x):
        bf16 = x.dtype == torch.bfloat16 or (torch.is_autocast_enabled()
                                             and torch.get_autocast_dtype('cuda') == torch.bfloat16)
        return (ENABLED and bf16 and x.is_cuda and x.dim() == 4 and self.kernel_size == (3, 3) and self.stride == (1, 1)
                and self.padding == (1, 1) and self.dilation == (1, 1) and self.groups == 1
                and self.padding_mode == 'zeros' and self.in_channels % 32 == 0
                and x.shape[0] * x.shape[2] * x.shape[3] * max(self.in_channels, _pad32(self.out_channels)) * 2
                < 2 ** 32 - 4096)

    bn_follows = False   # set by the owner when a BatchNormReLU2d consumes the output (dense2d.py)

    def forward(self, x):
        if not self._fast(x):
            return super().forward(x)
        link = getattr(x, "_pcd_bn_link", None)
        if x.dtype != torch.bfloat16:
            x, link = x.to(torch.bfloat16), None
        if not x.is_contiguous(memory_format=torch.channels_last):
            x, link = x.contiguous(memory_format=torch.channels_last), None
        pf, pd = self._take_packs()
        follows = bool(self.bn_follows and self.training and torch.is_grad_enabled())
        return _Conv3x3Function.apply(x, self.weight, self.bias, pf, pd, follows, link)

    def _take_packs(self):
        pf = pd = None
        ahead = getattr(self, "_packs_ahead", None)
        if ahead is not None:                        # Conv3x3Packs.run() since the last weight update: use them ONCE
            self._packs_ahead = None
            if ahead[2] == self.weight._version:     # (an in-place change since run() -- load_state_dict, broadcast,
                pf, pd = ahead[0], ahead[1]          #  a manual edit -- bumps the version: pack again from the weights)
        return pf, pd


class _ConvPlanesFunction(torch.autograd.Function):
    """The stride-2 3x3 conv (mode 2) and the k = stride transposed convs (modes 4, 6) of BaseBEVBackbone through the
    plane kernels of conv2d.hip; the data gradient is the same kernel family with the next odd pack mode, the weight
    gradient the sparse pair kernels over dense pair lists (_plane_pairs).  No bias (base_bev_backbone.py:38,58,66)."""

    @staticmethod
    def forward(ctx, x, weight, pack_f, pack_d, mode_f):
        xn = x.detach().permute(0, 2, 3, 1)
        assert xn.is_contiguous()
        B, H, W, _ = xn.shape
        _, cout = ops.conv2d_layer_channels(weight, mode_f)
        out_hw = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if mode_f == 2 else ((2 * H, 2 * W) if mode_f == 4 else (H, W))
        if pack_f is None:
            pack_f = ops.conv2d_pack_weight(weight, mode_f)
        y = ops.conv2d_planes_nhwc(mode_f, xn, pack_f, cout, out_hw)
        ctx.save_for_backward(xn, weight, pack_d)
        ctx.mode_f = mode_f
        ctx.weight_param = weight if isinstance(weight, nn.Parameter) else None
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        xn, weight, pack_d = ctx.saved_tensors
        mode_f = ctx.mode_f
        B, H, W, cin = xn.shape
        _, cout = ops.conv2d_layer_channels(weight, mode_f)
        dyn = dy.permute(0, 2, 3, 1)
        if dyn.dtype != torch.bfloat16:
            dyn = dyn.to(torch.bfloat16)
        dyn = dyn.contiguous()
        from ..spconv import functional as Fsp
        wp = ctx.weight_param
        kk = {2: 9, 4: 4, 6: 1}[mode_f]

        def dgrad():
            pd = pack_d if pack_d is not None else ops.conv2d_pack_weight(weight, mode_f + 1)
            return ops.conv2d_planes_nhwc(mode_f + 1, dyn, pd, cin, (H, W)).permute(0, 3, 1, 2)

        def wgrad(direct):
            fine, coarse = (xn, dyn) if mode_f == 2 else (dyn, xn)
            if ops.conv2d_wgrad_planes_splits(mode_f, B, coarse.shape[1], coarse.shape[2], fine.shape[3], coarse.shape[3]) > 0:
                if direct:                                        # the dense kernel (no pair lists)
                    ops.conv2d_wgrad_planes(mode_f, fine, coarse, out=wp.grad, defer=Fsp._WGRAD_JOBS)
                    return None
                return ops.conv2d_wgrad_planes(mode_f, fine, coarse).to(weight.dtype)
            pairs, num = _plane_pairs(mode_f, B, H, W, xn.device)
            if mode_f == 2:       # gathered rows = x (contraction c = cin), accumulated rows = dy (o = cout)
                a, ca, b_, cb = xn.reshape(-1, cin), cin, dyn.reshape(-1, cout), cout
            else:                 # transposed convs: roles swapped -> dW [cin, K, cout] = the parameter's layout
                a, ca, b_, cb = dyn.reshape(-1, cout), cout, xn.reshape(-1, cin), cin
            if direct:
                ops.wgrad(a, ca, b_, pairs, num, kk, out=wp.grad, defer=Fsp._WGRAD_JOBS, conv2d_layout=True)
                return None
            dwk = ops.wgrad(a, ca, b_, pairs, num, kk)                    # [cb, K, ca] f32
            k = int(round(kk ** 0.5))
            return dwk.permute(0, 2, 1).reshape(cb, ca, k, k).to(weight.dtype)

        dx, dw, _ = _scheduled_backward(ctx.needs_input_grad[0], ctx.needs_input_grad[1], False, wp, None, True, dgrad,
                                        wgrad, None, (xn, dyn))
        return dx, dw, None, None, None


def _plane_input(x):
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    return x


def _bf16_region(x):
    return x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16)


class _PlaneConvMixin:
    MODE_F = None

    def _take_packs(self):
        pf = pd = None
        ahead = getattr(self, "_packs_ahead", None)
        if ahead is not None:
            self._packs_ahead = None
            if ahead[2] == self.weight._version:
                pf, pd = ahead[0], ahead[1]
        return pf, pd


class Conv3x3S2(_PlaneConvMixin, nn.Conv2d):
    """nn.Conv2d(k = 3, stride 2, padding 1, bias=False) -- the conv that opens a down-sampling block
    (base_bev_backbone.py:36-41: ZeroPad2d(1) + Conv2d(padding=0), the same arithmetic) -- through the plane kernels."""
    MODE_F = 2

    def _fast(self, x):
        return (ENABLED and _bf16_region(x) and x.is_cuda and x.dim() == 4 and self.kernel_size == (3, 3)
                and self.stride == (2, 2) and self.padding == (1, 1) and self.dilation == (1, 1) and self.groups == 1
                and self.padding_mode == 'zeros' and self.bias is None and self.in_channels % 32 == 0
                and self.out_channels % 32 == 0
                and x.shape[0] * x.shape[2] * x.shape[3] * max(self.in_channels, self.out_channels) * 2 < 2 ** 32 - 4096)

    def forward(self, x):
        if not self._fast(x):
            return super().forward(x)
        pf, pd = self._take_packs()
        return _ConvPlanesFunction.apply(_plane_input(x), self.weight, pf, pd, 2)


class UpConvT(_PlaneConvMixin, nn.ConvTranspose2d):
    """nn.ConvTranspose2d(k = stride in {1, 2}, bias=False) -- the deblocks (base_bev_backbone.py:55-62) -- through the
    plane kernels (k = 2: four 1-tap planes written with stride 2, no intermediate + pixel shuffle pass)."""

    @property
    def MODE_F(self):
        return 4 if self.kernel_size == (2, 2) else 6

    def _fast(self, x):
        return (ENABLED and _bf16_region(x) and x.is_cuda and x.dim() == 4 and self.kernel_size in ((1, 1), (2, 2))
                and self.stride == self.kernel_size and self.padding == (0, 0) and self.output_padding == (0, 0)
                and self.dilation == (1, 1) and self.groups == 1 and self.bias is None
                and self.in_channels % 32 == 0 and self.out_channels % 32 == 0
                and x.shape[0] * x.shape[2] * x.shape[3] * 4 * max(self.in_channels, self.out_channels) * 2 < 2 ** 32 - 4096)

    def forward(self, x, output_size=None):
        if output_size is not None or not self._fast(x):
            return super().forward(x, output_size)
        pf, pd = self._take_packs()
        return _ConvPlanesFunction.apply(_plane_input(x), self.weight, pf, pd, self.MODE_F)


class _BranchConvsFunction(torch.autograd.Function):
    """The LAST 3x3 convs (64 -> 1..3 channels, bias) of all branches of a SeparateHead (center_head.py:21-24) on the
    channel blocks of ONE shared activation a [B, 64 n, H, W]: branch i reads channels [64 i, 64 i + 64) in place, its data
    gradient fills that block of da -- no slice copies forward, no zero-filled slice gradients and no adds backward.
    Outputs are the first cout_i channels of maps computed with zero-padded output channels (views, pixel stride 32).
    metas[i] = (pack_f, pack_d) made ahead or (None, None); wb = w_0, b_0, w_1, b_1, ..."""

    @staticmethod
    def forward(ctx, a, width, metas, *wb):
        an = a.detach().permute(0, 2, 3, 1)
        assert an.is_contiguous() and an.shape[3] == width * (len(wb) // 2)
        outs, packs_d = [], []
        for i in range(len(wb) // 2):
            w, b = wb[2 * i], wb[2 * i + 1]
            cout = w.shape[0]
            cp = _pad32(cout)
            bp = torch.nn.functional.pad(b.detach().float(), (0, cp - cout)) if b is not None else None
            pf = metas[i][0] if metas[i][0] is not None else ops.conv2d_pack_weight(w, 0)
            y = ops.conv2d_3x3_nhwc(an[..., width * i:width * (i + 1)], pf, cp, bp)
            outs.append(y[..., :cout].permute(0, 3, 1, 2))
            packs_d.append(metas[i][1])
        ctx.save_for_backward(an, *[wb[2 * i] for i in range(len(wb) // 2)])
        ctx.width, ctx.packs_d = width, packs_d
        ctx.w_params = [wb[2 * i] if isinstance(wb[2 * i], nn.Parameter) else None for i in range(len(wb) // 2)]
        ctx.b_params = [wb[2 * i + 1] if isinstance(wb[2 * i + 1], nn.Parameter) else None for i in range(len(wb) // 2)]
        ctx.has_bias = [wb[2 * i + 1] is not None for i in range(len(wb) // 2)]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        an, ws = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        from ..spconv import functional as Fsp
        B, H, W, C = an.shape
        n, width = len(ws), ctx.width
        cps = [_pad32(w.shape[0]) for w in ws]
        # the output gradients, zero-padded to the channel count the kernels contract over: ONE fill + n copies
        pad = torch.zeros((sum(cps) * B * H * W,), dtype=torch.bfloat16, device=an.device)
        dyp, off = [], 0
        for i in range(n):
            t = pad[off:off + cps[i] * B * H * W].view(B, H, W, cps[i])
            off += cps[i] * B * H * W
            if dys[i] is not None:
                t[..., :ws[i].shape[0]].copy_(dys[i].permute(0, 2, 3, 1))
            dyp.append(t)
        live = [i for i in range(n) if dys[i] is not None]
        want_w = any(ctx.needs_input_grad[3 + 2 * i] for i in live)
        want_b = any(ctx.has_bias[i] and ctx.needs_input_grad[4 + 2 * i] for i in live)

        def dgrad():
            da = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=an.device)
            for i in range(n):
                blk = da[..., width * i:width * (i + 1)]
                if dys[i] is None:
                    blk.zero_()
                    continue
                pd = ctx.packs_d[i] if ctx.packs_d[i] is not None else ops.conv2d_pack_weight(ws[i], 1)
                ops.conv2d_3x3_nhwc(dyp[i], pd, width, out=blk)
            return da.permute(0, 3, 1, 2)

        def wgrad(direct):
            pairs, num = _dense_pairs(B, H, W, an.device)
            a2 = an.reshape(-1, C)
            res = []
            for i in range(n):
                if dys[i] is None or not ctx.needs_input_grad[3 + 2 * i]:
                    res.append(None)
                    continue
                cout = ws[i].shape[0]
                if ops.conv2d_wgrad_splits(B, H, W, width, cps[i]) > 0:
                    xblk = an[..., width * i:width * (i + 1)]
                    if direct:
                        ops.conv2d_wgrad(xblk, dyp[i], cout, out=ctx.w_params[i].grad, defer=Fsp._WGRAD_JOBS)
                        res.append(None)
                    else:
                        res.append(ops.conv2d_wgrad(xblk, dyp[i], cout).to(ws[i].dtype))
                    continue
                xb = a2[:, width * i:width * (i + 1)]
                if direct:
                    ops.wgrad(xb, width, dyp[i].reshape(-1, cps[i]), pairs, num, 9, out=ctx.w_params[i].grad,
                              defer=Fsp._WGRAD_JOBS, conv2d_layout=True, cout_write=cout if cps[i] != cout else 0,
                              x_block=True)
                    res.append(None)
                else:
                    dwk = ops.wgrad(xb, width, dyp[i].reshape(-1, cps[i]), pairs, num, 9, x_block=True)
                    res.append(dwk[:cout].permute(0, 2, 1).reshape(cout, width, 3, 3).to(ws[i].dtype))
            return res

        def bsum(direct):
            res = []
            for i in range(n):
                if dys[i] is None or not (ctx.has_bias[i] and ctx.needs_input_grad[4 + 2 * i]):
                    res.append(None)
                    continue
                cs = ops.col_sum(dyp[i].reshape(-1, cps[i]))[:ws[i].shape[0]]
                if direct:
                    ctx.b_params[i].grad.copy_(cs)
                    res.append(None)
                else:
                    res.append(cs)
            return res

        wps = [ctx.w_params[i] for i in live if ctx.needs_input_grad[3 + 2 * i]]
        bps = [ctx.b_params[i] for i in live if ctx.has_bias[i] and ctx.needs_input_grad[4 + 2 * i]]
        dx, dws, dbs = _scheduled_backward(ctx.needs_input_grad[0], want_w, want_b, wps, bps, True, dgrad, wgrad, bsum,
                                           (an, pad))
        grads = [dx, None, None]
        for i in range(n):
            grads.append(dws[i] if dws is not None else None)
            grads.append(dbs[i] if dbs is not None else None)
        return tuple(grads)


class Conv3x3Packs:
    """Forward + data-gradient packs of all Conv3x3 modules of a model in ONE launch (pcd_conv2d_pack_weights_batched)
    into persistent buffers -- call `run()` right after every optimizer step (the weights do not change again before
    the next forward; 2 x 22 small pack launches otherwise sit in front of the convs of the CenterPoint BEV stack).
    A module uses the packs for exactly one forward / backward and packs by itself again afterwards, so a forgotten
    `run()` costs time, never correctness; packs made BEFORE a torch-visible in-place change of the weight
    (load_state_dict, checkpoint restore, dist.broadcast, manual edits: all bump `weight._version`) are dropped."""

    def __init__(self, model):
        import ctypes  # noqa: F401
        from .. import _lib as L
        self.convs, self.modes = [], []
        extra = []
        for m in model.modules():
            hook = getattr(m, "_pack_extra_convs", None)       # (SeparateHead: its batched first-stage conv, if any)
            if callable(hook):
                extra += list(hook())
        for m in list(model.modules()) + extra:
            if not (hasattr(m, "weight") and m.weight is not None and m.weight.is_cuda and m.weight.dtype == torch.float32):
                continue
            if isinstance(m, Conv3x3) and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) \
                    and m.in_channels % 32 == 0:
                self.convs.append(m), self.modes.append((0, 1))
            elif isinstance(m, Conv3x3S2) and m.stride == (2, 2) and m.padding == (1, 1) and m.bias is None \
                    and m.in_channels % 32 == 0 and m.out_channels % 32 == 0:
                self.convs.append(m), self.modes.append((2, 3))
            elif isinstance(m, UpConvT) and m.kernel_size in ((1, 1), (2, 2)) and m.stride == m.kernel_size \
                    and m.bias is None and m.in_channels % 32 == 0 and m.out_channels % 32 == 0:
                self.convs.append(m), self.modes.append((m.MODE_F, m.MODE_F + 1))
        rows, first, self.bufs = [], 0, []
        lib = L.lib()
        for m, modes in zip(self.convs, self.modes):
            cin, cout = m.in_channels, m.out_channels
            pair = []
            for mode in modes:
                nbytes = lib.pcd_conv2d_packed_weight_bytes(cin, cout, mode)
                buf = torch.empty((nbytes // 2,), dtype=torch.bfloat16, device=m.weight.device)
                rows.append([m.weight.data_ptr(), buf.data_ptr(), cin, cout, _pad32(cout), mode, first, nbytes // 16])
                first += (nbytes // 16 + 255) // 256
                pair.append(buf)
            self.bufs.append(tuple(pair))
        self.total_blocks = first
        self.keys = tuple(m.weight.data_ptr() for m in self.convs)
        self.table = torch.tensor(rows, dtype=torch.int64).to(self.convs[0].weight.device) if rows else None

    def run(self):
        from .. import _lib as L
        if self.table is None:
            return
        if self.keys != tuple(m.weight.data_ptr() for m in self.convs):
            raise L.PcdError("Conv3x3Packs: a weight moved since the plan was built (build a new plan)")
        L.check(L.lib().pcd_conv2d_pack_weights_batched(L.ptr(self.table), len(self.convs) * 2, self.total_blocks,
                                                        L.stream_ptr()), "pcd_conv2d_pack_weights_batched")
        for m, pair in zip(self.convs, self.bufs):
            m._packs_ahead = (pair[0], pair[1], m.weight._version)


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

    def forward(self, x, out=None):
        """out (optional): a [B, C, H, W] channel block of a wider channels-last map; when the fused training path
        runs, y is written THERE (self.wrote_out = True) and the returned tensor is that view -- otherwise `out` is
        ignored and the caller concatenates as usual."""
        from ..spconv import functional as Fsp
        self.wrote_out = False
        if ENABLED and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 \
                and x.is_contiguous(memory_format=torch.channels_last):
            B, C, H, W = x.shape
            rows = x.permute(0, 2, 3, 1).reshape(B * H * W, C)
            if Fsp._fusable(self, rows):
                out_rows = None
                if out is not None and self.training and self.relu and self.bias is not None \
                        and out.shape == x.shape and out.dtype == x.dtype and out.stride(1) == 1:
                    out_rows = out.permute(0, 2, 3, 1).reshape(B * H * W, C)
                    if out_rows.data_ptr() != out.data_ptr() or out_rows.stride(1) != 1:
                        out_rows = None                                   # (reshape had to copy: not a column block)
                st = getattr(x, "_pcd_stats", None)           # the conv in front took the statistics in its epilogue
                if st is not None:
                    rows._pcd_stats = st
                y = Fsp.batch_norm_act(self, rows, None, self.relu, out=out_rows)
                self.wrote_out = out_rows is not None
                res = y.view(B, H, W, C).permute(0, 3, 1, 2)
                link = getattr(y, "_pcd_bn_link", None)       # ... and the conv behind can take the backward reductions
                if link is not None:
                    res._pcd_bn_link = link
                return res
        if getattr(self, "_defer_nbt", False) and self.training and self.num_batches_tracked is not None:
            self.num_batches_tracked.sub_(1)         # bump_bn_counters() already counted this call
        y = super().forward(x)
        return torch.relu(y) if self.relu else y


class ConcatChannelBlocks(torch.autograd.Function):
    """torch.cat(parts, dim=1) when the parts already ARE the channel blocks of `wide` (each BatchNorm wrote its output
    there): returns `wide`, no copy; backward hands every part its channel block of the gradient as a view."""

    @staticmethod
    def forward(ctx, wide, *parts):
        off = 0
        for p in parts:
            assert p.data_ptr() == wide.data_ptr() + off * wide.element_size() and p.stride() == wide.stride()
            off += p.shape[1]
        assert off == wide.shape[1]
        ctx.widths = [p.shape[1] for p in parts]
        return wide.view_as(wide)

    @staticmethod
    def backward(ctx, g):
        outs, off = [None], 0
        for w in ctx.widths:
            outs.append(g[:, off:off + w])
            off += w
        return tuple(outs)


def bump_bn_counters(module):
    """num_batches_tracked += 1 for every BatchNormReLU2d below `module` in ONE multi-tensor launch (the modules then
    skip their own increment): 20 one-element launches per step otherwise sit between the convs of the BEV stack."""
    if not BATCH_BN_COUNTERS:
        return
    bns = module.__dict__.get("_bn2d_list")
    if bns is None:
        bns = [m for m in module.modules() if isinstance(m, BatchNormReLU2d)]
        for m in bns:
            m._defer_nbt = True
        module.__dict__["_bn2d_list"] = bns
    if module.training:
        counters = [m.num_batches_tracked for m in bns if m.num_batches_tracked is not None]
        if counters:
            torch._foreach_add_(counters, 1)
