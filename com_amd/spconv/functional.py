"""autograd.Function wrappers over the HIP kernels (binder convention of the reference's own ops:
pcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py:8-50)."""
import torch
from torch.autograd import Function

from .. import ops


OVERLAP_WGRAD = True
# Opt-in, needs DIRECT_GRAD: the join of layer i's weight-gradient kernels is postponed until WGRAD_JOIN_LAG
# later conv backward nodes have been issued (the side stream is in order, so only the LAST event matters),
# so the current stream never idles while wgrad(+reduce) of a layer outlasts its dgrad.  The inputs of the
# pending kernels are kept alive by reference until their join.  The training loop must call
# join_deferred_wgrad() after loss.backward() and before it reads any `.grad`.
WGRAD_JOIN_LAG = 0
# Opt-in for training loops whose parameters each receive exactly ONE gradient contribution per step and own
# pre-allocated fp32 `.grad` buffers (bench.py: views of one flat bucket): the kernels then write dW / dbias /
# dgamma / dbeta straight into `.grad` and autograd gets None for them -- no temporary, no AccumulateGrad add
# kernel per parameter (79 tiny launches per step on the critical path of VoxelResBackBone8x).
DIRECT_GRAD = False
USE_DGRAD_CLASSES = True      # strided-conv data gradient over parity-class row groups (ops.dgrad_classes)
# The conv kernels take the per-channel sums of the BatchNorm beside them in their epilogue (ops.BnReduce): the
# forward conv the batch statistics of the BatchNorm that follows it, the data-gradient kernel the two reductions
# of the BatchNorm whose output was the conv's input.  Each replaces one streaming pass + launch per BatchNorm.
# The hand-over goes through attributes on the tensors (`_pcd_stats` on a conv output, `_pcd_bn_link` on a
# BatchNorm output) and is only used when the consumer sees exactly that tensor.
FUSE_BN_REDUCTIONS = True
_SIDE = {}
_PENDING = []   # [(event, tensors kept alive)] of weight-gradient launches not yet joined


_COLSUM_JOBS = []   # [(partial, rows, bias.grad)] of deferred bias gradients: one launch at the join
_WGRAD_JOBS = []    # [(slab, dW, kvol, cin, cout, pmax)] of deferred slab reductions: one launch at the join
WGRAD_FLUSH_BYTES = 32 << 20     # deferred slab reductions are flushed whenever this many slab bytes have piled up (0: once, at the join)
_WGRAD_KEEP = []
_SP_SEQ = [0]       # diagnostics (ops.STAMPS["sparse"]): running index of the sparse conv layers in backward order
_STAMP_SEQ = [0]    # diagnostics (ops.STAMPS): running index of the dense weight-gradient launches of a step
_DIRECT_WRITTEN = set()   # ids of the parameters whose .grad a kernel has OVERWRITTEN since the last join (DIRECT_GRAD)


def reset_deferred():
    """Drop every pending job of the current step WITHOUT running it (call it when loss.backward() raised: the jobs
    hold raw pointers into tensors of the aborted step).  join_deferred_wgrad() is the normal end of a step."""
    _COLSUM_JOBS.clear()
    _WGRAD_JOBS.clear()
    _WGRAD_KEEP.clear()
    _PENDING.clear()
    _DIRECT_WRITTEN.clear()


def _claim_direct(param, what):
    """DIRECT_GRAD kernels overwrite `.grad`: a second contribution to the same parameter inside one step (a conv
    module used twice, gradient accumulation over micro-batches without a join in between) would silently replace
    the first one -- refuse it."""
    key = (id(param), what)
    if key in _DIRECT_WRITTEN:
        raise RuntimeError("DIRECT_GRAD: a second gradient contribution for the same parameter arrived before "
                           "join_deferred_wgrad(); direct writes overwrite .grad -- disable DIRECT_GRAD for shared "
                           "modules / micro-batch accumulation")
    _DIRECT_WRITTEN.add(key)


def join_deferred_wgrad():
    """Make the current stream wait for every side-stream weight-gradient kernel issued so far (and finish the
    deferred bias gradients with one launch on that stream).  Ends the step for the DIRECT_GRAD bookkeeping."""
    _DIRECT_WRITTEN.clear()
    _STAMP_SEQ[0] = 0
    _SP_SEQ[0] = 0
    if _COLSUM_JOBS or _WGRAD_JOBS:
        side = _side_stream((_COLSUM_JOBS or _WGRAD_JOBS)[0][0].device)
        with torch.cuda.stream(side):
            if _WGRAD_JOBS:
                ops.wgrad_reduce_batched(_WGRAD_JOBS)
            if _COLSUM_JOBS:
                ops.col_sum_finalize_batched(_COLSUM_JOBS)
            ops.stamp("side_end")
            ev = torch.cuda.Event()
            ev.record(side)
        torch.cuda.current_stream().wait_event(ev)
        _COLSUM_JOBS.clear()
        _WGRAD_JOBS.clear()
        _WGRAD_KEEP.clear()
        _PENDING.clear()
        return
    _WGRAD_KEEP.clear()
    if _PENDING:
        torch.cuda.current_stream().wait_event(_PENDING[-1][0])
        _PENDING.clear()


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _side_stream(device, role="wgrad"):
    key = (device.type, device.index) if role == "wgrad" else (device.type, device.index, role)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def _to_bf16_padded(x, c_pad):
    """[N, C] (f32 or bf16) -> contiguous bf16 [N, c_pad], zero padded channels."""
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    if x.shape[1] != c_pad:
        x = torch.nn.functional.pad(x, (0, c_pad - x.shape[1]))
    return x.contiguous()


def _window_wgrad(c):
    """The "subm_window_wgrad" option admits the window weight gradient for c channels (bit 0 = 64, bit 1 = 32, bit 2 = 16)."""
    from .. import _lib as L
    return bool(L.get_option("subm_window_wgrad") & {64: 1, 32: 2, 16: 4}.get(int(c), 0))


class SparseConvFunction(Function):
    """y = indice_conv(features, weight, rulebook) with bf16 MFMA, fp32 accumulate.

    forward : output-stationary gather-GEMM over rb.nbr_out
    backward: dX = gather-GEMM over the input-stationary view with W^T; dW = pair-wise X^T dY.
    `packed` = (packed_fwd, packed_dgrad or None) prepared by the module (cached per weight version).
    """

    @staticmethod
    def forward(ctx, features, weight, bias, rb, packed_fwd, packed_dgrad=None, passthrough=False, fp8=None, window=False):
        """window=True: `packed_fwd` / `packed_dgrad` are window-kernel packs (ops.pack_weight_window) and `rb` is a SubM
        3x3x3 rulebook over z-fastest rows: forward and data gradient run through ops.subm_window.
        passthrough=True also returns `features` itself as a second output (the identity branch of a residual
        block): backward then receives the gradients of BOTH branches in one call and adds the identity gradient
        inside the dgrad kernel's epilogue instead of through a separate elementwise kernel."""
        cout, cin = weight.shape[0], weight.shape[-1]
        cin_pad = ops.pow2_ge8(cin)
        if window and cin_pad < cout:
            cin_pad = cout                       # a window layer with fewer input channels (conv_input, 5 -> 16): rows zero-padded
        assert features.shape[1] in (cin, ops.pow2_ge8(cin), cin_pad), (features.shape, weight.shape)
        x = _to_bf16_padded(features.detach(), cin_pad)
        out_dtype = features.dtype if features.dtype in (torch.float32, torch.bfloat16) else torch.float32
        b = bias.detach().float().contiguous() if bias is not None else None
        stats = None
        if FUSE_BN_REDUCTIONS and out_dtype == torch.bfloat16 and cout % 16 == 0 and ctx.needs_input_grad[1]:
            stats = ops.BnReduce(1)              # training: a BatchNorm follows every conv of the backbones
        if fp8 is not None and cin_pad >= 16:
            # fp8 FORWARD (BASELINE config 5 as a training step): e4m3 operands with the layer's static per-tensor
            # scales, fp32 accumulation, bf16 / f32 result; the backward pass below is the bf16 one on the saved bf16
            # input (straight-through: the quantisation is treated as the identity).  No BatchNorm sums in this epilogue.
            y = fp8.forward(x, b, rb, cout, out_dtype)
            stats = None
        elif window:
            assert out_dtype == torch.bfloat16 and rb.subm
            y = ops.subm_window(x, packed_fwd, b, rb, cout, bn_reduce=stats)
        elif out_dtype == torch.bfloat16 and ops.pair_conv_usable(rb, cin_pad, cout):
            # strided conv of a z-fastest chain, 16 -> 32 channels: one gather per indice PAIR (ops.pair_conv)
            y = ops.pair_conv(x, packed_fwd, b, rb, 0, cout, out_dtype, bn_reduce=stats)
        else:
            packed = getattr(rb, "nbr_out_packed", None) is not None and out_dtype == torch.bfloat16 \
                and not ops.gather_gemm_is_wide(x.shape[0], cin_pad, rb.kvol, rb.n_out, cout)
            y = ops.gather_gemm(x, packed_fwd, b, rb.nbr_out_packed if packed else rb.nbr_out, rb.kvol, False, rb.n_out, cout,
                                out_dtype, n_dev=rb.n_out_dev, bn_reduce=stats, nbr_packed=packed,
                                zfast=rb.subm and rb.kvol == 27 and getattr(rb, "order", None) == ops.ROWS_YXZ)
        if stats is not None:
            y._pcd_stats = stats
        # bias gradient = column sum of our dy; when dy comes out of a fused BatchNorm backward that kernel sums it
        ctx.colsum_link = None
        if FUSE_BN_REDUCTIONS and bias is not None and ctx.needs_input_grad[2] and out_dtype == torch.bfloat16:
            ctx.colsum_link = _ColsumLink()
            y._pcd_colsum_link = ctx.colsum_link
        # the input is the output of a fused BatchNorm: its backward reductions can ride on our data gradient
        link = getattr(features, "_pcd_bn_link", None) if FUSE_BN_REDUCTIONS else None
        if link is not None and not (features.dtype == torch.bfloat16 and x.data_ptr() == features.data_ptr()
                                     and x.shape == features.shape):
            link = None
        ctx.bn_link = link
        ctx.rb = rb
        ctx.packed_dgrad = packed_dgrad          # callable returning the (cached) dgrad pack of the module
        ctx.cin, ctx.cout, ctx.cin_pad = cin, cout, cin_pad
        ctx.in_cols = features.shape[1]
        ctx.has_bias = bias is not None
        ctx.bias_param = bias if isinstance(bias, torch.nn.Parameter) else None
        ctx.weight_param = weight if isinstance(weight, torch.nn.Parameter) else None
        ctx.in_dtype = features.dtype
        ctx.passthrough = passthrough
        ctx.window = bool(window) and fp8 is None
        ctx.save_for_backward(x, weight)
        if passthrough:
            return y, features.view_as(features)
        return y

    @staticmethod
    def backward(ctx, dy, d_ident=None):
        x, weight = ctx.saved_tensors
        rb = ctx.rb
        if dy is None:                               # only the identity branch received a gradient
            return (d_ident, None, None, None, None, None, None, None, None, None)
        dy16 = _to_bf16_padded(dy, ctx.cout)
        dx = dw = db = None
        # dgrad and wgrad only share their inputs: at B = 4 neither fills the chip (1-4 waves per SIMD), so the
        # weight gradient (+ bias column sum) runs on a second HIP stream concurrently with the data gradient
        # and is joined before returning (fork/join is captured as such in hipGraph mode).
        side = None
        cur = torch.cuda.current_stream()
        # The data gradient is issued FIRST and the weight gradient forks from an event recorded before it: in a
        # captured graph the first successor of a node stays on its hardware queue, and the main chain
        # (BatchNorm backward -> dgrad -> BatchNorm backward ...) should be the one that never changes queues --
        # a cross-queue dependency costs several microseconds, a same-queue one ~1.5.
        ready = None
        if OVERLAP_WGRAD and ctx.needs_input_grad[0] and (ctx.needs_input_grad[1] or
                                                         (ctx.has_bias and ctx.needs_input_grad[2])):
            ready = torch.cuda.Event()
            ready.record(cur)
        if ctx.needs_input_grad[0]:
            if ctx.cin_pad % 16 != 0:
                raise RuntimeError("dgrad needs >= 16 input channels (the 5-channel input layer never "
                                   "requires an input gradient)")
            # (the pack's layout is bound to THIS forward's window decision: conv._packed_dgrad_for)
            if ctx.packed_dgrad is not None:
                packed_d = ctx.packed_dgrad()
            else:
                packed_d = ops.pack_weight_window(weight, 1) if ctx.window else ops.pack_weight(weight, 1)
            add = None
            if d_ident is not None and ctx.cin_pad == ctx.in_cols and d_ident.dtype == ctx.in_dtype:
                add = d_ident.contiguous()           # fused: dx = dgrad + identity-branch gradient
            link, red = ctx.bn_link, None
            if link is not None and (d_ident is None or add is not None):
                # the ReLU mask comes from the BatchNorm's output, which is this conv's saved input x
                red = ops.BnReduce(2, link.relu, x=link.x, y=x if link.relu else None, mean=link.mean,
                                   invstd=link.invstd)
                if not red.usable(ctx.cin_pad, ctx.in_dtype):
                    red = None
            if rb.subm and ctx.window and ctx.in_dtype == torch.bfloat16:
                dxp = ops.subm_window(dy16, packed_d, None, rb, ctx.cin_pad, addend=add, bn_reduce=red)
            elif rb.subm:
                assert not ctx.window, "window packs cannot feed the generic kernel"
                dxp = ops.gather_gemm(dy16, packed_d, None, rb.nbr_out, rb.kvol, True, rb.n_in, ctx.cin_pad,
                                      ctx.in_dtype, n_dev=rb.n_in_dev, addend=add, bn_reduce=red,
                                      zfast=rb.kvol == 27 and getattr(rb, "order", None) == ops.ROWS_YXZ)
            elif ctx.in_dtype == torch.bfloat16 and ops.pair_conv_usable(rb, ctx.cout, ctx.cin_pad):
                # strided conv of a z-fastest chain, 32 -> 16 channels backwards: one gather per indice pair
                dxp = ops.pair_conv(dy16, packed_d, None, rb, 1, ctx.cin_pad, ctx.in_dtype, addend=add, bn_reduce=red)
            elif USE_DGRAD_CLASSES and getattr(rb, "classes", None) is not None and ctx.cout >= 32 \
                    and (ctx.cout & (ctx.cout - 1)) == 0 and ctx.cin_pad % 16 == 0:
                # strided conv: rows grouped by parity class run only the 1..8 offsets they can use
                dxp = ops.dgrad_classes(dy16, packed_d, rb, ctx.cin_pad, ctx.in_dtype, addend=add, bn_reduce=red)
            else:
                dxp = ops.gather_gemm(dy16, packed_d, None, rb.nbr_in, rb.kvol, False, rb.n_in, ctx.cin_pad,
                                      ctx.in_dtype, n_dev=rb.n_in_dev, addend=add, bn_reduce=red)
            if red is not None:
                link.result = (dxp, red.partial, red.rows)   # dxp stays referenced: its address cannot be reused
            dx = dxp if ctx.cin_pad == ctx.in_cols else dxp[:, :ctx.in_cols].contiguous()
            if d_ident is not None and add is None:
                dx = dx + d_ident.to(dx.dtype)
        want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        if OVERLAP_WGRAD and want_w and ctx.needs_input_grad[0]:
            side = _side_stream(dy16.device)
            side.wait_event(ready)                   # dy16 is complete (recorded BEFORE the dgrad was issued)
        bias_p = ctx.bias_param
        weight_p = ctx.weight_param
        direct_w = DIRECT_GRAD and weight_p is not None and weight_p.grad is not None \
            and weight_p.grad.dtype == torch.float32 and weight_p.grad.is_contiguous()
        direct_b = DIRECT_GRAD and bias_p is not None and bias_p.grad is not None \
            and bias_p.grad.dtype == torch.float32 and bias_p.grad.is_contiguous()
        if direct_w and ctx.needs_input_grad[1]:
            _claim_direct(weight_p, "w")
        if direct_b and ctx.has_bias and ctx.needs_input_grad[2]:
            _claim_direct(bias_p, "b")
        keep_partial = None
        deferred = (WGRAD_JOIN_LAG > 0 and side is not None and (direct_w or not ctx.needs_input_grad[1])
                    and (direct_b or not (ctx.has_bias and ctx.needs_input_grad[2])))
        if ops.STAMPS is not None and ops.STAMPS.get("sparse"):
            _SP_SEQ[0] += 1
            ops.stamp(f"dg{_SP_SEQ[0]}e")                 # main stream: the data gradient of layer (backward order) is behind us
        with torch.cuda.stream(side) if side is not None else _NullCtx():
            if ops.STAMPS is not None and ops.STAMPS.get("sparse"):
                ops.stamp(f"wg{_SP_SEQ[0]}s")             # side stream: its weight gradient starts
            if ctx.needs_input_grad[1]:
                if ctx.window and rb.subm and ctx.cin <= ctx.cout == x.shape[1] and _window_wgrad(ctx.cout):
                    dwk = ops.subm_window_wgrad(x, dy16, rb, out=weight_p.grad if direct_w else None,
                                                defer=_WGRAD_JOBS if (deferred and direct_w) else None, cin=ctx.cin)
                else:
                    dwk = ops.wgrad(x, ctx.cin, dy16, None, None, rb.kvol,
                                    out=weight_p.grad if direct_w else None,                 # [Cout, K, Cin] f32
                                    defer=_WGRAD_JOBS if (deferred and direct_w) else None, rb=rb)
                dw = None if direct_w else dwk.view(weight.shape).to(weight.dtype)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                cl = ctx.colsum_link.result if ctx.colsum_link is not None else None
                if cl is not None and cl[0].data_ptr() == dy16.data_ptr() and cl[0].shape == dy16.shape:
                    # the BatchNorm backward that produced dy already summed its columns per workgroup
                    if deferred and direct_b:
                        _COLSUM_JOBS.append((cl[1], cl[2], bias_p.grad))     # finished in join_deferred_wgrad()
                        db = bias_p.grad
                    else:
                        db = ops.col_sum_finalize(cl[1], cl[2], out=bias_p.grad if direct_b else None)
                    keep_partial = cl[1]
                else:
                    db = ops.col_sum(dy16, n_dev=rb.n_out_dev, out=bias_p.grad if direct_b else None)
                if ctx.colsum_link is not None:
                    ctx.colsum_link.result = None
                if direct_b:
                    db = None
            if deferred and WGRAD_FLUSH_BYTES > 0 and sum(j[0].numel() for j in _WGRAD_JOBS) >= WGRAD_FLUSH_BYTES:
                # the slabs collected so far are summed NOW, on this (side) stream, in the middle of the backward pass: one
                # reduction of every layer at the very end read ~170 MB of 128-channel tiles on the step's critical tail
                ops.wgrad_reduce_batched(_WGRAD_JOBS)
                _WGRAD_KEEP.extend(_WGRAD_JOBS)          # (buffers stay referenced until the step's join)
                del _WGRAD_JOBS[:]
            if deferred:
                ev = torch.cuda.Event()
                ev.record(side)
        if deferred:
            _PENDING.append((ev, x, dy16, keep_partial))   # inputs stay alive until the lagged join below
            side = None
        if deferred and len(_PENDING) > WGRAD_JOIN_LAG:
            cur.wait_event(_PENDING[-1 - WGRAD_JOIN_LAG][0])
            del _PENDING[:len(_PENDING) - WGRAD_JOIN_LAG]
        if side is not None:
            cur.wait_stream(side)                    # join: dW / dbias are consumed on the current stream
            for t in (dw, db):
                if t is not None:
                    t.record_stream(cur)
        if dx is None and d_ident is not None:
            dx = d_ident
        return dx, dw, db, None, None, None, None, None, None, None


class BevDenseFunction(Function):
    """SparseConvTensor.dense() fused with the [B, C*D, H, W] view; backward = gather."""

    @staticmethod
    def forward(ctx, features, indices, batch_size, spatial_shape, n_dev=None, channels_last=False):
        f = features.detach().contiguous()
        if f.dtype not in (torch.float32, torch.bfloat16):
            f = f.float()
        pad = (-f.shape[1]) % (4 if f.dtype == torch.float32 else 8)
        c = f.shape[1]
        if pad:
            f = torch.nn.functional.pad(f, (0, pad))
        nhwc = bool(channels_last) and pad == 0
        out = ops.bev_scatter(f, indices, batch_size, spatial_shape, channels=c, n_dev=n_dev, channels_last=nhwc)
        ctx.meta = (indices, batch_size, list(spatial_shape), c, n_dev, nhwc)
        return out

    @staticmethod
    def backward(ctx, dout):
        indices, batch_size, spatial_shape, c, n_dev, nhwc = ctx.meta
        df = ops.bev_gather(dout, indices, batch_size, spatial_shape, c, n_dev=n_dev, channels_last=nhwc)
        return df, None, None, None, None, None


def bev_dense(features, indices, batch_size, spatial_shape, n_dev=None, channels_last=False):
    return BevDenseFunction.apply(features, indices, batch_size, spatial_shape, n_dev, channels_last)


EXACT_FP32 = False     # fp32 features -> the fp32-exact kernels (v_mfma_f32_16x16x4_f32) instead of bf16 operands


class SparseConvExactFunction(Function):
    """The same op in fp32 end to end (pcd_sparse_conv_gather_gemm_f32 / _wgrad_f32): what the reference computes
    (fp32 weights and features through spconv).  For parity work -- a whole backbone against an fp64 chain at 1e-3,
    reference checkpoints' activations -- not for speed."""

    @staticmethod
    def forward(ctx, features, weight, bias, rb, passthrough=False):
        x = features.detach().float().contiguous()
        cin = weight.shape[-1]
        if x.shape[1] != cin:
            x = x[:, :cin].contiguous()
        w = weight.detach().float().reshape(weight.shape[0], -1, cin).contiguous()    # [c_out, K, c_in]
        y = ops.gather_gemm_f32(x, w, bias, rb.nbr_out, rb.kvol, False, rb.n_out, n_dev=rb.n_out_dev)
        ctx.rb, ctx.has_bias, ctx.passthrough, ctx.in_cols = rb, bias is not None, passthrough, features.shape[1]
        ctx.wshape = weight.shape
        ctx.save_for_backward(x, w)
        if passthrough:
            return y, features.view_as(features)
        return y

    @staticmethod
    def backward(ctx, dy, d_ident=None):
        x, w = ctx.saved_tensors
        rb = ctx.rb
        if dy is None:
            return (d_ident, None, None, None, None)
        dy = dy.float().contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = w.permute(2, 1, 0).contiguous()                  # [c_in, K, c_out]
            add = d_ident.float().contiguous() if (d_ident is not None and ctx.in_cols == w.shape[2]) else None
            if rb.subm:
                dx = ops.gather_gemm_f32(dy, wt, None, rb.nbr_out, rb.kvol, True, rb.n_in, n_dev=rb.n_in_dev, addend=add)
            else:
                dx = ops.gather_gemm_f32(dy, wt, None, rb.nbr_in, rb.kvol, False, rb.n_in, n_dev=rb.n_in_dev, addend=add)
            if d_ident is not None and add is None:
                dx = dx + d_ident.float()[:, :dx.shape[1]]
            if ctx.in_cols != dx.shape[1]:
                dx = torch.nn.functional.pad(dx, (0, ctx.in_cols - dx.shape[1]))
        elif d_ident is not None:
            dx = d_ident
        if ctx.needs_input_grad[1]:
            dw = ops.wgrad_f32(x, dy, rb.pairs, rb.pair_num, rb.kvol).view(ctx.wshape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.col_sum(dy)
        return (dx, dw, db, None, None)


def sparse_conv(features, weight, bias, rb, packed_fwd, packed_dgrad=None, passthrough=False, fp8=None, window=False):
    if EXACT_FP32 and features.dtype == torch.float32:
        return SparseConvExactFunction.apply(features, weight, bias, rb, passthrough)
    return SparseConvFunction.apply(features, weight, bias, rb, packed_fwd, packed_dgrad, passthrough, fp8, window)


class _ColsumLink:
    """conv -> BatchNorm behind it: "sum the columns of the dx you produce"; result = (dx, partial, rows)."""
    __slots__ = ("result",)

    def __init__(self):
        self.result = None


class _BnLink:
    """What the data-gradient kernel of the NEXT conv needs to take this BatchNorm's backward reductions
    (attached to the BatchNorm output as `_pcd_bn_link`; holds no reference to that output)."""
    __slots__ = ("relu", "x", "mean", "invstd", "result")

    def __init__(self, relu, x, mean, invstd):
        self.relu, self.x, self.mean, self.invstd = relu, x, mean, invstd
        self.result = None     # (dx tensor, partial [rows, 2, c], rows) left by the data-gradient kernel


class FusedBNFunction(Function):
    """nn.BatchNorm1d (+ residual add) (+ nn.ReLU) on `.features` in two streaming passes
    (spconv_backbone.py:21-25,50-66); parameters / buffers stay in the caller's nn.BatchNorm1d."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, bn, relu, n_dev=None, out=None):
        # out: a [n, c] column block of a wider matrix that receives y (the concatenated map of BaseBEVBackbone)
        xc = x.detach().contiguous()
        rc = residual.detach().contiguous() if residual is not None else None
        if rc is not None and rc.dtype != xc.dtype:
            rc = rc.to(xc.dtype)
        training = bn.training or bn.running_mean is None
        momentum = bn.momentum if bn.momentum is not None else 0.1
        g = gamma.detach().float() if gamma is not None else None
        b = beta.detach().float() if beta is not None else None
        partials = None
        st = getattr(x, "_pcd_stats", None) if (FUSE_BN_REDUCTIONS and training) else None
        if st is not None and st.partial is not None and xc.data_ptr() == x.data_ptr() \
                and xc.dtype == torch.bfloat16 and st.partial.shape[2] == xc.shape[1]:
            partials = (st.partial, st.rows)         # the conv that produced x already summed its columns
        y, save_mean, save_invstd = ops.bn_forward(xc, rc, g, b, bn.eps, momentum, training, bn.running_mean,
                                                   bn.running_var, relu, n_dev=n_dev, partials=partials, out=out)
        if not training:
            save_mean = bn.running_mean
            save_invstd = torch.rsqrt(bn.running_var + bn.eps)
        ctx.relu, ctx.training, ctx.has_res = relu, training, residual is not None
        ctx.n_dev = n_dev
        ctx.bn = bn
        # without a residual the ReLU mask is recomputed from x in the backward (one [n][c] read less per pass)
        ctx.mask_from_x = bool(relu and residual is None and training and b is not None)
        if out is not None and not ctx.mask_from_x:
            raise RuntimeError("FusedBNFunction(out=...): only for the training form without a residual")
        ctx.save_for_backward(xc, None if ctx.mask_from_x else y, g, b, save_mean, save_invstd)
        cl = getattr(x, "_pcd_colsum_link", None) if FUSE_BN_REDUCTIONS else None
        ctx.colsum_link = cl if (cl is not None and xc.data_ptr() == x.data_ptr() and training) else None
        ctx.link = None
        if FUSE_BN_REDUCTIONS and training and xc.dtype == torch.bfloat16:
            ctx.link = _BnLink(bool(relu), xc, save_mean, save_invstd)
            y._pcd_bn_link = ctx.link
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, g, b, save_mean, save_invstd = ctx.saved_tensors
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        bn = ctx.bn
        gw = bn.weight.grad if (DIRECT_GRAD and bn.weight is not None and ctx.needs_input_grad[1]) else None
        gb = bn.bias.grad if (DIRECT_GRAD and bn.bias is not None and ctx.needs_input_grad[2]) else None
        direct = gw is not None and gb is not None and ops._usable_out(gw, x.shape[1]) \
            and ops._usable_out(gb, x.shape[1])
        partials = None
        link = ctx.link
        if link is not None and link.result is not None:
            rdx, part, rows = link.result
            link.result = None
            if rdx.data_ptr() == dy.data_ptr() and rdx.shape == dy.shape and dy.dtype == x.dtype \
                    and dy.is_contiguous():
                partials = (part, rows)              # dy IS the tensor whose kernel took the reductions
        res = ops.bn_backward(dy, x, y, g, save_mean, save_invstd, ctx.relu, ctx.training,
                              ctx.has_res and ctx.needs_input_grad[3], n_dev=ctx.n_dev,
                              dgamma_out=gw if direct else None, dbeta_out=gb if direct else None,
                              beta=b if ctx.mask_from_x else None, partials=partials,
                              colsum=ctx.colsum_link is not None)
        dx, dres, dgamma, dbeta = res[:4]
        if ctx.colsum_link is not None:
            ctx.colsum_link.result = (dx,) + res[4]
        if direct:
            dgamma = dbeta = None
        return (dx, dgamma if ctx.needs_input_grad[1] else None, dbeta if ctx.needs_input_grad[2] else None,
                dres, None, None, None, None)


def _fusable(bn, x):
    import torch.nn as nn
    if not isinstance(bn, nn.modules.batchnorm._BatchNorm) or not x.is_cuda or x.dim() != 2:   # (1d; 2d via hotpath.conv2d_fast)
        return False
    if isinstance(bn, nn.SyncBatchNorm) and bn.training:
        # --sync_bn (tools/train.py:34,134-135: convert_sync_batchnorm): statistics over ALL ranks.  The fused kernels take
        # per-rank statistics only, so a SyncBatchNorm in training mode goes through torch's own module (its all_gather of
        # the per-rank sums included) -- correct, unfused, eager only; in eval mode it is an ordinary BatchNorm.
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return False
    if x.dtype not in (torch.float32, torch.bfloat16):
        return False
    c = x.shape[1]
    piece = 4 if x.dtype == torch.float32 else 8
    if c % piece or c > 1024:
        return False
    pcs = c // piece
    return pcs <= 256 and (bn.running_mean is not None or bn.training)


def batch_norm_act(bn, x, residual=None, relu=True, n_dev=None, out=None):
    """y = relu?(bn(x) + residual?) through the fused HIP kernels when the shape allows, else torch.
    out (fused path only): column block of a wider matrix to write y into."""
    if _fusable(bn, x):
        if bn.training and bn.num_batches_tracked is not None and not getattr(bn, "_defer_nbt", False):
            bn.num_batches_tracked.add_(1)
        return FusedBNFunction.apply(x, bn.weight, bn.bias, residual, bn, relu, n_dev, out)
    assert out is None
    if n_dev is not None:
        raise RuntimeError("static-shape mode needs the fused BatchNorm path (unsupported channel count / dtype)")
    y = bn(x)
    if residual is not None:
        y = y + residual
    return torch.relu(y) if relu else y
