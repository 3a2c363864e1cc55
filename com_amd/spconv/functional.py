"""autograd.Function wrappers over the HIP kernels (binder convention of the reference's own ops:
pcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py:8-50)."""
import torch
from torch.autograd import Function

from .. import ops


def _to_bf16_padded(x, c_pad):
    """[N, C] (f32 or bf16) -> contiguous bf16 [N, c_pad], zero padded channels."""
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    if x.shape[1] != c_pad:
        x = torch.nn.functional.pad(x, (0, c_pad - x.shape[1]))
    return x.contiguous()


class SparseConvFunction(Function):
    """y = indice_conv(features, weight, rulebook) with bf16 MFMA, fp32 accumulate.

    forward : output-stationary gather-GEMM over rb.nbr_out
    backward: dX = gather-GEMM over the input-stationary view with W^T; dW = pair-wise X^T dY.
    `packed` = (packed_fwd, packed_dgrad or None) prepared by the module (cached per weight version).
    """

    @staticmethod
    def forward(ctx, features, weight, bias, rb, packed_fwd):
        cout, cin = weight.shape[0], weight.shape[-1]
        assert features.shape[1] == cin, (features.shape, weight.shape)
        cin_pad = ops.pow2_ge8(cin)
        x = _to_bf16_padded(features.detach(), cin_pad)
        out_dtype = features.dtype if features.dtype in (torch.float32, torch.bfloat16) else torch.float32
        b = bias.detach().float().contiguous() if bias is not None else None
        y = ops.gather_gemm(x, packed_fwd, b, rb.nbr_out, rb.kvol, False, rb.n_out, cout, out_dtype)
        ctx.rb = rb
        ctx.cin, ctx.cout, ctx.cin_pad = cin, cout, cin_pad
        ctx.has_bias = bias is not None
        ctx.in_dtype = features.dtype
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        rb = ctx.rb
        dy16 = _to_bf16_padded(dy, ctx.cout)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if ctx.cin_pad % 16 != 0:
                raise RuntimeError("dgrad needs >= 16 input channels (the 5-channel input layer never "
                                   "requires an input gradient)")
            packed_d = ops.pack_weight(weight, 1)
            if rb.subm:
                dxp = ops.gather_gemm(dy16, packed_d, None, rb.nbr_out, rb.kvol, True, rb.n_in, ctx.cin_pad,
                                      ctx.in_dtype)
            else:
                dxp = ops.gather_gemm(dy16, packed_d, None, rb.nbr_in, rb.kvol, False, rb.n_in, ctx.cin_pad,
                                      ctx.in_dtype)
            dx = dxp if ctx.cin_pad == ctx.cin else dxp[:, :ctx.cin].contiguous()
        if ctx.needs_input_grad[1]:
            dwk = ops.wgrad(x, ctx.cin, dy16, rb.pairs, rb.pair_num, rb.kvol)      # [Cout, K, Cin] f32
            dw = dwk.view(weight.shape).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.float().sum(0)
        return dx, dw, db, None, None


class BevDenseFunction(Function):
    """SparseConvTensor.dense() fused with the [B, C*D, H, W] view; backward = gather."""

    @staticmethod
    def forward(ctx, features, indices, batch_size, spatial_shape):
        f = features.detach().contiguous()
        if f.dtype not in (torch.float32, torch.bfloat16):
            f = f.float()
        pad = (-f.shape[1]) % (4 if f.dtype == torch.float32 else 8)
        c = f.shape[1]
        if pad:
            f = torch.nn.functional.pad(f, (0, pad))
        out = ops.bev_scatter(f, indices, batch_size, spatial_shape, channels=c)
        ctx.meta = (indices, batch_size, list(spatial_shape), c)
        return out

    @staticmethod
    def backward(ctx, dout):
        indices, batch_size, spatial_shape, c = ctx.meta
        df = ops.bev_gather(dout, indices, batch_size, spatial_shape, c)
        return df, None, None, None


def bev_dense(features, indices, batch_size, spatial_shape):
    return BevDenseFunction.apply(features, indices, batch_size, spatial_shape)


def sparse_conv(features, weight, bias, rb, packed_fwd):
    return SparseConvFunction.apply(features, weight, bias, rb, packed_fwd)
