"""CenterHead losses (pcdet/models/dense_heads/center_head.py:226-262 `get_loss`, pcdet/utils/loss_utils.py:611-643
`neg_loss_cornernet`, :1317-1345 `_reg_loss`, :1348-1362 `_gather_feat` / `_transpose_and_gather_feat`, :1364-1390
`RegLossCenterNet`) with the reference's arithmetic and WITHOUT its host round trips: the reference branches on
`num_pos == 0` (a device tensor in an `if`: one synchronisation per head and step) and fills `tb_dict` with
`.item()` (two more per head); here the branch is `clamp_min(num_pos, 1)` -- identical, because the positive term is
exactly zero when there is no positive -- and the logged scalars stay device tensors.  Nothing in this file blocks
the stream, so the whole loss is capturable in a hipGraph behind `assign_targets` (centerhead.hip)."""
import torch




def _like(t):
    """an uninitialised tensor with EXACTLY t's strides (torch.empty_like densifies a non-dense view, e.g. the 1-3 real
    channels of a prediction map that was computed with zero-padded channels)."""
    return torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device)


def sigmoid_clamped(x):
    """center_head.py:226-228"""
    return torch.clamp(x.sigmoid(), min=1e-4, max=1 - 1e-4)


def neg_loss_cornernet(pred, gt, mask=None):
    """loss_utils.py:611-643: (loss, mean confidence at the positives); pred / gt [B, C, H, W], mask [B, H, W]."""
    pos_inds = gt.eq(1).float()
    neg_inds = gt.lt(1).float()
    neg_weights = torch.pow(1 - gt, 4)
    pos_loss = torch.log(pred) * torch.pow(1 - pred, 2) * pos_inds
    neg_loss = torch.log(1 - pred) * torch.pow(pred, 2) * neg_weights * neg_inds
    if mask is not None:
        mask = mask[:, None, :, :].float()
        pos_loss = pos_loss * mask
        neg_loss = neg_loss * mask
        num_pos = (pos_inds * mask).sum()
    else:
        num_pos = pos_inds.sum()
    pos_loss = pos_loss.sum()
    neg_loss = neg_loss.sum()
    confidence = (pred * pos_inds).sum() / num_pos          # (nan without positives, as in the reference)
    # reference: `if num_pos == 0: -neg_loss else: -(pos_loss + neg_loss) / num_pos`; pos_loss == 0 when num_pos == 0
    loss = -(pos_loss + neg_loss) / torch.clamp_min(num_pos, 1.0)
    return loss, confidence


def _gather_feat(feat, ind):
    dim = feat.size(2)
    ind = ind.unsqueeze(2).expand(ind.size(0), ind.size(1), dim)
    return feat.gather(1, ind)


def _transpose_and_gather_feat(feat, ind):
    feat = feat.permute(0, 2, 3, 1).contiguous()
    feat = feat.view(feat.size(0), -1, feat.size(3))
    return _gather_feat(feat, ind)


def reg_loss(output, mask, ind, target):
    """RegLossCenterNet.forward (loss_utils.py:1364-1390): L1 per code dimension over the masked objects -> [dim]."""
    pred = _transpose_and_gather_feat(output, ind)
    num = mask.float().sum()
    m = mask.unsqueeze(2).expand_as(target).float()
    loss = torch.abs(pred * m - target * m)
    loss = loss.transpose(2, 0)
    loss = torch.sum(loss, dim=2)
    loss = torch.sum(loss, dim=1)
    return loss / torch.clamp_min(num, min=1.0)


class CenterHeadLoss(torch.nn.Module):
    """`CenterHead.get_loss` (center_head.py:230-262): sum over the heads of cls_weight * focal(hm) + loc_weight *
    sum(code_weights * L1).  Returns (loss, tb) with tb holding DEVICE scalars (read them when you log)."""

    def __init__(self, head_order, cls_weight=1.0, loc_weight=2.0, code_weights=(1.0,) * 8):
        super().__init__()
        self.head_order = list(head_order)
        self.cls_weight, self.loc_weight = float(cls_weight), float(loc_weight)
        self.register_buffer("code_weights", torch.tensor(code_weights, dtype=torch.float32), persistent=False)

    def forward(self, pred_dicts, target_dicts):
        tb, loss, confidence = {}, 0, 0
        for idx, pred in enumerate(pred_dicts):
            hm = sigmoid_clamped(pred['hm'].float())
            hm_loss, conf = neg_loss_cornernet(hm, target_dicts['heatmaps'][idx])
            hm_loss = hm_loss * self.cls_weight
            boxes = torch.cat([pred[name].float() for name in self.head_order], dim=1)
            rl = reg_loss(boxes, target_dicts['masks'][idx], target_dicts['inds'][idx], target_dicts['target_boxes'][idx])
            loc_loss = (rl * self.code_weights[:rl.shape[0]].to(rl.device)).sum() * self.loc_weight
            loss = loss + hm_loss + loc_loss
            tb['hm_loss_head_%d' % idx] = hm_loss.detach()
            tb['loc_loss_head_%d' % idx] = loc_loss.detach()
            confidence = confidence + conf.detach()
        tb['rpn_loss'] = loss.detach()
        tb['confidence'] = confidence / len(pred_dicts)
        return loss, tb


# ---------------------------------------------------------------------------------------------------------------
class _FusedHeadLoss(torch.autograd.Function):
    """One head of `get_loss` through pcd_centerhead_loss_forward / _backward (centerhead.hip): 2 + 2 launches."""

    @staticmethod
    def forward(ctx, hm, heatmap, inds, masks, target_boxes, code_weights, cls_weight, loc_weight, *regs):
        import ctypes
        from .. import _lib as L
        assert hm.is_cuda and hm.dim() == 4 and heatmap.dtype == torch.float32 and heatmap.is_contiguous()
        assert inds.dtype == torch.int64 and masks.dtype == torch.int64 and inds.is_contiguous() and masks.is_contiguous()
        assert target_boxes.dtype == torch.float32 and target_boxes.is_contiguous()
        B, C, H, W = hm.shape
        dims = sum(int(r.shape[1]) for r in regs)
        assert dims == target_boxes.shape[2] and all(r.dtype == regs[0].dtype for r in regs)
        lib = L.lib()
        code_weights = code_weights[:dims].to(device=hm.device, dtype=torch.float32).contiguous()
        out = torch.empty((6 + dims,), dtype=torch.float32, device=hm.device)
        ws = torch.empty((int(lib.pcd_centerhead_loss_workspace_bytes(dims)),), dtype=torch.uint8, device=hm.device)

        def dt(t):
            assert t.dtype in (torch.float32, torch.bfloat16), t.dtype
            return L.PCD_F32 if t.dtype == torch.float32 else L.PCD_BF16
        geo = dict(hm_strides=(ctypes.c_longlong * 4)(*hm.stride()),
                   reg_ptrs=(ctypes.c_void_p * len(regs))(*[r.data_ptr() for r in regs]),
                   reg_ch=(ctypes.c_int * len(regs))(*[int(r.shape[1]) for r in regs]),
                   reg_strides=(ctypes.c_longlong * (4 * len(regs)))(*[v for r in regs for v in r.stride()]))
        L.check(lib.pcd_centerhead_loss_forward(
            L.ptr(hm), dt(hm), geo["hm_strides"], L.ptr(heatmap), B, C, H, W, geo["reg_ptrs"], geo["reg_ch"],
            dt(regs[0]) if regs else L.PCD_F32, geo["reg_strides"], len(regs), L.ptr(inds), L.ptr(masks),
            L.ptr(target_boxes), int(inds.shape[1]), L.ptr(code_weights), float(cls_weight), float(loc_weight),
            L.ptr(out), L.ptr(ws), ws.numel(), L.stream_ptr()), "pcd_centerhead_loss_forward")
        ctx.save_for_backward(hm, heatmap, inds, masks, target_boxes, code_weights, out, *regs)
        ctx.weights = (float(cls_weight), float(loc_weight))
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        import ctypes
        from .. import _lib as L
        hm, heatmap, inds, masks, target_boxes, code_weights, out, *regs = ctx.saved_tensors
        B, C, H, W = hm.shape
        lib = L.lib()
        d_hm = _like(hm)                      # (same strides: the kernels address gradients like the inputs)
        d_regs = [_like(r) for r in regs]
        assert d_hm.stride() == hm.stride() and all(d.stride() == r.stride() for d, r in zip(d_regs, regs))
        g = g_loss.detach().to(torch.float32).reshape(1).contiguous()

        def dt(t):
            return L.PCD_F32 if t.dtype == torch.float32 else L.PCD_BF16
        L.check(lib.pcd_centerhead_loss_backward(
            L.ptr(hm), L.ptr(d_hm), dt(hm), (ctypes.c_longlong * 4)(*hm.stride()), L.ptr(heatmap), B, C, H, W,
            (ctypes.c_void_p * len(regs))(*[r.data_ptr() for r in regs]),
            (ctypes.c_void_p * len(regs))(*[d.data_ptr() for d in d_regs]),
            (ctypes.c_int * len(regs))(*[int(r.shape[1]) for r in regs]), dt(regs[0]) if regs else L.PCD_F32,
            (ctypes.c_longlong * (4 * len(regs)))(*[v for r in regs for v in r.stride()]), len(regs), L.ptr(inds),
            L.ptr(masks), L.ptr(target_boxes), int(inds.shape[1]), L.ptr(code_weights), ctx.weights[0], ctx.weights[1],
            L.ptr(out), L.ptr(g), L.stream_ptr()), "pcd_centerhead_loss_backward")
        return (d_hm, None, None, None, None, None, None, None, *d_regs)


class FusedCenterHeadLoss(CenterHeadLoss):
    """CenterHeadLoss with every head's loss and gradients in four HIP launches (no elementwise chain)."""

    def forward(self, pred_dicts, target_dicts):
        tb, loss, confidence = {}, 0, 0
        for idx, pred in enumerate(pred_dicts):
            regs = [pred[name] for name in self.head_order]
            head_loss, out = _FusedHeadLoss.apply(pred['hm'], target_dicts['heatmaps'][idx], target_dicts['inds'][idx],
                                                  target_dicts['masks'][idx], target_dicts['target_boxes'][idx],
                                                  self.code_weights, self.cls_weight, self.loc_weight, *regs)
            loss = loss + head_loss if idx else head_loss
            tb['hm_loss_head_%d' % idx] = out[1]
            tb['loc_loss_head_%d' % idx] = out[2]
            confidence = confidence + out[3] if idx else out[3]
        tb['rpn_loss'] = loss.detach()
        tb['confidence'] = confidence / len(pred_dicts)
        return loss, tb
