"""The COM curriculum head's per-step work on the device (BASELINE config 3).

Mirrors, with the reference's names and return shapes:
  * `cluster`               CurriculumCenterHead.cluster                 pcdet/models/dense_heads/curriculum_center_head.py:414-459
  * `assign_targets`        CurriculumCenterHead.assign_targets          same file :206-307 (+ :108-204)
  * `FocalLossCenterCurriculumState` / `CurriculumCenterHeadLoss`
                            FocalLossCenterCurriculum + get_loss          pcdet/utils/loss_utils.py:998-1310; head :309-358
through pcd_com_* (com_head.hip).  The reference runs these with Python loops over objects and groups, `.item()` per
object and 288 `torch.where` calls per step; here nothing blocks the stream, so the head can live inside the captured
step.  Logged scalars (`tb_dict`) and the `(3, 96)` confidence tensors stay DEVICE tensors -- read them when you log.

There is no CPU fallback: CPU tensors raise."""
import ctypes

import torch

from .. import _lib as L




def _like(t):
    """an uninitialised tensor with EXACTLY t's strides (torch.empty_like densifies a non-dense view, e.g. the 1-3 real
    channels of a prediction map that was computed with zero-padded channels)."""
    return torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device)


def _dev(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise L.PcdError(f"{what} needs a HIP device tensor (there is no CPU fallback)")
    return t


def cluster(gt_boxes, true_object, occupancy_ratio, facade_type):
    """[B, M] int64 difficulty groups (cars 1..96, pedestrians / cyclists 1..15, 0 = padding or not a real object).
    true_object=None (the reference would fail on `None == 1`) is rejected."""
    gt = _dev(gt_boxes, "cluster").contiguous().float()
    B, M, code = gt.shape
    if true_object is None:
        raise L.PcdError("cluster needs data_dict['true_object'] (COMAug's marker of real vs pasted objects)")
    to, occ, fac = (_dev(t, "cluster").contiguous().float() for t in (true_object, occupancy_ratio, facade_type))
    assert to.shape == (B, M) and occ.shape == (B, M) and fac.shape == (B, M)
    group = torch.empty((B, M), dtype=torch.int64, device=gt.device)
    L.check(L.lib().pcd_com_cluster_groups(L.ptr(gt), B, M, code, L.ptr(to), L.ptr(occ), L.ptr(fac),
                                           L.PCD_COM_CLUSTER_X5, L.ptr(group), L.stream_ptr()), "pcd_com_cluster_groups")
    return group


def assign_targets(gt_boxes, feature_map_size, class_names, class_names_each_head, point_cloud_range, voxel_size,
                   feature_map_stride, npgt, true_object=None, num_max_objs=500, gaussian_overlap=0.1, min_radius=2,
                   epoch=0, epoch_threshold=100, min_points=1):
    """The reference's `ret_dict`: lists over heads of heatmaps [B, C, H, W], target_boxes [B, n, code], inds [B, n]
    int64, masks [B, n] FLOAT, radius_map [B, n, 5] int64 (class, cx, cy, radius, group; 4 columns when `true_object`
    -- the group tensor -- is None), heatmap_mask [B, C, H, W] ones.  feature_map_size = [H, W].
    Every head filters on the ORIGINAL class ids (the reference rewrites gt_boxes' class column in place while
    filtering, curriculum_center_head.py:260, which later heads then see; identical for the single-head COM configs)."""
    gt = _dev(gt_boxes, "assign_targets").contiguous().float()
    B, M, code = gt.shape
    H, W = int(feature_map_size[0]), int(feature_map_size[1])
    npgt = _dev(npgt, "assign_targets").contiguous().float()
    assert npgt.shape == (B, M), "gt_boxes.shape[:-1] == npgt.shape (curriculum_center_head.py:236)"
    group = None
    if true_object is not None:
        group = _dev(true_object, "assign_targets").contiguous().to(torch.int64)
        assert group.shape == (B, M)
    cols = 5 if group is not None else 4
    lib = L.lib()
    ws = torch.empty((max(int(lib.pcd_com_assign_workspace_bytes(B, num_max_objs)), 256),), dtype=torch.uint8,
                     device=gt.device)
    ret = {'heatmaps': [], 'target_boxes': [], 'inds': [], 'masks': [], 'heatmap_masks': [], 'radius_map': [],
           'heatmap_mask': []}
    gate = int(epoch <= epoch_threshold)
    for head_names in class_names_each_head:
        cmap = [0] * (len(class_names) + 1)
        for i, name in enumerate(class_names):
            if name in head_names:
                cmap[i + 1] = list(head_names).index(name) + 1
        nc = len(head_names)
        dev = gt.device
        heatmap = torch.empty((B, nc, H, W), dtype=torch.float32, device=dev)
        hmask = torch.empty((B, nc, H, W), dtype=torch.float32, device=dev)
        boxes = torch.empty((B, num_max_objs, code), dtype=torch.float32, device=dev)
        inds = torch.empty((B, num_max_objs), dtype=torch.int64, device=dev)
        mask = torch.empty((B, num_max_objs), dtype=torch.float32, device=dev)
        rmap = torch.empty((B, num_max_objs, cols), dtype=torch.int64, device=dev)
        L.check(lib.pcd_com_assign_targets(
            L.ptr(gt), B, M, code, L.host_i32(cmap), len(cmap), nc, W, H, int(feature_map_stride),
            L.host_f32([voxel_size[0], voxel_size[1]]), L.host_f32([point_cloud_range[0], point_cloud_range[1]]),
            int(num_max_objs), float(gaussian_overlap), int(min_radius), L.ptr(npgt), L.ptr(group), gate,
            float(min_points), L.ptr(heatmap), L.ptr(boxes), L.ptr(inds), L.ptr(mask), L.ptr(rmap), cols, L.ptr(hmask),
            L.ptr(ws), ws.numel(), L.stream_ptr()), "pcd_com_assign_targets")
        ret['heatmaps'].append(heatmap)
        ret['target_boxes'].append(boxes)
        ret['inds'].append(inds)
        ret['masks'].append(mask)
        ret['radius_map'].append(rmap)
        ret['heatmap_mask'].append(hmask)
    return ret


def curriculum_struct(cur, epoch, conf_shape):
    """LOSS_CURRICULUM dict -> PcdComCurriculum, with FocalLossCenterCurriculum.__init__'s defaults
    (loss_utils.py:1020-1054).  `THRESHOLD` is NOT read by the reference (self.threshold = 0.5)."""
    cur = dict(cur or {})
    g = cur.get
    c = L.PcdComCurriculum()
    c.ucl, c.fix_threshold = int(bool(g('UCL', True))), int(bool(g('FIX', False)))
    c.straight, c.tuning, c.only_center = int(bool(g('STRAIGHT', False))), int(bool(g('TUNING', False))), int(bool(g('CENTER', False)))
    c.apply = int(g('START', 0) <= epoch <= g('END', 30))
    c.add, c.radius = int(g('ADD', 0)), int(g('RADIUS', 0))
    c.k_straight, c.elongation, c.height = float(g('K', 1.0)), float(g('ELONGATION', -10)), float(g('HEIGHT', 1))
    c.alpha, c.threshold = float(g('ALPHA', 0.001)), 0.5
    c.conf_classes, c.conf_groups = (int(conf_shape[0]), int(conf_shape[1])) if conf_shape is not None else (0, 0)
    return c


class _ComHeadLoss(torch.autograd.Function):
    """One head of CurriculumCenterHead.get_loss: 2 launches forward in the shipped COM setting (UCL False), 4-6 with
    the per-object weights, 2 backward."""

    @staticmethod
    def forward(ctx, hm, heatmap, inds, box_mask, target_boxes, radius_map, heatmap_mask, owner, cur, state,
                conf_all, num_all, conf_epoch, num_epoch, code_weights, cls_weight, loc_weight, *regs):
        assert hm.is_cuda and hm.dim() == 4 and heatmap.dtype == torch.float32 and heatmap.is_contiguous()
        assert inds.dtype == torch.int64 and inds.is_contiguous() and radius_map.dtype == torch.int64
        assert box_mask.dtype == torch.float32 and box_mask.is_contiguous() and radius_map.is_contiguous()
        assert target_boxes.dtype == torch.float32 and target_boxes.is_contiguous()
        B, C, H, W = hm.shape
        n = int(inds.shape[1])
        dims = sum(int(r.shape[1]) for r in regs)
        assert dims == target_boxes.shape[2] and all(r.dtype == regs[0].dtype for r in regs)
        lib = L.lib()
        code_weights = code_weights[:dims].to(device=hm.device, dtype=torch.float32).contiguous()
        out = torch.empty((6 + dims,), dtype=torch.float32, device=hm.device)
        ws = torch.empty((int(lib.pcd_com_loss_workspace_bytes(B, n)),), dtype=torch.uint8, device=hm.device)
        msum = None
        if cur.ucl:
            assert heatmap_mask is not None and heatmap_mask.is_contiguous() and heatmap_mask.shape == hm.shape
            assert owner is not None and owner.dtype == torch.int32 and owner.numel() >= hm.numel()
            msum = torch.empty((C, H, W), dtype=torch.float32, device=hm.device)

        def dt(t):
            assert t.dtype in (torch.float32, torch.bfloat16), t.dtype
            return L.PCD_F32 if t.dtype == torch.float32 else L.PCD_BF16
        L.check(lib.pcd_com_loss_forward(
            L.ptr(hm), dt(hm), (ctypes.c_longlong * 4)(*hm.stride()), L.ptr(heatmap), B, C, H, W,
            (ctypes.c_void_p * len(regs))(*[r.data_ptr() for r in regs]),
            (ctypes.c_int * len(regs))(*[int(r.shape[1]) for r in regs]), dt(regs[0]) if regs else L.PCD_F32,
            (ctypes.c_longlong * (4 * len(regs)))(*[v for r in regs for v in r.stride()]), len(regs), L.ptr(inds),
            L.ptr(box_mask), L.ptr(target_boxes), L.ptr(radius_map), int(radius_map.shape[2]), n,
            L.ptr(heatmap_mask) if cur.ucl else None, L.ptr(owner) if cur.ucl else None, L.ptr(msum),
            ctypes.cast(ctypes.pointer(cur), ctypes.c_void_p), L.ptr(code_weights), float(cls_weight), float(loc_weight),
            L.ptr(state), L.ptr(out), L.ptr(conf_all), L.ptr(num_all), L.ptr(conf_epoch), L.ptr(num_epoch), L.ptr(ws),
            ws.numel(), L.stream_ptr()), "pcd_com_loss_forward")
        ctx.save_for_backward(hm, heatmap, inds, box_mask, target_boxes, code_weights, out, *regs)
        ctx.msum = msum
        ctx.weights = (float(cls_weight), float(loc_weight))
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        hm, heatmap, inds, box_mask, target_boxes, code_weights, out, *regs = ctx.saved_tensors
        B, C, H, W = hm.shape
        d_hm = _like(hm)
        d_regs = [_like(r) for r in regs]
        assert d_hm.stride() == hm.stride() and all(d.stride() == r.stride() for d, r in zip(d_regs, regs))
        g = g_loss.detach().to(torch.float32).reshape(1).contiguous()

        def dt(t):
            return L.PCD_F32 if t.dtype == torch.float32 else L.PCD_BF16
        L.check(L.lib().pcd_com_loss_backward(
            L.ptr(hm), L.ptr(d_hm), dt(hm), (ctypes.c_longlong * 4)(*hm.stride()), L.ptr(heatmap), B, C, H, W,
            (ctypes.c_void_p * len(regs))(*[r.data_ptr() for r in regs]),
            (ctypes.c_void_p * len(regs))(*[d.data_ptr() for d in d_regs]),
            (ctypes.c_int * len(regs))(*[int(r.shape[1]) for r in regs]), dt(regs[0]) if regs else L.PCD_F32,
            (ctypes.c_longlong * (4 * len(regs)))(*[v for r in regs for v in r.stride()]), len(regs), L.ptr(inds),
            L.ptr(box_mask), L.ptr(target_boxes), int(inds.shape[1]), L.ptr(ctx.msum), L.ptr(code_weights),
            ctx.weights[0], ctx.weights[1], L.ptr(out), L.ptr(g), L.stream_ptr()), "pcd_com_loss_backward")
        return (d_hm,) + (None,) * 16 + tuple(d_regs)


class FocalLossCenterCurriculumState:
    """What FocalLossCenterCurriculum keeps between steps (loss_utils.py:1022,1049,1184-1188,1214), on the device:
    `state` double[2] = {EMA of the average confidence, this step's average}, `confidence_all` = [sums, counts] of the
    last step (the attribute train_utils.py:111-112 reads), and the running epoch sums of both."""

    def __init__(self, conf_shape, device):
        self.conf_shape = tuple(conf_shape) if conf_shape is not None else None
        self.state = torch.zeros((2,), dtype=torch.float64, device=device)
        shape = self.conf_shape if self.conf_shape is not None else (0, 0)
        self.confidence_all = [torch.zeros(shape, dtype=torch.float32, device=device) for _ in range(2)]
        self.epoch_confidence = torch.zeros(shape, dtype=torch.float32, device=device)
        self.epoch_num = torch.zeros(shape, dtype=torch.float32, device=device)
        self._owner = None

    @property
    def avg_confidence(self):
        """self.avg_confidence of the reference (a host read: use it for logging only)."""
        return float(self.state[0].item())

    def owner(self, like):
        if self._owner is None or self._owner.numel() < like.numel() or self._owner.device != like.device:
            if torch.cuda.is_current_stream_capturing():
                raise L.PcdError("COM loss (UCL): run one eager step before graph capture (the owner plane is allocated once)")
            self._owner = torch.zeros((like.numel(),), dtype=torch.int32, device=like.device)   # zeroed once; self-cleaning
        return self._owner

    def start_epoch(self):
        """train_utils.py:57-58: the per-epoch lists start empty."""
        self.epoch_confidence.zero_()
        self.epoch_num.zero_()


class CurriculumCenterHeadLoss(torch.nn.Module):
    """`CurriculumCenterHead.get_loss` (curriculum_center_head.py:313-358) for all heads: returns (loss, tb_dict) with
    tb_dict holding DEVICE scalars.  `curriculum` = MODEL.DENSE_HEAD.LOSS_CURRICULUM; conf_shape=(3, 96) is
    CurriculumCenterHead_x5 (head_zoo.py:145-149), None the base class.  As in the reference ONE loss module (one state)
    serves all heads, in head order."""

    def __init__(self, head_order, curriculum, conf_shape=(3, 96), cls_weight=1.0, loc_weight=2.0,
                 code_weights=(1.0,) * 8):
        super().__init__()
        self.head_order = list(head_order)
        self.curriculum = dict(curriculum or {})
        self.conf_shape = tuple(conf_shape) if conf_shape is not None else None
        self.cls_weight, self.loc_weight = float(cls_weight), float(loc_weight)
        self.register_buffer("code_weights", torch.tensor(code_weights, dtype=torch.float32), persistent=False)
        self.hm_loss_func = None          # FocalLossCenterCurriculumState, created on the first forward's device

    def init_state(self, device):
        """Create the device-side state (EMA, (3, 96) tensors, epoch sums).  Done by the first forward; call it yourself
        before capturing the first step into a hipGraph: state allocated during a capture would live in the graph's
        private pool and die with it."""
        if torch.cuda.is_current_stream_capturing():
            raise L.PcdError("CurriculumCenterHeadLoss: run one eager step (or init_state(device)) before graph capture")
        self.hm_loss_func = FocalLossCenterCurriculumState(self.conf_shape, device)
        return self.hm_loss_func

    def forward(self, pred_dicts, target_dicts, epoch=0):
        dev = pred_dicts[0]['hm'].device
        if self.hm_loss_func is None:
            self.init_state(dev)
        st = self.hm_loss_func
        cur = curriculum_struct(self.curriculum, epoch, self.conf_shape)
        tb, loss = {}, 0
        confidence, conf_true, conf_aug = 0, 0, 0
        for idx, pred in enumerate(pred_dicts):
            regs = [pred[name] for name in self.head_order]
            box_mask = target_dicts['masks'][idx].clone()                  # (get_loss passes a clone, :331)
            hmask = target_dicts['heatmap_mask'][idx] if cur.ucl else None
            head_loss, out = _ComHeadLoss.apply(
                pred['hm'], target_dicts['heatmaps'][idx], target_dicts['inds'][idx], box_mask,
                target_dicts['target_boxes'][idx], target_dicts['radius_map'][idx], hmask,
                st.owner(pred['hm']) if cur.ucl else None, cur, st.state,
                st.confidence_all[0] if self.conf_shape is not None else None,
                st.confidence_all[1] if self.conf_shape is not None else None,
                st.epoch_confidence if self.conf_shape is not None else None,
                st.epoch_num if self.conf_shape is not None else None,
                self.code_weights, self.cls_weight, self.loc_weight, *regs)
            loss = loss + head_loss if idx else head_loss
            tb['hm_loss_head_%d' % idx] = out[1]
            tb['loc_loss_head_%d' % idx] = out[2]
            tb['box_mask_head_%d' % idx] = box_mask
            confidence = (confidence + out[3]) / len(pred_dicts)           # (the reference divides inside the loop, :346-352)
            conf_true = (conf_true + 1) / len(pred_dicts)                  # neg_loss returns the constants 1 and 2 (:1200-1201)
            conf_aug = (conf_aug + 2) / len(pred_dicts)
        tb['rpn_loss'] = loss.detach()
        tb['confidence'] = confidence
        tb['confidence_true'] = conf_true
        tb['confidence_aug'] = conf_aug
        return loss, tb
