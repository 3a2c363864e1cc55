"""CPU restatement (numpy) of the COM curriculum head -- TEST INFRASTRUCTURE ONLY (see oracle/pcd_oracle.c header):
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product path (com_amd/) never
does.

What it restates (file:line relative to /root/reference):
  cluster_groups        pcdet/models/dense_heads/curriculum_center_head.py:414-459   (CurriculumCenterHead.cluster)
  assign_targets        same file :108-204 (assign_target_of_single_head) and :206-307 (the per-head / per-frame caller)
  gaussian_radius       pcdet/models/model_utils/centernet_utils.py:46-72
  draw_gaussian         pcdet/models/model_utils/centernet_utils.py:75-106
  group_confidence      pcdet/utils/loss_utils.py:1134-1176 (group_confifence / confidence_of_all_groups)
  com_loss              pcdet/utils/loss_utils.py:1178-1310 (FocalLossCenterCurriculum.neg_loss), :1317-1390 (_reg_loss,
                        RegLossCenterNet), curriculum_center_head.py:309-358 (sigmoid clamp + get_loss)
  epoch_gather          tools/train_utils/train_utils.py:57,111-112,208,269-287,325

PARITY STATUS: PINNED by fixtures G11 / G12 / G14 (tests/golden/make_golden.py runs the reference's own classes, extracted
from the files above at generation time): tests/test_com_oracle.py checks every function here against them.

Sums are accumulated in float64 (the reference sums float32 tensors in torch's internal order; the agreed tolerance on
sums is 1e-6 relative); everything that feeds an INDEX or a COUNT is computed in float32 exactly like the reference."""
import numpy as np

f32 = np.float32


def cluster_groups(gt_boxes, true_object, occupancy_ratio, facade_type):
    """[B, M] int64 difficulty group of every ground-truth box: cars 3 distance x 2 length x 4 facade x 4 occupancy bins
    -> 1..96, pedestrians / cyclists 3 distance x 5 occupancy bins -> 1..15; 0 for padding and for objects that are not
    `true_object == 1` (pasted by the augmentor).  Scalars are compared in float32, as torch does for a float32 tensor
    against a Python number (0.41 * 5 / 12 lands on the other side of its float32 neighbour in float64)."""
    g = np.asarray(gt_boxes, f32)
    to = np.asarray(true_object, f32)
    occ = np.asarray(occupancy_ratio, f32)
    fac = np.asarray(facade_type, f32)
    dist = np.sqrt(g[..., 0] * g[..., 0] + g[..., 1] * g[..., 1]).astype(f32)
    length = g[..., 3]
    cls = g[..., -1]
    dbin = np.where(dist <= f32(30), 0, np.where(dist <= f32(50), 1, 2))
    lbin = np.where(length <= f32(6), 0, 1)
    fbin = np.full(cls.shape, -1)
    for i, v in enumerate((3, 2, 1, 0)):
        fbin[fac == f32(v)] = i
    # occupancy lists are REVERSED in the reference ([::-1]): bin 0 is the highest occupancy
    t = [f32(c * 5 / 12) for c in (0.21, 0.41, 0.61, 0.81)]
    obin = np.where(occ > t[3], 0, np.where(occ > t[2], 1, np.where(occ > t[1], 2, np.where(occ > t[0], 3, 4))))
    tc = [f32(0.25), f32(0.5), f32(0.7)]
    obin_car = np.where(occ > tc[2], 0, np.where(occ > tc[1], 1, np.where(occ > tc[0], 2, 3)))
    group = np.zeros(cls.shape, np.int64)
    real = to == f32(1)
    car = real & (cls == f32(1)) & (fbin >= 0)
    group[car] = (1 + ((dbin * 2 + lbin) * 4 + fbin) * 4 + obin_car)[car]
    for c in (2, 3):
        sel = real & (cls == f32(c))
        group[sel] = (1 + dbin * 5 + obin)[sel]
    return group


def gaussian_radius(height, width, min_overlap):
    """float32, operation by operation (min_overlap and the integer literals enter as float32 scalars)."""
    h, w, mo, one = f32(height), f32(width), f32(min_overlap), f32(1)
    b1 = h + w
    c1 = w * h * (one - mo) / (one + mo)
    r1 = (b1 + np.sqrt(b1 * b1 - f32(4) * f32(1) * c1)) / f32(2)
    b2 = f32(2) * (h + w)
    c2 = (one - mo) * w * h
    r2 = (b2 + np.sqrt(b2 * b2 - f32(4) * f32(4) * c2)) / f32(2)
    a3 = f32(4) * mo
    b3 = f32(-2) * mo * (h + w)
    c3 = (mo - one) * w * h
    r3 = (b3 + np.sqrt(b3 * b3 - f32(4) * a3 * c3)) / f32(2)
    return min(min(r1, r2), r3)


def draw_gaussian(plane, cx, cy, radius):
    """max a (2r+1)^2 Gaussian, sigma = (2r+1)/6, computed in float64 and cast to float32, into plane [H, W]."""
    H, W = plane.shape
    left, right = min(cx, radius), min(W - cx, radius + 1)
    top, bottom = min(cy, radius), min(H - cy, radius + 1)
    if left + right <= 0 or top + bottom <= 0:
        return
    sigma = (2 * radius + 1) / 6
    ys, xs = np.ogrid[-top:bottom, -left:right]
    g = np.exp(-(xs * xs + ys * ys) / (2 * sigma * sigma))
    g[g < np.finfo(np.float64).eps * 1.0] = 0            # (h.max() of the full kernel is 1)
    view = plane[cy - top:cy + bottom, cx - left:cx + right]
    np.maximum(view, g.astype(f32), out=view)


def assign_targets(gt_boxes, num_points_in_gt, group, class_names, class_names_each_head, point_cloud_range, voxel_size,
                   feature_map_size, feature_map_stride, num_max_objs=500, gaussian_overlap=0.1, min_radius=2,
                   epoch=0, epoch_threshold=100, min_points=1):
    """Lists over heads of heatmaps [B, C, H, W] f32, target_boxes [B, n, code] f32, inds [B, n] i64, masks [B, n] f32,
    radius_map [B, n, 5] i64 (class, centre x, centre y, radius, group), heatmap_mask [B, C, H, W] f32 ones.
    feature_map_size = [H, W].  Every head sees the ORIGINAL class ids (the reference rewrites the class column of
    gt_boxes in place while filtering, :260, which only stays harmless for head layouts whose earlier heads do not
    renumber a class into a later head's range -- e.g. the single-head COM configs and 'Vehicle first')."""
    gt = np.asarray(gt_boxes, f32)
    npgt = np.asarray(num_points_in_gt, f32)
    B, M, code = gt.shape
    H, W = int(feature_map_size[0]), int(feature_map_size[1])
    vx, vy, st = f32(voxel_size[0]), f32(voxel_size[1]), f32(feature_map_stride)
    rx, ry = f32(point_cloud_range[0]), f32(point_cloud_range[1])
    ret = {k: [] for k in ("heatmaps", "target_boxes", "inds", "masks", "radius_map", "heatmap_mask")}
    for head in class_names_each_head:
        C = len(head)
        heat = np.zeros((B, C, H, W), f32)
        boxes = np.zeros((B, num_max_objs, code), f32)
        inds = np.zeros((B, num_max_objs), np.int64)
        mask = np.zeros((B, num_max_objs), f32)
        rmap = np.zeros((B, num_max_objs, 5 if group is not None else 4), np.int64)
        for b in range(B):
            k = -1
            for m in range(M):
                c0 = int(gt[b, m, -1])
                if c0 <= 0 or c0 > len(class_names) or class_names[c0 - 1] not in head:
                    continue
                k += 1
                if k >= num_max_objs:
                    break
                q = gt[b, m]
                cx = min(max((q[0] - rx) / vx / st, f32(0)), f32(W - 0.5))
                cy = min(max((q[1] - ry) / vy / st, f32(0)), f32(H - 0.5))
                ix, iy = int(cx), int(cy)
                dx, dy = q[3] / vx / st, q[4] / vy / st
                if dx <= 0 or dy <= 0:
                    continue
                if not (0 <= ix <= W and 0 <= iy <= H):
                    continue
                if epoch <= epoch_threshold and npgt[b, m] < min_points:
                    continue
                radius = max(int(gaussian_radius(dx, dy, gaussian_overlap)), int(min_radius))
                cls = list(head).index(class_names[c0 - 1])
                draw_gaussian(heat[b, cls], ix, iy, radius)
                inds[b, k] = iy * W + ix
                mask[b, k] = 1
                boxes[b, k, 0], boxes[b, k, 1], boxes[b, k, 2] = cx - f32(ix), cy - f32(iy), q[2]
                boxes[b, k, 3:6] = np.log(q[3:6])
                boxes[b, k, 6], boxes[b, k, 7] = np.cos(q[6]), np.sin(q[6])
                if code > 8:
                    boxes[b, k, 8:] = q[7:-1]
                rmap[b, k, 0:4] = (cls, ix, iy, radius)
                if group is not None:
                    rmap[b, k, 4] = group[b, m]
        ret["heatmaps"].append(heat)
        ret["target_boxes"].append(boxes)
        ret["inds"].append(inds)
        ret["masks"].append(mask)
        ret["radius_map"].append(rmap)
        ret["heatmap_mask"].append(np.ones((B, C, H, W), f32))
    return ret


def group_confidence(pred, radius_map, conf_shape):
    """(sums [C, G] f64, counts [C, G] f64): per (class, group) the sum of pred[b, class, cy, cx] over the objects of
    radius_map with that class and group id (1-based; group 0 = padding / skipped / not a real object)."""
    C, G = conf_shape
    sums = np.zeros((C, G), np.float64)
    nums = np.zeros((C, G), np.float64)
    B, n = radius_map.shape[:2]
    for b in range(B):
        for k in range(n):
            c, x, y, _, g = (int(v) for v in radius_map[b, k, :5])
            if 1 <= g <= G and 0 <= c < C:
                sums[c, g - 1] += float(pred[b, c, y, x])
                nums[c, g - 1] += 1
    return sums, nums


class ComLossState:
    """What FocalLossCenterCurriculum keeps between steps: the EMA of the average confidence (:1022, :1214)."""

    def __init__(self):
        self.avg_confidence = 0.0


def com_loss(hm_logit, regs, targets, cur, epoch, state, conf_shape=(3, 96), cls_weight=1.0, loc_weight=2.0,
             code_weights=None):
    """One head of CurriculumCenterHead.get_loss.  hm_logit [B, C, H, W] f32; regs = list of [B, c_i, H, W] in HEAD_ORDER;
    targets = dict(heatmap, radius_map, masks, inds, target_boxes) of that head; cur = the LOSS_CURRICULUM dict.
    Returns a dict with loss / hm_loss / loc_loss, avg_confidence, the (3, 96) sums / counts, box_mask, heatmap_mask and
    the gradients w.r.t. hm_logit and every reg map (analytic)."""
    x = np.asarray(hm_logit, f32)
    gt = np.asarray(targets["heatmap"], f32)
    rmap = np.asarray(targets["radius_map"])
    B, C, H, W = x.shape
    s = (f32(1) / (f32(1) + np.exp(-x, dtype=f32))).astype(f32)
    pred = np.clip(s, f32(1e-4), f32(1 - 1e-4))
    inside = (s >= f32(1e-4)) & (s <= f32(1 - 1e-4))
    conf_sum, conf_num = group_confidence(pred, rmap, conf_shape) if conf_shape is not None else (None, None)
    pos = gt == 1
    neg = gt < 1
    num_obj = float(pos.sum())
    with np.errstate(invalid="ignore", divide="ignore"):
        avg_conf = float(np.float64(pred[pos].astype(np.float64).sum()) / num_obj) if num_obj else float("nan")
    alpha = cur.get("ALPHA", 0.001)
    state.avg_confidence = alpha * float(f32(avg_conf)) + (1 - alpha) * state.avg_confidence
    box_mask = np.asarray(targets["masks"], f32).copy()
    hmask = np.ones((B, C, H, W), f32)
    if cur.get("UCL", True):
        thr = 0.5 if cur.get("FIX", False) else state.avg_confidence * 0.5          # (self.threshold = 0.5, :1054)
        elong, height = cur.get("ELONGATION", -10), cur.get("HEIGHT", 1)
        for b in range(B):
            for k in np.nonzero(rmap[b, :, 3] > 0)[0]:
                c, cx, cy = int(rmap[b, k, 0]), int(rmap[b, k, 1]), int(rmap[b, k, 2])
                radius = cur.get("RADIUS", 0) if cur.get("RADIUS", 0) != 0 else int(rmap[b, k, 3]) + cur.get("ADD", 0)
                p = float(pred[b, c, cy, cx])
                if cur.get("STRAIGHT", False):
                    w = cur.get("K", 1.0) * (p - thr) + 1
                elif cur.get("TUNING", False):
                    w = 1
                else:
                    w = height / (1 + np.exp(elong * (p - thr))) + 1 - height / 2
                if cur.get("START", 0) <= epoch <= cur.get("END", 30):
                    box_mask[b, k] = w
                    if cur.get("CENTER", False):
                        hmask[b, c, cy, cx] = w
                    else:
                        left, right = min(cx, radius), min(W - cx, radius + 1)
                        top, bottom = min(cy, radius), min(H - cy, radius + 1)
                        if left + right > 0 and top + bottom > 0:
                            hmask[b, c, cy - top:cy + bottom, cx - left:cx + right] = f32(w)
    # focal terms; the reference multiplies [B, C, H, W] terms by mask[:, None] = [B, 1, C, H, W]: a [B, B, C, H, W]
    # product, i.e. every frame's term is weighted by the SUM over frames of the mask at that (c, y, x)
    p64, g64 = pred.astype(np.float64), gt.astype(np.float64)
    pos_t = np.log(pred).astype(np.float64) * ((f32(1) - pred) ** 2).astype(np.float64) * pos
    neg_t = np.log(f32(1) - pred).astype(np.float64) * (pred * pred).astype(np.float64) * ((f32(1) - gt) ** 4).astype(np.float64) * neg
    msum = hmask.astype(np.float64).sum(0)                                            # [C, H, W]
    pos_loss = (pos_t * msum[None]).sum()
    neg_loss = (neg_t * msum[None]).sum()
    num_pos = (pos.astype(np.float64) * msum[None]).sum()
    div = num_pos if num_pos != 0 else 1.0
    hm_loss = -(pos_loss + neg_loss) / div if num_pos != 0 else -neg_loss
    # d hm_loss / d pred, through the clamp (gradient 1 inside, bounds included) and the sigmoid
    dpos = (((1 - p64) ** 2) / p64 - 2 * (1 - p64) * np.log(p64)) * pos
    dneg = (-(p64 ** 2) / (1 - p64) + 2 * p64 * np.log(1 - p64)) * ((1 - g64) ** 4) * neg
    dterm = (dpos + dneg) if num_pos != 0 else dneg
    d_hm = -(dterm * msum[None]) / div * cls_weight * (s.astype(np.float64) * (1 - s.astype(np.float64))) * inside
    # regression
    dims = [r.shape[1] for r in regs]
    tb = np.asarray(targets["target_boxes"], f32)
    inds = np.asarray(targets["inds"])
    cw = np.ones(sum(dims)) if code_weights is None else np.asarray(code_weights, np.float64)[:sum(dims)]
    num = float(box_mask.astype(np.float64).sum())
    denom = max(num, 1.0)
    l1 = np.zeros(sum(dims), np.float64)
    d_regs = [np.zeros(r.shape, np.float64) for r in regs]
    for b in range(B):
        for k in range(inds.shape[1]):
            m = box_mask[b, k]
            y, xx = int(inds[b, k]) // W, int(inds[b, k]) % W
            d0 = 0
            for r, c_r in enumerate(dims):
                for c in range(c_r):
                    diff = f32(regs[r][b, c, y, xx]) * m - tb[b, k, d0 + c] * m
                    l1[d0 + c] += abs(float(diff))
                    d_regs[r][b, c, y, xx] += np.sign(float(diff)) * float(m) * cw[d0 + c] * loc_weight / denom
                d0 += c_r
    reg = l1 / denom
    loc_loss = float((reg * cw).sum()) * loc_weight
    hm_loss = float(hm_loss) * cls_weight
    return dict(loss=hm_loss + loc_loss, hm_loss=hm_loss, loc_loss=loc_loss, avg_confidence=avg_conf, conf_sum=conf_sum,
                conf_num=conf_num, box_mask=box_mask, heatmap_mask=hmask, reg_loss=reg, grad_hm_logit=d_hm,
                grad_regs=d_regs, num_pos=num_pos)


def epoch_gather(per_rank_step_conf, per_rank_step_num):
    """per_rank_step_* [ranks][steps] of (C, G) float32 arrays -> what every rank hands to COMAug after the epoch:
    float32 sequential sums over the steps per rank (python `sum` of a list), over the ranks in rank order, then
    conf / (num + 0.1) in float32 (a Python float is a weak scalar for numpy: the arrays stay float32)."""
    tot_c = tot_n = None
    for conf_steps, num_steps in zip(per_rank_step_conf, per_rank_step_num):
        c = np.zeros_like(np.asarray(conf_steps[0], f32))
        n = np.zeros_like(c)
        for a in conf_steps:
            c = (c + np.asarray(a, f32)).astype(f32)
        for a in num_steps:
            n = (n + np.asarray(a, f32)).astype(f32)
        tot_c = c if tot_c is None else (tot_c + c).astype(f32)
        tot_n = n if tot_n is None else (tot_n + n).astype(f32)
    return (tot_c / (tot_n + f32(0.1))).astype(f32)
