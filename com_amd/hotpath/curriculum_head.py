"""`CurriculumCenterHead` / `CurriculumCenterHead_x5` -- the COM dense head as a registry drop-in
(pcdet/models/dense_heads/__init__.py:15-32; reference classes: curriculum_center_head.py:48-511, head_zoo.py:145-149).

Same constructor arguments, `forward(data_dict)` contract (`spatial_features_2d`, `gt_boxes`, `num_points_in_gt`,
`true_object`, `occupancy_ratio`, `facade_type` in; `rois` / `roi_scores` / `roi_labels` or `final_box_dicts` out),
`get_loss()`, `generate_predicted_boxes()` and module names (`shared_conv`, `heads_list`: state-dict compatible) -- with
the towers on the hand-written 3x3 conv kernels (bf16 channels-last), `cluster` / `assign_targets` / the curriculum
loss on the device (com_head.py) and NMS through com_amd.iou3d_nms.  `self.epoch` is set by the training loop as in the
reference (train_utils.py pushes it every epoch)."""
import torch
import torch.nn as nn

from . import com_head
from .dense2d import CenterHeadTowers, _get
from .. import iou3d_nms


def _topk(scores, K):
    """centernet_utils.py:199-214"""
    batch, num_class, height, width = scores.size()
    topk_scores, topk_inds = torch.topk(scores.flatten(2, 3), K)
    topk_inds = topk_inds % (height * width)
    topk_ys = (topk_inds // width).float()
    topk_xs = (topk_inds % width).int().float()
    topk_score, topk_ind = torch.topk(topk_scores.view(batch, -1), K)
    topk_classes = (topk_ind // K).int()
    pick = lambda t: t.view(batch, -1).gather(1, topk_ind)
    return topk_score, pick(topk_inds), topk_classes, pick(topk_ys), pick(topk_xs)


def _gather_map(feat, inds):
    """[B, C, H, W] at flat pixel indices [B, K] -> [B, K, C]  (centernet_utils.py:181-196)"""
    B, C = feat.shape[0], feat.shape[1]
    flat = feat.permute(0, 2, 3, 1).reshape(B, -1, C)
    return flat.gather(1, inds.unsqueeze(2).expand(B, inds.shape[1], C))


def decode_bbox_from_heatmap(heatmap, rot_cos, rot_sin, center, center_z, dim, point_cloud_range, voxel_size,
                             feature_map_stride, K=100, score_thresh=None, post_center_limit_range=None, vel=None):
    """centernet_utils.py:217-279 (circle_nms is `assert False` there: not offered).  Returns the per-frame list of
    dicts pred_boxes [n, 7(+2)], pred_scores [n], pred_labels [n] (int32, 0-based class inside the head)."""
    B = heatmap.shape[0]
    scores, inds, class_ids, ys, xs = _topk(heatmap, K)
    center = _gather_map(center, inds)
    rot_sin, rot_cos = _gather_map(rot_sin, inds), _gather_map(rot_cos, inds)
    center_z, dim = _gather_map(center_z, inds), _gather_map(dim, inds)
    angle = torch.atan2(rot_sin, rot_cos)
    xs = xs.view(B, K, 1) + center[:, :, 0:1]
    ys = ys.view(B, K, 1) + center[:, :, 1:2]
    xs = xs * feature_map_stride * voxel_size[0] + point_cloud_range[0]
    ys = ys * feature_map_stride * voxel_size[1] + point_cloud_range[1]
    parts = [xs, ys, center_z, dim, angle]
    if vel is not None:
        parts.append(_gather_map(vel, inds))
    boxes = torch.cat(parts, dim=-1)
    assert post_center_limit_range is not None
    mask = (boxes[..., :3] >= post_center_limit_range[:3]).all(2) & (boxes[..., :3] <= post_center_limit_range[3:]).all(2)
    if score_thresh is not None:
        mask &= scores > score_thresh
    return [{'pred_boxes': boxes[k, mask[k]], 'pred_scores': scores[k, mask[k]], 'pred_labels': class_ids[k, mask[k]]}
            for k in range(B)]


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    """model_nms_utils.py:5-27 over com_amd.iou3d_nms (nms_gpu / nms_normal_gpu)."""
    src = box_scores
    if score_thresh is not None:
        keep_mask = box_scores >= score_thresh
        box_scores, box_preds = box_scores[keep_mask], box_preds[keep_mask]
    selected = box_scores.new_zeros((0,), dtype=torch.int64)
    if box_scores.shape[0] > 0:
        top, indices = torch.topk(box_scores, k=min(int(_get(nms_config, 'NMS_PRE_MAXSIZE')), box_scores.shape[0]))
        fn = getattr(iou3d_nms, _get(nms_config, 'NMS_TYPE'))
        keep, _ = fn(box_preds[indices][:, 0:7].contiguous(), top, float(_get(nms_config, 'NMS_THRESH')))
        selected = indices[keep[:int(_get(nms_config, 'NMS_POST_MAXSIZE'))]]
    if score_thresh is not None:
        selected = keep_mask.nonzero().view(-1)[selected]
    return selected, src[selected]


class CurriculumCenterHead(nn.Module):
    conf_shape = None            # base class: FocalLossCenterCurriculum(conf_shape=None), curriculum_center_head.py:103-105

    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range, voxel_size,
                 predict_boxes_when_training=True):
        super().__init__()
        self.model_cfg, self.num_class, self.class_names = model_cfg, num_class, list(class_names)
        self.grid_size, self.point_cloud_range, self.voxel_size = grid_size, list(point_cloud_range), list(voxel_size)
        ta = _get(model_cfg, 'TARGET_ASSIGNER_CONFIG')
        self.feature_map_stride = _get(ta, 'FEATURE_MAP_STRIDE', None)
        self.epoch, self.cur_iter = 0, 0
        self.epoch_thredhold = _get(ta, 'EPOCH_THRED', 100)
        self.min_points = _get(ta, 'MIN_POINTS', 1)
        self.class_names_each_head = [[x for x in names if x in self.class_names]
                                      for names in _get(model_cfg, 'CLASS_NAMES_EACH_HEAD')]
        assert sum(len(x) for x in self.class_names_each_head) == len(self.class_names)
        self.class_id_mapping_each_head = [[self.class_names.index(x) for x in names] for names in self.class_names_each_head]
        towers = CenterHeadTowers(model_cfg, input_channels, self.class_names_each_head)
        self.shared_conv, self.heads_list = towers.shared_conv, towers.heads_list      # (reference module names)
        self._towers = [towers]                                                        # not a submodule: no duplicate keys
        self.separate_head_cfg = _get(model_cfg, 'SEPARATE_HEAD_CFG')
        self.predict_boxes_when_training = predict_boxes_when_training
        self.forward_ret_dict = {}
        lw = _get(_get(model_cfg, 'LOSS_CONFIG'), 'LOSS_WEIGHTS')
        self.loss = com_head.CurriculumCenterHeadLoss(
            _get(self.separate_head_cfg, 'HEAD_ORDER'), _get(model_cfg, 'LOSS_CURRICULUM', None), conf_shape=self.conf_shape,
            cls_weight=lw['cls_weight'], loc_weight=lw['loc_weight'], code_weights=lw['code_weights'])

    @property
    def hm_loss_func(self):
        """What train_utils.py:111-112 reads (`hm_loss_func.confidence_all`) -- the device-side state."""
        return self.loss.hm_loss_func

    def assign_targets(self, gt_boxes, feature_map_size=None, npgt=None, true_object=None, **kwargs):
        ta = _get(self.model_cfg, 'TARGET_ASSIGNER_CONFIG')
        return com_head.assign_targets(
            gt_boxes, feature_map_size, self.class_names, self.class_names_each_head, self.point_cloud_range,
            self.voxel_size, _get(ta, 'FEATURE_MAP_STRIDE'), npgt, true_object=true_object,
            num_max_objs=_get(ta, 'NUM_MAX_OBJS'), gaussian_overlap=_get(ta, 'GAUSSIAN_OVERLAP'),
            min_radius=_get(ta, 'MIN_RADIUS'), epoch=self.epoch, epoch_threshold=self.epoch_thredhold,
            min_points=self.min_points)

    def cluster(self, gt_boxes, true_object, occupancy_ratio, facade_type):
        return com_head.cluster(gt_boxes, true_object, occupancy_ratio, facade_type)

    def get_loss(self):
        return self.loss(self.forward_ret_dict['pred_dicts'], self.forward_ret_dict['target_dicts'], epoch=self.epoch)

    def generate_predicted_boxes(self, batch_size, pred_dicts):
        """curriculum_center_head.py:360-412"""
        pp = _get(self.model_cfg, 'POST_PROCESSING')
        nms_cfg = _get(pp, 'NMS_CONFIG')
        dev = pred_dicts[0]['hm'].device
        limit = torch.tensor(_get(pp, 'POST_CENTER_LIMIT_RANGE'), device=dev).float()
        ret = [{'pred_boxes': [], 'pred_scores': [], 'pred_labels': []} for _ in range(batch_size)]
        order = _get(self.separate_head_cfg, 'HEAD_ORDER')
        for idx, pd in enumerate(pred_dicts):
            f = lambda t: t.float()
            finals = decode_bbox_from_heatmap(
                heatmap=f(pd['hm']).sigmoid(), rot_cos=f(pd['rot'])[:, 0:1], rot_sin=f(pd['rot'])[:, 1:2],
                center=f(pd['center']), center_z=f(pd['center_z']), dim=f(pd['dim']).exp(),
                vel=f(pd['vel']) if 'vel' in order else None, point_cloud_range=self.point_cloud_range,
                voxel_size=self.voxel_size, feature_map_stride=self.feature_map_stride,
                K=_get(pp, 'MAX_OBJ_PER_SAMPLE'), score_thresh=_get(pp, 'SCORE_THRESH'), post_center_limit_range=limit)
            mapping = torch.tensor(self.class_id_mapping_each_head[idx], device=dev)
            for k, fd in enumerate(finals):
                fd['pred_labels'] = mapping[fd['pred_labels'].long()]
                if _get(nms_cfg, 'NMS_TYPE') != 'circle_nms':
                    sel, sel_scores = class_agnostic_nms(fd['pred_scores'], fd['pred_boxes'], nms_cfg, score_thresh=None)
                    fd['pred_boxes'], fd['pred_scores'], fd['pred_labels'] = fd['pred_boxes'][sel], sel_scores, fd['pred_labels'][sel]
                for key in ret[k]:
                    ret[k][key].append(fd[key])
        for k in range(batch_size):
            ret[k]['pred_boxes'] = torch.cat(ret[k]['pred_boxes'], dim=0)
            ret[k]['pred_scores'] = torch.cat(ret[k]['pred_scores'], dim=0)
            ret[k]['pred_labels'] = torch.cat(ret[k]['pred_labels'], dim=0) + 1
        return ret

    @staticmethod
    def reorder_rois_for_refining(batch_size, pred_dicts):
        """curriculum_center_head.py:394-412"""
        num_max = max(1, max(len(d['pred_boxes']) for d in pred_dicts))
        b0 = pred_dicts[0]['pred_boxes']
        rois = b0.new_zeros((batch_size, num_max, b0.shape[-1]))
        roi_scores = b0.new_zeros((batch_size, num_max))
        roi_labels = b0.new_zeros((batch_size, num_max)).long()
        for b in range(batch_size):
            n = len(pred_dicts[b]['pred_boxes'])
            rois[b, :n], roi_scores[b, :n], roi_labels[b, :n] = pred_dicts[b]['pred_boxes'], pred_dicts[b]['pred_scores'], pred_dicts[b]['pred_labels']
        return rois, roi_scores, roi_labels

    def forward(self, data_dict):
        """curriculum_center_head.py:461-487"""
        sf = data_dict['spatial_features_2d']
        pred_dicts = self._towers[0]({'spatial_features_2d': sf})['pred_dicts']
        if self.training:
            group = self.cluster(data_dict['gt_boxes'], data_dict.get('true_object', None), data_dict['occupancy_ratio'],
                                 data_dict['facade_type'])
            self.forward_ret_dict['target_dicts'] = self.assign_targets(
                data_dict['gt_boxes'], feature_map_size=sf.size()[2:], npgt=data_dict['num_points_in_gt'], true_object=group)
        self.forward_ret_dict['pred_dicts'] = pred_dicts
        if not self.training or self.predict_boxes_when_training:
            boxes = self.generate_predicted_boxes(data_dict['batch_size'], pred_dicts)
            if self.predict_boxes_when_training:
                rois, roi_scores, roi_labels = self.reorder_rois_for_refining(data_dict['batch_size'], boxes)
                data_dict.update(rois=rois, roi_scores=roi_scores, roi_labels=roi_labels, has_class_labels=True)
            else:
                data_dict['final_box_dicts'] = boxes
        return data_dict


class CurriculumCenterHead_x5(CurriculumCenterHead):
    """head_zoo.py:145-149: the (3, 96) group-confidence pass that feeds COMAug."""
    conf_shape = (3, 96)
