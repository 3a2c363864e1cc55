"""CenterHead.assign_targets (pcdet/models/dense_heads/center_head.py:163-225) without the host loop: one call of
pcd_centerhead_assign_targets per head builds heat maps, regression targets, indices and masks for the whole batch on
the device.  Returns the reference's `ret_dict` (lists over heads of [B, ...] tensors; `inds` / `masks` int64)."""
import torch

from .. import _lib as L


def assign_targets(gt_boxes, feature_map_size, class_names, class_names_each_head, point_cloud_range, voxel_size,
                   feature_map_stride, num_max_objs=500, gaussian_overlap=0.1, min_radius=2):
    """gt_boxes [B, M, 8+] (device, f32; last column = 1-based class id, 0 = padding); feature_map_size [H, W]."""
    if not gt_boxes.is_cuda:
        raise L.PcdError("assign_targets needs a HIP device tensor (there is no CPU fallback)")
    gt = gt_boxes.contiguous().float()
    B, M, code = gt.shape
    H, W = int(feature_map_size[0]), int(feature_map_size[1])
    lib = L.lib()
    ws = torch.empty((max(int(lib.pcd_centerhead_assign_workspace_bytes(B, num_max_objs)), 256),), dtype=torch.uint8,
                     device=gt.device)
    ret = {'heatmaps': [], 'target_boxes': [], 'inds': [], 'masks': [], 'heatmap_masks': []}
    for head_names in class_names_each_head:
        cmap = [0] * (len(class_names) + 1)
        for i, name in enumerate(class_names):
            if name in head_names:
                cmap[i + 1] = list(head_names).index(name) + 1
        nc = len(head_names)
        heatmap = torch.empty((B, nc, H, W), dtype=torch.float32, device=gt.device)
        boxes = torch.empty((B, num_max_objs, code), dtype=torch.float32, device=gt.device)
        inds = torch.empty((B, num_max_objs), dtype=torch.int64, device=gt.device)
        mask = torch.empty((B, num_max_objs), dtype=torch.int64, device=gt.device)
        L.check(lib.pcd_centerhead_assign_targets(
            L.ptr(gt), B, M, code, L.host_i32(cmap), len(cmap), nc, W, H, int(feature_map_stride),
            L.host_f32([voxel_size[0], voxel_size[1]]), L.host_f32([point_cloud_range[0], point_cloud_range[1]]),
            int(num_max_objs), float(gaussian_overlap), int(min_radius), L.ptr(heatmap), L.ptr(boxes), L.ptr(inds),
            L.ptr(mask), L.ptr(ws), ws.numel(), L.stream_ptr()), "pcd_centerhead_assign_targets")
        ret['heatmaps'].append(heatmap)
        ret['target_boxes'].append(boxes)
        ret['inds'].append(inds)
        ret['masks'].append(mask)
    return ret
