"""VFE modules (pcdet/models/backbones_3d/vfe/{mean_vfe,dynamic_mean_vfe,pillar_vfe,dynamic_pillar_vfe}.py)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def _cfg_get(cfg, name):
    return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)


class VFETemplate(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg

    def get_output_feature_dim(self):
        raise NotImplementedError


class MeanVFE(VFETemplate):
    """mean_vfe.py:14-31.  If the voxeliser already produced the fused mean ('voxel_features' present
    and 'voxels' absent) this is a no-op."""

    def __init__(self, model_cfg, num_point_features, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        if batch_dict.get('voxels', None) is None and 'voxel_features' in batch_dict:
            return batch_dict
        batch_dict['voxel_features'] = ops.mean_vfe(batch_dict['voxels'], batch_dict['voxel_num_points'])
        return batch_dict


class DynamicMeanVFE(VFETemplate):
    """dynamic_mean_vfe.py:13-76: on-GPU point -> voxel scatter (sorted by key, no caps)."""

    def __init__(self, model_cfg, num_point_features, voxel_size, grid_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.num_point_features = num_point_features
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.grid_size = [int(v) for v in grid_size]

    def get_output_feature_dim(self):
        return self.num_point_features

    @torch.no_grad()
    def forward(self, batch_dict, **kwargs):
        feats, coords, _ = ops.voxelize_dynamic_mean(batch_dict['points'], batch_dict['batch_size'],
                                                     self.point_cloud_range, self.voxel_size)
        batch_dict['voxel_features'] = feats.contiguous()
        batch_dict['voxel_coords'] = coords.contiguous()
        return batch_dict


class PFNLayer(nn.Module):
    """One PointNet stage of the pillar encoder (pillar_vfe.py:8-49): per-point Linear (+BatchNorm1d eps 1e-3,
    momentum 0.01) + ReLU, max over the points of a pillar; non-final stages append the pillar maximum to every
    point.  Sub-module names (`linear`, `norm`) match the reference state dict."""

    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.last_vfe, self.use_norm = last_layer, use_norm
        width = out_channels if last_layer else out_channels // 2
        self.linear = nn.Linear(in_channels, width, bias=not use_norm)
        if use_norm:
            self.norm = nn.BatchNorm1d(width, eps=1e-3, momentum=0.01)

    def forward(self, inputs):
        m, t, _ = inputs.shape
        h = self.linear(inputs)
        if self.use_norm:                       # BatchNorm over all (pillar, point) rows, channel last
            h = self.norm(h.reshape(m * t, -1)).reshape(m, t, -1)
        # ReLU + max over the points (+ [h, max] concatenation) in one HIP pass (pcd_pfn_relu_pool)
        out = _PfnReluPool.apply(h.contiguous().float(), self.last_vfe)
        return out.unsqueeze(1) if self.last_vfe else out


class _PfnReluPool(torch.autograd.Function):
    """pillar_vfe.py:44-49: relu -> max over the pillar's points -> (repeat + cat for non-final stages)."""

    @staticmethod
    def forward(ctx, h, last):
        out, arg = ops.pfn_relu_pool(h, last)
        ctx.save_for_backward(h, arg)
        ctx.last = last
        return out

    @staticmethod
    def backward(ctx, g):
        h, arg = ctx.saved_tensors
        return ops.pfn_relu_pool_backward(g, h, arg, ctx.last), None


class PillarVFE(VFETemplate):
    """Pillar feature encoder (pillar_vfe.py:52-123): decorate the <= T points of each pillar with their offset
    to the pillar mean (`f_cluster`) and to the pillar centre (`f_center`), zero the padding slots, run the PFN
    stack.  Linear / BatchNorm are plain library math and stay in torch."""

    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.use_norm = _cfg_get(model_cfg, 'USE_NORM')
        self.with_distance = _cfg_get(model_cfg, 'WITH_DISTANCE')
        self.use_absolute_xyz = _cfg_get(model_cfg, 'USE_ABSLOTE_XYZ')
        self.num_filters = list(_cfg_get(model_cfg, 'NUM_FILTERS'))
        assert self.num_filters
        width_in = num_point_features + (6 if self.use_absolute_xyz else 3) + (1 if self.with_distance else 0)
        widths = [width_in] + self.num_filters
        self.pfn_layers = nn.ModuleList(
            PFNLayer(widths[i], widths[i + 1], self.use_norm, last_layer=(i == len(widths) - 2))
            for i in range(len(widths) - 1))
        # pillar centre of voxel index c along an axis = c * size + (size / 2 + range_min)
        self.voxel_x, self.voxel_y, self.voxel_z = (float(v) for v in voxel_size[:3])
        self.x_offset = self.voxel_x / 2 + float(point_cloud_range[0])
        self.y_offset = self.voxel_y / 2 + float(point_cloud_range[1])
        self.z_offset = self.voxel_z / 2 + float(point_cloud_range[2])

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    def forward(self, batch_dict, **kwargs):
        # decoration (offset to the pillar mean and to the pillar centre, optional range) + padding mask: one HIP
        # pass over the padded pillars (pcd_pillar_decorate) instead of the reference's mean / sub / cat / mask chain
        feats = ops.pillar_decorate(batch_dict['voxels'], batch_dict['voxel_num_points'], batch_dict['voxel_coords'],
                                    (self.voxel_x, self.voxel_y, self.voxel_z),
                                    (self.x_offset, self.y_offset, self.z_offset), self.use_absolute_xyz,
                                    self.with_distance)
        for layer in self.pfn_layers:
            feats = layer(feats)
        batch_dict['pillar_features'] = feats.squeeze(1)
        return batch_dict


class _SegmentMax(torch.autograd.Function):
    """torch_scatter.scatter_max(x, seg, dim=0)[0] with its gradient (dynamic_pillar_vfe.py:40): HIP kernels, one
    64-bit atomicMax per element, deterministic argmax."""

    @staticmethod
    def forward(ctx, x, seg, m):
        out, arg = ops.segment_max(x.detach().float().contiguous(), seg, m)
        ctx.save_for_backward(arg)
        ctx.n, ctx.dtype = x.shape[0], x.dtype
        return out.to(x.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        (arg,) = ctx.saved_tensors
        return ops.segment_max_backward(grad_out, arg, ctx.n).to(ctx.dtype), None, None


class PFNLayerV2(nn.Module):
    """PointNet stage over DYNAMIC pillars (dynamic_pillar_vfe.py:14-47): per-point Linear (+BatchNorm1d) + ReLU,
    maximum over the points of each pillar through `unq_inv`; non-final stages append the pillar maximum to every
    point.  Sub-module names (`linear`, `norm`) match the reference state dict."""

    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.last_vfe, self.use_norm = last_layer, use_norm
        width = out_channels if last_layer else out_channels // 2
        self.linear = nn.Linear(in_channels, width, bias=not use_norm)
        if use_norm:
            self.norm = nn.BatchNorm1d(width, eps=1e-3, momentum=0.01)

    def forward(self, inputs, seg32, seg64, num_pillars):
        h = self.linear(inputs)
        if self.use_norm:
            h = self.norm(h)
        h = F.relu(h)
        pooled = _SegmentMax.apply(h, seg32, num_pillars)
        return pooled if self.last_vfe else torch.cat((h, pooled[seg64]), dim=1)


class DynamicPillarVFE(VFETemplate):
    """dynamic_pillar_vfe.py:49-142: every in-range point (x, y only -- z is not tested) joins the pillar of its
    (b, x, y) cell, pillars sorted by b * X * Y + cx * Y + cy, no caps; points are decorated with their offset to
    the pillar mean and to the pillar centre and run through the PFNLayerV2 stack.  Pillar ids, means and
    coordinates come from the bitmap-rank voxeliser (`pcd_voxelize_dynamic_mean` with a single z cell), the segment
    maxima from `pcd_segment_max`; Linear / BatchNorm stay in torch."""

    def __init__(self, model_cfg, num_point_features, voxel_size, grid_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.use_norm = _cfg_get(model_cfg, 'USE_NORM')
        self.with_distance = _cfg_get(model_cfg, 'WITH_DISTANCE')
        self.use_absolute_xyz = _cfg_get(model_cfg, 'USE_ABSLOTE_XYZ')
        self.num_filters = list(_cfg_get(model_cfg, 'NUM_FILTERS'))
        assert self.num_filters
        width_in = num_point_features + (6 if self.use_absolute_xyz else 3) + (1 if self.with_distance else 0)
        widths = [width_in] + self.num_filters
        self.pfn_layers = nn.ModuleList(
            PFNLayerV2(widths[i], widths[i + 1], self.use_norm, last_layer=(i >= len(widths) - 2))
            for i in range(len(widths) - 1))
        self.voxel_x, self.voxel_y, self.voxel_z = (float(v) for v in voxel_size[:3])
        self.x_offset = self.voxel_x / 2 + float(point_cloud_range[0])
        self.y_offset = self.voxel_y / 2 + float(point_cloud_range[1])
        self.z_offset = self.voxel_z / 2 + float(point_cloud_range[2])
        r = [float(v) for v in point_cloud_range]
        # one z cell that takes every finite z: the reference only range-checks x and y (dynamic_pillar_vfe.py:93-94)
        self._range = [r[0], r[1], -1.0e9, r[3], r[4], 1.0e9]
        self._vsize = [self.voxel_x, self.voxel_y, 2.0e9]
        self.grid_size = [int(v) for v in grid_size]

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    def forward(self, batch_dict, **kwargs):
        points = batch_dict['points'].float().contiguous()          # (batch_idx, x, y, z, i, e)
        mean, coords, _, inv = ops.voxelize_dynamic_mean(points, batch_dict['batch_size'], self._range, self._vsize,
                                                         return_inverse=True)
        keep = inv >= 0
        points, seg32 = points[keep], inv[keep].contiguous()
        seg64 = seg32.long()
        xyz = points[:, 1:4]
        f_cluster = xyz - mean[seg64, :3]
        cell = coords[seg64]                                         # (b, 0, cy, cx) of every point's pillar
        f_center = torch.stack((xyz[:, 0] - (cell[:, 3].to(xyz.dtype) * self.voxel_x + self.x_offset),
                                xyz[:, 1] - (cell[:, 2].to(xyz.dtype) * self.voxel_y + self.y_offset),
                                xyz[:, 2] - self.z_offset), dim=1)
        parts = [points[:, 1:] if self.use_absolute_xyz else points[:, 4:], f_cluster, f_center]
        if self.with_distance:
            parts.append(xyz.norm(dim=1, keepdim=True))
        feats = torch.cat(parts, dim=1)
        m = coords.shape[0]
        for layer in self.pfn_layers:
            feats = layer(feats, seg32, seg64, m)
        batch_dict['pillar_features'] = feats
        batch_dict['voxel_coords'] = coords.contiguous()
        return batch_dict
