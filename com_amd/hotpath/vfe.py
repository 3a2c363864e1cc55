"""VFE modules (pcdet/models/backbones_3d/vfe/{mean_vfe,dynamic_mean_vfe,pillar_vfe}.py)."""
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
    """pillar_vfe.py:8-49"""

    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.last_vfe = last_layer
        self.use_norm = use_norm
        if not self.last_vfe:
            out_channels = out_channels // 2
        if self.use_norm:
            self.linear = nn.Linear(in_channels, out_channels, bias=False)
            self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)
        else:
            self.linear = nn.Linear(in_channels, out_channels, bias=True)
        self.part = 50000

    def forward(self, inputs):
        x = self.linear(inputs)
        x = self.norm(x.permute(0, 2, 1)).permute(0, 2, 1) if self.use_norm else x
        x = F.relu(x)
        x_max = torch.max(x, dim=1, keepdim=True)[0]
        if self.last_vfe:
            return x_max
        x_repeat = x_max.repeat(1, inputs.shape[1], 1)
        return torch.cat([x, x_repeat], dim=2)


class PillarVFE(VFETemplate):
    """pillar_vfe.py:52-123 (decoration + PointNet).  Plain library math (Linear/BN) stays in torch."""

    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.use_norm = _cfg_get(model_cfg, 'USE_NORM')
        self.with_distance = _cfg_get(model_cfg, 'WITH_DISTANCE')
        self.use_absolute_xyz = _cfg_get(model_cfg, 'USE_ABSLOTE_XYZ')
        num_point_features += 6 if self.use_absolute_xyz else 3
        if self.with_distance:
            num_point_features += 1
        self.num_filters = list(_cfg_get(model_cfg, 'NUM_FILTERS'))
        assert len(self.num_filters) > 0
        num_filters = [num_point_features] + self.num_filters
        layers = []
        for i in range(len(num_filters) - 1):
            layers.append(PFNLayer(num_filters[i], num_filters[i + 1], self.use_norm,
                                   last_layer=(i >= len(num_filters) - 2)))
        self.pfn_layers = nn.ModuleList(layers)
        self.voxel_x, self.voxel_y, self.voxel_z = voxel_size[0], voxel_size[1], voxel_size[2]
        self.x_offset = self.voxel_x / 2 + point_cloud_range[0]
        self.y_offset = self.voxel_y / 2 + point_cloud_range[1]
        self.z_offset = self.voxel_z / 2 + point_cloud_range[2]

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    @staticmethod
    def get_paddings_indicator(actual_num, max_num, axis=0):
        actual_num = torch.unsqueeze(actual_num, axis + 1)
        shape = [1] * len(actual_num.shape)
        shape[axis + 1] = -1
        max_num = torch.arange(max_num, dtype=torch.int, device=actual_num.device).view(shape)
        return actual_num.int() > max_num

    def forward(self, batch_dict, **kwargs):
        vf, nump, coords = batch_dict['voxels'], batch_dict['voxel_num_points'], batch_dict['voxel_coords']
        points_mean = vf[:, :, :3].sum(dim=1, keepdim=True) / nump.type_as(vf).view(-1, 1, 1)
        f_cluster = vf[:, :, :3] - points_mean
        f_center = torch.zeros_like(vf[:, :, :3])
        f_center[:, :, 0] = vf[:, :, 0] - (coords[:, 3].to(vf.dtype).unsqueeze(1) * self.voxel_x + self.x_offset)
        f_center[:, :, 1] = vf[:, :, 1] - (coords[:, 2].to(vf.dtype).unsqueeze(1) * self.voxel_y + self.y_offset)
        f_center[:, :, 2] = vf[:, :, 2] - (coords[:, 1].to(vf.dtype).unsqueeze(1) * self.voxel_z + self.z_offset)
        features = [vf, f_cluster, f_center] if self.use_absolute_xyz else [vf[..., 3:], f_cluster, f_center]
        if self.with_distance:
            features.append(torch.norm(vf[:, :, :3], 2, 2, keepdim=True))
        features = torch.cat(features, dim=-1)
        mask = self.get_paddings_indicator(nump, features.shape[1], axis=0)
        features = features * torch.unsqueeze(mask, -1).type_as(vf)
        for pfn in self.pfn_layers:
            features = pfn(features)
        batch_dict['pillar_features'] = features.squeeze(1) if features.dim() == 3 else features
        return batch_dict
