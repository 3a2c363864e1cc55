"""Data-side boundary: the voxel generator wrapper and batch collation of
pcdet/datasets/processor/data_processor.py:15-60,125-153 and pcdet/datasets/dataset.py:252-259,
moved onto the GPU (one batched launch sequence per step instead of one CPU call per frame in a
DataLoader worker)."""
import numpy as np
import torch

from .. import ops
from ..spconv.utils import VoxelGeneratorV2


class VoxelGeneratorWrapper:
    """data_processor.py:15-60: same constructor and `generate(points) -> (voxels, coordinates, num_points)`."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels):
        self.spconv_ver = 1
        self._voxel_generator = VoxelGeneratorV2(voxel_size=vsize_xyz, point_cloud_range=coors_range_xyz,
                                                 max_num_points=max_num_points_per_voxel,
                                                 max_voxels=max_num_voxels)
        self.num_point_features = num_point_features

    def generate(self, points):
        out = self._voxel_generator.generate(points)
        return out['voxels'], out['coordinates'], out['num_points_per_voxel']


def collate_points(frames, device="cuda"):
    """dataset.py:252-259 for `points`: concat + left-pad the batch index -> ([sum N, 1+C] f32, offsets)."""
    offs = [0]
    rows = []
    for b, p in enumerate(frames):
        p = torch.as_tensor(p, dtype=torch.float32)
        rows.append(torch.nn.functional.pad(p, (1, 0), value=float(b)))
        offs.append(offs[-1] + p.shape[0])
    return torch.cat(rows, 0).to(device).contiguous(), offs


def transform_points_to_voxels(batch_dict, point_cloud_range, voxel_size, max_points_per_voxel,
                               max_voxels, fuse_mean=True, keep_voxels=False, bf16_features=False, out=None,
                               row_order="first", key_depth_extra=1, bf16_feature_stride=None):
    """Batched GPU form of data_processor.py:125-153 + collate: consumes batch_dict['points']
    ([sum N, 1+C] with batch index) and batch_dict['frame_offsets'], produces 'voxel_coords' [M,4],
    'voxel_num_points', and either 'voxels' (reference layout) or the fused MeanVFE 'voxel_features'."""
    pts = batch_dict['points']
    res = ops.voxelize_hard(pts, batch_dict['frame_offsets'], point_cloud_range, voxel_size,
                            max_points_per_voxel, max_voxels, feat_offset=1, num_features=pts.shape[1] - 1,
                            want_voxels=keep_voxels or not fuse_mean, want_mean=fuse_mean and not bf16_features,
                            # (bf16_feature_stride: channels of the bf16 MeanVFE rows, zero-padded -- 16 when the first conv runs
                            #  on the window tiles, which want rows of its OUTPUT width; default: the next power of two >= 8)
                            mean_bf16_stride=(int(bf16_feature_stride) if bf16_feature_stride else ops.pow2_ge8(pts.shape[1] - 1))
                            if (fuse_mean and bf16_features) else 0,
                            out=out, row_order=row_order,
                            # (the 3D backbones' sparse_shape = grid_size[::-1] + [1, 0, 0], spconv_backbone.py:87,187)
                            key_depth=(ops.grid_size(point_cloud_range, voxel_size)[2] + key_depth_extra)
                            if row_order in ("key", "yxz") else 0)
    batch_dict['voxelize_result'] = res          # pass as `out=` to write the same buffers again (static shapes)
    batch_dict['voxel_coords'] = res['coords']
    if res.get('rank', None) is not None:
        batch_dict['voxel_rank'] = res['rank']        # coordinate -> row map (key-ordered rows): level-1 SubM rulebook
    batch_dict['voxel_num_points'] = res['num_points']
    if res['voxels'] is not None:
        batch_dict['voxels'] = res['voxels']
    if fuse_mean:
        # bf16_features: the MeanVFE row is emitted directly as the zero-padded bf16 [M, 8] operand of the
        # first sparse conv (no cast / pad kernels in between)
        batch_dict['voxel_features'] = res['voxel_features_bf16'] if bf16_features else res['voxel_features']
    batch_dict['voxel_counts'] = res['counts']
    if res.get('num_rows', None) is not None:
        batch_dict['voxel_num_rows'] = res['num_rows']
    return batch_dict
