"""Host-side mirror of the reference's hot-path modules (same names, batch_dict keys and
constructor arguments as pcdet/models/backbones_3d/{vfe,spconv_backbone.py} and
pcdet/models/backbones_2d/map_to_bev), running on the HIP kernels."""
from .backbone3d import SparseBasicBlock, VoxelBackBone8x, VoxelResBackBone8x, post_act_block  # noqa: F401
from .data import VoxelGeneratorWrapper, collate_points, transform_points_to_voxels  # noqa: F401
from .map_to_bev import HeightCompression, PointPillarScatter  # noqa: F401
from .vfe import DynamicMeanVFE, DynamicPillarVFE, MeanVFE, PillarVFE  # noqa: F401
from .dense2d import BaseBEVBackbone, CenterHeadTowers, SeparateHead  # noqa: F401
from .curriculum_head import CurriculumCenterHead, CurriculumCenterHead_x5  # noqa: F401
