"""ctypes binding of ``libpcdops_hip.so`` (C ABI: ``include/pcd_ops.h``).

There is NO CPU fallback: if the shared library is missing or a call fails, this module raises.
Python passes raw device pointers (``tensor.data_ptr()``), element counts and the current HIP
stream; torch is only the owner of device memory and streams.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpcdops_hip.so")
EXPERIMENTS_LIB_PATH = os.path.join(_HERE, "lib_experiments", "libpcdops_hip.so")     # `make -C com_amd/csrc EXPERIMENTS=1`
CSRC = os.path.join(_HERE, "csrc")

PCD_F32 = 0
PCD_BF16 = 1

_vp = ctypes.c_void_p
_i = ctypes.c_int
_sz = ctypes.c_size_t

# name -> (restype, argtypes); must list EVERY symbol declared in include/pcd_ops.h
PROTOTYPES = {
    "pcd_version": (_i, []),
    "pcd_error_string": (ctypes.c_char_p, [_i]),
    "pcd_build_arch": (ctypes.c_char_p, []),
    "pcd_last_hip_error_string": (ctypes.c_char_p, []),
    "pcd_set_last_hip_error": (None, [_i]),
    "pcd_set_option": (_i, [ctypes.c_char_p, _i]),
    "pcd_get_option": (_i, [ctypes.c_char_p, _vp]),
    "pcd_voxelize_hard_workspace_bytes": (_sz, [_i, _i, _i]),
    "pcd_voxelize_hard": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                               _i, _vp, _vp, _sz, _vp]),
    "pcd_voxelize_hard_sorted_workspace_bytes": (_sz, [_i, _i, _i, _vp, _vp, _i]),
    "pcd_voxelize_hard_sorted_rank_words": (_i, [_i, _vp, _vp, _i, _vp, _vp]),
    "pcd_voxelize_hard_sorted": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                                      _i, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "pcd_voxelize_hard_yxz_workspace_bytes": (_sz, [_i, _i, _i, _vp, _vp, _i, _i]),
    "pcd_voxelize_hard_yxz": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                                   _i, _vp, _i, _vp, _sz, _vp, _sz, _vp]),
    "pcd_voxelize_hard_host": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "pcd_mean_vfe": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "pcd_voxelize_dynamic_workspace_bytes": (_sz, [_i, _i, _i, _vp, _vp]),
    "pcd_voxelize_dynamic_mean": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcd_segment_max_workspace_bytes": (_sz, [_i, _i]),
    "pcd_segment_max": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "pcd_segment_max_backward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "pcd_rulebook_subm_workspace_bytes": (_sz, [_i, _i]),
    "pcd_rulebook_subm": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "pcd_rulebook_conv_workspace_bytes": (_sz, [_i, _i, _vp, _vp, _vp, _vp, _vp]),
    "pcd_rulebook_conv_classes_workspace_bytes": (_sz, [_i]),
    "pcd_rulebook_conv_classes": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "pcd_sparse_conv_dgrad_classes": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp,
                                           _i, _vp, _vp, _vp]),
    "pcd_sparse_conv_dgrad_classes_tiles": (_i, [_i, _i]),
    "pcd_sparse_conv_gather_gemm_tiles": (_i, [_i, _i, _i, _i, _i]),
    "pcd_sparse_conv_gather_gemm_tiles_dir": (_i, [_i, _i, _i, _i, _i, _i]),
    "pcd_sparse_conv_gather_gemm_variant": (_i, [_i, _i, _i, _i, _i, _i]),
    "pcd_subm_window_tile_rows": (_i, [_i, _i]),
    "pcd_subm_window_partial_rows": (_i, [_i, _i]),
    "pcd_subm_window_set_trace": (_i, [_vp]),
    "pcd_subm_window_plan_bytes": (_sz, [_i, _i, _i]),
    "pcd_subm_window_plan": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "pcd_subm_window_plan_cm": (_i, [_vp, _i, _vp, _i, _vp, _vp, _sz, _i, _i, _i, _vp, _i, _vp, _vp]),
    "pcd_subm_window_packed_weight_bytes": (_sz, [_i, _i]),
    "pcd_subm_window_pack_weight": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "pcd_subm_window_pack_weights_batched": (_i, [_vp, _i, _i, _vp]),
    "pcd_subm_window_wgrad_splits": (_i, [_i]),
    "pcd_sparse_conv_subm_window_wgrad": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "pcd_sparse_conv_subm_window": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "pcd_sparse_conv_subm_window_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "pcd_rulebook_conv_rank_layout": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcd_rulebook_subm_ranked_workspace_bytes": (_sz, [_i, _i]),
    "pcd_rulebook_subm_pairs_workspace_bytes": (_sz, [_i, _i]),
    "pcd_rulebook_subm_pairs": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "pcd_rulebook_conv_pairs": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "pcd_rulebook_subm_ranked": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _sz,
                                      _vp, _i]),
    "pcd_rulebook_subm_ranked4": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _sz,
                                       _vp, _i]),
    "pcd_colmap_bytes": (_sz, [_i, _vp, _i]),
    "pcd_colmap_counts_offset": (_sz, [_i, _vp, _i, _vp]),
    "pcd_colmap_from_rows_workspace_bytes": (_sz, [_i, _vp]),
    "pcd_colmap_from_rows": (_i, [_vp, _i, _vp, _i, _vp, _vp, _sz, _vp, _sz, _vp]),
    "pcd_rulebook_subm_cm_workspace_bytes": (_sz, [_i]),
    "pcd_rulebook_subm_cm": (_i, [_vp, _i, _i, _vp, _vp, _sz, _i, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "pcd_rulebook_conv_cm_workspace_bytes": (_sz, [_i, _i, _vp, _vp, _vp, _vp]),
    "pcd_rulebook_conv_cm_count": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp, _vp, _sz, _vp]),
    "pcd_rulebook_conv_cm_fill": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp,
                                       _vp, _i, _vp, _vp, _sz, _vp]),
    "pcd_rulebook_conv_cm_build": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp,
                                        _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "pcd_conv_out_shape": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "pcd_rulebook_conv_count": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _i]),
    "pcd_rulebook_conv_fill": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp,
                                    _vp, _sz, _vp, _i]),
    "pcd_rulebook_conv_build": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp,
                                     _i, _vp, _vp, _vp, _sz, _vp, _i]),
    "pcd_debug_stamp": (_i, [_vp, _vp]),
    "pcd_debug_stream_create_cu_mask": (_i, [_vp, _i, _vp]),
    "pcd_debug_spin": (_i, [_i, ctypes.c_ulonglong, _vp, _vp]),
    "pcd_debug_spin_shape": (_i, [_i, _i, _i, _i, ctypes.c_ulonglong, _vp]),
    "pcd_pull_from_host": (_i, [_vp, _i, _vp, _vp, _sz, _i, _vp]),
    "pcd_counter_add": (_i, [_vp, _i, _vp]),
    "pcd_conv2d_packed_weight_bytes": (_sz, [_i, _i, _i]),
    "pcd_conv2d_pack_weight": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "pcd_conv2d_pack_weights_batched": (_i, [_vp, _i, _i, _vp]),
    "pcd_conv2d_3x3_nhwc": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "pcd_conv2d_3x3_nhwc_ld": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "pcd_conv2d_wgrad_3x3_splits": (_i, [_i, _i, _i, _i, _i]),
    "pcd_conv2d_wgrad_3x3_nhwc": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "pcd_conv2d_3x3_tiles": (_i, [_i, _i, _i]),
    "pcd_conv2d_3x3_nhwc_bn": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "pcd_conv2d_wgrad_planes_splits": (_i, [_i, _i, _i, _i, _i, _i]),
    "pcd_conv2d_wgrad_planes_nhwc": (_i, [_i, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "pcd_conv2d_planes_nhwc": (_i, [_i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "pcd_sparse_conv_gather_gemm_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "pcd_sparse_conv_wgrad_f32": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "pcd_packed_weight_bytes": (_sz, [_i, _i, _i, _i]),
    "pcd_pack_weight": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "pcd_pack_weights_batched": (_i, [_vp, _i, _i, _vp]),
    "pcd_sparse_conv_gather_gemm": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp,
                                         _vp]),
    "pcd_sparse_conv_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "pcd_sparse_conv_wgrad": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "pcd_sparse_conv_wgrad_v2": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "pcd_sparse_conv_wgrad_classes": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                           _vp, _i]),
    "pcd_sparse_conv_dgrad_classes_v2": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp,
                                              _vp, _vp]),
    "pcd_sparse_conv_gather_gemm_packed": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "pcd_rulebook_conv_cm_build_compact": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp, _i,
                                                _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "pcd_rulebook_conv_expand_nbr_out": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "pcd_rulebook_conv_expand_nbr_in": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "pcd_sparse_conv_wgrad_reduce": (_i, [_i, _i, _i, _i, _vp, _vp, _vp]),
    "pcd_sparse_conv_wgrad_reduce_batched": (_i, [_vp, _i, _vp]),
    "pcd_sparse_conv_wgrad_os_splits": (_i, [_i, _i, _i, _i]),
    "pcd_sparse_conv_wgrad_os": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _sz, _vp]),
    "pcd_bev_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "pcd_bev_scatter": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "pcd_bev_gather": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "pcd_bev_scatter_nhwc": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "pcd_bev_gather_nhwc": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "pcd_bn_workspace_bytes": (_sz, [_i]),
    "pcd_col_sum": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "pcd_bn_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, ctypes.c_float, ctypes.c_float, _i, _vp, _vp, _i,
                            _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "pcd_bn_forward_ld": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, ctypes.c_float, ctypes.c_float, _i, _vp, _vp, _i,
                               _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "pcd_bn_backward_ld": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                _i, _vp, _vp, _sz, _vp]),
    "pcd_adam_flat_workspace_bytes": (_sz, []),
    "pcd_adam_flat_step": (_i, [_vp, _vp, _vp, _vp, _sz, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _sz,
                                _vp]),
    "pcd_adam_flat_step_v2": (_i, [_vp, _vp, _vp, _vp, _sz, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                   ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _i, _vp, _vp, _vp,
                                   _vp, _sz, _vp]),
    "pcd_adam_flat_step_v3": (_i, [_vp, _vp, _vp, _vp, _sz, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                   ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _i, _vp, _vp, _i,
                                   _vp, _vp, _vp, _sz, _vp]),
    "pcd_adam_flat_step_v4": (_i, [_vp, _vp, _i, _vp, _vp, _sz, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                   ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _i, _vp, _vp, _i,
                                   _vp, _vp, _vp, _sz, _vp]),
    "pcd_stream_capture_id": (_i, [_vp, _vp]),
    "pcd_dot_bf16_workspace_bytes": (_sz, []),
    "pcd_dot_bf16": (_i, [_vp, _vp, _sz, _vp, _vp, _sz, _vp]),
    "pcd_scale_bf16": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "pcd_static_overflow_check": (_i, [_vp, _i, _vp, _vp]),
    "pcd_com_cluster_groups": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "pcd_com_assign_workspace_bytes": (_sz, [_i, _i]),
    "pcd_com_assign_targets": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, ctypes.c_float, _i, _vp, _vp,
                                    _i, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "pcd_com_loss_workspace_bytes": (_sz, [_i, _i]),
    "pcd_com_loss_forward": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i,
                                  _vp, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _sz, _vp]),
    "pcd_com_loss_backward": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i,
                                   _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp]),
    "pcd_centerhead_loss_workspace_bytes": (_sz, [_i]),
    "pcd_centerhead_loss_forward": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i,
                                         _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _sz, _vp]),
    "pcd_centerhead_loss_backward": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp,
                                          _vp, _vp, _i, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp]),
    "pcd_centerhead_assign_workspace_bytes": (_sz, [_i, _i]),
    "pcd_centerhead_assign_targets": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, ctypes.c_float, _i, _vp,
                                           _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcd_ball_query_stack": (_i, [_i, _i, ctypes.c_float, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcd_group_points_stack": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcd_group_points_stack_grad": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcd_stack_farthest_point_sampling": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcd_stack_fps_coop_workspace_bytes": (_sz, [_i]),
    "pcd_stack_fps_buckets_workspace_bytes": (_sz, [_i, _i]),
    "pcd_stack_farthest_point_sampling_buckets": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "pcd_stack_farthest_point_sampling_coop": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "pcd_three_nn_stack": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcd_three_interpolate_stack": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp]),
    "pcd_three_interpolate_stack_grad": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp]),
    "pcd_voxel_query_stack": (_i, [_i, _i, _i, _i, _i, ctypes.c_float, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcd_fp8_packed_weight_bytes": (_sz, [_i, _i, _i]),
    "pcd_fp8_pack_weight": (_i, [_vp, _i, _i, _i, _i, ctypes.c_float, _vp, _vp]),
    "pcd_fp8_quantize": (_i, [_vp, _i, _i, _vp, _i, _i, _i, ctypes.c_float, _vp, _vp]),
    "pcd_fp8_dequantize": (_i, [_vp, _sz, ctypes.c_float, _vp, _vp]),
    "pcd_sparse_conv_gather_gemm_fp8": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i,
                                             ctypes.c_float, _vp, _i, _i, _vp]),
    "pcd_pillar_decorate": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "pcd_pfn_relu_pool": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "pcd_pfn_relu_pool_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "pcd_boxes_overlap_bev": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp]),
    "pcd_boxes_iou_bev_host": (_i, [_vp, _i, _vp, _i, _vp]),
    "pcd_nms_workspace_bytes": (_sz, [_i]),
    "pcd_nms_bev": (_i, [_vp, _i, ctypes.c_float, _i, _vp, _vp, _vp, _sz, _vp]),
    "pcd_bn_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                             _i, _vp, _vp, _sz, _vp]),
    "pcd_bn_backward_colsum_rows": (_i, [_i, _i, _i]),
    "pcd_col_sum_finalize": (_i, [_vp, _i, _vp]),
}

_lib = None


class PcdError(RuntimeError):
    pass


class PcdColsumJob(ctypes.Structure):
    """include/pcd_ops.h: struct PcdColsumJob."""
    _fields_ = [("partial", ctypes.c_void_p), ("out", ctypes.c_void_p), ("rows", ctypes.c_int), ("c", ctypes.c_int)]


COLSUM_MAX_JOBS = 32


class PcdWgradReduceJob(ctypes.Structure):
    """include/pcd_ops.h: struct PcdWgradReduceJob."""
    _fields_ = [("workspace", ctypes.c_void_p), ("dweight", ctypes.c_void_p), ("kvol", ctypes.c_int),
                ("cin", ctypes.c_int), ("cout", ctypes.c_int), ("pmax", ctypes.c_int), ("splits", ctypes.c_int),
                ("layout", ctypes.c_int), ("cout_write", ctypes.c_int), ("cin_write", ctypes.c_int)]


WGRAD_MAX_JOBS = 32
COUNT_CHECK_MAX = 24


class PcdCountCheck(ctypes.Structure):
    """include/pcd_ops.h: struct PcdCountCheck (static-shape overflow guard)."""
    _fields_ = [("count", ctypes.c_void_p * COUNT_CHECK_MAX), ("cap", ctypes.c_int32 * COUNT_CHECK_MAX)]


BN_MID_ROWS = 16          # include/pcd_ops.h: PCD_BN_MID_ROWS
BN_EXT_MID = -1           # include/pcd_ops.h: PCD_BN_EXT_MID
BN_COUNTER_STRIDE = 32    # include/pcd_ops.h: PCD_BN_COUNTER_STRIDE


class PcdComCurriculum(ctypes.Structure):
    """include/pcd_ops.h: struct PcdComCurriculum (LOSS_CURRICULUM of the COM head)."""
    _fields_ = [("ucl", ctypes.c_int), ("fix_threshold", ctypes.c_int), ("straight", ctypes.c_int),
                ("tuning", ctypes.c_int), ("only_center", ctypes.c_int), ("apply", ctypes.c_int), ("add", ctypes.c_int),
                ("radius", ctypes.c_int), ("k_straight", ctypes.c_double), ("elongation", ctypes.c_double),
                ("height", ctypes.c_double), ("alpha", ctypes.c_double), ("threshold", ctypes.c_double),
                ("conf_classes", ctypes.c_int), ("conf_groups", ctypes.c_int)]


# include/pcd_ops_experiments.h: present only in a library built with `make EXPERIMENTS=1` (measured-slower kernels kept for
# reproduction); bound when exported, never required
EXPERIMENT_PROTOTYPES = {
    "pcd_sparse_conv_pairs_seg_bytes": (_sz, [_i, _i]),
    "pcd_sparse_conv_pairs_seg": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "pcd_sparse_conv_pairs_tiles": (_i, [_i, _i, _i, _i]),
    "pcd_sparse_conv_pairs": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "pcd_sparse_conv_gather_gemm_zfast": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp,
                                         _vp]),
}

PCD_COM_CLUSTER_X5 = 0


class PcdBnReduce(ctypes.Structure):
    """include/pcd_ops.h: struct PcdBnReduce (conv-epilogue reductions for the BatchNorm beside the conv)."""
    _fields_ = [("mode", ctypes.c_int), ("relu", ctypes.c_int), ("x", ctypes.c_void_p), ("y", ctypes.c_void_p),
                ("mean", ctypes.c_void_p), ("invstd", ctypes.c_void_p), ("partial", ctypes.c_void_p),
                ("partial_rows", ctypes.c_int), ("mid", ctypes.c_void_p), ("counters", ctypes.c_void_p)]


def build(force=False):
    """hipcc --offload-arch=gfx950 -> com_amd/lib/libpcdops_hip.so (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j4"])
    return LIB_PATH


def lib():
    """Load the HIP library; raise if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        # torch bundles its own HIP runtime (libamdhip64); it must be loaded FIRST so that this library's
        # NEEDED entry binds to the same runtime instance (two runtimes in one process do not share devices,
        # streams or allocations).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise PcdError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C com_amd/csrc`. There is no CPU fallback for the hot path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in EXPERIMENT_PROTOTYPES.items():
            fn = getattr(handle, name, None)
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        # (tuning options are set through set_option -- bench.py / tools/ forward PCD_OPT_<KEY> environment variables with
        #  tools/env_switches.py; neither this module nor the library reads the environment)
        _lib = handle
    return _lib


def use_experiments_library():
    """Tools that reproduce the measured-slower kernels (tools/exp_ggwin.py, tools/exp_pconv.py): load the EXPERIMENTS build
    instead of the default library -- call it before the first op of the process."""
    global LIB_PATH
    if _lib is not None:
        raise PcdError("use_experiments_library() must come before the first call into the library")
    if not os.path.exists(EXPERIMENTS_LIB_PATH):
        raise PcdError(f"{EXPERIMENTS_LIB_PATH} not found: make -C com_amd/csrc EXPERIMENTS=1")
    LIB_PATH = EXPERIMENTS_LIB_PATH


_has_experiments = None


def has_experiments():
    """True if the loaded library exports the entry points of include/pcd_ops_experiments.h."""
    global _has_experiments
    if _has_experiments is None:
        _has_experiments = all(hasattr(lib(), name) for name in EXPERIMENT_PROTOTYPES)
    return _has_experiments


def set_option(key, value):
    """PROCESS-WIDE (one table per loaded library, read at launch time without synchronisation): set options before
    the first launch, not from concurrent threads; tests restore what they change (tests/conftest.py::pcd_option)."""
    check(lib().pcd_set_option(key.encode(), int(value)), f"pcd_set_option({key})")


def get_option(key):
    v = ctypes.c_int(0)
    check(lib().pcd_get_option(key.encode(), ctypes.byref(v)), f"pcd_get_option({key})")
    return int(v.value)


def check(code, what):
    if code != 0:
        msg = lib().pcd_error_string(code).decode()
        if code == -5:
            msg += " [" + lib().pcd_last_hip_error_string().decode() + "]"
        raise PcdError(f"{what} failed: {msg} (code {code})")


def host_f32(values):
    arr = (ctypes.c_float * len(values))(*[float(v) for v in values])
    return arr


def host_i32(values):
    arr = (ctypes.c_int * len(values))(*[int(v) for v in values])
    return arr


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
