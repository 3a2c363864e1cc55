"""`com_amd.hotpath.CurriculumCenterHead(_x5)`: the COM dense head as a registry drop-in.

CPU: box decoding against fixture G15 (outputs of the reference's own `decode_bbox_from_heatmap`,
centernet_utils.py:217-279) -- same torch ops, bit-identical.
GPU: the whole module -- training forward (towers -> cluster -> COM targets), get_loss + backward, eval forward with
rotated NMS; targets / group counts against the numpy oracle, kept boxes against the C oracle's greedy NMS."""
import os

import numpy as np
import pytest
import torch

from com_amd.utils import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_decode_matches_the_reference_fixture_g15():
    from com_amd.hotpath import curriculum_head as H
    g = np.load(os.path.join(HERE, "golden", "g15_decode.npz"))
    t = lambda k: torch.from_numpy(g[k])
    out = H.decode_bbox_from_heatmap(t("hm"), t("rot")[:, 0:1], t("rot")[:, 1:2], t("center"), t("center_z"), t("dim"),
                                     list(synth.WAYMO_RANGE), list(synth.WAYMO_VOXEL), int(g["stride"][0]), K=int(g["K"][0]),
                                     score_thresh=0.1, post_center_limit_range=t("limit"))
    for k, d in enumerate(out):
        np.testing.assert_array_equal(d["pred_boxes"].numpy(), g[f"boxes{k}"])
        np.testing.assert_array_equal(d["pred_scores"].numpy(), g[f"scores{k}"])
        np.testing.assert_array_equal(d["pred_labels"].numpy(), g[f"labels{k}"])


COM_HEAD_CFG = dict(       # tools/cfgs/waymo_models/com/centercurriculum_pillar_3cls_b2_com.yaml:124-173 on the stride-8 voxel map
    CLASS_AGNOSTIC=False, CLASS_NAMES_EACH_HEAD=[['Vehicle', 'Pedestrian', 'Cyclist']], SHARED_CONV_CHANNEL=64,
    USE_BIAS_BEFORE_NORM=True, NUM_HM_CONV=2,
    SEPARATE_HEAD_CFG=dict(HEAD_ORDER=['center', 'center_z', 'dim', 'rot'], HEAD_DICT={
        'center': {'out_channels': 2, 'num_conv': 2}, 'center_z': {'out_channels': 1, 'num_conv': 2},
        'dim': {'out_channels': 3, 'num_conv': 2}, 'rot': {'out_channels': 2, 'num_conv': 2}}),
    TARGET_ASSIGNER_CONFIG=dict(FEATURE_MAP_STRIDE=8, NUM_MAX_OBJS=500, GAUSSIAN_OVERLAP=0.1, MIN_RADIUS=2, MIN_POINTS=0),
    LOSS_CONFIG=dict(LOSS_WEIGHTS={'cls_weight': 1.0, 'loc_weight': 2.0, 'code_weights': [1.0] * 8}),
    POST_PROCESSING=dict(SCORE_THRESH=0.1, POST_CENTER_LIMIT_RANGE=[-80, -80, -10.0, 80, 80, 10.0], MAX_OBJ_PER_SAMPLE=500,
                         NMS_CONFIG=dict(NMS_TYPE='nms_gpu', NMS_THRESH=0.7, NMS_PRE_MAXSIZE=4096, NMS_POST_MAXSIZE=500)),
    LOSS_CURRICULUM=dict(UCL=False, THRESHOLD=0.2, ELONGATION=-10, HEIGHT=1, FIX=True))


@pytest.mark.gpu
def test_gpu_curriculum_center_head_module_train_and_eval():
    from com_amd import hotpath
    from oracle import com_oracle as C
    from oracle import oracle as O
    dev = "cuda"
    torch.manual_seed(2)
    names = ['Vehicle', 'Pedestrian', 'Cyclist']
    head = hotpath.CurriculumCenterHead_x5(COM_HEAD_CFG, 128, 3, names, [1504, 1504, 40], synth.WAYMO_RANGE,
                                           synth.WAYMO_VOXEL, predict_boxes_when_training=False).to(dev)
    assert {k.split('.')[0] for k in head.state_dict()} == {"shared_conv", "heads_list"}      # the reference's keys
    rng = np.random.default_rng(4)
    B, M = 2, 40
    gt = np.zeros((B, M, 8), np.float32)
    gt[:, :30, 0:2] = rng.uniform(-70, 70, (B, 30, 2))
    gt[:, :30, 2] = rng.uniform(-1, 2, (B, 30))
    gt[:, :30, 3:6] = rng.uniform(0.6, 6.0, (B, 30, 3))
    gt[:, :30, 6] = rng.uniform(-3, 3, (B, 30))
    gt[:, :30, 7] = rng.integers(1, 4, (B, 30))
    extra = dict(num_points_in_gt=rng.integers(0, 50, (B, M)).astype(np.float32),
                 true_object=np.where(gt[..., 7] > 0, 1.0, 0.0).astype(np.float32),
                 occupancy_ratio=rng.random((B, M)).astype(np.float32), facade_type=rng.integers(0, 4, (B, M)).astype(np.float32))
    sf = torch.randn(B, 128, 188, 188, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    sf.requires_grad_(True)
    dd = dict(spatial_features_2d=sf, gt_boxes=torch.from_numpy(gt).to(dev), batch_size=B,
              **{k: torch.from_numpy(v).to(dev) for k, v in extra.items()})
    head.train()
    head.epoch = 3
    head(dd)
    td = head.forward_ret_dict['target_dicts']
    grp = C.cluster_groups(gt, extra["true_object"], extra["occupancy_ratio"], extra["facade_type"])
    ref = C.assign_targets(gt, extra["num_points_in_gt"], grp, names, [names], synth.WAYMO_RANGE, synth.WAYMO_VOXEL,
                           [188, 188], 8, 500, 0.1, 2, 3, 100, 0)
    np.testing.assert_array_equal(td["radius_map"][0].cpu().numpy(), ref["radius_map"][0])
    np.testing.assert_array_equal(td["inds"][0].cpu().numpy(), ref["inds"][0])
    loss, tb = head.get_loss()
    loss.backward()
    assert torch.isfinite(loss) and sf.grad is not None and torch.isfinite(sf.grad.float()).all()
    assert float(sf.grad.float().abs().sum()) > 0
    np.testing.assert_array_equal(head.hm_loss_func.confidence_all[1].cpu().numpy(),
                                  C.group_confidence(np.zeros((B, 3, 188, 188), np.float32), ref["radius_map"][0], (3, 96))[1])
    # eval: the module runs end to end (towers in bf16 -> decode -> rotated NMS) ...
    head.eval()
    with torch.no_grad():
        out = head(dict(spatial_features_2d=sf.detach(), batch_size=B))
    assert len(out['final_box_dicts']) == B and out['final_box_dicts'][0]['pred_boxes'].shape[1] == 7
    # ... and on fp32 maps with DISTINCT scores (bf16 logits tie, and the order of equal scores is unspecified in the
    # reference too) the kept boxes are exactly what the C oracle's greedy NMS keeps of the decoded candidates
    from com_amd.hotpath import curriculum_head as H
    gen = torch.Generator(device=dev).manual_seed(9)
    r = lambda c, s=1.0: torch.randn(B, c, 188, 188, device=dev, generator=gen) * s
    pd = {'hm': r(3, 2.0) - 4.0, 'center': torch.rand(B, 2, 188, 188, device=dev, generator=gen), 'center_z': r(1),
          'dim': r(3, 0.3) + 0.5, 'rot': r(2)}
    with torch.no_grad():
        finals = head.generate_predicted_boxes(B, [pd])
        cands = H.decode_bbox_from_heatmap(pd['hm'].sigmoid(), pd['rot'][:, 0:1], pd['rot'][:, 1:2], pd['center'], pd['center_z'],
                                           pd['dim'].exp(), synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 8, K=500, score_thresh=0.1,
                                           post_center_limit_range=torch.tensor(
                                               COM_HEAD_CFG['POST_PROCESSING']['POST_CENTER_LIMIT_RANGE'], device=dev).float())
    for k in range(B):
        boxes, scores = cands[k]['pred_boxes'].cpu().numpy(), cands[k]['pred_scores'].cpu().numpy()
        assert len(boxes) > 50 and len(set(scores.tolist())) == len(scores)
        order = np.argsort(-scores, kind="stable")
        keep = order[O.nms_bev(boxes[order][:, :7], 0.7)]
        np.testing.assert_array_equal(finals[k]['pred_boxes'].cpu().numpy(), boxes[keep])
        assert 0 < len(keep) < len(boxes)
        assert finals[k]['pred_labels'].min() >= 1 and finals[k]['pred_labels'].max() <= 3
