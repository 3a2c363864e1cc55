"""The numpy restatement of the COM curriculum head (oracle/com_oracle.py) against fixtures G11 / G12 / G14, which hold
outputs of the REFERENCE'S OWN classes (tests/golden/make_golden.py::g11/g12/g14: CurriculumCenterHead.cluster /
assign_targets / get_loss, FocalLossCenterCurriculum, RegLossCenterNet extracted from /root/reference at generation
time).  Bars: group ids, inds, radius_map, masks, counts bit-exact; heat maps / regression targets 1e-6 (libm log / cos /
sin / exp may differ in the last ulp between numpy and torch); sums, losses and gradients 1e-6 relative (2e-5 where float32
torch summation order of ~4000 terms enters)."""
import json
import os

import numpy as np
import pytest

from oracle import com_oracle as C

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ["Vehicle", "Pedestrian", "Cyclist"]
WAYMO_RANGE = [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
WAYMO_VOXEL = [0.1, 0.1, 0.15]


def _load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


def _dense(g, prefix, shape, fill=0.0):
    a = np.full(shape, fill, np.float32)
    nz = g[prefix + "_nz"]
    a[tuple(nz[:, i] for i in range(nz.shape[1]))] = g[prefix + "_val"]
    return a


def test_cluster_groups_match_the_reference_g11():
    g = _load("g11_com_targets")
    got = C.cluster_groups(g["gt_boxes"], g["true_object"], g["occupancy_ratio"], g["facade_type"])
    np.testing.assert_array_equal(got, g["group"])
    assert len(np.unique(g["group"])) > 25 and g["group"].max() > 60


@pytest.mark.parametrize("layout", ["one", "two"])
@pytest.mark.parametrize("gate", ["nogate", "gate", "late"])
def test_assign_targets_match_the_reference_g11(layout, gate):
    g = _load("g11_com_targets")
    heads = [NAMES] if layout == "one" else [["Vehicle"], ["Pedestrian", "Cyclist"]]
    epoch, thr, minp = {"nogate": (3, 100, 0), "gate": (3, 100, 5), "late": (101, 100, 5)}[gate]
    H, W = (int(v) for v in g["feature_map_size"])
    nmax = int(g["num_max_objs"][0])
    td = C.assign_targets(g["gt_boxes"], g["num_points_in_gt"], g["group"], NAMES, heads, WAYMO_RANGE, WAYMO_VOXEL,
                          [H, W], int(g["stride"][0]), nmax, 0.1, 2, epoch, thr, minp)
    for hi, head in enumerate(heads):
        k = f"{layout}_{gate}_h{hi}"
        np.testing.assert_array_equal(td["inds"][hi], g[k + "_inds"])
        np.testing.assert_array_equal(td["masks"][hi], g[k + "_mask"])
        np.testing.assert_array_equal(td["radius_map"][hi], g[k + "_radius_map"])
        np.testing.assert_allclose(td["target_boxes"][hi], g[k + "_boxes"], rtol=1e-6, atol=1e-6)
        want = _dense(g, k + "_heat", (g["gt_boxes"].shape[0], len(head), H, W))
        np.testing.assert_allclose(td["heatmaps"][hi], want, rtol=0, atol=1e-7)
        assert ((td["heatmaps"][hi] == 1) == (want == 1)).all()
    if gate == "gate":          # the gate really drops objects
        assert g["one_gate_h0_mask"].sum() < g["one_nogate_h0_mask"].sum()


def _case(g, name):
    cur = json.loads(str(g[name + "_curriculum"]))
    return cur, int(g[name + "_epoch"][0]), int(g[name + "_steps"][0])


def test_com_loss_matches_the_reference_g12():
    g = _load("g12_com_loss")
    H, W = (int(v) for v in g["feature_map_size"])
    order = [str(n) for n in g["head_order"]]
    seen_weighted = False
    for name in (str(n) for n in g["cases"]):
        cur, epoch, steps = _case(g, name)
        state = C.ComLossState()
        for st in range(steps):
            k = f"{name}_s{st}"
            B = g[k + "_hm_logit"].shape[0]
            group = C.cluster_groups(g[k + "_gt_boxes"], g[k + "_true_object"], g[k + "_occupancy_ratio"],
                                     g[k + "_facade_type"])
            td = C.assign_targets(g[k + "_gt_boxes"], g[k + "_num_points_in_gt"], group, NAMES, [NAMES],
                                  g["point_cloud_range"], g["voxel_size"], [H, W], int(g["stride"][0]),
                                  int(g["num_max_objs"][0]), 0.1, 2, epoch, 100, 0)
            np.testing.assert_array_equal(td["radius_map"][0], g[k + "_radius_map"])
            np.testing.assert_array_equal(td["inds"][0], g[k + "_inds"])
            np.testing.assert_array_equal(td["masks"][0], g[k + "_masks"])
            heat = _dense(g, k + "_heat", (B, 3, H, W))
            np.testing.assert_allclose(td["heatmaps"][0], heat, atol=1e-7)
            # the loss on the FIXTURE's targets (so that a last-ulp difference in log() of a box size cannot leak in)
            tg = dict(heatmap=heat, radius_map=g[k + "_radius_map"], masks=g[k + "_masks"], inds=g[k + "_inds"],
                      target_boxes=g[k + "_target_boxes"])
            regs = [g[f"{k}_{n}"] for n in order]
            r = C.com_loss(g[k + "_hm_logit"], regs, tg, cur, epoch, state, (3, 96), float(g["cls_weight"][0]),
                           float(g["loc_weight"][0]), g["code_weights"])
            np.testing.assert_array_equal(r["conf_num"], g[k + "_num_all"])
            np.testing.assert_allclose(r["conf_sum"], g[k + "_confidence_all"], rtol=1e-6, atol=1e-7)
            want_mask = _dense(g, k + "_heatmap_mask_after", (B, 3, H, W), fill=1.0)
            # (weights follow sigmoid(x) at the object's centre: numpy's and torch's float32 exp differ in the last ulp)
            np.testing.assert_array_equal(r["heatmap_mask"] != 1, want_mask != 1)
            np.testing.assert_allclose(r["heatmap_mask"], want_mask, rtol=5e-7, atol=0)
            seen_weighted |= bool((want_mask != 1).any())
            if name == "nopos":
                assert np.isnan(g[k + "_confidence"][0]) and np.isnan(r["avg_confidence"])
            else:
                np.testing.assert_allclose(r["avg_confidence"], g[k + "_confidence"][0], rtol=2e-6)
                np.testing.assert_allclose(state.avg_confidence, g[k + "_avg_confidence_ema"][0], rtol=2e-6)
            np.testing.assert_allclose(r["hm_loss"], g[k + "_hm_loss"][0], rtol=2e-5)
            np.testing.assert_allclose(r["loc_loss"], g[k + "_loc_loss"][0], rtol=2e-5, atol=1e-7)
            np.testing.assert_allclose(r["loss"], g[k + "_loss"][0], rtol=2e-5)
            gh = g[k + "_grad_hm_logit"]
            np.testing.assert_allclose(r["grad_hm_logit"], gh, rtol=2e-4, atol=2e-6 * np.abs(gh).max())
            for n, d in zip(order, r["grad_regs"]):
                np.testing.assert_allclose(d, g[f"{k}_grad_{n}"], rtol=1e-5, atol=1e-7)
    assert seen_weighted


def test_epoch_gather_matches_the_reference_arithmetic_g14():
    g = _load("g14_com_epoch_gather")
    conf, num = g["conf"], g["num"]
    got = C.epoch_gather([list(conf[r]) for r in range(conf.shape[0])], [list(num[r]) for r in range(num.shape[0])])
    assert str(g["result_dtype"]) == "float32"
    np.testing.assert_array_equal(got, g["result"])
