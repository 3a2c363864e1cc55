"""GPU: the COM curriculum head (pcd_com_* behind com_amd.hotpath.com_head) against

  * fixtures G11 / G12 = outputs of the REFERENCE'S OWN CurriculumCenterHead.cluster / assign_targets / get_loss and
    FocalLossCenterCurriculum (tests/golden/make_golden.py::g11/g12), and
  * the numpy oracle (oracle/com_oracle.py, itself pinned by those fixtures on the CPU) at the full size of BASELINE
    config 3 (B = 4, 188 x 188 map at stride 8, 500 object slots), incl. bf16 channels-last predictions and a replayed
    hipGraph.

Bars: group ids, inds, masks, radius_map, counts: bit-exact.  Heat maps: same support, 1e-6.  Regression targets 1e-6
(device logf / cosf / sinf).  UCL weights / box_mask / heatmap_mask: same support, 2e-6 relative (the weight follows
sigmoid(x) at the centre; device expf vs torch's differ in the last ulp).  Sums, losses: 1e-6..2e-5 relative; gradients
2e-4 relative of the largest entry."""
import json
import os

import numpy as np
import pytest
import torch

from com_amd.utils import synth
from oracle import com_oracle as C

pytestmark = pytest.mark.gpu
NAMES = ["Vehicle", "Pedestrian", "Cyclist"]
ORDER = [("center", 2), ("center_z", 1), ("dim", 3), ("rot", 2)]


def _dense(g, prefix, shape, fill=0.0):
    a = np.full(shape, fill, np.float32)
    nz = g[prefix + "_nz"]
    a[tuple(nz[:, i] for i in range(nz.shape[1]))] = g[prefix + "_val"]
    return a


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_cluster_matches_reference_fixture_and_oracle(golden):
    from com_amd.hotpath import com_head as H
    g = golden("g11_com_targets")
    got = H.cluster(_cuda(g["gt_boxes"]), _cuda(g["true_object"]), _cuda(g["occupancy_ratio"]), _cuda(g["facade_type"]))
    assert got.dtype == torch.int64
    np.testing.assert_array_equal(got.cpu().numpy(), g["group"])
    # every bin edge from both sides, random bulk
    rng = np.random.default_rng(3)
    B, M = 4, 500
    gt = np.zeros((B, M, 8), np.float32)
    edges = np.array([29.999998, 30.0, 30.000002, 49.999996, 50.0, 50.000004, 5.0, 70.0], np.float32)
    ang = rng.uniform(0, 2 * np.pi, (B, M))
    r = rng.choice(edges, (B, M))
    gt[..., 0], gt[..., 1] = r * np.cos(ang), r * np.sin(ang)
    gt[..., 0][:, :50] = rng.choice(edges, (B, 50))
    gt[..., 1][:, :50] = 0
    gt[..., 3] = rng.choice(np.array([5.9999995, 6.0, 6.0000005, 3.0, 9.0], np.float32), (B, M))
    gt[..., 7] = rng.integers(0, 4, (B, M))
    thr = [c * 5 / 12 for c in (0.21, 0.41, 0.61, 0.81)] + [0.25, 0.5, 0.7]
    vals = np.array([np.nextafter(np.float32(t), np.float32(d)) for t in thr for d in (0, 1)] + [np.float32(t) for t in thr]
                    + [0.0, 1.0, 0.33], np.float32)
    occ = rng.choice(vals, (B, M)).astype(np.float32)
    fac = rng.integers(0, 5, (B, M)).astype(np.float32)          # 4 = not a facade type: cars stay in group 0
    to = rng.choice([0.0, 1.0, 2.0], (B, M)).astype(np.float32)
    got = H.cluster(_cuda(gt), _cuda(to), _cuda(occ), _cuda(fac)).cpu().numpy()
    np.testing.assert_array_equal(got, C.cluster_groups(gt, to, occ, fac))
    assert len(np.unique(got)) > 55


@pytest.mark.parametrize("layout", ["one", "two"])
@pytest.mark.parametrize("gate", ["nogate", "gate", "late"])
def test_com_targets_match_reference_fixture(golden, layout, gate):
    from com_amd.hotpath import com_head as H
    g = golden("g11_com_targets")
    heads = [NAMES] if layout == "one" else [["Vehicle"], ["Pedestrian", "Cyclist"]]
    epoch, thr, minp = {"nogate": (3, 100, 0), "gate": (3, 100, 5), "late": (101, 100, 5)}[gate]
    Hh, W = (int(v) for v in g["feature_map_size"])
    gt = _cuda(g["gt_boxes"])
    group = H.cluster(gt, _cuda(g["true_object"]), _cuda(g["occupancy_ratio"]), _cuda(g["facade_type"]))
    td = H.assign_targets(gt, [Hh, W], NAMES, heads, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, int(g["stride"][0]),
                          _cuda(g["num_points_in_gt"]), true_object=group, num_max_objs=int(g["num_max_objs"][0]),
                          gaussian_overlap=0.1, min_radius=2, epoch=epoch, epoch_threshold=thr, min_points=minp)
    for hi, head in enumerate(heads):
        k = f"{layout}_{gate}_h{hi}"
        assert td["masks"][hi].dtype == torch.float32 and td["radius_map"][hi].dtype == torch.int64
        np.testing.assert_array_equal(td["inds"][hi].cpu().numpy(), g[k + "_inds"])
        np.testing.assert_array_equal(td["masks"][hi].cpu().numpy(), g[k + "_mask"])
        np.testing.assert_array_equal(td["radius_map"][hi].cpu().numpy(), g[k + "_radius_map"])
        np.testing.assert_allclose(td["target_boxes"][hi].cpu().numpy(), g[k + "_boxes"], rtol=1e-6, atol=1e-6)
        hm = td["heatmaps"][hi].cpu().numpy()
        want = _dense(g, k + "_heat", hm.shape)
        assert np.array_equal(hm != 0, want != 0) and np.array_equal(hm == 1, want == 1)
        np.testing.assert_allclose(hm, want, rtol=0, atol=1e-6)
        assert bool((td["heatmap_mask"][hi] == 1).all()) and td["heatmap_mask"][hi].shape == td["heatmaps"][hi].shape


def _run_head(mod, hm_logit, regs, tg, epoch):
    pred = {"hm": hm_logit.requires_grad_(True)}
    for (n, _), r in zip(ORDER, regs):
        pred[n] = r.requires_grad_(True)
    loss, tb = mod([pred], tg, epoch=epoch)
    loss.backward()
    return loss, tb, pred


def test_com_loss_matches_reference_fixture_g12(golden):
    from com_amd.hotpath import com_head as H
    g = golden("g12_com_loss")
    Hh, W = (int(v) for v in g["feature_map_size"])
    seen_weighted = False
    for name in (str(n) for n in g["cases"]):
        cur = json.loads(str(g[name + "_curriculum"]))
        epoch, steps = int(g[name + "_epoch"][0]), int(g[name + "_steps"][0])
        mod = H.CurriculumCenterHeadLoss([n for n, _ in ORDER], cur, conf_shape=(3, 96), cls_weight=1.0, loc_weight=2.0,
                                         code_weights=g["code_weights"].tolist()).cuda()
        conf_running = np.zeros((3, 96), np.float32)
        for st in range(steps):
            k = f"{name}_s{st}"
            gt = _cuda(g[k + "_gt_boxes"])
            group = H.cluster(gt, _cuda(g[k + "_true_object"]), _cuda(g[k + "_occupancy_ratio"]),
                              _cuda(g[k + "_facade_type"]))
            tg = H.assign_targets(gt, [Hh, W], NAMES, [NAMES], g["point_cloud_range"].tolist(), g["voxel_size"].tolist(),
                                  int(g["stride"][0]), _cuda(g[k + "_num_points_in_gt"]), true_object=group,
                                  num_max_objs=int(g["num_max_objs"][0]), epoch=epoch, epoch_threshold=100, min_points=0)
            np.testing.assert_array_equal(tg["radius_map"][0].cpu().numpy(), g[k + "_radius_map"])
            np.testing.assert_array_equal(tg["masks"][0].cpu().numpy(), g[k + "_masks"])
            # loss on the fixture's regression targets (a last-ulp logf difference must not leak into the L1 sums' bar)
            tg["target_boxes"][0] = _cuda(g[k + "_target_boxes"])
            regs = [_cuda(g[f"{k}_{n}"]) for n, _ in ORDER]
            loss, tb, pred = _run_head(mod, _cuda(g[k + "_hm_logit"]), regs, tg, epoch)
            st_ = mod.hm_loss_func
            np.testing.assert_array_equal(st_.confidence_all[1].cpu().numpy(), g[k + "_num_all"])
            np.testing.assert_allclose(st_.confidence_all[0].cpu().numpy(), g[k + "_confidence_all"], rtol=2e-6, atol=1e-7)
            conf_running = conf_running + g[k + "_confidence_all"]
            np.testing.assert_allclose(st_.epoch_confidence.cpu().numpy(), conf_running, rtol=2e-6, atol=1e-6)
            B = gt.shape[0]
            want_mask = _dense(g, k + "_heatmap_mask_after", (B, 3, Hh, W), fill=1.0)
            got_mask = tg["heatmap_mask"][0].cpu().numpy()
            np.testing.assert_array_equal(got_mask != 1, want_mask != 1)
            np.testing.assert_allclose(got_mask, want_mask, rtol=2e-6, atol=0)
            seen_weighted |= bool((want_mask != 1).any())
            assert int(mod.hm_loss_func._owner.abs().sum().item()) == 0 if cur.get("UCL", True) else True
            if name == "nopos":
                assert np.isnan(float(tb["confidence"]))
            else:
                np.testing.assert_allclose(float(tb["confidence"]), g[k + "_confidence"][0], rtol=2e-6)
                np.testing.assert_allclose(st_.avg_confidence, g[k + "_avg_confidence_ema"][0], rtol=2e-6)
            np.testing.assert_allclose(float(tb["hm_loss_head_0"]), g[k + "_hm_loss"][0], rtol=2e-5)
            np.testing.assert_allclose(float(tb["loc_loss_head_0"]), g[k + "_loc_loss"][0], rtol=2e-5, atol=1e-7)
            np.testing.assert_allclose(float(loss), g[k + "_loss"][0], rtol=2e-5)
            gh = g[k + "_grad_hm_logit"]
            np.testing.assert_allclose(pred["hm"].grad.cpu().numpy(), gh, rtol=2e-4, atol=2e-6 * np.abs(gh).max())
            for n, _ in ORDER:
                np.testing.assert_allclose(pred[n].grad.cpu().numpy(), g[f"{k}_grad_{n}"], rtol=2e-5, atol=1e-7)
    assert seen_weighted


def _full_size_problem(seed, B=4, M=200, n_obj=(150, 60, 0, 199), code=8):
    rng = np.random.default_rng(seed)
    gt = np.zeros((B, M, code), np.float32)
    npgt = np.zeros((B, M), np.float32)
    to, occ, fac = (np.zeros((B, M), np.float32) for _ in range(3))
    for b in range(B):
        n = n_obj[b]
        gt[b, :n, 0:2] = rng.uniform(-74, 74, (n, 2))
        gt[b, :n, 2] = rng.uniform(-1, 2, n)
        cls = rng.integers(1, 4, n)
        gt[b, :n, 3] = np.where(cls == 1, rng.uniform(3.5, 12, n), rng.uniform(0.5, 2.0, n))
        gt[b, :n, 4] = np.where(cls == 1, rng.uniform(1.6, 3.0, n), rng.uniform(0.4, 1.0, n))
        gt[b, :n, 5] = rng.uniform(1.0, 3.0, n)
        gt[b, :n, 6] = rng.uniform(-np.pi, np.pi, n)
        gt[b, :n, code - 1] = cls
        npgt[b, :n] = rng.integers(0, 40, n)
        to[b, :n] = rng.choice([1, 1, 1, 2], n)
        occ[b, :n] = rng.random(n)
        fac[b, :n] = rng.integers(0, 4, n)
    return gt, npgt, to, occ, fac


@pytest.mark.parametrize("cur", [dict(UCL=False, FIX=True), dict(UCL=True, FIX=False, ALPHA=0.25, ADD=1, HEIGHT=0.9)])
@pytest.mark.parametrize("layout", ["f32_nchw", "bf16_nhwc"])
def test_com_head_full_size_against_the_oracle(cur, layout):
    """BASELINE config 3 sizes: B = 4, 188 x 188 at stride 8, 500 slots (one frame empty, one nearly full)."""
    from com_amd.hotpath import com_head as H
    Hh = W = 188
    gt, npgt, to, occ, fac = _full_size_problem(31)
    group = H.cluster(_cuda(gt), _cuda(to), _cuda(occ), _cuda(fac))
    np.testing.assert_array_equal(group.cpu().numpy(), C.cluster_groups(gt, to, occ, fac))
    tg = H.assign_targets(_cuda(gt), [Hh, W], NAMES, [NAMES], synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 8, _cuda(npgt),
                          true_object=group, num_max_objs=500, epoch=2, epoch_threshold=100, min_points=3)
    ref = C.assign_targets(gt, npgt, group.cpu().numpy(), NAMES, [NAMES], synth.WAYMO_RANGE, synth.WAYMO_VOXEL, [Hh, W],
                           8, 500, 0.1, 2, 2, 100, 3)
    for key in ("inds", "masks", "radius_map"):
        np.testing.assert_array_equal(tg[key][0].cpu().numpy(), ref[key][0])
    np.testing.assert_allclose(tg["target_boxes"][0].cpu().numpy(), ref["target_boxes"][0], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(tg["heatmaps"][0].cpu().numpy(), ref["heatmaps"][0], rtol=0, atol=1e-6)
    assert np.array_equal(tg["heatmaps"][0].cpu().numpy() == 1, ref["heatmaps"][0] == 1)
    rng = np.random.default_rng(5)
    B = gt.shape[0]
    hm_logit = (rng.standard_normal((B, 3, Hh, W)) * 2 - 1).astype(np.float32)
    regs = [rng.standard_normal((B, c, Hh, W)).astype(np.float32) for _, c in ORDER]
    if layout == "bf16_nhwc":
        conv = lambda a: _cuda(a).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    else:
        conv = _cuda
    t_hm, t_regs = conv(hm_logit), [conv(r) for r in regs]
    hm_used = t_hm.float().cpu().numpy()
    regs_used = [r.float().cpu().numpy() for r in t_regs]
    mod = H.CurriculumCenterHeadLoss([n for n, _ in ORDER], cur, conf_shape=(3, 96)).cuda()
    state = C.ComLossState()
    for step in range(2):                                    # two steps: the EMA threshold of step 2 follows step 1
        tgs = {k: [v[0].clone()] for k, v in tg.items() if v}
        loss, tb, pred = _run_head(mod, t_hm.clone(), [r.clone() for r in t_regs], tgs, epoch=2)
        tref = dict(heatmap=tg["heatmaps"][0].cpu().numpy(), radius_map=ref["radius_map"][0], masks=ref["masks"][0],
                    inds=ref["inds"][0], target_boxes=tg["target_boxes"][0].cpu().numpy())
        r = C.com_loss(hm_used, regs_used, tref, cur, 2, state, (3, 96), 1.0, 2.0, None)
        np.testing.assert_array_equal(mod.hm_loss_func.confidence_all[1].cpu().numpy(), r["conf_num"])
        np.testing.assert_allclose(mod.hm_loss_func.confidence_all[0].cpu().numpy(), r["conf_sum"], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(mod.hm_loss_func.avg_confidence, state.avg_confidence, rtol=2e-6)
        got_mask = tgs["heatmap_mask"][0].cpu().numpy()
        np.testing.assert_array_equal(got_mask != 1, r["heatmap_mask"] != 1)
        np.testing.assert_allclose(got_mask, r["heatmap_mask"], rtol=2e-6)
        np.testing.assert_allclose(tb["box_mask_head_0"].cpu().numpy(), r["box_mask"], rtol=2e-6)
        np.testing.assert_allclose(float(tb["hm_loss_head_0"]), r["hm_loss"], rtol=2e-5)
        np.testing.assert_allclose(float(tb["loc_loss_head_0"]), r["loc_loss"], rtol=2e-5)
        gh = r["grad_hm_logit"]
        tol = 2e-4 if layout == "f32_nchw" else 6e-3        # (bf16 gradient storage: 2^-8 relative rounding)
        np.testing.assert_allclose(pred["hm"].grad.float().cpu().numpy(), gh, rtol=tol, atol=tol * 1e-2 * np.abs(gh).max())
        for (n, _), d in zip(ORDER, r["grad_regs"]):
            np.testing.assert_allclose(pred[n].grad.float().cpu().numpy(), d, rtol=tol, atol=1e-7)
    assert int(mod.hm_loss_func.confidence_all[1].sum().item()) > 100


def test_com_head_step_is_capturable_and_replays_identically():
    """cluster -> assign_targets -> loss -> backward in ONE hipGraph: replay == eager (bit for bit), epoch sums advance
    per replay, nothing synchronises."""
    from com_amd.hotpath import com_head as H
    Hh = W = 188
    gt, npgt, to, occ, fac = (_cuda(a) for a in _full_size_problem(41))
    cur = dict(UCL=True, FIX=False, ALPHA=0.2)
    torch.manual_seed(0)
    hm = torch.randn(4, 3, Hh, W, device="cuda", requires_grad=True)
    regs = [torch.randn(4, c, Hh, W, device="cuda", requires_grad=True) for _, c in ORDER]

    def step(mod):
        group = H.cluster(gt, to, occ, fac)
        tg = H.assign_targets(gt, [Hh, W], NAMES, [NAMES], synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 8, npgt, true_object=group)
        pred = {"hm": hm}
        pred.update({n: r for (n, _), r in zip(ORDER, regs)})
        loss, tb = mod([pred], tg, epoch=1)
        grads = torch.autograd.grad(loss, [hm] + regs)
        return loss.detach(), grads, tg["heatmap_mask"][0]

    eager = H.CurriculumCenterHeadLoss([n for n, _ in ORDER], cur).cuda()
    outs = [step(eager) for _ in range(3)]
    graphed = H.CurriculumCenterHeadLoss([n for n, _ in ORDER], cur).cuda()
    step(graphed)                                            # warm-up (allocations), then reset the state it advanced
    graphed.hm_loss_func.state.zero_()
    graphed.hm_loss_func.start_epoch()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        l_g, g_g, m_g = step(graphed)
    for i in range(3):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(l_g, outs[i][0]), i
        for a, b in zip(g_g, outs[i][1]):
            assert torch.equal(a, b)
        assert torch.equal(m_g, outs[i][2])
    assert torch.equal(graphed.hm_loss_func.epoch_num, eager.hm_loss_func.epoch_num)
    assert torch.equal(graphed.hm_loss_func.state, eager.hm_loss_func.state)
    assert float(graphed.hm_loss_func.epoch_num.sum()) == 3 * float(graphed.hm_loss_func.confidence_all[1].sum())
