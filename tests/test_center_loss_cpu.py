"""CenterHead losses (com_amd/hotpath/center_loss.py) against fixture G10 = the reference's own `neg_loss_cornernet` /
`_reg_loss` / `_transpose_and_gather_feat` and the `get_loss` arithmetic (tests/golden/make_golden.py::g10): values
and gradients, including a head without a single positive (the reference's `if num_pos == 0` branch, which the
device-side form replaces by clamp_min).  Pure torch: runs on the CPU."""
import numpy as np
import torch

from com_amd.hotpath import center_loss as CL


def _t(a, grad=False):
    t = torch.from_numpy(np.array(a))
    return t.requires_grad_(True) if grad else t


def test_center_head_loss_matches_reference_functions(golden):
    g = golden("g10_center_loss")
    order = [str(n) for n in g["head_order"]]
    loss_mod = CL.CenterHeadLoss(order, cls_weight=float(g["cls_weight"][0]), loc_weight=float(g["loc_weight"][0]),
                                 code_weights=tuple(float(v) for v in g["code_weights"]))
    pred_dicts, targets = [], {"heatmaps": [], "target_boxes": [], "inds": [], "masks": []}
    for hi in (0, 1):
        d = {"hm": _t(g[f"h{hi}_hm_logit"], True)}
        for n in order:
            d[n] = _t(g[f"h{hi}_{n}"], True)
        pred_dicts.append(d)
        targets["heatmaps"].append(_t(g[f"h{hi}_heatmap"]))
        targets["target_boxes"].append(_t(g[f"h{hi}_target_boxes"]))
        targets["inds"].append(_t(g[f"h{hi}_inds"]))
        targets["masks"].append(_t(g[f"h{hi}_mask"]))
    loss, tb = loss_mod(pred_dicts, targets)
    loss.backward()
    np.testing.assert_allclose(loss.detach().numpy(), g["loss"][0], rtol=1e-6)
    for hi in (0, 1):
        np.testing.assert_allclose(tb[f"hm_loss_head_{hi}"].numpy(), g[f"h{hi}_hm_loss"][0], rtol=1e-6)
        np.testing.assert_allclose(tb[f"loc_loss_head_{hi}"].numpy(), g[f"h{hi}_loc_loss"][0], rtol=1e-6)
        np.testing.assert_allclose(pred_dicts[hi]["hm"].grad.numpy(), g[f"h{hi}_grad_hm_logit"], rtol=1e-5, atol=1e-9)
        for n in order:
            np.testing.assert_allclose(pred_dicts[hi][n].grad.numpy(), g[f"h{hi}_grad_{n}"], rtol=1e-5, atol=1e-9)
    # pieces
    hm0 = CL.sigmoid_clamped(_t(g["h0_hm_logit"]))
    l0, c0 = CL.neg_loss_cornernet(hm0, _t(g["h0_heatmap"]))
    np.testing.assert_allclose(float(c0), g["h0_confidence"][0], rtol=1e-6)
    hm1 = CL.sigmoid_clamped(_t(g["h1_hm_logit"]))
    l1, c1 = CL.neg_loss_cornernet(hm1, _t(g["h1_heatmap"]))
    assert np.isnan(g["h1_confidence"][0]) and bool(torch.isnan(c1))       # no positive: 0 / 0, as in the reference
    boxes = torch.cat([_t(g[f"h0_{n}"]) for n in order], 1)
    rl = CL.reg_loss(boxes, _t(g["h0_mask"]), _t(g["h0_inds"]), _t(g["h0_target_boxes"]))
    np.testing.assert_allclose(rl.numpy(), g["h0_reg_loss"], rtol=1e-6)


def test_center_head_loss_has_no_host_round_trip():
    """The loss module must not synchronise: no `.item()` / `.cpu()` / `.tolist()` in its code (the reference has
    three host round trips per head and step)."""
    import inspect
    code_lines = [ln.split("#")[0] for ln in inspect.getsource(CL).splitlines()
                  if not ln.strip().startswith(("#", '"""')) and "`" not in ln]
    assert not any(tok in ln for ln in code_lines for tok in (".item()", ".cpu()", ".tolist()"))
