"""GPU: channels-last BEV hand-off and the dense BEV stack.

  * pcd_bev_scatter_nhwc / pcd_bev_gather_nhwc are pure data movement: BIT-EXACT against the NCHW kernels (which
    are pinned by the reference HeightCompression fixture G5) and against the reference formula
    dense().view(N, C*D, H, W) on fixture G5 itself;
  * BaseBEVBackbone + CenterHeadTowers in bf16 / channels_last against the SAME modules run in float32 NCHW by plain
    torch (the reference's arithmetic for these layers is exactly torch's conv2d / BatchNorm2d): relative L2 of
    every output <= 6e-2 (bf16 storage noise through 12-14 conv+BatchNorm layers with random weights: measured 3.2e-2), state-dict names equal to the reference's."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_nhwc_scatter_gather_bit_exact(golden, dtype):
    from com_amd import ops
    g = golden("g5_dense")
    B, C, D, H, W = [int(v) for v in g["shape"]]
    # pad C (6) to 8 channels so that both element types have whole 16-byte pieces
    feat = torch.zeros((g["features"].shape[0], 8), dtype=dtype, device=DEV)
    feat[:, :C] = torch.from_numpy(g["features"]).to(DEV).to(dtype)
    idx = torch.from_numpy(g["indices"]).to(DEV)
    nchw = ops.bev_scatter(feat, idx, B, [D, H, W])
    nhwc = ops.bev_scatter(feat, idx, B, [D, H, W], channels_last=True)
    assert nhwc.shape == nchw.shape and nhwc.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(nhwc.contiguous(), nchw)
    ref = torch.from_numpy(g["spatial_features"]).to(DEV).to(dtype)        # reference HeightCompression output
    assert torch.equal(nhwc.contiguous().view(B, 8, D, H, W)[:, :C].reshape(B, C * D, H, W), ref)
    dout = torch.randn(nchw.shape, device=DEV).to(dtype)
    ga = ops.bev_gather(dout, idx, B, [D, H, W], 8)
    gb = ops.bev_gather(dout.contiguous(memory_format=torch.channels_last), idx, B, [D, H, W], 8, channels_last=True)
    assert torch.equal(ga, gb)


def test_height_compression_channels_last_module_and_autograd():
    from com_amd import hotpath, spconv
    torch.manual_seed(0)
    B, D, H, W, C = 2, 2, 24, 20, 128
    lin = torch.randperm(B * D * H * W)[:300].sort()[0]
    idx = torch.stack([lin // (D * H * W), (lin // (H * W)) % D, (lin // W) % H, lin % W], 1).int().to(DEV)
    feat = torch.randn(300, C, device=DEV).bfloat16().requires_grad_(True)
    outs = []
    for cl in (False, True):
        sp = spconv.SparseConvTensor(feat, idx, [D, H, W], B)
        m = hotpath.HeightCompression({"NUM_BEV_FEATURES": C * D, "CHANNELS_LAST": cl})
        bd = m({"encoded_spconv_tensor": sp, "encoded_spconv_tensor_stride": 8})
        sf = bd["spatial_features"]
        assert sf.is_contiguous(memory_format=torch.channels_last) == cl
        w = torch.arange(sf.numel(), device=DEV, dtype=torch.float32).reshape(sf.shape).remainder(7.0).bfloat16()
        feat.grad = None
        (sf * w).sum().backward()
        outs.append((sf.detach().contiguous(), feat.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_dense_bev_stack_bf16_channels_last_vs_torch_fp32():
    from com_amd.hotpath import dense2d
    torch.manual_seed(1)
    B, H, W = 2, 48, 40
    bb = dense2d.BaseBEVBackbone(dense2d.CENTERPOINT_BACKBONE_2D, 256).to(DEV)
    head = dense2d.CenterHeadTowers(dense2d.CENTERPOINT_HEAD, bb.num_bev_features, [['Vehicle', 'Pedestrian', 'Cyclist']]).to(DEV)
    # the reference's names (base_bev_backbone.py:28-79, center_head.py:75-99)
    names = set(dict(bb.named_parameters())) | set(dict(head.named_parameters()))
    for k in ("blocks.0.1.weight", "blocks.1.16.weight", "deblocks.1.0.weight", "deblocks.0.1.bias",
              "shared_conv.0.bias", "heads_list.0.hm.1.bias", "heads_list.0.center.0.1.weight", "heads_list.0.rot.1.weight"):
        assert k in names, k
    assert bb.num_bev_features == 512
    x = torch.randn(B, 256, H, W, device=DEV) * (torch.rand(B, 1, H, W, device=DEV) < 0.15)   # sparse BEV map
    bb.train(); head.train()

    def run(xin, dtype):
        bb.compute_dtype = head.compute_dtype = dtype
        for m in list(bb.modules()) + list(head.modules()):
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        d = head(bb({"spatial_features": xin}))
        outs = {"spatial_features_2d": d["spatial_features_2d"]}
        outs.update({"pred_" + k: v for k, v in d["pred_dicts"][0].items()})
        return outs

    ref = run(x, torch.float32)
    got = run(x.bfloat16().contiguous(memory_format=torch.channels_last), torch.bfloat16)
    assert got["spatial_features_2d"].dtype == torch.bfloat16
    assert got["spatial_features_2d"].is_contiguous(memory_format=torch.channels_last)
    assert ref["spatial_features_2d"].shape == (B, 512, H, W) and ref["pred_hm"].shape == (B, 3, H, W)
    for k in ref:
        e = float((got[k].detach().float() - ref[k].detach()).norm() / (ref[k].detach().norm() + 1e-12))
        assert e < 6e-2, (k, e)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,hw,bias", [(128, 128, (47, 52), False), (256, 128, (40, 33), False),
                                               (64, 64, (24, 24), True), (512, 64, (20, 18), True),
                                               (64, 3, (31, 17), True)])
def test_conv3x3_hip_kernel_forward_and_gradients(cin, cout, hw, bias):
    """hotpath.conv2d_fast.Conv3x3 (implicit-GEMM MFMA kernel: forward + data gradient; weight gradient through the
    sparse pair kernel over dense pair lists) against torch conv2d in fp32 on the same bf16-rounded operands: outputs
    and dx within bf16 output rounding (2^-8 of the largest value), dW / dbias within 2e-3 (fp32 accumulation)."""
    from com_amd.hotpath.conv2d_fast import Conv3x3
    torch.manual_seed(cin + cout)
    B, (H, W) = 2, hw
    m = Conv3x3(cin, cout, 3, padding=1, bias=bias).cuda()
    x = torch.randn(B, cin, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    gy = torch.randn(B, cout, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    y = m(x)
    assert y.dtype == torch.bfloat16 and y.shape == (B, cout, H, W) and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gy)
    xr = x.detach().float().requires_grad_(True)
    wr = m.weight.detach().bfloat16().float().requires_grad_(True)
    br = m.bias.detach().clone().requires_grad_(True) if bias else None
    yr = torch.nn.functional.conv2d(xr, wr, br, padding=1)
    yr.backward(gy.float())
    rel = lambda a, b: float((a.float() - b).abs().max() / b.abs().max())
    assert rel(y, yr) <= 6e-3, rel(y, yr)
    assert rel(x.grad, xr.grad) <= 6e-3, rel(x.grad, xr.grad)
    assert rel(m.weight.grad, wr.grad) <= 2e-3, rel(m.weight.grad, wr.grad)
    if bias:
        assert rel(m.bias.grad, br.grad) <= 1e-4
    # what the kernel does not cover falls back to nn.Conv2d's own path
    m2 = Conv3x3(cin, cout, 3, stride=2, padding=1, bias=False).cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert not m2._fast(x) and m2(x.detach()).shape[2] == (H + 1) // 2
        y_ac = m(x.detach().float())                       # fp32 map inside a bf16 autocast region: fast path too
    assert y_ac.dtype == torch.bfloat16 and torch.equal(y_ac, y.detach())


@pytest.mark.gpu
def test_conv3x3_packs_made_ahead_give_identical_results():
    """Conv3x3Packs (all forward + data-gradient packs in one launch) vs the per-call packs: bit-identical outputs and
    gradients; the packs are consumed once, a second forward packs by itself again."""
    from com_amd.hotpath.conv2d_fast import Conv3x3, Conv3x3Packs
    torch.manual_seed(2)
    net = torch.nn.Sequential(Conv3x3(64, 128, 3, padding=1, bias=False), Conv3x3(128, 3, 3, padding=1, bias=True)).cuda()
    x = torch.randn(2, 64, 21, 19, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)

    def run():
        xx = x.clone().requires_grad_(True)
        for p in net.parameters():
            p.grad = None
        y = net(xx)
        y.float().square().sum().backward()
        return y.detach().clone(), xx.grad.clone(), [p.grad.clone() for p in net.parameters()]

    a = run()
    plan = Conv3x3Packs(net)
    assert len(plan.convs) == 2
    plan.run()
    assert all(m._packs_ahead is not None for m in plan.convs)
    b = run()
    assert all(m._packs_ahead is None for m in plan.convs)          # consumed
    c = run()
    for u, v, w in zip((a[0], a[1], *a[2]), (b[0], b[1], *b[2]), (c[0], c[1], *c[2])):
        assert torch.equal(u, v) and torch.equal(u, w)


@pytest.mark.parametrize("cin,cout,bias", [(128, 128, False), (64, 64, True), (256, 128, False), (64, 3, True)])
def test_conv3x3_direct_deferred_gradients_equal_the_plain_path(cin, cout, bias):
    """bench.py's mode for the dense convs (spconv.functional.DIRECT_GRAD + WGRAD_JOIN_LAG): weight / bias gradients on
    the side stream, written straight into pre-allocated .grad buffers -- the weight gradient by the deferred batched
    slab reduction in the parameter's own [cout, cin, 3, 3] layout (PcdWgradReduceJob.layout = 1) -- must equal the
    gradients autograd accumulates on the plain path, bit for bit (same kernels, same summation order)."""
    from com_amd.hotpath.conv2d_fast import Conv3x3
    from com_amd.spconv import functional as Fsp
    torch.manual_seed(cin * 3 + cout)
    B, H, W = 2, 37, 29
    m = Conv3x3(cin, cout, 3, padding=1, bias=bias).cuda()
    x = torch.randn(B, cin, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, cout, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)

    def run(direct):
        xi = x.clone().requires_grad_(True)
        for p in m.parameters():
            p.grad = torch.full_like(p, 7.0) if direct else None      # direct writes OVERWRITE (the bucket is zeroed per step)
        old = (Fsp.DIRECT_GRAD, Fsp.WGRAD_JOIN_LAG)
        Fsp.DIRECT_GRAD, Fsp.WGRAD_JOIN_LAG = (True, 8) if direct else (False, 0)
        try:
            m(xi).backward(gy)
            Fsp.join_deferred_wgrad()
        finally:
            Fsp.DIRECT_GRAD, Fsp.WGRAD_JOIN_LAG = old
            Fsp.reset_deferred()
        torch.cuda.synchronize()
        return xi.grad.clone(), [p.grad.clone() for p in m.parameters()]

    dx0, g0 = run(False)
    dx1, g1 = run(True)
    assert torch.equal(dx0, dx1)
    for a, b_, p in zip(g0, g1, m.parameters()):
        # (cout = 3 runs with output channels zero-padded to 32: the deferred reduction writes only the real rows)
        assert torch.equal(a, b_), float((a - b_).abs().max())


def test_batchnorm_into_a_column_block_and_from_a_gradient_block_is_bit_identical():
    """pcd_bn_forward_ld / pcd_bn_backward_ld: y written into a column block of a wider matrix, dy read from a column
    block of a wider gradient -- bit-identical to the dense calls (same kernels, same summation order)."""
    from com_amd import ops
    torch.manual_seed(3)
    n, c, wide = 5000, 64, 192
    x = torch.randn(n, c, device=DEV).bfloat16()
    g, b = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV)
    y0, m0, s0 = ops.bn_forward(x, None, g, b, 1e-3, 0.01, True, None, None, True)
    buf = torch.full((n, wide), 7.0, device=DEV).bfloat16()
    y1, m1, s1 = ops.bn_forward(x, None, g, b, 1e-3, 0.01, True, None, None, True, out=buf[:, 64:128])
    assert y1.data_ptr() == buf[:, 64:128].data_ptr()
    assert torch.equal(buf[:, 64:128], y0) and torch.equal(m0, m1) and torch.equal(s0, s1)
    assert bool((buf[:, :64] == 7).all()) and bool((buf[:, 128:] == 7).all())
    dyw = torch.randn(n, wide, device=DEV).bfloat16()
    r0 = ops.bn_backward(dyw[:, 64:128].contiguous(), x, None, g, m0, s0, True, True, False, beta=b)
    r1 = ops.bn_backward(dyw[:, 64:128], x, None, g, m0, s0, True, True, False, beta=b)
    for u, v in zip(r0, r1):
        assert (u is None and v is None) or torch.equal(u, v)


def test_bev_backbone_without_the_cat_pass_is_bit_identical_to_torch_cat():
    from com_amd.hotpath import dense2d
    torch.manual_seed(4)
    bb = dense2d.BaseBEVBackbone(dense2d.CENTERPOINT_BACKBONE_2D, 256).to(DEV).train()
    x = (torch.randn(2, 256, 40, 36, device=DEV) * (torch.rand(2, 1, 40, 36, device=DEV) < 0.2)).bfloat16()
    x = x.contiguous(memory_format=torch.channels_last)
    gy = None
    res = []
    for share in (False, True):
        bb.SHARE_CAT = share
        for m in bb.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        for p in bb.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        y = bb({"spatial_features": xi})["spatial_features_2d"]
        assert y.shape == (2, 512, 40, 36) and y.is_contiguous(memory_format=torch.channels_last)
        if gy is None:
            gy = torch.randn_like(y)
        y.backward(gy)
        res.append([y.detach().clone(), xi.grad.clone()] + [p.grad.clone() for p in bb.parameters()])
    for u, v in zip(*res):
        assert torch.equal(u, v)


def _head_and_input(seed=7, B=2, H=26, W=22):
    from com_amd.hotpath import dense2d
    torch.manual_seed(seed)
    head = dense2d.CenterHeadTowers(dense2d.CENTERPOINT_HEAD, 512, [['Vehicle', 'Pedestrian', 'Cyclist']]).to(DEV).train()
    x = torch.randn(B, 512, H, W, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    return head, x


def test_separate_head_batched_path_equals_the_per_branch_path():
    """SeparateHead with its branches' first-stage parameters back to back (`flatten_branches_` here, bench.py's flat
    buckets in the bench): ONE conv 64 -> 320 + ONE BatchNorm(320) + the last convs on channel blocks -- predictions,
    input gradient, every parameter gradient and the BatchNorm running statistics equal the per-branch path's (same
    kernels per channel; only the accumulation of the five branch gradients into dx differs: one conv over 320 channels
    instead of five partial sums added in bf16)."""
    from com_amd.hotpath import dense2d
    from com_amd.spconv import functional as Fsp
    head, x = _head_and_input()
    sh = head.heads_list[0]
    assert sh._batchable() and sh._wide_modules() is None               # fresh module: parameters are not adjacent
    sh.flatten_branches_()
    names = list(sh.sep_head_dict)
    gys = None
    res = []
    for batched in (False, True):
        dense2d.SeparateHead.BATCHED = batched
        for m in head.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        for p in head.parameters():
            p.grad = None
        if batched:                                                      # the gradients must be adjacent too
            flat = torch.zeros(sum(p.numel() for p in head.parameters()), device=DEV)
            off = 0
            for p in dense2d.batched_param_order(head):
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        old = (Fsp.DIRECT_GRAD, Fsp.WGRAD_JOIN_LAG)
        Fsp.DIRECT_GRAD, Fsp.WGRAD_JOIN_LAG = (True, 32) if batched else (False, 0)
        try:
            xi = x.clone().requires_grad_(True)
            d = head({"spatial_features_2d": xi})["pred_dicts"][0]
            assert (sh._wide_modules() is not None) == batched
            if gys is None:
                gys = {k: torch.randn(v.shape, device=DEV).bfloat16() for k, v in d.items()}
            torch.autograd.backward([d[k] for k in names], [gys[k] for k in names])
            Fsp.join_deferred_wgrad()
        finally:
            Fsp.DIRECT_GRAD, Fsp.WGRAD_JOIN_LAG = old
            Fsp.reset_deferred()
            dense2d.SeparateHead.BATCHED = True
        torch.cuda.synchronize()
        res.append(dict(preds={k: d[k].detach().float().clone() for k in names}, dx=xi.grad.float().clone(),
                        grads={k: p.grad.clone() for k, p in head.named_parameters()},
                        stats={k: b.clone() for k, b in head.named_buffers()}))
    a, b = res
    for k in names:
        assert torch.equal(a["preds"][k], b["preds"][k]), k
    for k in a["stats"]:     # (320 channels per row: another thread <-> element mapping, i.e. another fp32 summation order)
        torch.testing.assert_close(b["stats"][k].float(), a["stats"][k].float(), rtol=2e-6, atol=1e-7)
    for k in a["grads"]:
        ga, gb = a["grads"][k], b["grads"][k]
        if k.endswith(".0.0.bias") or k == "shared_conv.0.bias":
            continue                         # a conv bias in front of a BatchNorm: the exact gradient is 0, what is left is noise
        if k.startswith("shared_conv"):      # downstream of dx of the towers (bf16 partial sums vs one fp32 accumulation)
            torch.testing.assert_close(gb, ga, rtol=3e-2, atol=3e-2 * float(ga.abs().max()))
        elif ".0.0." in k or ".0.1." in k:   # first stage: behind the BatchNorm backward (statistics differ in the last bit)
            torch.testing.assert_close(gb, ga, rtol=2e-3, atol=2e-3 * float(ga.abs().max()))
        else:                                # last convs: same activation, same kernels
            assert torch.equal(ga, gb), k
    rel = float((a["dx"] - b["dx"]).norm() / a["dx"].norm())
    assert rel < 1e-2, rel


def test_separate_head_batched_path_in_eval_mode_and_state_dict_round_trip():
    from com_amd.hotpath import dense2d
    head, x = _head_and_input(seed=11)
    ref_sd = {k: v.clone() for k, v in head.state_dict().items()}
    with torch.no_grad():
        head.train()
        head({"spatial_features_2d": x})                                 # moves the running statistics off their init
        head.eval()
        d0 = {k: v.float().clone() for k, v in head({"spatial_features_2d": x})["pred_dicts"][0].items()}
        sh = head.heads_list[0].flatten_branches_()
        assert sh._wide_modules() is not None
        d1 = head({"spatial_features_2d": x})["pred_dicts"][0]
        for k in d0:
            assert torch.equal(d0[k], d1[k].float()), k
    assert set(head.state_dict()) == set(ref_sd)                         # aliases are not registered anywhere
    head.load_state_dict(ref_sd)                                         # in-place copies go through the views
    assert head.heads_list[0]._wide_modules() is None                    # training without adjacent .grad buffers: per branch
    with torch.no_grad():
        w = head.heads_list[0]._wide_modules()[0].weight
    assert torch.equal(w[:64], head.heads_list[0].center[0][0].weight) and torch.equal(w[64:128], head.heads_list[0].center_z[0][0].weight)


@pytest.mark.parametrize("c", [320, 96, 40])
def test_fused_batchnorm_channel_counts_whose_pieces_do_not_divide_256(c):
    """5 x 64 = 320 channels (the batched head towers): the streaming passes run with (256 / pieces) * pieces threads per
    block; against torch.nn.BatchNorm1d in fp32 on the same bf16 input (outputs within bf16 rounding, statistics and
    parameter gradients 1e-4), forward and backward."""
    from com_amd.spconv import functional as Fsp
    torch.manual_seed(c)
    n = 7001
    bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(DEV).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(DEV).train()
    ref.load_state_dict(bn.state_dict())
    x = (torch.randn(n, c, device=DEV) * 2 + 0.3).bfloat16().requires_grad_(True)
    assert Fsp._fusable(bn, x)
    y = Fsp.batch_norm_act(bn, x, None, True)
    gy = torch.randn_like(y)
    y.backward(gy)
    xr = x.detach().float().requires_grad_(True)
    yr = torch.relu(ref(xr))
    yr.backward(gy.float())
    assert float((y.float() - yr).abs().max()) <= 2 ** -7 * float(yr.abs().max())
    assert float((x.grad.float() - xr.grad).abs().max()) <= 2 ** -6 * float(xr.grad.abs().max())
    torch.testing.assert_close(bn.running_mean, ref.running_mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bn.running_var, ref.running_var, rtol=1e-4, atol=1e-5)
    # (the ReLU mask is taken from the bf16-rounded output on one side and the fp32 one on the other: a few elements differ)
    torch.testing.assert_close(bn.weight.grad, ref.weight.grad, rtol=2e-2, atol=2e-2 * float(ref.weight.grad.abs().max()))
    torch.testing.assert_close(bn.bias.grad, ref.bias.grad, rtol=2e-2, atol=2e-2 * float(ref.bias.grad.abs().max()))


def test_dense_conv_epilogue_batchnorm_sums_match_the_separate_passes():
    """conv3x3 -> BatchNorm -> ReLU -> conv3x3 -> BatchNorm -> ReLU with the BatchNorm statistics taken in the convs'
    forward epilogues and the backward reductions in the second conv's data-gradient epilogue (PcdBnReduce modes 1 / 2 of
    pcd_conv2d_3x3_nhwc_bn, mid rows folded by the last workgroup) against the same stack with the separate reduction
    passes (FUSE_BN_REDUCTIONS off): same outputs, statistics within fp32 summation-order noise, gradients within the
    bf16 rounding that noise can flip."""
    from com_amd.hotpath import conv2d_fast
    from com_amd.hotpath.conv2d_fast import BatchNormReLU2d, Conv3x3
    from com_amd.spconv import functional as Fsp
    torch.manual_seed(9)
    net = torch.nn.Sequential(Conv3x3(64, 128, 3, padding=1, bias=False), BatchNormReLU2d(128, eps=1e-3, momentum=0.01, relu=True),
                              Conv3x3(128, 128, 3, padding=1, bias=True), BatchNormReLU2d(128, eps=1e-3, momentum=0.01, relu=True),
                              Conv3x3(128, 64, 3, padding=1, bias=False)).to(DEV).train()
    net[0].bn_follows = net[2].bn_follows = True
    x = torch.randn(2, 64, 45, 37, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    gy = None
    res = []
    old, old_epi = Fsp.FUSE_BN_REDUCTIONS, conv2d_fast.DENSE_BN_EPILOGUE
    try:
        for fuse in (False, True):
            Fsp.FUSE_BN_REDUCTIONS = fuse
            conv2d_fast.DENSE_BN_EPILOGUE = 3                # both directions (the default runs the forward one only)
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.reset_running_stats()
            for p in net.parameters():
                p.grad = None
            xi = x.clone().requires_grad_(True)
            y = net(xi)
            gy = torch.randn_like(y) if gy is None else gy
            y.backward(gy)
            torch.cuda.synchronize()
            res.append((y.detach().float().clone(), xi.grad.float().clone(), {k: p.grad.clone() for k, p in net.named_parameters()},
                        {k: b.clone().float() for k, b in net.named_buffers()}))
    finally:
        Fsp.FUSE_BN_REDUCTIONS, conv2d_fast.DENSE_BN_EPILOGUE = old, old_epi
    (y0, dx0, g0, s0), (y1, dx1, g1, s1) = res
    assert float((y0 - y1).abs().max()) <= 2 ** -6 * float(y0.abs().max())
    for k in s0:
        torch.testing.assert_close(s1[k], s0[k], rtol=1e-5, atol=1e-6)
    assert float((dx0 - dx1).norm() / dx0.norm()) < 5e-3
    for k in g0:
        if k == "2.bias":
            continue                                   # conv bias in front of a BatchNorm: exact gradient 0
        assert float((g0[k] - g1[k]).norm() / g0[k].norm()) < 5e-3, k
