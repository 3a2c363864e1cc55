"""Column-map rulebook builds (com_amd/csrc/colmap.hip; include/pcd_ops.h "Column maps") against the CPU oracle, bit for bit.

Rows of every level in (b, y, x, z) order; the oracle's rulebooks are canonical (output rows by (b, z, y, x)), so strided
results are compared through the row permutation, SubM tables directly (the oracle's SubM build is order-agnostic).
Also: the same outputs as the flat-bitmap builds (rulebook.hip) they replace, padded capacities with the row counts in
device memory, capacity overflow, the spine path of the scans, empty / tiny / border inputs.
"""
import numpy as np
import pytest
import torch

from com_amd.utils import synth
from oracle import oracle as O
import contextlib
_plan_scope = contextlib.ExitStack()      # `with plan:` scopes opened / closed around try blocks (com_amd.ops.current_plan)

pytestmark = pytest.mark.gpu

DEV = "cuda"
CHAIN = [dict(k=(3, 3, 3), s=(2, 2, 2), p=(1, 1, 1)), dict(k=(3, 3, 3), s=(2, 2, 2), p=(1, 1, 1)),
         dict(k=(3, 3, 3), s=(2, 2, 2), p=(0, 1, 1)), dict(k=(3, 1, 1), s=(2, 1, 1), p=(0, 0, 0))]


def _ops():
    from com_amd import ops
    return ops


def _cpu(t):
    return t.detach().cpu().numpy()


def _yxz_order(idx):
    return np.lexsort((idx[:, 1], idx[:, 3], idx[:, 2], idx[:, 0]))


def _sorted_yxz(idx):
    return np.ascontiguousarray(idx[_yxz_order(idx)])


def _check_subm_cm(idx_t, batch, shape, cmap, n_dev=None, n_real=None):
    ops = _ops()
    n = idx_t.shape[0] if n_real is None else n_real
    rb_o = O.rulebook_subm(_cpu(idx_t)[:n], tuple(shape))
    rb = ops.rulebook_subm(idx_t, batch, list(shape), pad_pairs=True, rank=cmap, n_dev=n_dev)
    assert rb.order == ops.ROWS_YXZ and rb.rank is cmap
    np.testing.assert_array_equal(_cpu(rb.nbr_out)[:, :n], rb_o["nbr_out"])
    np.testing.assert_array_equal(_cpu(rb.pair_num), rb_o["pair_num"])
    np.testing.assert_array_equal(_cpu(rb.pairs)[:, :, :n], rb_o["pairs"][:, :, :n])
    return rb


def _check_conv_cm(idx_t, batch, shape, geo, cmap, n_dev=None, n_real=None, **kw):
    """strided build from the input level's column map: the oracle's canonical tables through the row permutation"""
    ops = _ops()
    n = idx_t.shape[0] if n_real is None else n_real
    idx_np = _cpu(idx_t)[:n]
    rb_o = O.rulebook_conv(idx_np, tuple(shape), geo["k"], geo["s"], geo["p"])
    rb = ops.rulebook_conv(idx_t, batch, list(shape), geo["k"], geo["s"], geo["p"], pad_pairs=True, order=ops.ROWS_YXZ,
                           in_rank=cmap, n_dev=n_dev, **kw)
    assert isinstance(rb.rank, ops.ColumnMap), "the column-map build did not run"
    perm = _yxz_order(rb_o["out_indices"])                # my row r = canonical row perm[r]
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size)
    ren = lambda a: np.where(a >= 0, inv[np.maximum(a, 0)], -1).astype(np.int32)
    m = rb_o["n_out"]
    got_m = rb.n_out if rb.n_out_dev is None else int(rb.n_out_dev.item())
    assert got_m == m and rb.out_shape == list(rb_o["out_shape"])
    np.testing.assert_array_equal(_cpu(rb.out_indices)[:m], rb_o["out_indices"][perm])
    np.testing.assert_array_equal(_cpu(rb.nbr_out)[:, :m], rb_o["nbr_out"][:, perm])
    assert bool((rb.nbr_out[:, m:] == -1).all())
    np.testing.assert_array_equal(_cpu(rb.nbr_in)[:, :n], ren(rb_o["nbr_in"]))
    np.testing.assert_array_equal(_cpu(rb.pair_num), rb_o["pair_num"])
    want_pairs = rb_o["pairs"].copy()
    want_pairs[:, 1, :] = ren(rb_o["pairs"][:, 1, :])     # pairs stay ascending in the INPUT row
    np.testing.assert_array_equal(_cpu(rb.pairs)[:, :, :n], want_pairs[:, :, :n])
    return rb, rb_o


def test_waymo_chain_from_column_maps_bit_exact():
    """Two full frames: level 1 from the voxeliser (row_order yxz -> ColumnMap), then every SubM rulebook and every strided
    build of VoxelResBackBone8x (spconv_backbone.py:199-229) from the column maps, each level's map produced by the build
    before it -- and the classes / tables equal to the flat-bitmap builds they replace."""
    from com_amd.hotpath import collate_points
    ops = _ops()
    frames = [synth.synth_cloud(f) for f in (0, 1)]
    pts, offs = collate_points(frames, DEV)
    shape = (41, 1504, 1504)
    res = ops.voxelize_hard(pts, offs, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000, feat_offset=1, num_features=5,
                            want_voxels=False, row_order="yxz", key_depth=41)
    idx_t, cmap = res["coords"], res["rank"]
    assert isinstance(cmap, ops.ColumnMap) and cmap.matches(idx_t, list(shape), [3, 3, 3])
    assert np.array_equal(_yxz_order(_cpu(idx_t)), np.arange(idx_t.shape[0]))
    expect_shapes = [[21, 752, 752], [11, 376, 376], [5, 188, 188], [2, 188, 188]]
    for geo, es in zip(CHAIN, expect_shapes):
        if shape[0] >= 3:
            _check_subm_cm(idx_t, 2, shape, cmap)
        rb, _ = _check_conv_cm(idx_t, 2, shape, geo, cmap)
        assert rb.out_shape == es
        old = ops.rulebook_conv(idx_t, 2, list(shape), geo["k"], geo["s"], geo["p"], pad_pairs=True, order=ops.ROWS_YXZ)
        assert isinstance(old.rank, ops.RankMap)
        for a, b in ((rb.out_indices, old.out_indices), (rb.nbr_in, old.nbr_in), (rb.nbr_out, old.nbr_out),
                     (rb.pairs, old.pairs), (rb.pair_num, old.pair_num), (rb.classes[0], old.classes[0]),
                     (rb.classes[1], old.classes[1])):
            assert torch.equal(a, b)
        idx_t, cmap, shape = rb.out_indices, rb.rank, tuple(rb.out_shape)


def test_waymo_chain_with_compact_tables_against_the_oracle():
    """The same two full frames with the strided rulebooks built as the training step builds them -- static plan, no pair
    lists, compact tables only (pcd_rulebook_conv_cm_build_compact) -- and everything the oracle defines checked on the
    EXPANDED tables and the DERIVED pair lists: out_indices, nbr_out, nbr_in, indice_pairs, indice_pair_num, bit for bit."""
    from com_amd.hotpath import collate_points
    ops = _ops()
    frames = [synth.synth_cloud(f) for f in (0, 1)]
    pts, offs = collate_points(frames, DEV)
    shape = (41, 1504, 1504)
    res = ops.voxelize_hard(pts, offs, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000, feat_offset=1, num_features=5,
                            want_voxels=False, row_order="yxz", key_depth=41)
    idx_t, cmap = res["coords"], res["rank"]
    n_dev, n_real = None, idx_t.shape[0]
    for lvl, geo in enumerate(CHAIN[:3]):
        m = O.rulebook_conv(_cpu(idx_t)[:n_real], tuple(shape), geo["k"], geo["s"], geo["p"])["n_out"]
        plan = ops.StaticPlan()
        plan.observe(("conv", lvl), m)                    # (capacity = 1.25 x the count, rounded: the levels below run on padded tensors)
        plan.active = True
        with plan:
            rb, _ = _check_conv_cm(idx_t, 2, shape, geo, cmap, n_dev=n_dev, n_real=n_real, pair_lists=False, compact=True,
                                   plan_key=("conv", lvl))
            plan.check()
        assert rb.nbr_out_packed is not None and rb.nbr_cls is not None and rb.implicit_pairs and rb.n_out > m
        idx_t, cmap, shape, n_dev, n_real = rb.out_indices, rb.rank, tuple(rb.out_shape), rb.n_out_dev, m


def test_golden_grid_from_column_maps(golden):
    """Fixture G3 (reduced grid; out_indices pinned by fp64 dense conv3d): level-1 map from the sorted rows, the three
    geometries, and the SubM rulebook of every output level."""
    ops = _ops()
    g = golden("g3_conv")
    idx = _sorted_yxz(g["indices"])
    shape = tuple(int(v) for v in g["spatial_shape"])
    batch = int(idx[:, 0].max()) + 1
    idx_t = torch.from_numpy(idx).to(DEV)
    cmap = ops.colmap_from_rows(idx_t, batch, list(shape))
    _check_subm_cm(idx_t, batch, shape, cmap)
    for name, geo in (("conv_k3_s2_p1", CHAIN[0]), ("conv_k3_s2_p011", CHAIN[2]), ("conv_k311_s211_p0", CHAIN[3])):
        rb, _ = _check_conv_cm(idx_t, batch, shape, geo, cmap)
        want = g[f"{name}_f32_out_indices"]
        np.testing.assert_array_equal(_cpu(rb.out_indices), want[_yxz_order(want)])
        if rb.out_shape[0] >= 3:
            _check_subm_cm(rb.out_indices, batch, rb.out_shape, rb.rank)


def test_static_capacities_device_counts_overflow_and_spine(pcd_option):
    """One full frame through the chain with padded capacities on both sides, the row counts in device memory, everything
    in ONE call per build (static plan: pcd_rulebook_conv_cm_build incl. the parity classes) -- equal to the two-phase build;
    then a capacity BELOW the real count (rows dropped, real count reported, no write out of bounds); the whole chain once
    more with the block sums forced through the spine launch."""
    from com_amd.hotpath import collate_points
    ops = _ops()
    pts, offs = collate_points([synth.synth_cloud(3)], DEV)
    res = ops.voxelize_hard(pts, offs, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000, feat_offset=1, num_features=5,
                            want_voxels=False, row_order="yxz", key_depth=41)
    for direct in (4096, 4):
        pcd_option("cm_direct_blocks", direct)
        idx_np, shape = _cpu(res["coords"]), [41, 1504, 1504]
        for lvl, geo in enumerate(CHAIN):
            n_real = idx_np.shape[0]
            cap_in = (n_real * 5 // 4 + 1023) // 1024 * 1024
            idx = torch.full((cap_in, 4), 7, dtype=torch.int32, device=DEV)     # garbage beyond the count must not be read
            idx[:n_real] = torch.from_numpy(idx_np).to(DEV)
            n_dev = torch.tensor([n_real], dtype=torch.int32, device=DEV)
            cmap = ops.colmap_from_rows(idx, 1, shape, n_dev=n_dev)
            if shape[0] >= 3:
                _check_subm_cm(idx, 1, shape, cmap, n_dev=n_dev, n_real=n_real)
            ref, rb_o = _check_conv_cm(idx, 1, shape, geo, cmap, n_dev=n_dev, n_real=n_real)
            m = ref.n_out
            plan = ops.StaticPlan()
            plan.observe(("conv", lvl), m)
            plan.active = True
            _plan_scope.close(); _plan_scope.enter_context(plan)
            try:
                rb, _ = _check_conv_cm(idx, 1, shape, geo, cmap, n_dev=n_dev, n_real=n_real, plan_key=("conv", lvl))
            finally:
                _plan_scope.close()
            assert rb.n_out > m and int(rb.n_out_dev.item()) == m
            (perm, vstart, vcap), (perm_r, vstart_r, vcap_r) = rb.classes, ref.classes
            assert vcap == vcap_r and torch.equal(vstart, vstart_r) and torch.equal(perm, perm_r)
            if ref.out_shape[0] >= 3:       # the padded output level's map serves its SubM rulebook
                a = ops.rulebook_subm(ref.out_indices, 1, ref.out_shape, rank=ref.rank, want_pairs=False)
                b = ops.rulebook_subm(rb.out_indices, 1, rb.out_shape, rank=rb.rank, n_dev=rb.n_out_dev, want_pairs=False)
                assert torch.equal(b.nbr_out[:, :m], a.nbr_out)
            if direct == 4096:
                # overflow: capacity below the real count
                plan = ops.StaticPlan(margin=1.0, round_to=1)
                plan.observe(("conv", lvl), m // 2)
                plan.active = True
                _plan_scope.close(); _plan_scope.enter_context(plan)
                try:
                    guard = torch.full((1 << 16,), 0x5A, dtype=torch.uint8, device=DEV)
                    small = ops.rulebook_conv(idx, 1, shape, geo["k"], geo["s"], geo["p"], n_dev=n_dev, order=ops.ROWS_YXZ,
                                              in_rank=cmap, plan_key=("conv", lvl))
                    guard2 = torch.full((1 << 16,), 0x5A, dtype=torch.uint8, device=DEV)
                finally:
                    _plan_scope.close()
                half = small.n_out
                assert m // 2 <= half < m and int(small.n_out_dev.item()) == m
                assert torch.equal(small.out_indices, ref.out_indices[:half])
                assert torch.equal(small.nbr_out, ref.nbr_out[:, :half])
                want_in = torch.where(ref.nbr_in >= half, torch.full_like(ref.nbr_in, -1), ref.nbr_in)
                assert torch.equal(small.nbr_in[:, :n_real], want_in[:, :n_real])
                assert bool((guard == 0x5A).all()) and bool((guard2 == 0x5A).all())
            idx_np, shape = _cpu(ref.out_indices), ref.out_shape


def test_tiny_border_and_dense_inputs():
    """One voxel; voxels in every corner of the grid (neighbours outside on all sides, stride parity at the borders); a
    fully occupied block (every column 62 deep would not fit: D = 20 here, all z set)."""
    ops = _ops()
    D, H, W = 20, 33, 47
    cases = [np.array([[0, 5, 7, 9]], np.int32)]
    corners = [(b, z, y, x) for b in (0, 1) for z in (0, D - 1) for y in (0, H - 1) for x in (0, W - 1)]
    cases.append(np.array(corners, np.int32))
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(10, 16), np.arange(20, 31), indexing="ij")
    dense = np.stack([np.ones(zz.size, np.int64), zz.ravel(), yy.ravel(), xx.ravel()], 1).astype(np.int32)
    cases.append(dense)
    rng = np.random.default_rng(7)
    rnd = np.unique(np.stack([rng.integers(0, 3, 4000), rng.integers(0, D, 4000), rng.integers(0, H, 4000),
                              rng.integers(0, W, 4000)], 1).astype(np.int32), axis=0)
    cases.append(rnd)
    for idx in cases:
        idx = _sorted_yxz(idx)
        batch = int(idx[:, 0].max()) + 1
        idx_t = torch.from_numpy(idx).to(DEV)
        cmap = ops.colmap_from_rows(idx_t, batch, [D, H, W])
        _check_subm_cm(idx_t, batch, (D, H, W), cmap)
        for geo in (CHAIN[0], CHAIN[2], CHAIN[3], dict(k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1)),
                    dict(k=(3, 3, 3), s=(2, 1, 2), p=(1, 0, 1))):
            rb, _ = _check_conv_cm(idx_t, batch, (D, H, W), geo, cmap)
            if rb.out_shape[0] >= 3 and rb.n_out > 0:
                _check_subm_cm(rb.out_indices, batch, rb.out_shape, rb.rank)


def test_random_grids_and_geometries():
    """Forty random cases: grid sizes whose width is / is not a multiple of 32 (the maps pad the BEV row pitch), 1-3 frames,
    occupancies from 0.1 % to 60 %, every kernel / stride / padding combination the column-map builds cover -- each strided
    build and the SubM rulebook of its output level against the oracle, bit for bit."""
    ops = _ops()
    rng = np.random.default_rng(2025)
    ran = 0
    for case in range(40):
        D, H, W = int(rng.integers(3, 44)), int(rng.integers(5, 90)), int(rng.choice([31, 32, 33, 47, 64, 65, 96, 100, 127]))
        batch = int(rng.integers(1, 4))
        dens = float(rng.choice([0.001, 0.01, 0.05, 0.2, 0.6]))
        n = max(1, int(batch * D * H * W * dens))
        idx = np.unique(np.stack([rng.integers(0, batch, n), rng.integers(0, D, n), rng.integers(0, H, n), rng.integers(0, W, n)], 1)
                        .astype(np.int32), axis=0)
        idx = _sorted_yxz(idx)
        idx_t = torch.from_numpy(idx).to(DEV)
        cmap = ops.colmap_from_rows(idx_t, batch, [D, H, W])
        _check_subm_cm(idx_t, batch, (D, H, W), cmap)
        if rng.random() < 0.5:
            k, s, p = (3, 3, 3), tuple(int(v) for v in rng.integers(1, 3, 3)), tuple(int(v) for v in rng.integers(0, 2, 3))
        else:
            k, s, p = (3, 1, 1), (int(rng.integers(1, 3)), 1, 1), (int(rng.integers(0, 2)), 0, 0)
        out = O.conv_out_shape((D, H, W), k, s, p, (1, 1, 1))
        if min(out) <= 0:
            continue
        geo = dict(k=k, s=s, p=p)
        rb, rb_o = _check_conv_cm(idx_t, batch, (D, H, W), geo, cmap)
        ran += 1
        if rb.out_shape[0] >= 3 and rb.n_out > 0:
            _check_subm_cm(rb.out_indices, batch, rb.out_shape, rb.rank)
    assert ran >= 30


def test_unsupported_geometries_fall_back_to_the_flat_builds():
    """Dilation, 5-wide kernels and rows that are not z-fastest are outside the column-map builds: the same call still
    returns the oracle's rulebook (through the flat-bitmap build)."""
    ops = _ops()
    rng = np.random.default_rng(11)
    D, H, W = 12, 40, 40
    idx = _sorted_yxz(np.unique(np.stack([np.zeros(3000, np.int64), rng.integers(0, D, 3000), rng.integers(0, H, 3000),
                                          rng.integers(0, W, 3000)], 1).astype(np.int32), axis=0))
    idx_t = torch.from_numpy(idx).to(DEV)
    cmap = ops.colmap_from_rows(idx_t, 1, [D, H, W])
    for k, s, p, d in (((5, 5, 5), (2, 2, 2), (2, 2, 2), 1), ((3, 3, 3), (2, 2, 2), (2, 2, 2), 2), ((3, 3, 3), (3, 3, 3), (1, 1, 1), 1)):
        rb = ops.rulebook_conv(idx_t, 1, [D, H, W], k, s, p, d, pad_pairs=True, order=ops.ROWS_YXZ, in_rank=cmap)
        assert isinstance(rb.rank, ops.RankMap)
        rb_o = O.rulebook_conv(idx, (D, H, W), k, s, p, d)
        perm = _yxz_order(rb_o["out_indices"])
        np.testing.assert_array_equal(_cpu(rb.out_indices), rb_o["out_indices"][perm])
        np.testing.assert_array_equal(_cpu(rb.nbr_out), rb_o["nbr_out"][:, perm])


@pytest.mark.parametrize("direction", [0, 1])
def test_pair_driven_strided_conv_against_the_oracle_and_the_gather_kernels(direction):
    """pcd_sparse_conv_pairs (pconv_kernel) on the level-1 -> 2 conv of a 2-frame Waymo-shaped batch, rows z-fastest: the
    forward (16 -> 32, + bias) and the data gradient (32 -> 16, + addend) from the rulebook's indice pairs -- first that the
    pairs of every offset ARE sorted by both rows (what the segment search relies on), then against the oracle's conv on the
    same bf16 operands (one bf16 ulp per element; the fp32-output form at 1e-3 per element), against the 27-slot gather
    kernels, the BatchNorm sums of the epilogue, and with padded capacities + device-side row counts."""
    from com_amd import _lib as L
    if not L.has_experiments():
        pytest.skip("pconv_kernel is an EXPERIMENTS-build kernel (make -C com_amd/csrc EXPERIMENTS=1)")
    from com_amd.hotpath import collate_points
    ops = _ops()
    pts, offs = collate_points([synth.synth_cloud(f, 32, 2500) for f in (0, 1)], DEV)
    res = ops.voxelize_hard(pts, offs, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000, feat_offset=1, num_features=5,
                            want_voxels=False, row_order="yxz", key_depth=41)
    idx, shape = res["coords"], [41, 1504, 1504]
    geo = CHAIN[0]
    rb = ops.rulebook_conv(idx, 2, shape, geo["k"], geo["s"], geo["p"], order=ops.ROWS_YXZ, in_rank=res["rank"])
    pn = _cpu(rb.pair_num)
    pairs = _cpu(rb.pairs)
    for k in range(27):
        assert (np.diff(pairs[k, 0, :pn[k]]) > 0).all() and (np.diff(pairs[k, 1, :pn[k]]) > 0).all(), k
    rb_o = O.rulebook_conv(_cpu(idx), tuple(shape), geo["k"], geo["s"], geo["p"])
    perm = _yxz_order(rb_o["out_indices"])
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size)
    n_in, n_out = idx.shape[0], rb.n_out
    g = torch.Generator().manual_seed(123 + direction)
    w = torch.randn(32, 3, 3, 3, 16, generator=g) * (1.0 / np.sqrt(27 * 16))
    wk = O.bf16_round(O.weight_from_spconv2(w.numpy()))                                        # [K, 16, 32]
    bf = lambda a: torch.from_numpy(a).to(torch.bfloat16)
    if direction == 0:
        x = bf(torch.randn(n_in, 16, generator=g).numpy())
        bias = torch.randn(32, generator=g) * 0.1
        ref = O.conv_fwd(x.float().numpy(), wk, bias.numpy(), rb_o, threads=8)[perm]           # rows in my order
        packed = ops.pack_weight(w.to(DEV), 0)
        run = lambda dt, red=None: ops.pair_conv(x.to(DEV), packed, bias.to(DEV), rb, 0, 32, dt, bn_reduce=red)
        gen = ops.gather_gemm(x.to(DEV), packed, bias.to(DEV), rb.nbr_out, 27, False, n_out, 32, torch.bfloat16)
    else:
        dy = bf(torch.randn(n_out, 32, generator=g).numpy())
        add = bf(torch.randn(n_in, 16, generator=g).numpy())
        dy_canon = np.zeros((n_out, 32), np.float32)
        dy_canon[perm] = dy.float().numpy()                                                    # my row r = canonical row perm[r]
        dxo, _, _ = O.conv_bwd(np.zeros((n_in, 16), np.float32), wk, dy_canon, rb_o, threads=8)
        ref = dxo + add.float().numpy()
        packed = ops.pack_weight(w.to(DEV), 1)
        run = lambda dt, red=None: ops.pair_conv(dy.to(DEV), packed, None, rb, 1, 16, dt,
                                                 addend=add.to(DEV) if dt == torch.bfloat16 else add.to(DEV).float(), bn_reduce=red)
        gen = ops.gather_gemm(dy.to(DEV), packed, None, rb.nbr_in, 27, False, n_in, 16, torch.bfloat16, addend=add.to(DEV))
    y = run(torch.bfloat16)
    rms = float(np.sqrt((ref.astype(np.float64) ** 2).mean()))
    err = np.abs(y.float().cpu().numpy() - ref)
    assert (err <= 2.0 ** -7 * np.abs(ref) + 2.0 ** -7 * rms).all(), float(err.max())
    y32 = run(torch.float32).cpu().numpy()
    assert (np.abs(y32 - ref) <= 1e-3 * (np.abs(ref) + 0.1 * rms)).all()
    d = (gen.float() - y.float()).abs()
    assert float((d > 0).float().mean()) < 2e-3 and float(d.max()) <= 2.0 ** -6 * float(gen.float().abs().max())
    if direction == 0:
        st = ops.BnReduce(1)
        yb = run(torch.bfloat16, st)
        torch.cuda.synchronize()
        assert torch.equal(yb, y)
        got = st.partial.double().sum(0)
        want = torch.stack([y.double().sum(0), (y.double() ** 2).sum(0)])
        mag = torch.stack([y.double().abs().sum(0), (y.double() ** 2).sum(0)]).clamp_min(1.0)
        assert float(((got - want).abs() / mag).max()) < 2e-6


def test_columns_without_rows_are_counted_against_the_map_capacity():
    """pad_z 0, k 3, s 2 on an EVEN depth: input plane D - 1 feeds no output z, so an input column that only holds that plane
    gives its output BEV cell a column WITHOUT rows -- more columns than rows are possible while the map's column capacity is sized
    by rows (round-5 advisor finding).  The build reports its column count; the host layer refuses a map that lost columns."""
    ops = _ops()
    D, H, W = 4, 24, 40
    # every second BEV cell holds ONLY z = D - 1 (no output rows), a few cells hold z = 0 (one output row each)
    rows = [(0, D - 1, y, x) for y in range(0, H, 2) for x in range(0, W, 2)] + [(0, 0, 1, 1), (0, 0, 5, 9)]
    idx = _sorted_yxz(np.array(rows, np.int32))
    idx_t = torch.from_numpy(idx).to(DEV)
    cmap = ops.colmap_from_rows(idx_t, 1, [D, H, W])
    with pytest.raises(ops.L.PcdError, match="columns > capacity"):
        ops.rulebook_conv(idx_t, 1, [D, H, W], (3, 3, 3), (2, 2, 2), (0, 1, 1), order=ops.ROWS_YXZ, in_rank=cmap)
    # the same rows through the flat build: the oracle's rulebook
    rb = ops.rulebook_conv(idx_t, 1, [D, H, W], (3, 3, 3), (2, 2, 2), (0, 1, 1), pad_pairs=True)
    rb_o = O.rulebook_conv(idx, (D, H, W), (3, 3, 3), (2, 2, 2), (0, 1, 1))
    np.testing.assert_array_equal(_cpu(rb.out_indices), rb_o["out_indices"])
