#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run ONCE, in the build container).

What pins what
--------------
The reference (ZZY816/COM, /root/reference) has no tests or golden vectors, and
its sparse-conv arithmetic lives in the un-vendored third-party package ``spconv`` which is not
installed here.  So the fixtures come from the two independent sources that ARE runnable:

* the reference's own pure-torch hot-path modules, imported file-by-file from /root/reference
  on CPU: MeanVFE, PillarVFE, PointPillarScatter, HeightCompression, DynamicMeanVFE and DynamicPillarVFE
  (the last two with a ``.cuda()`` identity patch and torch stand-ins for ``torch_scatter.scatter_mean`` /
  ``scatter_max``, because the image has neither a GPU nor torch_scatter);
* ``torch.nn.functional.conv3d`` (fp64) on the densified grid for the sparse-conv semantics
  (SubM k3; SparseConv3d k3/s2/p1, k3/s2/p(0,1,1), k(3,1,1)/s(2,1,1)/p0 -- the four geometries of
  pcdet/models/backbones_3d/spconv_backbone.py:191-232), including the active output set (from the
  occupancy mask) and dX / dW through autograd.

Nothing from /root/reference is copied: only inputs and outputs (data) are stored.  The CPU
oracle (oracle/) is then checked against these files by tests/test_oracle_golden.py.

Usage:  python tests/golden/make_golden.py [g1 .. g9]   (needs /root/reference; not needed at test time;
        with names only those fixtures are regenerated)
"""
import hashlib
import importlib
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from com_amd.utils import synth  # noqa: E402  (product-side synthetic generator, numpy only)
from oracle import oracle as O  # noqa: E402  (grid-size / collate helpers only; the voxels of G1/G4/G7 come
#                                  from voxelize_hard_np below, NOT from the oracle)
sys.path.insert(0, HERE)
import g7_params as P7  # noqa: E402


def voxelize_hard_np(points, point_cloud_range, voxel_size, max_points, max_voxels):
    """Literal numpy / Python-loop transcription of SURVEY.md A.1 (the spconv voxel generator called at
    pcdet/datasets/processor/data_processor.py:44-60), written independently of oracle/pcd_oracle.c so that
    the fixtures pin the oracle instead of replaying it: dense coor_to_voxelidx grid, points in caller order,
    float32 `floor((p - min) / vsize)`, voxel ids in first-appearance order, first `max_points` points kept,
    new voxels dropped once `max_voxels` exist."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    rng = np.asarray(point_cloud_range, dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    grid = np.round((rng[3:6] - rng[0:3]) / vs).astype(np.int64)              # data_processor.py:127-128 (x, y, z)
    coor_to_voxelidx = -np.ones((int(grid[2]), int(grid[1]), int(grid[0])), dtype=np.int32)
    C = pts.shape[1]
    voxels = np.zeros((max_voxels, max_points, C), dtype=np.float32)
    coors = np.zeros((max_voxels, 3), dtype=np.int32)
    num = np.zeros((max_voxels,), dtype=np.int32)
    voxel_num = 0
    for i in range(pts.shape[0]):
        c = np.floor((pts[i, 0:3] - rng[0:3]) / vs)                            # float32 arithmetic, true division
        if not np.all(np.isfinite(c)):
            continue
        c = c.astype(np.int64)
        if np.any(c < 0) or np.any(c >= grid):
            continue
        v = coor_to_voxelidx[c[2], c[1], c[0]]
        if v == -1:
            if voxel_num >= max_voxels:
                continue
            v = voxel_num
            voxel_num += 1
            coor_to_voxelidx[c[2], c[1], c[0]] = v
            coors[v] = (c[2], c[1], c[0])
        n = num[v]
        if n < max_points:
            voxels[v, n, :] = pts[i, :C]
            num[v] = n + 1
    return voxels[:voxel_num].copy(), coors[:voxel_num].copy(), num[:voxel_num].copy()


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def ref_module(subdir, name, pkg):
    """Import /root/reference/<subdir>/<name>.py under a fake parent package `pkg`."""
    if pkg not in sys.modules:
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, subdir)]
        sys.modules[pkg] = m
    return importlib.import_module(f"{pkg}.{name}")


class Cfg(dict):
    __getattr__ = dict.__getitem__


manifest = {}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    manifest[name] = {
        "sha256": sha(*[arrays[k] for k in sorted(arrays)]),
        "arrays": {k: [str(np.asarray(v).dtype), list(np.asarray(v).shape)] for k, v in arrays.items()},
        "bytes": os.path.getsize(path),
    }
    print(f"{name}: {manifest[name]['bytes']} bytes")


# ---------------------------------------------------------------------------------------------
# G1: config 1 (PointPillars, 4k points, B=1): hard voxels -> PillarVFE -> PointPillarScatter
def g1():
    torch.manual_seed(7)
    pts = synth.synth_cloud(0, n_beams=16, n_azimuth=250)              # 4000 points
    rng, vs = synth.PILLAR_RANGE, synth.PILLAR_VOXEL
    voxels, coords, nump = voxelize_hard_np(pts, rng, vs, synth.PILLAR_MAX_POINTS, 32000)
    coords4 = np.pad(coords, ((0, 0), (1, 0)))                          # batch idx 0
    pv = ref_module("pcdet/models/backbones_3d/vfe", "pillar_vfe", "refvfe")
    cfg = Cfg(USE_NORM=True, WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, NUM_FILTERS=[64, 64])
    vfe = pv.PillarVFE(cfg, num_point_features=5, voxel_size=list(vs), point_cloud_range=np.array(rng))
    # non-trivial BN statistics, then eval mode (fixed weights)
    with torch.no_grad():
        for m in vfe.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    vfe.eval()
    bd = {"voxels": torch.from_numpy(voxels), "voxel_num_points": torch.from_numpy(nump).float(),
          "voxel_coords": torch.from_numpy(coords4).float()}
    with torch.no_grad():
        bd = vfe(bd)
    pillar_features = bd["pillar_features"].numpy()
    sc = ref_module("pcdet/models/backbones_2d/map_to_bev", "pointpillar_scatter", "refbev")
    grid = O.grid_size(rng, vs)
    scat = sc.PointPillarScatter(Cfg(NUM_BEV_FEATURES=64), grid_size=[int(g) for g in grid])
    with torch.no_grad():
        bd = scat(bd)
    spatial = bd["spatial_features"].numpy()
    assert spatial.shape == (1, 64, 468, 468)
    state = {k.replace(".", "__"): v.numpy() for k, v in vfe.state_dict().items()}
    # the dense map is 56 MB: keep its hash, its sum and a 64x32x32 crop around the sensor
    save("g1_pillars", points=pts, voxels=voxels, coords=coords, num_points=nump,
         pillar_features=pillar_features, spatial_crop=spatial[0, :, 218:250, 218:250].copy(),
         spatial_sha=np.frombuffer(bytes.fromhex(sha(spatial)), dtype=np.uint8),
         spatial_nnz=np.array([np.count_nonzero(np.abs(spatial).sum(1))]), **{"w__" + k: v for k, v in state.items()})


# ---------------------------------------------------------------------------------------------
# reduced-grid cloud used by G2/G3/G4: range chosen so the grid is (x,y,z) = (96,96,40)
SMALL_RANGE = (-4.8, -4.8, -2.0, 4.8, 4.8, 4.0)
SMALL_VOXEL = (0.1, 0.1, 0.15)


def small_points(seed, n, batch):
    rng = np.random.default_rng(seed)
    out = []
    for b in range(batch):
        ctr = rng.uniform(-3.5, 3.5, (12, 3)) * np.array([1, 1, 0.3])
        which = rng.integers(0, 12, n)
        p = ctr[which] + rng.normal(0, 0.35, (n, 3)) * np.array([1, 1, 0.5])
        p[: n // 20] = rng.uniform(-6, 6, (n // 20, 3))                 # some out of range
        # exact-boundary probes: points exactly on min / max faces and on voxel faces
        p[n // 20] = [-4.8, 0.0, 0.0]
        p[n // 20 + 1] = [4.8, 0.0, 0.0]
        p[n // 20 + 2] = [0.0, 0.0, 4.0]
        p[n // 20 + 3] = [0.3, -0.7, -2.0]
        feat = np.concatenate([p, np.tanh(rng.uniform(0, 2, (n, 1))), rng.uniform(0, 1.5, (n, 1))], 1)
        out.append(feat.astype(np.float32))
    return out


def g2():
    frames = small_points(11, 1000, 2)
    pts_b = np.concatenate([np.pad(p, ((0, 0), (1, 0)), constant_values=float(b))
                            for b, p in enumerate(frames)], 0).astype(np.float32)
    # stubs: no GPU, no torch_scatter in this image
    ts = types.ModuleType("torch_scatter")

    def scatter_mean(src, index, dim=0):
        n = int(index.max()) + 1
        out = torch.zeros((n, src.shape[1]), dtype=src.dtype)
        out.index_add_(0, index, src)
        cnt = torch.zeros((n,), dtype=src.dtype).index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        return out / cnt.clamp(min=1).unsqueeze(1)

    ts.scatter_mean = scatter_mean
    sys.modules["torch_scatter"] = ts
    old_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        dm = ref_module("pcdet/models/backbones_3d/vfe", "dynamic_mean_vfe", "refvfe")
        grid = O.grid_size(SMALL_RANGE, SMALL_VOXEL)
        vfe = dm.DynamicMeanVFE(Cfg(), 5, list(SMALL_VOXEL), [int(g) for g in grid], list(SMALL_RANGE))
        bd = vfe({"batch_size": 2, "points": torch.from_numpy(pts_b)})
    finally:
        torch.Tensor.cuda = old_cuda
    save("g2_dynamic", points_b=pts_b, voxel_features=bd["voxel_features"].numpy(),
         voxel_coords=bd["voxel_coords"].numpy().astype(np.int32),
         range=np.array(SMALL_RANGE, np.float32), voxel_size=np.array(SMALL_VOXEL, np.float32))


# ---------------------------------------------------------------------------------------------
def densify(feat, idx, B, shape):
    d = torch.zeros((B, feat.shape[1]) + tuple(shape), dtype=torch.float64)
    d[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = feat
    return d


GEOMS = {
    "subm_k3": dict(subm=True, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1)),
    "conv_k3_s2_p1": dict(subm=False, k=(3, 3, 3), s=(2, 2, 2), p=(1, 1, 1)),
    "conv_k3_s2_p011": dict(subm=False, k=(3, 3, 3), s=(2, 2, 2), p=(0, 1, 1)),
    "conv_k311_s211_p0": dict(subm=False, k=(3, 1, 1), s=(2, 1, 1), p=(0, 0, 0)),
}


def g3():
    frames = small_points(23, 700, 2)
    B = 2
    shape = (41, 96, 96)
    per = [voxelize_hard_np(p, SMALL_RANGE, SMALL_VOXEL, 5, 5000) for p in frames]
    _, coords, _ = O.collate_voxels(per)
    idx = torch.from_numpy(coords).long()
    n = coords.shape[0]
    g = torch.Generator().manual_seed(5)
    cin, cout = 8, 16
    out = {"indices": coords, "spatial_shape": np.array(shape, np.int32)}
    for tag, rounded in (("f32", False), ("bf16in", True)):
        x = torch.randn((n, cin), generator=g, dtype=torch.float32)
        if rounded:
            x = x.bfloat16().float()
        out[f"x_{tag}"] = x.numpy()
        for name, geo in GEOMS.items():
            k = geo["k"]
            w = (torch.randn((cout, cin) + k, generator=g, dtype=torch.float32) * 0.2)
            gy_seed = torch.randn((1,), generator=g)
            if rounded:
                w = w.bfloat16().float()
            xd = densify(x.double(), idx, B, shape).requires_grad_(True)
            wd = w.double().requires_grad_(True)
            yd = F.conv3d(xd, wd, stride=geo["s"], padding=geo["p"])
            occ = densify(torch.ones((n, 1), dtype=torch.float64), idx, B, shape)
            if geo["subm"]:
                act = occ[:, 0] > 0
            else:
                act = F.conv3d(occ, torch.ones((1, 1) + k, dtype=torch.float64), stride=geo["s"],
                               padding=geo["p"])[:, 0] > 0.5
            oidx = act.nonzero()                                          # sorted by (b,z,y,x)
            y = yd[oidx[:, 0], :, oidx[:, 1], oidx[:, 2], oidx[:, 3]]
            if geo["subm"]:
                # keep the INPUT row order for SubM (out rows == in rows)
                oidx = idx
                y = yd[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]]
            gy = torch.randn(y.shape, generator=g, dtype=torch.float32).double()
            if rounded:
                gy = gy.float().bfloat16().double()
            (y * gy).sum().backward()
            dx = xd.grad[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]]
            dw = wd.grad                                                   # [cout,cin,kd,kh,kw]
            # store weights as [K, cin, cout]
            K = k[0] * k[1] * k[2]
            w_k = w.reshape(cout, cin, K).permute(2, 1, 0).contiguous().numpy()
            dw_k = dw.reshape(cout, cin, K).permute(2, 1, 0).contiguous().float().numpy()
            pre = f"{name}_{tag}_"
            out[pre + "w"] = w_k
            out[pre + "out_indices"] = oidx.numpy().astype(np.int32)
            out[pre + "out_shape"] = np.array(yd.shape[2:], np.int32)
            out[pre + "y"] = y.detach().float().numpy()
            out[pre + "gy"] = gy.float().numpy()
            out[pre + "dx"] = dx.float().numpy()
            out[pre + "dw"] = dw_k
            out[pre + "n_pairs"] = np.array([int(round(float(F.conv3d(
                occ, torch.ones((1, 1) + k, dtype=torch.float64), stride=geo["s"], padding=geo["p"])[
                    :, 0][act if not geo["subm"] else (occ[:, 0] > 0)].sum())))])
            del gy_seed
    save("g3_conv", **out)


# ---------------------------------------------------------------------------------------------
def g4():
    mv = ref_module("pcdet/models/backbones_3d/vfe", "mean_vfe", "refvfe")
    vfe = mv.MeanVFE(Cfg(), num_point_features=5)
    frames = small_points(31, 1500, 2)
    per = [voxelize_hard_np(p, SMALL_RANGE, SMALL_VOXEL, 5, 5000) for p in frames]
    voxels, coords, nump = O.collate_voxels(per)
    bd = vfe({"voxels": torch.from_numpy(voxels), "voxel_num_points": torch.from_numpy(nump).float()})
    save("g4_meanvfe", points0=frames[0], points1=frames[1], voxels=voxels, coords=coords,
         num_points=nump, voxel_features=bd["voxel_features"].numpy(),
         range=np.array(SMALL_RANGE, np.float32), voxel_size=np.array(SMALL_VOXEL, np.float32))


# ---------------------------------------------------------------------------------------------
def g5():
    hc = ref_module("pcdet/models/backbones_2d/map_to_bev", "height_compression", "refbev")
    rng = np.random.default_rng(3)
    B, C, D, H, W = 2, 6, 2, 12, 10
    lin = rng.permutation(B * D * H * W)[:57]
    idx = np.stack(np.unravel_index(lin, (B, D, H, W)), 1).astype(np.int32)
    feat = rng.normal(size=(57, C)).astype(np.float32)

    class FakeSp:  # dense() per SURVEY Appendix A.3 (spconv definition)
        def dense(self):
            out = torch.zeros((B, D, H, W, C))
            i = torch.from_numpy(idx).long()
            out[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = torch.from_numpy(feat)
            return out.permute(0, 4, 1, 2, 3).contiguous()

    m = hc.HeightCompression(Cfg(NUM_BEV_FEATURES=C * D))
    bd = m({"encoded_spconv_tensor": FakeSp(), "encoded_spconv_tensor_stride": 8})
    save("g5_dense", features=feat, indices=idx, shape=np.array([B, C, D, H, W], np.int32),
         spatial_features=bd["spatial_features"].numpy())


# ---------------------------------------------------------------------------------------------
# G6: DynamicPillarVFE (reference module imported with a .cuda() identity patch and torch stand-ins for
# torch_scatter.scatter_mean / scatter_max) on a 4000-point, 2-frame batch with the PointPillars grid
def g6():
    torch.manual_seed(13)
    frames = [synth.synth_cloud(20 + b, n_beams=16, n_azimuth=125) for b in range(2)]          # 2 x 2000 points
    pts_b = np.concatenate([np.pad(p, ((0, 0), (1, 0)), constant_values=float(b))
                            for b, p in enumerate(frames)], 0).astype(np.float32)
    rng, vs = synth.PILLAR_RANGE, synth.PILLAR_VOXEL
    ts = types.ModuleType("torch_scatter")

    def scatter_mean(src, index, dim=0):
        n = int(index.max()) + 1
        out = torch.zeros((n, src.shape[1]), dtype=src.dtype).index_add_(0, index, src)
        cnt = torch.zeros((n,), dtype=src.dtype).index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        return out / cnt.clamp(min=1).unsqueeze(1)

    def scatter_max(src, index, dim=0):
        n = int(index.max()) + 1
        out = torch.full((n, src.shape[1]), float("-inf"), dtype=src.dtype)
        out = out.scatter_reduce(0, index.unsqueeze(1).expand(-1, src.shape[1]), src, "amax", include_self=True)
        return out, None

    ts.scatter_mean, ts.scatter_max = scatter_mean, scatter_max
    sys.modules["torch_scatter"] = ts
    old_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        dp = ref_module("pcdet/models/backbones_3d/vfe", "dynamic_pillar_vfe", "refvfe")
        grid = [int(g) for g in O.grid_size(rng, vs)]
        cfg = Cfg(USE_NORM=True, WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, NUM_FILTERS=[64, 64])
        vfe = dp.DynamicPillarVFE(cfg, 5, list(vs), grid, list(rng))
        with torch.no_grad():
            for m in vfe.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.running_mean.uniform_(-0.2, 0.2)
                    m.running_var.uniform_(0.5, 1.5)
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.uniform_(-0.3, 0.3)
        vfe.eval()
        with torch.no_grad():
            bd = vfe({"batch_size": 2, "points": torch.from_numpy(pts_b)})
    finally:
        torch.Tensor.cuda = old_cuda
    state = {k.replace(".", "__"): v.numpy() for k, v in vfe.state_dict().items()}
    save("g6_dynamic_pillars", points_b=pts_b, pillar_features=bd["pillar_features"].numpy(),
         voxel_coords=bd["voxel_coords"].numpy().astype(np.int32), range=np.array(rng, np.float32),
         voxel_size=np.array(vs, np.float32), grid=np.array(grid, np.int32),
         **{"w__" + k: v for k, v in state.items()})


# ---------------------------------------------------------------------------------------------
# G7: the WHOLE VoxelResBackBone8x + HeightCompression graph (spconv_backbone.py:191-232,254-291,
# height_compression.py:20-25) forward AND backward on a reduced grid, expressed with dense
# torch.nn.functional.conv3d (fp64) masked to the active sets (as G3 does per layer), BatchNorm1d(eps 1e-3,
# training mode) over the active rows, ReLU, residual adds.  Two variants:
#   "exact": fp64 everywhere (fp32 parameters) -- the mathematically exact reference;
#   "bf16" : the same chain with every tensor the HIP path STORES in bf16 rounded to bf16 at that point
#            (input features, conv weights, every conv output, every BatchNorm(+residual)+ReLU output, and
#            the corresponding gradients in the backward pass) -- what remains between this variant and the
#            HIP path is fp32 accumulation order, so it is the one the 1e-3 north-star tolerance is checked on.
class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.float().bfloat16().double()

    @staticmethod
    def backward(ctx, g):
        return g.float().bfloat16().double()


def _g7_chain(feat_in, coords, state, emulate_bf16):
    B, shape0 = P7.BATCH, (P7.GRID[2] + 1, P7.GRID[1], P7.GRID[0])
    R = _RoundBF16.apply if emulate_bf16 else (lambda t: t)
    params = {k: torch.from_numpy(v.copy()).double().requires_grad_(True) for k, v in state.items()}
    running = {}

    def weight(name):
        w = params[name]                                          # [Cout, kd, kh, kw, Cin] (spconv 2.x)
        if emulate_bf16:
            w = _RoundBF16.apply(w)                               # bf16 weights; gradient goes to the fp32 master copy
        return w.permute(0, 4, 1, 2, 3)

    def occupancy(idx, shape):
        return densify(torch.ones((idx.shape[0], 1), dtype=torch.float64), idx, B, shape)

    def gather(dense, idx):
        return dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]]

    def scatter(rows, idx, shape):
        d = torch.zeros((B,) + tuple(shape) + (rows.shape[1],), dtype=torch.float64)
        d = d.index_put((idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]), rows)        # dense(): [B, D, H, W, C]
        return d.permute(0, 4, 1, 2, 3)

    def subm(rows, idx, shape, wname, bname=None):
        y = F.conv3d(scatter(rows, idx, shape), weight(wname), stride=1, padding=1)
        y = gather(y, idx)
        if bname is not None:
            y = y + params[bname]
        return R(y)

    def spconv(rows, idx, shape, L):
        k, st, pd = L["k"], L["s"], L["p"]
        y = F.conv3d(scatter(rows, idx, shape), weight(L["name"] + ".0.weight"), stride=st, padding=pd)
        act = F.conv3d(occupancy(idx, shape), torch.ones((1, 1) + tuple(k), dtype=torch.float64), stride=st,
                       padding=pd)[:, 0] > 0.5
        oidx = act.nonzero()                                       # sorted by (b, z, y, x): the canonical row order
        return R(gather(y, oidx)), oidx, tuple(y.shape[2:])

    def bn_act(x, prefix, residual=None):
        g, b = params[prefix + ".weight"], params[prefix + ".bias"]
        mean = x.mean(0)
        var = x.var(0, unbiased=False)
        n = x.shape[0]
        running[prefix + ".running_mean"] = (P7.BN_MOMENTUM * mean).detach().float().numpy()
        running[prefix + ".running_var"] = ((1 - P7.BN_MOMENTUM) + P7.BN_MOMENTUM * var * n / max(n - 1, 1)
                                            ).detach().float().numpy()
        y = (x - mean) / torch.sqrt(var + P7.BN_EPS) * g + b
        if residual is not None:
            y = y + residual
        return R(torch.relu(y))

    idx = torch.from_numpy(coords).long()
    shape = shape0
    x = R(torch.from_numpy(feat_in).double())
    taps, tap_idx = {}, {}
    for L in P7.LAYERS:
        n = L["name"]
        if L["kind"] == "subm":
            x = bn_act(subm(x, idx, shape, n + ".0.weight"), n + ".1")
        elif L["kind"] == "spconv":
            y, idx, shape = spconv(x, idx, shape, L)
            x = bn_act(y, n + ".1")
        else:                                                      # SparseBasicBlock, spconv_backbone.py:50-66
            ident = x
            out = bn_act(subm(x, idx, shape, n + ".conv1.weight", n + ".conv1.bias"), n + ".bn1")
            out = subm(out, idx, shape, n + ".conv2.weight", n + ".conv2.bias")
            x = bn_act(out, n + ".bn2", residual=ident)
        if n in P7.TAPS:
            taps[P7.TAPS[n]] = x
            tap_idx[P7.TAPS[n]] = (idx.numpy().astype(np.int32), shape)
    # HeightCompression: dense() [B, C, D, H, W] -> view [B, C*D, H, W]
    dense = scatter(x, idx, shape)
    sf = dense.reshape(B, dense.shape[1] * dense.shape[2], dense.shape[3], dense.shape[4])
    # loss: coherent quadratic term + a fixed random projection (bf16-representable coefficients)
    proj = torch.from_numpy(P7.loss_projection(sf.numel())).double().view_as(sf)
    loss = P7.LOSS_QUAD * 0.5 * (sf * sf).mean() + (sf * proj).sum()
    loss.backward()
    return dict(taps=taps, tap_idx=tap_idx, sf=sf, loss=loss, params=params, running=running)


def g7():
    mv = ref_module("pcdet/models/backbones_3d/vfe", "mean_vfe", "refvfe")
    frames = P7.points(P7.SEED, P7.POINTS_PER_FRAME, P7.BATCH)
    per = [voxelize_hard_np(p, P7.RANGE, P7.VOXEL, P7.MAX_POINTS, P7.MAX_VOXELS) for p in frames]
    voxels, coords, nump = O.collate_voxels(per)
    vfe = mv.MeanVFE(Cfg(), num_point_features=5)
    feat = vfe({"voxels": torch.from_numpy(voxels), "voxel_num_points": torch.from_numpy(nump).float()}
               )["voxel_features"].numpy()
    state = P7.state_dict()
    out = {"coords": coords, "num_points": nump, "voxel_features": feat,
           **{f"points{b}": f for b, f in enumerate(frames)}}
    for tag, emu in (("exact", False), ("bf16", True)):
        r = _g7_chain(feat, coords, state, emu)
        for name, t in r["taps"].items():
            out[f"{tag}_{name}"] = t.detach().float().numpy()
            if tag == "exact":
                out[f"idx_{name}"] = r["tap_idx"][name][0]
                out[f"shape_{name}"] = np.array(r["tap_idx"][name][1], np.int32)
        out[f"{tag}_spatial_features"] = r["sf"].detach().float().numpy()
        out[f"{tag}_loss"] = np.array([float(r["loss"])])
        for k, p in r["params"].items():
            g = p.grad.float().numpy()
            out[f"{tag}_grad__{k.replace('.', '__')}"] = P7.grad_sample(g).copy()
            out[f"{tag}_gnorm__{k.replace('.', '__')}"] = np.array([float(p.grad.norm())])
        for k, v in r["running"].items():
            out[f"{tag}_{k.replace('.', '__')}"] = v
        print(f"g7 {tag}: loss {float(r['loss']):.6f}, rows",
              {n: int(t.shape[0]) for n, t in r["taps"].items()})
    save("g7_backbone", **out)


# ---------------------------------------------------------------------------------------------
# G8: the reference's OWN optimizer code -- OptimWrapper(Adam(betas=(0.9, 0.99)), wd, true_wd=True, bn_wd=True) driven
# by OneCycle (tools/train_utils/optimization/{__init__.py:19-32,fastai_optim.py,learning_schedules_fastai.py}) in the
# order of tools/train_utils/train_utils.py:78-95 (scheduler.step(it); backward; clip_grad_norm_(10); step) on a small
# model with a BatchNorm layer: the parameter trajectory and the (lr, momentum) schedule, for com_amd.dist.FlatAdam /
# one_cycle.
def g8():
    from functools import partial
    fo = ref_module("tools/train_utils/optimization", "fastai_optim", "refopt")
    ls = ref_module("tools/train_utils/optimization", "learning_schedules_fastai", "refopt")
    torch.manual_seed(21)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    params = [p for p in model.parameters()]
    opt = fo.OptimWrapper.create(partial(torch.optim.Adam, betas=(0.9, 0.99)), 3e-3,
                                 [torch.nn.Sequential(*list(model.children()))], wd=0.01, true_wd=True, bn_wd=True)
    total = 20
    sched = ls.OneCycle(opt, total, 0.003, [0.95, 0.85], 10, 0.4)
    rng = np.random.default_rng(8)
    flat = lambda: torch.cat([p.detach().reshape(-1) for p in params]).numpy().copy()
    p0 = flat()
    n = p0.size
    grads, traj, lrs, moms = [], [], [], []
    for it in range(total):
        sched.step(it)
        lrs.append(float(opt.lr))
        moms.append(float(opt.mom))
        if it < 8:
            # every other step a gradient large enough for GRAD_NORM_CLIP = 10 to bind
            g = (rng.normal(size=n) * (6.0 if it % 2 else 0.05)).astype(np.float32)
            off = 0
            for p in params:
                p.grad = torch.from_numpy(g[off:off + p.numel()].copy()).view_as(p)
                off += p.numel()
            torch.nn.utils.clip_grad_norm_(params, 10.0)
            opt.step()
            grads.append(g)
            traj.append(flat())
    save("g8_adam_onecycle", p0=p0, grads=np.stack(grads), params=np.stack(traj), lr=np.array(lrs), mom=np.array(moms),
         total_steps=np.array([total]))


# ---------------------------------------------------------------------------------------------
# G9: CenterHead target assignment by the REFERENCE'S OWN code: `assign_target_of_single_head`
# (pcdet/models/dense_heads/center_head.py:104-161) is extracted from the reference file at generation time (the module
# itself cannot be imported: it pulls in the compiled iou3d_nms extension) and run with the reference's
# centernet_utils (imported with an empty `numba` stand-in: its jit-decorated helpers are not used here) on a
# 3-frame batch incl. the caller's per-head class filtering (center_head.py:193-207), for two head layouts.
def g9():
    import ast
    import textwrap
    sys.modules.setdefault("numba", types.SimpleNamespace(jit=lambda *a, **k: (lambda f: f)))
    cu = ref_module("pcdet/models/model_utils", "centernet_utils", "refmu")
    src = open(os.path.join(REF, "pcdet/models/dense_heads/center_head.py")).read()
    fn_src = None
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.FunctionDef) and node.name == "assign_target_of_single_head":
            fn_src = textwrap.dedent(ast.get_source_segment(src, node))
    ns = {"torch": torch, "centernet_utils": cu}
    exec(compile(fn_src, "center_head.assign_target_of_single_head", "exec"), ns)
    single = ns["assign_target_of_single_head"]
    rng_xy, vs = synth.WAYMO_RANGE, synth.WAYMO_VOXEL
    me = types.SimpleNamespace(point_cloud_range=list(rng_xy), voxel_size=list(vs))
    class_names = ["Vehicle", "Pedestrian", "Cyclist"]
    rng = np.random.default_rng(9)
    B, M, H, W, stride, nmax = 3, 60, 188, 188, 8, 50
    gt = np.zeros((B, M, 8), np.float32)
    for b in range(B):
        n = [40, 55, 7][b]
        gt[b, :n, 0:2] = rng.uniform(-74, 74, (n, 2))
        gt[b, :n, 2] = rng.uniform(-1, 2, n)
        cls = rng.integers(1, 4, n)
        gt[b, :n, 3] = np.where(cls == 1, rng.uniform(3.5, 12, n), rng.uniform(0.5, 2.0, n))
        gt[b, :n, 4] = np.where(cls == 1, rng.uniform(1.6, 3.0, n), rng.uniform(0.4, 1.0, n))
        gt[b, :n, 5] = rng.uniform(1.0, 3.0, n)
        gt[b, :n, 6] = rng.uniform(-np.pi, np.pi, n)
        gt[b, :n, 7] = cls
    gt[0, 3, 0:2] = [75.3, -75.3]          # outside the range: clamped centre
    gt[0, 4, 3] = 0.0                      # degenerate box: skipped, keeps its slot
    gt[1, 0, 0:2] = [75.19, 75.19]         # last cell
    out = {"gt_boxes": gt, "feature_map_size": np.array([H, W], np.int32), "stride": np.array([stride], np.int32),
           "num_max_objs": np.array([nmax], np.int32)}
    all_names = np.array(["bg", *class_names])
    for tag, heads in (("one", [class_names]), ("two", [["Vehicle"], ["Pedestrian", "Cyclist"]])):
        for hi, head in enumerate(heads):
            hm, tb, ii, mm = [], [], [], []
            for b in range(B):
                cur = torch.from_numpy(gt[b].copy())
                names = all_names[cur[:, -1].long().numpy()]
                rows = []
                for i, name in enumerate(names):                 # center_head.py:196-202
                    if name not in head:
                        continue
                    t = cur[i].clone()
                    t[-1] = head.index(name) + 1
                    rows.append(t[None, :])
                sel = torch.cat(rows, 0) if rows else cur[:0, :]
                h, r, i_, m_ = single(me, num_classes=len(head), gt_boxes=sel, feature_map_size=[W, H],
                                      feature_map_stride=stride, num_max_objs=nmax, gaussian_overlap=0.1, min_radius=2)
                hm.append(h.numpy()); tb.append(r.numpy()); ii.append(i_.numpy()); mm.append(m_.numpy())
            hm = np.stack(hm)
            nz = np.nonzero(hm)
            out[f"{tag}{hi}_heat_nz"] = np.stack(nz, 1).astype(np.int32)        # sparse form of the heat maps
            out[f"{tag}{hi}_heat_val"] = hm[nz].astype(np.float32)
            out[f"{tag}{hi}_boxes"] = np.stack(tb)
            out[f"{tag}{hi}_inds"] = np.stack(ii)
            out[f"{tag}{hi}_mask"] = np.stack(mm)
    save("g9_center_targets", **out)


def g10():
    """CenterHead losses: the reference's own `neg_loss_cornernet`, `_reg_loss`, `_gather_feat`,
    `_transpose_and_gather_feat` (pcdet/utils/loss_utils.py, extracted by name and executed as they stand) and the
    `get_loss` arithmetic of center_head.py:230-262 on small random predictions / targets, values and gradients."""
    import ast
    import textwrap
    src = open(os.path.join(REF, "pcdet/utils/loss_utils.py")).read()
    want = {"neg_loss_cornernet", "_reg_loss", "_gather_feat", "_transpose_and_gather_feat"}
    ns = {"torch": torch, "np": np}
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in want:
            exec(compile(textwrap.dedent(ast.get_source_segment(src, node)), "loss_utils." + node.name, "exec"), ns)
    rng = np.random.default_rng(10)
    B, H, W, nmax, code = 2, 24, 20, 12, 8
    order = [("center", 2), ("center_z", 1), ("dim", 3), ("rot", 2)]
    out = {"cls_weight": np.array([1.0], np.float32), "loc_weight": np.array([2.0], np.float32),
           "code_weights": np.array([1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0], np.float32),
           "head_order": np.array([n for n, _ in order])}
    total = None
    leaves = []
    for hi, nc in enumerate((1, 2)):
        hm_logit = torch.from_numpy(rng.standard_normal((B, nc, H, W)).astype(np.float32)).requires_grad_(True)
        gt = np.clip(rng.random((B, nc, H, W)).astype(np.float32) ** 6, 0, 0.98)
        n_obj = [5, 0][hi] if hi == 1 else 7                       # head 1: no positive at all in frame 1 .. see below
        inds = np.zeros((B, nmax), np.int64)
        mask = np.zeros((B, nmax), np.int64)
        for b in range(B):
            k = [7, 3][b] if hi == 0 else [5, 0][b]
            cells = rng.choice(H * W, k, replace=False)
            inds[b, :k] = cells
            mask[b, :k] = 1
            for c_ in cells:
                gt[b, rng.integers(0, nc), c_ // W, c_ % W] = 1.0
        if hi == 1:
            gt[:, :, :, :] = np.where(gt == 1.0, 0.97, gt)            # head 1: num_pos == 0 -> the reference's other branch
        boxes_t = rng.standard_normal((B, nmax, code)).astype(np.float32)
        regs = {n: torch.from_numpy(rng.standard_normal((B, c, H, W)).astype(np.float32)).requires_grad_(True)
                for n, c in order}
        pred = torch.clamp(hm_logit.sigmoid(), min=1e-4, max=1 - 1e-4)                  # center_head.py:226-228
        hm_loss, conf = ns["neg_loss_cornernet"](pred, torch.from_numpy(gt))
        pred_boxes = torch.cat([regs[n] for n, _ in order], dim=1)
        feat = ns["_transpose_and_gather_feat"](pred_boxes, torch.from_numpy(inds))     # RegLossCenterNet.forward
        rl = ns["_reg_loss"](feat, torch.from_numpy(boxes_t), torch.from_numpy(mask))
        loc_loss = (rl * torch.tensor(out["code_weights"])).sum() * 2.0
        head_loss = hm_loss * 1.0 + loc_loss
        total = head_loss if total is None else total + head_loss
        leaves.append((hi, hm_logit, regs))
        out.update({f"h{hi}_hm_logit": hm_logit.detach().numpy(), f"h{hi}_heatmap": gt, f"h{hi}_inds": inds,
                    f"h{hi}_mask": mask, f"h{hi}_target_boxes": boxes_t, f"h{hi}_hm_loss": hm_loss.detach().numpy()[None],
                    f"h{hi}_confidence": np.array([float(conf)], np.float32), f"h{hi}_reg_loss": rl.detach().numpy(),
                    f"h{hi}_loc_loss": loc_loss.detach().numpy()[None]})
        for n, _ in order:
            out[f"h{hi}_{n}"] = regs[n].detach().numpy()
    total.backward()
    out["loss"] = total.detach().numpy()[None]
    for hi, hm_logit, regs in leaves:
        out[f"h{hi}_grad_hm_logit"] = hm_logit.grad.numpy()
        for n, _ in order:
            out[f"h{hi}_grad_{n}"] = regs[n].grad.numpy()
    save("g10_center_loss", **out)


# ---------------------------------------------------------------------------------------------
# G11 / G12: the COM curriculum head by the REFERENCE'S OWN code.  `CurriculumCenterHead.cluster`,
# `.assign_targets`, `.assign_target_of_single_head`, `.sigmoid`, `.get_loss`
# (pcdet/models/dense_heads/curriculum_center_head.py:108-358,414-459) are extracted from the reference file at
# generation time as they stand and bound to a bare class (the module itself cannot be imported: it pulls in the compiled
# iou3d_nms extension), `FocalLossCenterCurriculum` + `RegLossCenterNet` + their helpers (pcdet/utils/loss_utils.py:
# 998-1390) likewise; centernet_utils is imported with an empty `numba` stand-in (its jit helpers are not used here).
def _ref_com_head():
    import ast
    import textwrap
    import torch.nn as nn
    sys.modules.setdefault("numba", types.SimpleNamespace(jit=lambda *a, **k: (lambda f: f)))
    cu = ref_module("pcdet/models/model_utils", "centernet_utils", "refmu")
    lsrc = open(os.path.join(REF, "pcdet/utils/loss_utils.py")).read()
    lns = {"torch": torch, "np": np, "nn": nn, "centernet_utils": cu}
    want = {"FocalLossCenterCurriculum", "RegLossCenterNet", "_reg_loss", "_gather_feat", "_transpose_and_gather_feat"}
    for node in ast.parse(lsrc).body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in want:
            exec(compile(textwrap.dedent(ast.get_source_segment(lsrc, node)), "loss_utils." + node.name, "exec"), lns)
    hsrc = open(os.path.join(REF, "pcdet/models/dense_heads/curriculum_center_head.py")).read()
    hns = {"torch": torch, "np": np, "centernet_utils": cu}
    methods = {}
    for node in ast.parse(hsrc).body:
        if isinstance(node, ast.ClassDef) and node.name == "CurriculumCenterHead":
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name in ("cluster", "assign_targets", "sigmoid", "get_loss",
                                                                   "assign_target_of_single_head", "group_classifier"):
                    exec(compile(textwrap.dedent(ast.get_source_segment(hsrc, fn)), "curriculum_center_head." + fn.name,
                                 "exec"), hns)
                    methods[fn.name] = hns[fn.name]
    Head = type("RefCurriculumCenterHead", (), methods)
    return Head, lns


class _AttrDict(dict):
    """EasyDict stand-in: attribute access + .get (what the reference's cfg objects offer)."""
    __getattr__ = dict.__getitem__


def _com_boxes(rng, B, M, counts, rng_lo, rng_hi):
    gt = np.zeros((B, M, 8), np.float32)
    npgt = np.zeros((B, M), np.float32)
    true_object = np.zeros((B, M), np.float32)
    occ = np.zeros((B, M), np.float32)
    facade = np.zeros((B, M), np.float32)
    for b in range(B):
        n = counts[b]
        gt[b, :n, 0] = rng.uniform(rng_lo[0], rng_hi[0], n)
        gt[b, :n, 1] = rng.uniform(rng_lo[1], rng_hi[1], n)
        gt[b, :n, 2] = rng.uniform(-1, 2, n)
        cls = rng.integers(1, 4, n)
        gt[b, :n, 3] = np.where(cls == 1, rng.uniform(3.5, 12, n), rng.uniform(0.5, 2.0, n))
        gt[b, :n, 4] = np.where(cls == 1, rng.uniform(1.6, 3.0, n), rng.uniform(0.4, 1.0, n))
        gt[b, :n, 5] = rng.uniform(1.0, 3.0, n)
        gt[b, :n, 6] = rng.uniform(-np.pi, np.pi, n)
        gt[b, :n, 7] = cls
        npgt[b, :n] = rng.integers(0, 40, n)
        true_object[b, :n] = rng.choice([1, 1, 1, 2], n)           # 1 = real object, 2 = pasted by the augmentor
        occ[b, :n] = rng.choice([0.0, 0.05, 0.0875, 0.1, 0.2, 0.21, 0.25, 0.3, 0.34, 0.41, 0.5, 0.61, 0.7, 0.81, 0.9, 1.0], n)
        facade[b, :n] = rng.integers(0, 4, n)
    return gt, npgt, true_object, occ, facade


def _sparse(prefix, arr, out, fill=0.0):
    nz = np.nonzero(arr != fill)
    out[prefix + "_nz"] = np.stack(nz, 1).astype(np.int32)
    out[prefix + "_val"] = arr[nz]


def g11():
    """COM target assignment: cluster() -> group ids, assign_targets() -> heat maps, boxes, inds, float masks,
    radius_map [B, num_max, 5] (class, cx, cy, radius, group), heatmap_mask (ones) -- on a 3-frame batch over the Waymo
    voxel grid at stride 8 (the composition BASELINE config 3 names) for the one-head layout of
    tools/cfgs/waymo_models/com/*.yaml and a two-head layout, with and without the MIN_POINTS epoch gate."""
    Head, _ = _ref_com_head()
    class_names = ["Vehicle", "Pedestrian", "Cyclist"]
    rng = np.random.default_rng(11)
    B, M, H, W, stride, nmax = 3, 60, 188, 188, 8, 50
    gt, npgt, true_object, occ, facade = _com_boxes(rng, B, M, [40, 55, 7], (-74, -74), (74, 74))
    gt[0, 3, 0:2] = [75.3, -75.3]          # outside the range: clamped centre
    gt[0, 4, 3] = 0.0                      # degenerate box: skipped, keeps its (all-zero) slot
    gt[1, 0, 0:2] = [75.19, 75.19]         # last cell
    gt[1, 1, 0:2] = [29.9, 0.0]            # distance bins 30 / 50 from both sides
    gt[1, 2, 0:2] = [30.0, 0.0]
    gt[1, 3, 0:2] = [30.000002, 0.0]
    gt[1, 4, 0:2] = [0.0, 50.0]
    gt[1, 5, 3] = 6.0                      # length bin edge
    gt[1, 5, 7] = 1
    out = {"gt_boxes": gt, "num_points_in_gt": npgt, "true_object": true_object, "occupancy_ratio": occ,
           "facade_type": facade, "feature_map_size": np.array([H, W], np.int32), "stride": np.array([stride], np.int32),
           "num_max_objs": np.array([nmax], np.int32)}
    layouts = (("one", [class_names]), ("two", [["Vehicle"], ["Pedestrian", "Cyclist"]]))
    for tag, heads in layouts:
        for gate, (epoch, thr, minp) in (("nogate", (3, 100, 0)), ("gate", (3, 100, 5)), ("late", (101, 100, 5))):
            me = Head()
            me.point_cloud_range, me.voxel_size = list(synth.WAYMO_RANGE), list(synth.WAYMO_VOXEL)
            me.class_names, me.class_names_each_head = class_names, heads
            me.epoch, me.epoch_thredhold, me.min_points = epoch, thr, minp
            me.model_cfg = _AttrDict(TARGET_ASSIGNER_CONFIG=_AttrDict(FEATURE_MAP_STRIDE=stride, NUM_MAX_OBJS=nmax,
                                                                      GAUSSIAN_OVERLAP=0.1, MIN_RADIUS=2))
            tg = torch.from_numpy(gt.copy())             # (the reference rewrites the class column in place, :260)
            group = me.cluster(gt_boxes=tg, true_object=torch.from_numpy(true_object),
                               occupancy_ratio=torch.from_numpy(occ), facade_type=torch.from_numpy(facade))
            out["group"] = group.numpy()
            td = me.assign_targets(tg, feature_map_size=torch.Size([H, W]), npgt=torch.from_numpy(npgt),
                                   true_object=group)
            for hi in range(len(heads)):
                k = f"{tag}_{gate}_h{hi}"
                _sparse(k + "_heat", td["heatmaps"][hi].numpy(), out)
                out[k + "_boxes"] = td["target_boxes"][hi].numpy()
                out[k + "_inds"] = td["inds"][hi].numpy()
                out[k + "_mask"] = td["masks"][hi].numpy()
                out[k + "_radius_map"] = td["radius_map"][hi].numpy()
                hmk = td["heatmap_mask"][hi].numpy()
                assert hmk.shape == (B, len(heads[hi]), H, W) and (hmk == 1).all()
                assert td["masks"][hi].dtype == torch.float32 and td["radius_map"][hi].dtype == torch.int64
    save("g11_com_targets", **out)


def g12():
    """COM loss: CurriculumCenterHead.get_loss = FocalLossCenterCurriculum (per-group confidence sums / counts,
    average confidence + its EMA, UCL per-object weights drawn into the heat-map mask and the box mask) +
    RegLossCenterNet, values and gradients, for: the shipped COM setting (UCL False, conf_shape (3, 96)); UCL True with a
    fixed threshold; UCL True with the EMA threshold over two consecutive steps; UCL True outside [START, END]; STRAIGHT;
    CENTER; a head without a single positive.  Targets come from the reference's assign_targets on a 40 x 36 map."""
    Head, lns = _ref_com_head()
    class_names = ["Vehicle", "Pedestrian", "Cyclist"]
    order = [("center", 2), ("center_z", 1), ("dim", 3), ("rot", 2)]
    B, M, H, W, stride, nmax = 3, 40, 36, 40, 2, 30
    pc_range, vs = [-20.0, -18.0, -2.0, 20.0, 18.0, 4.0], [0.5, 0.5, 0.15]
    rng = np.random.default_rng(12)
    out = {"point_cloud_range": np.array(pc_range, np.float32), "voxel_size": np.array(vs, np.float32),
           "feature_map_size": np.array([H, W], np.int32), "stride": np.array([stride], np.int32),
           "num_max_objs": np.array([nmax], np.int32), "head_order": np.array([n for n, _ in order]),
           "code_weights": np.ones(8, np.float32), "cls_weight": np.array([1.0], np.float32),
           "loc_weight": np.array([2.0], np.float32)}
    cases = {
        "com": dict(cur=dict(UCL=False, THRESHOLD=0.2, ELONGATION=-10, HEIGHT=1, FIX=True), epoch=5, steps=2, counts=[25, 31, 4]),
        "ucl_fix": dict(cur=dict(UCL=True, FIX=True, ELONGATION=-10, HEIGHT=1), epoch=5, steps=1, counts=[25, 31, 4]),
        "ucl_ema": dict(cur=dict(UCL=True, FIX=False, ELONGATION=-6, HEIGHT=0.8, ALPHA=0.3, ADD=1), epoch=5, steps=2, counts=[20, 12, 9]),
        "ucl_off_epoch": dict(cur=dict(UCL=True, FIX=True, START=10, END=20), epoch=5, steps=1, counts=[10, 12, 3]),
        "straight": dict(cur=dict(UCL=True, FIX=True, STRAIGHT=True, K=0.7), epoch=0, steps=1, counts=[14, 9, 11]),
        "center": dict(cur=dict(UCL=True, FIX=True, CENTER=True, RADIUS=3), epoch=1, steps=1, counts=[14, 9, 11]),
        "radius": dict(cur=dict(UCL=True, FIX=True, RADIUS=3), epoch=1, steps=1, counts=[14, 19, 11]),
        "nopos": dict(cur=dict(UCL=True, FIX=True), epoch=1, steps=1, counts=[0, 0, 0]),
    }
    out["cases"] = np.array(sorted(cases))
    for name, cs in cases.items():
        me = Head()
        me.point_cloud_range, me.voxel_size = pc_range, vs
        me.class_names, me.class_names_each_head = class_names, [class_names]
        me.epoch, me.epoch_thredhold, me.min_points = cs["epoch"], 100, 0
        me.model_cfg = _AttrDict(
            TARGET_ASSIGNER_CONFIG=_AttrDict(FEATURE_MAP_STRIDE=stride, NUM_MAX_OBJS=nmax, GAUSSIAN_OVERLAP=0.1, MIN_RADIUS=2),
            LOSS_CONFIG=_AttrDict(LOSS_WEIGHTS={"cls_weight": 1.0, "loc_weight": 2.0, "code_weights": [1.0] * 8}),
            LOSS_CURRICULUM=cs["cur"])
        me.separate_head_cfg = _AttrDict(HEAD_ORDER=[n for n, _ in order])
        me.hm_loss_func = lns["FocalLossCenterCurriculum"](me.model_cfg, conf_shape=(3, 96))
        me.reg_loss_func = lns["RegLossCenterNet"]()
        me.forward_ret_dict = {}
        out[name + "_curriculum"] = np.array(json.dumps(cs["cur"]))
        out[name + "_epoch"] = np.array([cs["epoch"]], np.int32)
        out[name + "_steps"] = np.array([cs["steps"]], np.int32)
        for st in range(cs["steps"]):
            gt, npgt, true_object, occ, facade = _com_boxes(rng, B, M, cs["counts"], (-19.5, -17.5), (19.5, 17.5))
            if name != "nopos" and st == 0:
                gt[0, 1, 0:2] = gt[0, 0, 0:2] + np.float32(0.3)           # two objects in one cell / overlapping masks
                gt[0, 1, 7] = gt[0, 0, 7]
            tg = torch.from_numpy(gt.copy())
            group = me.cluster(gt_boxes=tg, true_object=torch.from_numpy(true_object),
                               occupancy_ratio=torch.from_numpy(occ), facade_type=torch.from_numpy(facade))
            td = me.assign_targets(tg, feature_map_size=torch.Size([H, W]), npgt=torch.from_numpy(npgt), true_object=group)
            hm_logit = torch.from_numpy((rng.standard_normal((B, 3, H, W)) * 2.0 - 1.0).astype(np.float32)).requires_grad_(True)
            regs = {n: torch.from_numpy(rng.standard_normal((B, c, H, W)).astype(np.float32)).requires_grad_(True)
                    for n, c in order}
            me.forward_ret_dict["target_dicts"] = td
            me.forward_ret_dict["pred_dicts"] = [dict(hm=hm_logit, **regs)]
            masks_before = td["masks"][0].numpy().copy()
            loss, tb = me.get_loss()
            loss.backward()
            k = f"{name}_s{st}"
            out.update({k + "_gt_boxes": gt, k + "_num_points_in_gt": npgt, k + "_true_object": true_object,
                        k + "_occupancy_ratio": occ, k + "_facade_type": facade, k + "_hm_logit": hm_logit.detach().numpy(),
                        k + "_grad_hm_logit": hm_logit.grad.numpy(), k + "_loss": loss.detach().numpy()[None],
                        k + "_hm_loss": np.array([tb["hm_loss_head_0"]], np.float32),
                        k + "_loc_loss": np.array([tb["loc_loss_head_0"]], np.float32),
                        k + "_confidence": np.array([tb["confidence"]], np.float32),
                        k + "_avg_confidence_ema": np.array([me.hm_loss_func.avg_confidence], np.float64),
                        k + "_confidence_all": me.hm_loss_func.confidence_all[0].numpy(),
                        k + "_num_all": me.hm_loss_func.confidence_all[1].numpy(),
                        k + "_masks": masks_before, k + "_radius_map": td["radius_map"][0].numpy(),
                        k + "_inds": td["inds"][0].numpy(), k + "_target_boxes": td["target_boxes"][0].numpy()})
            _sparse(k + "_heat", td["heatmaps"][0].numpy(), out)
            _sparse(k + "_heatmap_mask_after", td["heatmap_mask"][0].numpy(), out, fill=1.0)   # (drawn in place by UCL)
            for n, _ in order:
                out[f"{k}_{n}"] = regs[n].detach().numpy()
                out[f"{k}_grad_{n}"] = regs[n].grad.numpy()
    save("g12_com_loss", **out)


def g14():
    """COM per-epoch group-confidence exchange (tools/train_utils/train_utils.py:57,111-112,208-216,269-287): per rank
    the epoch SUM of the per-step (3, 96) confidence sums / counts (python `sum` of a list of float32 tensors), the
    all_gather of both, the rank sum in rank order, conf / (num + 0.1).  Restated literally on synthetic per-step
    tensors of two ranks (no process group needed: all_gather just returns every rank's tensor)."""
    rng = np.random.default_rng(14)
    ranks, steps = 2, 7
    conf = rng.random((ranks, steps, 3, 96)).astype(np.float32) * rng.integers(0, 6, (ranks, steps, 3, 96)).astype(np.float32)
    num = rng.integers(0, 6, (ranks, steps, 3, 96)).astype(np.float32)
    per_rank_conf = [sum([torch.from_numpy(conf[r, s]) for s in range(steps)]) for r in range(ranks)]   # :208
    per_rank_num = [sum([torch.from_numpy(num[r, s]) for s in range(steps)]) for r in range(ranks)]
    confidence_list = sum([np.array(t) for t in per_rank_conf])                                          # :274-276
    num_list = sum([np.array(t) for t in per_rank_num])
    result = confidence_list / (num_list + 0.1)                                                          # :287
    single = np.array(per_rank_conf[0] / (per_rank_num[0] + 0.01))                                       # :325 (no dist)
    save("g14_com_epoch_gather", conf=conf, num=num, result=result.astype(np.float32), result_dtype=np.array(str(result.dtype)),
         single_rank0=single)


def g15():
    """Box decoding by the REFERENCE's own `decode_bbox_from_heatmap` (+ `_topk`, `_transpose_and_gather_feat`;
    pcdet/models/model_utils/centernet_utils.py:181-279, imported with an empty `numba` stand-in) on random head maps of a
    2-frame batch, K = 40, score threshold 0.1, the POST_CENTER_LIMIT_RANGE of the COM configs."""
    sys.modules.setdefault("numba", types.SimpleNamespace(jit=lambda *a, **k: (lambda f: f)))
    cu = ref_module("pcdet/models/model_utils", "centernet_utils", "refmu")
    rng = np.random.default_rng(15)
    B, C, H, W, K = 2, 3, 47, 47, 40
    t = lambda *sh, s=1.0: torch.from_numpy((rng.standard_normal(sh) * s).astype(np.float32))
    hm = torch.sigmoid(t(B, C, H, W, s=2.0) - 2.0)
    rot, center, center_z, dim = t(B, 2, H, W), torch.rand(B, 2, H, W), t(B, 1, H, W), torch.exp(t(B, 3, H, W, s=0.3))
    limit = torch.tensor([-80, -80, -10.0, 80, 80, 10.0]).float()
    out = cu.decode_bbox_from_heatmap(heatmap=hm, rot_cos=rot[:, 0:1], rot_sin=rot[:, 1:2], center=center, center_z=center_z,
                                      dim=dim, point_cloud_range=list(synth.WAYMO_RANGE), voxel_size=list(synth.WAYMO_VOXEL),
                                      feature_map_stride=32, K=K, circle_nms=False, score_thresh=0.1,
                                      post_center_limit_range=limit)
    arrays = dict(hm=hm.numpy(), rot=rot.numpy(), center=center.numpy(), center_z=center_z.numpy(), dim=dim.numpy(),
                  K=np.array([K], np.int32), stride=np.array([32], np.int32), limit=limit.numpy())
    for k, d in enumerate(out):
        arrays[f"boxes{k}"] = d["pred_boxes"].numpy()
        arrays[f"scores{k}"] = d["pred_scores"].numpy()
        arrays[f"labels{k}"] = d["pred_labels"].numpy()
    assert all(len(d["pred_boxes"]) > 5 for d in out)
    save("g15_decode", **arrays)


def g13():
    """Rotated BEV IoU by the REFERENCE ITSELF: pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp compiled unmodified into
    oracle/_ref/libiou3d_ref.so (oracle/ref_build/Makefile), `boxes_iou_bev_cpu` (:232-252) on 150 x 120 random boxes
    (18 000 pairs, ~16 % overlapping) + a 12 x 12 block of degenerate pairs (identical, touching, contained, crossed at
    90 degrees, zero-size, tiny angle).  Stored: the boxes and the reference's IoU matrix."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "ref_build"), "-s"])
    rng = np.random.default_rng(13)

    def boxes(n, spread):
        xy = rng.uniform(-spread, spread, (n, 2))
        z = rng.uniform(-1, 1, (n, 1))
        size = np.stack([rng.uniform(0.4, 12, n), rng.uniform(0.4, 3, n), rng.uniform(1, 3, n)], 1)
        ang = rng.uniform(-np.pi, np.pi, (n, 1))
        return np.concatenate([xy, z, size, ang], 1).astype(np.float32)
    a, b = boxes(150, 8.0), boxes(120, 8.0)
    d = np.array([[0, 0, 0, 4, 2, 1.5, 0], [0, 0, 0, 4, 2, 1.5, 0], [4, 0, 0, 4, 2, 1.5, 0], [0, 2, 0, 4, 2, 1.5, 0],
                  [0, 0, 0, 4, 2, 1.5, np.pi / 2], [0, 0, 0, 0, 0, 1, 0], [1, 1, 0, 4, 2, 1.5, 1e-3],
                  [0, 0, 0, 2, 4, 1.5, np.pi / 2], [0.5, 0.25, 0, 1, 0.5, 1, 0.3], [0, 0, 0, 4, 2, 1.5, np.pi],
                  [0, 0, 0, 4, 2, 1.5, -np.pi / 4], [100, 100, 0, 4, 2, 1.5, 0.7]], np.float32)
    save("g13_iou3d_ref", boxes_a=a, boxes_b=b, iou_ab=O.ref_boxes_iou_bev_cpu(a, b), boxes_d=d,
         iou_dd=O.ref_boxes_iou_bev_cpu(d, d))


def g16():
    """The dense BEV stack by the REFERENCE's own modules: `BaseBEVBackbone` (pcdet/models/backbones_2d/
    base_bev_backbone.py:6-112, imported as it stands) followed by the conv towers of CenterHead -- `shared_conv` built as
    center_head.py:75-83 builds it and the reference's `SeparateHead` class (center_head.py:11-46, extracted from the file:
    the module itself pulls in the compiled iou3d extension) -- at reduced widths, same structure as centerpoint.yaml
    (LAYER_NUMS [2, 2], strides [1, 2], k = stride deconvs, five two-conv branches, USE_BIAS_BEFORE_NORM).  Training-mode
    forward on a sparse-looking BEV map + backward of a fixed linear functional, float32 on the CPU.  Stored: the complete
    state dicts BEFORE the step (what `load_state_dict(strict=True)` of the drop-in modules must accept), the input, all
    outputs, the BatchNorm running statistics AFTER the step, the input gradient and four parameter gradients."""
    import ast
    import textwrap
    import torch.nn as nn
    bb_mod = ref_module("pcdet/models/backbones_2d", "base_bev_backbone", "refbev")
    src = open(os.path.join(REF, "pcdet/models/dense_heads/center_head.py")).read()
    ns = {"torch": torch, "nn": nn, "kaiming_normal_": torch.nn.init.kaiming_normal_}
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name == "SeparateHead":
            exec(compile(textwrap.dedent(ast.get_source_segment(src, node)), "center_head.SeparateHead", "exec"), ns)
    SeparateHead = ns["SeparateHead"]
    torch.manual_seed(16)
    cfg = _AttrDict(LAYER_NUMS=[2, 2], LAYER_STRIDES=[1, 2], NUM_FILTERS=[32, 64], UPSAMPLE_STRIDES=[1, 2],
                    NUM_UPSAMPLE_FILTERS=[32, 32])
    bb = bb_mod.BaseBEVBackbone(cfg, 64).train()
    shared = nn.Sequential(nn.Conv2d(bb.num_bev_features, 64, 3, stride=1, padding=1, bias=True), nn.BatchNorm2d(64),
                           nn.ReLU()).train()                                        # center_head.py:75-83
    head_dict = {"center": dict(out_channels=2, num_conv=2), "center_z": dict(out_channels=1, num_conv=2),
                 "dim": dict(out_channels=3, num_conv=2), "rot": dict(out_channels=2, num_conv=2),
                 "hm": dict(out_channels=3, num_conv=2)}
    head = SeparateHead(64, head_dict, init_bias=-2.19, use_bias=True).train()       # center_head.py:85-97
    with torch.no_grad():                    # non-trivial BatchNorm parameters (the constructors leave them at 1 / 0)
        for m in list(bb.modules()) + list(shared.modules()) + list(head.modules()):
            if isinstance(m, nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0.0, 0.2)
    arrays = {}
    for prefix, mod in (("bb.", bb), ("shared.", shared), ("head.", head)):
        for k, v in mod.state_dict().items():
            arrays["sd:" + prefix + k] = v.detach().numpy().copy()
    B, H, W = 2, 24, 20
    x = (torch.randn(B, 64, H, W) * (torch.rand(B, 1, H, W) < 0.3)).requires_grad_(True)
    d = bb({"spatial_features": x})
    f2d = d["spatial_features_2d"]
    preds = head(shared(f2d))
    ws = {k: torch.randn_like(v) for k, v in preds.items()}
    loss = sum((preds[k] * ws[k]).sum() for k in preds)
    loss.backward()
    arrays.update(x=x.detach().numpy(), spatial_features_2d=f2d.detach().numpy(), dx=x.grad.numpy())
    for k in preds:
        arrays["pred:" + k] = preds[k].detach().numpy()
        arrays["w:" + k] = ws[k].numpy()
    for prefix, mod in (("bb.", bb), ("shared.", shared), ("head.", head)):
        for k, v in mod.state_dict().items():
            if "running_" in k:
                arrays["after:" + prefix + k] = v.detach().numpy().copy()
    named = dict(bb.named_parameters())
    for k in ("blocks.0.1.weight", "blocks.1.1.weight", "deblocks.0.0.weight", "deblocks.1.0.weight"):
        arrays["grad:bb." + k] = named[k].grad.numpy()
    arrays["grad:head.hm.1.weight"] = dict(head.named_parameters())["hm.1.weight"].grad.numpy()
    arrays["grad:head.dim.0.0.weight"] = dict(head.named_parameters())["dim.0.0.weight"].grad.numpy()
    assert f2d.shape == (B, 64, H, W) and preds["hm"].shape == (B, 3, H, W)
    save("g16_dense_stack", **arrays)


def g17():
    """Host geometry of PV-RCNN's second stage by the REFERENCE's own functions, extracted from the files as they stand
    (the modules pull in compiled extensions / SharedArray): pcdet/utils/common_utils.py `rotate_points_along_z` (:35-54),
    `get_voxel_centers` (:66-82), voxel_set_abstraction.py `bilinear_interpolate_torch` (:11-44) and pvrcnn_head.py `get_global_grid_points_of_roi`
    / `get_dense_grid_points` (:111-132)."""
    import ast
    import textwrap
    csrc = open(os.path.join(REF, "pcdet/utils/common_utils.py")).read()
    cns = {"torch": torch, "np": np}
    for node in ast.parse(csrc).body:
        if isinstance(node, ast.FunctionDef) and node.name in ("rotate_points_along_z", "get_voxel_centers",
                                                               "bilinear_interpolate_torch", "check_numpy_to_torch"):
            exec(compile(textwrap.dedent(ast.get_source_segment(csrc, node)), "common_utils." + node.name, "exec"), cns)
    vsrc = open(os.path.join(REF, "pcdet/models/backbones_3d/pfe/voxel_set_abstraction.py")).read()
    for node in ast.parse(vsrc).body:
        if isinstance(node, ast.FunctionDef) and node.name == "bilinear_interpolate_torch":     # :11-44
            exec(compile(textwrap.dedent(ast.get_source_segment(vsrc, node)), "voxel_set_abstraction." + node.name, "exec"), cns)
    common_utils = types.SimpleNamespace(**{k: v for k, v in cns.items() if callable(v)})
    hsrc = open(os.path.join(REF, "pcdet/models/roi_heads/pvrcnn_head.py")).read()
    hns = {"torch": torch, "common_utils": common_utils}
    for node in ast.parse(hsrc).body:
        if isinstance(node, ast.ClassDef) and node.name == "PVRCNNHead":
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name in ("get_global_grid_points_of_roi", "get_dense_grid_points"):
                    seg = ast.get_source_segment(hsrc, fn)
                    exec(compile(textwrap.dedent(seg), "pvrcnn_head." + fn.name, "exec"), hns)
    holder = types.SimpleNamespace()
    holder.get_dense_grid_points = hns["get_dense_grid_points"]
    rng = np.random.default_rng(17)
    rois = np.zeros((2, 9, 7), np.float32)
    rois[..., 0:2] = rng.uniform(-60, 60, (2, 9, 2))
    rois[..., 2] = rng.uniform(-1, 2, (2, 9))
    rois[..., 3:6] = rng.uniform(0.5, 10, (2, 9, 3))
    rois[..., 6] = rng.uniform(-np.pi, np.pi, (2, 9))
    glob, local = hns["get_global_grid_points_of_roi"](holder, torch.from_numpy(rois), 6)
    coords = torch.from_numpy(rng.integers(0, 40, (50, 3)).astype(np.int32))
    centers = cns["get_voxel_centers"](coords, 4, list(synth.WAYMO_VOXEL), list(synth.WAYMO_RANGE))
    im = torch.from_numpy(rng.standard_normal((23, 31, 5)).astype(np.float32))
    bx = torch.from_numpy(rng.uniform(-2, 33, 64).astype(np.float32))
    by = torch.from_numpy(rng.uniform(-2, 25, 64).astype(np.float32))
    inter = cns["bilinear_interpolate_torch"](im, bx, by)
    pts = torch.from_numpy(rng.standard_normal((4, 11, 5)).astype(np.float32))
    ang = torch.from_numpy(rng.uniform(-np.pi, np.pi, 4).astype(np.float32))
    rot = cns["rotate_points_along_z"](pts, ang)
    save("g17_stage2_geometry", rois=rois, grid_global=glob.numpy(), grid_local=local.numpy(), coords=coords.numpy(),
         centers=centers.numpy(), im=im.numpy(), bx=bx.numpy(), by=by.numpy(), interp=inter.numpy(), pts=pts.numpy(),
         ang=ang.numpy(), rot=rot.numpy())


if __name__ == "__main__":
    only = set(sys.argv[1:])           # e.g. `make_golden.py g6`: regenerate just that fixture, keep the rest
    mpath = os.path.join(HERE, "MANIFEST.json")
    if only and os.path.exists(mpath):
        with open(mpath) as f:
            manifest.update(json.load(f))
    for fn in (g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11, g12, g13, g14, g15, g16, g17):
        if not only or fn.__name__ in only:
            fn()
    with open(mpath, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
