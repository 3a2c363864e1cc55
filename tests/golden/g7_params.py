"""Deterministic parameters of the G7 end-to-end fixture (shared by make_golden.py, which runs the dense
conv3d / BatchNorm1d chain with them, and by the tests, which load them into the HIP modules).

Nothing here depends on torch's RNG: every tensor comes from numpy's PCG64 seeded with (SEED, crc32(name)),
so the generator (this container) and the GPU box see the same numbers without a 10 MB state dict in the repo.

Layer table = pcdet/models/backbones_3d/spconv_backbone.py:191-232 (VoxelResBackBone8x), state-dict names per
SURVEY.md Appendix B; weights in spconv-2.x layout [Cout, kd, kh, kw, Cin]."""
import zlib

import numpy as np

SEED = 20240707
RANGE = (-4.8, -4.8, -2.0, 4.8, 4.8, 4.0)       # grid (x, y, z) = (96, 96, 40) -> sparse_shape (41, 96, 96)
VOXEL = (0.1, 0.1, 0.15)
GRID = (96, 96, 40)
BATCH = 2
POINTS_PER_FRAME = 6000
MAX_POINTS, MAX_VOXELS = 5, 5000
BN_EPS, BN_MOMENTUM = 1e-3, 0.01                # spconv_backbone.py:187
LOSS_QUAD = 64.0                                # loss = LOSS_QUAD * 0.5 * mean(sf^2) + <sf, P>


def _basic(prefix, c, key):
    return [dict(kind="block", name=prefix, c=c, key=key)]


# (kind, name, cin, cout, kernel, stride, padding)
LAYERS = (
    [dict(kind="subm", name="conv_input", cin=5, cout=16, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1))]
    + _basic("conv1.0", 16, "res1") + _basic("conv1.1", 16, "res1")
    + [dict(kind="spconv", name="conv2.0", cin=16, cout=32, k=(3, 3, 3), s=(2, 2, 2), p=(1, 1, 1))]
    + _basic("conv2.1", 32, "res2") + _basic("conv2.2", 32, "res2")
    + [dict(kind="spconv", name="conv3.0", cin=32, cout=64, k=(3, 3, 3), s=(2, 2, 2), p=(1, 1, 1))]
    + _basic("conv3.1", 64, "res3") + _basic("conv3.2", 64, "res3")
    + [dict(kind="spconv", name="conv4.0", cin=64, cout=128, k=(3, 3, 3), s=(2, 2, 2), p=(0, 1, 1))]
    + _basic("conv4.1", 128, "res4") + _basic("conv4.2", 128, "res4")
    + [dict(kind="spconv", name="conv_out", cin=128, cout=128, k=(3, 1, 1), s=(2, 1, 1), p=(0, 0, 0))]
)
# outputs named by spconv_backbone.py:284-291: the tensor after these layers
TAPS = {"conv1.1": "x_conv1", "conv2.2": "x_conv2", "conv3.2": "x_conv3", "conv4.2": "x_conv4", "conv_out": "out"}


def param_specs():
    """[(state-dict name, shape, kind)] in module order."""
    out = []

    def bn(prefix, c):
        out.append((prefix + ".weight", (c,), "gamma"))
        out.append((prefix + ".bias", (c,), "beta"))

    for L in LAYERS:
        if L["kind"] == "block":
            c, n = L["c"], L["name"]
            for j in (1, 2):
                out.append((f"{n}.conv{j}.weight", (c, 3, 3, 3, c), "w"))
                out.append((f"{n}.conv{j}.bias", (c,), "b"))
                bn(f"{n}.bn{j}", c)
        else:
            n = L["name"]
            out.append((n + ".0.weight", (L["cout"],) + tuple(L["k"]) + (L["cin"],), "w"))
            bn(n + ".1", L["cout"])
    return out


def make_param(name, shape, kind):
    rng = np.random.default_rng([SEED, zlib.crc32(name.encode())])
    if kind == "w":
        fan_in = int(np.prod(shape[1:]))
        bound = np.sqrt(6.0 / ((1 + 5.0) * fan_in))          # kaiming_uniform_(a=sqrt(5))
        return rng.uniform(-bound, bound, shape).astype(np.float32)
    if kind == "b":
        return rng.uniform(-0.05, 0.05, shape).astype(np.float32)
    if kind == "gamma":
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if kind == "beta":
        return rng.uniform(-0.3, 0.3, shape).astype(np.float32)
    raise ValueError(kind)


def state_dict():
    return {name: make_param(name, shape, kind) for name, shape, kind in param_specs()}


def loss_projection(numel):
    """Fixed projection P (bf16-representable values): loss = sum(spatial_features * P)."""
    rng = np.random.default_rng([SEED, 99])
    p = (rng.normal(size=numel) * 1e-2).astype(np.float32)
    # round to bf16 (round-to-nearest-even on the upper 16 bits) so every consumer multiplies by the same numbers
    u = p.view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def grad_sample(g):
    """What the fixture keeps of a gradient tensor: everything up to 16384 elements, else a strided subsample."""
    flat = np.asarray(g).reshape(-1)
    stride = max(1, -(-flat.size // 16384))
    return flat[::stride]


def points(seed, n, batch):
    """Clustered cloud on the reduced grid (some points out of range / exactly on faces)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(batch):
        ctr = rng.uniform(-3.5, 3.5, (12, 3)) * np.array([1, 1, 0.3])
        which = rng.integers(0, 12, n)
        p = ctr[which] + rng.normal(0, 0.35, (n, 3)) * np.array([1, 1, 0.5])
        p[: n // 20] = rng.uniform(-6, 6, (n // 20, 3))
        p[n // 20] = [-4.8, 0.0, 0.0]
        p[n // 20 + 1] = [4.8, 0.0, 0.0]
        p[n // 20 + 2] = [0.0, 0.0, 4.0]
        p[n // 20 + 3] = [0.3, -0.7, -2.0]
        feat = np.concatenate([p, np.tanh(rng.uniform(0, 2, (n, 1))), rng.uniform(0, 1.5, (n, 1))], 1)
        out.append(feat.astype(np.float32))
    return out
