"""CPU: the C-ABI library loads and exports every symbol include/pcd_ops.h declares (no compute
calls: there is no GPU here), plus host-side logic of the spconv mirror."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pcd_ops.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from com_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in pcd_ops.h but not exported"
    assert sorted(_lib.PROTOTYPES) == declared, "ctypes prototypes out of sync with the header"
    lib = _lib.lib()
    assert lib.pcd_version() >= 100
    assert lib.pcd_build_arch() == b"gfx950"
    assert lib.pcd_error_string(-3).decode().startswith("batch")


def test_experiment_kernels_stay_out_of_the_default_library():
    """include/pcd_ops_experiments.h declares the entry points of the measured-slower kernels; the DEFAULT build must not export
    them (they are compiled only by `make EXPERIMENTS=1` into com_amd/lib_experiments/), the ctypes layer binds them when present."""
    from com_amd import _lib
    text = open(os.path.join(ROOT, "include", "pcd_ops_experiments.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(pcd_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(_lib.EXPERIMENT_PROTOTYPES) and len(declared) == 5
    default = ctypes.CDLL(os.path.join(ROOT, "com_amd", "lib", "libpcdops_hip.so"))
    for name in declared:
        assert not hasattr(default, name), f"{name} is an experiment: it must not be in the default library"
    default.pcd_subm_window_tile_rows.restype = ctypes.c_int
    assert default.pcd_subm_window_tile_rows(128, 128) == 0 and default.pcd_subm_window_tile_rows(64, 64) > 0
    # (the optional build, when it is there and not older than the default one)
    if os.path.exists(_lib.EXPERIMENTS_LIB_PATH) and os.path.getmtime(_lib.EXPERIMENTS_LIB_PATH) >= os.path.getmtime(_lib.LIB_PATH):
        exp = ctypes.CDLL(_lib.EXPERIMENTS_LIB_PATH)
        for name in declared + _declared_symbols():
            assert hasattr(exp, name), name


def test_header_structs_match_ctypes_mirrors(tmp_path):
    """The by-value / by-pointer structs of the C ABI (PcdBnReduce, PcdColsumJob, PcdWgradReduceJob) as gcc lays
    them out from include/pcd_ops.h == the ctypes.Structure mirrors in com_amd/_lib.py (size and field offsets)."""
    import subprocess
    from com_amd import _lib
    structs = {"PcdBnReduce": _lib.PcdBnReduce, "PcdColsumJob": _lib.PcdColsumJob,
               "PcdWgradReduceJob": _lib.PcdWgradReduceJob, "PcdComCurriculum": _lib.PcdComCurriculum}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "pcd_ops.h"', 'int main(void) {']
    for name, st in structs.items():
        lines.append(f'printf("{name} size %zu\\n", sizeof({name}));')
        for field, _ in st._fields_:
            lines.append(f'printf("{name} {field} %zu\\n", offsetof({name}, {field}));')
    lines += ['return 0; }']
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = {}
    for line in subprocess.check_output([str(exe)]).decode().split("\n"):
        if line:
            a, b, c = line.split()
            got[(a, b)] = int(c)
    for name, st in structs.items():
        assert got[(name, "size")] == ctypes.sizeof(st), name
        for field, _ in st._fields_:
            assert got[(name, field)] == getattr(st, field).offset, (name, field)
    assert _lib.COLSUM_MAX_JOBS == 32 and _lib.WGRAD_MAX_JOBS == 32      # PCD_COLSUM_MAX_JOBS / PCD_WGRAD_MAX_JOBS


def test_host_only_entry_points_agree_with_oracle():
    """pcd_conv_out_shape and the workspace-size queries are pure host code."""
    from com_amd import ops, _lib
    from oracle import oracle as O
    cases = [((41, 1504, 1504), 3, 2, 1), ((21, 752, 752), 3, 2, 1), ((11, 376, 376), 3, 2, (0, 1, 1)),
             ((5, 188, 188), (3, 1, 1), (2, 1, 1), 0), ((4, 4, 4), 3, 1, 0), ((2, 2, 2), 3, 2, 0)]
    for shp, k, s, p in cases:
        assert ops.conv_out_shape(shp, k, s, p, 1) == list(O.conv_out_shape(shp, k, s, p, 1))
    # spconv_backbone.py:89-112 shape comments
    assert ops.conv_out_shape((41, 1504, 1504), 3, 2, 1, 1) == [21, 752, 752]
    assert ops.conv_out_shape((5, 188, 188), (3, 1, 1), (2, 1, 1), 0, 1) == [2, 188, 188]
    lib = _lib.lib()
    assert lib.pcd_voxelize_hard_workspace_bytes(160000, 5, 1) > 160000 * 8
    assert lib.pcd_packed_weight_bytes(27, 16, 16, 0) == 14 * 1 * 64 * 8 * 2      # ceil(27*16/32) steps
    assert lib.pcd_packed_weight_bytes(27, 5, 16, 0) == 7 * 1 * 64 * 8 * 2        # 5 -> 8 channels
    # 128 x 128: header + (512 equal-pair chunks + one tile per offset) tiles of 128 x 128 floats; else row-range slabs:
    # 64 x 64 -> 6144-row splits (7 for 40000 rows)
    assert lib.pcd_sparse_conv_wgrad_workspace_bytes(27, 128, 128, 40000) == 4096 + (512 + 27) * 128 * 128 * 4
    assert lib.pcd_sparse_conv_wgrad_workspace_bytes(27, 64, 64, 40000) == 7 * 64 * 27 * 64 * 4
    assert ops.grid_size((-75.2, -75.2, -2, 75.2, 75.2, 4), (0.1, 0.1, 0.15)) == [1504, 1504, 40]


def test_ops_refuse_cpu_tensors_loudly():
    from com_amd import ops, _lib
    with pytest.raises(_lib.PcdError):
        ops.rulebook_subm(torch.zeros((4, 4), dtype=torch.int32), 1, [4, 4, 4])
    with pytest.raises(_lib.PcdError):
        ops.voxelize_hard(torch.zeros((4, 5)), [0, 4], (0, 0, 0, 1, 1, 1), (0.5, 0.5, 0.5), 5, 10)


def test_backbone_state_dict_matches_reference_layout():
    """parameter names / shapes of VoxelResBackBone8x and VoxelBackBone8x."""
    from com_amd.hotpath import VoxelBackBone8x, VoxelResBackBone8x
    m = VoxelResBackBone8x({}, 5, [1504, 1504, 40])
    assert m.sparse_shape == [41, 1504, 1504]
    sd = m.state_dict()
    assert sd["conv_input.0.weight"].shape == (16, 3, 3, 3, 5) and "conv_input.0.bias" not in sd
    assert sd["conv1.0.conv1.weight"].shape == (16, 3, 3, 3, 16) and sd["conv1.1.conv2.bias"].shape == (16,)
    assert sd["conv2.0.0.weight"].shape == (32, 3, 3, 3, 16) and "conv2.0.0.bias" not in sd
    assert sd["conv4.0.0.weight"].shape == (128, 3, 3, 3, 64)
    assert sd["conv4.2.conv2.weight"].shape == (128, 3, 3, 3, 128)
    assert sd["conv_out.0.weight"].shape == (128, 3, 1, 1, 128)
    assert sd["conv_out.1.running_mean"].shape == (128,)
    n_sparse = sum(p.numel() for p in m.parameters())
    assert 2.6e6 < n_sparse < 2.8e6                         # ~2.69 M sparse-3D params
    assert m.conv4[0][0].padding == [0, 1, 1] and m.conv_out[0].stride == [2, 1, 1]
    assert m.conv1[0].conv1.indice_key == "res1" and m.conv_input[0].indice_key == "subm1"
    p = VoxelBackBone8x({}, 5, [1504, 1504, 40])
    sp = p.state_dict()
    assert sp["conv4.0.0.weight"].shape == (64, 3, 3, 3, 64) and sp["conv_out.0.weight"].shape == (128, 3, 1, 1, 64)
    assert p.conv1[0][0].indice_key == "subm1"              # shares the rulebook with conv_input
    assert all(".bias" not in k or ".1." in k or k.endswith("1.bias") for k in sp if "conv" in k and "weight" not in k)


def test_sparse_sequential_and_tensor_semantics():
    from com_amd import spconv
    seq = spconv.SparseSequential(torch.nn.BatchNorm1d(4), torch.nn.ReLU())
    assert list(seq._modules) == ["0", "1"] and len(seq) == 2
    feats = torch.randn(6, 4)
    idx = torch.zeros((6, 4), dtype=torch.int64)
    t = spconv.SparseConvTensor(feats, idx, [3, 3, 3], 1)
    assert t.indices.dtype == torch.int32
    out = seq(t)                                            # plain nn.Modules see .features
    assert isinstance(out, spconv.SparseConvTensor) and out.indice_dict is t.indice_dict
    t2 = t.replace_feature(feats * 2)
    assert t2.indices is t.indices and torch.equal(t2.features, feats * 2)
    t.features = feats + 1                                  # spconv-1.x spelling still works
    assert torch.equal(t.features, feats + 1)
    assert isinstance(spconv.SubMConv3d(4, 8, 3), spconv.conv.SparseConvolution)   # spconv_utils.py:19
    empty = spconv.SparseConvTensor(torch.zeros((0, 4)), torch.zeros((0, 4), dtype=torch.int32), [3, 3, 3], 1)
    assert seq(empty).features.shape == (0, 4)              # empty tensors skip dense modules


def test_synthetic_cloud_spec():
    from com_amd.utils import synth
    p = synth.synth_cloud(0)
    assert p.shape == (160000, 5) and p.dtype == np.float32
    assert np.array_equal(p, synth.synth_cloud(0)) and not np.array_equal(p, synth.synth_cloud(1))
    r = synth.WAYMO_RANGE
    inside = ((p[:, 0] >= r[0]) & (p[:, 0] < r[3]) & (p[:, 1] >= r[1]) & (p[:, 1] < r[4]) &
              (p[:, 2] >= r[2]) & (p[:, 2] < r[5]))
    assert 0.99 < inside.mean() < 1.0                       # outliers exercise the drop path
    frames, cat = synth.synth_batch(0, 2, 16, 250)
    assert cat.shape == (8000, 6) and set(np.unique(cat[:, 0])) == {0.0, 1.0}


def test_pillar_vfe_host_module_loads_reference_state_dict_and_has_no_cpu_fallback(golden):
    """BASELINE config 1 plumbing: the PillarVFE mirror takes the reference module's state dict (same parameter /
    buffer names, fixture G1) -- and, like every hot-path op, refuses CPU tensors instead of silently computing in
    torch (the numerics are checked on the GPU: tests/test_gpu_parity.py::test_pillar_vfe_*)."""
    import pytest
    from com_amd import _lib
    from com_amd.hotpath import PillarVFE
    from com_amd.utils import synth
    g = golden("g1_pillars")
    cfg = dict(USE_NORM=True, WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, NUM_FILTERS=[64, 64])
    vfe = PillarVFE(cfg, 5, list(synth.PILLAR_VOXEL), np.array(synth.PILLAR_RANGE))
    sd = {k[3:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("w__")}
    vfe.load_state_dict(sd)                                  # same parameter / buffer names as the reference
    vfe.eval()
    coords4 = torch.from_numpy(np.pad(g["coords"], ((0, 0), (1, 0)))).float()
    with pytest.raises(_lib.PcdError, match="no CPU fallback"), torch.no_grad():
        vfe({"voxels": torch.from_numpy(g["voxels"]),
             "voxel_num_points": torch.from_numpy(g["num_points"]).float(), "voxel_coords": coords4})
    assert vfe.get_output_feature_dim() == 64


def test_host_voxeliser_matches_the_oracle_bit_for_bit(golden):
    """pcd_voxelize_hard_host (the variant for forked DataLoader workers, pcdet/datasets/processor/data_processor.py:44-60,
    130-141): product code through the C ABI, no GPU -- against the oracle on the golden frames (fixture G4, whose voxels the
    reference's MeanVFE consumed) and on a Waymo-shaped cloud, with max_voxels and max_points binding, incl. the generator
    class the reference probes for (device="cpu")."""
    from com_amd.spconv import utils as U
    from com_amd.utils import synth
    from oracle import oracle as O
    g = golden("g4_meanvfe")
    cases = [(g["points0"], g["range"], g["voxel_size"], 5, 5000), (g["points1"], g["range"], g["voxel_size"], 5, 5000),
             (synth.synth_cloud(3, 16, 1250), synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000),
             (synth.synth_cloud(4, 16, 1250), synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 1, 700),       # both caps bind
             (synth.synth_cloud(5, 16, 250), [0, -39.68, -3, 69.12, 39.68, 1], [0.16, 0.16, 4], 20, 16000),
             (np.zeros((0, 5), np.float32), synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 100)]
    for pts, rng, vs, T, maxv in cases:
        v, c, n = U.voxelize_hard_host(pts, rng, vs, T, maxv)
        if pts.shape[0] == 0:
            assert v.shape[0] == 0
            continue
        vo, co, no = O.voxelize_hard(pts, rng, vs, T, maxv)
        np.testing.assert_array_equal(c, co)
        np.testing.assert_array_equal(n, no)
        np.testing.assert_array_equal(v, vo)
    gen = U.VoxelGeneratorV2(g["voxel_size"], g["range"], 5, 5000, device="cpu")
    out = gen.generate(g["points0"])
    vo, co, no = O.voxelize_hard(g["points0"], g["range"], g["voxel_size"], 5, 5000)
    np.testing.assert_array_equal(out["coordinates"], co)
    np.testing.assert_array_equal(out["voxels"], vo)
    np.testing.assert_array_equal(out["num_points_per_voxel"], no)
