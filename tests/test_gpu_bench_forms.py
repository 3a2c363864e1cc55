"""GPU: bench.py's execution forms must be the same computation.

  * the N = 1 form (ONE hipGraph per step) and the N > 1 form run on one GPU (PCD_FORCE_3GRAPH=1: one graph for
    forward+backward incl. the mid-forward voxelisation of the next batch, then the all-reduce and clip+Adam as plain
    launches; PCD_N_GT_1_FORM=3graph: the older voxelise-graph | forward+backward-graph form) must leave BIT-IDENTICAL
    parameters and gradients after the same steps on the same data;
  * two ranks (PCD_DIST_ONE_GPU=1, gloo, both on cuda:0; bench.py starts them itself from `--gpus 2`) that both
    process rank 0's frames (--same-shard) sum two identical gradients and divide by the world size -- exact in
    binary floating point -- so they too must end bit-identical to the one-rank run: this exercises the launcher,
    the broadcast of rank 0's parameters, the flat-bucket all-reduce and the rank mean inside the fused Adam.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(tmp_path, tag, args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PCD_FORCE_3GRAPH", "PCD_DIST_ONE_GPU",
              "PCD_DIST_BACKEND", "PCD_RCCL_WORLD1", "PCD_N_GT_1_FORM"):
        e.pop(k, None)
    e.update(env or {})
    dump = tmp_path / f"{tag}.json"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--batch", "2",
           "--no-cpu-baseline", "--no-roofline", "--no-h2d", "--no-ragged", "--no-full-model", "--no-stage2", "--no-fp8", "--no-seam-path", "--no-n-gt-1", "--distinct-batches", "3",
           "--same-shard", "--dump-state", str(dump)] + args
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert "running eagerly" not in r.stderr, r.stderr[-2000:]      # the graph forms really were captured
    return json.loads(line), json.load(open(dump))


@pytest.mark.timeout(1800)
def test_one_graph_three_graph_and_two_rank_forms_are_bit_identical(tmp_path):
    res1, st1 = _bench(tmp_path, "one_graph", ["--gpus", "1"])
    res3, st3 = _bench(tmp_path, "n_gt_1", ["--gpus", "1"], env={"PCD_FORCE_3GRAPH": "1"})
    assert res1["n_gpus"] == res3["n_gpus"] == 1
    assert "one graph" in res1["config"]["execution"] and "all-reduce, clip+Adam as plain launches" in res3["config"]["execution"]
    assert st1["param_sha256"] == st3["param_sha256"], (st1, st3)
    assert st1["grad_sha256"] == st3["grad_sha256"]
    # the older N > 1 form (voxelise-graph | forward+backward-graph on two streams), still selectable
    res3o, st3o = _bench(tmp_path, "three_graph", ["--gpus", "1"], env={"PCD_FORCE_3GRAPH": "1", "PCD_N_GT_1_FORM": "3graph"})
    assert "voxelise [prefetched" in res3o["config"]["execution"]
    assert st1["param_sha256"] == st3o["param_sha256"] and st1["grad_sha256"] == st3o["grad_sha256"]
    res2, st2 = _bench(tmp_path, "two_ranks", ["--gpus", "2"], env={"PCD_DIST_ONE_GPU": "1", "PCD_DIST_BACKEND": "gloo"})
    assert res2["n_gpus"] == 2 and res2["rccl_ranks"] == 2 and len(res2["ms_per_step_ranks"]) == 2
    assert res2["config"]["global_batch"] == 4 and res2["config"]["parallelism"] == "dp2"
    assert st2["world"] == 2
    # bucket.flat holds the rank SUM at N = 2: compare the parameters (which used sum / world) bit for bit
    assert st2["param_sha256"] == st1["param_sha256"], (st1, st2)


@pytest.mark.timeout(1800)
def test_rccl_communicator_runs_the_three_graph_form_on_one_gpu(tmp_path):
    """backend="nccl" (= RCCL) must have executed on hardware before an 8-GPU run depends on it: a ONE-rank RCCL
    communicator, the N > 1 three-graph form, the real dist.all_reduce of the flat gradient bucket between the graph
    replays.  A sum over one rank is the identity, so the parameters must equal the one-graph run bit for bit."""
    res1, st1 = _bench(tmp_path, "one_graph_b", ["--gpus", "1"])
    resr, str_ = _bench(tmp_path, "rccl_world1", ["--gpus", "1"], env={"PCD_RCCL_WORLD1": "1"})
    assert resr["collective_backend"] == "nccl" and "all-reduce, clip+Adam as plain launches" in resr["config"]["execution"]
    assert str_["param_sha256"] == st1["param_sha256"], (st1, str_)


@pytest.mark.timeout(2400)
def test_com_full_model_forms_agree(tmp_path):
    """BASELINE config 3 as bench.py composes it (`--dense-head --com`: hot path + BaseBEVBackbone + CenterHead towers +
    COM curriculum targets / FocalLossCenterCurriculum, all inside the captured step): the one-graph form, the three-graph
    form and two ranks with identical shards (gloo on one GPU) all capture, count the same objects into the same (3, 96)
    groups and end with BIT-IDENTICAL parameters.  (Until the stride-2 conv and the two ConvTranspose2d of
    BaseBEVBackbone ran in MIOpen, whose weight gradients use atomics, two runs of the SAME form differed in the last
    bits -- round 3 replaced them by the plane kernels of conv2d.hip and the whole step is reproducible:
    `tools/exp_forms.sh`.)"""
    args = ["--dense-head", "--com"]
    res1, st1 = _bench(tmp_path, "com_one", ["--gpus", "1"] + args)
    assert res1["config"]["com_head"] and res1["com"]["groups_seen"] > 10 and res1["com"]["objects_counted"] > 0
    res1b, st1b = _bench(tmp_path, "com_one_again", ["--gpus", "1"] + args)
    assert st1b["param_sha256"] == st1["param_sha256"] and st1b["grad_sha256"] == st1["grad_sha256"]   # run to run
    res3, st3 = _bench(tmp_path, "com_three", ["--gpus", "1"] + args, env={"PCD_FORCE_3GRAPH": "1"})
    assert st3["param_sha256"] == st1["param_sha256"], (st1, st3)
    res2, st2 = _bench(tmp_path, "com_two", ["--gpus", "2"] + args, env={"PCD_DIST_ONE_GPU": "1", "PCD_DIST_BACKEND": "gloo"})
    assert res2["n_gpus"] == 2 and res2["com"]["groups_seen"] == res1["com"]["groups_seen"]
    assert res2["com"]["objects_counted"] == res1["com"]["objects_counted"]
    assert st2["param_sha256"] == st1["param_sha256"], (st1, st2)


@pytest.mark.timeout(1200)
def test_config5_fp8_forward_training_step_captures_at_300k_points(tmp_path):
    """BASELINE config 5 as bench.py runs it: SECOND's VoxelBackBone8x, 300 k-point clouds (the 150 k voxel cap binds), e4m3
    forward convs on 11 layers + bf16 backward inside ONE captured step; and the same workload with the bf16 forward."""
    res8, st8 = _bench(tmp_path, "cfg5_fp8", ["--gpus", "1", "--config5"])
    assert "fp8 (e4m3) FORWARD convs on 11 layers" in res8["config"]["workload"] and res8["config"]["points_per_frame"] == 300000
    assert res8["config"]["voxels_per_frame"] > 100000 and res8["value"] > 0 and "one graph" in res8["config"]["execution"]
    res16, st16 = _bench(tmp_path, "cfg5_bf16", ["--gpus", "1", "--config5"], env={"PCD_CONFIG5_BF16": "1"})
    assert st8["param_sha256"] != st16["param_sha256"]                      # the fp8 forward really ran
    assert abs(st8["param_sum"] - st16["param_sum"]) <= 2e-2 * abs(st16["param_sum"])


@pytest.mark.timeout(1200)
def test_h2d_inclusive_leg_runs_in_both_forms(tmp_path):
    """The a3 leg of the bench (points of every batch arriving from pinned host memory inside the step): the copy-stream
    form with its trial-based stream choice, and the opt-in form where a kernel of the step's own graph pulls the batch over
    PCIe (pcd_pull_from_host) -- both must run the captured step to the end (sticky overflow flag checked) and report a
    rate of the order of the resident one."""
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PCD_H2D_PULL"):
        e.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--batch", "2", "--no-cpu-baseline",
           "--no-roofline", "--no-ragged", "--no-full-model", "--no-stage2", "--no-fp8", "--no-seam-path", "--no-n-gt-1", "--distinct-batches", "3", "--gpus", "1"]
    for env, form in (({"PCD_H2D_CANDIDATES": "2"}, "copy stream"), ({"PCD_H2D_PULL": "1"}, "pulled by a kernel")):
        r = subprocess.run(cmd, env={**e, **env}, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        assert "H2D-inclusive loop failed" not in r.stderr, r.stderr[-2000:]
        res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        h = res["h2d_inclusive"]
        assert form in h["form"] and 0.3 * res["value"] < h["value"] < 1.2 * res["value"], h
