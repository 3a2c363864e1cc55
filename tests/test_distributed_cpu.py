"""CPU, world_size 2, gloo: the N > 1 path of the hot path is "shard frames, sum gradients"."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from com_amd import dist as cdist
    from com_amd.hotpath import VoxelResBackBone8x
    r, lr, w = cdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(666 + rank)                       # per-rank seeds like tools/train.py:86-87 ...
    model = VoxelResBackBone8x({}, 5, [1504, 1504, 40])
    # ... DDP broadcasts rank 0's parameters and buffers at construction (tools/train.py:165-166)
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    w0 = model.conv_input[0].weight.detach().clone()
    gathered = [torch.zeros_like(w0) for _ in range(world)]
    dist.all_gather(gathered, w0)
    same_init = all(torch.equal(gathered[0], g) for g in gathered)

    # flat-bucket gradient averaging (the exchange step of the captured-graph training loop)
    bucket = cdist.FlatGradBucket(model.parameters())
    for i, p in enumerate(bucket.params):
        p.grad.fill_(float(rank + 1) * (i + 1))
    bucket.all_reduce_mean()
    expect = sum(range(1, world + 1)) / world
    ok_bucket = all(torch.allclose(p.grad, torch.full_like(p, expect * (i + 1))) for i, p in enumerate(bucket.params))
    n_flat = bucket.flat.numel()
    # sum-reduce + divisor form (what bench.py uses: the mean and the clip coefficient go into Adam's grad_scale)
    for i, p in enumerate(bucket.params):
        p.grad.fill_(float(rank + 1) * (i + 1))
    bucket.all_reduce_sum()
    div = bucket.clip_divisor_(1e9, torch.ones(()), pre_divisor=float(world))     # no clipping: divisor = world
    ok_bucket = ok_bucket and abs(float(div) - world) < 1e-6 and all(
        torch.allclose(p.grad / div, torch.full_like(p, expect * (i + 1))) for i, p in enumerate(bucket.params))

    # DDP's own reducer over the same parameters: grads of sum(p * (rank+1)) are averaged over ranks
    for p in model.parameters():
        p.grad = None
    out = sum((p * float(rank + 1)).sum() for p in ddp.module.parameters())
    # go through ddp.forward so the reducer is armed
    class Wrap(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m
        def forward(self, s):
            return sum((p * s).sum() for p in self.m.parameters())
    ddp2 = torch.nn.parallel.DistributedDataParallel(Wrap(model))
    ddp2(torch.tensor(float(rank + 1))).backward()
    ok_ddp = all(torch.allclose(p.grad, torch.full_like(p, expect)) for p in model.parameters())

    frames = cdist.shard_frames(3, rank, world, 4)
    tmax = cdist.max_over_ranks(10.0 + rank)
    q.put((rank, same_init, ok_bucket, ok_ddp, frames, tmax, n_flat))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharding_and_gradient_exchange():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    frames_all = []
    for rank, same_init, ok_bucket, ok_ddp, frames, tmax, n_flat in res:
        assert same_init and ok_bucket and ok_ddp
        assert tmax == 11.0                                   # MAX over ranks
        assert 2.6e6 < n_flat < 2.8e6                          # 10.8 MB fp32 exchanged per step
        frames_all += frames
    # step 3, 2 ranks x 4 frames: disjoint cover of frames 24..31, round-robin like DistributedSampler
    assert sorted(frames_all) == list(range(24, 32))
    assert res[0][4] == [24, 26, 28, 30] and res[1][4] == [25, 27, 29, 31]


def test_shard_frames_single_rank():
    from com_amd import dist as cdist
    assert cdist.shard_frames(0, 0, 1, 4) == [0, 1, 2, 3]
    assert cdist.shard_frames(2, 0, 1, 4) == [8, 9, 10, 11]
    assert cdist.max_over_ranks(3.5) == 3.5


# ---------------------------------------------------------------------------------------------
# bench.py's own launcher / result assembly (what the driver exercises with `python bench.py --gpus N`)
def _run_bench(args, env=None, timeout=240):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=e, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


@pytest.mark.timeout(300)
def test_bench_gpus_2_starts_its_own_two_ranks():
    """`python bench.py --gpus 2` with no launcher must start 2 ranks itself (never silently run one): the
    --selftest-launch form runs the launcher, the rank environment, shard_frames, the max-over-ranks timing and the
    result line over gloo without touching a GPU."""
    rc, res, err = _run_bench(["--gpus", "2", "--selftest-launch", "--batch", "4"])
    assert rc == 0, err
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["allreduce_sum"] == 3.0
    assert res["frames"] == [[0, 2, 4, 6], [1, 3, 5, 7]]          # DistributedSampler striding, disjoint cover
    assert res["slowest"] == 2.0                                   # MAX over ranks


@pytest.mark.timeout(600)
def test_bench_gpus_8_plumbing_over_gloo():
    """The 8-rank job the driver launches, as far as a CPU can take it: 8 ranks started by bench.py itself, rendezvous and the
    first all-reduce under their watchdogs, DistributedSampler striding over 8 ranks, the REAL exchange of a step (the 10.78 MB
    flat gradient bucket of VoxelResBackBone8x summed over 8 ranks), max-over-ranks timing, one result line."""
    rc, res, err = _run_bench(["--gpus", "8", "--selftest-launch", "--batch", "4"], timeout=540)
    assert rc == 0, err[-3000:]
    assert res["n_gpus"] == 8 and res["rccl_ranks"] == 8 and res["allreduce_sum"] == 36.0
    assert res["frames"] == [[r + 8 * j for j in range(4)] for r in range(8)]   # rank r: r, r + 8, r + 16, r + 24
    assert sorted(f for fr in res["frames"] for f in fr) == list(range(32))     # disjoint cover of the global batch
    assert res["slowest"] == 8.0                                                 # MAX over ranks
    assert res["bucket_sum_ok"] and 10.7 < res["bucket_MB"] < 10.9


@pytest.mark.timeout(300)
def test_a_rank_that_never_joins_the_first_all_reduce_cannot_hang_the_job():
    """One of four ranks never arrives at the first collective: the other ranks' watchdogs end them with code 124 after
    --init-timeout, the launcher stops the straggler, `python bench.py --gpus 4` returns non-zero -- in seconds, not after the
    driver's whole budget."""
    import time
    t0 = time.time()
    rc, res, err = _run_bench(["--gpus", "4", "--selftest-launch", "--selftest-hang", "2", "--init-timeout", "8",
                               "--launch-timeout", "120"], timeout=240)
    took = time.time() - t0
    assert rc == 1 and res is None, err[-2000:]
    assert "exceeded 8 s -- exiting with code 124" in err and "ranks failed" in err
    assert took < 100, took


@pytest.mark.timeout(120)
def test_launcher_deadline_stops_ranks_that_never_finish(tmp_path):
    import sys
    import time
    from com_amd import dist as cdist
    script = tmp_path / "sleeper.py"
    script.write_text("import time\ntime.sleep(600)\n")
    t0 = time.time()
    codes = cdist.launch_local_ranks(3, [sys.executable, str(script)], timeout=3)
    assert time.time() - t0 < 30 and all(c != 0 for c in codes)


def test_watchdog_fires_once_and_can_be_disarmed():
    import time
    from com_amd import dist as cdist
    fired = []
    w = cdist.Watchdog(0.2, "unit test phase", _exit=fired.append)
    time.sleep(0.6)
    assert fired == [124]
    fired2 = []
    with cdist.Watchdog(0.3, "disarmed phase", _exit=fired2.append):
        pass
    time.sleep(0.6)
    assert fired2 == []
    assert cdist.first_all_reduce() == 1.0                     # no process group: nothing to wait for


@pytest.mark.timeout(120)
def test_bench_refuses_a_world_size_that_is_not_gpus():
    rc, res, err = _run_bench(["--gpus", "2", "--selftest-launch"], env={"WORLD_SIZE": "3", "RANK": "0"})
    assert rc == 2 and res is None and "WORLD_SIZE=3" in err


@pytest.mark.timeout(120)
def test_launcher_propagates_a_failing_rank(tmp_path):
    import sys
    from com_amd import dist as cdist
    script = tmp_path / "child.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK'])\n"
                      "assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "assert os.environ['LOCAL_RANK'] == os.environ['RANK']\n"
                      "if r == 1: sys.exit(7)\n"
                      "time.sleep(60)\n")
    codes = cdist.launch_local_ranks(3, [sys.executable, str(script)])
    assert codes[1] == 7 and all(c != 0 for c in codes)            # the other ranks were stopped, not left running


def test_one_cycle_schedule_matches_reference_fixture(golden):
    """G8 holds (lr, momentum) of the reference's OneCycle (learning_schedules_fastai.py:60-77) for every step."""
    import numpy as np
    from com_amd import dist as cdist
    g = golden("g8_adam_onecycle")
    total = int(g["total_steps"][0])
    mine = np.array([cdist.one_cycle(i, total) for i in range(total)])
    np.testing.assert_allclose(mine[:, 0], g["lr"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(mine[:, 1], g["mom"], rtol=1e-12, atol=0)
    assert mine[0, 1] == 0.95 and abs(mine[0, 0] - 3e-4) < 1e-15   # MOMS[0], LR / DIV_FACTOR


def _gather_worker(rank, world, port, q):
    import numpy as np
    import torch
    import torch.distributed as dist
    from com_amd import dist as cdist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g14_com_epoch_gather.npz"))
    conf = torch.zeros(3, 96)
    num = torch.zeros(3, 96)
    for s in range(g["conf"].shape[1]):                      # what the loss kernel does per step: += in float32
        conf += torch.from_numpy(g["conf"][rank, s])
        num += torch.from_numpy(g["num"][rank, s])
    out = cdist.gather_group_confidence(conf, num)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_com_epoch_gather_two_ranks_matches_reference_arithmetic_g14():
    """COM's per-epoch (3, 96) all_gather (tools/train_utils/train_utils.py:269-287) over a world-size-2 gloo group:
    every rank ends with exactly the array the reference's arithmetic gives (fixture G14), float32."""
    import socket
    import numpy as np
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g14_com_epoch_gather.npz"))
    for r in range(2):
        assert got[r].dtype == np.float32
        np.testing.assert_array_equal(got[r], g["result"])


def test_com_epoch_gather_without_a_process_group_g14():
    import numpy as np
    import torch
    from com_amd import dist as cdist
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g14_com_epoch_gather.npz"))
    conf = sum(torch.from_numpy(g["conf"][0, s]) for s in range(g["conf"].shape[1]))
    num = sum(torch.from_numpy(g["num"][0, s]) for s in range(g["num"].shape[1]))
    np.testing.assert_array_equal(cdist.gather_group_confidence(conf, num), g["single_rank0"])
