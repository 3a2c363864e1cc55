"""Data-parallel plumbing of the hot path: frames are independent units, so the
only exchange per step is the gradient sum.  One process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).  Mirrors tools/train.py:69-76,165-166 +
pcdet/datasets/__init__.py:65-72 (DistributedSampler striding)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend="nccl", device=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from torch.distributed.run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def shard_frames(step, rank, world, frames_per_gpu):
    """Frame ids of `rank` for global step `step`: the global batch is world*frames_per_gpu consecutive
    frames, dealt round-robin like DistributedSampler (rank r gets indices r, r+world, ...)."""
    base = step * world * frames_per_gpu
    return [base + rank + j * world for j in range(frames_per_gpu)]


class FlatGradBucket:
    """All parameters' gradients as views into ONE contiguous fp32 buffer, so the per-step exchange is a
    single all-reduce (10.8 MB for the sparse backbone: one large collective suits xGMI's per-link-bound
    rings better than DDP's default 25 MB / many-small-bucket schedule, and it sits outside any captured
    hipGraph)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())
        return self.flat


def max_over_ranks(value, device="cpu"):
    """Slowest rank's time (the bench contract: MAX over ranks)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
