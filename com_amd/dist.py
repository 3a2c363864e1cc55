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
        self.numel = n
        self.flat = torch.zeros((n + 3) // 4 * 4, dtype=torch.float32, device=dev)   # padded for 16-byte kernels
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def flatten_parameters(self):
        """Also move the parameters themselves into ONE flat fp32 buffer (each `p.data` becomes a view of it) and
        return it as a single nn.Parameter whose `.grad` is the flat gradient buffer: the optimizer then updates
        the whole model with one fused kernel and gradient clipping is a norm + scale of one tensor."""
        flat = torch.empty_like(self.flat)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat[off:off + n].view_as(p)
                off += n
        self.flat_param = torch.nn.Parameter(flat, requires_grad=True)
        self.flat_param.grad = self.flat
        return self.flat_param

    def clip_grad_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ semantics on the flat buffer (L2 norm, coefficient clamped to 1)."""
        norm = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / (norm + 1e-6), max=1.0))
        return norm

    def clip_divisor_(self, max_norm, out, pre_divisor=1.0):
        """out <- pre_divisor * max((||g / pre_divisor|| + 1e-6) / max_norm, 1): the number the gradients must be
        DIVIDED by to implement (averaging over `pre_divisor` ranks +) clip_grad_norm_.  Handing it to a fused
        optimizer as `grad_scale` (torch.optim.Adam(fused=True) divides the gradients by it inside its kernel)
        saves the separate scaling passes over the whole buffer (mean after all_reduce_sum, clip coefficient)."""
        norm = torch.linalg.vector_norm(self.flat) / pre_divisor
        out.copy_(torch.clamp((norm + 1e-6) / max_norm, min=1.0) * pre_divisor)
        return out

    def all_reduce_sum(self):
        """Gradient SUM over the ranks (one collective); pair with clip_divisor_(..., pre_divisor=world_size)."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        return self.flat

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())
        return self.flat


class FlatAdam:
    """torch.optim.Adam (L2 weight decay, bias correction) + clip_grad_norm_ on the flat buffers of a
    FlatGradBucket with flattened parameters, through pcd_adam_flat_step: two passes over the buffers in three
    launches instead of torch's norm + scalar kernels + scaling pass + multi-tensor Adam.  `step()` expects
    bucket.flat to hold the SUM of the ranks' gradients (all_reduce_sum) and divides by `world` itself."""

    def __init__(self, bucket, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=0.0, world=1):
        assert getattr(bucket, "flat_param", None) is not None, "call bucket.flatten_parameters() first"
        from . import _lib as L
        self.L, self.bucket = L, bucket
        self.lr, self.betas, self.eps, self.wd = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.max_norm, self.world = float(max_norm), float(world)
        p = bucket.flat_param.data
        self.exp_avg = torch.zeros_like(p)
        self.exp_avg_sq = torch.zeros_like(p)
        self.step_dev = torch.zeros((1,), dtype=torch.float32, device=p.device)
        self.grad_norm = torch.zeros((1,), dtype=torch.float32, device=p.device)
        self.ws = torch.empty((max(int(L.lib().pcd_adam_flat_workspace_bytes()), 256),), dtype=torch.uint8, device=p.device)

    def step(self):
        L, b = self.L, self.bucket
        p = b.flat_param.data
        L.check(L.lib().pcd_adam_flat_step(L.ptr(p), L.ptr(b.flat), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                           p.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                           self.max_norm, self.world, L.ptr(self.step_dev), L.ptr(self.grad_norm),
                                           L.ptr(self.ws), self.ws.numel(), L.stream_ptr()), "pcd_adam_flat_step")


def max_over_ranks(value, device="cpu"):
    """Slowest rank's time (the bench contract: MAX over ranks)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
