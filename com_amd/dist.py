"""Data-parallel plumbing of the hot path: frames are independent units, so the
only exchange per step is the gradient sum.  One process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).  Mirrors tools/train.py:69-76,165-166 +
pcdet/datasets/__init__.py:65-72 (DistributedSampler striding)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend="nccl", device=None, timeout_s=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from torch.distributed.run.  `timeout_s`: the process group's
    collective timeout (torch's RCCL watchdog tears the rank down when a collective exceeds it)."""
    import datetime
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        if timeout_s is not None:
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


class Watchdog:
    """A deadline for a phase of a rank that must not hang a multi-GPU job (rendezvous + communicator set-up + the FIRST
    all-reduce; the whole run): a daemon thread that, unless `disarm()` came first, prints what timed out and ends THIS
    process with exit code 124 -- `os._exit`, no clean-up, because a rank stuck inside an RCCL call cannot unwind.  The
    launcher (launch_local_ranks) sees the non-zero code and stops the other ranks; under torch.distributed.run the agent
    does.  Nothing is exec'ed and no other process is signalled."""

    def __init__(self, seconds, what, _exit=None):
        import threading
        self.what, self.seconds = what, float(seconds)
        self._done = threading.Event()
        self._exit = _exit or os._exit
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def _run(self):
        import sys
        if not self._done.wait(self.seconds):
            print(f"[com_amd.dist] rank {os.environ.get('RANK', '0')}: {self.what} exceeded {self.seconds:.0f} s -- "
                  f"exiting with code 124", file=sys.stderr, flush=True)
            self._exit(124)

    def disarm(self):
        self._done.set()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.disarm()
        return False


def first_all_reduce(device="cpu", timeout_s=120.0):
    """The first collective of a job under a Watchdog: communicator creation is lazy, so THIS is where a bad fabric /
    a missing peer shows.  Returns the sum of the ranks' (rank + 1) -- every rank checks it."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return 1.0
    with Watchdog(timeout_s, "the first all-reduce (communicator set-up)"):
        t = torch.tensor([float(dist.get_rank() + 1)], dtype=torch.float32, device=device)
        dist.all_reduce(t)
        if t.is_cuda:
            torch.cuda.synchronize()
        got = float(t.item())
    world = dist.get_world_size()
    if got != world * (world + 1) / 2:
        raise RuntimeError(f"first all-reduce returned {got}, expected {world * (world + 1) / 2}")
    return got


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
        self.force_collective = False      # run the all-reduce even in a 1-rank group (exercises RCCL on a 1-GPU box)
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
        if dist.is_initialized() and (dist.get_world_size() > 1 or self.force_collective):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        return self.flat

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())
        return self.flat


def one_cycle(step, total_steps, lr_max=3e-3, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4):
    """(lr, beta1) of training step `step` under the reference's OneCycle schedule
    (tools/train_utils/optimization/learning_schedules_fastai.py:12-77 with centerpoint.yaml:81-88): cosine from
    lr_max / div_factor up to lr_max over the first pct_start of the steps while the momentum goes MOMS[0] ->
    MOMS[1], then cosine down to lr_max / div_factor / 1e4 while the momentum returns."""
    import math

    def cos(start, end, pct):
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    a1 = int(total_steps * pct_start)
    low = lr_max / div_factor
    if step < a1 or a1 >= total_steps:
        pct = step / max(a1, 1)
        return cos(low, lr_max, pct), cos(moms[0], moms[1], pct)
    pct = (step - a1) / max(total_steps - a1, 1)
    return cos(lr_max, low / 1e4, pct), cos(moms[1], moms[0], pct)


class FlatAdam:
    """clip_grad_norm_ + Adam on the flat buffers of a FlatGradBucket with flattened parameters, through
    pcd_adam_flat_step_v2: two passes over the buffers in three launches instead of torch's norm + scalar kernels +
    scaling pass + multi-tensor Adam.  `step()` expects bucket.flat to hold the SUM of the ranks' gradients
    (all_reduce_sum) and divides by `world` itself.

    decoupled=False (default) is torch.optim.Adam's L2 weight decay.  decoupled=True is the reference's
    `adam_onecycle` update (tools/train_utils/optimization/__init__.py:19-32: OptimWrapper(Adam(betas=(0.9, 0.99)), wd,
    true_wd=True, bn_wd=True); fastai_optim.py:135-150): every parameter -- BatchNorm ones included -- is multiplied by
    (1 - lr * wd), then Adam runs without a weight-decay term.  lr / beta1 live in a device float[2] (`hyper`) so that
    a OneCycle schedule can drive a replayed hipGraph: either `set_schedule(table)` (the kernel looks the step's row
    up itself) or `set_hyper(*one_cycle(it, total))` per step.  Assigning `opt.lr = x` / `opt.betas = (b1, b2)` takes
    effect on the next step() as with torch optimizers (the device pair is refreshed; beta2 is a launch argument,
    i.e. baked into an already captured graph)."""

    def __init__(self, bucket, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=0.0, world=1,
                 decoupled=False):
        assert getattr(bucket, "flat_param", None) is not None, "call bucket.flatten_parameters() first"
        from . import _lib as L
        self.L, self.bucket = L, bucket
        self._lr, self._betas, self.eps, self.wd = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self._hyper_dirty = False
        self.max_norm, self.world, self.decoupled = float(max_norm), float(world), bool(decoupled)
        p = bucket.flat_param.data
        self.exp_avg = torch.zeros_like(p)
        self.exp_avg_sq = torch.zeros_like(p)
        self.step_dev = torch.zeros((1,), dtype=torch.float32, device=p.device)
        self.grad_norm = torch.zeros((1,), dtype=torch.float32, device=p.device)
        self.hyper = torch.tensor([self._lr, self._betas[0]], dtype=torch.float32, device=p.device)
        # pinned staging RING for set_hyper: the host runs ahead of the device when steps are graph replays, so a
        # staging slot must not be rewritten before its async copy has executed (slot events guard the reuse)
        self._hyper_host = torch.zeros((64, 2), dtype=torch.float32)
        if p.is_cuda:
            self._hyper_host = self._hyper_host.pin_memory()
        self._hyper_ev = [None] * 64
        self._hyper_i = 0
        self.ws = torch.empty((max(int(L.lib().pcd_adam_flat_workspace_bytes()), 256),), dtype=torch.uint8, device=p.device)
        self.schedule = None

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr, self._hyper_dirty = float(value), True

    @property
    def betas(self):
        return self._betas

    @betas.setter
    def betas(self, value):
        self._betas, self._hyper_dirty = (float(value[0]), float(value[1])), True

    def set_schedule(self, pairs):
        """The whole (lr, beta1) schedule as a device table [T, 2] (row = number of updates done so far, the last row
        repeats): `step()` then looks this step's pair up on the device -- inside a replayed hipGraph too -- and no
        per-step host -> device copy (`set_hyper`) sits in front of the step."""
        t = torch.as_tensor(pairs, dtype=torch.float32)
        assert t.dim() == 2 and t.shape[1] == 2 and t.shape[0] >= 1
        self.schedule = t.to(self.hyper.device).contiguous()

    def set_hyper(self, lr, beta1):
        """lr / beta1 of the NEXT step(s) (async H2D of 8 bytes on the current stream; capturable graphs read the
        device copy)."""
        self._lr, self._betas, self._hyper_dirty = float(lr), (float(beta1), self._betas[1]), False
        i = self._hyper_i
        self._hyper_i = (i + 1) % len(self._hyper_ev)
        if self._hyper_ev[i] is not None:
            self._hyper_ev[i].synchronize()            # only blocks when the host is 64 steps ahead
        self._hyper_host[i, 0], self._hyper_host[i, 1] = self.lr, self.betas[0]
        self.hyper.copy_(self._hyper_host[i], non_blocking=True)
        if self.hyper.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._hyper_ev[i] = ev

    def step(self, zero_grad=False):
        """zero_grad=True: the gradient bucket is cleared by the Adam pass itself as it consumes it (the next step's zero_grad
        without a fill launch: do not call bucket.zero() then)."""
        L, b = self.L, self.bucket
        p = b.flat_param.data
        if self._hyper_dirty:                          # `opt.lr = ...` / `opt.betas = ...` since the last step
            self.set_hyper(self._lr, self._betas[0])
        sched = self.schedule                          # (the kernel reads row min(step, T - 1) itself: no lookup launches)
        L.check(L.lib().pcd_adam_flat_step_v4(L.ptr(p), L.ptr(b.flat), int(bool(zero_grad)), L.ptr(self.exp_avg),
                                              L.ptr(self.exp_avg_sq),
                                              p.numel(), self._lr, self._betas[0], self._betas[1], self.eps, self.wd,
                                              self.max_norm, self.world, int(self.decoupled), L.ptr(self.hyper),
                                              L.ptr(sched) if sched is not None else None,
                                              int(sched.shape[0]) if sched is not None else 0,
                                              L.ptr(self.step_dev), L.ptr(self.grad_norm), L.ptr(self.ws),
                                              self.ws.numel(), L.stream_ptr()), "pcd_adam_flat_step_v4")


def gather_group_confidence(conf_epoch, num_epoch):
    """COM's per-epoch exchange (tools/train_utils/train_utils.py:269-287): every rank contributes the epoch sums of
    its per-step (3, 96) group-confidence sums and counts (`FocalLossCenterCurriculumState.epoch_confidence` /
    `.epoch_num`); all_gather both (RCCL when the tensors live on the GPU), add the ranks' tensors in rank order and
    return conf / (num + 0.1) as the float32 numpy array the reference hands to COMAug's database sampler
    (`confidence_groups`, :321-323).  Without a process group: conf / (num + 0.01) (:325)."""
    import numpy as np
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return (conf_epoch / (num_epoch + 0.01)).cpu().numpy()
    world = dist.get_world_size()
    out = []
    for t in (conf_epoch, num_epoch):
        t = t.contiguous()
        if dist.get_backend() == "gloo":
            t = t.cpu()                                   # (CPU tests / one-GPU validation runs)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        acc = np.zeros(tuple(t.shape), np.float32)
        for p in parts:                                   # sum(list) = ((0 + r0) + r1) + ... in float32
            acc = acc + p.cpu().numpy()
        out.append(acc)
    return out[0] / (out[1] + 0.1)


def launch_local_ranks(n, argv, env=None, master_port=None, timeout=None):
    """Start `n` ranks of `argv` (a python command line) on THIS node, one process per GPU, the way
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` would (tools/scripts/dist_train.sh:18): RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in the environment.  The caller must NOT have
    touched the GPU (children are new processes, nothing is exec'ed over this one).  Rank 0 inherits stdout; the
    other ranks' stdout goes to stderr.  Returns the list of exit codes; if a rank fails the others are terminated."""
    import socket
    import subprocess
    import sys
    import time
    if master_port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            master_port = s.getsockname()[1]
    procs = []
    # Per-rank CPU affinity is OPT-IN (PCD_RANK_AFFINITY=1): contiguous blocks of the allowed cores ignore NUMA and SMT
    # numbering (where logical CPUs N/2.. are the hyperthreads of 0..N/2-1 the upper ranks would land on the lower ranks'
    # siblings), and binding to the GPU-local node measured slower for the H2D loop on the gpurun boxes.  The child applies it
    # to itself at start-up (PCD_PIN_CPUS, read by apply_rank_affinity()) -- no preexec_fn: that hook runs between fork and
    # exec of a multi-threaded parent (torch is loaded here), which Python documents as unsafe.
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cores = []
    per = len(cores) // n if (cores and os.environ.get("PCD_RANK_AFFINITY") == "1") else 0

    for r in range(n):
        e = dict(os.environ if env is None else env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(master_port)})
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if per >= 1:
            e["PCD_PIN_CPUS"] = ",".join(str(c) for c in cores[r * per:(r + 1) * per])
        procs.append(subprocess.Popen(list(argv), env=e, stdout=None if r == 0 else sys.stderr))
    t0 = time.time()
    codes = [None] * n
    try:
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
            if any(c not in (None, 0) for c in codes) or (timeout is not None and time.time() - t0 > timeout):
                break
            time.sleep(0.05)
    finally:
        for i, p in enumerate(procs):                 # a failed / timed-out job: stop exactly the PIDs started here
            if codes[i] is None:
                p.terminate()
                try:
                    codes[i] = p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    codes[i] = p.wait()
                codes[i] = codes[i] if codes[i] not in (None, 0) else -15
    return codes


def apply_rank_affinity():
    """Child side of launch_local_ranks' opt-in pinning: bind this process to the cores listed in PCD_PIN_CPUS (call it
    first thing in the rank's main, before anything touches the GPU)."""
    spec = os.environ.get("PCD_PIN_CPUS")
    if spec:
        try:
            os.sched_setaffinity(0, {int(c) for c in spec.split(",") if c})
        except (AttributeError, OSError, ValueError):
            pass


def max_over_ranks(value, device="cpu"):
    """Slowest rank's time (the bench contract: MAX over ranks)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _gpu_local_cpus(device_index):
    import os
    p = torch.cuda.get_device_properties(device_index)
    bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    text = open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read().strip()
    cpus = set()
    for part in text.split(","):
        if "-" in part:
            a, b = part.split("-")
            cpus |= set(range(int(a), int(b) + 1))
        elif part.strip():
            cpus.add(int(part))
    return cpus


def bind_to_gpu_numa(device_index, n_local=1):
    """Restrict this process to CPUs local to its GPU's PCIe root (sysfs `local_cpulist` of the device), so that the
    launch thread and the pinned staging buffers it allocates sit on the GPU's NUMA node; with n_local > 1 ranks on the
    node, the ranks whose GPUs share a NUMA node split that node's cores between them (rank = device index).  Call it
    right after torch.cuda.set_device(), before allocating pinned memory.  Returns the CPU set, or None when nothing was
    changed (no sysfs entry, PCD_NO_AFFINITY=1, one NUMA node, or nothing left after intersecting with the current mask)."""
    import os
    if os.environ.get("PCD_NO_AFFINITY"):
        return None
    try:
        allowed = set(os.sched_getaffinity(0))
        local = _gpu_local_cpus(device_index)
        mine = sorted(local & allowed)
        if not mine:
            return None
        if n_local > 1:
            peers = [j for j in range(n_local) if _gpu_local_cpus(j) == local]
            per = len(mine) // len(peers)
            if per >= 1:
                k = peers.index(device_index)
                mine = mine[k * per:(k + 1) * per]
        if set(mine) == allowed:
            return None
        os.sched_setaffinity(0, set(mine))
        return set(mine)
    except Exception:
        return None
