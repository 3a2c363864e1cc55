"""Where the step's device-to-device copies come from: one static-shape (capture-like) step of the bench workload under
torch.profiler with Python stacks; prints every aten::copy_ / clone / contiguous that launched work, grouped by call site.
usage: python tools/exp_copies.py"""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench

args = argparse.Namespace(batch=4, distinct_batches=3, same_shard=False, dense_head=False, com=False, com_ucl=False, config5=False)
dev = torch.device("cuda", 0)
W = bench.build_workload(args, 0, 1, dev)
step = W.step
step.observe(W.batches, steps=2)
step.plan.active = True
step.plan.prepare(dev)
pts, offs = W.batches[0]


def one():
    with step.plan, step.options:
        bd2 = step._voxelize(pts, offs)
        bd_in = dict(bd2)
        step._forward_backward(bd_in)
        step._optimizer_step()


for _ in range(2):
    one()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    one()
    torch.cuda.synchronize()
sites = collections.Counter()
shapes = {}
kern = collections.Counter()
for ev in prof.events():
    if ev.device_type.name != "CPU":
        if "emcpy" in ev.name or "copyBuffer" in ev.name:
            kern[ev.name] += 1
        continue
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::fill_", "aten::zero_"):
        st = [s for s in (ev.stack or []) if "/repo/" in s and "exp_copies" not in s]
        key = (ev.name, " <- ".join(s.split("/repo/")[-1] for s in st[:3]))
        sites[key] += 1
        shapes.setdefault(key, set()).add(str(ev.input_shapes))
print("device-side copy events:", dict(kern))
for (name, where), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {name:18s} {where}   {sorted(shapes[(name, where)])[:2]}")
