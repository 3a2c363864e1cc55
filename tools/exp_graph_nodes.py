"""Node census of the captured step (hipGraphDebugDotPrint): kernel / memcpy / memset nodes by name.
usage: python tools/exp_graph_nodes.py"""
import argparse, collections, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench
from com_amd import train

args = argparse.Namespace(batch=4, distinct_batches=3, same_shard=False, dense_head=False, com=False, com_ucl=False, config5=False)
dev = torch.device("cuda", 0)
W = bench.build_workload(args, 0, 1, dev)
step = W.step
step.observe(W.batches, steps=2)
orig = torch.cuda.CUDAGraph
made = []


class Dbg(orig):
    def __new__(cls, *a, **k):
        g = orig.__new__(cls, *a, **k)
        g.enable_debug_mode()
        made.append(g)
        return g


torch.cuda.CUDAGraph = Dbg
step.capture(W.batches[0])
torch.cuda.CUDAGraph = orig
out = os.path.join(ROOT, "gpurun_out", "r06_graph.dot")
made[-1].debug_dump(out)
text = open(out).read()
labels = re.findall(r'label="([^"]*)"', text)
census = collections.Counter()
for lb in labels:
    name = lb.split("\\n")[0] if "\\n" in lb else lb
    name = re.sub(r"<.*", "", name)[:60]
    census[name] += 1
edges = text.count("->")
print(f"nodes {len(labels)}  edges {edges}")
for k, v in census.most_common(80):
    print(f"{v:4d}  {k}")
os.remove(out)
