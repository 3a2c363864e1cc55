"""Kernel breakdown of the PV-RCNN stage-2 workload (bench.py's `stage2` key): run under rocprofv3 --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
r = bench.measure_stage2(4, torch.device("cuda:0"))
print(r)
