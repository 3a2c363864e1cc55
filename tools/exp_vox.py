"""Hard voxelisation timing: shuffled points (training: SHUFFLE_ENABLED train=True in the reference's
DATA_PROCESSOR) vs range-image order (inference), B = 4 frames of 160 k points."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from com_amd import ops
from com_amd.utils import synth


def run(frames, label, reps=30):
    pts = torch.from_numpy(np.concatenate(frames, 0)).cuda()
    offs = [0]
    for f in frames:
        offs.append(offs[-1] + len(f))
    args = (pts, offs, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, synth.WAYMO_MAX_POINTS, synth.WAYMO_MAX_VOXELS)
    for _ in range(3):
        res = ops.voxelize_hard(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.voxelize_hard(*args)
    e1.record()
    torch.cuda.synchronize()
    print(f"{label}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per batch, voxels {sum(res['counts'])}")


if __name__ == "__main__":
    shuffled = [synth.synth_cloud(f) for f in range(4)]
    # undo the shuffle: sort by (inclination, azimuth) = range-image order
    ordered = []
    for p in shuffled:
        r = np.linalg.norm(p[:, :2], axis=1)
        inc = np.round(np.rad2deg(np.arctan2(p[:, 2] - 2.0, r)) * 3.15).astype(np.int64)
        az = np.arctan2(p[:, 1], p[:, 0])
        ordered.append(np.ascontiguousarray(p[np.lexsort((az, inc))]))
    run(shuffled, "shuffled")
    run(ordered, "range-image order")
