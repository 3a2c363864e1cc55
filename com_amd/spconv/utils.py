"""Voxel generators with the signatures the reference probes for in
pcdet/datasets/processor/data_processor.py:17-42 (SURVEY.md 8b "Import seam 2"), backed by the HIP
hard-voxelisation kernels.  Inputs may be numpy arrays (as in the reference's DataLoader workers;
they are staged to the GPU) or device tensors (the MI355X-first path: no PCIe round trip)."""
import numpy as np
import torch

from .. import ops


def _in_forked_worker():
    """The reference calls the voxel generator inside DataLoader WORKER processes (data_processor.py:130-141, workers
    default to fork, tools/train.py `--workers 4`).  A forked child of a process that has initialised HIP cannot use the
    GPU ("Cannot re-initialize CUDA in forked subprocess"): numpy points arriving there take the library's HOST variant
    (pcd_voxelize_hard_host: product code, same results bit for bit) -- the literal drop-in for the seam; the MI355X-first
    integration keeps raw points in the batch and voxelises the collated device tensor."""
    return bool(getattr(torch.cuda, "_is_in_bad_fork", lambda: False)())


def voxelize_hard_host(points, point_cloud_range, voxel_size, max_num_points, max_voxels):
    """One frame on the CPU through the C ABI (no GPU work): (voxels [M, T, C] f32, coords [M, 3] (z, y, x) i32, num_points [M])."""
    import ctypes
    from .. import _lib as L
    pts = np.ascontiguousarray(points, dtype=np.float32)
    n, c = pts.shape
    cap = max(1, min(n, int(max_voxels)))
    voxels = np.empty((cap, int(max_num_points), c), np.float32)
    coords = np.empty((cap, 3), np.int32)
    nump = np.empty((cap,), np.int32)
    m = ctypes.c_int32(0)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    L.check(L.lib().pcd_voxelize_hard_host(p(pts), n, c, c, L.host_f32(point_cloud_range), L.host_f32(voxel_size),
                                           int(max_num_points), cap, p(voxels), p(coords), p(nump), ctypes.byref(m)),
            "pcd_voxelize_hard_host")
    return voxels[:m.value], coords[:m.value], nump[:m.value]


class VoxelGeneratorV2:
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, device="cuda"):
        self._voxel_size = [float(v) for v in voxel_size]
        self._point_cloud_range = [float(v) for v in point_cloud_range]
        self._max_num_points = int(max_num_points)
        self._max_voxels = int(max_voxels)
        self._grid_size = ops.grid_size(self._point_cloud_range, self._voxel_size)
        self._device = device

    def generate(self, points, max_voxels=None):
        """points [N, C] (x, y, z, ...).  Returns dict(voxels, coordinates (z,y,x), num_points_per_voxel)
        of the input's kind (numpy in -> numpy out)."""
        as_numpy = isinstance(points, np.ndarray)
        if as_numpy and (self._device == "cpu" or _in_forked_worker()):
            v, c, n = voxelize_hard_host(points, self._point_cloud_range, self._voxel_size, self._max_num_points,
                                         max_voxels or self._max_voxels)
            return {"voxels": v, "coordinates": c, "num_points_per_voxel": n}
        pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(self._device) \
            if as_numpy else points.contiguous().float()
        res = ops.voxelize_hard(pts, [0, pts.shape[0]], self._point_cloud_range, self._voxel_size,
                                self._max_num_points, max_voxels or self._max_voxels, want_mean=False)
        voxels, coords, nump = res["voxels"], res["coords"][:, 1:].contiguous(), res["num_points"]
        if as_numpy:
            voxels, coords, nump = voxels.cpu().numpy(), coords.cpu().numpy(), nump.cpu().numpy()
        return {"voxels": voxels, "coordinates": coords, "num_points_per_voxel": nump}

    @property
    def voxel_size(self):
        return self._voxel_size

    @property
    def max_num_points_per_voxel(self):
        return self._max_num_points

    @property
    def point_cloud_range(self):
        return self._point_cloud_range

    @property
    def grid_size(self):
        return self._grid_size


VoxelGenerator = VoxelGeneratorV2


class Point2VoxelGPU3d:
    """spconv-2.x style signature (data_processor.py:36-42); `point_to_voxel` takes a device tensor or
    numpy array and returns three tensors/arrays (no cumm.tensorview needed)."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel,
                 max_num_voxels, device="cuda"):
        self._gen = VoxelGeneratorV2(vsize_xyz, coors_range_xyz, max_num_points_per_voxel, max_num_voxels,
                                     device)
        self.num_point_features = num_point_features

    def point_to_voxel(self, points):
        out = self._gen.generate(points)
        return out["voxels"], out["coordinates"], out["num_points_per_voxel"]


Point2VoxelCPU3d = Point2VoxelGPU3d  # name the reference imports (data_processor.py:25)
