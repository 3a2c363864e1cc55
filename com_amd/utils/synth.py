"""Synthetic Waymo-shaped LiDAR clouds (SURVEY.md section 8d "Synthetic inputs").

Spinning-LiDAR model: n_beams x n_azimuth rays from a sensor at (0, 0, 2.0) over a ground plane
z = 0 and a smooth closed "wall" r_wall(azimuth); 0.5 % outliers so the out-of-range drop path is
always exercised; one permutation (cf. pcdet/datasets/processor/data_processor.py:103-113).
Point features: x, y, z, tanh(intensity) (cf. pcdet/datasets/waymo/waymo_dataset.py:210), elongation.
"""
import numpy as np

WAYMO_RANGE = (-75.2, -75.2, -2.0, 75.2, 75.2, 4.0)          # tools/cfgs/dataset_configs/waymo_dataset.yaml:5
WAYMO_VOXEL = (0.1, 0.1, 0.15)                                # waymo_dataset.yaml:79
WAYMO_MAX_POINTS = 5                                          # waymo_dataset.yaml:80
WAYMO_MAX_VOXELS = 150000                                     # waymo_dataset.yaml:81-84
PILLAR_RANGE = (-74.88, -74.88, -2.0, 74.88, 74.88, 4.0)      # tools/cfgs/waymo_models/pointpillar_1x.yaml:6
PILLAR_VOXEL = (0.32, 0.32, 6.0)                              # pointpillar_1x.yaml:18-23
PILLAR_MAX_POINTS = 20


def synth_cloud(frame, n_beams=64, n_azimuth=2500):
    """Return [n_beams*n_azimuth, 5] float32 points for frame index `frame` (seed 1000+frame)."""
    rng = np.random.default_rng(1000 + frame)
    n = n_beams * n_azimuth
    incl = np.deg2rad(np.linspace(-17.6, 2.4, n_beams))
    step = 2.0 * np.pi / n_azimuth
    az0 = -np.pi + step * np.arange(n_azimuth)
    a_m = rng.uniform(0.0, 0.5, 4) / np.arange(1, 5)
    phi_m = rng.uniform(0.0, 2.0 * np.pi, 4)
    jitter = rng.uniform(-0.5, 0.5, (n_beams, n_azimuth)) * step
    az = az0[None, :] + jitter
    inc = np.broadcast_to(incl[:, None], az.shape)
    r_wall = 25.0 + 20.0 * sum(a_m[m] * np.cos((m + 1) * az + phi_m[m]) for m in range(4))
    r_wall = np.clip(r_wall, 8.0, 74.0)
    with np.errstate(divide="ignore"):
        r_ground = np.where(inc < 0.0, 2.0 / np.sin(-np.minimum(inc, -1e-9)), np.inf)
    r = np.minimum(np.minimum(r_ground, r_wall), 74.0)
    r = r * (1.0 + rng.normal(0.0, 0.002, r.shape))
    x = r * np.cos(inc) * np.cos(az)
    y = r * np.cos(inc) * np.sin(az)
    z = 2.0 + r * np.sin(inc)
    intensity = np.tanh(rng.uniform(0.0, 2.0, r.shape))
    elong = rng.uniform(0.0, 1.5, r.shape)
    pts = np.stack([x, y, z, intensity, elong], axis=-1).reshape(n, 5)
    n_out = max(1, int(round(0.005 * n)))
    pts[n - n_out:, 0] = rng.uniform(-80.0, 80.0, n_out)
    pts[n - n_out:, 1] = rng.uniform(-80.0, 80.0, n_out)
    pts[n - n_out:, 2] = rng.uniform(-3.0, 5.0, n_out)
    pts = pts.astype(np.float32)
    return np.ascontiguousarray(pts[rng.permutation(n)])


def synth_batch(first_frame, batch_size, n_beams=64, n_azimuth=2500):
    """List of per-frame clouds plus the collated [sum N, 6] (b, x, y, z, i, e) array
    (pcdet/datasets/dataset.py:252-259 batch-index padding)."""
    frames = [synth_cloud(first_frame + b, n_beams, n_azimuth) for b in range(batch_size)]
    cat = np.concatenate(
        [np.pad(p, ((0, 0), (1, 0)), constant_values=float(b)) for b, p in enumerate(frames)], 0)
    return frames, np.ascontiguousarray(cat.astype(np.float32))
