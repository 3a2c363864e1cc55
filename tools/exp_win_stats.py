import sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle as O
from com_amd.utils import synth
pts = synth.synth_cloud(0)
v, c, n = O.voxelize_hard(pts, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000)
idx = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
order = np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 0]))
idx = np.ascontiguousarray(idx[order])
shape = (41, 1504, 1504)
for lvl, geo in ((2, (3, 2, 1)), (3, (3, 2, 1))):
    rc = O.rulebook_conv(idx, shape, (geo[0],) * 3, (geo[1],) * 3, (geo[2],) * 3)
    idx, shape = rc["out_indices"], tuple(int(s) for s in rc["out_shape"])
    nb = O.rulebook_subm(idx, shape)["nbr_out"]          # [27][N]
    N = nb.shape[1]
    for T in (32, 64, 128):
        need = []
        for t0 in range(0, N - T + 1, T):
            tile = nb[:, t0:t0 + T]
            w = 0
            for q in range(3):
                g = tile[9 * q:9 * q + 9]
                val = g[g >= 0]
                if val.size:
                    w = max(w, int(val.max()) - int(val.min()) + 1)
            need.append(w)
        need = np.array(need)
        print(f"level {lvl} rows {N} tile {T}: window rows needed median {int(np.median(need))} p90 {int(np.percentile(need, 90))} p99 {int(np.percentile(need, 99))} max {need.max()};"
              f" fits 1.5x: {(need <= 1.5 * T).mean():.3f} fits 2x: {(need <= 2 * T).mean():.3f} fits 3x: {(need <= 3 * T).mean():.3f}")
