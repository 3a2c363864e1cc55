"""Window-fit statistics of SubM 3x3x3 neighbourhoods under different ROW ORDERS (CPU only).

For every level of VoxelResBackBone8x on frames 0..3 (one batch, B = 4) the rows are numbered in a candidate order, cut
into tiles of T consecutive rows, and for every tile the 27 neighbour offsets are grouped into three RUNS (the nine
offsets that share the slowest-varying spatial step of that order).  A run's window is max - min + 1 over the valid
neighbour rows of its nine offsets.  Reported per (level, order, T): window rows median / p90 / p99 / max of the LARGEST
run, of the SUM of the three runs, and the share of tiles whose three runs each fit 1.5x / 2x / 3x the tile.

orders:  zyx   (b, z, y, x)                    -- round 1-3 layout, runs grouped by dz
         yxz   (b, y, x, z)  z fastest          -- runs grouped by dy
         brick (b, y>>3, x>>3, z, y&7, x&7)     -- runs grouped by dz (inside a brick) -- neighbours in other bricks are far
usage: python tools/exp_win_stats.py > profiles/r04_win_stats.txt
"""
import sys
import numpy as np
sys.path.insert(0, '.')
from oracle import oracle as O
from com_amd.utils import synth

B = 4
LEVELS = ((2, (3, 2, 1)), (3, (3, 2, 1)), (4, (3, 2, (0, 1, 1))))


def level_indices():
    rows = []
    for f in range(B):
        pts = synth.synth_cloud(f)
        _, c, _ = O.voxelize_hard(pts, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000)
        rows.append(np.concatenate([np.full((c.shape[0], 1), f, np.int32), c], 1))
    idx = np.concatenate(rows, 0)
    order = np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 0]))
    idx = np.ascontiguousarray(idx[order])
    shape = (41, 1504, 1504)
    out = [(1, idx, shape)]
    for lvl, (k, s, p) in LEVELS:
        pd = p if isinstance(p, tuple) else (p,) * 3
        rc = O.rulebook_conv(idx, shape, (k,) * 3, (s,) * 3, pd)
        idx, shape = rc["out_indices"], tuple(int(v) for v in rc["out_shape"])
        out.append((lvl, idx, shape))
    return out


def order_key(idx, shape, name):
    b, z, y, x = (idx[:, i].astype(np.int64) for i in range(4))
    D, H, W = shape
    if name == "zyx":
        return ((b * D + z) * H + y) * W + x
    if name == "yxz":
        return ((b * H + y) * W + x) * 64 + z
    if name == "brick":
        return ((((b * ((H + 7) >> 3) + (y >> 3)) * ((W + 7) >> 3) + (x >> 3)) * 64 + z) * 8 + (y & 7)) * 8 + (x & 7)
    raise ValueError(name)


GROUP_AXIS = {"zyx": 0, "yxz": 1, "brick": 0}          # which of (dz, dy, dx) defines a run


def neighbour_rows(idx, shape, name):
    """rank of every row under the order and nbr[27][N] in that numbering (-1 = absent)."""
    D, H, W = shape
    key = order_key(idx, shape, name)
    perm = np.argsort(key, kind="stable")
    skey = key[perm]
    idx_s = idx[perm]
    n = idx.shape[0]
    nbr = np.full((27, n), -1, np.int64)
    k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                q = idx_s.copy()
                q[:, 1] += dz; q[:, 2] += dy; q[:, 3] += dx
                ok = (q[:, 1] >= 0) & (q[:, 1] < D) & (q[:, 2] >= 0) & (q[:, 2] < H) & (q[:, 3] >= 0) & (q[:, 3] < W)
                qk = order_key(np.where(ok[:, None], q, 0), shape, name)
                pos = np.searchsorted(skey, qk)
                pos = np.minimum(pos, n - 1)
                hit = ok & (skey[pos] == qk)
                nbr[k, hit] = pos[hit]
                k += 1
    return nbr


def stats(nbr, name, T):
    n = nbr.shape[1]
    ax = GROUP_AXIS[name]
    offs = [(dz, dy, dx) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    groups = [[k for k, o in enumerate(offs) if o[ax] == g] for g in (-1, 0, 1)]
    nt = n // T
    big = np.iinfo(np.int64).max
    wins = np.zeros((3, nt), np.int64)
    for gi, ks in enumerate(groups):
        g = nbr[ks][:, :nt * T].reshape(len(ks), nt, T)
        lo = np.where(g >= 0, g, big).min(axis=(0, 2))
        hi = g.max(axis=(0, 2))
        wins[gi] = np.where(hi >= 0, hi - lo + 1, 0)
    largest = wins.max(0)
    total = wins.sum(0)
    pct = lambda a, p: int(np.percentile(a, p))
    fit = lambda f: float((largest <= f * T).mean())
    return (f"largest run median {pct(largest, 50)} p90 {pct(largest, 90)} p99 {pct(largest, 99)} max {int(largest.max())} | "
            f"sum of 3 runs median {pct(total, 50)} p90 {pct(total, 90)} p99 {pct(total, 99)} max {int(total.max())} | "
            f"tiles whose 3 runs each fit 1.5x {fit(1.5):.3f} 2x {fit(2):.3f} 3x {fit(3):.3f}")


def main():
    print(f"# frames 0..{B - 1} as one batch; rows per level, mean valid neighbours per row")
    for lvl, idx, shape in level_indices():
        for name in ("zyx", "yxz", "brick"):
            nbr = neighbour_rows(idx, shape, name)
            if name == "zyx":
                print(f"level {lvl} shape {shape} rows {idx.shape[0]} neighbours/row {float((nbr >= 0).sum()) / idx.shape[0]:.2f}")
            for T in (64, 128):
                print(f"  level {lvl} order {name:5s} tile {T:3d}: {stats(nbr, name, T)}")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
