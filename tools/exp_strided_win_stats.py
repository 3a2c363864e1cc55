"""Window-fit statistics of the four STRIDED convs of VoxelResBackBone8x in z-fastest (b, y, x, z) row order (CPU only;
VERDICT r4 item 1 step 0; the SubM counterpart is tools/exp_win_stats.py).

Forward (output-stationary): a tile of T consecutive OUTPUT rows; its <= K offsets are grouped into runs by the kernel
index along y (ky): run ky holds the input rows of BEV row  oy * sy - py + ky  over the tile's x-range, a contiguous
stretch of the input level's rows.  Data gradient (input-stationary): a tile of T consecutive INPUT rows; run ky holds
the output rows of BEV row (y + py - ky) / sy (where the parity admits one).  A run's window is max - min + 1 over the
rows its offsets touch.  Reported per (conv, direction, T): window rows median / p90 / p99 / max of the LARGEST run and
of the SUM of the runs, the share of tiles whose runs each fit 1x / 1.5x / 2x / 3x the tile, the single window that
covers ALL offsets of the tile (adjacent output BEV rows read input BEV rows two apart: the runs of a tile that crosses
a BEV row boundary are not contiguous, their union is), and the pairs per tile row (the gathers the window replaces).

usage: python tools/exp_strided_win_stats.py > profiles/r05_strided_win_stats.txt
"""
import sys
import numpy as np
sys.path.insert(0, '.')
from oracle import oracle as O
from com_amd.utils import synth

B = 4
CONVS = (("spconv2 16->32", (3, 3, 3), (2, 2, 2), (1, 1, 1)),
         ("spconv3 32->64", (3, 3, 3), (2, 2, 2), (1, 1, 1)),
         ("spconv4 64->128", (3, 3, 3), (2, 2, 2), (0, 1, 1)),
         ("spconv_down2 128->128", (3, 1, 1), (2, 1, 1), (0, 0, 0)))


def yxz_perm(idx, shape):
    """permutation old row -> rank in (b, y, x, z) order, and its inverse"""
    D, H, W = shape
    b, z, y, x = (idx[:, i].astype(np.int64) for i in range(4))
    key = ((b * H + y) * W + x) * 64 + z
    order = np.argsort(key, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.shape[0])
    return order, rank


def renumber(tbl, rank_rows, order_cols):
    """tbl[k][col] = row (or -1): columns reordered by order_cols, entries mapped through rank_rows"""
    t = tbl[:, order_cols]
    return np.where(t >= 0, rank_rows[np.maximum(t, 0)], -1)


def stats(tbl, ks, T, group_of_k):
    n = tbl.shape[1]
    nt = n // T
    big = np.iinfo(np.int64).max
    groups = sorted(set(group_of_k))
    wins = np.zeros((len(groups), nt), np.int64)
    for gi, g in enumerate(groups):
        sel = [k for k in range(tbl.shape[0]) if group_of_k[k] == g]
        a = tbl[sel][:, :nt * T].reshape(len(sel), nt, T)
        lo = np.where(a >= 0, a, big).min(axis=(0, 2))
        hi = a.max(axis=(0, 2))
        wins[gi] = np.where(hi >= 0, hi - lo + 1, 0)
    largest, total = wins.max(0), wins.sum(0)
    allk = tbl[:, :nt * T].reshape(tbl.shape[0], nt, T)
    union = np.where(allk.max(axis=(0, 2)) >= 0, allk.max(axis=(0, 2)) - np.where(allk >= 0, allk, big).min(axis=(0, 2)) + 1, 0)
    touched = float((tbl[:, :nt * T] >= 0).sum()) / (nt * T)
    pct = lambda a, p: int(np.percentile(a, p))
    fit = lambda f: float((largest <= f * T).mean())
    return (f"runs {len(groups)} | largest run median {pct(largest, 50)} p90 {pct(largest, 90)} p99 {pct(largest, 99)} max {int(largest.max())}"
            f" | sum of runs median {pct(total, 50)} p90 {pct(total, 90)} p99 {pct(total, 99)} max {int(total.max())}"
            f" | tiles whose runs each fit 1x {fit(1):.3f} 1.5x {fit(1.5):.3f} 2x {fit(2):.3f} 3x {fit(3):.3f}"
            f" | ONE window over all offsets median {pct(union, 50)} p90 {pct(union, 90)} p99 {pct(union, 99)} max {int(union.max())}"
            f" | pairs per tile row {touched:.2f}")


def main():
    rows = []
    for f in range(B):
        pts = synth.synth_cloud(f)
        _, c, _ = O.voxelize_hard(pts, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000)
        rows.append(np.concatenate([np.full((c.shape[0], 1), f, np.int32), c], 1))
    idx = np.ascontiguousarray(np.concatenate(rows, 0))
    shape = (41, 1504, 1504)
    print(f"# frames 0..{B - 1} as one batch; rows of both levels numbered by (b, y, x, z); runs = offsets grouped by ky")
    for name, ks, st, pd in CONVS:
        rc = O.rulebook_conv(idx, shape, ks, st, pd)
        oidx, oshape = rc["out_indices"], tuple(int(v) for v in rc["out_shape"])
        in_order, in_rank = yxz_perm(idx, shape)
        out_order, out_rank = yxz_perm(oidx, oshape)
        nbr_out = renumber(rc["nbr_out"].astype(np.int64), in_rank, out_order)    # [K][n_out] -> input row
        nbr_in = renumber(rc["nbr_in"].astype(np.int64), out_rank, in_order)      # [K][n_in]  -> output row
        K = ks[0] * ks[1] * ks[2]
        ky_of = [(k // ks[2]) % ks[1] for k in range(K)]
        print(f"{name}: in {shape} rows {idx.shape[0]} -> out {oshape} rows {oidx.shape[0]}, pairs {int((nbr_out >= 0).sum())}")
        for T in (32, 64, 128):
            print(f"  forward  (tiles of {T:3d} output rows): {stats(nbr_out, ks, T, ky_of)}")
        for T in (32, 64, 128):
            print(f"  dgrad    (tiles of {T:3d} input rows):  {stats(nbr_in, ks, T, ky_of)}")
        sys.stdout.flush()
        idx, shape = oidx, oshape


if __name__ == "__main__":
    main()
