"""Kernels of a rocprofv3 kernel trace (rocpd .db) that run on FEW workgroups and still take long: serial tails hiding inside a
launch or a chain (round 6: one such block was the whole duration of the strided builds' emit launch).
usage: python tools/small_grids.py <results.db> [max_workgroups=16] [min_us=4]"""
import sqlite3, sys, collections


def main():
    db = sys.argv[1]
    max_wg = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    gx = "grid_size_x" if "grid_size_x" in cols else "grid_x"
    wx = "workgroup_size_x" if "workgroup_size_x" in cols else "workgroup_x"
    rows = c.execute(f"select s.kernel_name, d.{gx}, d.{wx}, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
    agg = collections.defaultdict(list)
    for name, g, w, dur in rows:
        nwg = (g + w - 1) // max(w, 1)
        agg[(name[:90], nwg)].append(dur / 1e3)
    out = [(sum(v) / len(v), len(v), k) for k, v in agg.items() if k[1] <= max_wg and sum(v) / len(v) >= min_us]
    for avg, n, (name, nwg) in sorted(out, reverse=True):
        print(f"{avg:9.1f} us  x{n:5d}  {nwg:4d} wg  {name}")


main()
