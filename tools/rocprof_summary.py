#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel --stats style text table that is
committed under profiles/.  Usage: rocprof_summary.py <results.db> [steps] > profiles/<name>.txt"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    # durations in the view are nanoseconds in some rocprofv3 builds and microseconds in others: use the
    # raw dispatch table to be unambiguous
    disp = c.execute("select min(start), max(end), sum(end - start), count(*) from kernels").fetchone()
    unit = 1e-3  # ns -> us
    tot_view = sum(r[2] for r in rows)
    scale = (disp[2] / tot_view) if tot_view else 1.0
    print(f"# rocprofv3 --kernel-trace --stats summary of {db}")
    print(f"# dispatches {disp[3]}, total kernel time {disp[2] * 1e-6:.3f} ms, wall span {(disp[1] - disp[0]) * 1e-6:.3f} ms"
          + (f", steps {steps:g} -> {disp[2] * 1e-6 / steps:.3f} ms kernel time / step" if steps else ""))
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}  name")
    for name, calls, total, avg, pct in rows:
        print(f"{calls:7d} {total * scale * unit:12.1f} {total * scale * unit / calls:10.2f} {pct:6.2f}  {name[:150]}")


if __name__ == "__main__":
    main()
