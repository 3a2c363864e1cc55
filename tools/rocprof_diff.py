#!/usr/bin/env python3
"""Per-REPLAY kernel statistics of the captured step: the difference of two rocprofv3 kernel traces of the same command that
differ only in the number of timed steps (the eager warm-up steps, the capture's own warm-up and every one-off launch cancel).
Usage: rocprof_diff.py <few_steps.db> <many_steps.db> <extra_steps> > profiles/<name>.txt"""
import collections
import sqlite3
import sys


def load(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kt = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    st = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute(f"select s.kernel_name, count(*), sum(d.end - d.start) from {kt} d join {st} s on d.kernel_id = s.id "
                     f"group by s.kernel_name").fetchall()
    return {n: (cnt, tot) for n, cnt, tot in rows}


def main():
    a, b, extra = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3])
    rows = []
    for name in b:
        ca, ta = a.get(name, (0, 0))
        cb, tb = b[name]
        if cb - ca > 0:
            rows.append((name, (cb - ca) / extra, (tb - ta) * 1e-3 / extra, (tb - ta) * 1e-3 / (cb - ca)))
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    print(f"# per-replay kernel statistics: ({sys.argv[2]}) - ({sys.argv[1]}) over {extra:g} extra replayed steps")
    print(f"# launches per step {sum(r[1] for r in rows):.1f}, kernel time per step {tot * 1e-3:.3f} ms")
    print(f"{'per_step':>9} {'us_per_step':>12} {'avg_us':>9} {'pct':>6}  name")
    for name, n, us, avg in rows:
        print(f"{n:9.2f} {us:12.1f} {avg:9.2f} {100 * us / tot:6.2f}  {name[:150]}")


if __name__ == "__main__":
    main()
