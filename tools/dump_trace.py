#!/usr/bin/env python3
"""Dump the last N kernel dispatches of a rocprofv3 rocpd sqlite trace: queue, start (us), duration (us), name."""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
ci = {k: i for i, k in enumerate(cols)}
rows = c.execute("select * from kernels order by start").fetchall()[-n:]
name = "name" if "name" in ci else "kernel_name"
t0 = rows[0][ci["start"]]
for r in rows:
    nm = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r[ci[name]])[:60]
    print(f"{r[ci['queue_id']]} {1e-3 * (r[ci['start']] - t0):9.1f} {1e-3 * (r[ci['end']] - r[ci['start']]):8.1f} {nm}")
