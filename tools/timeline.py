#!/usr/bin/env python3
"""Per-queue busy/idle analysis of the LAST step in a rocprofv3 kernel trace (rocpd sqlite)."""
import sqlite3, sys
from collections import defaultdict
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select * from kernels order by start").fetchall()
ci = {n: i for i, n in enumerate(cols)}
qcol = "queue_id" if "queue_id" in ci else ("stream_id" if "stream_id" in ci else None)
name = "name" if "name" in ci else "kernel_name"
# find the last occurrence of vox_insert (start of a step) 
starts = [r[ci["start"]] for r in rows if "vox_insert" in r[ci[name]]]
t0, t1 = starts[-2], starts[-1]
step = [r for r in rows if t0 <= r[ci["start"]] < t1]
print(f"step wall {1e-6 * (t1 - t0):.3f} ms, {len(step)} kernels; queue column: {qcol}")
byq = defaultdict(list)
for r in step:
    byq[r[ci[qcol]] if qcol else 0].append(r)
for q, rs in byq.items():
    busy = sum(r[ci["end"]] - r[ci["start"]] for r in rs)
    print(f" queue {q}: {len(rs)} kernels, busy {busy * 1e-6:.3f} ms")
# union busy time over all queues and idle gaps
ev = sorted((r[ci["start"]], r[ci["end"]]) for r in step)
cur_s, cur_e = ev[0]; union = 0; gaps = []
for s, e in ev[1:]:
    if s > cur_e:
        union += cur_e - cur_s; gaps.append((s - cur_e, cur_e)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f" GPU busy (union) {union * 1e-6:.3f} ms, idle {(t1 - t0 - union) * 1e-6:.3f} ms in {len(gaps)} gaps; "
      f"gaps > 5us: {sum(1 for g, _ in gaps if g > 5000)} totalling {sum(g for g, _ in gaps if g > 5000) * 1e-6:.3f} ms")
big = sorted(gaps, reverse=True)[:8]
for g, at in big:
    prev = max((r for r in step if r[ci['end']] <= at + 1), key=lambda r: r[ci['end']])
    print(f"   gap {g * 1e-3:.1f} us after {prev[ci[name]][:60]}")
if len(sys.argv) > 2:  # dump the step's kernel sequence: queue, start offset (us), duration (us), short name
    import re
    with open(sys.argv[2], "w") as f:
        for r in step:
            nm = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r[ci[name]])[:70]
            f.write(f"{r[ci[qcol]] if qcol else 0} {1e-3 * (r[ci['start']] - t0):9.1f} {1e-3 * (r[ci['end']] - r[ci['start']]):8.1f} {nm}\n")
