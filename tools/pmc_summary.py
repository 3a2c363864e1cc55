#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc results (rocpd sqlite) per kernel name: sum of each counter / dispatch count."""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows = c.execute("select * from counters_collection").fetchall()
ci = {n: i for i, n in enumerate(cols)}
name_col = "kernel_name" if "kernel_name" in ci else [n for n in cols if "name" in n and "kernel" in n][0]
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for r in rows:
    k = r[ci[name_col]].replace("void ", "").replace("(anonymous namespace)::", "").replace("HIP_vector_type", "vec")[:96]
    agg[k][r[ci["counter_name"]]] += float(r[ci["value"]])
    cnt[k].add(r[ci["dispatch_id"]])
names = sorted({n for v in agg.values() for n in v})
derived = "SQ_VALU_MFMA_BUSY_CYCLES" in names and "SQ_BUSY_CYCLES" in names
# MFMA busy %: SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD, SQ_BUSY_CYCLES per shader engine -- the ratio is normalised so that a
# kernel issuing MFMAs back to back on every SIMD of every CU it occupies reads 100 (guide: MI355X_MICROARCH.md, SQ counters)
print("kernel".ljust(96), "disp", *[n[:22].rjust(22) for n in names], *(["mfma_busy/wave_cycles %".rjust(24), "wait/wave_cycles %".rjust(20)] if derived else []))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    d = max(len(cnt[k]), 1)
    extra = []
    if derived:
        wc = max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        extra = [f"{100.0 * v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / wc:24.1f}", f"{100.0 * v.get('SQ_WAIT_ANY', 0.0) / wc:20.1f}"]
    print(k.ljust(96), f"{d:4d}", *[f"{v.get(n, 0) / d:22.1f}" for n in names], *extra)
