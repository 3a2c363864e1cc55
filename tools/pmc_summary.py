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
    k = r[ci[name_col]][:70]
    agg[k][r[ci["counter_name"]]] += float(r[ci["value"]])
    cnt[k].add(r[ci["dispatch_id"]])
names = sorted({n for v in agg.values() for n in v})
print("kernel".ljust(70), "disp", *[n[:22].rjust(22) for n in names])
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    d = max(len(cnt[k]), 1)
    print(k.ljust(70), f"{d:4d}", *[f"{v.get(n, 0) / d:22.1f}" for n in names])
