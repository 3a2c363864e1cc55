#!/usr/bin/env python3
"""Combine the FETCH_SIZE pass and the WRITE_SIZE pass of `rocprofv3 --pmc` (two rocpd sqlite DBs) into
profiles/*_pmc_traffic.json: HBM bytes per launch and kernel, FETCH_SIZE doubled per MI355X_MICROARCH.md
(gfx950 reports half of wide coalesced reads), both counters in KB.
Usage: pmc_traffic.py <fetch.db> <write.db> > traffic.json"""
import json, re, sqlite3, sys
from collections import defaultdict


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    ci = {n: i for i, n in enumerate(cols)}
    name_col = "kernel_name" if "kernel_name" in ci else [n for n in cols if "name" in n and "kernel" in n][0]
    tot, disp = defaultdict(float), defaultdict(set)
    for r in c.execute("select * from counters_collection"):
        if r[ci["counter_name"]] != counter:
            continue
        k = short(r[ci[name_col]])
        tot[k] += float(r[ci["value"]])
        disp[k].add(r[ci["dispatch_id"]])
    return {k: (tot[k] / max(len(disp[k]), 1), len(disp[k])) for k in tot}


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|^void ", "", name)
    m = re.match(r"(gather_gemm_kernel)<(\d+),", name)
    if m:
        return f"{m.group(1)}<NB={m.group(2)}>"        # the key bench.py's roofline uses
    m = re.match(r"(wgrad_kernel<\d+, \d+>)", name)
    if m:
        return m.group(1)
    m = re.match(r"(subm_win_kernel|subm_wgrad_win_kernel|cm_win_plan_kernel|win_plan_kernel)<WinCfg<(\d+),", name)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"            # (channels: one entry per instantiation)
    m = re.match(r"(gather_gemm_cls_kernel|ggw_kernel)<(\d+),", name)
    if m:
        return f"{m.group(1)}<NB={m.group(2)}>"
    return re.split(r"[<(]", name)[0]


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (bench.py --mode eager --steps 2 "
               "--warmup 2 --no-cpu-baseline --no-roofline); KB units; FETCH_SIZE doubled per MI355X_MICROARCH.md "
               "(gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE uncorrected; averages over all launches of "
               "the kernel name", "kernels": {}}
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 0))[0] + write.get(k, (0, 0))[0])):
    f, nf = fetch.get(k, (0.0, 0))
    w, _ = write.get(k, (0.0, 0))
    out["kernels"][k] = {"launches_sampled": nf, "FETCH_SIZE_KB_per_launch": round(f, 1),
                         "WRITE_SIZE_KB_per_launch": round(w, 1),
                         "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
print(json.dumps(out, indent=1))
