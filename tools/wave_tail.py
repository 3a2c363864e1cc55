"""Tail / imbalance suspects of a step: per kernel, the average wave lifetime (SQ_WAVE_CYCLES / SQ_WAVES) against the launch's
duration (GRBM_GUI_ACTIVE, same --pmc pass).  A launch that lasts many times its average wave is waiting for few long waves
(round 6: the strided builds' emit launch, 4.5 k waves of 3 us in a launch of 87 us -- one workgroup scanning the parity classes).
usage: rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d DIR -o r -- python3 bench.py --steps 4 --warmup 2 --light
       python tools/wave_tail.py DIR/.../r_results.db"""
import sqlite3, sys, collections, re


def main():
    c = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    ip = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute(f"select s.kernel_name, d.id, p.name, e.value from {pe} e join {ip} p on e.pmc_id = p.id "
                     f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id").fetchall()
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    name_of = {}
    for name, did, cname, v in rows:
        per[did][cname] += v
        name_of[did] = name
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for did, cs in per.items():
        n = re.sub(r"\(.*", "", name_of[did])[:70]
        a = agg[n]
        a[0] += 1; a[1] += cs.get("SQ_WAVES", 0); a[2] += cs.get("SQ_WAVE_CYCLES", 0); a[3] += cs.get("GRBM_GUI_ACTIVE", 0)
    out = []
    for n, (k, w, wc, gui) in agg.items():
        if w <= 0 or gui <= 0:
            continue
        out.append((gui / k, n, k, w / k, wc / w, (gui / k) / (wc / w)))
    print(f"{'launch cycles':>14} {'waves':>9} {'wave cycles':>12} {'launch/wave':>12}  kernel (launches)")
    for gui, n, k, w, wl, ratio in sorted(out, reverse=True)[:60]:
        print(f"{gui:14.0f} {w:9.0f} {wl:12.0f} {ratio:12.1f}  {n} ({k})")


main()
