"""Window gather-GEMM (spconv_win.hip) against the generic kernels on the SubM layers of the B = 4 batch, rows numbered
z-fastest (PCD_ROWS_YXZ): max difference (both are bf16-rounded fp32 sums in different orders) and time per launch.
usage: python tools/exp_subm_win.py [levels e.g. 23]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from com_amd import ops, hotpath
sys.path.insert(0, 'tools')
import env_switches
env_switches.apply()          # PCD_OPT_* -> pcd_set_option
from com_amd.utils import synth

dev = torch.device("cuda")
B = 4
frames = [synth.synth_cloud(f) for f in range(B)]
pts, offs = hotpath.collate_points(frames, dev)
res = ops.voxelize_hard(pts, offs, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000, feat_offset=1, num_features=5,
                        want_voxels=False, row_order="yxz", key_depth=41)
idx, rank, shape = res["coords"], res["rank"], [41, 1504, 1504]
levels = {1: (idx, rank, shape, 16)}
chain = [((3, 3, 3), (2, 2, 2), (1, 1, 1), 32), ((3, 3, 3), (2, 2, 2), (1, 1, 1), 64), ((3, 3, 3), (2, 2, 2), (0, 1, 1), 128)]
lvl = 1
for k, s, p, ch in chain:
    rbc = ops.rulebook_conv(idx, B, shape, k, s, p, want_pairs=False, order=ops.ROWS_YXZ)
    idx, rank, shape = rbc.out_indices, rbc.rank, rbc.out_shape
    lvl += 1
    levels[lvl] = (idx, rank, shape, ch)


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters * 1e3


_scratch = None


def timeit_cold(fn, iters=12, mb=1024):
    """One launch at a time behind a `mb`-MB fill (L2 and the 256 MB MALL hold nothing of the conv's inputs, plan or weights):
    what a conv of the training step sees -- its input was written two launches ago, its plan a millisecond ago."""
    global _scratch
    if _scratch is None or _scratch.numel() != mb << 20:
        _scratch = torch.empty((mb << 20,), dtype=torch.uint8, device=dev)
    tot = 0.0
    for i in range(iters + 2):
        _scratch.fill_(i & 255)
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        fn()
        t1.record()
        torch.cuda.synchronize()
        if i >= 2:
            tot += t0.elapsed_time(t1)
    return tot / iters * 1e3


want = sys.argv[1] if len(sys.argv) > 1 else "23"
for lvl in (1, 2, 3, 4):
    if str(lvl) not in want:
        continue
    idx, rank, shape, ch = levels[lvl]
    n = idx.shape[0]
    rb = ops.rulebook_subm(idx, B, shape, rank=rank, want_pairs=False)
    g = torch.Generator(device="cpu").manual_seed(5)
    w = (torch.randn(ch, 3, 3, 3, ch, generator=g) * (1.0 / np.sqrt(27 * ch))).to(dev)
    bias = (torch.randn(ch, generator=g) * 0.1).to(dev)
    x = torch.randn(n, ch, generator=g).to(dev).to(torch.bfloat16)
    add = torch.randn(n, ch, generator=g).to(dev).to(torch.bfloat16)
    pf, pd = ops.pack_weight(w, 0), ops.pack_weight(w, 1)
    wf, wd = ops.pack_weight_window(w, 0), ops.pack_weight_window(w, 1)
    pairs = int((rb.nbr_out >= 0).sum().item())
    T = ops.subm_window_tile_rows(ch, ch)
    nt = (n + T - 1) // T
    off = 512 * 64 + 2048 + (nt * 4 + 31) // 32 * 32              # (spconv_win.hip: win_hdr_off)
    pl = ops.subm_window_plan(rb, ch, ch)
    hdr = pl[off:off + nt * 32].view(torch.int32).view(nt, 8).cpu().numpy()
    ent = pl[:256 * 64].view(torch.int32).view(256, 16).cpu().numpy()        # (the 8-wave configurations' 256 entries)
    share = (ent[:, 1] - ent[:, 0])[:256 // (4 if ch == 128 else 1)]
    print(f"level {lvl}: shares of the workgroups: tiles min {share.min()} median {int(np.median(share))} max {share.max()}")
    runs = hdr[:, [1, 3, 5]]
    print(f"level {lvl}: {nt} tiles of {T} rows; passes histogram {np.bincount(hdr[:, 6]).tolist()}; run length median "
          f"{int(np.median(runs))} p90 {int(np.percentile(runs, 90))} p99 {int(np.percentile(runs, 99))} max {int(runs.max())}; "
          f"multi-pass tiles: {np.nonzero(hdr[:, 6] > 1)[0][:24].tolist()}")
    if os.environ.get("WIN_HDR"):
        ic = idx.cpu().numpy()
        for tt in np.nonzero(hdr[:, 6] > 1)[0][:6]:
            rows = ic[tt * T:(tt + 1) * T]
            print(f"   tile {tt}: header {hdr[tt].tolist()}; rows {tt * T}..: b {rows[:, 0].min()}-{rows[:, 0].max()} "
                  f"z {rows[:, 1].min()}-{rows[:, 1].max()} y {rows[:, 2].min()}-{rows[:, 2].max()} x {rows[:, 3].min()}-{rows[:, 3].max()}")
    for name, flip, pk, pw, b_, a_ in (("fwd", False, pf, wf, bias, None), ("dgrad+addend", True, pd, wd, None, add)):
        y0 = ops.gather_gemm(x, pk, b_, rb.nbr_out, 27, flip, n, ch, torch.bfloat16, addend=a_)
        y1 = ops.subm_window(x, pw, b_, rb, ch, addend=a_)
        torch.cuda.synchronize()
        d = (y0.float() - y1.float()).abs()
        scale = y0.float().abs().max().item()
        t_gen = timeit(lambda: ops.gather_gemm(x, pk, b_, rb.nbr_out, 27, flip, n, ch, torch.bfloat16, addend=a_))
        t_win = timeit(lambda: ops.subm_window(x, pw, b_, rb, ch, addend=a_))
        if os.environ.get("WIN_COLD"):
            t_cold = timeit_cold(lambda: ops.subm_window(x, pw, b_, rb, ch, addend=a_))
            t_empty = timeit_cold(lambda: None)
            print(f"   cold caches (one launch behind a 1 GB fill): window {t_cold:.1f} us (event pair alone: {t_empty:.1f} us); warm loop {t_win:.1f} us")
        fl = 2.0 * pairs * ch * ch
        print(f"level {lvl} {ch}->{ch} rows {n} pairs {pairs} {name}: max|diff| {d.max().item():.4g} (scale {scale:.3g}, "
              f"mismatching elements {(d > 0).float().mean().item():.4f}) generic {t_gen:.1f} us, window {t_win:.1f} us "
              f"({fl / t_win * 1e-6:.0f} TFLOP/s algorithmic = {fl / t_win * 1e-6 / 2500:.3f} of the bf16 peak)", flush=True)
    # BatchNorm sums in the epilogue
    st0, st1 = ops.BnReduce(1), ops.BnReduce(1)
    ops.BN_FUSED_MID = False
    y0 = ops.gather_gemm(x, pf, bias, rb.nbr_out, 27, False, n, ch, torch.bfloat16, bn_reduce=st0)
    y1 = ops.subm_window(x, wf, bias, rb, ch, bn_reduce=st1)
    torch.cuda.synchronize()
    s0, s1 = st0.partial.double().sum(0), st1.partial.double().sum(0)
    ref = torch.stack([y1.double().sum(0), (y1.double() ** 2).sum(0)])
    print(f"   BN sums (mode 1): window vs its own output {((s1 - ref).abs() / ref.abs().clamp_min(1)).max().item():.3g}, "
          f"generic vs window {((s0 - s1).abs() / s1.abs().clamp_min(1)).max().item():.3g}")
    ops.BN_FUSED_MID = True

if os.environ.get("WIN_TRACE"):
    from com_amd import _lib as L
    idx, rank, shape, ch = levels[int(os.environ["WIN_TRACE"])]
    n = idx.shape[0]
    rb = ops.rulebook_subm(idx, B, shape, rank=rank, want_pairs=False)
    w = (torch.randn(ch, 3, 3, 3, ch) * 0.02).to(dev)
    x = torch.randn(n, ch).to(dev).to(torch.bfloat16)
    wf = ops.pack_weight_window(w, 0)
    tr = torch.zeros(1024, dtype=torch.int64, device=dev)
    for _ in range(3):
        ops.subm_window(x, wf, None, rb, ch)
    L.lib().pcd_subm_window_set_trace(L.ptr(tr))
    ops.subm_window(x, wf, None, rb, ch)
    torch.cuda.synchronize()
    L.lib().pcd_subm_window_set_trace(None)
    t = tr.cpu().numpy()
    t = t[t > 0]
    names = ["B1", "prefetch issued", "MFMA done", "prefetch landed", "B2", "sums written+B3", "epilogue"]
    print(f"workgroup 0: entry -> weights + first window landed {int(t[1] - t[0])} clk; entry -> exit {int(t[-1] - t[0])} clk; "
          f"last tile epilogue -> exit {int(t[-1] - t[-2])} clk")
    t = t[2:-1]
    print("shader clocks between stamps (rows = tiles):")
    print("   " + " | ".join(names[1:]) + " | -> next B1")
    for i in range(0, len(t) - 6, 7):
        d = np.diff(t[i:i + 8])
        print("   " + " ".join(f"{int(v):7d}" for v in d))

if os.environ.get("WIN_WGRAD"):
    for lvl in (1, 2, 3):
        if str(lvl) not in want:
            continue
        idx, rank, shape, ch = levels[lvl]
        n = idx.shape[0]
        rb = ops.rulebook_subm(idx, B, shape, rank=rank, want_pairs=True)
        x = torch.randn(n, ch).to(dev).to(torch.bfloat16)
        dy = torch.randn(n, ch).to(dev).to(torch.bfloat16)
        d0 = ops.wgrad(x, ch, dy, None, None, 27, rb=rb)
        d1 = ops.subm_window_wgrad(x, dy, rb)
        torch.cuda.synchronize()
        jobs = []
        t_gen = timeit(lambda: ops.wgrad(x, ch, dy, None, None, 27, rb=rb, defer=jobs))
        t_win = timeit(lambda: ops.subm_window_wgrad(x, dy, rb, defer=jobs))
        t_red0 = timeit(lambda: ops.wgrad(x, ch, dy, None, None, 27, rb=rb)) - t_gen
        t_red1 = timeit(lambda: ops.subm_window_wgrad(x, dy, rb)) - t_win
        if os.environ.get("WIN_WGRAD") == "2":
            from com_amd import _lib as L
            tr = torch.zeros(1024, dtype=torch.int64, device=dev)
            L.lib().pcd_subm_window_set_trace(L.ptr(tr))
            ops.subm_window_wgrad(x, dy, rb, defer=jobs)
            torch.cuda.synchronize()
            L.lib().pcd_subm_window_set_trace(None)
            tt = tr.cpu().numpy()
            ent, ext = tt[256:512], tt[512:768]
            live = ext > 0
            t0 = ent[live].min()
            dur = (ext - ent)[live] * 10e-3
            print(f"   per workgroup (us): start after the first {np.percentile((ent[live] - t0) * 10e-3, [50, 100]).round(1).tolist()} "
                  f"(median, max); duration min {dur.min():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f}; "
                  f"end of the last {((ext[live] - t0) * 10e-3).max():.1f}; longest: block {np.nonzero(live)[0][dur.argmax()]}")
            tt = tt[:256]
            tt = tt[tt > 0]
            print(f"   wgrad workgroup 0: entry -> first tile {int(tt[1] - tt[0])} clk, entry -> exit {int(tt[-1] - tt[0])} clk, "
                  f"last barrier -> exit {int(tt[-1] - tt[-2])}; per tile (issue | compute | wait | barrier):")
            body = tt[1:-2]
            for i in range(0, min(len(body) - 4, 48), 4):
                d = np.diff(body[i:i + 5])
                print("      " + " ".join(f"{int(v):6d}" for v in d))
        print(f"level {lvl} wgrad {ch}x{ch}: max|diff| {(d0 - d1).abs().max().item():.4g} (scale {d0.abs().max().item():.3g}); "
              f"generic {t_gen:.1f} us (+ reduce {t_red0:.1f}), window {t_win:.1f} us (+ reduce {t_red1:.1f})", flush=True)

if os.environ.get("WIN_CONTEND"):
    # how a window launch starts while another stream keeps the chip busy: per-workgroup entry / exit times
    from com_amd import _lib as L
    lvl = int(os.environ["WIN_CONTEND"])
    idx, rank, shape, ch = levels[lvl]
    n = idx.shape[0]
    rb = ops.rulebook_subm(idx, B, shape, rank=rank, want_pairs=False)
    w = (torch.randn(ch, 3, 3, 3, ch) * 0.02).to(dev)
    x = torch.randn(n, ch).to(dev).to(torch.bfloat16)
    wf = ops.pack_weight_window(w, 0)
    # the competitor: the pair-based weight gradient of the 128-channel level (what runs beside the level-1 dgrads in the step)
    i4, r4, s4, c4 = levels[4]
    rb4 = ops.rulebook_subm(i4, B, s4, rank=r4, want_pairs=True)
    x4 = torch.randn(i4.shape[0], c4).to(dev).to(torch.bfloat16)
    d4 = torch.randn(i4.shape[0], c4).to(dev).to(torch.bfloat16)
    side = torch.cuda.Stream()
    tr = torch.zeros(1024, dtype=torch.int64, device=dev)
    for contend in (False, True):
        for _ in range(3):
            ops.subm_window(x, wf, None, rb, ch)
        torch.cuda.synchronize()
        L.lib().pcd_subm_window_set_trace(L.ptr(tr))
        if contend:
            with torch.cuda.stream(side):
                jobs = []
                for _ in range(4):
                    ops.wgrad(x4, c4, d4, None, None, 27, rb=rb4, defer=jobs)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        ops.subm_window(x, wf, None, rb, ch)
        t1.record()
        torch.cuda.synchronize()
        L.lib().pcd_subm_window_set_trace(None)
        tt = tr.cpu().numpy()
        ent, ext = tt[256:512], tt[512:768]
        live = ext > 0
        tz = ent[live].min()
        st = (ent[live] - tz) * 10e-3
        dur = (ext - ent)[live] * 10e-3
        print(f"level {lvl} window launch {'beside 4 wgrad128 launches' if contend else 'alone'}: {t0.elapsed_time(t1) * 1e3:.1f} us; "
              f"workgroup start after the first: median {np.median(st):.1f} p90 {np.percentile(st, 90):.1f} max {st.max():.1f} us; "
              f"duration median {np.median(dur):.1f} max {dur.max():.1f}; last exit {((ext[live] - tz) * 10e-3).max():.1f} us")
        tr.zero_()

if os.environ.get("WIN_BALANCE"):
    # per-workgroup duration against its share: least squares  dur = a + b tiles + c extra passes
    from com_amd import _lib as L
    for lvl in (1, 2, 3):
        idx, rank, shape, ch = levels[lvl]
        n = idx.shape[0]
        rb = ops.rulebook_subm(idx, B, shape, rank=rank, want_pairs=False)
        w = (torch.randn(ch, 3, 3, 3, ch) * 0.02).to(dev)
        x = torch.randn(n, ch).to(dev).to(torch.bfloat16)
        wf = ops.pack_weight_window(w, 0)
        T = ops.subm_window_tile_rows(ch, ch)
        nt = (n + T - 1) // T
        pl = ops.subm_window_plan(rb, ch, ch)
        off = 512 * 64 + 2048 + (nt * 4 + 31) // 32 * 32
        hdr = pl[off:off + nt * 32].view(torch.int32).view(nt, 8).cpu().numpy()
        ent_tab = pl[:256 * 64].view(torch.int32).view(256, 16).cpu().numpy()
        tr = torch.zeros(1024, dtype=torch.int64, device=dev)
        for _ in range(3):
            ops.subm_window(x, wf, None, rb, ch)
        torch.cuda.synchronize()
        L.lib().pcd_subm_window_set_trace(L.ptr(tr))
        ops.subm_window(x, wf, None, rb, ch)
        torch.cuda.synchronize()
        L.lib().pcd_subm_window_set_trace(None)
        tt = tr.cpu().numpy()
        dur = (tt[512:768] - tt[256:512]) * 10e-3
        # workgroup b -> XCD-major index
        rowsA = []
        for b in range(256):
            wxm = (b & 7) * 32 + (b >> 3)
            tb, te = ent_tab[wxm, 0], ent_tab[wxm, 1]
            extra = int((hdr[tb:te, 6] - 1).sum())
            rowsA.append((1.0, te - tb, extra, dur[b]))
        A = np.array(rowsA)
        coef, *_ = np.linalg.lstsq(A[:, :3], A[:, 3], rcond=None)
        print(f"level {lvl}: dur = {coef[0]:.2f} + {coef[1]:.2f} tiles + {coef[2]:.2f} extra passes (us); extra pass = {coef[2] / coef[1]:.2f} tiles; "
              f"dur median {np.median(dur):.1f} max {dur.max():.1f}; tiles {A[:, 1].min():.0f}-{A[:, 1].max():.0f}, extra 0-{A[:, 2].max():.0f}")
