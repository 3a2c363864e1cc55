"""Bandwidth-regime measurements asked for by SURVEY.md section 8(d): the rulebook builds (and the whole
forward+backward step) at B = 4 (reference batch) and B = 32 frames per GPU, with the algorithmic bytes of
section 8(d) (rulebook: 16*N_in + 8*P, + 16*N_out for strided convs) over the measured time vs 8 TB/s.
Usage: python tools/regime.py [B ...]   -> one JSON line per batch size."""
import json, os, sys, time, torch
sys.path.insert(0, '.')
from com_amd import ops, hotpath
from com_amd.utils import synth
dev = 'cuda'
sys.path.insert(0, 'tools')
import env_switches
import contextlib
_plan_scope = contextlib.ExitStack()      # `with plan:` scopes opened / closed around try blocks (com_amd.ops.current_plan)
env_switches.apply()          # (PCD_COLMAP=0: the flat key-space bitmap builds of rounds 1-4; PCD_OPT_*)


def timed_graph(fn, key, n_out, reps=10, inner=8):
    """The same build replayed from a hipGraph with its output capacity known (static plan): what a training step
    executes -- no host round trip for the row count, no Python between the launches.  `inner` builds per graph so
    that the cost of launching the graph itself (which a training step pays once for ~280 kernels) is amortised."""
    plan = ops.StaticPlan()
    plan.observe(key, n_out)
    plan.active = True
    _plan_scope.close(); _plan_scope.enter_context(plan)
    try:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(inner):
                    fn()
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) / (reps * inner) * 1e-3
        del g                                     # free the graph's pool NOW, not inside the next timed region
        import gc; gc.collect(); torch.cuda.synchronize()
        return dt
    finally:
        _plan_scope.close()


def timed(fn, reps=10):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / reps * 1e-3


AS_BUILT_FREE = {int(v) for v in os.environ.get('PCD_REGIME_FREE', '1,2').split(',') if v}    # levels built without a neighbour table
for B in [int(a) for a in sys.argv[1:]] or [4, 32]:
    frames = [synth.synth_cloud(f) for f in range(B)]
    pts, offs = hotpath.collate_points(frames, dev)
    res = ops.voxelize_hard(pts, offs, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000, feat_offset=1,
                            num_features=5, want_voxels=False,
                            row_order=os.environ.get('PCD_ROW_ORDER', 'yxz'), key_depth=41)   # (what bench.py runs)
    ORDER = ops.ROW_ORDERS[os.environ.get('PCD_ROW_ORDER', 'yxz')]
    idx, shape = res['coords'], [41, 1504, 1504]
    rows = []
    tot_bytes = tot_t = tot_g = tot_step = tot_tables = 0.0
    rank = res.get('rank', None)       # key-ordered rows: the voxeliser's coordinate -> row map serves level 1
    geos = [None, (3, 2, 1), (3, 2, 1), (3, 2, (0, 1, 1)), ((3, 1, 1), (2, 1, 1), 0)]
    for lvl, geo in enumerate(geos):
        if geo is not None:
            n_in = idx.shape[0]
            rbc, t = timed(lambda: ops.rulebook_conv(idx, B, shape, geo[0], geo[1], geo[2], order=ORDER, in_rank=rank))
            P = int(rbc.pair_num.sum())
            by = 16 * n_in + 8 * P + 16 * rbc.out_indices.shape[0]
            SKIP_PLAIN = os.environ.get('PCD_REGIME_SKIP_PLAIN') == '1'      # (profiling runs: only the as-built forms in the graphs)
            tg = 1e-9 if SKIP_PLAIN else timed_graph(lambda: ops.rulebook_conv(idx, B, shape, geo[0], geo[1], geo[2], plan_key=("conv", lvl), order=ORDER, in_rank=rank),
                             ("conv", lvl), int(rbc.out_indices.shape[0]))
            # as the training step builds it (round 6): no indice_pairs for the 3 x 3 x 3 strided convs -- their weight gradient
            # reads the pairs off the parity classes (pcd_sparse_conv_wgrad_classes) -- and compact neighbour tables only
            # (pcd_rulebook_conv_cm_build_compact); conv_out (128 x 128) keeps its lists and full tables
            tgs = tg
            if (ops.IMPLICIT_STRIDED_PAIRS and lvl < 4) or SKIP_PLAIN:
                tgs = timed_graph(lambda: ops.rulebook_conv(idx, B, shape, geo[0], geo[1], geo[2], plan_key=("conv", lvl), order=ORDER,
                                                            in_rank=rank, pair_lists=False, compact=ops.COMPACT_STRIDED_TABLES),
                                   ("conv", lvl), int(rbc.out_indices.shape[0]))
            rows.append(dict(kind="strided" + ("_cm" if isinstance(rbc.rank, ops.ColumnMap) else ""), level=lvl + 1, n_in=n_in, n_out=int(rbc.out_indices.shape[0]), pairs=P,
                             us=round(t * 1e6, 1), graph_us=round(tg * 1e6, 1), alg_MB=round(by / 1e6, 1),
                             GBps=round(by / t / 1e9, 1), graph_GBps=round(by / tg / 1e9, 1), as_built_us=round(tgs * 1e6, 1)))
            tot_bytes += by; tot_t += t; tot_g += tg; tot_step += tgs; tot_tables += tg
            idx, shape, rank = rbc.out_indices, rbc.out_shape, rbc.rank
        if lvl < 4:
            n = idx.shape[0]
            rb, t = timed(lambda: ops.rulebook_subm(idx, B, shape, rank=rank))
            P = int(rb.pair_num.sum())
            by = 16 * n + 8 * P
            tg = 1e-9 if os.environ.get('PCD_REGIME_SKIP_PLAIN') == '1' else timed_graph(lambda: ops.rulebook_subm(idx, B, shape, rank=rank), None, 0)
            # as the training step builds it: the 16 / 32-channel levels run their weight gradient over the window tiles and
            # need no pair lists (hotpath/backbone3d.py, spconv/conv.py::_rulebook)
            # ... and their window PLAN comes out of the same pass, straight from the column map (round 6: pcd_subm_window_plan_cm);
            # where every conv of the level runs on window tiles (AS_BUILT_FREE: level 2, and level 1 once conv_input does) no
            # neighbour table is written at all.  (Until round 5 the as-built figure was the table build alone, WITHOUT the plan
            # kernels that followed it: `as_built_tables_us`.)
            if lvl >= 2 or not isinstance(rank, ops.ColumnMap):
                tgs = tgt = tg
            else:
                ch = 16 << lvl
                tgt = timed_graph(lambda: ops.rulebook_subm(idx, B, shape, rank=rank, want_pairs=False), None, 0)
                free = (lvl + 1) in AS_BUILT_FREE
                tgs = timed_graph(lambda: ops.rulebook_subm(idx, B, shape, rank=rank, want_pairs=False, window=(ch, ch),
                                                            nbr_tables=not free), None, 0)
            tot_step += tgs
            tot_tables += tgt
            rows.append(dict(kind="subm" + ("_cm" if isinstance(rank, ops.ColumnMap) else "_ranked" if rank is not None else "_hash"), level=lvl + 1, n_in=n, pairs=P, us=round(t * 1e6, 1),
                             graph_us=round(tg * 1e6, 1), alg_MB=round(by / 1e6, 1), GBps=round(by / t / 1e9, 1),
                             graph_GBps=round(by / tg / 1e9, 1), as_built_us=round(tgs * 1e6, 1), as_built_tables_us=round(tgt * 1e6, 1)))
            tot_bytes += by; tot_t += t; tot_g += tg
    print(json.dumps(dict(frames=B, rulebook_chain_us=round(tot_t * 1e6, 1), rulebook_chain_graph_us=round(tot_g * 1e6, 1),
                          alg_MB=round(tot_bytes / 1e6, 1),
                          achieved_GBps=round(tot_bytes / tot_t / 1e9, 1), frac_of_8TBps=round(tot_bytes / tot_t / 8e12, 4),
                          graph_GBps=round(tot_bytes / tot_g / 1e9, 1), graph_frac_of_8TBps=round(tot_bytes / tot_g / 8e12, 4),
                          as_built_chain_graph_us=round(tot_step * 1e6, 1), as_built_frac_of_8TBps=round(tot_bytes / tot_step / 8e12, 4),
                          as_built_free_levels=sorted(AS_BUILT_FREE),
                          as_built_tables_chain_graph_us=round(tot_tables * 1e6, 1),
                          builds=rows)), flush=True)
    del frames, pts, res, idx
    torch.cuda.empty_cache()
