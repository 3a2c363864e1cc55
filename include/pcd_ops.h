/*
 * pcd_ops.h -- C ABI of libpcdops_hip.so: the MI355X (gfx950) implementation of the sparse-voxel
 * hot path the reference (ZZY816/COM, an OpenPCDet fork) inherits from `spconv` + a few torch modules.
 *
 * Conventions (the reference's own binder convention is
 * pcdet/ops/pointnet2/pointnet2_stack/src/ball_query.cpp:29-45: caller-allocated outputs, int return,
 * raw pointers, contiguous int32 / float32 tensors):
 *   - every function returns 0 on success or a negative PCD_ERR_* code (never exit()s);
 *   - every pointer is a DEVICE pointer unless the name ends in _host; host arrays are read
 *     before the function returns;
 *   - the caller owns every buffer (outputs, workspaces); the library never allocates or frees
 *     device memory and keeps no pointer after returning; all work is enqueued on `stream`
 *     (a hipStream_t, may be NULL for the default stream) and NOT synchronised;
 *   - functions keep no state between calls and may be called from several host threads; the ONLY process-wide state is the
 *     tuning-option table (pcd_set_option: set before the first launch, read unsynchronised at launch time) and the two
 *     profiling pointers that are NULL unless a tool sets them (pcd_subm_window_set_trace);
 *   - index tensors are int32, coordinates are (batch, z, y, x) rows of 4 int32.
 *
 * Device-side row counts: wherever a function takes a row count `n` together with a `const int32_t *n_dev`
 * argument, `n` is the CAPACITY (buffer strides, grid size) and, when n_dev != NULL, the kernels process
 * min(*n_dev, n) rows.  A caller can therefore chain the whole path (voxelise -> rulebooks -> convs -> BEV)
 * without ever reading a data-dependent size back to the host, i.e. inside one captured hipGraph.
 *
 * Feature element types: PCD_BF16 (MFMA path, fp32 accumulate) for sparse-conv features,
 * PCD_F32 for point / voxel payloads.  A feature row has `c_pad` elements, c_pad % 8 == 0,
 * padding channels must be zero.
 */
#ifndef PCD_OPS_H_
#define PCD_OPS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCD_OK 0
#define PCD_ERR_INVALID_ARG (-1)  /* NULL pointer, negative size, bad enum                       */
#define PCD_ERR_UNSUPPORTED (-2)  /* shape / dtype combination this build has no kernel for      */
#define PCD_ERR_KEYSPACE (-3)     /* batch * D * H * W does not fit the 32-bit coordinate key    */
#define PCD_ERR_WORKSPACE (-4)    /* workspace too small                                         */
#define PCD_ERR_LAUNCH (-5)       /* hipGetLastError() != hipSuccess after a launch              */

#define PCD_F32 0
#define PCD_BF16 1

/* Row order of a sparse level whose rows are numbered by a linear coordinate key (the key-ordered voxeliser, the strided
 * rulebook builds and the rank-map SubM builds take it as `row_order`).  Every consumer of the path is invariant to the
 * order (spconv itself numbers GPU-built levels by hash / atomics order); the VOXEL SETS and, through the row permutation,
 * the rulebooks are identical in both.
 *   PCD_ROWS_ZYX  key = ((b*D + z)*H + y)*W + x   x fastest -- spconv's / torch.unique's sorted order
 *   PCD_ROWS_YXZ  key = ((b*H + y)*W + x)*D + z   z fastest: the voxels of a BEV column are consecutive rows and the 27
 *                 neighbours of a run of rows lie in three short runs of rows (BEV rows y-1, y, y+1) -- what the window
 *                 gather-GEMM (pcd_sparse_conv_subm_window) stages in LDS */
#define PCD_ROWS_ZYX 0
#define PCD_ROWS_YXZ 1

/* ---- library info -------------------------------------------------------------------------- */
int pcd_version(void);                    /* 10000*major + 100*minor + patch */
const char *pcd_error_string(int code);
const char *pcd_build_arch(void);         /* "gfx950" */
/* text of the HIP runtime error behind the calling thread's last PCD_ERR_LAUNCH */
const char *pcd_last_hip_error_string(void);
void pcd_set_last_hip_error(int code);    /* internal use */
/* *id_out = 0 when `stream` is not being captured into a hipGraph, else the id of that capture (host-side keying of
 * per-graph resources; no reference counterpart: the reference has no graph capture) */
int pcd_stream_capture_id(void *stream, unsigned long long *id_out);

/* ---- tuning options --------------------------------------------------------------------------
 * Integer knobs of the kernel dispatch (tile shapes, kernel variants, ablation switches).  They select between
 * implementations that produce the same results; the defaults are the measured optima and nothing in the product sets them.
 * Explicit and inspectable (no environment reads anywhere in the library): pcd_set_option returns PCD_ERR_INVALID_ARG for an
 * unknown key.  Process-wide, not synchronised: set them before the first call that uses them.  (No reference counterpart.)
 *   "gg_resident_kb" 32   packed weight up to this size stays resident in LDS in gather_gemm_kernel
 *   "ggw" 1               LDS-DMA gather-GEMM for C_in = 128 (0 off, 2..4: rows-per-wave forced also for C_in = 64, 6: forward only)
 *   "gg1" 1               16-channel gather-GEMM variant (1: 32 rows per wave)
 *   "subm_window" 23      window gather-GEMM for SubM 3x3x3 layers over PCD_ROWS_YXZ rows: bit 0 = 64 channels, bit 1 = 32,
 *                         bit 2 = 16, bit 3 = 128 (EXPERIMENTS builds; off: its dense 27-offset MFMA work makes it slower than the
 *                         step-skipping generic kernel there, 61.6 vs 49.0 us), bit 4 = layers with FEWER input than output
 *                         channels run on zero-padded rows (conv_input 5 -> 16; forward + weight gradient, no data gradient)
 *                         (0: generic kernels) -- read by the host-side layer, the C entry points take any of these widths
 *   "subm_window_wgrad" 6 the same bits for the window weight gradient (pcd_sparse_conv_subm_window_wgrad); 64 channels off:
 *                         its 80 partial slabs (35 MB per layer) cost the training step more than the kernel saves
 *   "wg128" 1             equal-pair weight-gradient kernel at 128 x 128 channels
 *   "wg128_chunks" 512    its workgroup count
 *   "wg_rows" 6144        row-range split of the generic weight-gradient kernel
 *   "conv2d_wb" 1, "conv2d_wg_blocks" 128, "conv2d_wgp_mode2" 0, "conv2d_wgp_blocks" 512   dense 3x3 conv variants
 *   "vox_emit_rows" 1     pcd_voxelize_hard_yxz writes its output rows in row order (a workgroup per 256 columns of the map; 64 / 128:
 *                         that many columns per workgroup; 0 = one thread per point, rows scattered: the round-4 form, same results)
 *   "ggw_cw" 2            consumer waves per SIMD of the wide gather-GEMM at 128 -> 128 channels, 192-row tiles (1 = one, the round-4 form)
 *   "ggw_mi" 0            rows per workgroup / 64 of the wide gather-GEMM: 0 = by rule (forward convs of small levels 2, else 3), 2 / 3 forced
 *   "fps_g" 0             workgroups per frame of the cooperative farthest point sampling (0: from the device's CU count)
 *   "gg_dbg" 0, "ggw_dbg" 0, "win_dbg" 0   ablation bit masks of the gather-GEMM kernels (profiling only)
 *   "ggwin" 0             (EXPERIMENTS build only, pcd_ops_experiments.h) 1: 128 -> 128 SubM layers through ggwin_kernel
 *   "subm_window_half" 0  4-wave window configurations (256 threads, <= 80 KB of LDS: two workgroups per CU): bit 1 = 32 channels,
 *                         bit 2 = 16 channels.  Set it before the first plan is built: plans, packs and launches of a width must agree
 *   "subm_window_grid" 256   workgroups of a window launch (a multiple of 8, <= 256); fewer leave CUs to other streams --
 *                         measured: no gain (240 / 224 / 192: 0 / -0.5 / -1 % in the step)
 *   "cm_direct_blocks" 4096   column-map builds: up to this many scan blocks add up the block sums themselves, beyond it a
 *                             spine launch runs (tests lower it to reach the spine path on small inputs) */
int pcd_set_option(const char *key, int value);
int pcd_get_option(const char *key, int *value_out);

/* ============================================================================================
 * (a1) hard voxelisation -- replaces spconv.utils.VoxelGeneratorV2.generate /
 *      Point2VoxelCPU3d.point_to_voxel as called from
 *      pcdet/datasets/processor/data_processor.py:44-60 (+ the batch collation of
 *      pcdet/datasets/dataset.py:252-259), and optionally fuses MeanVFE
 *      (pcdet/models/backbones_3d/vfe/mean_vfe.py:25-29).
 *
 * points: rows of `point_stride` floats; the C features copied into voxels start at column
 * `feat_offset` and begin with x, y, z (feat_offset = 1 for collated (b,x,y,z,..) rows).
 * Frame b owns point rows [frame_offsets[b], frame_offsets[b+1]).
 * Semantics per frame = reference: point order, coordinate floor((p-min)/vsize) in IEEE f32,
 * voxel ids in first-appearance order, at most max_voxels voxels, first max_points points kept.
 * Output rows are compacted over frames: frame b starts at sum_{b'<b} M_b'.
 *   voxels      [cap][max_points][C] f32, zero padded (may be NULL)
 *   coords      [cap][4] i32 (b, z, y, x)
 *   num_points  [cap] i32
 *   mean_f32    [cap][C] f32 = sum / max(num,1)  (may be NULL)
 *   mean_bf16   [cap][mean_bf16_stride] bf16, zero padded channels (may be NULL)
 *   voxel_counts[batch+1] i32: M_b per frame, last = total rows written
 * cap = capacity in rows of the outputs (>= min(n_points, batch*max_voxels) is always enough).
 * ============================================================================================ */
size_t pcd_voxelize_hard_workspace_bytes(int n_points, int max_points, int batch);
int pcd_voxelize_hard(const float *points, int n_points, int point_stride, int feat_offset,
                      int num_features, const int32_t *frame_offsets, int batch,
                      const float *range_host /*[6]*/, const float *vsize_host /*[3]*/,
                      int max_points, int max_voxels, int cap, float *voxels, int32_t *coords,
                      int32_t *num_points, float *mean_f32, void *mean_bf16, int mean_bf16_stride,
                      int32_t *voxel_counts, void *workspace, size_t workspace_bytes, void *stream);

/* Same operation, same kept set (the first max_voxels voxels per frame in first-appearance order, their first
 * max_points points in point order -- data_processor.py:44-60), rows numbered by ascending (b, z, y, x) key instead of
 * by first appearance: the order torch.unique (dynamic_mean_vfe.py:57-66) and spconv's strided convs give their rows.
 * Every consumer of the hot path (mean_vfe.py:25-29 -> spconv_backbone.py:239-246) is invariant to the row order;
 * with training's shuffled points (data_processor.py:103-113) the first-appearance order is random in space, the key
 * order keeps the 27 neighbours of a level-1 row in nearby rows.  Needs batch*gz*gy*gx < 2^32 - 1024 (PCD_ERR_KEYSPACE
 * otherwise); the workspace holds an occupancy bitmap of that many bits.
 *   rank_bitmap / rank_prefix (both or neither): caller-owned buffers of pcd_voxelize_hard_sorted_rank_words() words
 *   that receive the coordinate -> row map of the output -- bit (key) of rank_bitmap set for every kept voxel,
 *   rank_prefix[g] = number of set bits in front of the 128-bit group g -- the map pcd_rulebook_subm_ranked4 builds the
 *   level-1 SubM rulebook from (no hash table).  NULL: the map lives in the workspace and dies with the call.
 *   key_depth: z extent D of the key space (PCD_ROWS_* above with H = gy, W = gx); 0 = gz.  The 3D backbones use
 *   sparse_shape = grid_size[::-1] + [1, 0, 0] (spconv_backbone.py:87,187): pass gz + 1 so that the map has the layout
 *   the rulebook of that shape addresses.  The row ORDER does not depend on it.
 *   row_order: PCD_ROWS_ZYX / PCD_ROWS_YXZ. */
size_t pcd_voxelize_hard_sorted_workspace_bytes(int n_points, int max_points, int batch,
                                                const float *range_host /*[6]*/, const float *vsize_host /*[3]*/,
                                                int key_depth);
int pcd_voxelize_hard_sorted_rank_words(int batch, const float *range_host /*[6]*/, const float *vsize_host /*[3]*/,
                                        int key_depth, size_t *bitmap_words, size_t *prefix_words);
int pcd_voxelize_hard_sorted(const float *points, int n_points, int point_stride, int feat_offset,
                             int num_features, const int32_t *frame_offsets, int batch,
                             const float *range_host /*[6]*/, const float *vsize_host /*[3]*/,
                             int max_points, int max_voxels, int cap, float *voxels, int32_t *coords,
                             int32_t *num_points, float *mean_f32, void *mean_bf16, int mean_bf16_stride,
                             int32_t *voxel_counts, int key_depth, int row_order, uint32_t *rank_bitmap,
                             int32_t *rank_prefix, void *workspace, size_t workspace_bytes, void *stream);
/* pcd_voxelize_hard_sorted with row_order PCD_ROWS_YXZ whose coordinate -> row map is the level's COLUMN MAP ("Column maps"
 * below): `colmap` is a caller-owned buffer of pcd_colmap_bytes(batch, (max(gz, key_depth), gy, gx), cap) bytes that the
 * rulebook builds of level 1 read (pcd_rulebook_subm_cm, pcd_rulebook_conv_cm_*; pass the same `cap`).  Same voxels, same
 * rows as pcd_voxelize_hard_sorted(PCD_ROWS_YXZ); the ranks come from two order-free marks (BEV occupancy bit, z bit of the
 * column's mask) and two small scans instead of a bitmap over the (b, y, x, z) key space (46 MB zero-filled, scanned and
 * probed per 4-frame Waymo batch) -- and no separate pcd_colmap_from_rows pass.  key depth <= 62, batch <= 256
 * (PCD_ERR_UNSUPPORTED / 0 bytes otherwise: use pcd_voxelize_hard_sorted + pcd_colmap_from_rows).
 * Replaces the same call site (pcdet/datasets/processor/data_processor.py:44-60,125-153). */
/* Host-side variant for ONE frame: what the reference calls inside forked DataLoader worker processes
 * (pcdet/datasets/processor/data_processor.py:44-60,130-141), where a HIP call is impossible.  HOST pointers, no stream, no GPU
 * work: the sequential algorithm of SURVEY.md A.1 with the kernels' own coordinate arithmetic -- first-appearance voxel ids,
 * the first max_points points of every voxel, the max_voxels cut.  voxels [max_voxels][max_points][num_features] (rows used
 * are zero padded), coords [max_voxels][3] (z, y, x), num_points [max_voxels]; *num_voxels_out = M.  Same results as
 * pcd_voxelize_hard on one frame, bit for bit. */
int pcd_voxelize_hard_host(const float *points_host, int n_points, int point_stride, int num_features,
                           const float *range_host, const float *vsize_host, int max_points, int max_voxels,
                           float *voxels_host, int32_t *coords_host, int32_t *num_points_host, int32_t *num_voxels_out);
size_t pcd_voxelize_hard_yxz_workspace_bytes(int n_points, int max_points, int batch, const float *range_host,
                                             const float *vsize_host, int key_depth, int cap);
int pcd_voxelize_hard_yxz(const float *points, int n_points, int point_stride, int feat_offset, int num_features,
                          const int32_t *frame_offsets, int batch, const float *range_host, const float *vsize_host,
                          int max_points, int max_voxels, int cap, float *voxels, int32_t *coords, int32_t *num_points,
                          float *mean_f32, void *mean_bf16, int mean_bf16_stride, int32_t *voxel_counts, int key_depth,
                          void *colmap, size_t colmap_bytes, void *workspace, size_t workspace_bytes, void *stream);
/* (a4) MeanVFE on materialised voxels: mean_vfe.py:25-29.  out [m][C] f32. */
int pcd_mean_vfe(const float *voxels, const int32_t *num_points, int m, int max_points,
                 int num_features, float *out, void *stream);

/* ============================================================================================
 * (a5) dynamic voxelisation + mean -- replaces torch.unique + torch_scatter.scatter_mean in
 *      pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:53-72.
 * points_b rows = (batch_idx, x, y, z, f3, ...) with 1 + C floats.  Output sorted by the
 * reference key b*XYZ + cx*YZ + cy*Z + cz; coords (b, z, y, x); no caps.
 *   features [cap][C] f32, coords [cap][4], counts [cap] (may be NULL), num_voxels [1] i32.
 * ============================================================================================ */
size_t pcd_voxelize_dynamic_workspace_bytes(int n_points, int num_features, int batch,
                                            const float *range_host, const float *vsize_host);
int pcd_voxelize_dynamic_mean(const float *points_b, int n_points, int num_features, int batch,
                              const float *range_host, const float *vsize_host, int cap,
                              float *features, int32_t *coords, int32_t *counts,
                              int32_t *num_voxels, int32_t *point_voxel, void *workspace,
                              size_t workspace_bytes, void *stream);
/* point_voxel (may be NULL) [n_points] i32: output row of every point (-1: dropped) = `unq_inv` of the
 * torch.unique(..., return_inverse=True) call of the dynamic VFEs (dynamic_mean_vfe.py:63,
 * dynamic_pillar_vfe.py:103).
 *
 * (a6) dynamic pillar encoder -- replaces torch_scatter.scatter_max in PFNLayerV2 (dynamic_pillar_vfe.py:36-47).
 *   pcd_segment_max:  out[s][ch] = max over the points i with seg[i] == s of x[i][ch]  (f32; seg < 0 skipped),
 *   arg[s][ch] = the smallest such i attaining it (deterministic; torch_scatter leaves ties unspecified).
 *   pcd_segment_max_backward:  grad_x [n][c] = 0 except grad_x[arg[s][ch]][ch] = grad_out[s][ch]. */
size_t pcd_segment_max_workspace_bytes(int m, int c);
int pcd_segment_max(const float *x, const int32_t *seg, int n, int c, int m, float *out, int32_t *arg,
                    void *workspace, size_t workspace_bytes, void *stream);
int pcd_segment_max_backward(const float *grad_out, const int32_t *arg, int n, int c, int m, float *grad_x,
                             void *stream);

/* ============================================================================================
 * (a8) SubMConv3d rulebook -- replaces spconv's get_indice_pairs(subm=True) behind
 *      spconv.SubMConv3d (pcdet/models/backbones_3d/spconv_backbone.py:12,38-45).
 * indices [n][4].  K = kd*kh*kw, offsets enumerated k = (kd*KH+kh)*KW+kw,
 * in_pos = out_pos - (ksize/2)*dil + k_idx*dil.
 *   nbr      [K][n] i32: nbr[k][o] = input row feeding output row o through offset k, or -1.
 *            (SubM symmetry: the input-stationary view is nbr_in[k] = nbr[K-1-k].)
 *   pairs    [K][2][n] i32 (may be NULL): spconv's indice_pairs, canonical order = ascending input
 *            row within each k, -1 padded.
 *   pair_num [K] i32 (required iff pairs != NULL)
 *   pad_pairs != 0: entries of pairs beyond pair_num[k] are set to -1 (spconv's padding; costs a memset
 *            of the whole table -- the compute kernels never read them)
 * ============================================================================================ */
size_t pcd_rulebook_subm_workspace_bytes(int n, int kvol);
int pcd_rulebook_subm(const int32_t *indices, int n, int batch, const int *shape_host /*[3]*/,
                      const int *ksize_host, const int *dil_host, int32_t *nbr, int32_t *pairs,
                      int32_t *pair_num, int pad_pairs, const int32_t *n_dev, void *workspace,
                      size_t workspace_bytes, void *stream);

/* ============================================================================================
 * (a9) SparseConv3d (strided) rulebook -- replaces get_indice_pairs(subm=False) behind
 *      spconv.SparseConv3d (spconv_backbone.py:14-15,205-229).  Two phases because the number of
 *      output rows is data dependent:
 *   phase 1 `_count`: marks the active output cells, ranks them (sorted by the linear key of
 *           `row_order`, PCD_ROWS_*) and writes n_out to n_out_dev[0]; state lives in `workspace`.
 *   phase 2 `_fill` : (after the caller read n_out and allocated) emits
 *           out_indices [n_out][4], nbr_in [K][n] (output row fed by input i, or -1),
 *           nbr_out [K][n_out] (input row feeding output o, or -1),
 *           pairs [K][2][n] + pair_num [K] (may be NULL/NULL).
 *   The same `workspace` (untouched in between) must be passed to both phases.
 * out_shape_host [3] receives floor((in + 2p - d(k-1) - 1)/s) + 1.
 * ============================================================================================ */
size_t pcd_rulebook_conv_workspace_bytes(int n, int batch, const int *in_shape_host,
                                         const int *ksize_host, const int *stride_host,
                                         const int *pad_host, const int *dil_host);
int pcd_conv_out_shape(const int *in_shape_host, const int *ksize_host, const int *stride_host,
                       const int *pad_host, const int *dil_host, int *out_shape_host);
int pcd_rulebook_conv_count(const int32_t *indices, int n, int batch, const int *in_shape_host,
                            const int *ksize_host, const int *stride_host, const int *pad_host,
                            const int *dil_host, int32_t *n_out_dev, const int32_t *n_dev, void *workspace,
                            size_t workspace_bytes, void *stream, int row_order);
int pcd_rulebook_conv_fill(const int32_t *indices, int n, int batch, const int *in_shape_host,
                           const int *ksize_host, const int *stride_host, const int *pad_host,
                           const int *dil_host, int n_out, int32_t *out_indices, int32_t *nbr_in,
                           int32_t *nbr_out, int32_t *pairs, int32_t *pair_num, int pad_pairs,
                           const int32_t *n_dev, void *workspace, size_t workspace_bytes, void *stream,
                           int row_order);

/* Both phases (+ the parity classes of pcd_rulebook_conv_classes when perm != NULL) in ONE call for callers that
 * bound the number of output rows on the host (`n_out_cap`: buffers are sized for it, rows beyond it are dropped and
 * n_out_dev[0] still receives the real count -- static plans / hipGraph capture, nothing is read back).  Same
 * results as _count + _fill + _classes; six dependent launches (five without pair lists) instead of fifteen:
 * fill, mark (+ class counts), pack + block sums, scan + emit + 0xFF fills + class offsets, neighbour tables
 * (+ class permutation), pair lists.  Workspace: pcd_rulebook_conv_workspace_bytes.
 * Replaces the same get_indice_pairs(subm=False) call (spconv_backbone.py:14-15,205-229). */
int pcd_rulebook_conv_build(const int32_t *indices, int n, int batch, const int *in_shape_host,
                            const int *ksize_host, const int *stride_host, const int *pad_host,
                            const int *dil_host, int n_out_cap, int32_t *n_out_dev, int32_t *out_indices,
                            int32_t *nbr_in, int32_t *nbr_out, int32_t *pairs, int32_t *pair_num, int pad_pairs,
                            int cls_tile, int32_t *perm, int vcap, int32_t *vstart_dev, const int32_t *n_dev,
                            void *workspace, size_t workspace_bytes, void *stream, int row_order);

/* indice_pairs / indice_pair_num of a SubM rulebook from its nbr table [kvol][n] alone -- for rulebooks built with
 * pairs == NULL (the forward and the output-stationary kernels only need nbr) whose pairs are wanted later. */
size_t pcd_rulebook_subm_pairs_workspace_bytes(int n, int kvol);
int pcd_rulebook_subm_pairs(const int32_t *nbr, int n, int kvol, int32_t *pairs, int32_t *pair_num, int pad_pairs,
                            const int32_t *n_dev, void *workspace, size_t workspace_bytes, void *stream);
/* ... and of a STRIDED rulebook from its nbr_in table [kvol][n] (same workspace size): the training step builds strided
 * rulebooks without pair lists (pcd_sparse_conv_wgrad_classes reads the pairs off the parity classes); whoever still wants
 * spconv's indice_pairs (spconv/pytorch/ops.py get_indice_pairs' outputs) derives them here, same values as the build's. */
int pcd_rulebook_conv_pairs(const int32_t *nbr_in, int n, int kvol, int32_t *pairs, int32_t *pair_num, int pad_pairs,
                            const int32_t *n_dev, void *workspace, size_t workspace_bytes, void *stream);
/* The rank structure the strided build leaves in its workspace is exactly a coordinate -> row map of the OUTPUT
 * level (row id = rank of the linear key): the SubM rulebook of that level (the 'subm2..4' keys that follow
 * every strided conv of spconv_backbone.py:205-229) can use it instead of building and probing a hash table.
 *   pcd_rulebook_conv_rank_layout: byte offsets of the occupancy bitmap (u32 [nwords]) and of its exclusive
 *       popcount prefix (i32 [nwords]) inside a workspace of pcd_rulebook_conv_workspace_bytes(same arguments);
 *       valid after pcd_rulebook_conv_count, for as long as the caller keeps that workspace untouched.
 *   pcd_rulebook_subm_ranked: same outputs (bit for bit) as pcd_rulebook_subm for indices == the out_indices of
 *       that strided build, shape_host == its output shape; 3x3x3 kernels only (PCD_ERR_UNSUPPORTED otherwise).
 *       Ranks >= n (rows dropped by a static capacity) count as missing neighbours. */
int pcd_rulebook_conv_rank_layout(int n, int batch, const int *in_shape_host, const int *ksize_host,
                                  const int *stride_host, const int *pad_host, const int *dil_host,
                                  size_t *bitmap_offset, size_t *prefix_offset, size_t *nwords);
size_t pcd_rulebook_subm_ranked_workspace_bytes(int n, int kvol);
/* pcd_rulebook_subm_ranked4: the same for the map pcd_voxelize_hard_sorted hands out (one prefix per group of 4 bitmap
 * words): the SubM rulebook of level 1 (spconv_backbone.py:199-203, 'subm1' / 'res1') without a hash table. */
int pcd_rulebook_subm_ranked4(const int32_t *indices, int n, int batch, const int *shape_host,
                              const int *ksize_host, const int *dil_host, const uint32_t *bitmap,
                              const int32_t *prefix, int32_t *nbr, int32_t *pairs, int32_t *pair_num,
                              int pad_pairs, const int32_t *n_dev, void *workspace, size_t workspace_bytes,
                              void *stream, int row_order);
int pcd_rulebook_subm_ranked(const int32_t *indices, int n, int batch, const int *shape_host,
                             const int *ksize_host, const int *dil_host, const uint32_t *bitmap,
                             const int32_t *prefix, int32_t *nbr, int32_t *pairs, int32_t *pair_num,
                             int pad_pairs, const int32_t *n_dev, void *workspace, size_t workspace_bytes,
                             void *stream, int row_order);

/* ---- Column maps: the coordinate -> row map of a level whose rows are numbered z-fastest (PCD_ROWS_YXZ) ----------
 * In (b, y, x, z) order the rows of one BEV cell (a COLUMN) are consecutive and ordered by z, so a level's map is
 *     words  [B*H*W / 32]: { occupancy bits of 32 BEV cells, occupied cells in front of the word }       (8 bytes each)
 *     columns [<= rows]   : { z mask (64 bits), first row, rows }                                        (16 bytes each)
 * -- O(BEV area / 32) + O(rows) bytes instead of a bitmap over the (b, y, x, z) key space (371 M cells at level 1 of a
 * 4-frame Waymo batch).  The map lives in ONE caller-owned buffer of pcd_colmap_bytes(batch, shape (D, H, W), n_cap)
 * bytes (n_cap = row capacity of the level = column capacity; the same n_cap must be passed wherever the buffer is read).
 * D <= 62; the builds below cover dilation 1, kernels 3x3x3 and (3,1,1), strides 1 / 2 per axis (every conv of
 * pcdet/models/backbones_3d/spconv_backbone.py:69-293) and return PCD_ERR_UNSUPPORTED otherwise (0 from the
 * _workspace_bytes query): callers then take pcd_rulebook_conv_* / pcd_rulebook_subm.
 *   pcd_colmap_from_rows: the map of a row set given in (b, y, x, z) order (level 1: the key-ordered voxeliser's output).
 *   pcd_rulebook_subm_cm: pcd_rulebook_subm's outputs (bit for bit) for a 3x3x3 SubM conv (spconv_backbone.py:12-13) from
 *       the level's map: nine column lookups per row serve the 27 offsets; no hash table, no atomics.
 *   pcd_rulebook_conv_cm_{count,fill,build}: pcd_rulebook_conv_{count,fill,build}'s outputs for a strided conv
 *       (spconv_backbone.py:14-15,205-229) with row_order PCD_ROWS_YXZ, from the INPUT level's map; additionally writes
 *       the OUTPUT level's map (out_colmap, sized for n_out / n_out_cap rows).  An output column exists iff one of its
 *       kh x kw input columns does; its z mask is the OR of theirs, shifted by the padding, smeared over the kd taps and
 *       compressed by the stride: one thread per output BEV cell, no atomics, no map of the output volume.  Launches:
 *       count (+ class counts) -> emit map + coordinates (+ class offsets) -> both neighbour tables (+ class
 *       permutation) -> pair lists.  _count / _fill share one untouched workspace (the host reads n_out in between). */
size_t pcd_colmap_bytes(int batch, const int *shape_host, int n_cap);
/* Byte offset, inside a buffer of pcd_colmap_bytes(batch, shape, n_cap) bytes, of the map's counters {columns, rows} (two int32,
 * written by the build that produced the map); *ncol_cap_out receives its column capacity.  A strided build whose output z
 * range does not cover every input z numbers columns WITHOUT rows, so columns > rows is possible: check columns <= capacity
 * (com_amd.ops does: it rebuilds with a larger map in eager mode and records the count with the static plan's overflow guard). */
size_t pcd_colmap_counts_offset(int batch, const int *shape_host, int n_cap, int *ncol_cap_out);
size_t pcd_colmap_from_rows_workspace_bytes(int batch, const int *shape_host);
int pcd_colmap_from_rows(const int32_t *indices, int n, const int32_t *n_dev, int batch, const int *shape_host,
                         void *colmap, size_t colmap_bytes, void *workspace, size_t workspace_bytes, void *stream);
size_t pcd_rulebook_subm_cm_workspace_bytes(int n);
int pcd_rulebook_subm_cm(const int32_t *indices, int n, int batch, const int *shape_host, const void *colmap,
                         size_t colmap_bytes, int colmap_cap, int32_t *nbr, int32_t *pairs, int32_t *pair_num,
                         int pad_pairs, const int32_t *n_dev, void *workspace, size_t workspace_bytes, void *stream);
size_t pcd_rulebook_conv_cm_workspace_bytes(int n, int batch, const int *in_shape_host, const int *ksize_host,
                                            const int *stride_host, const int *pad_host);
int pcd_rulebook_conv_cm_count(int n, int batch, const int *in_shape_host, const int *ksize_host,
                               const int *stride_host, const int *pad_host, const void *in_colmap,
                               size_t in_colmap_bytes, int in_cap, int32_t *n_out_dev, void *workspace,
                               size_t workspace_bytes, void *stream);
int pcd_rulebook_conv_cm_fill(const int32_t *indices, int n, int batch, const int *in_shape_host,
                              const int *ksize_host, const int *stride_host, const int *pad_host,
                              const void *in_colmap, size_t in_colmap_bytes, int in_cap, int n_out,
                              int32_t *out_indices, void *out_colmap, size_t out_colmap_bytes, int32_t *nbr_in,
                              int32_t *nbr_out, int32_t *pairs, int32_t *pair_num, int pad_pairs,
                              const int32_t *n_dev, void *workspace, size_t workspace_bytes, void *stream);
int pcd_rulebook_conv_cm_build(const int32_t *indices, int n, int batch, const int *in_shape_host,
                               const int *ksize_host, const int *stride_host, const int *pad_host,
                               const void *in_colmap, size_t in_colmap_bytes, int in_cap, int n_out_cap,
                               int32_t *n_out_dev, int32_t *out_indices, void *out_colmap, size_t out_colmap_bytes,
                               int32_t *nbr_in, int32_t *nbr_out, int32_t *pairs, int32_t *pair_num, int pad_pairs,
                               int cls_tile, int32_t *perm, int vcap, int32_t *vstart_dev, const int32_t *n_dev,
                               void *workspace, size_t workspace_bytes, void *stream);
/* The same one-call build with COMPACT neighbour tables and no pair lists -- what a training step needs of a strided rulebook
 * (replaces the indice_pairs / indice_pair_num that spconv/pytorch/ops.py get_indice_pairs hands to SparseConv3d,
 * spconv_backbone.py:205,212,219, for kernel depth 3 and at most 8 stride-parity classes):
 *   nbr_out_packed [kh * kw][n_out_cap] u32: per (ky, kx) the three kz neighbours of an output row -- consecutive rows of one input
 *       column, rows being z-fastest -- as { first present row : 29 bits, presence of kz = 0, 1, 2 : 3 bits } (0 = none);
 *       read by pcd_sparse_conv_gather_gemm_packed;
 *   nbr_cls [8][vcap] i32: entry (j, v) = output row reached from input row perm[v] through the j-th kernel offset its class can
 *       use (ascending k), or -1: 27 / 8 entries per row on average; read by pcd_sparse_conv_dgrad_classes_v2 and
 *       pcd_sparse_conv_wgrad_classes (nbr_compact = 1).
 * 4 x 9 + 4 x 27 / 8 = 50 bytes per row instead of 2 x 108 (+ 8 per pair).  pcd_rulebook_conv_expand_nbr_out / _nbr_in rebuild
 * the 27-wide tables (nbr_out [3 kq][n_out], nbr_in [kvol][n]) for whoever wants them: same values as pcd_rulebook_conv_cm_build's. */
int pcd_rulebook_conv_cm_build_compact(const int32_t *indices, int n, int batch, const int *in_shape_host,
                                       const int *ksize_host, const int *stride_host, const int *pad_host,
                                       const void *in_colmap, size_t in_colmap_bytes, int in_cap, int n_out_cap,
                                       int32_t *n_out_dev, int32_t *out_indices, void *out_colmap, size_t out_colmap_bytes,
                                       uint32_t *nbr_out_packed, int32_t *nbr_cls, int cls_tile, int32_t *perm, int vcap,
                                       int32_t *vstart_dev, const int32_t *n_dev, void *workspace, size_t workspace_bytes,
                                       void *stream);
int pcd_rulebook_conv_expand_nbr_out(const uint32_t *nbr_out_packed, int kq, int n_out, const int32_t *n_out_dev,
                                     int32_t *nbr_out, void *stream);
int pcd_rulebook_conv_expand_nbr_in(const int32_t *nbr_cls, int vcap, const int32_t *perm, const int32_t *vstart_dev,
                                    const int *ksize_host, const int *stride_host, int n, int32_t *nbr_in, void *stream);

/* Optional per-channel reductions of the OUTPUT tile in the epilogue of pcd_sparse_conv_gather_gemm /
 * pcd_sparse_conv_dgrad_classes (bf16 outputs only; NULL or mode 0 = off).  The BatchNorm1d that follows every conv of
 * the reference backbones (spconv_backbone.py:21-25,50-66) needs these sums; taking them while the values are still in
 * registers replaces one streaming pass per BatchNorm and direction:
 *   mode 1 (forward)       partial[t] = { sum y, sum y^2 } of the rounded outputs  -> pcd_bn_forward(ext_partial)
 *   mode 2 (data gradient) the output is dy of the BatchNorm(+ReLU) whose output was the conv's input:
 *                          partial[t] = { sum dz, sum dz*xhat }, dz = relu ? dy*(y > 0) : dy, xhat = (x-mean)*invstd;
 *                          y = that BatchNorm's output = the conv's own input features (required when relu)
 *                                                                               -> pcd_bn_backward(ext_partial)
 * One row [2][c_out] per workgroup tile t; partial_rows must be >= pcd_sparse_conv_gather_gemm_tiles(..) /
 * pcd_sparse_conv_dgrad_classes_tiles(..), which is also the row count to hand to the BatchNorm call.
 * mean / invstd must be 16-byte aligned.  Fixed summation order (deterministic). */
typedef struct PcdBnReduce {
    int mode;
    int relu;
    const void *x;            /* mode 2: BatchNorm input  [rows][c_out] bf16 */
    const void *y;            /* mode 2: BatchNorm output [rows][c_out] bf16 (NULL when relu == 0) */
    const float *mean, *invstd;
    float *partial;           /* out: [partial_rows][2][c_out] f32 */
    int partial_rows;
    /* Optional: the launch itself folds the partial rows into the PCD_BN_MID_ROWS rows the BatchNorm apply passes start
     * from (otherwise pcd_bn_forward / _backward run a small kernel for that -- one more dependent launch on the chain):
     * mid [PCD_BN_MID_ROWS][2][c_out] f64, handed to the BatchNorm call as ext_partial with ext_rows = PCD_BN_EXT_MID;
     * counters [PCD_BN_MID_ROWS * PCD_BN_COUNTER_STRIDE] i32 (one counter per 128-byte line) must be ZERO when the launch
     * starts and are zero again when it ends (keep them in a buffer zeroed once); partial_rows must then equal the
     * tile count exactly. */
    double *mid;
    int32_t *counters;
} PcdBnReduce;
#define PCD_BN_MID_ROWS 16
#define PCD_BN_COUNTER_STRIDE 32
#define PCD_BN_EXT_MID (-1)

/* Parity classes of the input rows of a strided conv, for its data gradient: input coordinate c reaches an output
 * cell through kernel index k only if (c + p - k*d) is a multiple of the stride, so the residues ((c + p) mod s) of
 * the three axes select the 1..8 offsets (of 27 for k = 3, s = 2) a row can use at all.
 *   perm   [vcap] i32 : virtual row -> input row (stable inside a class), -1 = padding; every class starts at a
 *                       multiple of `tile`; vcap >= round_up(n, tile) + (sd*sh*sw) * tile
 *   vstart [sd*sh*sw + 1] i32 on the DEVICE: first virtual row of every class, last entry = end.
 * pcd_sparse_conv_dgrad_classes then runs, per class, only that class's offsets (bit-identical to
 * pcd_sparse_conv_gather_gemm on nbr_in; c_dy >= 32, a power of two; tile must be 256). */
size_t pcd_rulebook_conv_classes_workspace_bytes(int n);
int pcd_rulebook_conv_classes(const int32_t *indices, int n, const int *stride_host, const int *pad_host, int tile,
                              int32_t *perm, int vcap, int32_t *vstart_dev, const int32_t *n_dev, void *workspace,
                              size_t workspace_bytes, void *stream);
int pcd_sparse_conv_dgrad_classes(const void *dy, int n_dy_rows, int c_dy, const void *packed_w,
                                  const int32_t *nbr_in, int nbr_stride, const int *ksize_host,
                                  const int *stride_host, const int *pad_host, const int *dil_host,
                                  const int32_t *perm, const int32_t *vstart_dev, int vcap, int n_rows_in, int c_in,
                                  void *dx, int dx_dtype, const void *addend, const PcdBnReduce *bn_reduce,
                                  void *stream);
/* v2: nbr_compact = 1 reads the class-compact table nbr_cls [8][nbr_stride] (pcd_rulebook_conv_cm_build_compact; nbr_stride = the
 * permutation's capacity) instead of nbr_in [kvol][nbr_stride]: coalesced table reads instead of a gather through perm. */
int pcd_sparse_conv_dgrad_classes_v2(const void *dy, int n_dy_rows, int c_dy, const void *packed_w, const int32_t *nbr_in,
                                     int nbr_stride, int nbr_compact, const int *ksize_host, const int *stride_host,
                                     const int *pad_host, const int *dil_host, const int32_t *perm, const int32_t *vstart_dev,
                                     int vcap, int n_rows_in, int c_in, void *dx, int dx_dtype, const void *addend,
                                     const PcdBnReduce *bn_reduce, void *stream);
/* The forward of a strided conv over the PACKED output-side table nbr_out_packed [kvol / 3][nbr_stride] of
 * pcd_rulebook_conv_cm_build_compact; otherwise pcd_sparse_conv_gather_gemm (bit-identical result). */
int pcd_sparse_conv_gather_gemm_packed(const void *x, int n_rows_in, int c_in, const void *packed_w, const float *bias,
                                       const uint32_t *nbr_out_packed, int nbr_stride, int kvol, int n_rows_out,
                                       const int32_t *n_rows_out_dev, int c_out, void *y, int y_dtype, const void *addend,
                                       const PcdBnReduce *bn_reduce, void *stream);
int pcd_sparse_conv_dgrad_classes_tiles(int vcap, int n_rows_in);

/* ============================================================================================
 * (a8-a10) sparse convolution arithmetic -- replaces spconv's indice_conv fwd/bwd.
 *
 * Weights: the module parameter keeps spconv-2.x layout  weight [Cout][K][Cin] f32
 * (pcdet/models/detectors/detector3d_template.py:341-348).  pcd_pack_weight converts it to the
 * bf16 MFMA-fragment order the kernels read (call after every weight update):
 *   mode 0 (forward):  contraction over (k, cin),  outputs cout
 *   mode 1 (dgrad):    contraction over (k, cout), outputs cin
 * packed size in bytes = pcd_packed_weight_bytes(K, c_contract_pad, c_out_pad_to_16).
 * ============================================================================================ */
size_t pcd_packed_weight_bytes(int kvol, int cin, int cout, int mode);
int pcd_pack_weight(const float *weight, int kvol, int cin, int cout, int mode, void *packed,
                    void *stream);
/* The same conversion for a whole list of weights in ONE launch (a backbone re-packs ~40 small weights after
 * every optimizer step).  `table` is DEVICE memory, int64 [n][8], row i =
 *   { weight pointer, packed pointer, kvol, cin, cout, mode, first_block_i, 0 },
 * first_block_0 = 0, first_block_{i+1} = first_block_i + ceil(pcd_packed_weight_bytes(..)/2 / 2048)   (2048 packed elements per block);
 * total_blocks = first_block_n.  The table can be built once and reused while the pointers stay valid. */
int pcd_pack_weights_batched(const void *table, int n, int total_blocks, void *stream);

/* y[o] = bias + sum_k x[nbr[k'][o]] @ W[k],  k' = flip_k ? K-1-k : k.
 * Output-stationary gather-GEMM (no atomics, deterministic).  Used for
 *   forward : x = features [n_in][cin_pad] bf16, nbr = nbr_out [K][n_out], packed mode 0;
 *   dgrad   : x = dY [n_out][cout_pad], nbr = nbr_in [K][n_in] (SubM: nbr with flip_k = 1),
 *             packed mode 1.
 * n_rows_in = rows of x (bounds of the gathers), c_in = contraction channels (row stride of x, a power
 * of two >= 8), c_out = output channels (% 16 == 0),
 * nbr_stride = row stride (elements) of the nbr table.  y dtype PCD_BF16 or PCD_F32; bias f32 or NULL.
 * addend (NULL or [n_rows_out][c_out] of y's dtype) is added before the single rounding of y: the gradient of the
 * residual branch in the dgrad of a SparseBasicBlock's first conv (spconv_backbone.py:56-63) -- replaces the
 * elementwise add autograd would launch. */
int pcd_sparse_conv_gather_gemm(const void *x, int n_rows_in, int c_in, const void *packed_w, const float *bias,
                                const int32_t *nbr, int nbr_stride, int kvol, int flip_k,
                                int n_rows_out, const int32_t *n_rows_out_dev, int c_out, void *y,
                                int y_dtype, const void *addend, const PcdBnReduce *bn_reduce, void *stream);
/* number of workgroup tiles (= partial rows of bn_reduce) of that launch; < 0: error code */
int pcd_sparse_conv_gather_gemm_tiles(int n_rows_in, int c_in, int kvol, int n_rows_out, int c_out);
/* same for a launch that is a DATA GRADIENT (flip_k != 0 or bn_reduce->mode == 2): the library may pick a different
 * kernel (tile height) for the two directions */
int pcd_sparse_conv_gather_gemm_tiles_dir(int n_rows_in, int c_in, int kvol, int n_rows_out, int c_out, int is_dgrad);
/* which kernel that launch runs: 0 = gather_gemm_kernel (fragment loads), 1 = ggw_kernel (LDS-DMA, loader / consumer
 * waves); for profiling tools that group launches by kernel name */
int pcd_sparse_conv_gather_gemm_variant(int n_rows_in, int c_in, int kvol, int n_rows_out, int c_out, int is_dgrad);

/* dW[cout][k][cin] = sum_{(i,o) in pairs[k]} dY[o][cout] * X[i][cin]   (f32, parameter layout).
 * Two launches: pcd_sparse_conv_wgrad fills partial results in `workspace` (MFMA kernel) -- per row-range slabs, or at
 * 128 x 128 channels one tile per (equal-pair chunk, offset) behind a small header --,
 * pcd_sparse_conv_wgrad_reduce sums them in a fixed order into dweight (deterministic, no atomics).
 * `dweight` of the first call is only used when pmax == 0 (it is zeroed); it may be NULL otherwise.
 * pairs[k][0] (rows of x, n_x_rows of them) must be ascending inside each k (canonical order): the
 * kernel partitions the work by ranges of x rows and binary-searches the pair list. */
size_t pcd_sparse_conv_wgrad_workspace_bytes(int kvol, int cin, int cout, int pmax);
int pcd_sparse_conv_wgrad(const void *x, int n_x_rows, int cin_pad, int cin, const void *dy, int n_dy_rows, int cout,
                          const int32_t *pairs, const int32_t *pair_num, int kvol, int pmax,
                          float *dweight, void *workspace, size_t workspace_bytes, void *stream);
/* v2: n_x_dev (may be NULL) = device-side count of the real rows of x when n_x is a capacity (static-shape mode): the
 * row-range splits then partition the real rows, so no workgroup -- and no XCD -- is left with an empty range */
int pcd_sparse_conv_wgrad_v2(const void *x, int n_x, const int32_t *n_x_dev, int cin_pad, int cin, const void *dy,
                             int n_dy, int cout, const int32_t *pairs, const int32_t *pair_num, int kvol, int pmax,
                             float *dweight, void *workspace, size_t workspace_bytes, void *stream);
/* The weight gradient of a STRIDED conv without pair lists (replaces the indice_pairs argument of spconv's
 * indice_conv_backward, spconv/pytorch/ops.py, for SparseConv3d layers spconv_backbone.py:205,212,219): every kernel offset
 * k is usable by the input rows of exactly one stride-parity class, and for those rows it (almost) always has an output, so
 * the pairs of offset k are {(i, nbr_in[k][i]) : i in class(k)} -- read off `perm` / `vstart_dev` (pcd_rulebook_conv_cm_build /
 * pcd_rulebook_conv_classes) and nbr_in [kvol][nbr_stride]; a missing output gathers a zero row.  Same workspace
 * (pcd_sparse_conv_wgrad_workspace_bytes(kvol, cin, cout, n_x)), same reduction (pmax = n_x); the result equals
 * pcd_sparse_conv_wgrad_v2's over the rulebook's pair lists to fp32 summation order (a cut output holds a slot with a zero row).  kvol <= 27, at most 8 classes, not 128 x 128 channels. */
int pcd_sparse_conv_wgrad_classes(const void *x, int n_x, const int32_t *n_x_dev, int cin_pad, int cin, const void *dy,
                                  int n_dy, int cout, const int32_t *nbr_in, int nbr_stride, const int *ksize_host,
                                  const int *stride_host, const int *dil_host, const int32_t *perm,
                                  const int32_t *vstart_dev, float *dweight, void *workspace, size_t workspace_bytes,
                                  void *stream, int nbr_compact);
/* nbr_compact = 1: `nbr_in` is the class-compact table nbr_cls [8][nbr_stride] of pcd_rulebook_conv_cm_build_compact (entry
 * (j, v): the j-th usable offset of the class, permutation slot v) -- the pairs of an offset are then two plain arrays. */
int pcd_sparse_conv_wgrad_reduce(int kvol, int cin, int cout, int pmax, float *dweight,
                                 const void *workspace, void *stream);
/* The same reduction for up to PCD_WGRAD_MAX_JOBS layers in ONE launch (each layer then needs its own workspace
 * until the call; jobs_host is read during the call, the jobs travel as kernel arguments). */
#define PCD_WGRAD_MAX_JOBS 32
typedef struct PcdWgradReduceJob {
    const void *workspace;
    float *dweight;
    int kvol, cin, cout, pmax;
    int splits;              /* 0: as planned by pcd_sparse_conv_wgrad; > 0: slabs written by pcd_sparse_conv_wgrad_os */
    int layout;              /* of dweight: 0 = [cout][K][cin] (spconv weights), 1 = [cout][cin][K] (nn.Conv2d weights
                              * [cout, cin, 3, 3]: the dense 3x3 convs of the BEV stack write straight into .grad) */
    int cout_write;          /* 0 = cout; else only the first cout_write output channels are reduced and written (dweight
                              * holds cout_write rows): convs run with zero-padded output channels (the 1-3 channel
                              * final convs of the head towers, padded to 32) */
    int cin_write;           /* 0 = cin; else (layout 0, splits > 0) the slabs carry cin input channels per (co, k) and dweight
                              * only the first cin_write: a layer run on zero-padded input rows (conv_input 5 -> 16 on the window
                              * tiles: pcd_sparse_conv_subm_window_wgrad with c = 16) */
} PcdWgradReduceJob;
int pcd_sparse_conv_wgrad_reduce_batched(const PcdWgradReduceJob *jobs_host, int n_jobs, void *stream);
/* Output-stationary form for layers with 16 output channels (cin_pad 8 or 16, 3x3x3): walks the OUTPUT rows, reads dY in
 * order once and gathers only X through nbr_out [K][n_out] -- half the gathered rows of the pair form, which binds at
 * this width.  Writes pcd_sparse_conv_wgrad_os_splits(..) slabs [cout][K][cin] into `workspace` (splits * cout * K * cin
 * floats); reduce with pcd_sparse_conv_wgrad_reduce_batched (job.splits).  0 splits / PCD_ERR_UNSUPPORTED: not covered. */
int pcd_sparse_conv_wgrad_os_splits(int n_out_rows, int kvol, int cin_pad, int cout);
int pcd_sparse_conv_wgrad_os(const void *x, int n_x_rows, int cin_pad, int cin, const void *dy, int n_out_rows,
                             const int32_t *n_out_dev, int cout, const int32_t *nbr_out, int nbr_stride, int kvol,
                             void *workspace, size_t workspace_bytes, void *stream);

/* ============================================================================================
 * (a13/a14) BEV scatter -- replaces SparseConvTensor.dense() + the view in
 * pcdet/models/backbones_2d/map_to_bev/height_compression.py:20-25 and the per-batch loop of
 * pointpillar_scatter.py:17-37 (D == 1).   out [B][C*D][H][W], channel index c*D + z, every
 * element written (zeros included; no separate memset).  dtype: PCD_F32 or PCD_BF16 for both
 * features [n][c_stride] and out.  `gather` is the backward: dfeat[r][c] = dout[b][c*D+z][y][x].
 * workspace: pcd_bev_workspace_bytes(B, D, H, W) (a dense int32 row map).
 * ============================================================================================ */
size_t pcd_bev_workspace_bytes(int batch, int d, int h, int w);
int pcd_bev_scatter(const void *features, int c, int c_stride, int dtype, const int32_t *indices,
                    int n, const int32_t *n_dev, int batch, int d, int h, int w, void *out,
                    void *workspace, size_t workspace_bytes, void *stream);
int pcd_bev_gather(const void *dout, int c, int c_stride, int dtype, const int32_t *indices, int n,
                   const int32_t *n_dev, int batch, int d, int h, int w, void *dfeatures, void *stream);
/* channels-last variants: out / dout are [batch][h][w][c * d] (channel = c * d + z), i.e. the
 * torch.channels_last layout of the [batch, c * d, h, w] tensor HeightCompression returns
 * (height_compression.py:20-25) -- what MIOpen's bf16 convolutions of BaseBEVBackbone / CenterHead
 * (base_bev_backbone.py:30-112, center_head.py:75-99) consume without a layout change.  c % 8 == 0 (bf16) / 4 (f32),
 * d <= 8. */
int pcd_bev_scatter_nhwc(const void *features, int c, int c_stride, int dtype, const int32_t *indices, int n,
                         const int32_t *n_dev, int batch, int d, int h, int w, void *out, void *workspace,
                         size_t workspace_bytes, void *stream);
int pcd_bev_gather_nhwc(const void *dout, int c, int c_stride, int dtype, const int32_t *indices, int n,
                        const int32_t *n_dev, int batch, int d, int h, int w, void *dfeatures, void *stream);

/* ============================================================================================
 * (a11) fused sparse epilogue -- replaces the nn.BatchNorm1d(eps=1e-3, momentum=0.01) -> (+ residual)
 * -> nn.ReLU chains that SparseSequential / SparseBasicBlock apply to `.features` between convs
 * (pcdet/models/backbones_3d/spconv_backbone.py:21-25,50-66,73).
 *   forward : y = relu?( (x - mean) * invstd * gamma + beta + residual? ),  x/y/residual [n][c] of `dtype`;
 *             training != 0: batch statistics (biased var for normalisation, unbiased for running_var,
 *             running = (1-momentum)*running + momentum*batch), saved to save_mean / save_invstd [c];
 *             training == 0: running statistics.
 *   backward: dz = relu ? dy * (y > 0) : dy; dresidual = dz (may be NULL); dgamma, dbeta [c];
 *             dx = gamma*invstd*(dz - dbeta/n - xhat*dgamma/n)   (training) or gamma*invstd*dz (eval).
 *             y may be NULL when relu != 0 and the forward had NO residual: the mask is then recomputed from
 *             x, gamma, beta (required in that case), save_mean, save_invstd exactly as the forward computed
 *             it, which saves one [n][c] read in each of the two backward passes.
 * ext_partial (NULL = compute here): [ext_rows][2][c] sums already taken by a conv epilogue (PcdBnReduce); the
 *             statistics / reduction pass is then skipped.
 * c % 8 == 0 (bf16) / c % 4 == 0 (f32), c/piece a power of two <= 256.  Deterministic (no atomics).
 * ============================================================================================ */
size_t pcd_bn_workspace_bytes(int c);
/* out[c] = sum over the n rows of x [n][c] (bias gradient); workspace = pcd_bn_workspace_bytes(c). */
int pcd_col_sum(const void *x, int dtype, int n, int c, float *out, const int32_t *n_dev, void *workspace,
                size_t workspace_bytes, void *stream);
int pcd_bn_forward(const void *x, const void *residual, int dtype, int n, int c, const float *gamma,
                   const float *beta, float eps, float momentum, int training, float *running_mean,
                   float *running_var, int relu, void *y, float *save_mean, float *save_invstd,
                   const int32_t *n_dev, const float *ext_partial, int ext_rows, void *workspace,
                   size_t workspace_bytes, void *stream);
int pcd_bn_backward(const void *dy, const void *x, const void *y, int dtype, int n, int c,
                    const float *gamma, const float *beta, const float *save_mean, const float *save_invstd,
                    int relu, int training, void *dx, void *dresidual, float *dgamma, float *dbeta,
                    const int32_t *n_dev, const float *ext_partial, int ext_rows, float *colsum_partial,
                    void *workspace, size_t workspace_bytes, void *stream);
/* The same two calls for a BatchNorm whose OUTPUT (forward) / output GRADIENT (backward) is a column block of a wider
 * row-major matrix: y_ld / dy_ld = row stride in elements (>= c, a multiple of the 16-byte piece).  Lets the two
 * deblock BatchNorms of BaseBEVBackbone write straight into the concatenated map and read their halves of its
 * gradient (base_bev_backbone.py:103-108 torch.cat) without a copy either way. */
int pcd_bn_forward_ld(const void *x, const void *residual, int dtype, int n, int c, const float *gamma,
                      const float *beta, float eps, float momentum, int training, float *running_mean,
                      float *running_var, int relu, void *y, int y_ld, float *save_mean, float *save_invstd,
                      const int32_t *n_dev, const float *ext_partial, int ext_rows, void *workspace,
                      size_t workspace_bytes, void *stream);
int pcd_bn_backward_ld(const void *dy, int dy_ld, const void *x, const void *y, int dtype, int n, int c,
                       const float *gamma, const float *beta, const float *save_mean, const float *save_invstd,
                       int relu, int training, void *dx, void *dresidual, float *dgamma, float *dbeta,
                       const int32_t *n_dev, const float *ext_partial, int ext_rows, float *colsum_partial,
                       void *workspace, size_t workspace_bytes, void *stream);
/* colsum_partial (NULL = off): [pcd_bn_backward_colsum_rows(dtype, n, c)][c] f32, per-workgroup column sums of dx as
 * stored.  dx is dy of the conv in front of the BatchNorm, its column sum that conv's bias gradient
 * (spconv_backbone.py:37-44, bias=True inside SparseBasicBlock): pcd_col_sum_finalize adds the rows up in a fixed
 * order and replaces the pcd_col_sum pass over dx. */
int pcd_bn_backward_colsum_rows(int dtype, int n, int c);
/* up to PCD_COLSUM_MAX_JOBS column sums in ONE launch (jobs_host is read during the call; the jobs travel as kernel
 * arguments): a backbone finishes all its bias gradients at the end of the backward pass with a single kernel */
#define PCD_COLSUM_MAX_JOBS 32
typedef struct PcdColsumJob {
    const float *partial;   /* [rows][c] */
    float *out;             /* [c] */
    int rows, c;
} PcdColsumJob;
int pcd_col_sum_finalize(const PcdColsumJob *jobs_host, int n_jobs, void *stream);

/* ============================================================================================
 * (a15) update end of the data-parallel step: gradient-norm clipping + Adam on ONE flat fp32 buffer
 * (tools/train.py:165-166 DDP, tools/train_utils/train_utils.py:93-96 clip_grad_norm_ + optimizer.step();
 * torch.optim.Adam semantics: L2 weight decay added to the gradient, bias-corrected moments).
 *   grad holds the SUM of the ranks' gradients, pre_divisor = world size (1 on a single GPU): the update uses
 *   g = grad / pre_divisor * min(1, max_norm / (||grad / pre_divisor|| + 1e-6))   (max_norm <= 0: no clipping).
 *   step_dev[0] (device float) = number of updates done so far; incremented by the call.  norm_out (may be NULL)
 *   receives the gradient norm.  n % 4 == 0, buffers 16-byte aligned.  Two passes over the buffers, 3 launches.
 * ============================================================================================ */
size_t pcd_adam_flat_workspace_bytes(void);
int pcd_adam_flat_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, float max_norm, float pre_divisor,
                       float *step_dev, float *norm_out, void *workspace, size_t workspace_bytes, void *stream);
/* v2: decoupled_wd != 0 gives the reference's adam_onecycle rule -- OptimWrapper(true_wd=True, bn_wd=True),
 * tools/train_utils/optimization/__init__.py:19-32 + fastai_optim.py:135-150: every parameter is first multiplied
 * by (1 - lr * weight_decay), then Adam runs WITHOUT a weight-decay term; decoupled_wd == 0 is torch.optim.Adam's L2
 * form (v1).  hyper_dev (may be NULL): device float[2] = {lr, beta1} of THIS step, overriding the host arguments, so a
 * OneCycle schedule (learning_schedules_fastai.py:60-77: lr and MOMS vary every step) can drive a replayed hipGraph;
 * the bias correction uses the current beta1 like torch.optim.Adam does. */
int pcd_adam_flat_step_v2(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr,
                          float beta1, float beta2, float eps, float weight_decay, float max_norm, float pre_divisor,
                          int decoupled_wd, const float *hyper_dev, float *step_dev, float *norm_out, void *workspace,
                          size_t workspace_bytes, void *stream);
/* v3: the WHOLE OneCycle schedule as a device table schedule_dev[schedule_len][2] = {lr, beta1} per update
 * (learning_schedules_fastai.py:60-77 evaluated once on the host): the kernel reads row min(*step_dev, len - 1) itself
 * -- no lookup launches in front of the step -- and mirrors the pair into hyper_dev (may be NULL).  schedule_dev ==
 * NULL: v2 behaviour. */
int pcd_adam_flat_step_v3(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr,
                          float beta1, float beta2, float eps, float weight_decay, float max_norm, float pre_divisor,
                          int decoupled_wd, float *hyper_dev, const float *schedule_dev, int schedule_len,
                          float *step_dev, float *norm_out, void *workspace, size_t workspace_bytes, void *stream);
/* v3 + zero_grad: the gradient buffer is cleared as it is consumed (the next step's zero_grad without a fill launch). */
int pcd_adam_flat_step_v4(float *param, float *grad, int zero_grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr,
                          float beta1, float beta2, float eps, float weight_decay, float max_norm, float pre_divisor,
                          int decoupled_wd, float *hyper_dev, const float *schedule_dev, int schedule_len, float *step_dev,
                          float *norm_out, void *workspace, size_t workspace_bytes, void *stream);

/* ============================================================================================
 * (f2) CenterHead target assignment on the device -- replaces the per-object Python / CPU loop of
 *      pcdet/models/dense_heads/center_head.py:104-161 (assign_target_of_single_head) for one head, all batch
 *      elements at once (the caller's loop :163-225), incl. centernet_utils.py:46-107 (Gaussian radius, drawing).
 *   gt_boxes [batch][n_boxes][code] f32 (x, y, z, dx, dy, dz, heading, ..., class), class 0 = padding;
 *   class_map_host[n_class_map]: dataset class id -> 1-based id inside this head, 0 = not in this head;
 *   outputs: heatmap [batch][head_classes][fm_h][fm_w] f32, ret_boxes [batch][num_max_objs][code] f32,
 *   inds / mask [batch][num_max_objs] int64 (the reference's dtypes); all zero-filled by the call.
 * ============================================================================================ */
size_t pcd_centerhead_assign_workspace_bytes(int batch, int num_max_objs);
int pcd_centerhead_assign_targets(const float *gt_boxes, int batch, int n_boxes, int code_size, const int *class_map_host,
                                  int n_class_map, int head_classes, int fm_w, int fm_h, int feature_map_stride,
                                  const float *voxel_size_xy_host, const float *range_xy_host, int num_max_objs,
                                  float gaussian_overlap, int min_radius, float *heatmap, float *ret_boxes,
                                  long long *inds, long long *mask, void *workspace, size_t workspace_bytes, void *stream);

/* (f2) CenterHead.get_loss of one head (center_head.py:226-262): cls_weight * neg_loss_cornernet(clamp(sigmoid(hm)),
 *      heatmap) (loss_utils.py:611-643) + loc_weight * sum_d code_weights[d] * RegLossCenterNet_d (loss_utils.py:
 *      1317-1390) -- and its gradients -- in 2 + 2 launches instead of ~100 elementwise launches, without the
 *      reference's host round trips (`if num_pos == 0`, `.item()`).  Predictions are addressed through element
 *      strides {batch, channel, y, x} (NCHW or channels-last), dtype PCD_F32 / PCD_BF16; arithmetic in fp32.
 *   forward : out[0] = loss, [1] = hm_loss, [2] = loc_loss, [3] = mean confidence at the positives (nan without any,
 *             as the reference), [4] = num_pos, [5] = number of objects, [6 .. 6 + D) = L1 per code dimension
 *   backward: d_hm / reg_grads (same layout and dtype as the predictions; every element is written) for an upstream
 *             gradient *grad_out (device scalar).  `out` is the forward's.
 *   regression branches in HEAD_ORDER (their channels concatenated give the D code dimensions of target_boxes). */
size_t pcd_centerhead_loss_workspace_bytes(int code_dims);
int pcd_centerhead_loss_forward(const void *hm, int hm_dtype, const long long *hm_strides_host /*[4]*/,
                                const float *gt_heatmap, int batch, int num_classes, int height, int width,
                                const void *const *reg_ptrs_host, const int *reg_channels_host, int reg_dtype,
                                const long long *reg_strides_host /*[n_reg][4]*/, int n_reg, const long long *inds,
                                const long long *masks, const float *target_boxes, int num_max_objs,
                                const float *code_weights /*device [D]*/, float cls_weight, float loc_weight,
                                float *out /*device [6 + D]*/, void *workspace, size_t workspace_bytes, void *stream);
int pcd_centerhead_loss_backward(const void *hm, void *d_hm, int hm_dtype, const long long *hm_strides_host,
                                 const float *gt_heatmap, int batch, int num_classes, int height, int width,
                                 const void *const *reg_ptrs_host, void *const *reg_grads_host,
                                 const int *reg_channels_host, int reg_dtype, const long long *reg_strides_host,
                                 int n_reg, const long long *inds, const long long *masks, const float *target_boxes,
                                 int num_max_objs, const float *code_weights, float cls_weight, float loc_weight,
                                 const float *out, const float *grad_out, void *stream);

/* ============================================================================================
 * (f2, BASELINE config 3) The COM curriculum head on the device.  Replaces, for one head and the whole batch:
 *   pcd_com_cluster_groups   CurriculumCenterHead.cluster         pcdet/models/dense_heads/curriculum_center_head.py:414-459
 *   pcd_com_assign_targets   assign_targets / assign_target_of_single_head  same file :108-307 (+ centernet_utils.py:46-106)
 *   pcd_com_loss_forward     CurriculumCenterHead.get_loss of one head (:309-358) = FocalLossCenterCurriculum.neg_loss
 *                            (pcdet/utils/loss_utils.py:1178-1310) incl. confidence_of_all_groups (:1134-1176: the
 *                            conf_shape (3, 96) sums / counts a 288-iteration host loop produces every step), the
 *                            average-confidence EMA (:1214), the UCL per-object weights (:1231-1291, drawn into
 *                            heatmap_mask and box_mask in place as the reference does) + RegLossCenterNet with the
 *                            float box mask (:1317-1390)
 *   pcd_com_loss_backward    autograd of the above w.r.t. the heat-map logits and the regression maps
 * No host synchronisation anywhere (the reference: one .item() per object and per head, torch.where per group).
 *
 *   gt_boxes [B][M][code] f32, last column = 1-based class id (0 = padding); true_object / occupancy_ratio /
 *   facade_type / num_points_in_gt [B][M] f32 (load_data_to_gpu casts everything to float); group [B][M] i64.
 *   radius_map [B][num_max][cols] i64, cols = 5: (class in head, centre x, centre y, radius, group), 4 without groups.
 *   mask / box_mask [B][num_max] f32 (the reference's float `masks`), heatmap_mask [B][C][H][W] f32.
 *   owner [B][C][H][W] i32: scratch, zeroed ONCE by its owner (the kernels return it to zero); mask_sum [C][H][W] f32:
 *   written by the forward pass for the backward pass (both only touched when cur.ucl != 0; may be NULL otherwise).
 *   state double[2]: {EMA of the average confidence (in/out), this step's average confidence (out)}.
 *   out f32[6 + dims]: loss, hm_loss, loc_loss, avg_confidence, num_pos (mask-weighted), sum of box_mask, L1 per dim.
 *   conf_all / num_all f32[conf_classes * conf_groups]: this step's sums / counts; conf_epoch / num_epoch (may be NULL):
 *   += them (the epoch sums train_utils.py:111-112,208 builds from a Python list).
 * The [B, 1, C, H, W] mask broadcast of loss_utils.py:1293-1297 (every frame's term weighted by the frame-SUM of the
 * mask) is reproduced as it stands -- fixture G12 holds the reference's numbers. */
#define PCD_COM_CLUSTER_X5 0        /* CurriculumCenterHead.cluster (used by CurriculumCenterHead_x5, head_zoo.py:145-149) */
typedef struct PcdComCurriculum {   /* MODEL.DENSE_HEAD.LOSS_CURRICULUM as FocalLossCenterCurriculum.__init__ reads it */
    int ucl;                        /* UCL (default True) */
    int fix_threshold;              /* FIX */
    int straight;                   /* STRAIGHT */
    int tuning;                     /* TUNING */
    int only_center;                /* CENTER */
    int apply;                      /* START <= epoch <= END, evaluated by the caller */
    int add;                        /* ADD */
    int radius;                     /* RADIUS (0 = the object's own Gaussian radius + ADD) */
    double k_straight;              /* K */
    double elongation;              /* ELONGATION */
    double height;                  /* HEIGHT */
    double alpha;                   /* ALPHA */
    double threshold;               /* self.threshold: 0.5 (loss_utils.py:1054; the YAML's THRESHOLD key is not read) */
    int conf_classes, conf_groups;  /* conf_shape; 0, 0 = None */
} PcdComCurriculum;
int pcd_com_cluster_groups(const float *gt_boxes, int batch, int n_boxes, int code_size, const float *true_object,
                           const float *occupancy_ratio, const float *facade_type, int variant, long long *group,
                           void *stream);
size_t pcd_com_assign_workspace_bytes(int batch, int num_max_objs);
/* gate_min_points = (epoch <= EPOCH_THRED): objects with num_points_in_gt < min_points are skipped (:178-179);
 * class_map_host as for pcd_centerhead_assign_targets; group may be NULL (radius_map column 4 = 0) */
int pcd_com_assign_targets(const float *gt_boxes, int batch, int n_boxes, int code_size, const int *class_map_host,
                           int n_class_map, int head_classes, int fm_w, int fm_h, int feature_map_stride,
                           const float *voxel_size_xy_host, const float *range_xy_host, int num_max_objs,
                           float gaussian_overlap, int min_radius, const float *num_points_in_gt, const long long *group,
                           int gate_min_points, float min_points, float *heatmap, float *ret_boxes, long long *inds,
                           float *mask, long long *radius_map, int radius_map_cols, float *heatmap_mask, void *workspace,
                           size_t workspace_bytes, void *stream);
size_t pcd_com_loss_workspace_bytes(int batch, int num_max_objs);
int pcd_com_loss_forward(const void *hm, int hm_dtype, const long long *hm_strides_host, const float *gt_heatmap, int batch,
                         int num_classes, int height, int width, const void *const *reg_ptrs_host,
                         const int *reg_channels_host, int reg_dtype, const long long *reg_strides_host, int n_reg,
                         const long long *inds, float *box_mask, const float *target_boxes, const long long *radius_map,
                         int radius_map_cols, int num_max_objs, float *heatmap_mask, int32_t *owner, float *mask_sum,
                         const PcdComCurriculum *cur_host, const float *code_weights, float cls_weight, float loc_weight,
                         double *state, float *out, float *conf_all, float *num_all, float *conf_epoch, float *num_epoch,
                         void *workspace, size_t workspace_bytes, void *stream);
int pcd_com_loss_backward(const void *hm, void *d_hm, int hm_dtype, const long long *hm_strides_host,
                          const float *gt_heatmap, int batch, int num_classes, int height, int width,
                          const void *const *reg_ptrs_host, void *const *reg_grads_host, const int *reg_channels_host,
                          int reg_dtype, const long long *reg_strides_host, int n_reg, const long long *inds,
                          const float *box_mask, const float *target_boxes, int num_max_objs, const float *mask_sum,
                          const float *code_weights, float cls_weight, float loc_weight, const float *out,
                          const float *grad_out, void *stream);

/* ============================================================================================
 * (f4) PV-RCNN stage-2 natives -- the stacked-batch PointNet++ ops of pcdet/ops/pointnet2/pointnet2_stack (binder
 *      src/pointnet2_api.cpp; Python callers pointnet2_utils.py:8-303, voxel_query_utils.py:9-47).  "Stacked": the
 *      points of all batch elements are concatenated, *_batch_cnt[B] (device int32) give the counts.  Semantics are
 *      the reference kernels' (first nsample hits in ascending index, their tie rules); float32 / int32.
 *   pcd_ball_query_stack        ball_query_gpu.cu:16-66   idx [M][nsample]; row 0 = -1 when the ball is empty
 *   pcd_group_points_stack      group_points_gpu.cu:76-107 out [M][C][nsample] (+ _grad: scatter-add, atomics)
 *   pcd_stack_farthest_point_sampling  sampling_gpu.cu:188-327 (temp must be filled with 1e10 by the caller)
 *   pcd_three_nn_stack          interpolate_gpu.cu:16-76  dist2 [N][3] (squared), idx [N][3] (global row ids)
 *   pcd_three_interpolate_stack interpolate_gpu.cu:100-117 (+ _grad :140-155, atomics)
 *   pcd_voxel_query_stack       voxel_query_gpu.cu:10-88
 * ============================================================================================ */
int pcd_ball_query_stack(int B, int M, float radius, int nsample, const float *new_xyz, const int32_t *new_xyz_batch_cnt,
                         const float *xyz, const int32_t *xyz_batch_cnt, int32_t *idx, void *stream);
int pcd_group_points_stack(int B, int M, int C, int nsample, const float *features, const int32_t *features_batch_cnt,
                           const int32_t *idx, const int32_t *idx_batch_cnt, float *out, void *stream);
int pcd_group_points_stack_grad(int B, int M, int C, int nsample, const float *grad_out, const int32_t *idx,
                                const int32_t *idx_batch_cnt, const int32_t *features_batch_cnt,
                                float *grad_features_zeroed, void *stream);
int pcd_stack_farthest_point_sampling(int B, const float *xyz, float *temp_1e10, const int32_t *xyz_batch_cnt,
                                      int32_t *idxs, const int32_t *num_sampled_points, void *stream);
/* cooperative form for large frames (4096 keypoints of ~160 k raw points, voxel_set_abstraction.py:236-263): 256 / B (<= 64)
 * workgroups share a frame, same selected points (same total order of the argmax); PCD_ERR_UNSUPPORTED when B > 128 or a
 * frame's slice does not fit LDS -- fall back to the call above.  max_cnt_host >= every frame's point count. */
/* Bucket-pruned EXACT form for large frames (round 6; the default of com_amd.pointnet2_stack): the frame's points are binned
 * once along a Z-curve into buckets of 64..256 points that keep their bounding box and their farthest point; a new centre only visits the buckets
 * whose box is closer than their largest running distance -- provably the only ones that can change, in float arithmetic --
 * so an iteration touches ~5 k of 160 k points and needs no inter-workgroup exchange (one workgroup per frame).  Same selected
 * points as pcd_stack_farthest_point_sampling, ties included (sampling_gpu.cu:188-348).  total_points = rows of xyz;
 * max_cnt_host >= every frame's point count (PCD_ERR_UNSUPPORTED beyond 2048 x 256 points per frame). */
size_t pcd_stack_fps_buckets_workspace_bytes(int B, int total_points);
int pcd_stack_farthest_point_sampling_buckets(int B, const float *xyz, const int32_t *xyz_batch_cnt, int32_t *idxs,
                                              const int32_t *num_sampled_points, int total_points, int max_cnt_host,
                                              void *workspace, size_t workspace_bytes, void *stream);
size_t pcd_stack_fps_coop_workspace_bytes(int B);
int pcd_stack_farthest_point_sampling_coop(int B, const float *xyz, const int32_t *xyz_batch_cnt, int32_t *idxs,
                                           const int32_t *num_sampled_points, int max_cnt_host, void *workspace,
                                           size_t workspace_bytes, void *stream);
int pcd_three_nn_stack(int B, int N, const float *unknown, const int32_t *unknown_batch_cnt, const float *known,
                       const int32_t *known_batch_cnt, float *dist2, int32_t *idx, void *stream);
int pcd_three_interpolate_stack(int N, int C, const float *features, const int32_t *idx, const float *weight, float *out,
                                void *stream);
int pcd_three_interpolate_stack_grad(int N, int C, const float *grad_out, const int32_t *idx, const float *weight,
                                     float *grad_features_zeroed, void *stream);
int pcd_voxel_query_stack(int M, int R1, int R2, int R3, int nsample, float radius, int z_range, int y_range, int x_range,
                          const float *new_xyz, const float *xyz, const int32_t *new_coords, const int32_t *point_indices,
                          int32_t *idx, void *stream);

/* ============================================================================================
 * (g) fp8 feature path (BASELINE config 5; build-side precision, the reference is fp32): OCP e4m3 activations and
 *     weights with per-tensor scales, fp32 accumulation (v_mfma_f32_16x16x32_fp8_fp8), for the INFERENCE form of
 *     post_act_block (spconv_backbone.py:8-27): conv + eval BatchNorm + ReLU + quantisation of the next layer's
 *     input in ONE kernel.
 *   pcd_fp8_pack_weight: weight [Cout][K][Cin] f32 -> e4m3(weight * scale_inv) in MFMA fragment order;
 *       cin_pad = power of two >= 16 (the fp8 row width of the input features).
 *   pcd_fp8_quantize: x [n][c] (PCD_F32 / PCD_BF16, row stride c_stride) -> e4m3(x * scale_inv) [n][cb], zero padded.
 *   pcd_fp8_dequantize: count e4m3 bytes -> f32 * scale.
 *   pcd_sparse_conv_gather_gemm_fp8: y[o][c] = q( relu?( (sum_k x8[nbr[k][o]] . w8[k][:, c]) * alpha[c] + beta[c] ) ),
 *       alpha = x_scale * w_scale * gamma / sqrt(var + eps), beta = bn_beta - mean * gamma / sqrt(var + eps) (or any
 *       per-channel affine); y_kind 0: f32, 1: bf16, 2: e4m3(value * out_scale_inv); y_stride in elements.
 * ============================================================================================ */
size_t pcd_fp8_packed_weight_bytes(int kvol, int cin_pad, int cout);
int pcd_fp8_pack_weight(const float *weight, int kvol, int cin, int cin_pad, int cout, float scale_inv, void *packed,
                        void *stream);
int pcd_fp8_quantize(const void *x, int dtype, int n, const int32_t *n_dev, int c, int c_stride, int cb,
                     float scale_inv, void *out, void *stream);
int pcd_fp8_dequantize(const void *x8, size_t count, float scale, float *out, void *stream);
int pcd_sparse_conv_gather_gemm_fp8(const void *x8, int n_rows_in, int cin_pad, const void *packed_w, const int32_t *nbr,
                                    int nbr_stride, int kvol, int flip_k, int n_rows_out,
                                    const int32_t *n_rows_out_dev, int c_out, const float *alpha, const float *beta,
                                    int relu, float out_scale_inv, void *y, int y_kind, int y_stride, void *stream);

/* ============================================================================================
 * (a6) PillarVFE pieces -- pcdet/models/backbones_3d/vfe/pillar_vfe.py:94-118 (decoration + padding mask) and the
 *      ReLU / max-over-points / concatenation half of PFNLayer (pillar_vfe.py:40-49).  float32.
 *   pcd_pillar_decorate: voxels [m][T][C] (zero padded), num_points [m], coords [m][4] (b, z, y, x) ->
 *       out [m][T][C' ], C' = (use_absolute_xyz ? C : C - 3) + 6 (+ 1 with_distance): point features, xyz - pillar mean
 *       (sum over the T slots / num_points, not clamped), xyz - pillar centre (coord * voxel_size + offset, offset =
 *       voxel_size / 2 + range_min), optional |xyz|; padding slots (t >= num_points) zeroed.
 *   pcd_pfn_relu_pool: x [m][T][C] (BatchNorm / Linear output) -> last_layer: out [m][C] = max_t relu(x);
 *       else out [m][T][2C] = [relu(x), broadcast max].  arg [m][C] = first maximal slot (for the backward).
 *   pcd_pfn_relu_pool_backward: grad_x [m][T][C] from grad_out of that layout.
 * ============================================================================================ */
int pcd_pillar_decorate(const float *voxels, const int32_t *num_points, const int32_t *coords, int m, int T, int C,
                        int use_absolute_xyz, int with_distance, const float *voxel_size_host,
                        const float *offset_host, float *out, void *stream);
int pcd_pfn_relu_pool(const float *x, int m, int T, int C, int last_layer, float *out, int32_t *arg, void *stream);
int pcd_pfn_relu_pool_backward(const float *grad_out, const float *x, const int32_t *arg, int m, int T, int C,
                               int last_layer, float *grad_x, void *stream);

/* ============================================================================================
 * (f3) Rotated BEV overlap / IoU and NMS -- replaces pcdet/ops/iou3d_nms (binder: src/iou3d_nms_api.cpp:12-16;
 *      kernels src/iou3d_nms_kernel.cu:236-413; host reduction src/iou3d_nms.cpp:60-188; Python callers
 *      iou3d_nms_utils.py:31-116).  Boxes are rows of 7 float32 (x, y, z, dx, dy, dz, heading).
 *   pcd_boxes_overlap_bev: out[a][b] = area of the BEV intersection (want_iou == 0; boxes_overlap_bev_gpu) or
 *                          the BEV IoU (want_iou != 0; boxes_iou_bev_gpu), out = [num_a][num_b] f32.
 *   pcd_nms_bev: greedy NMS over boxes ALREADY SORTED by descending score (as nms_gpu / nms_normal_gpu receive them):
 *                box i suppresses every later box whose IoU with it exceeds thresh (normal != 0: axis-aligned IoU of
 *                nms_normal_gpu).  keep = int64[num_boxes] receives the kept indices in ascending order,
 *                *num_keep_dev (device int32) their number; both stay on the device (the reference returns the
 *                count through a host loop over a mask it copies back).  workspace: pcd_nms_workspace_bytes(n).
 *   pcd_boxes_iou_bev_host: the reference's CPU variant boxes_iou_bev_cpu (iou3d_nms_utils.py:12-28 ->
 *                src/iou3d_cpu.cpp:232-252), called by COMAug's database sampler per frame inside DataLoader workers
 *                (datasets/augmentor/database_sampler_v2.py:600-601).  HOST pointers, synchronous, touches no GPU state
 *                (safe in forked workers); out_host = [num_a][num_b] f32 IoU.  Same float arithmetic as the reference
 *                file, bit for bit (fixture G13).
 * ============================================================================================ */
int pcd_boxes_iou_bev_host(const float *boxes_a_host, int num_a, const float *boxes_b_host, int num_b,
                           float *out_host);
int pcd_boxes_overlap_bev(const float *boxes_a, int num_a, const float *boxes_b, int num_b, float *out, int want_iou,
                          void *stream);
size_t pcd_nms_workspace_bytes(int num_boxes);
int pcd_nms_bev(const float *boxes, int num_boxes, float thresh, int normal, long long *keep, int32_t *num_keep_dev,
                void *workspace, size_t workspace_bytes, void *stream);

/* ============================================================================================
 * Static-shape execution guard.  Buffers of a captured step are allocated at CAPACITIES (see "Device-side row
 * counts" above) and the kernels clamp to them, so a batch denser than the capacity would be truncated silently.
 * pcd_static_overflow_check enqueues a one-thread kernel that compares up to PCD_COUNT_CHECK_MAX device-side counts
 * with their capacities: flag[0] |= 1 (sticky) and flag[1] = max(flag[1], count - capacity) on overflow.  The
 * table is passed by value (host struct); flag is a device int32[2] the caller zeroes once and polls.
 * ============================================================================================ */
#define PCD_COUNT_CHECK_MAX 24
typedef struct PcdCountCheck {
    const int32_t *count[PCD_COUNT_CHECK_MAX];   /* device pointers to the counts */
    int32_t cap[PCD_COUNT_CHECK_MAX];            /* capacities                      */
} PcdCountCheck;
int pcd_static_overflow_check(const PcdCountCheck *table_host, int n, int32_t *flag, void *stream);

/* ============================================================================================
 * fp32-exact forms of the sparse convolution (features, weights, results in fp32; v_mfma_f32_16x16x4_f32).  The
 * reference runs fp32 end to end (spconv with fp32 weights behind spconv_backbone.py:12-15, pcdet/utils/spconv_utils.py:3-6);
 * these entry points exist for PARITY work -- reproducing a reference checkpoint's activations to fp32 accuracy and
 * checking a whole backbone end to end at 1e-3 without bf16 rounding noise -- not for speed.
 *   weight  [c_out][kvol][c_in] f32 (the parameter layout, no packing); for a data gradient pass the transposed
 *           weight [c_in][kvol][c_out] as `weight` and dy as `x` (flip_k as pcd_sparse_conv_gather_gemm)
 *   wgrad   dweight [c_out][kvol][c_in], pairs in canonical order; fixed summation order (deterministic)
 * c_out <= 128 per call. */
int pcd_sparse_conv_gather_gemm_f32(const float *x, int n_rows_in, int c_in, const float *weight, const float *bias,
                                    const int32_t *nbr, int nbr_stride, int kvol, int flip_k, int n_rows_out,
                                    const int32_t *n_rows_out_dev, int c_out, float *y, const float *addend,
                                    void *stream);
int pcd_sparse_conv_wgrad_f32(const float *x, int n_x_rows, int c_in, const float *dy, int n_dy_rows, int c_out,
                              const int32_t *pairs, const int32_t *pair_num, int kvol, int pmax, float *dweight,
                              void *stream);

/* ============================================================================================
 * (f1) Dense 3x3 convolution, stride 1, padding 1, over channels-last bf16 maps: the nn.Conv2d(k = 3, padding = 1) layers
 *      of BaseBEVBackbone / CenterHead (pcdet/models/backbones_2d/base_bev_backbone.py:30-112,
 *      dense_heads/center_head.py:11-46; MIOpen in the reference).  Implicit GEMM on MFMA, fp32 accumulate.
 *   weight  [cout][cin][3][3] f32 (torch OIHW); pcd_conv2d_pack_weight(mode 0) -> forward pack,
 *           (mode 1) -> data-gradient pack: run the SAME conv entry point on dy with cin / cout swapped
 *   x, y    [batch][height][width][channels] bf16 (torch.channels_last storage of an NCHW tensor)
 * cin % 32 == 0; packs pad the output channels with zeros to a multiple of 32 (run the conv with that padded count:
 * the data gradient contracts over it in steps of 32). */
size_t pcd_conv2d_packed_weight_bytes(int cin, int cout, int mode);
int pcd_conv2d_pack_weight(const float *weight, int cin, int cout, int mode, void *packed, void *stream);
/* all packs of a model in one launch: `table` = n device rows of 8 int64 {weight ptr, packed ptr, cin, cout,
 * cout padded to 32, mode, first workgroup, 16-byte pieces = packed bytes / 16}, total_blocks = sum of
 * ceil(pieces / 256) (rows ordered by first workgroup) */
int pcd_conv2d_pack_weights_batched(const void *table, int n, int total_blocks, void *stream);
int pcd_conv2d_3x3_nhwc(const void *x, int batch, int height, int width, int cin, const void *packed_w, int cout,
                        const float *bias, void *y, void *stream);
/* the same conv on CHANNEL BLOCKS of wider maps: x_cs / y_cs = channels per pixel of the buffers x / y point into
 * (>= cin / cout, multiples of 8): the five branches of a SeparateHead (center_head.py:11-46) read their 64 channels of
 * one 320-channel activation and their data gradients fill its gradient block by block -- no slice copies, no adds */
int pcd_conv2d_3x3_nhwc_ld(const void *x, int x_cs, int batch, int height, int width, int cin, const void *packed_w,
                           int cout, const float *bias, void *y, int y_cs, void *stream);
/* ... and with the BatchNorm sums of the sparse path's PcdBnReduce taken in its epilogue (y_cs == cout required): mode 1 on
 * the forward launch for the BatchNorm behind the conv, mode 2 on the data-gradient launch for the BatchNorm whose output
 * was the conv's input (x / y = that BatchNorm's input / output as [pixels][cout] bf16).  partial_rows must equal
 * pcd_conv2d_3x3_tiles(batch, height, width); hand partial / mid to pcd_bn_forward / _backward as ext_partial. */
int pcd_conv2d_3x3_tiles(int batch, int height, int width);
int pcd_conv2d_3x3_nhwc_bn(const void *x, int x_cs, int batch, int height, int width, int cin, const void *packed_w,
                           int cout, const float *bias, void *y, int y_cs, const PcdBnReduce *bn_reduce, void *stream);
/* Weight gradient of that conv without pair lists: x [b][h][w][cin] (pixel stride x_cs channels), dy [b][h][w][cout]
 * contiguous, cin % 64 == 0, cout % 32 == 0 (zero-padded output channels allowed).  Writes
 * pcd_conv2d_wgrad_3x3_splits(..) slabs [cout][9][cin] f32 into `slabs`; finish with
 * pcd_sparse_conv_wgrad_reduce_batched (job.kvol = 9, job.splits = that count, job.pmax = 1, layout 1 for an
 * nn.Conv2d .grad, cout_write for padded convs).  0 splits / PCD_ERR_UNSUPPORTED: shape not covered (use the pair form). */
int pcd_conv2d_wgrad_3x3_splits(int batch, int height, int width, int cin, int cout);
int pcd_conv2d_wgrad_3x3_nhwc(const void *x, int x_cs, const void *dy, int batch, int height, int width, int cin, int cout,
                              void *slabs, size_t slab_bytes, void *stream);
/* ... and of the plane operators below (forward pack modes 2 / 4 / 6): `fine` / `coarse` = the layer's two maps (mode 2: x / dy;
 * modes 4, 6: dy / x), contiguous bf16; slabs [cc][k * k][cf] f32 -> pcd_sparse_conv_wgrad_reduce_batched (kvol = k * k,
 * cin = cf, cout = cc, layout 1 = the torch parameter's layout for Conv2d AND ConvTranspose2d).  cf % 64 == 0, cc % 32 == 0. */
int pcd_conv2d_wgrad_planes_splits(int mode, int batch, int hc, int wc, int cf, int cc);
int pcd_conv2d_wgrad_planes_nhwc(int mode, const void *fine, int hf, int wf, int cf, const void *coarse, int batch, int hc,
                                 int wc, int cc, void *slabs, size_t slab_bytes, void *stream);
/* The other three layers of BaseBEVBackbone (base_bev_backbone.py:36-75; MIOpen in the reference), forward and data
 * gradient, as per-parity-plane stencils on the same tiles.  pack modes (pcd_conv2d_pack_weight / _packed_weight_bytes /
 * the batched table take them too; cin / cout are the LAYER's channel counts, the weight is the torch parameter):
 *   2  Conv2d(cin, cout, 3, stride 2, padding 1) forward          x [b][hi][wi][cin]  -> y [b][ho][wo][cout], ho = (hi-1)/2+1
 *   3  its data gradient                                           dy [b][hi][wi][cout] -> dx [b][ho][wo][cin], hi = (ho-1)/2+1
 *   4  ConvTranspose2d(cin, cout, 2, stride 2) forward             x [b][hi][wi][cin]  -> y [b][2hi][2wi][cout]
 *   5  its data gradient                                           dy [b][hi][wi][cout] -> dx [b][hi/2][wi/2][cin]
 *   6 / 7  ConvTranspose2d(cin, cout, 1, stride 1) forward / data gradient (a 1 x 1 conv)
 * `cin` / `cout` of THIS call are the channel counts of x and y (i.e. swapped for the data gradients); both % 32 == 0. */
int pcd_conv2d_planes_nhwc(int pack_mode, const void *x, int batch, int hi, int wi, int cin, const void *packed_w,
                           int cout, const float *bias, void *y, int ho, int wo, void *stream);

/* (a3, dataset.py:252-259 + pcdet/models/__init__.py:23-34: collate + .cuda()) the host -> device hop INSIDE a captured step:
 * `host_ptr_table_dev` = n_slots device-side entries holding the addresses of PINNED host buffers (each >= bytes); the kernel
 * copies buffer (*counter_dev % n_slots) into dst over PCIe.  No copy engine, no second stream, no host call per step: a
 * replayed hipGraph pulls a different batch every time because the counter (pcd_counter_add, also a graph node) moves.
 * bytes % 16 == 0, dst 16-byte aligned; `workgroups` <= 0: 64. */
int pcd_pull_from_host(const void *host_ptr_table_dev, int n_slots, const int32_t *counter_dev, void *dst, size_t bytes,
                       int workgroups, void *stream);
int pcd_counter_add(int32_t *counter_dev, int delta, void *stream);

/* <a, b> of two bf16 vectors (fp32 products, fp64 partial sums) -> out[0], and y = bf16(a * scale_dev[0]): forward and
 * backward of a fixed linear functional of the BEV map -- bench.py's stand-in for the dense head's loss when only the
 * sparse hot path is timed (no reference counterpart).  n % 8 == 0, 16-byte aligned. */
size_t pcd_dot_bf16_workspace_bytes(void);
int pcd_dot_bf16(const void *a, const void *b, size_t n, float *out, void *workspace, size_t workspace_bytes, void *stream);
int pcd_scale_bf16(const void *a, const float *scale_dev, size_t n, void *y, void *stream);

/* Diagnostics: the device clock (100 MHz) into slot[0] at this point of the stream -- a time point inside a replayed
 * hipGraph, which events cannot give and a profiler perturbs (tools/exp_stamps.py).  No reference counterpart. */
int pcd_debug_stamp(uint64_t *slot, void *stream);
/* Diagnostics (tools/exp_cu_mask.py): a stream restricted to the compute units of `cu_mask` (hipExtStreamCreateWithCUMask;
 * the caller owns it), and a launch of `blocks` 1024-thread workgroups that each stay busy for `ticks` ticks of the 100 MHz
 * clock (xcc_seen, optional: bit x set when a workgroup ran on XCD x) -- the duration of 256 of them tells how many CUs the
 * launch was given, eagerly and as a replayed hipGraph node.  No reference counterpart. */
int pcd_debug_stream_create_cu_mask(const uint32_t *cu_mask, int words, void **stream_out);
int pcd_debug_spin(int blocks, unsigned long long ticks, uint32_t *xcc_seen, void *stream);
int pcd_debug_spin_shape(int blocks, int threads, int lds_bytes, int vgprs, unsigned long long ticks, void *stream);

/* ============================================================================================
 * (a8) SubMConv3d arithmetic over z-fastest rows: the WINDOW gather-GEMM (spconv_win.hip) -- forward and data gradient of
 *      spconv.SubMConv3d (spconv_backbone.py:12,38-45) for 3x3x3 kernels with c_in == c_out in {16, 32, 64, 128}, bf16 features.
 * Same operation as pcd_sparse_conv_gather_gemm on a SubM neighbour table: y[o] = bias + sum_k W_k x[nbr[k][o]] (+ addend)
 * with weights packed in mode 0; with weights packed in mode 1 the data gradient dx[i] = sum_k W_k^T dy[nbr[26 - k][i]]
 * (+ addend) -- the k flip of the rulebook view is folded into that pack, the launch has no flip argument.  Results agree
 * with the generic kernels up to the fp32 summation order.  It is FAST when the rows are numbered
 * PCD_ROWS_YXZ (the 27 neighbours of a tile of consecutive rows then lie in three short runs of rows that are staged in LDS
 * once per tile) and correct, only slow, for any other numbering.
 *   pcd_subm_window_tile_rows      rows per tile T of the (c_in, c_out) kernel; 0 = no window kernel for these widths
 *   pcd_subm_window_plan           per tile of T rows the three runs of neighbour rows + the tile's table of LDS operand slots
 *                                  (u16, 32 per row): plan = pcd_subm_window_plan_bytes(n, c_in, c_out) bytes, built once per rulebook
 *                                  (every conv of the indice_key, forward and backward, uses it)
 *   pcd_subm_window_pack_weight    weight [c_out][27][c_in] f32 -> the kernel's register-resident slices (mode 0 forward,
 *                                  1 data gradient: transposed, offsets reversed), pcd_subm_window_packed_weight_bytes bytes;
 *                                  _batched: table rows of 8 x i64
 *                                  {weight ptr, packed ptr, c_in, mode, first 256-thread block, 0, 0, 0}
 *   pcd_subm_window_partial_rows   rows of PcdBnReduce.partial a launch of these widths writes (one per persistent workgroup: 256, or 512
 *                                  for the 4-wave configurations of option "subm_window_half")
 * ============================================================================================ */
int pcd_subm_window_tile_rows(int c_in, int c_out);
int pcd_subm_window_partial_rows(int c_in, int c_out);
/* profiling aid: a device buffer of 1024 x u64 whose first 256 entries receive shader-clock stamps of workgroup 0 at the phase boundaries of its
 * tiles (7 per tile: barrier, prefetch issued, MFMA loop done, prefetch landed, barrier, partial sums written + barrier,
 * epilogue done); NULL (the default) = off.  Process-wide; tools/exp_subm_win.py */
int pcd_subm_window_set_trace(void *buf256_u64);
size_t pcd_subm_window_plan_bytes(int n_cap, int c_in, int c_out);
int pcd_subm_window_plan(const int32_t *nbr, int nbr_stride, int n_cap, const int32_t *n_dev, int c_in, int c_out,
                         void *plan, void *stream);
/* The same plan built STRAIGHT from the level's column map (rows numbered PCD_ROWS_YXZ; `indices` [n_cap][4], colmap as handed
 * out by pcd_voxelize_hard_yxz / pcd_rulebook_conv_cm_* / pcd_colmap_from_rows) -- what pcd_rulebook_subm_cm + pcd_subm_window_plan
 * produce together, in ONE pass and without the 27 x 4 B per row of neighbour table in between (spconv_backbone.py:38-45,199-218:
 * the SubM rulebooks of levels whose convs all run on window tiles).  nbr [27][n_cap] must be a buffer of that size:
 *   nbr_full != 0  the complete table is written too (bit-identical to pcd_rulebook_subm_cm's), for consumers that read it;
 *   nbr_full == 0  only the columns of tiles whose run exceeds the window (header.passes > 1) are written -- all that
 *                  pcd_sparse_conv_subm_window / _wgrad ever read of it; the rest of the buffer is left untouched. */
int pcd_subm_window_plan_cm(const int32_t *indices, int n_cap, const int32_t *n_dev, int batch, const int *shape_host,
                            const void *colmap, size_t colmap_bytes, int colmap_cap, int c_in, int c_out, int32_t *nbr,
                            int nbr_full, void *plan, void *stream);
size_t pcd_subm_window_packed_weight_bytes(int c_in, int c_out);
int pcd_subm_window_pack_weight(const float *weight, int c_in, int c_out, int mode, void *packed, void *stream);
int pcd_subm_window_pack_weights_batched(const void *table, int n, int total_blocks, void *stream);
/* Weight gradient over the same tiles: slab[s][c_out][27][c_in] f32 for s < pcd_subm_window_wgrad_splits() partial sums
 * (slab_bytes >= splits * 27 * c * c * 4), to be summed over s in order -- pcd_sparse_conv_wgrad_reduce_batched with
 * job.splits = that count does it (same job as pcd_sparse_conv_wgrad_os).  x, dy: bf16 [n_rows][c]; nbr / plan as above. */
int pcd_subm_window_wgrad_splits(int c);
int pcd_sparse_conv_subm_window_wgrad(const void *x, const void *dy, int n_rows, int c, const int32_t *nbr, int nbr_stride,
                                      const int32_t *n_rows_dev, const void *plan, void *slab, size_t slab_bytes,
                                      void *stream);
int pcd_sparse_conv_subm_window(const void *x, int n_rows, int c_in, const void *packed_w, const float *bias,
                                const int32_t *nbr, int nbr_stride, const int32_t *n_rows_dev,
                                const void *plan, int c_out, void *y, const void *addend,
                                const PcdBnReduce *bn_reduce, void *stream);
/* The same launch that ALSO writes y_f32 [n_rows][c_out]: the fp32 sums (bias and addend included) every bf16 output is
 * rounded from -- what the parity tests compare with the oracle at the 1e-3 bar of BASELINE.json (the bf16 rounding of y
 * alone is 2^-9 relative per element).  Not used by the training step. */
int pcd_sparse_conv_subm_window_f32(const void *x, int n_rows, int c_in, const void *packed_w, const float *bias,
                                    const int32_t *nbr, int nbr_stride, const int32_t *n_rows_dev, const void *plan,
                                    int c_out, void *y, float *y_f32, const void *addend, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PCD_OPS_H_ */
