// The COM curriculum head on the device (BASELINE config 3): what the reference does per training step
// with Python loops and host round trips --
//   * CurriculumCenterHead.cluster                  pcdet/models/dense_heads/curriculum_center_head.py:414-459
//   * assign_targets / assign_target_of_single_head  same file :108-307 (per-object loop on CPU tensors, .item() per box)
//   * FocalLossCenterCurriculum.neg_loss             pcdet/utils/loss_utils.py:1178-1310, with
//       confidence_of_all_groups (:1134-1176): a 288-iteration loop of torch.where + gather + len(), every step;
//       avg_confidence.item() (:1214, :1310); the UCL per-object loop (.item() x 5 per object, :1231-1291)
//   * RegLossCenterNet with the float box mask       pcdet/utils/loss_utils.py:1317-1390
//   * get_loss                                       curriculum_center_head.py:309-358
// -- as a handful of launches with no host synchronisation, so the whole head sits inside the captured step.
//
// Arithmetic follows the reference: float32 wherever a value feeds an index, a count or a comparison (cluster bins,
// centres, radii), float64 accumulation for sums (the reference sums float32 tensors in torch's order; agreed tolerance
// 1e-6), Python-float (double) arithmetic for the UCL weights and the confidence EMA.
//
// One reference quirk is reproduced on purpose: neg_loss multiplies the [B, C, H, W] focal terms by
// `mask[:, None, :, :]` where mask is heatmap_mask [B, C, H, W] -- a [B, 1, C, H, W] tensor, so the product broadcasts to
// [B, B, C, H, W]: every frame's term at (c, y, x) is weighted by the SUM OVER FRAMES of the mask at (c, y, x), and
// num_pos likewise.  With an all-ones mask (UCL False) numerator and denominator both pick up a factor B (which cancels,
// except in the `num_pos == 0` branch, where the loss is B x the plain one).  Fixture G12 holds the reference's numbers.
#include "centerhead_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// cluster(): difficulty group per ground-truth box.  Scalars are compared in float32 (torch casts a Python number to
// the tensor's dtype), hence the (float)(double expression) constants.
__global__ __launch_bounds__(256) void com_cluster_kernel(const float *__restrict__ gt, int total, int code,
                                                          const float *__restrict__ true_object,
                                                          const float *__restrict__ occupancy,
                                                          const float *__restrict__ facade,
                                                          long long *__restrict__ group) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float *q = gt + (size_t)i * code;
    const float x = q[0], y = q[1];
    const float dist = sqrtf(x * x + y * y);                 // torch.sqrt(pow(x, 2) + pow(y, 2)), float32 (no fma: -ffp-contract=off)
    const float length = q[3], cls = q[code - 1];
    const float occ = occupancy[i], fac = facade[i];
    const int dbin = dist <= 30.0f ? 0 : (dist <= 50.0f ? 1 : 2);
    long long g = 0;
    if (true_object[i] == 1.0f) {
        if (cls == 1.0f) {
            const int lbin = length <= 6.0f ? 0 : 1;
            const int fbin = fac == 3.0f ? 0 : (fac == 2.0f ? 1 : (fac == 1.0f ? 2 : (fac == 0.0f ? 3 : -1)));
            const int obin = occ > 0.7f ? 0 : (occ > 0.5f ? 1 : (occ > 0.25f ? 2 : (occ <= 0.25f ? 3 : -1)));
            if (fbin >= 0 && obin >= 0) g = 1 + ((dbin * 2 + lbin) * 4 + fbin) * 4 + obin;
        } else if (cls == 2.0f || cls == 3.0f) {
            const float t0 = (float)(0.21 * 5 / 12), t1 = (float)(0.41 * 5 / 12), t2 = (float)(0.61 * 5 / 12),
                        t3 = (float)(0.81 * 5 / 12);
            const int obin = occ > t3 ? 0 : (occ > t2 ? 1 : (occ > t1 ? 2 : (occ > t0 ? 3 : (occ <= t0 ? 4 : -1))));
            if (obin >= 0) g = 1 + dbin * 5 + obin;
        }
    }
    group[i] = g;
}

// ---------------------------------------------------------------------------------------------------------------
// targets: one launch initialises every output, one wave per frame fills the object rows, one wave per object draws
struct ComInit {
    float *heatmap, *heatmap_mask, *ret_boxes, *mask;
    long long *inds, *radius_map;
    int4 *draw;
    size_t n_heat, n_boxes, n_obj, n_rmap;        // element counts
};

__global__ __launch_bounds__(256) void com_targets_init_kernel(ComInit I) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (size_t i = i0; i < I.n_heat; i += stride) {
        I.heatmap[i] = 0.0f;
        I.heatmap_mask[i] = 1.0f;
    }
    for (size_t i = i0; i < I.n_boxes; i += stride) I.ret_boxes[i] = 0.0f;
    for (size_t i = i0; i < I.n_obj; i += stride) {
        I.mask[i] = 0.0f;
        I.inds[i] = 0;
        I.draw[i] = make_int4(-1, 0, 0, 0);
    }
    for (size_t i = i0; i < I.n_rmap; i += stride) I.radius_map[i] = 0;
}

__global__ __launch_bounds__(64) void com_assign_rows_kernel(const float *__restrict__ gt, int n, AssignGeom G,
                                                             const float *__restrict__ npgt,
                                                             const long long *__restrict__ group, int gate,
                                                             float min_points, int rmap_cols,
                                                             float *__restrict__ ret_boxes, long long *__restrict__ inds,
                                                             float *__restrict__ mask, long long *__restrict__ radius_map,
                                                             int4 *__restrict__ draw) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float *rows = gt + (size_t)b * n * G.code;
    int count = 0;
    for (int base = 0; base < n && count < G.num_max; base += 64) {
        const int r = base + lane;
        int local = 0;
        if (r < n) {
            const int cls = (int)rows[(size_t)r * G.code + G.code - 1];
            local = (cls >= 0 && cls < 16) ? G.cls_map[cls] : 0;
        }
        int total;
        const int k = count + wave_rank(local > 0, total);
        count += total;
        if (local <= 0 || k >= G.num_max) continue;
        const float *q = rows + (size_t)r * G.code;
        float cx = (q[0] - G.range_x) / G.vs_x / (float)G.stride;
        float cy = (q[1] - G.range_y) / G.vs_y / (float)G.stride;
        cx = fminf(fmaxf(cx, 0.0f), (float)G.W - 0.5f);
        cy = fminf(fmaxf(cy, 0.0f), (float)G.H - 0.5f);
        const int ix = (int)cx, iy = (int)cy;
        const float dx = q[3] / G.vs_x / (float)G.stride, dy = q[4] / G.vs_y / (float)G.stride;
        if (!(dx > 0.0f && dy > 0.0f && ix >= 0 && ix <= G.W && iy >= 0 && iy <= G.H)) continue;   // :168-172
        if (gate && npgt[(size_t)b * n + r] < min_points) continue;                                  // :178-179
        int radius = (int)gaussian_radius_f32(dx, dy, G.overlap);
        radius = radius < G.min_radius ? G.min_radius : radius;
        const size_t at = (size_t)b * G.num_max + k;
        inds[at] = (long long)iy * G.W + ix;
        mask[at] = 1.0f;
        float *o = ret_boxes + at * G.code;
        o[0] = cx - (float)ix;
        o[1] = cy - (float)iy;
        o[2] = q[2];
        o[3] = logf(q[3]);
        o[4] = logf(q[4]);
        o[5] = logf(q[5]);
        o[6] = cosf(q[6]);
        o[7] = sinf(q[6]);
        for (int j = 8; j < G.code; ++j) o[j] = q[j - 1];
        long long *rm = radius_map + at * rmap_cols;
        rm[0] = local - 1;
        rm[1] = ix;
        rm[2] = iy;
        rm[3] = radius;
        if (rmap_cols > 4) rm[4] = group ? group[(size_t)b * n + r] : 0;
        draw[at] = make_int4(local - 1, ix, iy, radius);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// loss
struct ComCur {                     // LOSS_CURRICULUM as the kernels see it (PcdComCurriculum + derived)
    int ucl, fix, straight, tuning, only_center, apply, add, radius;
    double k_straight, elongation, height, alpha, threshold;
    int conf_c, conf_g;
};

constexpr int COM_BLOCKS = 256;
constexpr int COM_PSTRIDE = 8 + CHL_MAX_DIM + 2;    // doubles per partial row

__device__ __forceinline__ float com_pred(const ChlMap &hm, int b, int c, int y, int x) {
    const float s = chl_sigmoid(chl_load(hm, b * hm.sb + c * hm.sc + y * hm.sh + x * hm.sw));
    return fminf(fmaxf(s, 1e-4f), 1.0f - 1e-4f);
}

// (UCL with the EMA threshold only) sum of pred at the positives / number of positives -> partial[blk][6..7]
__global__ __launch_bounds__(256) void com_conf_partials_kernel(ChlMap hm, const float *__restrict__ gt, int B, int C,
                                                                int H, int W, double *__restrict__ partial) {
    __shared__ double lds[4];
    const unsigned total = (unsigned)B * C * H * W;
    double conf = 0.0, nobj = 0.0;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += COM_BLOCKS * 256u) {
        if (gt[e] != 1.0f) continue;
        const int x = (int)(e % (unsigned)W);
        unsigned t = e / (unsigned)W;
        const int y = (int)(t % (unsigned)H);
        t /= (unsigned)H;
        conf += (double)com_pred(hm, (int)(t / (unsigned)C), (int)(t % (unsigned)C), y, x);
        nobj += 1.0;
    }
    const double s0 = chl_block_sum(conf, lds), s1 = chl_block_sum(nobj, lds);
    if (threadIdx.x == 0) {
        partial[(size_t)blockIdx.x * COM_PSTRIDE + 6] = s0;
        partial[(size_t)blockIdx.x * COM_PSTRIDE + 7] = s1;
    }
}

// state[0] = EMA of the average confidence (double, loss_utils.py:1214); state[1] = this step's average confidence
__global__ __launch_bounds__(256) void com_ema_kernel(const double *__restrict__ partial, double alpha,
                                                      double *__restrict__ state) {
    __shared__ double lds[4];
    const double c = chl_block_sum(partial[(size_t)threadIdx.x * COM_PSTRIDE + 6], lds);
    const double n = chl_block_sum(partial[(size_t)threadIdx.x * COM_PSTRIDE + 7], lds);
    if (threadIdx.x == 0) {
        const float avg = (float)c / (float)n;                      // float32 tensor division, then .item()
        state[1] = (double)avg;
        state[0] = alpha * (double)avg + (1.0 - alpha) * state[0];
    }
}

// one wave per object slot: the UCL weight (loss_utils.py:1247-1280), box_mask, and the mask's OWNER: later objects
// overwrite earlier ones in the reference's frame-major, slot-ascending loop, so a pixel belongs to the largest slot
// index covering it (atomicMax over slot + 1), frames being independent planes.
__device__ __forceinline__ bool com_rect(const ComCur &cur, const long long *rm, int W, int H, int &c, int &cx, int &cy,
                                         int &x0, int &y0, int &w, int &h) {
    if (rm[3] <= 0) return false;                                   // nonzero_idx = where(radius_map[b][:, 3] > 0)
    c = (int)rm[0];
    cx = (int)rm[1];
    cy = (int)rm[2];
    if (cur.only_center) {
        x0 = cx; y0 = cy; w = 1; h = 1;
        return true;
    }
    const int radius = cur.radius != 0 ? cur.radius : (int)rm[3] + cur.add;
    const int left = min(cx, radius), right = min(W - cx, radius + 1);
    const int top = min(cy, radius), bottom = min(H - cy, radius + 1);
    x0 = cx - left; y0 = cy - top; w = left + right; h = top + bottom;
    return w > 0 && h > 0;
}

__global__ __launch_bounds__(64) void com_weights_kernel(ChlMap hm, int C, int H, int W, const long long *__restrict__ radius_map,
                                                         int rmap_cols, int num_max, ComCur cur,
                                                         const double *__restrict__ state, float *__restrict__ box_mask,
                                                         float *__restrict__ weights, int *__restrict__ owner) {
    const size_t at = blockIdx.x;
    const long long *rm = radius_map + at * rmap_cols;
    const int b = (int)(at / num_max), k = (int)(at % num_max);
    int c, cx, cy, x0, y0, w, h;
    if (rm[3] <= 0) return;
    const bool rect = com_rect(cur, rm, W, H, c, cx, cy, x0, y0, w, h);
    const double p = (double)com_pred(hm, b, c, cy, cx);            // pred_confidence.item()
    const double thr = cur.fix ? cur.threshold : state[0] * cur.threshold;
    double wt;
    if (cur.straight) wt = cur.k_straight * (p - thr) + 1.0;
    else if (cur.tuning) wt = 1.0;
    else wt = cur.height / (1.0 + exp(cur.elongation * (p - thr))) + 1.0 - cur.height / 2.0;
    if (!cur.apply) return;                                         // outside [START, END]: nothing is written
    if (threadIdx.x == 0) {
        box_mask[at] = (float)wt;
        weights[at] = (float)wt;
    }
    if (!rect) return;
    int *plane = owner + ((size_t)b * C + c) * H * W;
    for (int q = threadIdx.x; q < w * h; q += 64) {
        const int py = q / w, px = q - py * w;
        atomicMax(plane + (size_t)(y0 + py) * W + (x0 + px), k + 1);
    }
}

// second walk: the owner of a pixel stores its weight into heatmap_mask (in place, as the reference does) and
// returns the owner plane to zero (it is zeroed once by its owner and cleans itself)
__global__ __launch_bounds__(64) void com_mask_write_kernel(int C, int H, int W, const long long *__restrict__ radius_map,
                                                            int rmap_cols, int num_max, ComCur cur,
                                                            const float *__restrict__ weights, int *__restrict__ owner,
                                                            float *__restrict__ heatmap_mask) {
    const size_t at = blockIdx.x;
    const long long *rm = radius_map + at * rmap_cols;
    const int b = (int)(at / num_max), k = (int)(at % num_max);
    int c, cx, cy, x0, y0, w, h;
    if (!cur.apply || !com_rect(cur, rm, W, H, c, cx, cy, x0, y0, w, h)) return;
    const float wt = weights[at];
    const size_t base = ((size_t)b * C + c) * H * W;
    for (int q = threadIdx.x; q < w * h; q += 64) {
        const int py = q / w, px = q - py * w;
        const size_t e = base + (size_t)(y0 + py) * W + (x0 + px);
        if (owner[e] == k + 1) {
            heatmap_mask[e] = wt;
            owner[e] = 0;
        }
    }
}

// partial[blk][0..5] = pos_loss, neg_loss, num_pos (all three already multiplied by the frame-summed mask), sum of pred at
// the positives, number of positives, --; [8 .. 8 + dims) L1 sums; [8 + dims] = sum of the box mask.
// Work item = one (c, y, x) column over the frames.  msum (may be NULL when mask == NULL) receives the frame-summed
// mask for the backward pass.
__global__ __launch_bounds__(256) void com_forward_kernel(ChlMap hm, const float *__restrict__ gt,
                                                          const float *__restrict__ hmask, float *__restrict__ msum,
                                                          int B, int C, int H, int W, ChlRegs regs,
                                                          const long long *__restrict__ ind,
                                                          const float *__restrict__ box_mask,
                                                          const float *__restrict__ target, int M,
                                                          double *__restrict__ partial) {
    __shared__ double lds[4];
    const unsigned chw = (unsigned)C * H * W;
    double pos = 0.0, neg = 0.0, npos = 0.0, conf = 0.0, nobj = 0.0;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < chw; e += COM_BLOCKS * 256u) {
        const int x = (int)(e % (unsigned)W);
        unsigned t = e / (unsigned)W;
        const int y = (int)(t % (unsigned)H);
        const int c = (int)(t / (unsigned)H);
        double mk = (double)B;
        if (hmask) {
            float m = 0.0f;
            for (int b = 0; b < B; ++b) m += hmask[(size_t)b * chw + e];     // (<= a few frames: exact enough in f32)
            mk = (double)m;
            if (msum) msum[e] = m;
        }
        double p_col = 0.0, n_col = 0.0, i_col = 0.0;
        for (int b = 0; b < B; ++b) {
            const float g = gt[(size_t)b * chw + e];
            const float p = com_pred(hm, b, c, y, x);
            if (g == 1.0f) {
                const float q = 1.0f - p;
                p_col += (double)(logf(p) * (q * q));
                i_col += 1.0;
                conf += (double)p;
            } else if (g < 1.0f) {
                const float w1 = 1.0f - g, w2 = w1 * w1;
                n_col += (double)(logf(1.0f - p) * (p * p) * (w2 * w2));
            }
        }
        pos += p_col * mk;
        neg += n_col * mk;
        npos += i_col * mk;
        nobj += i_col;
    }
    double *row = partial + (size_t)blockIdx.x * COM_PSTRIDE;
    const double s0 = chl_block_sum(pos, lds), s1 = chl_block_sum(neg, lds), s2 = chl_block_sum(npos, lds),
                 s3 = chl_block_sum(conf, lds), s4 = chl_block_sum(nobj, lds);
    if (threadIdx.x == 0) {
        row[0] = s0; row[1] = s1; row[2] = s2; row[3] = s3; row[4] = s4;
    }
    // regression with the FLOAT box mask (loss_utils.py:1317-1345): |pred * m - target * m| per code dimension
    double acc[CHL_MAX_DIM + 1];
#pragma unroll
    for (int d = 0; d <= CHL_MAX_DIM; ++d) acc[d] = 0.0;
    if ((int)blockIdx.x < B) {
        const int b = blockIdx.x;
        for (int m0 = threadIdx.x; m0 < M; m0 += 256) {
            const int o = b * M + m0;
            const float mk = box_mask[o];
            const long long pix = ind[o];
            const int y = (int)(pix / W), x = (int)(pix % W);
            acc[CHL_MAX_DIM] += (double)mk;
            int d0 = 0;
            for (int r = 0; r < regs.n; ++r) {
                const ChlMap &m = regs.m[r];
                for (int c = 0; c < m.c; ++c) {
                    const float pr = chl_load(m, b * m.sb + c * m.sc + y * m.sh + x * m.sw);
                    const float v = fabsf(pr * mk - target[(size_t)o * regs.dims + d0 + c] * mk);
#pragma unroll
                    for (int d = 0; d < CHL_MAX_DIM; ++d)
                        if (d == d0 + c) acc[d] += (double)v;
                }
                d0 += m.c;
            }
        }
    }
#pragma unroll
    for (int d = 0; d < CHL_MAX_DIM; ++d) {
        const double t = chl_block_sum(acc[d], lds);
        if (threadIdx.x == 0 && d < regs.dims) row[8 + d] = t;
    }
    const double nbox = chl_block_sum(acc[CHL_MAX_DIM], lds);
    if (threadIdx.x == 0) row[8 + regs.dims] = nbox;
}

// out[0] = loss, [1] = hm_loss, [2] = loc_loss, [3] = avg_confidence, [4] = num_pos (mask-weighted; 0 selects the
// reference's other branch), [5] = sum of the box mask, [6 .. 6 + dims) = L1 per dim.
// Also: EMA update (unless com_ema_kernel already did it), the per-(class, group) confidence sums / counts of THIS step
// (conf_all / num_all, loss_utils.py:1166-1176) and their running epoch sums (train_utils.py:111-112,208).
constexpr int COM_CHUNK = 1024;
__global__ __launch_bounds__(256) void com_finalize_kernel(const double *__restrict__ partial, int dims,
                                                           const float *__restrict__ code_weights, float cls_weight,
                                                           float loc_weight, ChlMap hm, const long long *__restrict__ radius_map,
                                                           int rmap_cols, int n_slots, int num_max, ComCur cur,
                                                           int ema_done, double *__restrict__ state,
                                                           float *__restrict__ out, float *__restrict__ conf_all,
                                                           float *__restrict__ num_all, float *__restrict__ conf_epoch,
                                                           float *__restrict__ num_epoch) {
    __shared__ double tot[8 + CHL_MAX_DIM + 1];
    __shared__ int cell_s[COM_CHUNK];
    __shared__ float val_s[COM_CHUNK];
    __shared__ int lds[4];
    if ((int)threadIdx.x < 8 + dims + 1) {
        double s = 0.0;
        if (threadIdx.x < 5 || threadIdx.x >= 8)
            for (int b = 0; b < COM_BLOCKS; ++b) s += partial[(size_t)b * COM_PSTRIDE + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float pos = (float)tot[0], neg = (float)tot[1], npos = (float)tot[2];
        const float hm_loss = (npos == 0.0f ? -neg : -(pos + neg) / npos) * cls_weight;
        const float nbox = (float)tot[8 + dims];
        float loc = 0.0f;
        for (int d = 0; d < dims; ++d) {
            const float l = (float)tot[8 + d] / fmaxf(nbox, 1.0f);
            out[6 + d] = l;
            loc += l * code_weights[d];
        }
        loc *= loc_weight;
        const float avg = (float)tot[3] / (float)tot[4];              // nan without positives, as in the reference
        out[0] = hm_loss + loc;
        out[1] = hm_loss;
        out[2] = loc;
        out[3] = avg;
        out[4] = npos;
        out[5] = nbox;
        if (!ema_done) {
            state[1] = (double)avg;
            state[0] = cur.alpha * (double)avg + (1.0 - cur.alpha) * state[0];
        }
    }
    if (!conf_all || cur.conf_c <= 0 || cur.conf_g <= 0) return;
    // group confidences: the slots with a group id are compacted chunk by chunk into LDS (slot order), then one thread
    // per (class, group) cell adds its values in that order: deterministic, no atomics
    const int cells = cur.conf_c * cur.conf_g;
    double acc[2] = {0.0, 0.0}, cnt[2] = {0.0, 0.0};                 // cells handled by this thread: t, t + 256 (<= 512)
    for (int base = 0; base < n_slots; base += COM_CHUNK) {
        int filled = 0;
        for (int sub = 0; sub < COM_CHUNK; sub += 256) {
            const int at = base + sub + threadIdx.x;
            int cell = -1;
            float v = 0.0f;
            if (at < n_slots && sub + (int)threadIdx.x < COM_CHUNK) {
                const long long *rm = radius_map + (size_t)at * rmap_cols;
                const long long g = rmap_cols > 4 ? rm[4] : 0, c = rm[0];
                if (g >= 1 && g <= cur.conf_g && c >= 0 && c < cur.conf_c) {
                    cell = (int)c * cur.conf_g + (int)g - 1;
                    v = com_pred(hm, at / num_max, (int)c, (int)rm[2], (int)rm[1]);
                }
            }
            int total;
            const int pos = filled + block_exclusive_scan(cell >= 0 ? 1 : 0, lds, total);
            if (cell >= 0) {
                cell_s[pos] = cell;
                val_s[pos] = v;
            }
            filled += total;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int mine = threadIdx.x + 256 * j;
            if (mine < cells)
                for (int i = 0; i < filled; ++i)
                    if (cell_s[i] == mine) {
                        acc[j] += (double)val_s[i];
                        cnt[j] += 1.0;
                    }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int mine = threadIdx.x + 256 * j;
        if (mine < cells) {
            conf_all[mine] = (float)acc[j];
            num_all[mine] = (float)cnt[j];
            if (conf_epoch) conf_epoch[mine] += (float)acc[j];        // float32 `sum` of the per-step tensors, in step order
            if (num_epoch) num_epoch[mine] += (float)cnt[j];
        }
    }
}

// d loss / d hm logits; regression gradients zeroed here and filled by com_scatter_kernel
__global__ __launch_bounds__(256) void com_backward_kernel(ChlMap hm, const float *__restrict__ gt,
                                                           const float *__restrict__ msum, int B, int C, int H, int W,
                                                           ChlRegs regs, const float *__restrict__ out,
                                                           const float *__restrict__ grad_out, float cls_weight) {
    const unsigned chw = (unsigned)C * H * W;
    const unsigned total = (unsigned)B * chw;
    const float npos = out[4];
    const bool has_pos = npos != 0.0f;
    const float scale = -cls_weight / (has_pos ? npos : 1.0f) * grad_out[0];
    const long long hw = (long long)H * W;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
        const unsigned col = e % chw;
        const int x = (int)(col % (unsigned)W);
        unsigned t = col / (unsigned)W;
        const int y = (int)(t % (unsigned)H);
        const int c = (int)(t / (unsigned)H), b = (int)(e / chw);
        const float g = gt[e];
        const long long off = b * hm.sb + c * hm.sc + y * hm.sh + x * hm.sw;
        const float s = chl_sigmoid(chl_load(hm, off));
        const bool inside = s >= 1e-4f && s <= 1.0f - 1e-4f;
        const float p = fminf(fmaxf(s, 1e-4f), 1.0f - 1e-4f);
        const float mk = msum ? msum[col] : (float)B;
        float dp = 0.0f;
        if (g == 1.0f) {
            const float q = 1.0f - p;
            dp = has_pos ? q * q / p - 2.0f * q * logf(p) : 0.0f;   // (num_pos == 0 branch: loss = -neg_loss only)
        } else if (g < 1.0f) {
            const float w1 = 1.0f - g, w2 = w1 * w1;
            dp = (-(p * p) / (1.0f - p) + 2.0f * p * logf(1.0f - p)) * (w2 * w2);
        }
        chl_store_grad(hm, off, inside ? scale * dp * mk * (s * (1.0f - s)) : 0.0f);
    }
    for (int r = 0; r < regs.n; ++r) {
        const ChlMap &m = regs.m[r];
        const long long n = (long long)B * m.c * hw;
        for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < (unsigned)n; e += gridDim.x * 256u) {
            const int x = (int)(e % (unsigned)W);
            unsigned t = e / (unsigned)W;
            const int y = (int)(t % (unsigned)H);
            t /= (unsigned)H;
            const int c = (int)(t % (unsigned)m.c), b = (int)(t / (unsigned)m.c);
            chl_store_grad(m, b * m.sb + c * m.sc + y * m.sh + x * m.sw, 0.0f);
        }
    }
}

// like chl_scatter_kernel with float weights: d |pr m - t m| / d pr = sign(pr m - t m) m
__global__ __launch_bounds__(256) void com_scatter_kernel(ChlRegs regs, int W, const long long *__restrict__ ind,
                                                          const float *__restrict__ box_mask,
                                                          const float *__restrict__ target, int M,
                                                          const float *__restrict__ code_weights,
                                                          const float *__restrict__ out,
                                                          const float *__restrict__ grad_out, float loc_weight) {
    __shared__ int pix_s[CHL_MAX_OBJS];
    __shared__ int obj_s[CHL_MAX_OBJS];
    __shared__ int lds[4];
    const int b = blockIdx.x;
    int K = 0;
    for (int base = 0; base < M; base += 256) {
        const int m0 = base + threadIdx.x;
        const bool on = m0 < M && box_mask[b * M + m0] != 0.0f;
        int total;
        const int pos = K + block_exclusive_scan(on ? 1 : 0, lds, total);
        if (on && pos < CHL_MAX_OBJS) {
            pix_s[pos] = (int)ind[b * M + m0];
            obj_s[pos] = b * M + m0;
        }
        K += total;
    }
    __syncthreads();
    K = K < CHL_MAX_OBJS ? K : CHL_MAX_OBJS;
    const float scale = loc_weight / fmaxf(out[5], 1.0f) * grad_out[0];
    for (int k = threadIdx.x; k < K; k += 256) {
        const int pix = pix_s[k];
        bool first = true;
        for (int j = 0; j < k; ++j) first = first && pix_s[j] != pix;
        if (!first) continue;
        const int y = pix / W, x = pix - y * W;
        int d0 = 0;
        for (int r = 0; r < regs.n; ++r) {
            const ChlMap &mp = regs.m[r];
            for (int c = 0; c < mp.c; ++c) {
                const long long off = b * mp.sb + c * mp.sc + y * mp.sh + x * mp.sw;
                const float pr = chl_load(mp, off);
                float gsum = 0.0f;
                for (int j = k; j < K; ++j) {
                    if (pix_s[j] != pix) continue;
                    const float mk = box_mask[obj_s[j]];
                    const float diff = pr * mk - target[(size_t)obj_s[j] * regs.dims + d0 + c] * mk;
                    gsum += diff > 0.0f ? mk : (diff < 0.0f ? -mk : 0.0f);
                }
                chl_store_grad(mp, off, scale * code_weights[d0 + c] * gsum);
            }
            d0 += mp.c;
        }
    }
}

static ComCur com_cur(const PcdComCurriculum *c, int has_rmap5) {
    ComCur k = {};
    k.ucl = c->ucl; k.fix = c->fix_threshold; k.straight = c->straight; k.tuning = c->tuning;
    k.only_center = c->only_center; k.apply = c->apply; k.add = c->add; k.radius = c->radius;
    k.k_straight = c->k_straight; k.elongation = c->elongation; k.height = c->height; k.alpha = c->alpha;
    k.threshold = c->threshold;
    k.conf_c = has_rmap5 ? c->conf_classes : 0;
    k.conf_g = has_rmap5 ? c->conf_groups : 0;
    return k;
}

}  // namespace

extern "C" int pcd_com_cluster_groups(const float *gt_boxes, int batch, int n_boxes, int code_size,
                                      const float *true_object, const float *occupancy_ratio, const float *facade_type,
                                      int variant, long long *group, void *stream) {
    PCD_ENTER();
    if (batch <= 0 || n_boxes < 0 || code_size < 8 || !group) return PCD_ERR_INVALID_ARG;
    if (variant != PCD_COM_CLUSTER_X5) return PCD_ERR_UNSUPPORTED;
    const long long total = (long long)batch * n_boxes;
    if (total == 0) return PCD_OK;
    if (!gt_boxes || !true_object || !occupancy_ratio || !facade_type || total > 2147483647LL) return PCD_ERR_INVALID_ARG;
    com_cluster_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        gt_boxes, (int)total, code_size, true_object, occupancy_ratio, facade_type, group);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" size_t pcd_com_assign_workspace_bytes(int batch, int num_max_objs) {
    return pcd_centerhead_assign_workspace_bytes(batch, num_max_objs);
}

extern "C" int pcd_com_assign_targets(const float *gt_boxes, int batch, int n_boxes, int code_size,
                                      const int *class_map_host, int n_class_map, int head_classes, int fm_w, int fm_h,
                                      int feature_map_stride, const float *voxel_size_xy_host, const float *range_xy_host,
                                      int num_max_objs, float gaussian_overlap, int min_radius,
                                      const float *num_points_in_gt, const long long *group, int gate_min_points,
                                      float min_points, float *heatmap, float *ret_boxes, long long *inds, float *mask,
                                      long long *radius_map, int radius_map_cols, float *heatmap_mask, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    PCD_ENTER();
    if (batch <= 0 || n_boxes < 0 || code_size < 8 || !class_map_host || n_class_map <= 0 || n_class_map > 16 ||
        head_classes <= 0 || fm_w <= 0 || fm_h <= 0 || feature_map_stride <= 0 || num_max_objs <= 0 || !voxel_size_xy_host ||
        !range_xy_host || (radius_map_cols != 4 && radius_map_cols != 5))
        return PCD_ERR_INVALID_ARG;
    if (!heatmap || !ret_boxes || !inds || !mask || !radius_map || !heatmap_mask) return PCD_ERR_INVALID_ARG;
    if (n_boxes > 0 && (!gt_boxes || (gate_min_points && !num_points_in_gt))) return PCD_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < pcd_com_assign_workspace_bytes(batch, num_max_objs)) return PCD_ERR_WORKSPACE;
    AssignGeom G = {};
    G.range_x = range_xy_host[0]; G.range_y = range_xy_host[1];
    G.vs_x = voxel_size_xy_host[0]; G.vs_y = voxel_size_xy_host[1];
    G.stride = feature_map_stride; G.W = fm_w; G.H = fm_h; G.num_max = num_max_objs; G.min_radius = min_radius;
    G.code = code_size; G.head_classes = head_classes; G.overlap = gaussian_overlap;
    for (int i = 0; i < n_class_map; ++i) G.cls_map[i] = class_map_host[i];
    hipStream_t st = (hipStream_t)stream;
    int4 *draw = (int4 *)workspace;
    ComInit I = {heatmap, heatmap_mask, ret_boxes, mask, inds, radius_map, draw,
                 (size_t)batch * head_classes * fm_h * fm_w, (size_t)batch * num_max_objs * code_size,
                 (size_t)batch * num_max_objs, (size_t)batch * num_max_objs * radius_map_cols};
    size_t blocks = (I.n_heat + 1023) / 1024;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    com_targets_init_kernel<<<(unsigned)blocks, 256, 0, st>>>(I);
    com_assign_rows_kernel<<<batch, 64, 0, st>>>(gt_boxes, n_boxes, G, num_points_in_gt, group, gate_min_points,
                                                 min_points, radius_map_cols, ret_boxes, inds, mask, radius_map, draw);
    draw_gaussian_kernel<<<(unsigned)((size_t)batch * num_max_objs), 64, 0, st>>>(draw, G, heatmap);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" size_t pcd_com_loss_workspace_bytes(int batch, int num_max_objs) {
    if (batch <= 0 || num_max_objs < 0) return 0;
    return ws_piece((size_t)COM_BLOCKS * COM_PSTRIDE, sizeof(double)) + ws_piece((size_t)batch * num_max_objs, sizeof(float));
}

extern "C" int pcd_com_loss_forward(const void *hm, int hm_dtype, const long long *hm_strides_host,
                                    const float *gt_heatmap, int batch, int num_classes, int height, int width,
                                    const void *const *reg_ptrs_host, const int *reg_channels_host, int reg_dtype,
                                    const long long *reg_strides_host, int n_reg, const long long *inds,
                                    float *box_mask, const float *target_boxes, const long long *radius_map,
                                    int radius_map_cols, int num_max_objs, float *heatmap_mask, int32_t *owner,
                                    float *mask_sum, const PcdComCurriculum *cur_host, const float *code_weights,
                                    float cls_weight, float loc_weight, double *state, float *out, float *conf_all,
                                    float *num_all, float *conf_epoch, float *num_epoch, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    PCD_ENTER();
    if (batch <= 0 || num_classes <= 0 || height <= 0 || width <= 0 || num_max_objs <= 0 || !gt_heatmap || !out ||
        !workspace || !cur_host || !state || !inds || !box_mask || !target_boxes || !radius_map || !code_weights ||
        (radius_map_cols != 4 && radius_map_cols != 5))
        return PCD_ERR_INVALID_ARG;
    ChlMap H_;
    ChlRegs R;
    int rc = chl_pack(hm, nullptr, hm_dtype, hm_strides_host, num_classes, reg_ptrs_host, nullptr, reg_channels_host,
                      reg_dtype, reg_strides_host, n_reg, &H_, &R);
    if (rc != PCD_OK) return rc;
    if (workspace_bytes < pcd_com_loss_workspace_bytes(batch, num_max_objs)) return PCD_ERR_WORKSPACE;
    if ((double)batch * (num_classes > CHL_MAX_DIM ? num_classes : CHL_MAX_DIM) * height * width >= 2147483647.0)
        return PCD_ERR_UNSUPPORTED;
    if (num_max_objs > CHL_MAX_OBJS) return PCD_ERR_UNSUPPORTED;
    if (batch > COM_BLOCKS) return PCD_ERR_UNSUPPORTED;   // (the regression sums take one workgroup per frame of a fixed grid)
    ComCur cur = com_cur(cur_host, radius_map_cols == 5);
    if (cur.conf_c * cur.conf_g > 512) return PCD_ERR_UNSUPPORTED;
    if (cur.conf_c > 0 && (!conf_all || !num_all)) return PCD_ERR_INVALID_ARG;
    if (cur.ucl && (!heatmap_mask || !owner || !mask_sum)) return PCD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    double *partial = (double *)workspace;
    float *weights = (float *)((char *)workspace + ws_piece((size_t)COM_BLOCKS * COM_PSTRIDE, sizeof(double)));
    const int n_slots = batch * num_max_objs;
    int ema_done = 0;
    if (cur.ucl) {
        if (!cur.fix) {                                   // the threshold follows the EMA, which this step's average updates first
            com_conf_partials_kernel<<<COM_BLOCKS, 256, 0, st>>>(H_, gt_heatmap, batch, num_classes, height, width, partial);
            com_ema_kernel<<<1, 256, 0, st>>>(partial, cur.alpha, state);
            ema_done = 1;
        }
        com_weights_kernel<<<n_slots, 64, 0, st>>>(H_, num_classes, height, width, radius_map, radius_map_cols,
                                                   num_max_objs, cur, state, box_mask, weights, owner);
        if (cur.apply)
            com_mask_write_kernel<<<n_slots, 64, 0, st>>>(num_classes, height, width, radius_map, radius_map_cols,
                                                          num_max_objs, cur, weights, owner, heatmap_mask);
    }
    com_forward_kernel<<<COM_BLOCKS, 256, 0, st>>>(H_, gt_heatmap, cur.ucl ? heatmap_mask : nullptr,
                                                   cur.ucl ? mask_sum : nullptr, batch, num_classes, height, width, R,
                                                   inds, box_mask, target_boxes, num_max_objs, partial);
    com_finalize_kernel<<<1, 256, 0, st>>>(partial, R.dims, code_weights, cls_weight, loc_weight, H_, radius_map,
                                           radius_map_cols, n_slots, num_max_objs, cur, ema_done, state, out, conf_all,
                                           num_all, conf_epoch, num_epoch);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_com_loss_backward(const void *hm, void *d_hm, int hm_dtype, const long long *hm_strides_host,
                                     const float *gt_heatmap, int batch, int num_classes, int height, int width,
                                     const void *const *reg_ptrs_host, void *const *reg_grads_host,
                                     const int *reg_channels_host, int reg_dtype, const long long *reg_strides_host,
                                     int n_reg, const long long *inds, const float *box_mask, const float *target_boxes,
                                     int num_max_objs, const float *mask_sum, const float *code_weights, float cls_weight,
                                     float loc_weight, const float *out, const float *grad_out, void *stream) {
    PCD_ENTER();
    if (batch <= 0 || num_classes <= 0 || height <= 0 || width <= 0 || num_max_objs <= 0 || !gt_heatmap || !out ||
        !grad_out || !d_hm || (n_reg > 0 && !reg_grads_host) || !inds || !box_mask || !target_boxes || !code_weights)
        return PCD_ERR_INVALID_ARG;
    ChlMap H_;
    ChlRegs R;
    int rc = chl_pack(hm, d_hm, hm_dtype, hm_strides_host, num_classes, reg_ptrs_host, reg_grads_host,
                      reg_channels_host, reg_dtype, reg_strides_host, n_reg, &H_, &R);
    if (rc != PCD_OK) return rc;
    for (int r = 0; r < n_reg; ++r)
        if (!reg_grads_host[r]) return PCD_ERR_INVALID_ARG;
    if ((double)batch * (num_classes > CHL_MAX_DIM ? num_classes : CHL_MAX_DIM) * height * width >= 2147483647.0)
        return PCD_ERR_UNSUPPORTED;
    if (num_max_objs > CHL_MAX_OBJS) return PCD_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    com_backward_kernel<<<1024, 256, 0, st>>>(H_, gt_heatmap, mask_sum, batch, num_classes, height, width, R, out,
                                              grad_out, cls_weight);
    if (n_reg > 0)
        com_scatter_kernel<<<batch, 256, 0, st>>>(R, width, inds, box_mask, target_boxes, num_max_objs, code_weights, out,
                                                  grad_out, loc_weight);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
