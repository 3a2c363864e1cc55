// CenterHead target assignment on the GPU (SURVEY.md 8f #2 "host-side step overheads"): the reference builds the
// training targets with a Python loop over ground-truth boxes, on CPU tensors, with one `.item()` and one numpy ->
// torch -> device copy per box (pcdet/models/dense_heads/center_head.py:104-161, :163-225 and
// pcdet/models/model_utils/centernet_utils.py:46-107) -- a per-step host stall that grows with the number of
// objects.  Here: one wave per batch element compacts the boxes of the head's classes (ballot prefix keeps their
// order), computes centre / Gaussian radius / regression targets with the reference's float32 formulas, and a second
// kernel draws every Gaussian with an order-free atomic max (heat-map values are >= 0, so their float bits order like
// integers) -- deterministic, no host round trip.
#include "centerhead_common.h"

namespace {

__global__ __launch_bounds__(64) void assign_rows_kernel(const float *__restrict__ gt, int n, AssignGeom G,
                                                         float *__restrict__ ret_boxes, long long *__restrict__ inds,
                                                         long long *__restrict__ mask, int4 *__restrict__ draw) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float *rows = gt + (size_t)b * n * G.code;
    const int out_code = G.code;                              // ret_boxes rows: code - 1 + 1 values
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
        int radius = (int)gaussian_radius_f32(dx, dy, G.overlap);
        radius = radius < G.min_radius ? G.min_radius : radius;
        const size_t at = (size_t)b * G.num_max + k;
        int4 d = make_int4(-1, 0, 0, 0);
        if (dx > 0.0f && dy > 0.0f && ix >= 0 && ix <= G.W && iy >= 0 && iy <= G.H) {
            inds[at] = (long long)iy * G.W + ix;
            mask[at] = 1;
            float *o = ret_boxes + at * out_code;
            o[0] = cx - (float)ix;
            o[1] = cy - (float)iy;
            o[2] = q[2];
            o[3] = logf(q[3]);
            o[4] = logf(q[4]);
            o[5] = logf(q[5]);
            o[6] = cosf(q[6]);
            o[7] = sinf(q[6]);
            for (int j = 8; j < out_code; ++j) o[j] = q[j - 1];
            d = make_int4(local - 1, ix, iy, radius);
        }
        draw[at] = d;
    }
}

}  // namespace

extern "C" size_t pcd_centerhead_assign_workspace_bytes(int batch, int num_max_objs) {
    if (batch <= 0 || num_max_objs <= 0) return 0;
    return ws_piece((size_t)batch * num_max_objs, sizeof(int4));
}

extern "C" int pcd_centerhead_assign_targets(const float *gt_boxes, int batch, int n_boxes, int code_size,
                                             const int *class_map_host, int n_class_map, int head_classes, int fm_w, int fm_h,
                                             int feature_map_stride, const float *voxel_size_xy_host,
                                             const float *range_xy_host, int num_max_objs, float gaussian_overlap,
                                             int min_radius, float *heatmap, float *ret_boxes, long long *inds,
                                             long long *mask, void *workspace, size_t workspace_bytes, void *stream) {
    PCD_ENTER();
    if (batch <= 0 || n_boxes < 0 || code_size < 8 || !class_map_host || n_class_map <= 0 || n_class_map > 16 ||
        head_classes <= 0 || fm_w <= 0 || fm_h <= 0 || feature_map_stride <= 0 || num_max_objs <= 0 || !voxel_size_xy_host ||
        !range_xy_host)
        return PCD_ERR_INVALID_ARG;
    if (!heatmap || !ret_boxes || !inds || !mask) return PCD_ERR_INVALID_ARG;
    if (n_boxes > 0 && !gt_boxes) return PCD_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < pcd_centerhead_assign_workspace_bytes(batch, num_max_objs)) return PCD_ERR_WORKSPACE;
    AssignGeom G = {};
    G.range_x = range_xy_host[0]; G.range_y = range_xy_host[1];
    G.vs_x = voxel_size_xy_host[0]; G.vs_y = voxel_size_xy_host[1];
    G.stride = feature_map_stride; G.W = fm_w; G.H = fm_h; G.num_max = num_max_objs; G.min_radius = min_radius;
    G.code = code_size; G.head_classes = head_classes; G.overlap = gaussian_overlap;
    for (int i = 0; i < n_class_map; ++i) G.cls_map[i] = class_map_host[i];
    hipStream_t st = (hipStream_t)stream;
    int4 *draw = (int4 *)workspace;
    pcd_fill(heatmap, 0, (size_t)batch * head_classes * fm_h * fm_w * sizeof(float), st);
    pcd_fill(ret_boxes, 0, (size_t)batch * num_max_objs * code_size * sizeof(float), st);
    pcd_fill(inds, 0, (size_t)batch * num_max_objs * sizeof(long long), st);
    pcd_fill(mask, 0, (size_t)batch * num_max_objs * sizeof(long long), st);
    pcd_fill(draw, 0xFF, (size_t)batch * num_max_objs * sizeof(int4), st);
    assign_rows_kernel<<<batch, 64, 0, st>>>(gt_boxes, n_boxes, G, ret_boxes, inds, mask, draw);
    draw_gaussian_kernel<<<(unsigned)((size_t)batch * num_max_objs), 64, 0, st>>>(draw, G, heatmap);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

// =============================================================================================================
// CenterHead.get_loss of one head in four launches (center_head.py:226-262; loss_utils.py:611-643 neg_loss_cornernet,
// :1317-1390 RegLossCenterNet / _transpose_and_gather_feat) instead of ~100 elementwise / reduction launches:
//   forward : partial sums per workgroup (focal terms, number of positives, confidence; L1 sums per code dimension,
//             number of objects) -> one finalising workgroup -> out[] (loss, hm_loss, loc_loss, confidence, ...)
//   backward: d loss / d hm logits (dense) and zeros for the regression maps, then the <= B x M object positions
//             scattered into the regression gradients (objects sharing a pixel are summed by the first of them, in
//             object order: no atomics, deterministic).
// Same arithmetic as the reference in fp32: p = clamp(sigmoid(x), 1e-4, 1 - 1e-4), the clamp's gradient is 1 inside
// [1e-4, 1 - 1e-4] (inclusive) and 0 outside, |.|'s gradient is sign(.) with sign(0) = 0.  Sums are taken in a fixed
// order of this kernel's own (per-thread strided -> wave -> workgroup -> 256 partials in order).
// Predictions are addressed through element strides (NCHW or channels-last, bf16 or f32).
namespace {

// partial[blk][0..3] = pos_loss, neg_loss, num_pos, sum of p at the positives; [4 .. 4 + dims) = L1 sums; [4 + dims] =
// number of objects (the regression part by block 0 only)
__global__ __launch_bounds__(256) void chl_forward_kernel(ChlMap hm, const float *__restrict__ gt, int B, int C, int H,
                                                          int W, ChlRegs regs, const long long *__restrict__ ind,
                                                          const long long *__restrict__ mask,
                                                          const float *__restrict__ target, int M,
                                                          double *__restrict__ partial, int pstride) {
    __shared__ double lds[4];
    const long long total = (long long)B * C * H * W;
    double pos = 0.0, neg = 0.0, npos = 0.0, conf = 0.0;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < (unsigned)total; e += CHL_BLOCKS * 256u) {
        const int x = (int)(e % (unsigned)W);
        unsigned t = e / (unsigned)W;
        const int y = (int)(t % (unsigned)H);
        t /= (unsigned)H;
        const int c = (int)(t % (unsigned)C), b = (int)(t / (unsigned)C);
        const float g = gt[e];
        float p = chl_sigmoid(chl_load(hm, b * hm.sb + c * hm.sc + y * hm.sh + x * hm.sw));
        p = fminf(fmaxf(p, 1e-4f), 1.0f - 1e-4f);
        if (g == 1.0f) {
            const float q = 1.0f - p;
            pos += (double)(logf(p) * (q * q));
            npos += 1.0;
            conf += (double)p;
        } else if (g < 1.0f) {
            const float w1 = 1.0f - g, w2 = w1 * w1;
            neg += (double)(logf(1.0f - p) * (p * p) * (w2 * w2));
        }
    }
    double *row = partial + (size_t)blockIdx.x * pstride;
    const double s0 = chl_block_sum(pos, lds), s1 = chl_block_sum(neg, lds), s2 = chl_block_sum(npos, lds),
                 s3 = chl_block_sum(conf, lds);
    if (threadIdx.x == 0) {
        row[0] = s0; row[1] = s1; row[2] = s2; row[3] = s3;
    }
    // regression: workgroup b < B takes the objects of frame b (a thread per object, all code dimensions: the gathers of
    // one object are independent loads), every workgroup writes its row (zeros beyond the frames)
    double acc[CHL_MAX_DIM + 1];
#pragma unroll
    for (int d = 0; d <= CHL_MAX_DIM; ++d) acc[d] = 0.0;
    if ((int)blockIdx.x < B) {
        const int b = blockIdx.x;
        for (int m0 = threadIdx.x; m0 < M; m0 += 256) {
            const int o = b * M + m0;
            const float mk = mask[o] != 0 ? 1.0f : 0.0f;
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
                    for (int d = 0; d < CHL_MAX_DIM; ++d)      // (static register index)
                        if (d == d0 + c) acc[d] += (double)v;
                }
                d0 += m.c;
            }
        }
    }
#pragma unroll
    for (int d = 0; d < CHL_MAX_DIM; ++d) {
        const double t = chl_block_sum(acc[d], lds);
        if (threadIdx.x == 0 && d < regs.dims) row[4 + d] = t;
    }
    const double nobj = chl_block_sum(acc[CHL_MAX_DIM], lds);
    if (threadIdx.x == 0) row[4 + regs.dims] = nobj;
}

// out[0] = loss, [1] = hm_loss, [2] = loc_loss, [3] = confidence, [4] = num_pos, [5] = num_obj, [6 .. 6 + dims) = L1 per dim
__global__ __launch_bounds__(256) void chl_finalize_kernel(const double *__restrict__ partial, int pstride, int dims,
                                                           const float *__restrict__ code_weights, float cls_weight,
                                                           float loc_weight, float *__restrict__ out) {
    __shared__ double tot[4 + CHL_MAX_DIM + 1];
    if ((int)threadIdx.x < 4 + dims + 1) {
        double s = 0.0;
        for (int b = 0; b < CHL_BLOCKS; ++b) s += partial[(size_t)b * pstride + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float pos = (float)tot[0], neg = (float)tot[1], npos = (float)tot[2], conf = (float)tot[3];
    const float hm_loss = -(pos + neg) / fmaxf(npos, 1.0f) * cls_weight;
    const float nobj = (float)tot[4 + dims];
    float loc = 0.0f;
    for (int d = 0; d < dims; ++d) {
        const float l = (float)tot[4 + d] / fmaxf(nobj, 1.0f);
        out[6 + d] = l;
        loc += l * code_weights[d];
    }
    loc *= loc_weight;
    out[0] = hm_loss + loc;
    out[1] = hm_loss;
    out[2] = loc;
    out[3] = conf / npos;          // (nan without positives, as in the reference)
    out[4] = npos;
    out[5] = nobj;
}

// d loss / d hm logits for every element; the regression gradients are zeroed here and filled by chl_scatter_kernel
__global__ __launch_bounds__(256) void chl_backward_kernel(ChlMap hm, const float *__restrict__ gt, int B, int C, int H,
                                                           int W, ChlRegs regs, const float *__restrict__ out,
                                                           const float *__restrict__ grad_out, float cls_weight) {
    const long long total = (long long)B * C * H * W;
    const float scale = -cls_weight / fmaxf(out[4], 1.0f) * grad_out[0];
    const long long hw = (long long)H * W;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < (unsigned)total; e += gridDim.x * 256u) {
        const int x = (int)(e % (unsigned)W);
        unsigned t = e / (unsigned)W;
        const int y = (int)(t % (unsigned)H);
        t /= (unsigned)H;
        const int c = (int)(t % (unsigned)C), b = (int)(t / (unsigned)C);
        const float g = gt[e];
        const long long off = b * hm.sb + c * hm.sc + y * hm.sh + x * hm.sw;
        const float s = chl_sigmoid(chl_load(hm, off));
        const bool inside = s >= 1e-4f && s <= 1.0f - 1e-4f;
        const float p = fminf(fmaxf(s, 1e-4f), 1.0f - 1e-4f);
        float dp = 0.0f;
        if (g == 1.0f) {
            const float q = 1.0f - p;
            dp = q * q / p - 2.0f * q * logf(p);
        } else if (g < 1.0f) {
            const float w1 = 1.0f - g, w2 = w1 * w1;
            dp = (-(p * p) / (1.0f - p) + 2.0f * p * logf(1.0f - p)) * (w2 * w2);
        }
        chl_store_grad(hm, off, inside ? scale * dp * (s * (1.0f - s)) : 0.0f);
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

// A block per frame.  The masked objects are first compacted into LDS in object order (typically < 100 of the 500
// slots); then one thread per object: the FIRST object of a pixel adds up the gradients of all objects of that pixel (in
// object order) and stores them.
__global__ __launch_bounds__(256) void chl_scatter_kernel(ChlRegs regs, int W, const long long *__restrict__ ind,
                                                          const long long *__restrict__ mask,
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
        const bool on = m0 < M && mask[b * M + m0] != 0;
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
                    const float diff = pr - target[(size_t)obj_s[j] * regs.dims + d0 + c];
                    gsum += diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
                }
                chl_store_grad(mp, off, scale * code_weights[d0 + c] * gsum);
            }
            d0 += mp.c;
        }
    }
}

}  // namespace

extern "C" size_t pcd_centerhead_loss_workspace_bytes(int code_dims) {
    if (code_dims < 0 || code_dims > CHL_MAX_DIM) return 0;
    return (size_t)CHL_BLOCKS * (4 + CHL_MAX_DIM + 2) * sizeof(double);
}

extern "C" int pcd_centerhead_loss_forward(const void *hm, int hm_dtype, const long long *hm_strides_host,
                                           const float *gt_heatmap, int batch, int num_classes, int height, int width,
                                           const void *const *reg_ptrs_host, const int *reg_channels_host, int reg_dtype,
                                           const long long *reg_strides_host, int n_reg, const long long *inds,
                                           const long long *masks, const float *target_boxes, int num_max_objs,
                                           const float *code_weights, float cls_weight, float loc_weight, float *out,
                                           void *workspace, size_t workspace_bytes, void *stream) {
    PCD_ENTER();
    if (batch <= 0 || num_classes <= 0 || height <= 0 || width <= 0 || num_max_objs < 0 || !gt_heatmap || !out ||
        !workspace || (num_max_objs > 0 && n_reg > 0 && (!inds || !masks || !target_boxes || !code_weights)))
        return PCD_ERR_INVALID_ARG;
    ChlMap H_;
    ChlRegs R;
    int rc = chl_pack(hm, nullptr, hm_dtype, hm_strides_host, num_classes, reg_ptrs_host, nullptr, reg_channels_host,
                      reg_dtype, reg_strides_host, n_reg, &H_, &R);
    if (rc != PCD_OK) return rc;
    if (workspace_bytes < pcd_centerhead_loss_workspace_bytes(R.dims)) return PCD_ERR_WORKSPACE;
    if ((double)batch * (num_classes > CHL_MAX_DIM ? num_classes : CHL_MAX_DIM) * height * width >= 4294967295.0)
        return PCD_ERR_UNSUPPORTED;
    if (num_max_objs > CHL_MAX_OBJS) return PCD_ERR_UNSUPPORTED;   // (the backward pass could not follow)
    hipStream_t st = (hipStream_t)stream;
    const int pstride = 4 + CHL_MAX_DIM + 2;
    chl_forward_kernel<<<CHL_BLOCKS, 256, 0, st>>>(H_, gt_heatmap, batch, num_classes, height, width, R, inds, masks,
                                                   target_boxes, num_max_objs, (double *)workspace, pstride);
    chl_finalize_kernel<<<1, 256, 0, st>>>((const double *)workspace, pstride, R.dims, code_weights, cls_weight,
                                           loc_weight, out);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_centerhead_loss_backward(const void *hm, void *d_hm, int hm_dtype, const long long *hm_strides_host,
                                            const float *gt_heatmap, int batch, int num_classes, int height, int width,
                                            const void *const *reg_ptrs_host, void *const *reg_grads_host,
                                            const int *reg_channels_host, int reg_dtype,
                                            const long long *reg_strides_host, int n_reg, const long long *inds,
                                            const long long *masks, const float *target_boxes, int num_max_objs,
                                            const float *code_weights, float cls_weight, float loc_weight,
                                            const float *out, const float *grad_out, void *stream) {
    PCD_ENTER();
    if (batch <= 0 || num_classes <= 0 || height <= 0 || width <= 0 || num_max_objs < 0 || !gt_heatmap || !out ||
        !grad_out || !d_hm || (n_reg > 0 && !reg_grads_host) ||
        (num_max_objs > 0 && n_reg > 0 && (!inds || !masks || !target_boxes || !code_weights)))
        return PCD_ERR_INVALID_ARG;
    ChlMap H_;
    ChlRegs R;
    int rc = chl_pack(hm, d_hm, hm_dtype, hm_strides_host, num_classes, reg_ptrs_host, reg_grads_host,
                      reg_channels_host, reg_dtype, reg_strides_host, n_reg, &H_, &R);
    if (rc != PCD_OK) return rc;
    for (int r = 0; r < n_reg; ++r)
        if (!reg_grads_host[r]) return PCD_ERR_INVALID_ARG;
    if ((double)batch * (num_classes > CHL_MAX_DIM ? num_classes : CHL_MAX_DIM) * height * width >= 4294967295.0 ||
        (double)height * width >= 2147483647.0)
        return PCD_ERR_UNSUPPORTED;
    if (num_max_objs > CHL_MAX_OBJS) return PCD_ERR_UNSUPPORTED;   // chl_scatter_kernel compacts a frame's objects in LDS
    hipStream_t st = (hipStream_t)stream;
    chl_backward_kernel<<<1024, 256, 0, st>>>(H_, gt_heatmap, batch, num_classes, height, width, R, out, grad_out,
                                              cls_weight);
    if (n_reg > 0 && num_max_objs > 0)
        chl_scatter_kernel<<<batch, 256, 0, st>>>(R, width, inds, masks, target_boxes, num_max_objs, code_weights, out,
                                                  grad_out, loc_weight);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
