// Gradient clipping + Adam over ONE flat fp32 parameter buffer, in two launches (the exchange / update end of the
// data-parallel step: tools/train.py:165-166 wraps the model in DDP, tools/train_utils/train_utils.py:93-96 clips
// the gradient norm and steps the optimizer).  torch's route is clip_grad_norm_ (norm + five scalar kernels + a
// scaling pass) and a multi-tensor Adam kernel: ~200 us on the tail of every step for a 2.7 M-parameter backbone;
// here: one pass for the squared norm, one pass that derives the clip coefficient from the 256 partials and
// applies torch.optim.Adam's update (L2 weight decay, bias correction) -- same formula, fp32.
#include "common.h"

namespace {

constexpr int NORM_BLOCKS = 256;

__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float4 *__restrict__ g, size_t n4,
                                                             const float *__restrict__ tail, int ntail,
                                                             double *__restrict__ partial, float *step_dev = nullptr) {
    __shared__ double lds[4];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)NORM_BLOCKS * 256) {
        float4 v = g[i];
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) s += (double)tail[threadIdx.x] * tail[threadIdx.x];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    // the update counter moves HERE, in the launch in front of the Adam pass (which then finds it already incremented in every
    // block): no one-thread kernel behind it on the serial tail of the step
    if (step_dev && blockIdx.x == 0 && threadIdx.x == 0) step_dev[0] += 1.0f;
}

// step_dev[0] holds the number of updates done so far (a device float, as torch keeps it for capturable
// optimizers, so the step can be replayed from a hipGraph) -- incremented by the norm launch in front of this one: THIS update
// is number step_dev[0], the schedule row step_dev[0] - 1.  zero_grad: the gradient is cleared as it is consumed (the next
// step's zero_grad, without a fill launch).
__global__ __launch_bounds__(256) void adam_flat_kernel(float4 *__restrict__ p, float4 *__restrict__ g, int zero_grad,
                                                        float4 *__restrict__ m, float4 *__restrict__ v, size_t n4,
                                                        const double *__restrict__ norm_partial, float max_norm,
                                                        float pre_divisor, const float *__restrict__ step_dev,
                                                        float lr, float beta1, float beta2, float eps, float wd,
                                                        float *__restrict__ norm_out, int decoupled,
                                                        float *__restrict__ hyper_dev,
                                                        const float *__restrict__ schedule, int schedule_len) {
    __shared__ double total_s;
    if (threadIdx.x < 64) {
        double s = 0.0;
        for (int i = threadIdx.x; i < NORM_BLOCKS; i += 64) s += norm_partial[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (threadIdx.x == 0) total_s = s;
    }
    __syncthreads();
    const float norm = (float)(sqrt(total_s) / (double)pre_divisor);          // norm of the rank-mean gradient
    const float coef = fminf(max_norm > 0.0f ? max_norm / (norm + 1e-6f) : 1.0f, 1.0f);
    const float gscale = coef / pre_divisor;                                  // g_used = g_sum * gscale
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = norm;
    if (schedule) {             // the whole OneCycle table on the device: row = number of updates done so far
        int row = (int)step_dev[0] - 1;
        row = row < schedule_len - 1 ? row : schedule_len - 1;
        lr = schedule[2 * row];
        beta1 = schedule[2 * row + 1];
        if (hyper_dev && blockIdx.x == 0 && threadIdx.x == 0) {      // (kept observable; nobody in this launch reads it)
            hyper_dev[0] = lr;
            hyper_dev[1] = beta1;
        }
    } else if (hyper_dev) {     // lr / beta1 of THIS step from device memory (set_hyper driving a replayed graph)
        lr = hyper_dev[0];
        beta1 = hyper_dev[1];
    }
    const float decay = decoupled ? 1.0f - lr * wd : 1.0f;     // fastai_optim.py:135-150: p *= 1 - wd * lr, then wd = 0
    const float wd_l2 = decoupled ? 0.0f : wd;
    const float step = step_dev[0];
    const float bc1 = 1.0f - powf(beta1, step);
    const float bc2 = 1.0f - powf(beta2, step);
    const float step_size = lr / bc1;
    const float bc2_sqrt = sqrtf(bc2);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        float P[4] = {pp.x, pp.y, pp.z, pp.w}, G[4] = {gg.x, gg.y, gg.z, gg.w}, M[4] = {mm.x, mm.y, mm.z, mm.w},
              V[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            P[j] *= decay;
            float gr = G[j] * gscale + wd_l2 * P[j];
            M[j] = M[j] + (gr - M[j]) * (1.0f - beta1);
            V[j] = beta2 * V[j] + (1.0f - beta2) * gr * gr;
            float denom = sqrtf(V[j]) / bc2_sqrt + eps;
            P[j] = P[j] - step_size * (M[j] / denom);
        }
        p[i] = make_float4(P[0], P[1], P[2], P[3]);
        m[i] = make_float4(M[0], M[1], M[2], M[3]);
        v[i] = make_float4(V[0], V[1], V[2], V[3]);
        if (zero_grad) g[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

// <a, b> of two bf16 vectors (fp32 products, double partial sums) and y = a * s[0]: the forward / backward of a linear
// functional of the BEV map (bench.py's stand-in for the dense head's loss when only the sparse hot path is timed) in
// 2 + 1 launches instead of torch's cast / mul / sum chain.
constexpr int DOT_BLOCKS = 512;
__global__ __launch_bounds__(256) void dot_bf16_partials_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ b,
                                                                size_t n8, double *__restrict__ partial) {
    __shared__ double lds[4];
    float acc = 0.0f;
    double s = 0.0;
    int k = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)DOT_BLOCKS * 256) {
        const uint4 va = a[i], vb = b[i];
        const u32 wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc += __uint_as_float(wa[j] << 16) * __uint_as_float(wb[j] << 16);
            acc += __uint_as_float(wa[j] & 0xffff0000u) * __uint_as_float(wb[j] & 0xffff0000u);
        }
        if (++k == 16) {                     // spill the fp32 run into the double sum every 128 products
            s += (double)acc;
            acc = 0.0f;
            k = 0;
        }
    }
    s += (double)acc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}
__global__ __launch_bounds__(64) void dot_finalize_kernel(const double *__restrict__ partial, float *__restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < DOT_BLOCKS; i += 64) s += partial[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) out[0] = (float)s;
}
__global__ __launch_bounds__(256) void scale_bf16_kernel(const uint4 *__restrict__ a, const float *__restrict__ sc,
                                                         size_t n8, uint4 *__restrict__ y) {
    const float s = sc[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const uint4 va = a[i];
        const u32 w[4] = {va.x, va.y, va.z, va.w};
        u32 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = (u32)f32_to_bf16_bits(__uint_as_float(w[j] << 16) * s) |
                   ((u32)f32_to_bf16_bits(__uint_as_float(w[j] & 0xffff0000u) * s) << 16);
        y[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// device clock (100 MHz, shared by all queues) into slot[0]: time points INSIDE a replayed hipGraph, where neither
// events nor the profiler (which changes how the graph is scheduled) can be used
__global__ void stamp_kernel(unsigned long long *slot) { slot[0] = wall_clock64(); }

// Sticky capacity-overflow flag of static-shape execution: flag[0] |= 1 (and flag[1] = max over-capacity count
// seen) when any device-side row count exceeds the capacity its buffers were allocated with.  One thread.
__global__ void overflow_check_kernel(PcdCountCheck tab, int n, int32_t *flag) {
    int bad = 0, worst = 0;
    for (int i = 0; i < n; ++i) {
        const int v = *tab.count[i];
        if (v > tab.cap[i]) {
            bad = 1;
            worst = max(worst, v - tab.cap[i]);
        }
    }
    if (bad) {
        flag[0] = 1;
        if (worst > flag[1]) flag[1] = worst;
    }
}

}  // namespace

extern "C" int pcd_static_overflow_check(const PcdCountCheck *table_host, int n, int32_t *flag, void *stream) {
    PCD_ENTER();
    if (!table_host || !flag || n < 0 || n > PCD_COUNT_CHECK_MAX) return PCD_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i)
        if (!table_host->count[i]) return PCD_ERR_INVALID_ARG;
    if (n == 0) return PCD_OK;
    overflow_check_kernel<<<1, 1, 0, (hipStream_t)stream>>>(*table_host, n, flag);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" size_t pcd_adam_flat_workspace_bytes(void) { return ws_piece(NORM_BLOCKS, sizeof(double)); }

extern "C" int pcd_adam_flat_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                                  float lr, float beta1, float beta2, float eps, float weight_decay,
                                  float max_norm, float pre_divisor, float *step_dev, float *norm_out,
                                  void *workspace, size_t workspace_bytes, void *stream) {
    return pcd_adam_flat_step_v2(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, max_norm,
                                 pre_divisor, 0, nullptr, step_dev, norm_out, workspace, workspace_bytes, stream);
}

extern "C" int pcd_adam_flat_step_v2(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                                     float lr, float beta1, float beta2, float eps, float weight_decay,
                                     float max_norm, float pre_divisor, int decoupled_wd, const float *hyper_dev,
                                     float *step_dev, float *norm_out, void *workspace, size_t workspace_bytes,
                                     void *stream) {
    return pcd_adam_flat_step_v3(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, max_norm,
                                 pre_divisor, decoupled_wd, const_cast<float *>(hyper_dev), nullptr, 0, step_dev,
                                 norm_out, workspace, workspace_bytes, stream);
}

extern "C" int pcd_adam_flat_step_v3(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                                     float lr, float beta1, float beta2, float eps, float weight_decay,
                                     float max_norm, float pre_divisor, int decoupled_wd, float *hyper_dev,
                                     const float *schedule_dev, int schedule_len, float *step_dev, float *norm_out,
                                     void *workspace, size_t workspace_bytes, void *stream) {
    return pcd_adam_flat_step_v4(param, const_cast<float *>(grad), 0, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                                 max_norm, pre_divisor, decoupled_wd, hyper_dev, schedule_dev, schedule_len, step_dev, norm_out,
                                 workspace, workspace_bytes, stream);
}

extern "C" int pcd_adam_flat_step_v4(float *param, float *grad, int zero_grad, float *exp_avg, float *exp_avg_sq, size_t n,
                                     float lr, float beta1, float beta2, float eps, float weight_decay,
                                     float max_norm, float pre_divisor, int decoupled_wd, float *hyper_dev,
                                     const float *schedule_dev, int schedule_len, float *step_dev, float *norm_out,
                                     void *workspace, size_t workspace_bytes, void *stream) {
    PCD_ENTER();
    if (schedule_dev && schedule_len <= 0) return PCD_ERR_INVALID_ARG;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev || pre_divisor <= 0.0f) return PCD_ERR_INVALID_ARG;
    if ((n & 3) != 0) return PCD_ERR_UNSUPPORTED;   // the flat buffer is padded to a multiple of 4 by its owner
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15u) != 0)
        return PCD_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < pcd_adam_flat_workspace_bytes()) return PCD_ERR_WORKSPACE;
    if (n == 0) return PCD_OK;
    hipStream_t st = (hipStream_t)stream;
    double *partial = (double *)workspace;
    const size_t n4 = n / 4;
    sumsq_partials_kernel<<<NORM_BLOCKS, 256, 0, st>>>((const float4 *)grad, n4, nullptr, 0, partial, step_dev);
    int blocks = (int)((n4 + 1023) / 1024);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    adam_flat_kernel<<<blocks, 256, 0, st>>>((float4 *)param, (float4 *)grad, zero_grad ? 1 : 0, (float4 *)exp_avg,
                                             (float4 *)exp_avg_sq, n4, partial, max_norm, pre_divisor, step_dev, lr,
                                             beta1, beta2, eps, weight_decay, norm_out, decoupled_wd, hyper_dev,
                                             schedule_dev, schedule_len);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" size_t pcd_dot_bf16_workspace_bytes(void) { return ws_piece(DOT_BLOCKS, sizeof(double)); }

extern "C" int pcd_dot_bf16(const void *a, const void *b, size_t n, float *out, void *workspace, size_t workspace_bytes,
                            void *stream) {
    PCD_ENTER();
    if (!a || !b || !out || (n & 7) != 0 || ((((uintptr_t)a) | ((uintptr_t)b)) & 15u) != 0) return PCD_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < pcd_dot_bf16_workspace_bytes()) return PCD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    dot_bf16_partials_kernel<<<DOT_BLOCKS, 256, 0, st>>>((const uint4 *)a, (const uint4 *)b, n / 8, (double *)workspace);
    dot_finalize_kernel<<<1, 64, 0, st>>>((const double *)workspace, out);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_scale_bf16(const void *a, const float *scale_dev, size_t n, void *y, void *stream) {
    PCD_ENTER();
    if (!a || !y || !scale_dev || (n & 7) != 0 || ((((uintptr_t)a) | ((uintptr_t)y)) & 15u) != 0) return PCD_ERR_INVALID_ARG;
    if (n == 0) return PCD_OK;
    size_t blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    scale_bf16_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const uint4 *)a, scale_dev, n / 8, (uint4 *)y);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_debug_stamp(uint64_t *slot, void *stream) {
    PCD_ENTER();
    if (!slot) return PCD_ERR_INVALID_ARG;
    stamp_kernel<<<1, 1, 0, (hipStream_t)stream>>>((unsigned long long *)slot);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

// ---------------------------------------------------------------------------------------------
// (a3) Host -> device inside the captured step without a copy engine, a second stream or the host: the collated points
// of every batch sit in PINNED host memory (device-visible under HIP's unified addressing); a kernel of the step's own
// graph reads slot (*counter % n_slots) of a device-side table of those host pointers over PCIe and writes the staging
// buffer the step's voxeliser reads.  A replayed graph cannot change its arguments, the table + counter can: every replay
// pulls the next batch.  (The copy-stream form of the same thing was bimodal on the gpurun boxes, 3.5 vs 4.2-4.8 ms per
// step: how the copy queue is arbitrated against the queues the hipGraph executor uses; bench.py H2DSource.)
namespace {
__global__ __launch_bounds__(256) void pull_host_kernel(const unsigned long long *__restrict__ table, int n_slots,
                                                        const int32_t *__restrict__ counter, uint4 *__restrict__ dst,
                                                        size_t n16) {
    const int slot = (int)((unsigned)(*counter) % (unsigned)n_slots);
    const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(table[slot]);
    const size_t S = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * S < n16; i += 4 * S) {                 // 4 x 16 bytes in flight per thread: PCIe reads are latency-bound
        const uint4 a = src[i], b = src[i + S], c = src[i + 2 * S], d = src[i + 3 * S];
        dst[i] = a; dst[i + S] = b; dst[i + 2 * S] = c; dst[i + 3 * S] = d;
    }
    for (; i < n16; i += S) dst[i] = src[i];
}
__global__ void counter_add_kernel(int32_t *counter, int delta) { *counter += delta; }
}  // namespace

extern "C" int pcd_pull_from_host(const void *host_ptr_table_dev, int n_slots, const int32_t *counter_dev, void *dst,
                                  size_t bytes, int workgroups, void *stream) {
    PCD_ENTER();
    if (!host_ptr_table_dev || n_slots <= 0 || !counter_dev || !dst || (bytes & 15) || ((uintptr_t)dst & 15))
        return PCD_ERR_INVALID_ARG;
    if (bytes == 0) return PCD_OK;
    if (workgroups <= 0) workgroups = 64;
    pull_host_kernel<<<workgroups, 256, 0, (hipStream_t)stream>>>((const unsigned long long *)host_ptr_table_dev, n_slots,
                                                                 counter_dev, (uint4 *)dst, bytes / 16);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_counter_add(int32_t *counter_dev, int delta, void *stream) {
    PCD_ENTER();
    if (!counter_dev) return PCD_ERR_INVALID_ARG;
    counter_add_kernel<<<1, 1, 0, (hipStream_t)stream>>>(counter_dev, delta);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
