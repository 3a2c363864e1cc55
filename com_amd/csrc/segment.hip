// Segment reductions over per-point pillar ids for the dynamic pillar encoder
// (pcdet/models/backbones_3d/vfe/dynamic_pillar_vfe.py:36-47: torch_scatter.scatter_max of the PFN output over
// `unq_inv`, and its gradient).  Points of a pillar are scattered over the cloud (the ids come from the bitmap ranks
// of pcd_voxelize_dynamic_mean, the points keep their order), so the maximum is taken with ONE 64-bit atomicMax per
// (point, channel) on  (order-preserving float bits << 32 | ~point index):  the largest value wins, among equal
// values the SMALLEST point index -- the result and the argmax do not depend on the execution order.
#include "common.h"

namespace {

typedef unsigned long long u64s;

// monotone map float -> u32 (total order of the values, -0 < +0)
__device__ __forceinline__ u32 ordered_bits(float v) {
    const u32 b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_ordered(u32 o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__global__ __launch_bounds__(256) void seg_max_kernel(const float *__restrict__ x, const int32_t *__restrict__ seg,
                                                      int n, int c, int m, u64s *__restrict__ best) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)n * c) return;
    const int i = (int)(e / c), ch = (int)(e - (size_t)i * c);
    const int s = seg[i];
    if (s < 0 || s >= m) return;
    const u64s key = ((u64s)ordered_bits(x[e]) << 32) | (u64s)(~(u32)i);
    u64s *slot = best + (size_t)s * c + ch;
    if (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < key) atomicMax(slot, key);
}

__global__ __launch_bounds__(256) void seg_max_decode_kernel(const u64s *__restrict__ best, size_t total,
                                                             float *__restrict__ out, int32_t *__restrict__ arg) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const u64s k = best[e];
    if (k == 0ull) {            // empty segment (cannot happen for ids made by the voxeliser): torch_scatter leaves 0
        out[e] = 0.0f;
        arg[e] = -1;
        return;
    }
    out[e] = from_ordered((u32)(k >> 32));
    arg[e] = (int32_t)(~(u32)k);
}

__global__ __launch_bounds__(256) void seg_max_bwd_kernel(const float *__restrict__ gout,
                                                          const int32_t *__restrict__ arg, size_t total, int c,
                                                          float *__restrict__ gx) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int i = arg[e];
    if (i >= 0) gx[(size_t)i * c + (e % c)] = gout[e];   // one writer per (point, channel): a point has one pillar
}

}  // namespace

extern "C" size_t pcd_segment_max_workspace_bytes(int m, int c) {
    if (m < 0 || c <= 0) return 0;
    return ws_piece((size_t)(m > 0 ? m : 1) * c, sizeof(u64s));
}

extern "C" int pcd_segment_max(const float *x, const int32_t *seg, int n, int c, int m, float *out, int32_t *arg,
                               void *workspace, size_t workspace_bytes, void *stream) {
    PCD_ENTER();
    if (n < 0 || m < 0 || c <= 0) return PCD_ERR_INVALID_ARG;
    if (m == 0) return PCD_OK;
    if (!out || !arg || (n > 0 && (!x || !seg))) return PCD_ERR_INVALID_ARG;
    WsCarver ws(workspace, workspace_bytes);
    u64s *best = ws.take<u64s>((size_t)m * c);
    if (!ws.ok) return PCD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)m * c;
    pcd_fill(best, 0, total * sizeof(u64s), st);
    if (n > 0) seg_max_kernel<<<(unsigned)(((size_t)n * c + 255) / 256), 256, 0, st>>>(x, seg, n, c, m, best);
    seg_max_decode_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(best, total, out, arg);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_segment_max_backward(const float *grad_out, const int32_t *arg, int n, int c, int m, float *grad_x,
                                        void *stream) {
    PCD_ENTER();
    if (n < 0 || m < 0 || c <= 0) return PCD_ERR_INVALID_ARG;
    if (n > 0 && !grad_x) return PCD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n > 0) pcd_fill(grad_x, 0, (size_t)n * c * sizeof(float), st);
    if (m > 0 && n > 0) {
        if (!grad_out || !arg) return PCD_ERR_INVALID_ARG;
        const size_t total = (size_t)m * c;
        seg_max_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(grad_out, arg, total, c, grad_x);
    }
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
