// Shared pieces of the CenterHead kernels (centerhead.hip: plain CenterHead; com_head.hip: the COM curriculum head).
#pragma once
#include "common.h"

namespace {

struct AssignGeom {
    float range_x, range_y, vs_x, vs_y;
    int stride, W, H, num_max, min_radius, code, head_classes;
    float overlap;
    int cls_map[16];       // dataset class id (1-based; 0 = padding) -> 1-based id inside this head, 0 = not in the head
};

// centernet_utils.py:46-72 in float32, operation by operation
__device__ __forceinline__ float gaussian_radius_f32(float height, float width, float min_overlap) {
    const float a1 = 1.0f;
    const float b1 = height + width;
    const float c1 = width * height * (1.0f - min_overlap) / (1.0f + min_overlap);
    const float sq1 = sqrtf(b1 * b1 - 4.0f * a1 * c1);
    const float r1 = (b1 + sq1) / 2.0f;
    const float a2 = 4.0f;
    const float b2 = 2.0f * (height + width);
    const float c2 = (1.0f - min_overlap) * width * height;
    const float sq2 = sqrtf(b2 * b2 - 4.0f * a2 * c2);
    const float r2 = (b2 + sq2) / 2.0f;
    const float a3 = 4.0f * min_overlap;
    const float b3 = -2.0f * min_overlap * (height + width);
    const float c3 = (min_overlap - 1.0f) * width * height;
    const float sq3 = sqrtf(b3 * b3 - 4.0f * a3 * c3);
    const float r3 = (b3 + sq3) / 2.0f;
    return fminf(fminf(r1, r2), r3);
}

// one wave per (batch, object): max the object's Gaussian into its class plane (centernet_utils.py:75-107)
__global__ __launch_bounds__(64) void draw_gaussian_kernel(const int4 *__restrict__ draw, AssignGeom G,
                                                           float *__restrict__ heatmap) {
    const size_t at = blockIdx.x;
    const int4 d = draw[at];
    if (d.x < 0) return;
    const int b = (int)(at / G.num_max);
    const int radius = d.w, x = d.y, y = d.z;
    const int left = min(x, radius), right = min(G.W - x, radius + 1);
    const int top = min(y, radius), bottom = min(G.H - y, radius + 1);
    const int w = left + right, h = top + bottom;
    if (w <= 0 || h <= 0) return;
    const double sigma = (double)(2 * radius + 1) / 6.0;
    const double eps_cut = 2.220446049250313e-16;             // np.finfo(float64).eps * h.max(), h.max() = 1
    int *plane = reinterpret_cast<int *>(heatmap + ((size_t)b * G.head_classes + d.x) * G.H * G.W);
    for (int p = threadIdx.x; p < w * h; p += 64) {
        const int py = p / w, px = p - py * w;
        const int gx = px - left, gy = py - top;               // offset from the centre
        double v = exp(-(double)(gx * gx + gy * gy) / (2.0 * sigma * sigma));
        if (v < eps_cut) v = 0.0;
        const float f = (float)v;
        atomicMax(plane + (size_t)(y + gy) * G.W + (x + gx), __float_as_int(f));
    }
}

constexpr int CHL_BLOCKS = 256;     // partial rows of the forward pass
constexpr int CHL_MAX_REG = 8;      // regression branches per head
constexpr int CHL_MAX_DIM = 16;     // code dimensions
constexpr int CHL_MAX_OBJS = 2048;  // objects per frame (num_max_objs; the reference uses 500): LDS list of the backward

struct ChlMap {                     // one prediction map [B][c][H][W] behind strides
    const void *p;
    void *g;                        // its gradient (same layout), backward only
    long long sb, sc, sh, sw;
    int c, dtype;                   // PCD_F32 / PCD_BF16
};
struct ChlRegs {
    ChlMap m[CHL_MAX_REG];
    int n, dims;                    // dims = sum of c
};

__device__ __forceinline__ float chl_load(const ChlMap &m, long long off) {
    return m.dtype == PCD_BF16 ? bf16_bits_to_f32(((const unsigned short *)m.p)[off]) : ((const float *)m.p)[off];
}
__device__ __forceinline__ void chl_store_grad(const ChlMap &m, long long off, float v) {
    if (m.dtype == PCD_BF16) ((unsigned short *)m.g)[off] = f32_to_bf16_bits(v);
    else ((float *)m.g)[off] = v;
}
__device__ __forceinline__ float chl_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ double chl_block_sum(double v, double *lds /*[4]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    const double t = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return t;
}

static int chl_pack(const void *hm, void *d_hm, int hm_dtype, const long long *hm_strides, int C,
                    const void *const *reg_ptrs, void *const *reg_grads, const int *reg_channels, int reg_dtype,
                    const long long *reg_strides, int n_reg, ChlMap *H_, ChlRegs *R) {
    if (!hm || !hm_strides || n_reg < 0 || n_reg > CHL_MAX_REG || (n_reg > 0 && (!reg_ptrs || !reg_channels || !reg_strides)))
        return PCD_ERR_INVALID_ARG;
    if ((hm_dtype != PCD_F32 && hm_dtype != PCD_BF16) || (reg_dtype != PCD_F32 && reg_dtype != PCD_BF16))
        return PCD_ERR_UNSUPPORTED;
    *H_ = ChlMap{hm, d_hm, hm_strides[0], hm_strides[1], hm_strides[2], hm_strides[3], C, hm_dtype};
    R->n = n_reg;
    R->dims = 0;
    for (int r = 0; r < n_reg; ++r) {
        if (!reg_ptrs[r] || reg_channels[r] <= 0) return PCD_ERR_INVALID_ARG;
        R->m[r] = ChlMap{reg_ptrs[r], reg_grads ? reg_grads[r] : nullptr, reg_strides[4 * r], reg_strides[4 * r + 1],
                         reg_strides[4 * r + 2], reg_strides[4 * r + 3], reg_channels[r], reg_dtype};
        R->dims += reg_channels[r];
    }
    if (R->dims > CHL_MAX_DIM) return PCD_ERR_UNSUPPORTED;
    return PCD_OK;
}

}  // namespace
