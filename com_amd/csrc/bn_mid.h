// Shared by the conv kernels that take BatchNorm sums in their epilogue (spconv.hip, conv2d.hip): the DPP row sum and
// the "mid" reduction run by the workgroup that delivers the last partial row of a group (see spconv.hip).
#pragma once
#include "common.h"

namespace {

// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), result in all of them
__device__ __forceinline__ float row16_sum(float v) {
    int t;
    t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, t);
    t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, t);
    t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true);  // row_half_mirror
    v += __builtin_bit_cast(float, t);
    t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true);  // row_mirror
    v += __builtin_bit_cast(float, t);
    return v;
}

constexpr int BN_MID_ROWS = 16;
constexpr int BN_COUNTER_STRIDE = PCD_BN_COUNTER_STRIDE;   // ints between two group counters

__device__ __forceinline__ float ld_agent(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// mid[r][col] = sum of the partial rows r, r + 16, r + 32, ... (all threads of the workgroup; `lds`: 1024 doubles).
// tpc threads per column take every tpc-th row of the group, 8 loads in flight, then one thread per column adds the
// tpc sums in order: fixed order, ~20 registers (the narrow conv kernels run at 7 waves / SIMD, i.e. 72 VGPRs).
__device__ __forceinline__ void bn_mid_row(const float *partial, int nblocks, int c, int r, double *__restrict__ mid,
                                           double *lds) {
    const int cols = 2 * c;
    int cw = 1;                                            // columns handled per pass: a power of two <= blockDim
    while (cw < cols && cw < (int)blockDim.x) cw <<= 1;
    const int tpc = (int)blockDim.x / cw;                  // >= 1 (blockDim is a power of two)
    const int j = threadIdx.x / cw, t = threadIdx.x - j * cw;
    const int stride = BN_MID_ROWS * tpc;
    for (int col0 = 0; col0 < cols; col0 += cw) {
        const int col = col0 + t;
        double a = 0.0;
        if (col < cols) {
            for (int blk = r + BN_MID_ROWS * j; blk < nblocks; blk += 8 * stride) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int bq = blk + q * stride;
                    v[q] = ld_agent(partial + (size_t)(bq < nblocks ? bq : blk) * cols + col);
                    if (bq >= nblocks) v[q] = 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) a += (double)v[q];
            }
        }
        lds[threadIdx.x] = a;
        __syncthreads();
        if (j == 0 && col < cols) {
            double s2 = 0.0;
            for (int q = 0; q < tpc; ++q) s2 += lds[q * cw + t];
            // (agent scope: a launch that applies the BatchNorm itself -- spconv_win.hip, PcdBnFold -- reads the rows from other XCDs)
            __hip_atomic_store(mid + (size_t)r * cols + col, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
}


}  // namespace
