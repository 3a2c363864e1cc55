// BatchNorm reductions in the epilogue of the conv kernels (shared by spconv.hip and spconv_win.hip).
#pragma once
#include "common.h"
#include "bn_mid.h"

namespace {

// Optional per-channel reductions of the output tile in the gather-GEMM epilogue (mode 0 = off).  The values are
// in registers anyway; a separate kernel would read them back from memory:
//   mode 1  forward: sum / sum of squares of the ROUNDED outputs = the batch statistics of the BatchNorm that
//           follows the conv (replaces its statistics pass);
//   mode 2  data gradient: the output is dy of the BatchNorm (+ReLU) that produced this conv's input; accumulates
//           sum(dz) and sum(dz * xhat), dz = dy where y > 0 (ReLU), xhat = (x - mean) * invstd (replaces the
//           reduction pass of the BatchNorm backward; the second
//           sum is centred per lane: (sum dz*x - mean * sum dz) * invstd over the lane's <= 4 rows).
// One row of partial[gridDim.x][2][c_out] per workgroup (zeros from workgroups without rows); fixed summation
// tree (mi, DPP row, waves) -> deterministic.
struct BnRed {
    int mode;
    int relu;
    const unsigned short *x;      // mode 2: input of the BatchNorm, [rows][c_out] bf16
    const unsigned short *y;      // mode 2, relu: BatchNorm(+residual)+ReLU output (the conv's own input features)
    const float *mean, *invstd;
    float *partial;
    double *mid;                  // != NULL: the launch also folds the partial rows into MID_ROWS rows (below)
    int *counters;                // [MID_ROWS][BN_COUNTER_STRIDE], zero between launches
};


// ---- the BatchNorm "mid" reduction inside the conv launch ------------------------------------------------------
// pcd_bn_forward / pcd_bn_backward fold the partial rows of a launch into BN_MID_ROWS rows of doubles with a kernel of
// their own (fused.hip: bn_mid_kernel, mid row r = sum of the partial rows t with t % 16 == r) before the apply pass;
// that kernel is 16 small workgroups, but as a graph node on the critical chain it costs ~10 us of latency, 42 times
// per step.  With BnRed.mid the conv launch does it: the workgroup that delivers the LAST partial row of a group
// (a counter per group; rows are published with agent-scope stores and read back with agent-scope loads -- plain
// stores are only visible to other XCDs after an L2 write-back) sums the group's rows, in a fixed order of its own
// (doubles: it agrees with bn_mid_kernel's order to ~1e-16 relative).  The counter returns to zero for the next launch.
// partial row `tile` = get(e), e in [0, 2 c_out); all threads of the workgroup, uniformly
// returns (to every thread): this workgroup delivered the last row of its group and wrote the group's mid row
template <class F>
__device__ __forceinline__ bool bnred_publish(const BnRed &bn, int tile, int c_out, F get, int nrows_arg = -1) {
    float *dst = bn.partial + (size_t)tile * 2 * c_out;
    if (!bn.mid) {
        for (int e = threadIdx.x; e < 2 * c_out; e += blockDim.x) dst[e] = get(e);
        return false;
    }
    extern __shared__ __attribute__((aligned(16))) char smem_base[];   // the launch's dynamic LDS (>= 8 KiB, free now)
    __shared__ int last_s;
    for (int e = threadIdx.x; e < 2 * c_out; e += blockDim.x)
        __hip_atomic_store(dst + e, get(e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the row has reached the coherence point ...
    __syncthreads();
    const int nrows = nrows_arg >= 0 ? nrows_arg : (int)gridDim.x, r = tile & (BN_MID_ROWS - 1);
    if (tile == 0)                                         // fewer tiles than groups: the empty groups' rows are zero
        for (int e = nrows * 2 * c_out + threadIdx.x; e < BN_MID_ROWS * 2 * c_out; e += blockDim.x) bn.mid[e] = 0.0;
    if (threadIdx.x == 0) {                                // ... before this workgroup counts as arrived
        // (one counter per 128-byte line: agent-scope atomics on ONE line serialise at ~35 ns each -- 46 us for the
        //  1320 workgroups of a level-1 conv when the 16 counters shared a line)
        int *cnt = bn.counters + r * BN_COUNTER_STRIDE;
        const int expect = (nrows - r + BN_MID_ROWS - 1) / BN_MID_ROWS;
        const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = old == expect - 1;
        if (last_s) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (last_s) {
        // Memory-model note: publication is done at the ISA level -- every row element is an agent-scope (sc1, write-
        // through to the device coherence point) store whose completion the writer awaits (s_waitcnt vmcnt(0)) before
        // the workgroup barrier that precedes its counter increment, and bn_mid_row reads with agent-scope loads.  A
        // C++-level release on the fetch_add would add an L2 write-back (buffer_wbl2) to EVERY workgroup of the conv
        // launch; the acquire side is cheap (16 workgroups per launch) and is taken here so that nothing the compiler
        // or a cache could hold from before the counter read is reused.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        bn_mid_row(bn.partial, nrows, c_out, r, bn.mid, reinterpret_cast<double *>(smem_base));
    }
    return last_s != 0;
}

__device__ __forceinline__ void bnred_zero_row(const BnRed &bn, int tile, int c_out) {
    bnred_publish(bn, tile, c_out, [](int) { return 0.0f; });
}


// host: PcdBnReduce (C ABI) -> kernel argument; `grid` partial rows will be written
static int make_bnred(const PcdBnReduce *r, int y_dtype, int c_out, int grid, BnRed *out) {
    BnRed b = {};
    *out = b;
    if (!r || r->mode == 0) return PCD_OK;
    if (r->mode != 1 && r->mode != 2) return PCD_ERR_INVALID_ARG;
    if (y_dtype != PCD_BF16) return PCD_ERR_UNSUPPORTED;
    if (!r->partial || r->partial_rows < grid) return PCD_ERR_INVALID_ARG;
    if (r->mode == 2) {
        if (!r->x || !r->mean || !r->invstd) return PCD_ERR_INVALID_ARG;
        if (r->relu && !r->y) return PCD_ERR_INVALID_ARG;
        if (((uintptr_t)r->mean | (uintptr_t)r->invstd) & 15u)
            return PCD_ERR_UNSUPPORTED;   // the epilogue reads 4 channels per 16-byte load
    }
    (void)c_out;
    b.mode = r->mode;
    b.relu = r->relu;
    b.x = (const unsigned short *)r->x;
    b.y = (const unsigned short *)r->y;
    b.mean = r->mean;
    b.invstd = r->invstd;
    b.partial = r->partial;
    if (r->mid) {
        if (!r->counters || r->partial_rows != grid) return PCD_ERR_INVALID_ARG;   // groups are counted over the grid
        b.mid = r->mid;
        b.counters = r->counters;
    }
    *out = b;
    return PCD_OK;
}


}  // namespace
