// Shared device/host helpers for libpcdops_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pcd_ops.h"

#define PCD_WAVE 64
#define PCD_BLOCK 256

// hipGetLastError() is per host thread and sticky across unrelated runtime calls (torch's own
// probing included): every entry point clears it on entry and checks it after its launches.
extern "C" void pcd_set_last_hip_error(int code);
#define PCD_ENTER() (void)hipGetLastError()
#define PCD_RETURN_IF_LAUNCH_FAILED()                  \
    do {                                               \
        hipError_t e__ = hipGetLastError();            \
        if (e__ != hipSuccess) {                       \
            pcd_set_last_hip_error((int)e__);          \
            return PCD_ERR_LAUNCH;                     \
        }                                              \
    } while (0)

// tuning options (pcd_ops.h: pcd_set_option); the table lives in capi.hip
enum PcdOpt {
    PCD_OPT_GG_RESIDENT_KB, PCD_OPT_GGW, PCD_OPT_GG1, PCD_OPT_SUBM_WINDOW, PCD_OPT_SUBM_WINDOW_WGRAD, PCD_OPT_WG128, PCD_OPT_WG128_CHUNKS,
    PCD_OPT_WG_ROWS, PCD_OPT_CONV2D_WB, PCD_OPT_CONV2D_WG_BLOCKS, PCD_OPT_CONV2D_WGP_MODE2, PCD_OPT_CONV2D_WGP_BLOCKS,
    PCD_OPT_FPS_G, PCD_OPT_GG_DBG, PCD_OPT_GGW_DBG, PCD_OPT_WIN_DBG, PCD_OPT_CM_DIRECT_BLOCKS, PCD_OPT_SUBM_WINDOW_GRID, PCD_OPT_GGWIN, PCD_OPT_GG2, PCD_OPT_GGW_MI, PCD_OPT_GGW_CW, PCD_OPT_VOX_EMIT_ROWS, PCD_OPT_VOX_GRID, PCD_OPT_SUBM_WINDOW_HALF, PCD_OPT_COUNT
};
int pcd_opt(int which);

typedef unsigned long long u64;
typedef unsigned int u32;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static inline size_t pcd_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int pcd_div_up(int a, int b) { return (a + b - 1) / b; }

// Bump allocator over a caller-provided workspace (256-B aligned pieces).
struct WsCarver {
    char *base;
    size_t off;
    size_t cap;
    bool ok;
    WsCarver(void *p, size_t bytes) : base((char *)p), off(0), cap(bytes), ok(true) {}
    template <class T>
    T *take(size_t n) {
        size_t bytes = pcd_align_up(n * sizeof(T), 256);
        if (base == nullptr || off + bytes > cap) {
            ok = false;
            off += bytes;
            return nullptr;
        }
        T *r = (T *)(base + off);
        off += bytes;
        return r;
    }
};
static inline size_t ws_piece(size_t n, size_t elem) { return pcd_align_up(n * elem, 256); }

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    u32 u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                                  // RNE
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((u32)b) << 16);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Row count of a launch: `n_cap` is the host-side capacity (grid size, strides); when `n_dev` is non-NULL the
// real count lives in device memory (so the caller never has to read it back -> hipGraph capturable) and is
// clamped to the capacity.
__device__ __forceinline__ int eff_rows(const int32_t *n_dev, int n_cap) {
    if (!n_dev) return n_cap;
    int v = *n_dev;
    return v < n_cap ? (v < 0 ? 0 : v) : n_cap;
}

// exclusive prefix of a 0/1 predicate inside a wave + wave total (ballot / popcount)
__device__ __forceinline__ int wave_rank(bool pred, int &total) {
    u64 m = __ballot(pred);
    total = __popcll(m);
    u64 lt = (lane_id() == 0) ? 0ull : (~0ull >> (64 - lane_id()));
    return __popcll(m & lt);
}

// inclusive scan of an int across the 64 lanes of a wave
__device__ __forceinline__ int wave_inclusive_scan(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane_id() >= d) v += t;
    }
    return v;
}

// exclusive scan over a 256-thread block; `lds` needs 4 ints; returns exclusive prefix, sets total
__device__ __forceinline__ int block_exclusive_scan(int v, int *lds, int &total) {
    int inc = wave_inclusive_scan(v);
    int w = threadIdx.x >> 6;
    if (lane_id() == 63) lds[w] = inc;
    __syncthreads();
    int off = 0;
    int tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int s = lds[i];
        if (i < w) off += s;
        tot += s;
    }
    total = tot;
    __syncthreads();
    return off + inc - v;
}

// 32-bit multiplicative hash -> table index
__device__ __forceinline__ u32 hash_u32(u32 k) {
    k ^= k >> 16;
    k *= 0x7feb352du;
    k ^= k >> 15;
    k *= 0x846ca68bu;
    k ^= k >> 16;
    return k;
}
__device__ __forceinline__ u32 hash_u64(u64 k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    return (u32)k ^ (u32)(k >> 32);
}

// ---------------------------------------------------------------------------------------------
// Fill kernel used instead of hipMemsetAsync: memset nodes recorded during stream capture were observed not
// to be re-executed reliably on hipGraph replay (ROCm 7.2: hash tables stayed dirty -> probe loops never
// ended on the 3rd replay); a plain kernel node always replays.  `p` must be 16-byte aligned, bytes % 4 == 0.
static __global__ __launch_bounds__(256) void pcd_fill_kernel(uint4 *p, u32 word, size_t n16, size_t tail_words) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    const uint4 v = make_uint4(word, word, word, word);
    for (size_t e = i; e < n16; e += stride) p[e] = v;
    if (i < tail_words) reinterpret_cast<u32 *>(p + n16)[i] = word;
}
static inline void pcd_fill(void *p, int byte_value, size_t bytes, hipStream_t st) {
    if (bytes == 0 || p == nullptr) return;
    u32 b = (u32)(byte_value & 0xFF);
    u32 word = b | (b << 8) | (b << 16) | (b << 24);
    size_t n16 = bytes / 16, tail = (bytes % 16) / 4;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    pcd_fill_kernel<<<(unsigned)blocks, 256, 0, st>>>((uint4 *)p, word, n16, tail);
}

// ---------------------------------------------------------------------------------------------
// Generic 3-launch exclusive scan of f(i), i in [0,n): out[i] = sum_{j<i} f(j), out[n] = total.
// block_sums needs pcd_div_up(n,256)+1 ints.
template <class F>
__global__ __launch_bounds__(256) void scan_reduce_kernel(F f, int n, int *block_sums) {
    __shared__ int lds[4];
    int i = blockIdx.x * 256 + threadIdx.x;
    int v = (i < n) ? f(i) : 0;
    int total;
    block_exclusive_scan(v, lds, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single block: in-place exclusive scan of block_sums[0..nb), total -> block_sums[nb] (and *total_out)
static __global__ __launch_bounds__(256) void scan_spine_kernel(int *block_sums, int nb, int *total_out) {
    __shared__ int lds[4];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    // 8 consecutive elements per thread and iteration (8 loads in flight; 2048 entries per barrier pair)
    for (int base = 0; base < nb; base += 2048) {
        const int i0 = base + threadIdx.x * 8;
        int v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (i0 + j < nb) ? block_sums[i0 + j] : 0;
        int sum = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int t = v[j];
            v[j] = sum;
            sum += t;
        }
        int total;
        int ex = block_exclusive_scan(sum, lds, total);
        int carry = carry_s;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (i0 + j < nb) block_sums[i0 + j] = carry + ex + v[j];
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        block_sums[nb] = carry_s;
        if (total_out) *total_out = carry_s;
    }
}

template <class F>
__global__ __launch_bounds__(256) void scan_down_kernel(F f, int n, const int *block_sums, int *out) {
    __shared__ int lds[4];
    int i = blockIdx.x * 256 + threadIdx.x;
    int v = (i < n) ? f(i) : 0;
    int total;
    int ex = block_exclusive_scan(v, lds, total);
    int base = block_sums[blockIdx.x];
    if (i < n) out[i] = base + ex;
    if (i == n - 1) out[n] = base + ex + v;
}

template <class F>
static inline int scan_exclusive(F f, int n, int *out, int *block_sums, int *total_out,
                                 hipStream_t st) {
    int nb = pcd_div_up(n, 256);
    if (n <= 0) {
        // out[0] = 0, total = 0
        pcd_fill(out, 0, sizeof(int), st);
        if (total_out) pcd_fill(total_out, 0, sizeof(int), st);
        return PCD_OK;
    }
    scan_reduce_kernel<F><<<nb, 256, 0, st>>>(f, n, block_sums);
    scan_spine_kernel<<<1, 256, 0, st>>>(block_sums, nb, total_out);
    scan_down_kernel<F><<<nb, 256, 0, st>>>(f, n, block_sums, out);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

// Row-wise exclusive scan of a count matrix cnt[rows][cols] (one block per row, 8 consecutive elements per
// thread and iteration so that 8 loads are in flight), totals[row]; optionally also
// totals_out[flip ? rows - 1 - row : row] (spconv's indice_pair_num order).
static __global__ __launch_bounds__(256) void scan_rows_kernel(const int *cnt, int *off, int cols,
                                                        int *totals, int *totals_out = nullptr,
                                                        int flip = 0) {
    __shared__ int lds[4];
    __shared__ int carry_s;
    const int *c = cnt + (size_t)blockIdx.x * cols;
    int *o = off + (size_t)blockIdx.x * cols;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < cols; base += 2048) {
        const int i0 = base + threadIdx.x * 8;
        int v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (i0 + j < cols) ? c[i0 + j] : 0;
        int sum = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int t = v[j];
            v[j] = sum;
            sum += t;
        }
        int total;
        int ex = block_exclusive_scan(sum, lds, total);
        int carry = carry_s;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (i0 + j < cols) o[i0 + j] = carry + ex + v[j];
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (totals) totals[blockIdx.x] = carry_s;
        if (totals_out) totals_out[flip ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x] = carry_s;
    }
}
