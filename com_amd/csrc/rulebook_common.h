// Pieces shared by the rulebook builders (rulebook.hip: hash / flat-bitmap builds; colmap.hip: column-map builds).
#pragma once
#include "common.h"

namespace {

struct ConvGeom {
    int D, H, W;        // input spatial shape
    int Do, Ho, Wo;     // output spatial shape
    int kd, kh, kw;
    int sd, sh, sw;
    int pd, ph, pw;
    int dd, dh, dw;
    int K;
    int order;          // PCD_ROWS_ZYX / PCD_ROWS_YXZ: the linear key that numbers the rows of a bitmap-ranked level
};

__device__ __forceinline__ u32 lin_key(int b, int z, int y, int x, int D, int H, int W) {
    return (((u32)b * D + z) * H + y) * W + x;
}
// key of a bitmap-ranked level in the given row order (pcd_ops.h: PCD_ROWS_*)
__device__ __forceinline__ u32 ord_key(int order, int b, int z, int y, int x, int D, int H, int W) {
    return order == PCD_ROWS_YXZ ? (((u32)b * H + y) * W + x) * D + z : (((u32)b * D + z) * H + y) * W + x;
}

// pairs[k] = {(i, tbl[kr][i])} for i ascending, kr = flip ? K-1-k : k  (tbl is an input-stationary
// view: for SubM the symmetric row of the output-stationary table).
template <int KT>   // KT > 0: compile-time K (unrolled: all table loads of a thread in flight); 0: runtime K
__global__ __launch_bounds__(256) void pairs_fill_kernel(const int32_t *__restrict__ tbl, int n,
                                                         const int32_t *n_dev, int K, int flip,
                                                         const int *__restrict__ wave_off, int nwaves,
                                                         int32_t *__restrict__ pairs) {
    int i = blockIdx.x * 256 + threadIdx.x;
    int wave = i >> 6;
    const int nn = eff_rows(n_dev, n);
    if (KT > 0) {
        int o[KT > 0 ? KT : 1];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            int kr = flip ? KT - 1 - k : k;
            o[k] = (i < nn) ? tbl[(size_t)kr * n + i] : -1;
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            int kr = flip ? KT - 1 - k : k;
            int tot;
            int r = wave_rank(o[k] >= 0, tot);
            if (o[k] >= 0) {
                int pos = wave_off[(size_t)kr * nwaves + wave] + r;
                pairs[((size_t)k * 2 + 0) * n + pos] = i;
                pairs[((size_t)k * 2 + 1) * n + pos] = o[k];
            }
        }
        return;
    }
    for (int k = 0; k < K; ++k) {
        int kr = flip ? K - 1 - k : k;
        int o = (i < nn) ? tbl[(size_t)kr * n + i] : -1;
        int tot;
        int r = wave_rank(o >= 0, tot);
        if (o >= 0) {
            int pos = wave_off[(size_t)kr * nwaves + wave] + r;
            pairs[((size_t)k * 2 + 0) * n + pos] = i;
            pairs[((size_t)k * 2 + 1) * n + pos] = o;
        }
    }
}

static void launch_pairs_fill(const int32_t *tbl, int n, const int32_t *n_dev, int K, int flip,
                              const int *wave_off, int nwaves, int32_t *pairs, hipStream_t st) {
    int nb = pcd_div_up(n, 256);
    if (K == 27)
        pairs_fill_kernel<27><<<nb, 256, 0, st>>>(tbl, n, n_dev, K, flip, wave_off, nwaves, pairs);
    else
        pairs_fill_kernel<0><<<nb, 256, 0, st>>>(tbl, n, n_dev, K, flip, wave_off, nwaves, pairs);
}

// t / s for t >= 0 with the strides that occur (1, 2) as shifts
__device__ __forceinline__ int div_stride(int t, int s) { return s == 1 ? t : (s == 2 ? (t >> 1) : t / s); }

// Output coordinate along one axis reached from input coordinate c through kernel index k, or -1.
__device__ __forceinline__ int axis_out(int c, int p, int d, int s, int k, int n_out) {
    int t = c + p - k * d;
    if (t < 0) return -1;
    int o = div_stride(t, s);
    return (o * s == t && o < n_out) ? o : -1;
}

constexpr int CLS_MAX = 8;

__device__ __forceinline__ int row_class(int4 c, int pd, int ph, int pw, int sd, int sh, int sw) {
    return (((c.y + pd) % sd) * sh + ((c.z + ph) % sh)) * sw + ((c.w + pw) % sw);
}

// per block of 256 input rows: rows per stride-parity class -> blk_cnt[cls][blk]  (see "Parity classes" below)
__device__ __forceinline__ void block_class_counts(int cls, int ncls, int *cnt /* LDS [CLS_MAX] */,
                                                   int *__restrict__ blk_cnt) {
    if (threadIdx.x < CLS_MAX) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int q = 0; q < ncls; ++q) {
        u64 m = __ballot(cls == q);
        if (lane_id() == 0 && m) atomicAdd(&cnt[q], __popcll(m));
    }
    __syncthreads();
    if ((int)threadIdx.x < ncls) blk_cnt[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = cnt[threadIdx.x];
}

__device__ __forceinline__ int block_sum(int v, int *lds /* [4] */) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (lane_id() == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    const int t = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return t;
}

__device__ __forceinline__ void fill_ff(void *p, size_t bytes, size_t tid, size_t nthreads) {   // bytes % 4 == 0
    const size_t n16 = bytes / 16;
    const uint4 v = make_uint4(~0u, ~0u, ~0u, ~0u);
    for (size_t e = tid; e < n16; e += nthreads) reinterpret_cast<uint4 *>(p)[e] = v;
    if (tid < (bytes % 16) / 4) reinterpret_cast<u32 *>(p)[n16 * 4 + tid] = ~0u;
}

// blk_cnt[cls][blk] -> virtual row of the block's first row of that class (exclusive prefix inside the class +
// the tile-aligned class start), vstart[0 .. ncls]; one WAVE per class, run by ONE block of 256 threads
__device__ void class_offsets(int *blk_cnt, int nblk, int ncls, int tile, int *vstart, int *tot /* LDS [CLS_MAX] */,
                              int *start /* LDS [CLS_MAX + 1] */) {
    for (int q = threadIdx.x >> 6; q < ncls; q += 4) {
        int *row = blk_cnt + (size_t)q * nblk;
        int carry = 0;
        // 8 chunks of 64 counts in flight: the loop is a chain of load latencies
        for (int base = 0; base < nblk; base += 8 * 64) {
            int v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 64 + lane_id();
                v[u] = (i < nblk) ? row[i] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 64 + lane_id();
                const int inc = wave_inclusive_scan(v[u]);
                if (i < nblk) row[i] = carry + inc - v[u];
                carry += __shfl(inc, 63);
            }
        }
        if (lane_id() == 0) tot[q] = carry;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int vb = 0;
        for (int c = 0; c < ncls; ++c) {
            start[c] = vb;
            vstart[c] = vb;
            vb = (vb + tot[c] + tile - 1) / tile * tile;
        }
        vstart[ncls] = vb;
    }
    __syncthreads();
    for (int q = threadIdx.x >> 6; q < ncls; q += 4) {
        int *row = blk_cnt + (size_t)q * nblk;
        const int st = start[q];
        for (int i = lane_id(); i < nblk; i += 64) row[i] += st;
    }
}

// Pair-list offsets without a scan launch: the producer adds every wave's count into the sum of its group of 64
// waves (wsuper[k][wave / 64], zeroed beforehand; integer atomics: order-free); pairs_fill_super_kernel rebuilds a
// wave's offset from the groups before it + the waves of its own group before it.
__device__ __forceinline__ void publish_wave_count(int *__restrict__ wave_cnt, int *blk_sum /* LDS [K] */, int k,
                                                   int wave, int nwaves, bool hit) {
    const u64 m = __ballot(hit);
    if (lane_id() == 0 && wave < nwaves) {
        const int c = __popcll(m);
        wave_cnt[(size_t)k * nwaves + wave] = c;
        if (c) atomicAdd(&blk_sum[k], c);   // the block's 4 waves lie in one group: one global atomic per (block, k)
    }
}

// pairs_fill_kernel with the offsets rebuilt from (wave_cnt, wsuper) -- see publish_wave_count; also writes
// pair_num[k] = total of table row kr (block 0).  K <= 343.
template <int KT>
__global__ __launch_bounds__(256) void pairs_fill_super_kernel(const int32_t *__restrict__ tbl, int n,
                                                               const int32_t *n_dev, int K, int flip,
                                                               const int *__restrict__ wave_cnt, int nwaves,
                                                               const int *__restrict__ wsuper, int nws,
                                                               int32_t *__restrict__ pairs,
                                                               int32_t *__restrict__ pair_num, int scanned) {
    // scanned: wsuper holds exclusive prefixes already (a scan launch ran: beyond PAIRS_SUPER_SCAN groups every block
    // adding up the groups in front of it -- K x sb loads -- was most of this kernel's time) and pair_num is written
    __shared__ int off_s[343][4];
    const int w = threadIdx.x >> 6, lane = lane_id();
    const int wave0 = blockIdx.x * 4, sb = wave0 >> 6, m = wave0 & 63;   // m <= 60: the block's 4 waves share a group
    for (int kr = w; kr < K; kr += 4) {
        int acc = 0;
        if (scanned) {
            acc = lane == 0 ? wsuper[(size_t)kr * nws + sb] : 0;
        } else {
            for (int i = lane; i < sb; i += 64) acc += wsuper[(size_t)kr * nws + i];
        }
        const int wi = (sb << 6) + lane;
        const int c = (wi < nwaves && lane < m + 4) ? wave_cnt[(size_t)kr * nwaves + wi] : 0;
        const int inc = wave_inclusive_scan(c);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if (lane >= m && lane < m + 4) off_s[kr][lane - m] = acc + inc - c;
        if (blockIdx.x == 0 && pair_num && !scanned) {
            int t = 0;
            for (int i = lane; i < nws; i += 64) t += wsuper[(size_t)kr * nws + i];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
            if (lane == 0) pair_num[flip ? K - 1 - kr : kr] = t;
        }
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nn = eff_rows(n_dev, n);
    if (KT > 0) {
        int o[KT > 0 ? KT : 1];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            int kr = flip ? KT - 1 - k : k;
            o[k] = (i < nn) ? tbl[(size_t)kr * n + i] : -1;
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            int kr = flip ? KT - 1 - k : k;
            int tot;
            int r = wave_rank(o[k] >= 0, tot);
            if (o[k] >= 0) {
                int pos = off_s[kr][w] + r;
                pairs[((size_t)k * 2 + 0) * n + pos] = i;
                pairs[((size_t)k * 2 + 1) * n + pos] = o[k];
            }
        }
        return;
    }
    for (int k = 0; k < K; ++k) {
        int kr = flip ? K - 1 - k : k;
        int o = (i < nn) ? tbl[(size_t)kr * n + i] : -1;
        int tot;
        int r = wave_rank(o >= 0, tot);
        if (o >= 0) {
            int pos = off_s[kr][w] + r;
            pairs[((size_t)k * 2 + 0) * n + pos] = i;
            pairs[((size_t)k * 2 + 1) * n + pos] = o;
        }
    }
}

constexpr int PAIRS_SUPER_SCAN = 256;
static void launch_pairs_fill_super(const int32_t *tbl, int n, const int32_t *n_dev, int K, int flip,
                                    const int *wave_cnt, int nwaves, const int *wsuper, int nws, int32_t *pairs,
                                    int32_t *pair_num, hipStream_t st) {
    int nb = pcd_div_up(n, 256);
    const int scanned = nws > PAIRS_SUPER_SCAN;
    // (in place: every thread of scan_rows_kernel reads its 8 entries before it writes them)
    if (scanned) scan_rows_kernel<<<K, 256, 0, st>>>(wsuper, const_cast<int *>(wsuper), nws, nullptr, pair_num, flip);
    if (K == 27)
        pairs_fill_super_kernel<27><<<nb, 256, 0, st>>>(tbl, n, n_dev, K, flip, wave_cnt, nwaves, wsuper, nws, pairs,
                                                        pair_num, scanned);
    else
        pairs_fill_super_kernel<0><<<nb, 256, 0, st>>>(tbl, n, n_dev, K, flip, wave_cnt, nwaves, wsuper, nws, pairs,
                                                       pair_num, scanned);
}

static int make_geom(const int *shape, const int *ks, const int *st, const int *pd, const int *dl,
                     ConvGeom &G) {
    for (int d = 0; d < 3; ++d)
        if (shape[d] <= 0 || ks[d] <= 0 || st[d] <= 0 || pd[d] < 0 || dl[d] <= 0)
            return PCD_ERR_INVALID_ARG;
    G.D = shape[0]; G.H = shape[1]; G.W = shape[2];
    G.kd = ks[0]; G.kh = ks[1]; G.kw = ks[2];
    G.sd = st[0]; G.sh = st[1]; G.sw = st[2];
    G.pd = pd[0]; G.ph = pd[1]; G.pw = pd[2];
    G.dd = dl[0]; G.dh = dl[1]; G.dw = dl[2];
    G.K = ks[0] * ks[1] * ks[2];
    int out[3];
    pcd_conv_out_shape(shape, ks, st, pd, dl, out);
    G.Do = out[0]; G.Ho = out[1]; G.Wo = out[2];
    if (G.K > 343) return PCD_ERR_UNSUPPORTED;
    G.order = PCD_ROWS_ZYX;
    return PCD_OK;
}

}  // namespace
