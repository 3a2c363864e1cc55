// Sparse-convolution arithmetic for gfx950 (replaces spconv's indice_conv fwd / bwd; semantics
// SURVEY.md A.5; call sites pcdet/models/backbones_3d/spconv_backbone.py:12-17,38-45).
//
// Forward and data-gradient: OUTPUT-STATIONARY gather-GEMM.  A wave owns MI*16 output rows and all
// output channels; the contraction runs over the flattened (kernel offset, input channel) axis in
// steps of 32 so that one v_mfma_f32_16x16x32_bf16 consumes, per lane, 8 contiguous bf16 (16 B) of
// ONE gathered feature row: the A/B fragments are filled straight from global memory (L2) with
// 16-byte loads -- no LDS round trip, no transposition.  Narrow layers pack several offsets into one
// MFMA step (Cin = 8: 4 offsets / step, Cin = 16: 2).  Operands are swapped (D^T = W^T X^T) so every
// lane ends with 4 consecutive output channels of one row -> 8/16-byte stores.  Steps whose 16-row
// tile has no neighbour at all are skipped (wave-uniform ballot test).  No atomics: bit-reproducible.
// Weights are pre-packed into MFMA fragment order (pcd_pack_weight), one coalesced 1-KiB load per
// fragment per wave, L2-resident (<= 884 KB per layer).
//
// Weight gradient: per kernel offset k a dense  dW_k = X_gathered^T dY_gathered  over the rulebook
// pairs of k (no wasted MACs).  Each wave stages 32 gathered rows of X and dY in its private LDS
// slice (16-byte global loads -> ds_write_b128, padded rows), then builds both MFMA operands with the
// gfx950 transpose read ds_read_b64_tr_b16 (the contraction index = pair index is the slow axis of
// both row-major tiles).  Waves of a workgroup split the pair range, are reduced through LDS in a
// fixed order, written to per-split slabs and summed by a second kernel: deterministic, no atomics.
#include <type_traits>

#include "common.h"
#include "bn_mid.h"
#include "bnred.h"
#include "cls_table.h"

// Wave priority of the kernels on the step's critical chain (forward convs, data gradients, BatchNorm passes): the
// weight-gradient and rulebook kernels that run beside them on other streams keep priority 0, so on a shared SIMD the
// chain's waves issue first.  Any value > 0 measured the same: 3.675 -> 3.635 ms/step (3 alternating runs each).
#define PCD_MAIN_PRIO 3
namespace {

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

static int log2_exact(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return (1 << s) == v ? s : -1;
}
static int pow2_ge8(int c) {
    int p = 8;
    while (p < c) p <<= 1;
    return p;
}

// blocks interleaved in the packed channel order (see the epilogue of the gather-GEMM kernels)
__host__ __device__ constexpr int gg_quad(int nb) { return nb % 4 == 0 ? 4 : (nb % 2 == 0 ? 2 : 1); }

// ---------------------------------------------------------------------------------------------
// the eight elements e0 .. e0 + 7 (e0 a multiple of 8: one lane's 16 bytes of one fragment = eight consecutive contraction
// indices of one kernel offset, cshift >= 3): one 16-byte store; forward packs read 32 contiguous bytes of the weight
__device__ __forceinline__ void pack_eight(const float *__restrict__ w, int K, int cin, int cout, int mode,
                                           int cshift, int NB, size_t e0, unsigned short *out) {
    const int lane = (int)(e0 >> 3) & 63;
    const size_t t = e0 >> 9;
    const int nb = (int)(t % NB);
    const int s = (int)(t / NB);
    const int q0 = s * 32 + (lane >> 4) * 8;
    const int k = q0 >> cshift;
    const int c0 = q0 & ((1 << cshift) - 1);
    const int Q = gg_quad(NB), m = lane & 15;
    const int col = (nb / Q) * 16 * Q + (m >> 2) * 4 * Q + (nb % Q) * 4 + (m & 3);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (k < K) {
        if (mode == 0) {
            if (col < cout) {
                const float *src = w + ((size_t)col * K + k) * cin + c0;
                if (c0 + 8 <= cin && (cin & 3) == 0 && (((uintptr_t)src) & 15u) == 0) {
                    const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (c0 + j < cin) v[j] = src[j];
                }
            }
        } else if (col < cin) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (c0 + j < cout) v[j] = w[((size_t)(c0 + j) * K + k) * cin + col];
        }
    }
    u32 o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (u32)f32_to_bf16_bits(v[2 * j]) | ((u32)f32_to_bf16_bits(v[2 * j + 1]) << 16);
    *reinterpret_cast<uint4 *>(out + e0) = make_uint4(o[0], o[1], o[2], o[3]);
}

constexpr int PACK_PER_BLOCK = 2048;     // elements per 256-thread block (eight per thread)

__global__ __launch_bounds__(256) void pack_weight_kernel(const float *__restrict__ w, int K, int cin,
                                                          int cout, int mode, int cshift, int NB,
                                                          size_t total, unsigned short *out) {
    const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (e >= total) return;
    pack_eight(w, K, cin, cout, mode, cshift, NB, e, out);
}

// One launch for a whole list of weights.  table[i] = {weight ptr, packed ptr, kvol, cin, cout, mode,
// first block of entry i, 0}; every entry owns whole 256-thread blocks, found by a binary search over the
// first-block column.
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const long long *__restrict__ table, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (table[(size_t)mid * 8 + 6] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long *row = table + (size_t)lo * 8;
    const float *w = (const float *)row[0];
    unsigned short *out = (unsigned short *)row[1];
    const int K = (int)row[2], cin = (int)row[3], cout = (int)row[4], mode = (int)row[5];
    int cc = 8;
    while (cc < (mode == 0 ? cin : cout)) cc <<= 1;
    int cshift = 0;
    while ((1 << cshift) < cc) ++cshift;
    const int NB = ((mode == 0 ? cout : cin) + 15) / 16;
    const size_t total = (size_t)((K * cc + 31) / 32) * NB * 512;
    const size_t e = (((size_t)blockIdx.x - (size_t)row[6]) * 256 + threadIdx.x) * 8;
    if (e >= total) return;
    pack_eight(w, K, cin, cout, mode, cshift, NB, e, out);
}

// Tile of workgroup blockIdx.x when the REAL row tiles (ceil(n / rows_per_tile), n known on the device only) are dealt
// to the 8 XCDs in contiguous runs; gridDim.x (a multiple of 8, sized from the capacity) may exceed them.  A bijection
// of [0, gridDim.x): blocks beyond an XCD's share map to the tile ids past the live range (they own no rows; their
// BatchNorm partial row is zeroed like any empty tile's).
__device__ __forceinline__ int xcd_tile(int n, int rows_per_tile) {
    const int nt = (n + rows_per_tile - 1) / rows_per_tile;
    const int tpx = min((nt + 7) >> 3, (int)(gridDim.x >> 3));
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    return j < tpx ? xcd * tpx + j : 8 * tpx + (j - tpx) * 8 + xcd;
}

// Channel order inside the packed weights.  The MFMA leaves lane (g, rl) with rows m = 4g..4g+3 of every 16-channel
// block nb; with the natural order (channel = 16 nb + m) a lane owns 4 consecutive channels per block = 8-byte
// accesses, 32 bytes per output row and instruction.  The packs instead interleave Q = 4 (2, 1) blocks:
//     channel(nb, m) = (nb / Q) * 16Q + (m / 4) * 4Q + (nb % Q) * 4 + (m % 4),   Q = gg_quad(blocks)
// so a lane owns 4Q CONSECUTIVE channels of its row across the Q blocks: 16-byte accesses, and the 4 lanes of a
// row cover 16Q channels = one full 128-byte line at Q = 4 (the vector L1 works per line touched).

// Epilogue of both gather-GEMM kernels: lane (g, rl) of a wave holds, for every mi, the output channels described
// above of row rows[mi] (-1: no row).  (+bias) (+addend), one rounding, 8/16-byte stores, optional BnRed
// reductions (`red`: 4 * 2 * c_out floats of LDS not used by the main loop).
template <int MI, int NBW, bool OUT_BF16>
__device__ __forceinline__ void gg_epilogue(const f32x4 (&acc)[MI][NBW], const int (&rows)[MI], int c_out, int col0,
                                            int g, int rl, int wave, int tile, const float *__restrict__ bias,
                                            const void *__restrict__ addend, void *__restrict__ yv, const BnRed &bn,
                                            float *red, const bool bystander = false) {
    // bystander: a wave that owns no rows (the loader waves of ggw_kernel) only joins the final barrier / reduction
    constexpr int Q = gg_quad(NBW);
    constexpr int U = Q >= 2 ? 2 : 1;        // blocks per access unit: 8 channels = 16 bytes of bf16 (4 at Q = 1)
    constexpr int CH = 4 * U;
    const bool has_bias = bias != nullptr;
    auto ld_f = [](const float *p, float (&v)[CH]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 t = *reinterpret_cast<const float4 *>(p + 4 * u);
            v[4 * u] = t.x; v[4 * u + 1] = t.y; v[4 * u + 2] = t.z; v[4 * u + 3] = t.w;
        }
    };
    auto ld_h = [](const unsigned short *p, float (&v)[CH]) {     // CH bf16 -> float
        u32 w[2 * U];
        if (U == 2) {
            const uint4 t = *reinterpret_cast<const uint4 *>(p);
            w[0] = t.x; w[1] = t.y; w[2 % (2 * U)] = t.z; w[3 % (2 * U)] = t.w;
        } else {
            const uint2 t = *reinterpret_cast<const uint2 *>(p);
            w[0] = t.x; w[1] = t.y;
        }
#pragma unroll
        for (int j = 0; j < 2 * U; ++j) {
            v[2 * j] = __uint_as_float(w[j] << 16);
            v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        }
    };
    if (!bystander)
#pragma unroll
    for (int qd = 0; qd < NBW / Q; ++qd)
#pragma unroll
        for (int un = 0; un < Q / U; ++un) {
            const int nb0 = qd * Q + un * U;                              // first of the unit's U blocks
            const int col = col0 + qd * 16 * Q + g * 4 * Q + un * CH;     // first of the lane's CH channels
            float bs[CH], bq[CH], bv[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) bs[j] = bq[j] = bv[j] = 0.0f;
            if (has_bias) ld_f(bias + col, bv);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = rows[mi];
                if (row < 0) continue;
                float v[CH];
#pragma unroll
                for (int j = 0; j < CH; ++j) v[j] = acc[mi][nb0 + j / 4][j % 4];
                if (has_bias) {
#pragma unroll
                    for (int j = 0; j < CH; ++j) v[j] += bv[j];
                }
                const size_t at = (size_t)row * c_out + col;
                if (OUT_BF16) {
                    if (addend) {  // y = conv + addend (e.g. the residual branch's gradient), rounded once
                        float ad[CH];
                        ld_h((const unsigned short *)addend + at, ad);
#pragma unroll
                        for (int j = 0; j < CH; ++j) v[j] += ad[j];
                    }
                    u32 o[2 * U];
#pragma unroll
                    for (int j = 0; j < 2 * U; ++j)
                        o[j] = (u32)f32_to_bf16_bits(v[2 * j]) | ((u32)f32_to_bf16_bits(v[2 * j + 1]) << 16);
                    unsigned short *y = (unsigned short *)yv + at;
                    if (U == 2)
                        *reinterpret_cast<uint4 *>(y) = make_uint4(o[0], o[1], o[2 % (2 * U)], o[3 % (2 * U)]);
                    else
                        *reinterpret_cast<uint2 *>(y) = make_uint2(o[0], o[1]);
                    if (bn.mode) {
                        float d[CH];
#pragma unroll
                        for (int j = 0; j < 2 * U; ++j) {
                            d[2 * j] = __uint_as_float(o[j] << 16);
                            d[2 * j + 1] = __uint_as_float(o[j] & 0xffff0000u);
                        }
                        if (bn.mode == 1) {
#pragma unroll
                            for (int j = 0; j < CH; ++j) {
                                bs[j] += d[j];
                                bq[j] += d[j] * d[j];
                            }
                        } else {
                            float xv[CH], tv[CH];
                            ld_h(bn.x + at, xv);
                            if (bn.relu) ld_h(bn.y + at, tv);
#pragma unroll
                            for (int j = 0; j < CH; ++j) {
                                const float dz = (bn.relu && !(tv[j] > 0.0f)) ? 0.0f : d[j];
                                bs[j] += dz;
                                bq[j] += dz * xv[j];     // centred below: sum dz*(x - mean) = sum dz*x - mean * sum dz
                            }
                        }
                    }
                } else {
                    float *y = (float *)yv + at;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        float4 o = make_float4(v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3]);
                        if (addend) {
                            const float4 ad = *reinterpret_cast<const float4 *>((const float *)addend + at + 4 * u);
                            o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
                        }
                        *reinterpret_cast<float4 *>(y + 4 * u) = o;
                    }
                }
            }
            if (OUT_BF16 && bn.mode) {
                if (bn.mode == 2) {
                    // the lane's <= MI rows: (sum dz*x - mean*sum dz) * invstd = sum dz*xhat; mean and invstd are
                    // only needed here, not across the row loop (registers)
                    float mu[CH], is[CH];
                    ld_f(bn.mean + col, mu);
                    ld_f(bn.invstd + col, is);
#pragma unroll
                    for (int j = 0; j < CH; ++j) bq[j] = (bq[j] - mu[j] * bs[j]) * is[j];
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    bs[j] = row16_sum(bs[j]);
                    bq[j] = row16_sum(bq[j]);
                }
                if (rl == 0) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        *reinterpret_cast<float4 *>(red + (wave * 2 + 0) * c_out + col + 4 * u) =
                            make_float4(bs[4 * u], bs[4 * u + 1], bs[4 * u + 2], bs[4 * u + 3]);
                        *reinterpret_cast<float4 *>(red + (wave * 2 + 1) * c_out + col + 4 * u) =
                            make_float4(bq[4 * u], bq[4 * u + 1], bq[4 * u + 2], bq[4 * u + 3]);
                    }
                }
            }
        }
    if (OUT_BF16 && bn.mode) {
        __syncthreads();
        bnred_publish(bn, tile, c_out, [&](int e) {
            return ((red[e] + red[2 * c_out + e]) + red[4 * c_out + e]) + red[6 * c_out + e];
        });
    }
}

// ---------------------------------------------------------------------------------------------
// Output-stationary gather-GEMM.  Workgroup = 4 waves x (MI*16) output rows, all NB*16 output channels.
//   * the workgroup's rulebook tile nbr[K][ROWS] is staged in LDS once (coalesced k-major reads) plus one
//     row of -1 that padded contraction steps read;
//   * packed weights are shared by the 4 waves through LDS: narrow layers keep the whole packed weight
//     resident (no barrier in the main loop); wide layers stream it through a double-buffered stage of
//     SG contraction steps, filled global -> registers -> LDS one stage ahead (one barrier per stage).
//     Fragments come back with conflict-free linear ds_read_b128; with MI = 2 each feeds two MFMAs;
//   * gathered feature rows go straight from L2 into MFMA operand registers, issued one GROUP of G
//     contraction steps ahead of their use (two register sets, ping-pong, no copies);
//   * a contraction step whose 16-row tile has no neighbour at all is skipped (wave-uniform ballot).
// EVERY load in the main loop is an unconditional raw buffer load (out-of-range -> zeros, no memory access):
// with conditional loads hipcc cannot count how many younger loads follow an operand's load and falls back to
// s_waitcnt vmcnt(~0) at the first MFMA of each group, which waits for the loads issued for the NEXT group and
// exposes the full L2 latency every G steps (seen in the ISA of the previous version).
// WN = 2: 8 waves per workgroup, waves 4-7 take the upper half of the output channels of the same rows.
// Waves per SIMD the main loop's registers allow (accumulators + two gather register sets + ~24): the epilogue
// (BnRed) must not push the allocation past that -- these kernels live on occupancy.
constexpr int gg_waves(int nbw, int mi, int g) {
    const int est = mi * nbw * 4 + 2 * g * mi * 4 + 32 + (mi == 1 ? 16 : (mi == 2 && g == 1 ? 8 : 0)) +
                    (g >= 2 && mi >= 2 ? 16 : 0);
    const int w = 512 / est;
    return w > 7 ? 7 : (w < 1 ? 1 : w);   // (LDS limits the narrow kernels to <= 7 waves per SIMD anyway)
}

template <int NB, int MI, int G, int SG, bool OUT_BF16, int WN = 1>   // SG == 0: weights resident in LDS
__global__ __launch_bounds__(256 * WN, gg_waves(NB / WN, MI, G)) void gather_gemm_kernel(
    const unsigned short *__restrict__ x, int c_in, int cshift, const uint4 *__restrict__ wp,
    const float *__restrict__ bias, const int32_t *__restrict__ nbr, int nbr_stride, int K, int flip,
    int n_out_cap, const int32_t *__restrict__ n_out_dev, void *__restrict__ yv, int nsteps,
    unsigned x_bytes, int dbg, const void *__restrict__ addend, BnRed bn) {
    __builtin_amdgcn_s_setprio(PCD_MAIN_PRIO);   // main-chain kernel: issue ahead of the weight-gradient waves sharing the SIMD
    constexpr int ROWS = 4 * MI * 16;
    constexpr int THREADS = 256 * WN;
    constexpr int NBW = NB / WN;                           // 16-channel blocks per wave
    static_assert(NB % WN == 0, "channel split");
    static_assert(WN == 1 || gg_quad(NB) == gg_quad(NB / WN), "interleave groups must not straddle the channel split");
    constexpr bool STAGED = SG > 0;
    constexpr int GPS = STAGED ? SG / G : 1;               // gather groups per weight stage: 1 or 2
    static_assert(!STAGED || (SG == G || SG == 2 * G), "a stage is one or two gather groups");
    constexpr int VEC = STAGED ? SG * NB * 64 : 1;         // uint4 per stage
    constexpr int WPT = STAGED ? VEC / THREADS : 1;        // uint4 per thread per stage
    static_assert(!STAGED || VEC % THREADS == 0, "stage size");
    const int n_out = eff_rows(n_out_dev, n_out_cap);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned wtotal = (unsigned)nsteps * NB * 64;  // uint4 in the packed weight
    uint4 *wbuf = (uint4 *)smem;                         // staged: [2][VEC]; resident: [wtotal]
    int *nbr_s = (int *)(smem + (STAGED ? (size_t)2 * VEC : (size_t)wtotal) * sizeof(uint4));  // [K+1][ROWS]
    float *red_s = (float *)(nbr_s + (K + 1) * ROWS);      // [4][2][c_out], only with bn.mode

    const int wave = (threadIdx.x >> 6) & 3;               // row-tile index inside the workgroup
    const int wn = threadIdx.x >> 8;                       // channel half (0 when WN == 1)
    const int lane = threadIdx.x & 63;
    const int rl = lane & 15;
    const int g = lane >> 4;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order, speed only): give each
    // XCD a CONTIGUOUS run of row tiles so its private L2 holds 1/8 of the feature matrix (+ halo) instead
    // of every XCD streaming all of it.  gridDim.x is a multiple of 8; surplus tiles exit.
    const int tile = xcd_tile(n_out, ROWS);                // contiguous runs of the REAL tiles per XCD
    const int r0wg = tile * ROWS;
    constexpr int c_out = NB * 16;
    if (r0wg >= n_out) {
        if (bn.mode) bnred_zero_row(bn, tile, c_out);
        return;
    }

    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, (int)(wtotal * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)nbr, 0, (int)((unsigned)K * (unsigned)nbr_stride * 4u), 0x00020000);
    // rulebook tile -> LDS.  With a row stride that is a multiple of 4 (always in static-shape mode, where the
    // capacities are rounded) a lane fetches 4 consecutive rows of one offset with ONE 16-byte load: 4x fewer
    // load instructions in a prologue that is pure latency (27 dword loads per thread otherwise).
    const bool packed = (flip & 2) != 0;
    flip &= 1;
    if (packed) {
        // strided rulebook in its packed form (pcd_rulebook_conv_cm_build_compact): [K / 3][nbr_stride] words, per (ky, kx)
        // { first present row : 29, presence of kz = 0, 1, 2 : 3 } -- expanded into the same [K + 1][ROWS] tile
        const int KQ = K / 3;
        const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)nbr, 0, (int)((unsigned)KQ * (unsigned)nbr_stride * 4u), 0x00020000);
        const int total = KQ * ROWS;
        for (int base = threadIdx.x; base < total; base += 4 * THREADS) {
            unsigned v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * THREADS;
                const int q = idx / ROWS, r = idx - q * ROWS;
                const int row = r0wg + r;
                const unsigned off = (idx < total && row < n_out) ? ((unsigned)q * (unsigned)nbr_stride + (unsigned)row) * 4u
                                                                  : 0xFFFFFFF0u;
                v[u] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(prsrc, off, 0, 0);      // beyond the table: 0 = no neighbour
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * THREADS;
                if (idx < total) {
                    const int q = idx / ROWS, r = idx - q * ROWS;
                    const unsigned m = v[u] >> 29;
                    const int first = (int)(v[u] & 0x1FFFFFFFu);
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                        nbr_s[(a * KQ + q) * ROWS + r] = ((m >> a) & 1u) ? first + __builtin_popcount(m & ((1u << a) - 1u)) : -1;
                }
            }
        }
        for (int r = threadIdx.x; r < ROWS; r += THREADS) nbr_s[K * ROWS + r] = -1;   // row K: padded steps
    } else if ((nbr_stride & 3) == 0) {
        constexpr int QR = ROWS / 4;                       // 16-byte pieces per offset
        const int total = K * QR;
        for (int base = threadIdx.x; base < total; base += 4 * THREADS) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * THREADS;
                const int k = idx / QR, r = (idx - k * QR) * 4;
                const int krow = flip ? (K - 1 - k) : k;
                const unsigned off = idx < total ? ((unsigned)krow * (unsigned)nbr_stride + (unsigned)(r0wg + r)) * 4u
                                                 : 0xFFFFFFF0u;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(nrsrc, off, 0, 0);   // beyond the table: zeros
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * THREADS;
                if (idx < total) {
                    const int k = idx / QR, r = (idx - k * QR) * 4;
                    int4 w = make_int4((int)v[u][0], (int)v[u][1], (int)v[u][2], (int)v[u][3]);
                    const int row = r0wg + r;       // rows at / beyond n_out have no neighbours
                    if (row + 0 >= n_out) w.x = -1;
                    if (row + 1 >= n_out) w.y = -1;
                    if (row + 2 >= n_out) w.z = -1;
                    if (row + 3 >= n_out) w.w = -1;
                    *reinterpret_cast<int4 *>(nbr_s + k * ROWS + r) = w;
                }
            }
        }
        for (int r = threadIdx.x; r < ROWS; r += THREADS) nbr_s[K * ROWS + r] = -1;   // row K: padded steps
    } else {
        const int total = K * ROWS;
        for (int base = threadIdx.x; base < total + ROWS; base += 4 * THREADS) {
            int v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * THREADS;
                const int k = idx / ROWS, r = idx - k * ROWS;
                const int row = r0wg + r;
                const int krow = flip ? (K - 1 - k) : k;
                const unsigned off = (idx < total && row < n_out)
                                         ? ((unsigned)krow * (unsigned)nbr_stride + (unsigned)row) * 4u
                                         : 0xFFFFFFF0u;
                v[u] = __builtin_amdgcn_raw_buffer_load_b32(nrsrc, off, 0, 0);
                if (!(idx < total && row < n_out)) v[u] = -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * THREADS;
                if (idx < total + ROWS) nbr_s[idx] = v[u];  // row K of the tile = -1 (padded steps)
            }
        }
    }
    if (STAGED) {
        u32x4 w0[WPT];
#pragma unroll
        for (int j = 0; j < WPT; ++j)
            w0[j] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (unsigned)(j * THREADS + threadIdx.x) * 16u, 0, 0);
#pragma unroll
        for (int j = 0; j < WPT; ++j) reinterpret_cast<u32x4 *>(wbuf)[j * THREADS + threadIdx.x] = w0[j];
    } else {
        for (unsigned e = threadIdx.x; e < wtotal; e += 4 * THREADS) {
            u32x4 w0[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                w0[u] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (e + u * THREADS) * 16u, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e + u * THREADS < wtotal) reinterpret_cast<u32x4 *>(wbuf)[e + u * THREADS] = w0[u];
        }
    }
    __syncthreads();

    f32x4 acc[MI][NBW];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[mi][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int tile_row = wave * (MI * 16) + rl;
    // Feature rows are fetched with raw buffer loads: a missing neighbour (index -1) becomes an offset beyond
    // num_records, which the hardware answers with zeros WITHOUT a memory access, a branch or an exec-mask
    // dance, and the address is one 32-bit shift-add instead of a 64-bit multiply-add.
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    const int row_shift = cshift + 1;  // log2(bytes per feature row)
    u32x4 a0[G][MI], a1[G][MI];
    bool v0[G], v1[G];

    auto gather_group = [&](int s0, u32x4(&a)[G][MI], bool(&valid)[G]) {
#pragma unroll
        for (int gg = 0; gg < G; ++gg) {
            const int q0 = (s0 + gg) * 32 + g * 8;
            int k = q0 >> cshift;
            k = k < K ? k : K;                                  // padded steps read the -1 row
            const unsigned c0b = (unsigned)(q0 & (c_in - 1)) * 2u;
            valid[gg] = false;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                int i = nbr_s[k * ROWS + tile_row + mi * 16];
                if (dbg & 1) i = (i >= 0) ? (tile_row & 63) : i;  // ablation: gathers hit a few hot rows
                a[gg][mi] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ((unsigned)i << row_shift) + c0b, 0, 0);
                valid[gg] |= (i >= 0);
            }
        }
    };

    u32x4 wreg[WPT];
    int cur = 0;
    // one group of G contraction steps.  FIRST / LAST: position of the group inside its weight stage
    // (compile time), so the stage prefetch / publish code is straight-line.
    auto body = [&](int grp, auto first_tag, auto last_tag, u32x4(&ac)[G][MI], bool(&vc)[G], u32x4(&an)[G][MI],
                    bool(&vn)[G]) {
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
        const int s0 = grp * G;
        if (STAGED && FIRST && !(dbg & 2)) {
            const unsigned base = (unsigned)(s0 / (STAGED ? SG : 1) + 1) * VEC;  // next stage (OOB -> zeros)
#pragma unroll
            for (int j = 0; j < WPT; ++j)
                wreg[j] = __builtin_amdgcn_raw_buffer_load_b128(
                    wrsrc, (base + (unsigned)(j * THREADS) + threadIdx.x) * 16u, 0, 0);
        }
        gather_group(s0 + G, an, vn);  // beyond the last step: index -1 everywhere -> no memory traffic
        const uint4 *wcur = STAGED ? wbuf + (size_t)cur * VEC + (size_t)(FIRST ? 0 : G) * NB * 64
                                   : wbuf + (size_t)s0 * NB * 64;
#pragma unroll
        for (int gg = 0; gg < G; ++gg) {
            if (__any(vc[gg]) && !(dbg & 4)) {
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    bf16x8 b = as_bf16x8(wcur[(gg * NB + wn * NBW + nb) * 64 + lane]);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            b, __builtin_bit_cast(bf16x8, ac[gg][mi]), acc[mi][nb], 0, 0, 0);
                }
            }
        }
        if (STAGED && LAST) {
            u32x4 *wnext = reinterpret_cast<u32x4 *>(wbuf) + (size_t)(cur ^ 1) * VEC;
#pragma unroll
            for (int j = 0; j < WPT; ++j) wnext[j * THREADS + threadIdx.x] = wreg[j];
            __syncthreads();
            cur ^= 1;
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;

    gather_group(0, a0, v0);
    const int ngroups = (nsteps + G - 1) / G;
    const int npairs = (dbg & 8) ? 0 : (ngroups + 1) / 2;  // groups are processed in ping-pong pairs
    for (int pr = 0; pr < npairs; ++pr) {
        if (GPS == 2) {  // one stage = this pair
            body(2 * pr, T_{}, F_{}, a0, v0, a1, v1);
            body(2 * pr + 1, F_{}, T_{}, a1, v1, a0, v0);
        } else {         // every group is a stage (or the weights are resident)
            body(2 * pr, T_{}, T_{}, a0, v0, a1, v1);
            body(2 * pr + 1, T_{}, T_{}, a1, v1, a0, v0);
        }
    }

    int rows[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = r0wg + tile_row + mi * 16;
        rows[mi] = row < n_out ? row : -1;
    }
    BnRed bne = bn;
    if (WN != 1) bne.mode = 0;   // the reductions assume 4 waves x all channels (every launch configuration in use)
    gg_epilogue<MI, NBW, OUT_BF16>(acc, rows, c_out, wn * NBW * 16, g, rl, wave, tile, bias, addend, yv, bne, red_s);
}

template <int NB, int MI, int G, int SG, int WN = 1>
static int launch_gg(const void *x, int c_in, int cshift, const void *wp, const float *bias,
                     const int32_t *nbr, int nbr_stride, int K, int flip, int n_out, const int32_t *n_out_dev,
                     void *y, int y_dtype, int nsteps, unsigned x_bytes, hipStream_t st, const void *addend,
                     const PcdBnReduce *bnr, int *tiles_only) {
    constexpr int ROWS = 4 * MI * 16;
    int grid = pcd_div_up(pcd_div_up(n_out, ROWS), 8) * 8;
    if (tiles_only) {
        *tiles_only = grid;
        return PCD_OK;
    }
    BnRed bn;
    if (int rc = make_bnred(bnr, y_dtype, NB * 16, grid, &bn)) return rc;
    if (bn.mode && WN != 1) return PCD_ERR_UNSUPPORTED;
    size_t wbytes = SG > 0 ? (size_t)2 * SG * NB * 64 * sizeof(uint4) : (size_t)nsteps * NB * 64 * sizeof(uint4);
    size_t lds = wbytes + (size_t)(K + 1) * ROWS * sizeof(int) + (bn.mode ? (size_t)8 * NB * 16 * sizeof(float) : 0);
    if (lds > 160 * 1024) return PCD_ERR_UNSUPPORTED;
    const int dbg = pcd_opt(PCD_OPT_GG_DBG);   // ablation switches (0 in production)
    auto kb = gather_gemm_kernel<NB, MI, G, SG, true, WN>;
    auto kf = gather_gemm_kernel<NB, MI, G, SG, false, WN>;
    if (lds > 64 * 1024) {
        // above the default dynamic-LDS limit: raise it for this kernel (host-side attribute, set once per
        // instantiation -- benign race: the call is idempotent)
        static size_t raised[2] = {0, 0};
        const int which = y_dtype == PCD_BF16 ? 0 : 1;
        if (raised[which] < lds) {
            if (hipFuncSetAttribute((const void *)(which == 0 ? kb : kf),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return PCD_ERR_LAUNCH;
            raised[which] = lds;
        }
    }
    if (y_dtype == PCD_BF16)
        kb<<<grid, 256 * WN, lds, st>>>((const unsigned short *)x, c_in, cshift, (const uint4 *)wp, bias, nbr,
                                   nbr_stride, K, flip, n_out, n_out_dev, y, nsteps, x_bytes, dbg, addend, bn);
    else
        kf<<<grid, 256 * WN, lds, st>>>((const unsigned short *)x, c_in, cshift, (const uint4 *)wp, bias, nbr,
                                   nbr_stride, K, flip, n_out, n_out_dev, y, nsteps, x_bytes, dbg, addend, bn);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

// ---------------------------------------------------------------------------------------------
// WIDE gather-GEMM (C_in = 64 / 128): every operand reaches LDS by LDS-DMA (buffer_load_dwordx4 ... lds), nothing
// is staged through VGPRs.
//   * Gathered rows are fetched as WHOLE 128-byte lines: one wave instruction moves 8 rows x 128 B (64 input
//     channels = 2 contraction steps) instead of the 16 rows x 64 B a fragment-shaped load touches -- half the
//     lines per contraction step through the texture-address path, which is what bounds the fragment-loading
//     kernel above (MI355X guide: "x operand through LDS in full 128-B lines, filled by glds").  The LDS image of a
//     DMA is lane-linear, so the XOR swizzle that makes the ds_read_b128 operand reads conflict-free is applied on
//     the SOURCE side: lane l of instruction j fetches piece (l & 7) ^ ((row >> 1) & 7) of row 8j + (l >> 3), and
//     the reader of (row, piece) looks at piece ^ ((row >> 1) & 7).
//   * Packed weights (already in MFMA fragment order = lane-linear) stream through a ring of R stages of 2
//     contraction steps, each wave issuing a quarter of a stage; gathered rows through a per-wave ring of the same
//     depth.  A stage is issued R - 1 iterations before it is used; the only waits are one counted
//     s_waitcnt vmcnt((R - 2) * (GI + WPW)) and one raw s_barrier per stage (never vmcnt(0) at R = 3, never a
//     __syncthreads(): its fence would drain the DMAs in flight).
//   * Missing neighbours (index -1) are DMA'd from beyond num_records: the buffer unit returns zeros to LDS
//     without touching memory; 16-row tiles without any neighbour at an offset skip their MFMAs (wave-uniform mask
//     from a ballot taken when the stage was issued).
// Same arithmetic, same summation order per output element as gather_gemm_kernel (steps ascending, fp32
// accumulate in the MFMA) -> bit-identical results; same epilogue.
// (a __device__ function, not a lambda: inside a lambda of a __global__ template the target-feature check of the
// builtin is deferred on the HOST side and silently drops the kernel's host stub)
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char *lds_dst_wave_uniform, unsigned voffset) {
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_dst_wave_uniform, 16, voffset, 0, 0, 0);
}

// CW = 2 (round 5): TWO consumer waves per SIMD, each with half of the output channels of the SIMD's MI x 16 rows -- while one
// waits for its operand fragments the other issues MFMAs (one consumer per SIMD read all fragments of a stage, then ran its
// MFMAs: ~1100 clk per stage for 768 clk of matrix work).  Every output element is still accumulated in the same order: the
// results stay bit-identical.  12 waves: consumers 0-7 (row group w & 3, channel half w >> 2), loaders 8-11.
template <int NB, int SOFF, int MI, int R, bool OUT_BF16, int CW = 1>   // SOFF = stages per offset = C_in / 64
__global__ __launch_bounds__(256 * (1 + CW), 1) void ggw_kernel(
    const unsigned short *__restrict__ x, const uint4 *__restrict__ wp, const float *__restrict__ bias,
    const int32_t *__restrict__ nbr, int nbr_stride, int K, int flip, int n_out_cap,
    const int32_t *__restrict__ n_out_dev, void *__restrict__ yv, unsigned x_bytes, unsigned w_bytes,
    const void *__restrict__ addend, BnRed bn, int dbg) {
    __builtin_amdgcn_s_setprio(PCD_MAIN_PRIO);   // main-chain kernel: issue ahead of the weight-gradient waves sharing the SIMD
    // 8 waves: waves 0-3 are CONSUMERS (MI x 16 output rows each: LDS operand reads + MFMAs, nothing else), waves
    // 4-7 are LOADERS (wave 4 + w feeds consumer w: neighbour indices -> DMA offsets, all LDS-DMA instructions).
    // A wave issues in order, and a VMEM instruction waits in the issue stage while the CU's one texture-address
    // unit is busy (~32 clk per 1-KiB instruction): with loads and MFMAs in ONE wave per SIMD the MFMA pipe idles
    // during every such wait (measured: loads alone 32 us, MFMAs alone 31 us, both in one wave 57-63 us whether
    // issued back to back or interleaved).  Loader wave 4 + w shares SIMD w with consumer w.
    constexpr int ROWS = 64 * MI;
    constexpr int ROWB = 128 * SOFF;                       // bytes per feature row
    constexpr int GI = 2 * MI;                             // DMA instructions per loader per gather stage
    constexpr int WFR = 2 * NB;                            // 1-KiB weight fragments per stage
    constexpr int WPW = WFR / 4;                           // ... per loader
    constexpr int A_STAGE = MI * 2048;                     // bytes per consumer per stage (MI*16 rows x 128 B)
    constexpr int W_STAGE = WFR * 1024;
    constexpr int c_out = NB * 16;
    constexpr int THREADS = 256 * (1 + CW);
    constexpr int NBH = NB / CW;                           // output-channel blocks per consumer
    static_assert(CW == 1 || (CW == 2 && NB % 2 == 0 && gg_quad(NB) == gg_quad(NB / 2)), "channel split");
    static_assert(R == 2 || R == 3, "ring depth");
    static_assert((R - 2) * (GI + WPW) < 64, "vmcnt field");
    const int n_out = eff_rows(n_out_dev, n_out_cap);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *wring = smem;                                    // [R][W_STAGE]
    char *aring = smem + R * W_STAGE;                      // [4 consumers][R][A_STAGE]
    int *nbr_s = (int *)(aring + 4 * R * A_STAGE);         // [K + 1][ROWS]
    float *red_s = (float *)(nbr_s + (K + 1) * ROWS);      // [4][2][c_out], only with bn.mode

    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = wave8 >= 4 * CW;
    const int wave = wave8 & 3;                            // row group (of the loader: the one it feeds)
    const int half = CW == 2 ? (wave8 >> 2) & 1 : 0;       // consumer: its half of the output channels
    const int lane = threadIdx.x & 63;
    const int rl = lane & 15, g = lane >> 4;
    // XCD-aware tile order (workgroup b runs on XCD b % 8): every XCD gets a CONTIGUOUS run of row tiles -- of the
    // REAL tiles.  The grid is sized from the capacity (1.25 x the row count in static-shape mode); splitting the
    // grid instead of the real tiles would leave the last XCDs idle while the first ones run a second round.
    const int tile = xcd_tile(n_out, ROWS);
    const int r0wg = tile * ROWS;
    if (r0wg >= n_out) {
        if (bn.mode) bnred_zero_row(bn, tile, c_out);
        return;
    }
    const __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)nbr, 0, (int)((unsigned)K * (unsigned)nbr_stride * 4u), 0x00020000);
    {   // rulebook tile -> LDS (k-major, coalesced), row K = -1 for the stages beyond the last offset
        const int total = K * ROWS;
        for (int base = threadIdx.x; base < total + ROWS; base += 4 * THREADS) {
            int v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * THREADS;
                const int k = idx / ROWS, r = idx - k * ROWS;
                const int row = r0wg + r;
                const int krow = flip ? (K - 1 - k) : k;
                const bool ok = idx < total && row < n_out;
                const unsigned off = (ok && !(dbg & 16)) ? ((unsigned)krow * (unsigned)nbr_stride + (unsigned)row) * 4u : 0xFFFFFFF0u;
                v[u] = __builtin_amdgcn_raw_buffer_load_b32(nrsrc, off, 0, 0);
                if (!ok) v[u] = -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * THREADS;
                if (idx < total + ROWS) nbr_s[idx] = v[u];
            }
        }
    }
    __syncthreads();      // (no DMA in flight yet: a plain barrier)

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, (int)w_bytes, 0x00020000);
    char *const awave = aring + wave * (R * A_STAGE);
    const int wrow0 = wave * (MI * 16);
    const int T = (dbg & 8) ? 0 : K * SOFF;                // stages (ablation 8: prologue + epilogue only)
    const int NIT = (T + R - 1) / R * R;                   // iterations (both roles run the same number of barriers)

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    // validity of (stage, consumer): any of the consumer's rows has a neighbour at the stage's offset.  Both
    // roles derive it from the rulebook tile in LDS (the consumer to skip the stage's MFMAs).
    auto stage_k = [&](int stage) {
        const int k = SOFF == 2 ? (stage >> 1) : stage;
        return k < K ? k : K;                              // beyond the last offset: the all -1 row
    };

    if (loader) {
        // ------------------------------------------------------------------------------------------- loader
        // the loader's few instructions go ahead of the consumer's MFMA stream on the shared SIMD
        if (!(dbg & 64)) __builtin_amdgcn_s_setprio(3);
        const int gl_row = lane >> 3;                      // row slot of DMA instruction j: 8 j + (lane >> 3)
        // source piece = (l & 7) ^ ((row_slot >> 1) & 7) = (l & 7) ^ ((4 j + (l >> 4)) & 7)
        const unsigned gl_p0 = (unsigned)(lane & 7), gl_p1 = (unsigned)(lane >> 4);
        auto fire = [&](int stage, auto slot_tag) {
            constexpr int SLOT = decltype(slot_tag)::value;
            if (dbg & 32) return;                              // ablation: no DMA instructions at all
            const unsigned h = SOFF == 2 ? (unsigned)(stage & 1) * 128u : 0u;
            const int *irow = nbr_s + stage_k(stage) * ROWS + wrow0 + gl_row;
            int idx[GI];
#pragma unroll
            for (int j = 0; j < GI; ++j) idx[j] = irow[8 * j];
#pragma unroll
            for (int j = 0; j < GI; ++j) {
                const unsigned piece = gl_p0 ^ ((4u * j + gl_p1) & 7u);
                unsigned off = (unsigned)idx[j] * (unsigned)ROWB + h + piece * 16u;   // idx = -1 -> beyond num_records
                if (dbg & 1) off = 0xFFFFFF00u;                                     // ablation: no gather traffic
                glds16(xrsrc, awave + SLOT * A_STAGE + j * 1024, off);
            }
#pragma unroll
            for (int f = 0; f < WPW; ++f) {
                const int frag = wave + 4 * f;
                const unsigned off = (stage < T && !(dbg & 2))
                                         ? (unsigned)stage * (unsigned)W_STAGE + (unsigned)lane * 16u + (unsigned)frag * 1024u
                                         : 0xFFFFFFF0u;
                glds16(wrsrc, wring + SLOT * W_STAGE + frag * 1024, off);
            }
        };
        // iteration t: stage t must have landed (all but the (R - 2) younger stages' instructions retired), then
        // the barrier publishes it and frees slot (t - 1) % R for stage t + R - 1
#define GGW_LOADER_STEP(FIRE_STAGE, SLOT_TAG)                                                             \
    do {                                                                                                   \
        if (R == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * (GI + WPW)) : "memory");            \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                              \
        __builtin_amdgcn_s_barrier();                                                                      \
        fire(FIRE_STAGE, SLOT_TAG);                                                                        \
    } while (0)
        if (R == 3) {
            fire(0, S0{});
            fire(1, S1{});
            for (int t = 0; t < NIT; t += 3) {
                GGW_LOADER_STEP(t + 2, S2{});
                GGW_LOADER_STEP(t + 3, S0{});
                GGW_LOADER_STEP(t + 4, S1{});
            }
        } else {
            fire(0, S0{});
            for (int t = 0; t < NIT; t += 2) {
                GGW_LOADER_STEP(t + 1, S1{});
                GGW_LOADER_STEP(t + 2, S0{});
            }
        }
#undef GGW_LOADER_STEP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stages beyond T: zeros from beyond num_records
        // (the SAME instantiation as the consumers': bnred_publish keeps a __shared__ flag per instantiation -- with
        //  <MI, NB> here and <MI, NB / 2> there the two halves of the workgroup read different flags)
        f32x4 none[MI][NBH];
        int norows[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) norows[mi] = -1;
        gg_epilogue<MI, NBH, OUT_BF16>(none, norows, c_out, 0, g, rl, wave, tile, bias, addend, yv, bn, red_s, true);
        return;
    }

    // ----------------------------------------------------------------------------------------------- consumer
    f32x4 acc[MI][NBH];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nb = 0; nb < NBH; ++nb) acc[mi][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned a_lane = (unsigned)rl * 128u;           // + mi * 2048 + (piece ^ swz) * 16
    const unsigned a_swz = (unsigned)((rl >> 1) & 7);
    // does any of this consumer's rows have a neighbour at the stage's offset?  (lane l checks row l of the wave's
    // MI*16 <= 64 rows; read one iteration ahead so that the LDS latency hides behind MFMAs)
    auto valid_of = [&](int stage) -> bool {
        const int i = lane < MI * 16 ? nbr_s[stage_k(stage) * ROWS + wrow0 + lane] : -1;
        return __builtin_amdgcn_ballot_w64(i >= 0) != 0ull;
    };
    auto compute = [&](auto slot_tag, bool valid) {
        constexpr int SLOT = decltype(slot_tag)::value;
        if (!valid || (dbg & 4)) return;
        const char *ab = awave + SLOT * A_STAGE;
        const char *wb = wring + SLOT * W_STAGE;
        // all operand fragments of the stage (2 contraction steps) are requested up front, then consumed in order:
        // the LDS returns in issue order, so the MFMAs of step 0 start when its fragments have landed while those
        // of step 1 are still in flight (the compiler's own schedule kept ~3 reads ahead of the MFMAs: with one
        // consumer wave per SIMD that exposed an LDS latency every few MFMAs)
        bf16x8 xa[2][MI], bw[2][NBH];
#pragma unroll
        for (int cs = 0; cs < 2; ++cs) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                xa[cs][mi] = *reinterpret_cast<const bf16x8 *>(ab + mi * 2048 + a_lane + (((unsigned)(cs * 4 + g) ^ a_swz) << 4));
#pragma unroll
            for (int nb = 0; nb < NBH; ++nb)
                bw[cs][nb] = *reinterpret_cast<const bf16x8 *>(wb + ((cs * NB + half * NBH + nb) * 64 + lane) * 16);
        }
#pragma unroll
        for (int cs = 0; cs < 2; ++cs)
#pragma unroll
            for (int nb = 0; nb < NBH; ++nb)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)   // (a 16-row tile without neighbours holds zeros: no branch per MFMA)
                    acc[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[cs][nb], xa[cs][mi], acc[mi][nb], 0, 0, 0);
    };
#define GGW_CONSUMER_STEP(SLOT_TAG, STAGE)                                                                 \
    do {                                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* this wave's reads of slot (t - 1) % R are done */ \
        __builtin_amdgcn_s_barrier();                                                                      \
        const bool vnext = valid_of((STAGE) + 1);                                                          \
        compute(SLOT_TAG, vcur && (STAGE) < T);                                                            \
        vcur = vnext;                                                                                      \
    } while (0)
    bool vcur = valid_of(0);
    if (R == 3) {
        for (int t = 0; t < NIT; t += 3) {
            GGW_CONSUMER_STEP(S0{}, t);
            GGW_CONSUMER_STEP(S1{}, t + 1);
            GGW_CONSUMER_STEP(S2{}, t + 2);
        }
    } else {
        for (int t = 0; t < NIT; t += 2) {
            GGW_CONSUMER_STEP(S0{}, t);
            GGW_CONSUMER_STEP(S1{}, t + 1);
        }
    }
#undef GGW_CONSUMER_STEP

    int rows[MI];
    const int tile_row = wrow0 + rl;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = r0wg + tile_row + mi * 16;
        rows[mi] = row < n_out ? row : -1;
    }
    gg_epilogue<MI, NBH, OUT_BF16>(acc, rows, c_out, half * NBH * 16, g, rl, wave, tile, bias, addend, yv, bn, red_s);
}

template <int NB, int SOFF, int MI, int R, int CW = 1>
static int launch_ggw(const void *x, const void *wp, const float *bias, const int32_t *nbr, int nbr_stride, int K,
                      int flip, int n_out, const int32_t *n_out_dev, void *y, int y_dtype, unsigned x_bytes,
                      unsigned w_bytes, hipStream_t st, const void *addend, const PcdBnReduce *bnr, int *tiles_only) {
    constexpr int ROWS = 64 * MI;
    int grid = pcd_div_up(pcd_div_up(n_out, ROWS), 8) * 8;
    if (tiles_only) {
        *tiles_only = grid;
        return PCD_OK;
    }
    BnRed bn;
    if (int rc = make_bnred(bnr, y_dtype, NB * 16, grid, &bn)) return rc;
    const size_t lds = (size_t)R * (2 * NB * 1024) + (size_t)4 * R * (MI * 2048) + (size_t)(K + 1) * ROWS * sizeof(int) +
                       (bn.mode ? (size_t)8 * NB * 16 * sizeof(float) : 0);
    if (lds > 160 * 1024) return PCD_ERR_UNSUPPORTED;
    const int dbg = pcd_opt(PCD_OPT_GGW_DBG);   // ablation switches (0 in production)
    auto kb = ggw_kernel<NB, SOFF, MI, R, true, CW>;
    auto kf = ggw_kernel<NB, SOFF, MI, R, false, CW>;
    if (lds > 64 * 1024) {
        static size_t raised[2] = {0, 0};
        const int which = y_dtype == PCD_BF16 ? 0 : 1;
        if (raised[which] < lds) {
            if (hipFuncSetAttribute((const void *)(which == 0 ? kb : kf), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess)
                return PCD_ERR_LAUNCH;
            raised[which] = lds;
        }
    }
    if (y_dtype == PCD_BF16)
        kb<<<grid, 256 * (1 + CW), lds, st>>>((const unsigned short *)x, (const uint4 *)wp, bias, nbr, nbr_stride, K, flip, n_out,
                                              n_out_dev, y, x_bytes, w_bytes, addend, bn, dbg);
    else
        kf<<<grid, 256 * (1 + CW), lds, st>>>((const unsigned short *)x, (const uint4 *)wp, bias, nbr, nbr_stride, K, flip, n_out,
                                              n_out_dev, y, x_bytes, w_bytes, addend, bn, dbg);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

#ifdef PCD_EXPERIMENTS
#include "experiments/spconv_kernels.inc"     // ggwin_kernel (128-channel window/stream hybrid), pconv_kernel (pair-driven strided convs)
#endif

// ---------------------------------------------------------------------------------------------
// Data gradient of a STRIDED conv over rows grouped by parity class (pcd_rulebook_conv_classes): a workgroup's
// rows all share the residues ((c + p) mod s) of the three axes, hence the same 1..8 usable offsets (of 27 for
// k = 3, s = 2), and only those are executed -- the generic kernel runs all K offsets for every tile although
// two thirds of its (tile, offset) steps are empty.  Same arithmetic in the same order (skipped steps add exact
// zeros), so the result is bit-identical to the generic kernel's.

template <int NB, int MI, bool OUT_BF16>
__global__ __launch_bounds__(256, gg_waves(NB, MI, 1)) void gather_gemm_cls_kernel(
    const unsigned short *__restrict__ x, int c_in, int cshift, const uint4 *__restrict__ wp,
    const int32_t *__restrict__ nbr, int nbr_stride, int K, const int32_t *__restrict__ perm,
    const int32_t *__restrict__ vstart, ClsTable T, void *__restrict__ yv, int nsteps_total, unsigned x_bytes,
    const void *__restrict__ addend, BnRed bn, int compact) {
    __builtin_amdgcn_s_setprio(PCD_MAIN_PRIO);   // main-chain kernel: issue ahead of the weight-gradient waves sharing the SIMD
    constexpr int ROWS = 4 * MI * 16;
    constexpr int VEC = NB * 64;                           // uint4 per weight stage (one contraction step)
    constexpr int WPT = (VEC + 255) / 256;
    constexpr int c_out = NB * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4 *wbuf = (uint4 *)smem;                           // [2][VEC]
    int *row_s = (int *)(smem + (size_t)2 * VEC * sizeof(uint4));   // [ROWS] real row of every tile row, or -1
    int *nbr_s = row_s + ROWS;                             // [8 + 1][ROWS]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rl = lane & 15, g = lane >> 4;
    // Workgroup b runs on XCD b % 8.  The classes differ 8 x in work per tile (1 .. 8 usable offsets) and the virtual rows are
    // ordered by class, so contiguous runs of tiles per XCD -- the generic kernels' rule, used here until round 6 -- gave XCD 0
    // the 8-offset class and XCD 7 the 1-offset class: the launch lasted 8 / 3.4 = 2.4 x the balanced time.  Now XCD x takes the
    // x-th EIGHTH of every class's tiles: equal work, and the same spatial eighth of all classes (they gather the same dy
    // neighbourhood) in one L2.  The BatchNorm partial row of a workgroup is its block index (any bijection onto the grid works).
    const int tile = blockIdx.x;
    int v0 = -1, cls_v = 0;
    if (compact & 2) {                                     // (ablation, option "gg_dbg" bit 8: contiguous runs of tiles per XCD)
        const int t = xcd_tile(vstart[T.ncls], ROWS);
        if (t * ROWS < vstart[T.ncls]) {
            v0 = t * ROWS;
            for (int q = 1; q < T.ncls; ++q)
                if (vstart[q] <= v0) cls_v = q;
        }
    } else {
        const int xcd = blockIdx.x & 7;
        int j = blockIdx.x >> 3;
        for (int q = 0; q < T.ncls; ++q) {
            const int t0 = vstart[q] / ROWS, tc = vstart[q + 1] / ROWS - t0;     // (class starts are multiples of the class tile >= ROWS)
            const int lo = (xcd * tc) >> 3, hi = ((xcd + 1) * tc) >> 3;
            if (j < hi - lo) {
                v0 = (t0 + lo + j) * ROWS;
                cls_v = q;
                break;
            }
            j -= hi - lo;
        }
    }
    if (v0 < 0) {
        if (bn.mode) bnred_zero_row(bn, tile, c_out);
        return;
    }
    // the class is uniform over the workgroup: keep it in an SGPR so that the by-value table is read with scalar
    // loads (a divergent index would make the compiler copy the whole struct to scratch memory per thread)
    const int cls = __builtin_amdgcn_readfirstlane(cls_v);
    const int nk = T.nk[cls];
    int *kk_s = nbr_s + 9 * ROWS;                          // [8] usable offsets of this class
    float *red_s = (float *)(kk_s + 8);                    // [4][2][c_out], only with bn.mode
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) kk_s[j] = T.k[cls][j];
    }
    const int spk_shift = cshift - 5;                      // log2(contraction steps per offset)
    const int S = nk << spk_shift;
    const unsigned wtotal = (unsigned)nsteps_total * VEC;
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, (int)(wtotal * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    auto real_step = [&](int s) {                          // contraction step s of this class -> step of the pack
        const int j = s >> spk_shift;
        return j < nk ? (kk_s[j] << spk_shift) + (s & ((1 << spk_shift) - 1)) : nsteps_total;
    };
    for (int r = threadIdx.x; r < ROWS; r += 256) row_s[r] = perm[v0 + r];
    __syncthreads();
    for (int e = threadIdx.x; e < (nk + 1) * ROWS; e += 256) {
        const int j = e / ROWS, r = e - j * ROWS;
        const int i = row_s[r];
        // (compact: the class-compact table nbr_cls [8][nbr_stride], indexed by permutation slot -- coalesced, no gather)
        nbr_s[e] = (j < nk && i >= 0) ? ((compact & 1) ? nbr[(size_t)j * nbr_stride + v0 + r] : nbr[(size_t)kk_s[j] * nbr_stride + i]) : -1;
    }
    u32x4 wreg[WPT];
    auto load_w = [&](int s) {
        const unsigned base = (unsigned)real_step(s) * VEC;
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const unsigned e = (unsigned)(j * 256) + threadIdx.x;
            wreg[j] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, e < (unsigned)VEC ? (base + e) * 16u : 0xFFFFFFF0u, 0, 0);
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const unsigned e = (unsigned)(j * 256) + threadIdx.x;
            if (e < (unsigned)VEC) reinterpret_cast<u32x4 *>(wbuf)[buf * VEC + e] = wreg[j];
        }
    };
    load_w(0);
    store_w(0);
    __syncthreads();

    f32x4 acc[MI][NB];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mi][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tile_row = wave * (MI * 16) + rl;
    const int row_shift = cshift + 1;
    auto gather = [&](int s, u32x4(&a)[MI], bool &valid) {
        const int j = min(s >> spk_shift, nk);             // beyond the class's offsets: the all -1 row
        const unsigned c0b = (unsigned)(((s & ((1 << spk_shift) - 1)) * 32 + g * 8) * 2);
        valid = false;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int i = nbr_s[j * ROWS + tile_row + mi * 16];
            a[mi] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ((unsigned)i << row_shift) + c0b, 0, 0);
            valid |= (i >= 0);
        }
    };
    auto compute = [&](int cur, const u32x4(&a)[MI], bool valid) {
        if (__any(valid)) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                bf16x8 b = as_bf16x8(wbuf[cur * VEC + nb * 64 + lane]);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, __builtin_bit_cast(bf16x8, a[mi]),
                                                                          acc[mi][nb], 0, 0, 0);
            }
        }
    };
    u32x4 a0[MI], a1[MI];
    bool v0v, v1v;
    gather(0, a0, v0v);
    for (int s = 0; s < S; s += 2) {
        load_w(s + 1);
        gather(s + 1, a1, v1v);
        compute(0, a0, v0v);
        store_w(1);
        __syncthreads();
        load_w(s + 2);
        gather(s + 2, a0, v0v);
        compute(1, a1, v1v);                                // step S (odd S): all -1 -> skipped
        store_w(0);
        __syncthreads();
    }

    int rows[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) rows[mi] = row_s[tile_row + mi * 16];
    gg_epilogue<MI, NB, OUT_BF16>(acc, rows, c_out, 0, g, rl, wave, tile, nullptr, addend, yv, bn, red_s);
}

template <int NB, int MI>
static int launch_gg_cls(const void *x, int c_in, int cshift, const void *wp, const int32_t *nbr, int nbr_stride,
                         int K, const int32_t *perm, const int32_t *vstart, const ClsTable &T, int vcap, void *y,
                         int y_dtype, int nsteps, unsigned x_bytes, hipStream_t st, const void *addend,
                         const PcdBnReduce *bnr, int *tiles_only, int compact) {
    constexpr int ROWS = 4 * MI * 16;
    int grid = (pcd_div_up(pcd_div_up(vcap, ROWS), 8) + 8) * 8;       // (+ 8 per XCD: its eighths of up to 8 classes round up)
    if (tiles_only) {
        *tiles_only = grid;
        return PCD_OK;
    }
    BnRed bn;
    if (int rc = make_bnred(bnr, y_dtype, NB * 16, grid, &bn)) return rc;
    size_t lds = (size_t)2 * NB * 64 * sizeof(uint4) + (size_t)(1 + 9) * ROWS * sizeof(int) + 8 * sizeof(int) +
                 (bn.mode ? (size_t)8 * NB * 16 * sizeof(float) : 0);
    if (y_dtype == PCD_BF16)
        gather_gemm_cls_kernel<NB, MI, true><<<grid, 256, lds, st>>>(
            (const unsigned short *)x, c_in, cshift, (const uint4 *)wp, nbr, nbr_stride, K, perm, vstart, T, y,
            nsteps, x_bytes, addend, bn, compact);
    else
        gather_gemm_cls_kernel<NB, MI, false><<<grid, 256, lds, st>>>(
            (const unsigned short *)x, c_in, cshift, (const uint4 *)wp, nbr, nbr_stride, K, perm, vstart, T, y,
            nsteps, x_bytes, addend, bn, compact);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

// ---------------------------------------------------------------------------------------------
// weight gradient
template <int C>
struct WgradStride {  // elements; C/16 even -> pad by 16 elements (8 dwords)
    static constexpr int value = ((C / 16) % 2 == 0) ? C + 16 : C;
};

// Pair lists of a STRIDED conv without the lists (pcd_sparse_conv_wgrad_classes): offset k is usable by the rows of exactly one
// stride-parity class (ClsTable above), and for those rows it almost always has an output (every input row creates the
// outputs it reaches; only the grid border cuts some).  So the pairs of offset k ARE {(i, nbr_in[k][i]) : i in class(k)}:
// `pairs` is then the class permutation (ascending rows per class, -1 behind each class's last row), the second index
// comes from the neighbour table, and a missing output (-1) gathers a zero row.
struct WgClasses {
    const int32_t *nbr_in;          // nullptr: explicit pair lists
    const int32_t *vstart;          // [ncls + 1] first permutation slot of every class
    int stride;                     // row stride of nbr_in
    int compact;                    // nbr_in is the class-compact table nbr_cls [8][stride]: entry (j_of_k[k], permutation slot)
    unsigned char cls_of_k[28];
    unsigned char j_of_k[28];
};

template <int MB, int NBW>
__global__ __launch_bounds__(256) void wgrad_kernel(
    const unsigned short *__restrict__ x, int cin_pad, int cin, const unsigned short *__restrict__ dy,
    int cout, const int32_t *__restrict__ pairs, const int32_t *__restrict__ pair_num, int K, int pmax,
    int rows_per_split, int n_splits, int n_chunks, int n_cout_chunks, float *__restrict__ slab,
    unsigned x_bytes, unsigned dy_bytes, int n_x_cap, const int32_t *__restrict__ n_x_dev, WgClasses I) {
    constexpr int CI = MB * 16, CO = NBW * 16;
    // the row-range splits partition the REAL input rows (static-shape mode: n_x_cap is a capacity, the count lives
    // on the device): with capacity-sized ranges the last splits -- the last XCDs' share -- would be empty
    if (n_x_dev) rows_per_split = (eff_rows(n_x_dev, n_x_cap) + n_splits - 1) / n_splits;
    // LDS row strides (elements).  In dwords the stride is 8 * odd, so that (i) the 8 consecutive pair rows a
    // 32-lane half touches in one ds_read_b64_tr_b16 start on 8 distinct 8-dword windows of the 64 banks and
    // (ii) 4 consecutive rows x 32 B written by 8 consecutive lanes (ds_write_b128) cover 32 distinct banks.
    constexpr int XS = WgradStride<CI>::value, YS = WgradStride<CO>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    unsigned short *Xs = (unsigned short *)smem + (size_t)wave * 32 * (XS + YS);
    unsigned short *Ys = Xs + 32 * XS;
    float *tile = (float *)smem;  // reduction tile aliases the staging area (used after the main loop)

    // 1-D grid of (split, k, chunk) items, contiguous runs per XCD (block b -> XCD b % 8): all offsets and
    // channel chunks of one ROW RANGE run on one XCD, whose L2 then holds that range of X and dY.
    const int items = K * n_splits * n_chunks;
    const int per_xcd = gridDim.x >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= items) return;
    const int split = item / (K * n_chunks);
    const int rem = item - split * (K * n_chunks);
    const int k = rem / n_chunks;
    const int chunk = rem - k * n_chunks;
    const int cic = chunk / n_cout_chunks;
    const int coc = chunk % n_cout_chunks;
    const int ci0 = cic * CI, co0 = coc * CO;
    bool implicit = I.nbr_in != nullptr;
    int P;
    const int32_t *pin, *pout;
    if (implicit) {
        const int c = I.cls_of_k[k];
        const int v0 = I.vstart[c];
        P = I.vstart[c + 1] - v0;
        pin = pairs + v0;                                   // (the class permutation)
        pout = I.nbr_in + (size_t)k * I.stride;             // indexed by the INPUT ROW
        if (I.compact) {                                    // indexed by the permutation slot, like a pair list
            pout = I.nbr_in + (size_t)I.j_of_k[k] * I.stride + v0;
            implicit = false;
        }
    } else {
        P = pair_num[k];
        pin = pairs + ((size_t)k * 2 + 0) * pmax;
        pout = pairs + ((size_t)k * 2 + 1) * pmax;
    }
    // pairs of offset k are ascending in input row: this split owns input rows [row_lo, row_hi)
    // (unsigned compares: the -1 slots behind a class's last row sort after every row)
    const int row_lo = split * rows_per_split, row_hi = row_lo + rows_per_split;
    // Both lower bounds at once with a 32-ary search: lanes 0-31 look for row_lo, lanes 32-63 for row_hi; every
    // round probes 32 evenly spaced positions per half and a ballot narrows the range 32x, so a 150k-pair list
    // costs 4 dependent loads instead of the 2 x 18 of two scalar binary searches (which was ~1/3 of the
    // kernel's run time at 4096-row splits).
    int p_begin, p_end;
    {
        const int half = lane >> 5, l32 = lane & 31;
        const int target = half ? row_hi : row_lo;
        int lo = 0, hi = P;
        while (__builtin_amdgcn_ballot_w64(hi > lo) != 0ull) {
            const int width = hi - lo;
            const int step = (width + 31) >> 5;
            const int q = lo + l32 * step;
            const bool pred = width > 0 && q < hi && (unsigned)pin[q] < (unsigned)target;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
            const int c = __builtin_popcount((unsigned)(half ? (m >> 32) : m));
            if (width > 0) {
                const int nlo = c ? lo + (c - 1) * step + 1 : lo;
                const int nhi = c ? min(hi, lo + c * step) : lo;
                lo = nlo;
                hi = nhi;
            }
        }
        p_begin = __builtin_amdgcn_readlane(lo, 0);
        p_end = __builtin_amdgcn_readlane(lo, 32);
    }

    f32x4 acc[MB][NBW];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging: one load instruction fetches WHOLE chunk rows -- lane -> (row, 16-byte piece) with all 2*MB pieces of
    // a row in adjacent lanes (8 rows x 128 B per instruction at 64 channels).  The vector L1 works per 128-byte
    // line touched: 32 rows x 32 B per instruction (the previous mapping) cost 4x the line accesses of 8 full rows.
    constexpr int LPX = 2 * MB, LPY = 2 * NBW;           // lanes (pieces) per row
    constexpr int RPX = 64 / LPX, RPY = 64 / LPY;         // rows per load instruction
    const int xrow = lane / LPX, xpc = lane % LPX;
    const int yrow = lane / LPY, ypc = lane % LPY;
    const int g = lane >> 4, t = lane & 15;
    // contraction index k = g*8 + j of the MFMA  <->  staged pair row (j < 4 ? 4g + j : 16 + 4g + j - 4):
    // any bijection works as long as X and dY use the same one; this one keeps each transpose read on 8
    // consecutive LDS rows
    const int trow = 4 * g + (t >> 2);
    // software pipeline: the gathers of the NEXT 32 pairs are in flight while the current 32 are
    // transposed out of LDS and multiplied
    // (rows one step ahead, pair indices two steps ahead, so no dependent-load latency is exposed)
    u32x4 xr[MB], yr[NBW];
    int idx_n = -1;   // lanes 0-31: input row of pair p0 + lane; lanes 32-63: output row of pair p0 + lane - 32
    // implicit lists: the output row is one more dependent load (nbr_in[k][input row]); the input rows are therefore
    // requested one call EARLIER (calls come with p0 ascending by 128), so that no call waits on a load it issued itself
    int a_cur = -1;
    auto fetch_a = [&](int p0) {
        const int p = p0 + (lane & 31);
        a_cur = (p < p_end) ? pin[p] : -1;
    };
    auto load_idx = [&](int p0) {
        if (implicit) {
            const int a = a_cur;
            idx_n = (lane < 32 || a < 0) ? a : pout[a];
            fetch_a(p0 + 128);
            return;
        }
        const int p = p0 + (lane & 31);
        idx_n = -1;
        if (p < p_end) idx_n = (lane < 32) ? pin[p] : pout[p];
    };
    // gathered rows come through raw buffer loads: index -1 (beyond the pair range) or a channel piece beyond
    // the row -> offset past num_records -> hardware returns zeros, no branch, no memory access
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, (int)dy_bytes, 0x00020000);
    const unsigned x_row_bytes = (unsigned)cin_pad * 2u, y_row_bytes = (unsigned)cout * 2u;
    auto load_rows = [&](int idx) {
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int i = __shfl(idx, j * RPX + xrow);
            const int c = ci0 + xpc * 8;
            unsigned off = (c < cin_pad) ? (unsigned)i * x_row_bytes + (unsigned)c * 2u : 0xFFFFFFF0u;
            xr[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int o = __shfl(idx, 32 + j * RPY + yrow);
            const int c = co0 + ypc * 8;
            unsigned off = (c < cout) ? (unsigned)o * y_row_bytes + (unsigned)c * 2u : 0xFFFFFFF0u;
            yr[j] = __builtin_amdgcn_raw_buffer_load_b128(yrs, off, 0, 0);
        }
    };
    const int p_first = p_begin + wave * 32;
    if (implicit) fetch_a(p_first);
    load_idx(p_first);
    load_rows(idx_n);
    load_idx(p_first + 128);
    for (int p0 = p_first; p0 < p_end; p0 += 128) {
        // stage the 32 gathered rows of X and dY (this wave's private LDS slice)
#pragma unroll
        for (int j = 0; j < MB; ++j)
            *reinterpret_cast<u32x4 *>(Xs + (j * RPX + xrow) * XS + xpc * 8) = xr[j];
#pragma unroll
        for (int j = 0; j < NBW; ++j)
            *reinterpret_cast<u32x4 *>(Ys + (j * RPY + yrow) * YS + ypc * 8) = yr[j];
        load_rows(idx_n);
        load_idx(p0 + 256);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // transpose reads: lane (g,t) supplies the address of 4 bf16 of pair row g*8 + (t>>2)
        bf16x8 af[MB], bfr[NBW];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const unsigned short *a0 = Xs + trow * XS + mb * 16 + (t & 3) * 4;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3))) *)(a0));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3))) *)(a0 + 16 * XS));
            s16x8 cat = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            af[mb] = __builtin_bit_cast(bf16x8, cat);
        }
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const unsigned short *b0 = Ys + trow * YS + nb * 16 + (t & 3) * 4;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3))) *)(b0));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3))) *)(b0 + 16 * YS));
            s16x8 cat = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            bfr[nb] = __builtin_bit_cast(bf16x8, cat);
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
                acc[mb][nb] =
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mb], bfr[nb], acc[mb][nb], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }

    // fixed-order reduction of the 4 waves through LDS, in REGISTER order: tile[(mb * NBW + nb) * 64 + lane] is the
    // lane's f32x4 -- lane-linear 16-byte accesses, no bank conflicts (the [cout][cin] scatter this replaces put the
    // 16 lanes of a row 64 dwords apart: 16-way conflicts, 3.9 M cycles per launch at 64 x 64 = 9 % of the kernel).
    // (The main loop's padded staging is conflict-free for the transposing reads and nearly so for the writes; an
    //  unpadded XOR-swizzled layout measured the same in isolation and, by letting more of these workgroups become
    //  resident beside the data-gradient kernels, 2 % SLOWER in the training step.)
    __syncthreads();
    float4 *tile4 = reinterpret_cast<float4 *>(tile);
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    float4 *dst = tile4 + (mb * NBW + nb) * 64 + lane;
                    float4 v = make_float4(acc[mb][nb][0], acc[mb][nb][1], acc[mb][nb][2], acc[mb][nb][3]);
                    if (w != 0) {
                        const float4 o = *dst;
                        v.x = o.x + v.x; v.y = o.y + v.y; v.z = o.z + v.z; v.w = o.w + v.w;
                    }
                    *dst = v;
                }
        }
        __syncthreads();
    }
    // slab[split][cout][K][cin]; element (co, ci..ci+3) lives at lane (g = ci % 16 / 4, t = co % 16) of block
    // (mb = ci / 16, nb = co / 16)
    float *sl = slab + (size_t)split * cout * K * cin;
    for (int e = threadIdx.x; e < CI * CO / 4; e += 256) {
        const int ci = (e % (CI / 4)) * 4, co = e / (CI / 4);
        const float4 v = tile4[((ci >> 4) * NBW + (co >> 4)) * 64 + ((ci & 15) >> 2) * 16 + (co & 15)];
        if (co0 + co >= cout) continue;
        float *d = sl + ((size_t)(co0 + co) * K + k) * cin + ci0 + ci;
        if ((cin & 3) == 0) {   // 16-byte stores: consecutive cin are contiguous in the slab
            if (ci0 + ci < cin) *reinterpret_cast<float4 *>(d) = v;
        } else {
            const float q[4] = {v.x, v.y, v.z, v.w};
            for (int j = 0; j < 4; ++j)
                if (ci0 + ci + j < cin) d[j] = q[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 128 x 128 channels.  Two things differ from wgrad_kernel:
//  * the four 64 x 64 chunks of dW_k belong to the four WAVES of one workgroup, which stage every gathered row of X and
//    dY ONCE per workgroup (double-buffered in LDS) instead of once per chunk: 16 load instructions per 32 pairs
//    instead of 32, and no cross-wave reduction (every wave owns its chunk of the tile);
//  * the work is cut into EQUAL PAIR COUNTS, not into (offset, row range) items.  The pair lists of the 27 offsets
//    differ 2.4 x in length (centre : corner) and all workgroups are resident at once, so with one row-range grid for
//    every offset the launch lasted as long as its centre-offset items while the SIMDs held waves 50 % of the time
//    (SQ_WAVE_CYCLES).  Here the concatenation of all pair lists (offset-major) is cut into `nb` chunks of T pairs
//    (T a multiple of the 32-pair step); a chunk that crosses an offset boundary writes one tile per offset it
//    touches.  Tile (chunk c, offset k) has index c + k -- unique and, per offset, a contiguous run -- so the fixed-order
//    reduction of offset k is  sum of tiles [first_k, first_k + count_k)  (header at the start of the workspace).
//    A tile is stored in register order (1 KiB per store instruction); wgrad_tiles_reduce_body undoes it.
// Chunks are dealt to the XCDs by their position inside their offset (chunk i of m -> XCD 8 i / m): pair lists are
// ascending in input row, so every XCD sees one eighth of the rows of X and dY for all offsets, which its L2 holds.
constexpr int WG128_HDR_BYTES = 4096;   // {first tile, tile count} per offset (K <= 343), then the tiles

__global__ __launch_bounds__(256, 4) void wgrad128_kernel(
    const unsigned short *__restrict__ x, const unsigned short *__restrict__ dy, const int32_t *__restrict__ pairs,
    const int32_t *__restrict__ pair_num, int K, int pmax, int nb, void *__restrict__ ws, unsigned x_bytes,
    unsigned dy_bytes) {
    constexpr int C = 128, XS = WgradStride<C>::value;   // 144 elements = 72 dwords = 8 * odd
    constexpr int BUF = 32 * XS;                          // elements of one staged 32-row block
    __shared__ __attribute__((aligned(16))) unsigned short stage[2 * 2 * BUF];   // [buffer][X | dY][32][XS]
    __shared__ int pre_s[344];                            // pre_s[k] = pairs of the offsets before k
    __shared__ int sel_s[2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {
        int carry = 0;
        for (int base = 0; base < K; base += 64) {
            const int i = base + lane;
            const int v = i < K ? pair_num[i] : 0;
            const int inc = wave_inclusive_scan(v);
            if (i < K) pre_s[i + 1] = carry + inc;
            carry += __shfl(inc, 63);
        }
        if (lane == 0) pre_s[0] = 0;
    }
    __syncthreads();
    const int Ptot = pre_s[K];
    int T = ((Ptot + nb - 1) / nb + 31) & ~31;
    if (T < 32) T = 32;
    int *hdr = reinterpret_cast<int *>(ws);
    float *tiles = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + WG128_HDR_BYTES);
    if (blockIdx.x == 0) {
        for (int k = threadIdx.x; k < K; k += 256) {
            const int lo = pre_s[k], hi = pre_s[k + 1];
            hdr[2 * k] = lo / T + k;
            hdr[2 * k + 1] = hi > lo ? (hi - 1) / T - lo / T + 1 : 0;
        }
    }
    if (threadIdx.x == 0) {
        // the j-th chunk of XCD xc: offsets in order, chunk i of the m chunks an offset owns (those that START in it)
        // goes to XCD floor(8 i / m)
        const int xc = blockIdx.x & 7;
        int j = blockIdx.x >> 3, c = -1, kk = 0;
        for (int k = 0; k < K; ++k) {
            const int a0 = (pre_s[k] + T - 1) / T, a1 = (pre_s[k + 1] + T - 1) / T, m = a1 - a0;
            const int i0 = (xc * m + 7) / 8, i1 = ((xc + 1) * m + 7) / 8;
            if (j < i1 - i0) {
                c = a0 + i0 + j;
                kk = k;
                break;
            }
            j -= i1 - i0;
        }
        sel_s[0] = c;
        sel_s[1] = kk;
    }
    __syncthreads();
    const int c = sel_s[0];
    if (c < 0) return;
    int k = sel_s[1];

    const int cic = wave >> 1, coc = wave & 1;            // this wave's 64 x 64 chunk
    // staging: wave w gathers pair rows [8w, 8w + 8) of the 32-pair step, 4 whole rows (256 B) per load instruction
    const int srow = lane >> 4, spc = lane & 15;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, (int)dy_bytes, 0x00020000);
    const int g = lane >> 4, t = lane & 15;
    const int trow = 4 * g + (t >> 2);                     // pair row <-> contraction index: as in wgrad_kernel
    struct Rows { u32x4 x[2], y[2]; };

    int q0 = c * T;
    const int q1 = min(q0 + T, Ptot);
    while (q0 < q1) {
        while (pre_s[k + 1] <= q0) ++k;                    // (offsets without pairs)
        const int kend = min(q1, pre_s[k + 1]);
        const int p_begin = q0 - pre_s[k], p_end = kend - pre_s[k];
        const int32_t *pin = pairs + ((size_t)k * 2 + 0) * pmax;
        const int32_t *pout = pairs + ((size_t)k * 2 + 1) * pmax;
        f32x4 acc[4][4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb_ = 0; nb_ < 4; ++nb_) acc[mb][nb_] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // ONE index load per wave and step (lanes 0-31: input rows of the 32 pairs, lanes 32-63: output rows), handed
        // to the staging lanes by shuffles
        auto load_idx = [&](int p0) {
            const int pp = p0 + (lane & 31);
            int v = -1;
            if (pp < p_end) v = (lane < 32) ? pin[pp] : pout[pp];
            return v;
        };
        auto load_rows = [&](int idx) {   // index -1 -> offset beyond num_records -> zeros, no memory access
            Rows R;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int xi = __shfl(idx, wave * 8 + j * 4 + srow);
                const int yi = __shfl(idx, 32 + wave * 8 + j * 4 + srow);
                R.x[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (unsigned)xi * (C * 2u) + spc * 16u, 0, 0);
                R.y[j] = __builtin_amdgcn_raw_buffer_load_b128(yrs, (unsigned)yi * (C * 2u) + spc * 16u, 0, 0);
            }
            return R;
        };
        auto store_rows = [&](int buf, const Rows &R) {
            unsigned short *Xs = stage + buf * 2 * BUF, *Ys = Xs + BUF;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wave * 8 + j * 4 + srow;
                *reinterpret_cast<u32x4 *>(Xs + r * XS + spc * 8) = R.x[j];
                *reinterpret_cast<u32x4 *>(Ys + r * XS + spc * 8) = R.y[j];
            }
        };
        auto compute = [&](int buf) {
            const unsigned short *Xs = stage + buf * 2 * BUF, *Ys = Xs + BUF;
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const unsigned short *a0 = Xs + trow * XS + cic * 64 + mb * 16 + (t & 3) * 4;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(a0));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3))) *)(a0 + 16 * XS));
                af[mb] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int nb_ = 0; nb_ < 4; ++nb_) {
                const unsigned short *b0 = Ys + trow * XS + coc * 64 + nb_ * 16 + (t & 3) * 4;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(b0));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3))) *)(b0 + 16 * XS));
                bfr[nb_] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb_ = 0; nb_ < 4; ++nb_)
                    acc[mb][nb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mb], bfr[nb_], acc[mb][nb_], 0, 0, 0);
        };
        // Software pipeline: rows one step ahead in registers, pair indices one step ahead of their rows (a second
        // step of rows in flight costs 32 VGPRs = one resident workgroup per CU less, and gained nothing).
        // Step s stages pairs [p_begin + 32 s, + 32); beyond p_end everything is -1 / zeros.
        int i1 = load_idx(p_begin);
        Rows A = load_rows(i1);                            // step 0
        i1 = load_idx(p_begin + 32);
        store_rows(0, A);
        __syncthreads();
        int buf = 0;
        for (int p0 = p_begin; p0 < p_end; p0 += 32, buf ^= 1) {
            A = load_rows(i1);                             // step s + 1
            i1 = load_idx(p0 + 64);
            compute(buf);
            store_rows(buf ^ 1, A);                        // (read in the previous step, before that step's barrier)
            __syncthreads();
        }
        // tile (c, k) in register order: acc[mb][nb][r] = dW_k[co = coc*64 + nb*16 + t][ci = cic*64 + mb*16 + g*4 + r]
        float *tile = tiles + (size_t)(c + k) * (C * C);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb_ = 0; nb_ < 4; ++nb_)
                *reinterpret_cast<float4 *>(tile + (((wave * 16 + mb * 4 + nb_) * 64 + lane) << 2)) =
                    make_float4(acc[mb][nb_][0], acc[mb][nb_][1], acc[mb][nb_][2], acc[mb][nb_][3]);
        q0 = kend;
    }
}

// dW[co][k][ci] = sum of the tiles of offset k in tile order (wgrad128_kernel); block -> (k, 1/16 of the tile)
__device__ __forceinline__ void wgrad_tiles_reduce_body(const void *__restrict__ ws, int K, float *__restrict__ dw,
                                                        unsigned block, int tr = 0) {
    constexpr int C = 128;
    const int k = block >> 4, e4 = ((block & 15) << 8) + threadIdx.x;   // float4 index inside the tile (4096 of them)
    const int *hdr = reinterpret_cast<const int *>(ws);
    const float4 *tiles = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(ws) + WG128_HDR_BYTES);
    const int first = hdr[2 * k], count = hdr[2 * k + 1];
    const float4 *src = tiles + (size_t)first * (C * C / 4) + e4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int q = 0;
    for (; q + 3 < count; q += 4) {   // 4 loads in flight, summed in tile order
        const float4 a = src[(size_t)q * (C * C / 4)], b = src[(size_t)(q + 1) * (C * C / 4)];
        const float4 c2 = src[(size_t)(q + 2) * (C * C / 4)], d = src[(size_t)(q + 3) * (C * C / 4)];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        s.x += c2.x; s.y += c2.y; s.z += c2.z; s.w += c2.w;
        s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    }
    for (; q < count; ++q) {
        const float4 a = src[(size_t)q * (C * C / 4)];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    const int lane = e4 & 63, mn = (e4 >> 6) & 15, wave = e4 >> 10;
    const int ci = (wave >> 1) * 64 + (mn >> 2) * 16 + (lane >> 4) * 4, co = (wave & 1) * 64 + (mn & 3) * 16 + (lane & 15);
    if (tr) {          // [cout][cin][K] (see wgrad_reduce_body)
        dw[((size_t)co * C + ci + 0) * K + k] = s.x;
        dw[((size_t)co * C + ci + 1) * K + k] = s.y;
        dw[((size_t)co * C + ci + 2) * K + k] = s.z;
        dw[((size_t)co * C + ci + 3) * K + k] = s.w;
        return;
    }
    *reinterpret_cast<float4 *>(dw + ((size_t)co * K + k) * C + ci) = s;
}

__global__ __launch_bounds__(256) void wgrad_tiles_reduce_kernel(const void *__restrict__ ws, int K,
                                                                 float *__restrict__ dw) {
    wgrad_tiles_reduce_body(ws, K, dw, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Output-stationary weight gradient for the 16-channel layers (level 1: 338 k rows at B = 4, 5 SubM convs + the
// input conv).  The pair-based kernel above gathers TWO 32-byte rows per pair (x and dy), 2 x 7.9 per output row --
// at this width it is bound by those line fetches (~11 TB/s of L2 -> L1 traffic), and these layers are the last of
// the backward pass, where nothing hides them.  Here a wave walks 32 consecutive OUTPUT rows per step: dy is read
// once, in order, and only x is gathered, through nbr_out[k][o], for all K = 27 offsets; the 27 accumulators
// dW_k (16 x 16 each) stay in registers for the workgroup's whole row range.  Half the gathered lines for 3.4x the
// (idle) MFMA work.  Rows staged through LDS and transposed with ds_read_b64_tr_b16 as in wgrad_body.
template <int K>
__global__ __launch_bounds__(256, 2) void wgrad_os16_kernel(
    const unsigned short *__restrict__ x, int cin_pad, int cin, const unsigned short *__restrict__ dy,
    const int32_t *__restrict__ nbr, int nbr_stride, int n_cap, const int32_t *__restrict__ n_dev, int rows_per_wg,
    float *__restrict__ slab, unsigned x_bytes, unsigned dy_bytes) {
    constexpr int G = 9;                                   // offsets staged per group
    static_assert(K % G == 0, "offset groups");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned short *Xs = (unsigned short *)smem + (size_t)wave * (G + 1) * 512;   // [G][32 rows][16 ch]
    unsigned short *Ys = Xs + G * 512;                                            // [32 rows][16 ch]
    float *tile = (float *)smem;                           // [K][16 co][16 ci], after the main loop
    const int g = lane >> 4, t = lane & 15;
    const int xrow = lane >> 1, xpc = lane & 1;            // lane -> (row, 16-byte half of the 32-byte row)
    const int trow = 4 * g + (t >> 2);
    const int n = eff_rows(n_dev, n_cap);
    const int r_begin = blockIdx.x * rows_per_wg;
    const int r_end = min(n, r_begin + rows_per_wg);
    f32x4 acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, (int)dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t nrs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)nbr, 0, (int)((unsigned)K * (unsigned)nbr_stride * 4u), 0x00020000);
    const unsigned x_row_bytes = (unsigned)cin_pad * 2u;
    const bool x_piece_ok = xpc * 8 < cin_pad;
    int idx_cur[K], idx_nxt[K];
    auto load_idx = [&](int r0, int(&idx)[K]) {            // neighbour rows of output row r0 + xrow, all offsets
        const int row = r0 + xrow;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned off = row < r_end ? ((unsigned)k * (unsigned)nbr_stride + (unsigned)row) * 4u : 0xFFFFFFF0u;
            const int v = (int)__builtin_amdgcn_raw_buffer_load_b32(nrs, off, 0, 0);
            idx[k] = row < r_end ? v : -1;
        }
    };
    const int r_first = r_begin + wave * 32;
    load_idx(r_first, idx_cur);
    for (int r0 = r_first; r0 < r_end; r0 += 128) {
        load_idx(r0 + 128, idx_nxt);                        // one step ahead
        const int row = r0 + xrow;
        const u32x4 yv = __builtin_amdgcn_raw_buffer_load_b128(
            yrs, row < r_end ? (unsigned)row * 32u + (unsigned)xpc * 16u : 0xFFFFFFF0u, 0, 0);
        bf16x8 bfr;
#pragma unroll
        for (int grp = 0; grp < K / G; ++grp) {
            u32x4 xr[G];
#pragma unroll
            for (int j = 0; j < G; ++j) {                   // index -1 -> offset beyond the buffer -> zeros
                const int i = idx_cur[grp * G + j];
                const unsigned off = x_piece_ok ? (unsigned)i * x_row_bytes + (unsigned)xpc * 16u : 0xFFFFFFF0u;
                xr[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
            }
            if (grp == 0) *reinterpret_cast<u32x4 *>(Ys + xrow * 16 + xpc * 8) = yv;
#pragma unroll
            for (int j = 0; j < G; ++j) *reinterpret_cast<u32x4 *>(Xs + j * 512 + xrow * 16 + xpc * 8) = xr[j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (grp == 0) {
                const unsigned short *b0 = Ys + trow * 16 + (t & 3) * 4;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(b0));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(b0 + 16 * 16));
                bfr = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const unsigned short *a0 = Xs + j * 512 + trow * 16 + (t & 3) * 4;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(a0));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(a0 + 16 * 16));
                const bf16x8 af = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                acc[grp * G + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[grp * G + j], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int k = 0; k < K; ++k) idx_cur[k] = idx_nxt[k];
    }
    // fixed-order reduction of the 4 waves through LDS, then one slab [cout][K][cin] per workgroup
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float *dst = tile + k * 256 + t * 16 + g * 4 + r;     // [co = t][ci = 4g + r]
                    *dst = (w == 0) ? acc[k][r] : (*dst + acc[k][r]);
                }
        }
        __syncthreads();
    }
    float *sl = slab + (size_t)blockIdx.x * 16 * K * cin;
    for (int e = threadIdx.x; e < K * 256; e += 256) {
        const int k = e >> 8, co = (e >> 4) & 15, ci = e & 15;
        if (ci < cin) sl[((size_t)co * K + k) * cin + ci] = tile[e];
    }
}

constexpr int WGRAD_OS_ROWS = 1024;     // output rows per workgroup (= per slab); 256 / 512 / 2048: 4.20 / 4.14 / 4.12 vs 4.10 ms

// dw[e] = sum over splits of slab[q][e] in a FIXED order: 8 thread groups sum interleaved subsets of the splits
// (q = g, g + 8, ...) with independent loads in flight, then the 8 partial sums are added in group order.
// (One thread per element looping over up to 64 splits left a 16-channel layer with 27 workgroups of 64
// dependent-latency iterations: 20+ us for a 1.7 MB reduction.)
template <int V>   // V = 4: four consecutive elements per thread (16-byte loads), n % 4 == 0;  V = 1: scalar
__device__ __forceinline__ void wgrad_reduce_body(const float *__restrict__ slab, int splits, size_t n,
                                                  float *__restrict__ dw, unsigned block, float *lds,
                                                  int tr_k = 0, int tr_cin = 0, size_t stride = 0, int cin_dst = 0) {
    // n = elements to reduce and write (a PREFIX of every slab when only the first output channels are real),
    // stride = elements between two slabs (0: n)
    if (stride == 0) stride = n;
    float(*part)[32][V] = reinterpret_cast<float(*)[32][V]>(lds);   // [8][32][V]
    const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
    const size_t e = ((size_t)block * 32 + el) * V;
    float s[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = 0.0f;
    auto ld = [&](int q, float (&v)[V]) {
        if (V == 4) {
            float4 t = *reinterpret_cast<const float4 *>(slab + (size_t)q * stride + e);
            v[0] = t.x; v[1 % V] = t.y; v[2 % V] = t.z; v[3 % V] = t.w;
        } else {
            v[0] = slab[(size_t)q * stride + e];
        }
    };
    if (e < n) {
        int q = g;
        for (; q + 24 < splits; q += 32) {
            float a0[V], a1[V], a2[V], a3[V];
            ld(q, a0); ld(q + 8, a1); ld(q + 16, a2); ld(q + 24, a3);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                s[j] += a0[j];
                s[j] += a1[j];
                s[j] += a2[j];
                s[j] += a3[j];
            }
        }
        for (; q < splits; q += 8) {
            float a0[V];
            ld(q, a0);
#pragma unroll
            for (int j = 0; j < V; ++j) s[j] += a0[j];
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) part[g][el][j] = s[j];
    __syncthreads();
    if (g == 0 && e < n) {
        float r[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            r[j] = part[0][el][j];
#pragma unroll
            for (int t = 1; t < 8; ++t) r[j] += part[t][el][j];
        }
        if (cin_dst > 0) {
            // the slabs carry tr_cin input channels per (co, k), the parameter only the first cin_dst (a layer run on zero-padded
            // input rows): element e = (co K + k) tr_cin + ci -> (co K + k) cin_dst + ci, ci < cin_dst
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const size_t ej = e + j;
                const size_t ci = ej % (size_t)tr_cin, t = ej / (size_t)tr_cin;
                if (ci < (size_t)cin_dst) dw[t * cin_dst + ci] = r[j];
            }
        } else if (tr_k > 0) {
            // nn.Conv2d parameter layout [cout][cin][K] instead of [cout][K][cin] (the dense 3x3 convs of the BEV stack
            // write their gradient straight into .grad): element e = (co K + k) cin + ci -> (co cin + ci) K + k
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const size_t ej = e + j;
                const size_t ci = ej % (size_t)tr_cin, t = ej / (size_t)tr_cin;
                const size_t k = t % (size_t)tr_k, co = t / (size_t)tr_k;
                dw[(co * tr_cin + ci) * tr_k + k] = r[j];
            }
        } else if (V == 4)
            *reinterpret_cast<float4 *>(dw + e) = make_float4(r[0], r[1 % V], r[2 % V], r[3 % V]);
        else
            dw[e] = r[0];
    }
}

template <int V>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ slab, int splits,
                                                           size_t n, float *__restrict__ dw) {
    __shared__ float lds[8 * 32 * V];
    wgrad_reduce_body<V>(slab, splits, n, dw, blockIdx.x, lds);
}

// The slab reductions of up to PCD_WGRAD_MAX_JOBS layers in ONE launch (jobs in the kernel arguments): the
// weight-gradient stream is as long as the main chain, and every small kernel on it costs ~3.7 us of latency.
struct RedJobs {
    struct {
        const float *slab;
        float *dw;
        unsigned long long n, stride;
        int splits, vec;
        unsigned first_block;
        int tr_k, tr_cin;        // > 0: write [cout][cin][K]
        int cin_dst;             // > 0: dw has cin_dst < tr_cin input channels per (co, k)
    } job[PCD_WGRAD_MAX_JOBS];
    int n_jobs;
};
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(RedJobs J) {
    __shared__ float lds[8 * 32 * 4];
    int j = 0;
    for (int q = 1; q < J.n_jobs; ++q)
        if (J.job[q].first_block <= blockIdx.x) j = q;
    j = __builtin_amdgcn_readfirstlane(j);
    const unsigned block = blockIdx.x - J.job[j].first_block;
    if (J.job[j].vec == 2)   // tiles of wgrad128_kernel (n = kernel volume)
        wgrad_tiles_reduce_body(J.job[j].slab, (int)J.job[j].n, J.job[j].dw, block, J.job[j].tr_k);
    else if (J.job[j].vec)
        wgrad_reduce_body<4>(J.job[j].slab, J.job[j].splits, (size_t)J.job[j].n, J.job[j].dw, block, lds,
                             J.job[j].tr_k, J.job[j].tr_cin, (size_t)J.job[j].stride, J.job[j].cin_dst);
    else
        wgrad_reduce_body<1>(J.job[j].slab, J.job[j].splits, (size_t)J.job[j].n, J.job[j].dw, block, lds,
                             J.job[j].tr_k, J.job[j].tr_cin, (size_t)J.job[j].stride, J.job[j].cin_dst);
}

// splits = ranges of INPUT rows (pmax = number of input rows = row stride of `pairs`)
// (layers that are cut into >= 4 channel chunks already have 4x the workgroups: twice the rows per split there,
// measured 79 -> 72 us at 128 x 128 channels)
static bool wgrad128_enabled() { return pcd_opt(PCD_OPT_WG128) != 0; }
static int wgrad128_chunks() {   // equal-pair chunks of wgrad128_kernel: two workgroups per CU
    // (768 = three per CU was the isolated optimum; in the step 512 wins by 0.6 % -- 3.424 vs 3.447 ms, 384 / 640: 3.436 /
    //  3.451 -- a third less tile traffic for the reduction: 512 x 64 KiB per layer)
    int v = pcd_opt(PCD_OPT_WG128_CHUNKS);
    if (v < 8) v = 8;
    return (v + 7) / 8 * 8;
}
static bool wgrad128_use(int cin, int cout) { return cin == 128 && cout == 128 && wgrad128_enabled(); }

static void wgrad_plan(int pmax, int cin, int cout, int *splits, int *rows_per_split) {
    const int chunks = pcd_div_up(cin, 64) * pcd_div_up(cout, 64);
    // input rows per split (x 2 for layers cut into >= 4 channel chunks): 6144 measured best in the step (3.420 ms;
    // 3072 / 4096 / 5120 / 7168 / 8192: 3.450 / 3.432 / 3.430 / 3.429 / 3.425) -- fewer, larger slabs to reduce
    const int scale = pcd_opt(PCD_OPT_WG_ROWS) >= 256 ? pcd_opt(PCD_OPT_WG_ROWS) : 6144;
    int s = pcd_div_up(pmax > 0 ? pmax : 1, chunks >= 4 ? 2 * scale : scale);
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    int per = pcd_div_up(pmax > 0 ? pmax : 1, s);
    *splits = s;
    *rows_per_split = per;
}

template <int MB, int NBW>
static int launch_wgrad(const void *x, int n_x, int cin_pad, int cin, const void *dy, int n_dy, int cout,
                        const int32_t *pairs, const int32_t *pair_num, int K, int pmax, float *slab,
                        hipStream_t st, const int32_t *n_x_dev, const WgClasses &I) {
    constexpr int CI = MB * 16, CO = NBW * 16;
    int splits, per;
    wgrad_plan(pmax, cin, cout, &splits, &per);
    per = pcd_div_up(n_x > 0 ? n_x : 1, splits);  // the splits partition the rows of X (pairs[k][0] values)
    int ncic = pcd_div_up(cin, CI), ncoc = pcd_div_up(cout, CO);
    size_t lds_stage = (size_t)4 * 32 * (WgradStride<CI>::value + WgradStride<CO>::value) * 2;
    size_t lds_tile = (size_t)CI * CO * 4;
    size_t lds = lds_stage > lds_tile ? lds_stage : lds_tile;
    int items = K * splits * ncic * ncoc;
    int grid = pcd_div_up(items, 8) * 8;
    wgrad_kernel<MB, NBW><<<grid, 256, lds, st>>>((const unsigned short *)x, cin_pad, cin,
                                                  (const unsigned short *)dy, cout, pairs, pair_num, K,
                                                  pmax, per, splits, ncic * ncoc, ncoc, slab,
                                                  (unsigned)((size_t)n_x * cin_pad * 2),
                                                  (unsigned)((size_t)n_dy * cout * 2), n_x, n_x_dev, I);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

static int chunk_blocks(int c) {  // 16-channel blocks per workgroup tile: 1, 2 or 4
    int b = pcd_div_up(c, 16);
    return b >= 4 ? 4 : (b >= 2 ? 2 : 1);
}

}  // namespace

// =============================================================================================
extern "C" size_t pcd_packed_weight_bytes(int kvol, int cin, int cout, int mode) {
    if (kvol <= 0 || cin <= 0 || cout <= 0 || (mode != 0 && mode != 1)) return 0;
    int cc = pow2_ge8(mode == 0 ? cin : cout);
    int ncol = mode == 0 ? cout : cin;
    size_t nsteps = ((size_t)kvol * cc + 31) / 32;
    size_t nb = (ncol + 15) / 16;
    return nsteps * nb * 64 * 8 * sizeof(unsigned short);
}

extern "C" int pcd_pack_weight(const float *weight, int kvol, int cin, int cout, int mode,
                               void *packed, void *stream) {
    PCD_ENTER();
    if (!weight || !packed || kvol <= 0 || cin <= 0 || cout <= 0 || (mode != 0 && mode != 1))
        return PCD_ERR_INVALID_ARG;
    int cc = pow2_ge8(mode == 0 ? cin : cout);
    int ncol = mode == 0 ? cout : cin;
    int nsteps = (kvol * cc + 31) / 32;
    int NB = (ncol + 15) / 16;
    size_t total = (size_t)nsteps * NB * 64 * 8;
    pack_weight_kernel<<<(unsigned)((total + PACK_PER_BLOCK - 1) / PACK_PER_BLOCK), 256, 0, (hipStream_t)stream>>>(
        weight, kvol, cin, cout, mode, log2_exact(cc), NB, total, (unsigned short *)packed);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_pack_weights_batched(const void *table, int n, int total_blocks, void *stream) {
    PCD_ENTER();
    if (n < 0 || total_blocks < 0 || (n > 0 && !table)) return PCD_ERR_INVALID_ARG;
    if (n == 0 || total_blocks == 0) return PCD_OK;
    pack_weights_batched_kernel<<<(unsigned)total_blocks, 256, 0, (hipStream_t)stream>>>((const long long *)table, n);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

static int gg_dispatch(const void *x, int n_rows_in, int c_in, const void *packed_w, const float *bias,
                       const int32_t *nbr, int nbr_stride, int kvol, int flip_k, int n_rows_out,
                       const int32_t *n_rows_out_dev, int c_out, void *y, int y_dtype, const void *addend,
                       const PcdBnReduce *bnr, int *tiles_only, void *stream, int dir_hint = -1, int zfast = 0,
                       int nbr_packed = 0) {
    if (n_rows_out < 0 || kvol <= 0 || c_in <= 0 || c_out <= 0) return PCD_ERR_INVALID_ARG;
    if (nbr_packed && (flip_k || kvol % 3 != 0)) return PCD_ERR_INVALID_ARG;
    if (y_dtype != PCD_BF16 && y_dtype != PCD_F32) return PCD_ERR_INVALID_ARG;
    if (n_rows_out == 0) {
        if (tiles_only) *tiles_only = 0;
        return PCD_OK;
    }
    if (!tiles_only && (!x || !packed_w || !nbr || !y || nbr_stride < n_rows_out)) return PCD_ERR_INVALID_ARG;
    int cshift = log2_exact(c_in);
    if (cshift < 3 || (c_out % 16) != 0) return PCD_ERR_UNSUPPORTED;
    if (n_rows_in < 0 || (double)n_rows_in * c_in * 2 >= 4294967040.0) return PCD_ERR_UNSUPPORTED;
    const unsigned x_bytes = (unsigned)((size_t)n_rows_in * c_in * 2);
    int nsteps = (kvol * c_in + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    // Launch configuration per output width (NB = c_out/16):  <NB, MI, G (gather look-ahead), SG (stage)>
    //   packed weight <= 32 KB  -> resident in LDS, no barrier in the main loop (narrow / early layers)
    //   else                    -> double-buffered stages of SG steps
    // Narrow layers (NB 1/2, 300k+ rows): short look-ahead, many waves per SIMD.
    const size_t wbytes = (size_t)nsteps * (c_out / 16) * 1024;
    const size_t resident_kb = (size_t)pcd_opt(PCD_OPT_GG_RESIDENT_KB);
    const bool resident = wbytes <= resident_kb * 1024;
    // Wide layers: a contraction step lasts one memory latency (~3.7k clk measured at 128 channels: the gathers
    // and the weight stage are fetched one step ahead of ~256 clk of MFMA work), so two steps of look-ahead with
    // 32 rows per wave win once the register budget is sized for them (gg_waves: 3-4 waves per SIMD, no spills):
    // whole step 4.32 -> 4.19 ms at B = 4.  4 steps ahead, 16 or 64 rows per wave, or 4-step weight stages are
    // slower (4.21-4.63); choosing 16 rows per wave for layers with few rows (< 48k) was also slower
    // (B = 1 / 2: 432 / 676 -> 457 / 712 frames/s without that rule).
    // Wide layers (C_in = 64 / 128, C_out = 64 / 128): the LDS-DMA kernel (ggw_kernel).  Rows per workgroup follow the
    // row count so that the tiles of the LARGEST layers of a level fill the 256 CUs in whole rounds.
    const int ggw_mode = pcd_opt(PCD_OPT_GGW);
    // (C_in = 64: measured equal to the fragment-loading kernel, 55 us at 115 k rows -- both at the texture-address
    //  limit of one 1-KiB instruction per ~32 clk; only PCD_GGW >= 2 routes it here)
    const bool is_dgrad = dir_hint >= 0 ? dir_hint != 0 : (flip_k || (bnr && bnr->mode == 2));
    if (tiles_only && tiles_only[0] == -12345) {       // variant query (pcd_sparse_conv_gather_gemm_variant)
        const int m = pcd_opt(PCD_OPT_GGW);
        tiles_only[0] = (m && (c_in == 128 || (c_in == 64 && m >= 2 && m <= 4)) && (c_out == 64 || c_out == 128) &&
                         x_bytes <= 0xFFFF0000u && !(is_dgrad && m == 6)) ? 1 : 0;
        return PCD_OK;
    }
    // PCD_GGW: 0 = off, 1 = on (default), 2..4 = on with MI rows-per-wave forced (also for C_in = 64), 6 = forward only
    if (ggw_mode && (c_in == 128 || (c_in == 64 && ggw_mode >= 2 && ggw_mode <= 4)) && (c_out == 64 || c_out == 128) &&
        x_bytes <= 0xFFFF0000u && !(is_dgrad && ggw_mode == 6)) {
        if (nbr_packed) return PCD_ERR_UNSUPPORTED;        // (the LDS-DMA kernel stages full tables only)
        const unsigned w_bytes = (unsigned)wbytes;
        int mi = (ggw_mode >= 2 && ggw_mode <= 4) ? ggw_mode : ((c_in == 128 && n_rows_out <= 256 * 192 * 5 / 4) ? 3 : 2)   /* (capacities are 1.25 x the row counts) */;
        // SubM 3x3x3 over z-fastest rows at 128 -> 128 channels: x through windows (ggwin_kernel; same tiles, same BatchNorm rows)
#ifdef PCD_EXPERIMENTS
        if (zfast && !tiles_only && pcd_opt(PCD_OPT_GGWIN) && c_in == 128 && c_out == 128 && kvol == 27 && mi == 3)
            return launch_ggwin<8, 3>(x, packed_w, bias, nbr, nbr_stride, flip_k, n_rows_out, n_rows_out_dev, y, y_dtype, x_bytes,
                                      w_bytes, st, addend, bnr);
#endif
        // Few rows (one round of 128-row tiles fits the chip): the FORWARD conv takes 128-row tiles -- 36-38 us isolated against
        // 42-44 at 21-32 k rows (tools/exp_ggw.py) and nothing runs beside levels 3-4 of the forward pass; the data gradient
        // keeps 192 rows: its workgroups leave ~45 % of the CUs to the weight-gradient kernel running beside it, and with
        // 128-row tiles everywhere the training step was 7 % SLOWER (3.35 against 3.12 ms).  Option "ggw_mi": 2 / 3 = forced.
        // (not with option "ggwin": that kernel and its tile count are built on 192 rows)
        if (mi == 3 && !is_dgrad && c_in == 128 && n_rows_out <= 256 * 128 * 5 / 4 && !pcd_opt(PCD_OPT_GGWIN)) mi = 2;
        if (pcd_opt(PCD_OPT_GGW_MI) == 2 || pcd_opt(PCD_OPT_GGW_MI) == 3) mi = pcd_opt(PCD_OPT_GGW_MI);
#define GGW_ARGS x, packed_w, bias, nbr, nbr_stride, kvol, flip_k, n_rows_out, n_rows_out_dev, y, y_dtype, x_bytes, w_bytes, st, addend, bnr, tiles_only
        // two consumer waves per SIMD (ggw_kernel CW = 2) at 128 -> 128, 192-row tiles: option "ggw_cw" (1 = the one-consumer form)
        if (c_in == 128 && c_out == 128 && mi == 3 && pcd_opt(PCD_OPT_GGW_CW) == 2) return launch_ggw<8, 2, 3, 3, 2>(GGW_ARGS);
#define GGW_MI(NBV, SOFFV)                                                         \
        (mi == 3 ? launch_ggw<NBV, SOFFV, 3, 3>(GGW_ARGS) : launch_ggw<NBV, SOFFV, 2, 3>(GGW_ARGS))
        if (c_in == 64) return c_out == 64 ? GGW_MI(4, 1) : GGW_MI(8, 1);
        return c_out == 64 ? GGW_MI(4, 2) : GGW_MI(8, 2);
#undef GGW_MI
#undef GGW_ARGS
    }
    const int flip_arg = (flip_k ? 1 : 0) | (nbr_packed ? 2 : 0);     // (the kernel's `flip` argument: bit 0 = flipped k, bit 1 = packed table)
#define GG_ARGS x, c_in, cshift, packed_w, bias, nbr, nbr_stride, kvol, flip_arg, n_rows_out, n_rows_out_dev, y, y_dtype, nsteps, x_bytes, st, addend, bnr, tiles_only
    switch (c_out / 16) {
        case 1: {
            // 16 channels (level 1).  With key-ordered voxel rows (pcd_voxelize_hard_sorted) 32 rows per wave and one
            // step of look-ahead win: whole step 3.46 -> 3.42 ms (<1,2,2,0> 3.43, <1,4,*,0> 3.47-3.48, <1,1,1/4,0> 3.46);
            // rows in first-appearance order preferred <1,1,2,0> (PCD_GG1=0).
            const int v1 = pcd_opt(PCD_OPT_GG1);
            if (resident) return v1 ? launch_gg<1, 2, 1, 0>(GG_ARGS) : launch_gg<1, 1, 2, 0>(GG_ARGS);
            return launch_gg<1, 1, 2, 4>(GG_ARGS);
        }
        case 2:
            // (32 channels, staged: <2,2,2,2> / <2,2,2,4> / <2,4,2,2> / <2,2,4,4> / <2,4,1,2> measured 4.20-4.34 vs 4.18)
            if (resident) {
                // (the one resident 32-output-channel layer left on this kernel is the strided 16 -> 32 conv: option "gg2"
                //  selects look-ahead / rows per wave for it)
                switch (pcd_opt(PCD_OPT_GG2)) {
                    case 1: return launch_gg<2, 2, 2, 0>(GG_ARGS);
                    case 2: return launch_gg<2, 1, 2, 0>(GG_ARGS);
                    case 3: return launch_gg<2, 4, 1, 0>(GG_ARGS);
                    case 4: return launch_gg<2, 1, 4, 0>(GG_ARGS);
                    default: return launch_gg<2, 2, 1, 0>(GG_ARGS);
                }
            }
            return launch_gg<2, 2, 1, 2>(GG_ARGS);
        case 4:
            return resident ? launch_gg<4, 2, 2, 0>(GG_ARGS) : launch_gg<4, 2, 2, 2>(GG_ARGS);
        case 8:
            return resident ? launch_gg<8, 2, 2, 0>(GG_ARGS) : launch_gg<8, 2, 2, 2>(GG_ARGS);
        default:
            return PCD_ERR_UNSUPPORTED;
    }
#undef GG_ARGS
}

extern "C" int pcd_sparse_conv_gather_gemm(const void *x, int n_rows_in, int c_in, const void *packed_w,
                                           const float *bias, const int32_t *nbr, int nbr_stride,
                                           int kvol, int flip_k, int n_rows_out,
                                           const int32_t *n_rows_out_dev, int c_out, void *y, int y_dtype,
                                           const void *addend, const PcdBnReduce *bn_reduce, void *stream) {
    PCD_ENTER();
    return gg_dispatch(x, n_rows_in, c_in, packed_w, bias, nbr, nbr_stride, kvol, flip_k, n_rows_out, n_rows_out_dev,
                       c_out, y, y_dtype, addend, bn_reduce, nullptr, stream);
}

// The forward of a strided conv over the PACKED output-side table of pcd_rulebook_conv_cm_build_compact
// (nbr_out_packed [kvol / 3][nbr_stride]); everything else as pcd_sparse_conv_gather_gemm, same result bit for bit.
extern "C" int pcd_sparse_conv_gather_gemm_packed(const void *x, int n_rows_in, int c_in, const void *packed_w,
                                                  const float *bias, const uint32_t *nbr_out_packed, int nbr_stride, int kvol,
                                                  int n_rows_out, const int32_t *n_rows_out_dev, int c_out, void *y,
                                                  int y_dtype, const void *addend, const PcdBnReduce *bn_reduce, void *stream) {
    PCD_ENTER();
    return gg_dispatch(x, n_rows_in, c_in, packed_w, bias, (const int32_t *)nbr_out_packed, nbr_stride, kvol, 0, n_rows_out,
                       n_rows_out_dev, c_out, y, y_dtype, addend, bn_reduce, nullptr, stream, 0, 0, 1);
}

#ifdef PCD_EXPERIMENTS
#include "experiments/spconv_entries.inc"     // pcd_sparse_conv_gather_gemm_zfast, pcd_sparse_conv_pairs*
#endif

extern "C" int pcd_sparse_conv_gather_gemm_tiles_dir(int n_rows_in, int c_in, int kvol, int n_rows_out, int c_out,
                                                      int is_dgrad) {
    int tiles = 0;
    int rc = gg_dispatch(nullptr, n_rows_in, c_in, nullptr, nullptr, nullptr, 0, kvol, 0, n_rows_out, nullptr, c_out,
                         nullptr, PCD_BF16, nullptr, nullptr, &tiles, nullptr, is_dgrad ? 1 : 0);
    return rc == PCD_OK ? tiles : rc;
}

extern "C" int pcd_sparse_conv_gather_gemm_tiles(int n_rows_in, int c_in, int kvol, int n_rows_out, int c_out) {
    return pcd_sparse_conv_gather_gemm_tiles_dir(n_rows_in, c_in, kvol, n_rows_out, c_out, 0);
}

extern "C" int pcd_sparse_conv_gather_gemm_variant(int n_rows_in, int c_in, int kvol, int n_rows_out, int c_out,
                                                   int is_dgrad) {
    int v = -12345;
    int rc = gg_dispatch(nullptr, n_rows_in, c_in, nullptr, nullptr, nullptr, 0, kvol, 0, n_rows_out > 0 ? n_rows_out : 1,
                         nullptr, c_out, nullptr, PCD_BF16, nullptr, nullptr, &v, nullptr, is_dgrad ? 1 : 0);
    return rc == PCD_OK ? v : rc;
}

// a class tile runs only 1..8 of the K offsets: little work per workgroup, so small tiles (more workgroups in
// flight) hide its prologue better than the generic kernel's row-count rule (measured: 1 / 2 / 4 within 10 %)
static int cls_mi(int n_rows_in) { return n_rows_in >= 64 * 1024 ? 2 : 1; }

extern "C" int pcd_sparse_conv_dgrad_classes_tiles(int vcap, int n_rows_in) {
    if (vcap < 0 || n_rows_in < 0) return PCD_ERR_INVALID_ARG;
    if (vcap == 0 || n_rows_in == 0) return 0;
    return (pcd_div_up(pcd_div_up(vcap, 64 * cls_mi(n_rows_in)), 8) + 8) * 8;
}

extern "C" int pcd_sparse_conv_dgrad_classes(const void *dy, int n_dy_rows, int c_dy, const void *packed_w,
                                             const int32_t *nbr_in, int nbr_stride, const int *ksize_host,
                                             const int *stride_host, const int *pad_host, const int *dil_host,
                                             const int32_t *perm, const int32_t *vstart_dev, int vcap,
                                             int n_rows_in, int c_in, void *dx, int dx_dtype, const void *addend,
                                             const PcdBnReduce *bn_reduce, void *stream) {
    return pcd_sparse_conv_dgrad_classes_v2(dy, n_dy_rows, c_dy, packed_w, nbr_in, nbr_stride, 0, ksize_host, stride_host, pad_host,
                                            dil_host, perm, vstart_dev, vcap, n_rows_in, c_in, dx, dx_dtype, addend, bn_reduce,
                                            stream);
}

// nbr_compact = 1: `nbr_in` is the class-compact table nbr_cls [8][nbr_stride] of pcd_rulebook_conv_cm_build_compact
// (nbr_stride = the permutation's capacity vcap): the table reads of a tile are coalesced instead of a gather through perm.
extern "C" int pcd_sparse_conv_dgrad_classes_v2(const void *dy, int n_dy_rows, int c_dy, const void *packed_w,
                                                const int32_t *nbr_in, int nbr_stride, int nbr_compact, const int *ksize_host,
                                                const int *stride_host, const int *pad_host, const int *dil_host,
                                                const int32_t *perm, const int32_t *vstart_dev, int vcap,
                                                int n_rows_in, int c_in, void *dx, int dx_dtype, const void *addend,
                                                const PcdBnReduce *bn_reduce, void *stream) {
    PCD_ENTER();
    if (!ksize_host || !stride_host || !pad_host || !dil_host) return PCD_ERR_INVALID_ARG;
    if (nbr_compact && nbr_stride < vcap) return PCD_ERR_INVALID_ARG;
    if (n_rows_in < 0 || vcap < 0 || c_dy <= 0 || c_in <= 0) return PCD_ERR_INVALID_ARG;
    if (dx_dtype != PCD_BF16 && dx_dtype != PCD_F32) return PCD_ERR_INVALID_ARG;
    if (n_rows_in == 0 || vcap == 0) return PCD_OK;
    if (!dy || !packed_w || !nbr_in || !perm || !vstart_dev || !dx || (!nbr_compact && nbr_stride < n_rows_in))
        return PCD_ERR_INVALID_ARG;
    const int cshift = log2_exact(c_dy);
    if (cshift < 5 || (c_in % 16) != 0) return PCD_ERR_UNSUPPORTED;   // a contraction step must stay inside one offset
    if (n_dy_rows < 0 || (double)n_dy_rows * c_dy * 2 >= 4294967040.0) return PCD_ERR_UNSUPPORTED;
    const int K = ksize_host[0] * ksize_host[1] * ksize_host[2];
    ClsTable T;
    if (int rc = make_cls_table(ksize_host, stride_host, dil_host, T)) return rc;
    const unsigned x_bytes = (unsigned)((size_t)n_dy_rows * c_dy * 2);
    const int nsteps = (K * c_dy + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    // a class tile runs only 1..8 of the K offsets: little work per workgroup, so small tiles (more workgroups in
    // flight) hide its prologue better than the generic kernel's row-count rule
    const int mi = cls_mi(n_rows_in);
    int *tiles_only = nullptr;
    const int compact = (nbr_compact ? 1 : 0) | ((pcd_opt(PCD_OPT_GG_DBG) & 256) ? 2 : 0);
#define CLS_ARGS dy, c_dy, cshift, packed_w, nbr_in, nbr_stride, K, perm, vstart_dev, T, vcap, dx, dx_dtype, nsteps, x_bytes, st, addend, bn_reduce, tiles_only, compact
#define CLS_MI(NBV) (mi == 4 ? launch_gg_cls<NBV, 4>(CLS_ARGS) : mi == 2 ? launch_gg_cls<NBV, 2>(CLS_ARGS) : launch_gg_cls<NBV, 1>(CLS_ARGS))
    switch (c_in / 16) {
        case 1: return CLS_MI(1);
        case 2: return CLS_MI(2);
        case 4: return CLS_MI(4);
        case 8: return CLS_MI(8);
        default: return PCD_ERR_UNSUPPORTED;
    }
#undef CLS_MI
#undef CLS_ARGS
}

extern "C" size_t pcd_sparse_conv_wgrad_workspace_bytes(int kvol, int cin, int cout, int pmax) {
    if (kvol <= 0 || cin <= 0 || cout <= 0 || pmax < 0) return 0;
    if (wgrad128_use(cin, cout))
        return (size_t)WG128_HDR_BYTES + (size_t)(wgrad128_chunks() + kvol) * cin * cout * sizeof(float);
    int splits, per;
    wgrad_plan(pmax, cin, cout, &splits, &per);
    return (size_t)splits * cout * kvol * cin * sizeof(float);
}

extern "C" int pcd_sparse_conv_wgrad(const void *x, int n_x, int cin_pad, int cin, const void *dy, int n_dy,
                                     int cout,
                                     const int32_t *pairs, const int32_t *pair_num, int kvol, int pmax,
                                     float *dweight, void *workspace, size_t workspace_bytes,
                                     void *stream) {
    return pcd_sparse_conv_wgrad_v2(x, n_x, nullptr, cin_pad, cin, dy, n_dy, cout, pairs, pair_num, kvol, pmax, dweight,
                                    workspace, workspace_bytes, stream);
}

static int wgrad_impl(const void *x, int n_x, const int32_t *n_x_dev, int cin_pad, int cin, const void *dy, int n_dy, int cout,
                      const int32_t *pairs, const int32_t *pair_num, int kvol, int pmax, float *dweight, void *workspace,
                      size_t workspace_bytes, void *stream, const WgClasses &I);

extern "C" int pcd_sparse_conv_wgrad_v2(const void *x, int n_x, const int32_t *n_x_dev, int cin_pad, int cin,
                                        const void *dy, int n_dy, int cout, const int32_t *pairs,
                                        const int32_t *pair_num, int kvol, int pmax, float *dweight, void *workspace,
                                        size_t workspace_bytes, void *stream) {
    PCD_ENTER();
    if (pmax > 0 && (!pairs || !pair_num)) return PCD_ERR_INVALID_ARG;
    WgClasses I = {};
    return wgrad_impl(x, n_x, n_x_dev, cin_pad, cin, dy, n_dy, cout, pairs, pair_num, kvol, pmax, dweight, workspace,
                      workspace_bytes, stream, I);
}

// The weight gradient of a strided conv WITHOUT pair lists: the pairs of offset k are read off the parity-class permutation of
// the input rows (pcd_rulebook_conv_cm_build / pcd_rulebook_conv_classes: perm, vstart_dev) and the neighbour table nbr_in
// [kvol][nbr_stride] (see WgClasses).  Same slabs, same reduction job (pcd_sparse_conv_wgrad_reduce*, pmax = n_x) and -- the
// pairs being the same pairs in the same order, plus zero rows where an output is missing -- the same values as
// pcd_sparse_conv_wgrad_v2 over the rulebook's pair lists, to fp32 summation order.  kvol <= 27; not for 128 x 128 channels (PCD_ERR_UNSUPPORTED).
extern "C" int pcd_sparse_conv_wgrad_classes(const void *x, int n_x, const int32_t *n_x_dev, int cin_pad, int cin,
                                             const void *dy, int n_dy, int cout, const int32_t *nbr_in, int nbr_stride,
                                             const int *ksize_host, const int *stride_host, const int *dil_host,
                                             const int32_t *perm, const int32_t *vstart_dev, float *dweight, void *workspace,
                                             size_t workspace_bytes, void *stream, int nbr_compact) {
    PCD_ENTER();
    if (!ksize_host || !stride_host || !dil_host) return PCD_ERR_INVALID_ARG;
    const int kvol = ksize_host[0] * ksize_host[1] * ksize_host[2];
    const int ncls = stride_host[0] * stride_host[1] * stride_host[2];
    if (kvol <= 0 || kvol > 27 || ncls <= 0 || ncls > 8) return PCD_ERR_UNSUPPORTED;
    if (cin > 0 && cout > 0 && wgrad128_use(cin, cout)) return PCD_ERR_UNSUPPORTED;
    if (n_x > 0 && (!nbr_in || !perm || !vstart_dev || nbr_stride < n_x)) return PCD_ERR_INVALID_ARG;
    WgClasses I = {};
    I.nbr_in = nbr_in;
    I.vstart = vstart_dev;
    I.stride = nbr_stride;
    I.compact = nbr_compact ? 1 : 0;
    if (nbr_compact) {                 // (the compact table exists only where a class has at most 8 usable offsets)
        ClsTable T;
        if (int rc = make_cls_table(ksize_host, stride_host, dil_host, T)) return rc;
        for (int c = 0; c < T.ncls; ++c)
            for (int j = 0; j < T.nk[c]; ++j) I.j_of_k[T.k[c][j]] = (unsigned char)j;
    }
    for (int kz = 0; kz < ksize_host[0]; ++kz)
        for (int ky = 0; ky < ksize_host[1]; ++ky)
            for (int kx = 0; kx < ksize_host[2]; ++kx) {
                const int k = (kz * ksize_host[1] + ky) * ksize_host[2] + kx;
                const int rz = (kz * dil_host[0]) % stride_host[0], ry = (ky * dil_host[1]) % stride_host[1],
                          rx = (kx * dil_host[2]) % stride_host[2];       // the class with (r - k d) % s == 0 on every axis
                I.cls_of_k[k] = (unsigned char)((rz * stride_host[1] + ry) * stride_host[2] + rx);
            }
    return wgrad_impl(x, n_x, n_x_dev, cin_pad, cin, dy, n_dy, cout, perm, nullptr, kvol, n_x, dweight, workspace,
                      workspace_bytes, stream, I);
}

static int wgrad_impl(const void *x, int n_x, const int32_t *n_x_dev, int cin_pad, int cin, const void *dy, int n_dy, int cout,
                      const int32_t *pairs, const int32_t *pair_num, int kvol, int pmax, float *dweight, void *workspace,
                      size_t workspace_bytes, void *stream, const WgClasses &I) {
    if (kvol <= 0 || cin <= 0 || cout <= 0 || pmax < 0 || cin_pad < cin || n_x < 0 || n_dy < 0)
        return PCD_ERR_INVALID_ARG;
    if ((double)n_x * cin_pad * 2 >= 4294966000.0 || (double)n_dy * cout * 2 >= 4294966000.0)
        return PCD_ERR_UNSUPPORTED;
    if ((cin_pad % 8) != 0 || (cout % 8) != 0) return PCD_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    size_t n = (size_t)cout * kvol * cin;
    if (pmax == 0) {
        if (dweight) pcd_fill(dweight, 0, n * sizeof(float), st);
        return PCD_OK;
    }
    if (!x || !dy || !pairs) return PCD_ERR_INVALID_ARG;
    if (workspace_bytes < pcd_sparse_conv_wgrad_workspace_bytes(kvol, cin, cout, pmax) || !workspace)
        return PCD_ERR_WORKSPACE;
    float *slab = (float *)workspace;
    if (wgrad128_use(cin, cout)) {
        if (cin_pad != 128 || kvol > 343) return PCD_ERR_UNSUPPORTED;
        const int nb = wgrad128_chunks();
        const int grid = nb + 8 * kvol;   // per XCD: its share of the chunks, + up to one more per offset (rounding)
        wgrad128_kernel<<<grid, 256, 0, st>>>((const unsigned short *)x, (const unsigned short *)dy, pairs, pair_num,
                                              kvol, pmax, nb, workspace, (unsigned)((size_t)n_x * cin_pad * 2),
                                              (unsigned)((size_t)n_dy * cout * 2));
        PCD_RETURN_IF_LAUNCH_FAILED();
        return PCD_OK;
    }
    int mb = chunk_blocks(cin), nb = chunk_blocks(cout);
    int rc = PCD_ERR_UNSUPPORTED;
#define WG(M, N)                                                                                  \
    if (mb == M && nb == N)                                                                       \
        rc = launch_wgrad<M, N>(x, n_x, cin_pad, cin, dy, n_dy, cout, pairs, pair_num, kvol, pmax, slab, st, n_x_dev, I);
    WG(1, 1) WG(1, 2) WG(1, 4) WG(2, 1) WG(2, 2) WG(2, 4) WG(4, 1) WG(4, 2) WG(4, 4)
#undef WG
    return rc;
}

extern "C" int pcd_sparse_conv_wgrad_reduce(int kvol, int cin, int cout, int pmax, float *dweight,
                                            const void *workspace, void *stream) {
    PCD_ENTER();
    if (kvol <= 0 || cin <= 0 || cout <= 0 || pmax < 0 || !dweight) return PCD_ERR_INVALID_ARG;
    size_t n = (size_t)cout * kvol * cin;
    if (pmax == 0) return PCD_OK;  // pcd_sparse_conv_wgrad already zeroed dweight
    if (!workspace) return PCD_ERR_WORKSPACE;
    if (wgrad128_use(cin, cout)) {
        wgrad_tiles_reduce_kernel<<<(unsigned)kvol * 16, 256, 0, (hipStream_t)stream>>>(workspace, kvol, dweight);
        PCD_RETURN_IF_LAUNCH_FAILED();
        return PCD_OK;
    }
    int splits, per;
    wgrad_plan(pmax, cin, cout, &splits, &per);
    if ((n & 3) == 0 && (((uintptr_t)dweight | (uintptr_t)workspace) & 15u) == 0)
        wgrad_reduce_kernel<4><<<(unsigned)((n / 4 + 31) / 32), 256, 0, (hipStream_t)stream>>>(
            (const float *)workspace, splits, n, dweight);
    else
        wgrad_reduce_kernel<1><<<(unsigned)((n + 31) / 32), 256, 0, (hipStream_t)stream>>>(
            (const float *)workspace, splits, n, dweight);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_sparse_conv_wgrad_reduce_batched(const PcdWgradReduceJob *jobs_host, int n_jobs, void *stream) {
    PCD_ENTER();
    if (n_jobs < 0 || n_jobs > PCD_WGRAD_MAX_JOBS || (n_jobs > 0 && !jobs_host)) return PCD_ERR_INVALID_ARG;
    RedJobs J = {};
    unsigned blocks = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const PcdWgradReduceJob &q = jobs_host[i];
        if (q.kvol <= 0 || q.cin <= 0 || q.cout <= 0 || q.pmax < 0 || !q.dweight || (q.layout != 0 && q.layout != 1) ||
            q.cout_write < 0 || q.cout_write > q.cout || q.cin_write < 0 || q.cin_write > q.cin ||
            (q.cin_write > 0 && q.cin_write < q.cin && (q.layout != 0 || q.splits <= 0)))
            return PCD_ERR_INVALID_ARG;
        if (q.pmax == 0 && q.splits <= 0) continue;   // pcd_sparse_conv_wgrad already zeroed dweight
        if (!q.workspace) return PCD_ERR_WORKSPACE;
        int splits, per;
        if (q.splits <= 0 && wgrad128_use(q.cin, q.cout)) {
            if (q.cout_write > 0 && q.cout_write != q.cout) return PCD_ERR_UNSUPPORTED;
            auto &d = J.job[J.n_jobs++];
            d.slab = (const float *)q.workspace;
            d.dw = q.dweight;
            d.n = (unsigned long long)q.kvol;
            d.splits = 0;
            d.vec = 2;
            d.first_block = blocks;
            d.tr_k = q.layout == 1 ? q.kvol : 0;
            d.tr_cin = q.cin;
            blocks += (unsigned)q.kvol * 16;
            continue;
        }
        wgrad_plan(q.pmax, q.cin, q.cout, &splits, &per);
        if (q.splits > 0) splits = q.splits;          // slabs written by pcd_sparse_conv_wgrad_os
        const size_t stride = (size_t)q.cout * q.kvol * q.cin;
        const size_t n = q.cout_write > 0 ? (size_t)q.cout_write * q.kvol * q.cin : stride;   // real rows = a slab prefix
        const bool vec = (n & 3) == 0 && (stride & 3) == 0 && (((uintptr_t)q.dweight | (uintptr_t)q.workspace) & 15u) == 0;
        auto &d = J.job[J.n_jobs++];
        d.cin_dst = (q.cin_write > 0 && q.cin_write < q.cin) ? q.cin_write : 0;
        d.slab = (const float *)q.workspace;
        d.dw = q.dweight;
        d.n = n;
        d.stride = stride;
        d.splits = splits;
        d.vec = vec ? 1 : 0;
        d.first_block = blocks;
        d.tr_k = q.layout == 1 ? q.kvol : 0;
        d.tr_cin = q.cin;
        blocks += (unsigned)(((vec ? n / 4 : n) + 31) / 32);
    }
    if (blocks == 0) return PCD_OK;
    wgrad_reduce_batched_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(J);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

// 16-output-channel layers: see wgrad_os16_kernel.  splits (= slabs to reduce) = ceil(n_out / 1024).
extern "C" int pcd_sparse_conv_wgrad_os_splits(int n_out_rows, int kvol, int cin_pad, int cout) {
    if (n_out_rows <= 0 || kvol != 27 || cout != 16 || (cin_pad != 8 && cin_pad != 16)) return 0;
    return pcd_div_up(n_out_rows, WGRAD_OS_ROWS);
}

extern "C" int pcd_sparse_conv_wgrad_os(const void *x, int n_x, int cin_pad, int cin, const void *dy, int n_out_rows,
                                        const int32_t *n_out_dev, int cout, const int32_t *nbr_out, int nbr_stride,
                                        int kvol, void *workspace, size_t workspace_bytes, void *stream) {
    PCD_ENTER();
    const int splits = pcd_sparse_conv_wgrad_os_splits(n_out_rows, kvol, cin_pad, cout);
    if (splits <= 0 || cin <= 0 || cin > cin_pad) return PCD_ERR_UNSUPPORTED;
    if (!x || !dy || !nbr_out || !workspace || n_x < 0 || nbr_stride < n_out_rows) return PCD_ERR_INVALID_ARG;
    if ((double)n_x * cin_pad * 2 >= 4294966000.0 || (double)n_out_rows * cout * 2 >= 4294966000.0)
        return PCD_ERR_UNSUPPORTED;
    if (workspace_bytes < (size_t)splits * cout * kvol * cin * sizeof(float)) return PCD_ERR_WORKSPACE;
    const size_t lds_stage = (size_t)4 * 10 * 512 * sizeof(unsigned short);   // 4 waves x (9 + 1) row tiles
    const size_t lds_tile = (size_t)27 * 256 * sizeof(float);
    wgrad_os16_kernel<27><<<splits, 256, lds_stage > lds_tile ? lds_stage : lds_tile, (hipStream_t)stream>>>(
        (const unsigned short *)x, cin_pad, cin, (const unsigned short *)dy, nbr_out, nbr_stride, n_out_rows, n_out_dev,
        WGRAD_OS_ROWS, (float *)workspace, (unsigned)((size_t)n_x * cin_pad * 2),
        (unsigned)((size_t)n_out_rows * cout * 2));
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
