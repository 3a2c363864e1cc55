// WINDOW gather-GEMM for SubM 3x3x3 layers over rows numbered z-fastest (PCD_ROWS_YXZ) -- forward and data gradient of
// spconv.SubMConv3d (pcdet/models/backbones_3d/spconv_backbone.py:12,38-45; arithmetic SURVEY.md A.5).
//
// Why a second kernel family.  gather_gemm_kernel / ggw_kernel (spconv.hip) fetch every (output row, offset) operand row by
// itself: 27 x 128 B per output row through the CU's texture-address unit (one 1-KiB instruction per ~34 clk whatever it
// touches), and they re-stream the packed weights through LDS for every tile: 60 us for a 64 -> 64 layer of 115 k rows, 9 % of
// the MFMA peak.  With the rows of a level numbered by (b, y, x, z) the 27 neighbours of a tile of T consecutive rows lie in
// THREE short runs of rows -- the nine offsets that share dy read BEV row y + dy over the tile's x range: median 1.15 x T rows
// each, 94-99 % of the tiles within 2 T (profiles/r04_win_stats.txt; in (b, z, y, x) order 76-86 % of the tiles overflow).  So:
//   * the three runs are DMA'd into LDS once per tile as contiguous 1-KiB instructions (3.3 x T rows instead of 27 x T
//     gathered rows), XOR-swizzled on the source side like ggw_kernel's gather image; tile t + 1 is fetched during tile t;
//   * the rulebook tile nbr[27][T] becomes a table of LDS row indices (0 = a row of zeros for missing neighbours);
//   * the packed weights never move: every wave keeps ITS slice of them in registers for the whole launch (a persistent
//     workgroup per CU, 8 waves x ~110 VGPRs = the 221 KB of a 64 -> 64 layer's 27 offsets).  The slices partition the
//     (offset, output-channel block) space, so the waves that share an output block hold partial sums; these meet in LDS once
//     per tile (fixed order -> deterministic), where bias / addend / the bf16 rounding / the BatchNorm sums are applied and the
//     tile leaves as whole 128-byte lines;
//   * v_mfma_f32_32x32x16_bf16: one 16-byte LDS read per lane feeds 32 output channels (the 16x16x32 form: 16).
// Per 64-row tile at 64 -> 64: 3.5 k clk of MFMA per SIMD, 1.75 k clk of LDS operand reads, 28 KB of DMA.
// Runs longer than the window (rare; any row order at all is still correct, only slow) are processed in several passes over
// chunks of the run: a neighbour lies in exactly one chunk, the other passes read the zero row for it.
// Results equal gather_gemm_kernel's up to the fp32 summation order (offsets are summed per wave slice, then across slices).
#include <type_traits>

#include "common.h"
#include "bn_mid.h"
#include "bnred.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// LDS-DMA (buffer_load_dwordx4 ... lds: lane l's 16 bytes land at M0 + 16 l) as inline assembly, NOT through
// __builtin_amdgcn_raw_ptr_buffer_load_lds: the compiler tracks the builtin as a store to LDS and puts an s_waitcnt vmcnt(0)
// in front of the next ds_read of the same wave -- here that is the first operand read of the MFMA loop, i.e. every wave would
// wait for the prefetch of the NEXT tile before computing the current one (seen in the ISA: the prefetch overlapped nothing).
// The waits are placed by hand (vmcnt(0) + barrier before a window is read).
#ifndef WIN_VARIANT
#define WIN_VARIANT 0
#endif
__device__ __forceinline__ void win_glds16(u32x4 rsrc, char *lds_dst_wave_uniform, unsigned voffset) {
    typedef __attribute__((address_space(3))) char *lds_ptr_t;
#if WIN_VARIANT & 1
    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)(((u64)rsrc[1] << 32) | rsrc[0]), 0, (int)rsrc[2], (int)rsrc[3]),
                                             (lds_ptr_t)lds_dst_wave_uniform, 16, voffset, 0, 0, 0);
    return;
#endif
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_ptr_t)lds_dst_wave_uniform);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(voffset), "s"(rsrc)
                 : "memory", "m0");
}
__device__ __forceinline__ u32x4 win_rsrc(const void *base, unsigned bytes) {
    const u64 a = (u64)(uintptr_t)base;
    return (u32x4){(u32)a, (u32)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}

// raw barrier (a __syncthreads() would drain the DMAs / loads in flight with its fence): this wave's LDS traffic is waited
// for explicitly, the asm statements keep the compiler from moving LDS accesses across the barrier
#define WIN_BARRIER()                                         \
    do {                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_s_barrier();                         \
        asm volatile("" ::: "memory");                        \
    } while (0)

constexpr int WIN_THREADS = 512;
constexpr int WIN_GRID = 256;            // one persistent workgroup per CU; also the number of BatchNorm partial rows
constexpr int WIN_SCRATCH = 8192;        // head of the dynamic LDS: bnred_publish's scratch
constexpr int WIN_PLAN_CAP = 64;         // tiles whose plans are staged in LDS at a time

// Wave roles: wave = (cb, rg, oq) -- output-channel block of 32, group of row blocks, slice of the 27 offsets.
template <int CIN_, int NCB_, int NRG_, int NOQ_, int T_, int R_>
struct WinCfg {
    static constexpr int CIN = CIN_, NCB = NCB_, NRG = NRG_, NOQ = NOQ_, T = T_, R = R_;
    static_assert(NCB * NRG * NOQ == 8, "8 waves");
    static constexpr int COUT = 32 * NCB;
    static constexpr int ROWB = CIN * 2;                 // bytes per feature row
    static constexpr int S = ROWB / 16;                  // 16-byte slots per row
    static constexpr int P = ROWB >= 256 ? 1 : 256 / ROWB;   // rows per 256 bytes of LDS (one sweep of the 64 banks)
    static constexpr int RPI = 1024 / ROWB;              // rows per DMA instruction
    static constexpr int KS = CIN / 16;                  // contraction steps per offset
    static constexpr int OPW = (27 + NOQ - 1) / NOQ;     // offsets per wave
    static constexpr int RB = T / 32, RBW = RB / NRG;    // 32-row blocks per tile / per wave
    static_assert(RB % NRG == 0 && RBW >= 1, "row blocks");
    static constexpr int Z0 = 16;                        // LDS row index of the first window row (row 0 = zeros); 16: one swizzle period
    static constexpr int WINROWS = 3 * R;
    static constexpr int WINB = WINROWS * ROWB;
    static constexpr int TABROWS = 28;                   // table rows: 27 offsets + a row of zeros (the padding offset of the last slice)
    static_assert(NOQ * OPW <= TABROWS, "slices");
    static constexpr int TABN = TABROWS * T;
    static constexpr int TABB = (TABN * 2 + 1023) / 1024 * 1024;   // whole 1-KiB DMA instructions
    static constexpr int NTABI = TABB / 1024;            // ... of them (<= 8: one per wave)
    static_assert(NTABI <= 8, "table pieces");
    static_assert(R / RPI == 16, "two window instructions per run and wave");
    static constexpr int REDSTRIDE = COUT * 4 + 16;      // bytes per (slice, row) of partial sums: +16 keeps b128 stores conflict-free
    static constexpr int REDB = NOQ * T * REDSTRIDE;
    static constexpr int EXTRA = REDB > WINB ? (REDB - WINB + 16 * ROWB - 1) / (16 * ROWB) * (16 * ROWB) : 0;   // whole swizzle periods
    static_assert(WIN_THREADS * 16 * 4 <= WINB + EXTRA, "BatchNorm column staging fits the reduction area");
    // byte offsets into the dynamic LDS
    static constexpr int ROWBASE = WIN_SCRATCH;          // row index 0 lives here
    static constexpr int WIN0 = ROWBASE + Z0 * ROWB;
    static constexpr int XTR = WIN0 + WINB;              // the reduction area of buffer 0 = [WIN0, +REDB), of buffer 1 = [XTR, +REDB)
    static constexpr int WIN1 = XTR + EXTRA;
    static constexpr int WIN1ROW = Z0 + WINROWS + EXTRA / ROWB;
    static constexpr int BOFF = WIN1ROW - Z0;            // table values are buffer-0 row indices; buffer 1 = + BOFF (a multiple of 16: same swizzle)
    static_assert(BOFF % 16 == 0 && WINROWS % 16 == 0, "swizzle period");
    static constexpr int TAB0 = WIN1 + WINB;
    static constexpr int PLAN = TAB0 + 2 * TABB;
    static constexpr int COLS = PLAN + WIN_PLAN_CAP * 32;   // [2 COUT] floats: the workgroup's BatchNorm row
    static constexpr int LDS_BYTES = COLS + 2 * COUT * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static constexpr int NL = (27 * T + WIN_THREADS - 1) / WIN_THREADS;   // rulebook entries per thread and tile (multi-pass tiles only)
    static constexpr size_t plan_bytes(int ntiles) { return (size_t)ntiles * (32 + TABB); }
    static constexpr int CG = COUT / 8;                  // 8-channel groups per row in the tile epilogue
    static_assert(T * CG == WIN_THREADS, "epilogue: one (row, 8 channels) per thread");
    __host__ __device__ static constexpr unsigned swz(unsigned row) { return (row / P) & (S - 1); }
};

using Win64 = WinCfg<64, 2, 1, 4, 64, 128>;      // 64 -> 64: waves = 2 channel blocks x 4 offset slices (7 offsets each)
using Win32 = WinCfg<32, 1, 4, 2, 128, 256>;     // 32 -> 32: waves = 4 row blocks x 2 offset slices (14 offsets each)

// ---- plan: the three runs of every tile + its table of LDS row indices -----------------------------------------------
// header[tile] = {lo0, n0, lo1, n1, lo2, n2, passes, 0}: run g = rows [lo_g, lo_g + n_g) = min .. max of the valid entries of the
// nine table rows k with (k / 3) % 3 == g (dy = g - 1) over the tile's rows; passes = max_g ceil(n_g / R), >= 1.
// table[tile][28][T] u16 (behind the headers, TABB bytes per tile): LDS row of the operand of (offset k, row r) in window
// buffer 0 = Z0 + g R + (nbr[k][r] - lo_g) for neighbours inside the first R rows of their run, 0 (the zero row) otherwise
// (missing neighbour, row beyond n, the part of an over-long run that a later pass covers); row 27 = zeros.  The kernel
// DMAs a tile's table straight into LDS -- nothing is converted on the fly.  The k-flipped view of the data gradient
// reads table row 26 - k for offset k (same windows).  One wave per tile.
template <int T>
__global__ __launch_bounds__(256) void win_plan_kernel(const int32_t *__restrict__ nbr, int nbr_stride, int n_cap,
                                                       const int32_t *__restrict__ n_dev, int R, int Z0, int tabb,
                                                       int4 *__restrict__ plan, int ntiles_cap) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles_cap) return;
    const int lane = threadIdx.x & 63;
    const int n = eff_rows(n_dev, n_cap);
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
    int v[T / 64][27];
#pragma unroll
    for (int u = 0; u < T / 64; ++u) {
        const int row = tile * T + u * 64 + lane;
#pragma unroll
        for (int k = 0; k < 27; ++k) v[u][k] = row < n ? nbr[(size_t)k * nbr_stride + row] : -1;
    }
#pragma unroll
    for (int u = 0; u < T / 64; ++u)
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const int g = (k / 3) % 3;
            if (v[u][k] >= 0) {
                lo[g] = min(lo[g], v[u][k]);
                hi[g] = max(hi[g], v[u][k]);
            }
        }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo[g] = min(lo[g], __shfl_xor(lo[g], d, 64));
            hi[g] = max(hi[g], __shfl_xor(hi[g], d, 64));
        }
    int nn[3], passes = 1;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        nn[g] = hi[g] >= 0 ? hi[g] - lo[g] + 1 : 0;
        if (nn[g] == 0) lo[g] = 0;
        passes = max(passes, (nn[g] + R - 1) / R);
    }
    if (lane == 0) {
        plan[(size_t)tile * 2] = make_int4(lo[0], nn[0], lo[1], nn[1]);
        plan[(size_t)tile * 2 + 1] = make_int4(lo[2], nn[2], passes, 0);
    }
    unsigned short *tab = (unsigned short *)((char *)plan + (size_t)ntiles_cap * 32 + (size_t)tile * tabb);
#pragma unroll
    for (int u = 0; u < T / 64; ++u) {
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const int g = (k / 3) % 3;
            const unsigned rel = (unsigned)(v[u][k] - lo[g]);
            tab[k * T + u * 64 + lane] = (v[u][k] >= 0 && rel < (unsigned)R) ? (unsigned short)(Z0 + g * R + (int)rel) : (unsigned short)0;
        }
        tab[27 * T + u * 64 + lane] = 0;
    }
}

// ---- weight pack: every wave's slice contiguous, MFMA 32x32x16 A-operand order --------------------------------------
// packed[(((cb * NOQ + oq) * OPW + j) * KS + ks) * 64 + lane][e] = W_k[c_out = 32 cb + lane % 32][c_in = 16 ks + 8 (lane / 32) + e],
// k = oq * OPW + j (zeros for k >= 27).  mode 0: forward, weight [c_out][K][c_in]; mode 1: data gradient -- the contraction runs
// over the forward's OUTPUT channels: W_k[co][ci] := weight[ci][k][co] (the k flip is in the rulebook view, like pcd_pack_weight).
template <class C>
__device__ __forceinline__ void win_pack_one(const float *__restrict__ w, int mode, size_t e, unsigned short *out) {
    const int j8 = (int)(e & 7), lane = (int)((e >> 3) & 63);
    size_t t = e >> 9;
    const int ks = (int)(t % C::KS);
    t /= C::KS;
    const int j = (int)(t % C::OPW);
    t /= C::OPW;
    const int oq = (int)(t % C::NOQ), cb = (int)(t / C::NOQ);
    const int k = oq * C::OPW + j;
    const int co = 32 * cb + (lane & 31), ci = 16 * ks + 8 * (lane >> 5) + j8;
    float v = 0.0f;
    if (k < 27) v = mode == 0 ? w[((size_t)co * 27 + k) * C::CIN + ci] : w[((size_t)ci * 27 + k) * C::COUT + co];
    out[e] = f32_to_bf16_bits(v);
}
template <class C>
constexpr size_t win_pack_elems() { return (size_t)C::NCB * C::NOQ * C::OPW * C::KS * 512; }

__global__ __launch_bounds__(256) void win_pack_kernel(const float *__restrict__ w, int cin, int mode, unsigned short *out) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (cin == 64) {
        if (e < win_pack_elems<Win64>()) win_pack_one<Win64>(w, mode, e, out);
    } else {
        if (e < win_pack_elems<Win32>()) win_pack_one<Win32>(w, mode, e, out);
    }
}

// table[i] = {weight ptr, packed ptr, c_in, mode, first block, 0, 0, 0}
__global__ __launch_bounds__(256) void win_pack_batched_kernel(const long long *__restrict__ table, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(size_t)mid * 8 + 4] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long *row = table + (size_t)lo * 8;
    const float *w = (const float *)row[0];
    unsigned short *out = (unsigned short *)row[1];
    const int cin = (int)row[2], mode = (int)row[3];
    const size_t e = ((size_t)blockIdx.x - (size_t)row[4]) * 256 + threadIdx.x;
    if (cin == 64) {
        if (e < win_pack_elems<Win64>()) win_pack_one<Win64>(w, mode, e, out);
    } else {
        if (e < win_pack_elems<Win32>()) win_pack_one<Win32>(w, mode, e, out);
    }
}

// ---- the kernel ------------------------------------------------------------------------------------------------------
struct WinPlan {                 // scalars only (an array member sent the struct to scratch memory)
    int lo0, lo1, lo2, n0, n1, n2, passes;
    __device__ __forceinline__ int lo(int g) const { return g == 0 ? lo0 : g == 1 ? lo1 : lo2; }
    __device__ __forceinline__ int cnt(int g) const { return g == 0 ? n0 : g == 1 ? n1 : n2; }
};

template <class C>
__global__ __launch_bounds__(WIN_THREADS, 1) void subm_win_kernel(
    const unsigned short *__restrict__ x, const uint4 *__restrict__ wp, const float *__restrict__ bias,
    const int32_t *__restrict__ nbr, int nbr_stride, int flip, int n_cap, const int32_t *__restrict__ n_dev,
    const int4 *__restrict__ plan_g, unsigned short *__restrict__ y, unsigned x_bytes,
    const unsigned short *__restrict__ addend, BnRed bn, int dbg, unsigned long long *trace) {
    __builtin_amdgcn_s_setprio(3);       // main-chain kernel (see spconv.hip: PCD_MAIN_PRIO)
    // profiling aid (pcd_subm_window_set_trace): shader-clock stamps of workgroup 0 / wave 0 at the phase boundaries of its tiles
    int trace_at = 0;
    auto stamp = [&]() {
        if (trace && blockIdx.x == 0 && threadIdx.x == 0 && trace_at < 256) trace[trace_at++] = __builtin_readcyclecounter();
    };
    constexpr int T = C::T, R = C::R, ROWB = C::ROWB, COUT = C::COUT, OPW = C::OPW, KS = C::KS, RBW = C::RBW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stamp();                                 // (trace slot 0: kernel entry)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave8 % C::NCB, rg = (wave8 / C::NCB) % C::NRG, oq = wave8 / (C::NCB * C::NRG);
    const int n = eff_rows(n_dev, n_cap);
    const int nt = (n + T - 1) / T;
    // tiles of this workgroup: every XCD (workgroup b runs on XCD b % 8 -- speed only) gets a contiguous eighth of the real
    // tiles, its workgroups contiguous shares of that: neighbouring tiles' windows overlap, they share the XCD's L2
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, wpx = (int)gridDim.x >> 3;
    const int tpx = (nt + 7) >> 3;
    const int x0 = min(nt, xcd * tpx), nx = min(nt, x0 + tpx) - x0;
    const int t_begin = x0 + (int)((long long)nx * jw / wpx), t_end = x0 + (int)((long long)nx * (jw + 1) / wpx);

    float *cols = (float *)(smem + C::COLS);
    if (t_begin >= t_end) {              // no tile: the BatchNorm row of this workgroup is zero
        if (bn.mode) bnred_publish(bn, blockIdx.x, COUT, [](int) { return 0.0f; }, (int)gridDim.x);
        return;
    }

    // this wave's weights: OPW offsets x KS steps, 4 VGPRs each, resident for the whole launch
    bf16x8 wreg[OPW][KS];
    {
        const uint4 *wsrc = wp + (size_t)((cb * C::NOQ + oq) * OPW) * KS * 64 + lane;
#pragma unroll
        for (int j = 0; j < OPW; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wreg[j][ks] = __builtin_bit_cast(bf16x8, wsrc[(j * KS + ks) * 64]);
    }
    // bias and (BatchNorm mode 2) mean of the epilogue: LDS copies (zeros without a bias) in the area of the final BatchNorm row
    for (int e = tid; e < 2 * COUT; e += WIN_THREADS)
        cols[e] = e < COUT ? (bias ? bias[e] : 0.0f) : (bn.mode == 2 ? bn.mean[e - COUT] : 0.0f);
    // the zero row
    for (int e = tid; e < C::Z0 * ROWB / 4; e += WIN_THREADS) ((int *)(smem + C::ROWBASE))[e] = 0;

    const u32x4 xdma = win_rsrc(x, x_bytes);
    const __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)nbr, 0, (int)(27u * (unsigned)nbr_stride * 4u), 0x00020000);
    const size_t out_bytes = (size_t)n_cap * COUT * 2;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, (int)out_bytes, 0x00020000);

    // (epilogue role: one (row, 8 channels) per thread, erow = tid / CG, ecg = tid % CG)
    float bs[8], bq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[e] = bq[e] = 0.0f;

    const int4 *plan_s = (const int4 *)(smem + C::PLAN);
    auto get_plan = [&](int t, int chunk0) {
        const int4 a = plan_s[(t - chunk0) * 2], b = plan_s[(t - chunk0) * 2 + 1];
        WinPlan p;
        p.lo0 = __builtin_amdgcn_readfirstlane(a.x);
        p.n0 = __builtin_amdgcn_readfirstlane(a.y);
        p.lo1 = __builtin_amdgcn_readfirstlane(a.z);
        p.n1 = __builtin_amdgcn_readfirstlane(a.w);
        p.lo2 = __builtin_amdgcn_readfirstlane(b.x);
        p.n2 = __builtin_amdgcn_readfirstlane(b.y);
        p.passes = __builtin_amdgcn_readfirstlane(b.z);
        return p;
    };
    const int ntiles_cap = (n_cap + T - 1) / T;
    const u32x4 tdma = win_rsrc((const char *)plan_g + (size_t)ntiles_cap * 32, (unsigned)((size_t)ntiles_cap * C::TABB));

    // The prefetch of a tile is NSLOT VMEM instructions per wave, issued one at a time BETWEEN the MFMA steps of the previous
    // tile (a 1-KiB instruction occupies the CU's texture-address unit for ~34 clk and the in-order wave behind it: issued as one
    // burst after the barrier, the 40-80 instructions of a tile cost every wave 1.8 k clk before its first MFMA and skewed the
    // waves by another 3 k clk -- measured with pcd_subm_window_set_trace).
    //   slots 0..5: window pieces -- run g = s / 2, 1-KiB instruction wave8 + 8 (s % 2) of the run (R / RPI = 16 of them at most)
    //   slot  6   : piece wave8 of the tile's table (NTABI pieces; table row k of the flipped view = stored row 26 - k)
    constexpr int NSLOT = 7;
    auto issue_slot = [&](const WinPlan p, int tile, int buf, int pass, int slot) {
        if (dbg & 1) return;
        if (slot < 6) {
            const int g = slot >> 1, i = wave8 + 8 * (slot & 1);
            const int cnt = min(R, p.cnt(g) - pass * R);
            if (i * C::RPI < cnt) {
                const int wrow = (buf ? C::WIN1ROW : C::Z0) + g * R + i * C::RPI;          // first LDS row of the piece (wave-uniform)
                const int r_in = i * C::RPI + lane / C::S;
                const unsigned chunk = (unsigned)(lane % C::S) ^ C::swz((unsigned)(wrow + lane / C::S));
                const unsigned off = r_in < cnt ? (unsigned)(p.lo(g) + pass * R + r_in) * (unsigned)ROWB + chunk * 16u : 0xFFFFFF00u;
                win_glds16(xdma, smem + C::ROWBASE + (size_t)wrow * ROWB, off);
            }
        } else if (wave8 < C::NTABI && pass == 0) {
            constexpr int PPR = T * 2 / 16;                      // 16-byte pieces per table row
            const int trow = wave8 * (64 / PPR) + lane / PPR;
            const int src = trow < 27 ? (flip ? 26 - trow : trow) : 27;
            const unsigned off = trow < C::TABROWS
                                     ? (unsigned)tile * (unsigned)C::TABB + (unsigned)(src * T * 2) + (unsigned)(lane % PPR) * 16u
                                     : 0xFFFFFF00u;
            win_glds16(tdma, smem + C::TAB0 + buf * C::TABB + wave8 * 1024, off);
        }
    };

    // (thread-index arithmetic of the non-MFMA phases is recomputed where it is used -- `fresh(tid)` hides the value from
    //  loop-invariant hoisting: every register that lives across the MFMA loop is one the operand pipeline cannot have)
    auto fresh = [](int v) {
        asm volatile("" : "+v"(v));
        return v;
    };
    // multi-pass tiles only (a run longer than the window): the table of pass >= 1 is built here from the int32 rulebook tile
    auto load_nbr = [&](int t, int (&nv)[C::NL]) {
        const int tq = fresh(tid);
#pragma unroll
        for (int u = 0; u < C::NL; ++u) {
            const int e = tq + u * WIN_THREADS;
            const int k = e / T, r = e - k * T;
            const int src = flip ? 26 - k : k;
            const unsigned off = e < 27 * T ? ((unsigned)src * (unsigned)nbr_stride + (unsigned)(t * T + r)) * 4u : 0xFFFFFFF0u;
            nv[u] = __builtin_amdgcn_raw_buffer_load_b32(nrsrc, off, 0, 0);
        }
    };
    auto build_tab = [&](int t, const WinPlan p, const int (&nv)[C::NL], int buf, int pass) {
        unsigned short *tab = (unsigned short *)(smem + C::TAB0 + buf * C::TABB);
        const int tq = fresh(tid);
#pragma unroll
        for (int u = 0; u < C::NL; ++u) {
            const int e = tq + u * WIN_THREADS;
            const int k = e / T, r = e - k * T;
            const int g = ((flip ? 26 - k : k) / 3) % 3;          // group of the STORED row the view's offset k reads
            const int lo = p.lo(g) + pass * R;
            const int v = nv[u];
            const unsigned rel = (unsigned)(v - lo);
            // (every loaded value is consumed unconditionally -- bitwise &, no short circuit: the compiler then places its own
            //  wait for the load here and does not carry "maybe still in flight" registers into the MFMA loop, where it would
            //  protect their reuse with an s_waitcnt vmcnt(0) that also waits for the prefetch of the next tile)
            const bool ok = (e < 27 * T) & (v >= 0) & (t * T + r < n) & (rel < (unsigned)R);
            if (e < 27 * T) tab[e] = ok ? (unsigned short)(C::Z0 + g * R + (int)rel) : (unsigned short)0;
        }
    };

    f32x16 acc[RBW];
    // MFMA loop of a tile: step = (offset j of this wave's slice, row block): KS operand fragments (one 16-byte LDS read per
    // lane each) feed KS MFMAs.  The fragments of step s + 1 are requested before the MFMAs of step s are issued (two register
    // sets): an LDS read takes longer than one MFMA, with a single fragment in flight the matrix pipe idled half of the time.
    // PF: between the steps the wave issues its share of the NEXT tile's prefetch (issue_slot), one instruction per two steps.
    auto compute = [&](int buf, auto pf_tag, const WinPlan pn, int tnext) {
        constexpr bool PF = decltype(pf_tag)::value;
        if (dbg & 4) {
            if (PF)
                for (int q = 0; q < NSLOT; ++q) issue_slot(pn, tnext, buf ^ 1, 0, q);
            return;
        }
        const char *rowbase = smem + C::ROWBASE;
        const unsigned short *tabw =
            (const unsigned short *)(smem + C::TAB0 + buf * C::TABB) + (oq * OPW) * T + rg * RBW * 32 + (lane & 31);
        const unsigned half = (unsigned)lane >> 5;
        const unsigned boff = buf ? (unsigned)C::BOFF : 0u;
        constexpr int NSTEP = OPW * RBW;
        constexpr int PF_FIRST = 3;      // first step followed by a prefetch slot
        static_assert(NSTEP >= PF_FIRST + NSLOT, "a prefetch slot per step");
        bf16x8 fr[2][KS];
        unsigned idxs[NSTEP];            // all table entries of the tile up front: no table read (and its wait) between steps
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) idxs[step] = tabw[(step / RBW) * T + (step % RBW) * 32];
        auto fetch = [&](int step, bf16x8 (&dst)[KS]) {
            // table values are rows of buffer 0 (0 = the zero row); buffer 1 lies BOFF rows (whole swizzle periods) further
            const unsigned idx = idxs[step] ? idxs[step] + boff : 0u;
            // slot of contraction step ks = (2 ks + half) ^ swz(row) = (half ^ swz) ^ 2 ks: one XOR with a constant per step
            const unsigned a0 = idx * (unsigned)ROWB + ((C::swz(idx) ^ half) << 4);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                dst[ks] = *reinterpret_cast<const bf16x8 *>(rowbase + (a0 ^ ((unsigned)ks << 5)));
        };
        fetch(0, fr[0]);
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            if (step + 1 < NSTEP) fetch(step + 1, fr[(step + 1) & 1]);
            // (the scheduler, short of registers, sinks the reads back to one per MFMA unless told not to)
            __builtin_amdgcn_sched_barrier(0);
            const int j = step / RBW, rbw = step % RBW;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc[rbw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[j][ks], fr[step & 1][ks], acc[rbw], 0, 0, 0);
            if (PF) {
                // B1, inside the loop: the prefetch overwrites the area the PREVIOUS tile's epilogue read its partial sums from, so
                // every wave must have left that epilogue -- but nothing before the first slot needs the barrier: the waves run
                // their first steps as they arrive and meet here (the skew of the epilogue is absorbed by MFMA work)
                if (step == PF_FIRST - 1) WIN_BARRIER();
                if (step >= PF_FIRST && step < PF_FIRST + NSLOT) issue_slot(pn, tnext, buf ^ 1, 0, step - PF_FIRST);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using PF1 = std::true_type;
    using PF0 = std::false_type;

    // ---- tile loop, plans staged WIN_PLAN_CAP tiles at a time ----
    for (int chunk0 = t_begin; chunk0 < t_end; chunk0 += WIN_PLAN_CAP) {
        const int chunk1 = min(t_end, chunk0 + WIN_PLAN_CAP);
        __syncthreads();
        for (int e = tid; e < (chunk1 - chunk0) * 2; e += WIN_THREADS) ((int4 *)(smem + C::PLAN))[e] = plan_g[(size_t)chunk0 * 2 + e];
        __syncthreads();
        {   // prologue: tile chunk0 into buffer 0
            const WinPlan p0 = get_plan(chunk0, chunk0);
            for (int q = 0; q < NSLOT; ++q) issue_slot(p0, chunk0, 0, 0, q);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp();                         // (trace slot 1: weights + first window have landed)
            WIN_BARRIER();
        }
        for (int t = chunk0; t < chunk1; ++t) {
            const int buf = (t - chunk0) & 1;
            const WinPlan p = get_plan(t, chunk0);
            // (window + table of tile t were published by the barriers of tile t - 1; B1 = "buffer buf ^ 1 is free" sits inside
            //  compute(), in front of the first prefetch slot)
            stamp();
            const bool more = t + 1 < chunk1;
            // (the plan of the last tile is read twice rather than copied conditionally: a struct merged over a branch was
            //  kept in scratch memory)
            const WinPlan pn = get_plan(more ? t + 1 : t, chunk0);
#pragma unroll
            for (int rbw = 0; rbw < RBW; ++rbw)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[rbw][i] = 0.0f;
            stamp();
            if (more) compute(buf, PF1{}, pn, t + 1); else compute(buf, PF0{}, pn, t);
            stamp();
            for (int pass = 1; pass < p.passes; ++pass) {
                // a run longer than the window: next chunk of every run into the same buffer (slow path, rare)
                int nv2[C::NL];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                WIN_BARRIER();
                for (int q = 0; q < 6; ++q) issue_slot(p, t, buf, pass, q);
                load_nbr(t, nv2);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                build_tab(t, p, nv2, buf, pass);
                WIN_BARRIER();
                compute(buf, PF0{}, p, t);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // window + table of t + 1 landed (this wave's part)
            stamp();
            // the epilogue's operands (this thread's row and 8 channels): in flight across the two barriers below
            const int te = fresh(tid);
            const int erow = te / C::CG, ecg = te % C::CG;
            const int orow = t * T + erow;
            const bool olive = orow < n;
            const size_t oelem = (size_t)orow * COUT + ecg * 8;
            const unsigned ooff = olive ? (unsigned)oelem * 2u : 0xFFFFFFF0u;
            // (rows beyond n read row n - 1: unconditional loads, no select that would wait for them here; the launch-uniform
            //  branches leave the registers undefined when the operand does not exist -- they are only read under the same test)
            const size_t lelem = olive ? oelem : (size_t)(n - 1) * COUT + ecg * 8;
            uint4 av, xv, yv;
            if (addend) av = *reinterpret_cast<const uint4 *>(addend + lelem);
            if (bn.mode == 2) {
                xv = *reinterpret_cast<const uint4 *>(bn.x + lelem);
                if (bn.relu) yv = *reinterpret_cast<const uint4 *>(bn.y + lelem);
            }
            WIN_BARRIER();                               // B2: every wave is done reading window buf -> it becomes the reduction area
            stamp();
            char *red = smem + (buf ? C::XTR : C::WIN0);
            if (!(dbg & 8)) {   // partial sums: lane (n = lane & 31, h = lane >> 5) holds rows (i & 3) + 8 (i >> 2) + 4 h of the 32 x 32 block
                const int h = lane >> 5;
#pragma unroll
                for (int rbw = 0; rbw < RBW; ++rbw) {
                    const int row = (rg * RBW + rbw) * 32 + (lane & 31);
                    char *dst = red + (size_t)(oq * T + row) * C::REDSTRIDE + (cb * 32 + 4 * h) * 4;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        *reinterpret_cast<f32x4 *>(dst + q4 * 32) =
                            (f32x4){acc[rbw][q4 * 4], acc[rbw][q4 * 4 + 1], acc[rbw][q4 * 4 + 2], acc[rbw][q4 * 4 + 3]};
                }
            }
            WIN_BARRIER();                               // B3
            stamp();
            if (!(dbg & 8)) {   // tile epilogue: sum the slices in order, + bias (+ addend), one rounding, whole lines out, BatchNorm sums
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.0f;
#pragma unroll
                for (int q = 0; q < C::NOQ; ++q) {
                    const char *src = red + (size_t)(q * T + erow) * C::REDSTRIDE + ecg * 32;
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(src), b = *reinterpret_cast<const f32x4 *>(src + 16);
                    v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
                    v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
                }
                {
                    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(cols + ecg * 8),
                                b1 = *reinterpret_cast<const f32x4 *>(cols + ecg * 8 + 4);
                    v[0] += b0[0]; v[1] += b0[1]; v[2] += b0[2]; v[3] += b0[3];
                    v[4] += b1[0]; v[5] += b1[1]; v[6] += b1[2]; v[7] += b1[3];
                }
                if (addend) {
                    const u32 aw[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] += __uint_as_float(aw[e] << 16);
                        v[2 * e + 1] += __uint_as_float(aw[e] & 0xffff0000u);
                    }
                }
                u32x4 o;      // v_cvt_pk_bf16_f32: round to nearest even, two values per instruction (= f32_to_bf16_bits for finite values)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = __builtin_bit_cast(u32, __builtin_convertvector((f32x2){v[2 * e], v[2 * e + 1]}, bf16x2));
                __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, ooff, 0, 0);     // (rows >= n: beyond num_records, dropped)
                if (bn.mode && olive) {
                    u32 xw[4] = {0u, 0u, 0u, 0u}, yw[4] = {0u, 0u, 0u, 0u};
                    float mu[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (bn.mode == 2) {
                        xw[0] = xv.x; xw[1] = xv.y; xw[2] = xv.z; xw[3] = xv.w;
                        if (bn.relu) {
                            yw[0] = yv.x; yw[1] = yv.y; yw[2] = yv.z; yw[3] = yv.w;
                        }
                        const f32x4 m0 = *reinterpret_cast<const f32x4 *>(cols + COUT + ecg * 8),
                                    m1 = *reinterpret_cast<const f32x4 *>(cols + COUT + ecg * 8 + 4);
                        mu[0] = m0[0]; mu[1] = m0[1]; mu[2] = m0[2]; mu[3] = m0[3];
                        mu[4] = m1[0]; mu[5] = m1[1]; mu[6] = m1[2]; mu[7] = m1[3];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d0 = __uint_as_float(o[e] << 16), d1 = __uint_as_float(o[e] & 0xffff0000u);
                        if (bn.mode == 1) {
                            bs[2 * e] += d0; bq[2 * e] += d0 * d0;
                            bs[2 * e + 1] += d1; bq[2 * e + 1] += d1 * d1;
                        } else {
                            const float x0f = __uint_as_float(xw[e] << 16), x1f = __uint_as_float(xw[e] & 0xffff0000u);
                            const float y0f = __uint_as_float(yw[e] << 16), y1f = __uint_as_float(yw[e] & 0xffff0000u);
                            const float z0 = (bn.relu && !(y0f > 0.0f)) ? 0.0f : d0, z1 = (bn.relu && !(y1f > 0.0f)) ? 0.0f : d1;
                            bs[2 * e] += z0; bq[2 * e] += z0 * (x0f - mu[2 * e]);
                            bs[2 * e + 1] += z1; bq[2 * e + 1] += z1 * (x1f - mu[2 * e + 1]);
                        }
                    }
                }
            }
            stamp();
        }
    }
    if (bn.mode) {
        // the workgroup's BatchNorm row: column sums over its threads' rows in a fixed order -- rows of a wave by lane
        // exchanges (lanes that differ in the row bits), then the 8 waves through LDS
#pragma unroll
        for (int d = C::CG; d < 64; d <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bs[e] += __shfl_xor(bs[e], d, 64);
                bq[e] += __shfl_xor(bq[e], d, 64);
            }
        __syncthreads();
        float *stage = (float *)(smem + C::WIN0);                        // [8 waves][CG][16]
        if (lane < C::CG) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                stage[(wave8 * C::CG + lane) * 16 + e] = bs[e];
                stage[(wave8 * C::CG + lane) * 16 + 8 + e] = bq[e];
            }
        }
        __syncthreads();
        for (int e = tid; e < 2 * COUT; e += WIN_THREADS) {
            const int which = e / COUT, c = e % COUT;
            float s = 0.0f;
            for (int w = 0; w < 8; ++w) s += stage[(w * C::CG + c / 8) * 16 + which * 8 + (c & 7)];
            if (which == 1 && bn.mode == 2) s *= bn.invstd[c];
            cols[e] = s;
        }
        __syncthreads();
        bnred_publish(bn, blockIdx.x, COUT, [&](int e) { return cols[e]; }, (int)gridDim.x);
    }
    stamp();                                 // (last trace slot: kernel exit)
}

unsigned long long *g_win_trace = nullptr;     // profiling aid, NULL in production (pcd_subm_window_set_trace)

template <class C>
static int launch_win(const void *x, int n_rows, const void *wp, const float *bias, const int32_t *nbr, int nbr_stride,
                      int flip, const int32_t *n_dev, const void *plan, void *y, const void *addend,
                      const PcdBnReduce *bnr, hipStream_t st) {
    BnRed bn;
    if (int rc = make_bnred(bnr, PCD_BF16, C::COUT, WIN_GRID, &bn)) return rc;
    if ((double)n_rows * C::ROWB >= 4294967040.0) return PCD_ERR_UNSUPPORTED;
    auto k = subm_win_kernel<C>;
    // (set per call: the attribute is per device, the call idempotent)
    if (hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
        return PCD_ERR_LAUNCH;
    k<<<WIN_GRID, WIN_THREADS, C::LDS_BYTES, st>>>((const unsigned short *)x, (const uint4 *)wp, bias, nbr, nbr_stride, flip,
                                                    n_rows, n_dev, (const int4 *)plan, (unsigned short *)y,
                                                    (unsigned)((size_t)n_rows * C::ROWB), (const unsigned short *)addend, bn,
                                                    pcd_opt(PCD_OPT_WIN_DBG), g_win_trace);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

static bool win_supported(int c_in, int c_out) { return (c_in == 64 && c_out == 64) || (c_in == 32 && c_out == 32); }

}  // namespace

// =============================================================================================
extern "C" int pcd_subm_window_tile_rows(int c_in, int c_out) {
    if (!win_supported(c_in, c_out)) return 0;
    return c_in == 64 ? Win64::T : Win32::T;
}

extern "C" int pcd_subm_window_partial_rows(void) { return WIN_GRID; }

extern "C" int pcd_subm_window_set_trace(void *buf256_u64) {
    g_win_trace = (unsigned long long *)buf256_u64;
    return PCD_OK;
}

extern "C" size_t pcd_subm_window_plan_bytes(int n_cap, int c_in, int c_out) {
    if (n_cap < 0 || !win_supported(c_in, c_out)) return 0;
    const int nc = n_cap > 0 ? n_cap : 1;
    return c_in == 64 ? Win64::plan_bytes(pcd_div_up(nc, Win64::T)) : Win32::plan_bytes(pcd_div_up(nc, Win32::T));
}

extern "C" int pcd_subm_window_plan(const int32_t *nbr, int nbr_stride, int n_cap, const int32_t *n_dev, int c_in,
                                    int c_out, void *plan, void *stream) {
    PCD_ENTER();
    if (n_cap < 0 || !win_supported(c_in, c_out)) return PCD_ERR_INVALID_ARG;
    if (n_cap == 0) return PCD_OK;
    if (!nbr || !plan || nbr_stride < n_cap) return PCD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (c_in == 64) {
        const int nt = pcd_div_up(n_cap, Win64::T);
        win_plan_kernel<Win64::T><<<pcd_div_up(nt, 4), 256, 0, st>>>(nbr, nbr_stride, n_cap, n_dev, Win64::R, Win64::Z0, Win64::TABB,
                                                                     (int4 *)plan, nt);
    } else {
        const int nt = pcd_div_up(n_cap, Win32::T);
        win_plan_kernel<Win32::T><<<pcd_div_up(nt, 4), 256, 0, st>>>(nbr, nbr_stride, n_cap, n_dev, Win32::R, Win32::Z0, Win32::TABB,
                                                                     (int4 *)plan, nt);
    }
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" size_t pcd_subm_window_packed_weight_bytes(int c_in, int c_out) {
    if (!win_supported(c_in, c_out)) return 0;
    return (c_in == 64 ? win_pack_elems<Win64>() : win_pack_elems<Win32>()) * 2;
}

extern "C" int pcd_subm_window_pack_weight(const float *weight, int c_in, int c_out, int mode, void *packed, void *stream) {
    PCD_ENTER();
    if (!weight || !packed || (mode != 0 && mode != 1) || !win_supported(c_in, c_out)) return PCD_ERR_INVALID_ARG;
    const size_t total = pcd_subm_window_packed_weight_bytes(c_in, c_out) / 2;
    win_pack_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(weight, c_in, mode,
                                                                                     (unsigned short *)packed);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_subm_window_pack_weights_batched(const void *table, int n, int total_blocks, void *stream) {
    PCD_ENTER();
    if (n < 0 || total_blocks < 0 || (n > 0 && !table)) return PCD_ERR_INVALID_ARG;
    if (n == 0 || total_blocks == 0) return PCD_OK;
    win_pack_batched_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>((const long long *)table, n);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_sparse_conv_subm_window(const void *x, int n_rows, int c_in, const void *packed_w, const float *bias,
                                           const int32_t *nbr, int nbr_stride, int flip_k, const int32_t *n_rows_dev,
                                           const void *plan, int c_out, void *y, const void *addend,
                                           const PcdBnReduce *bn_reduce, void *stream) {
    PCD_ENTER();
    if (n_rows < 0 || !win_supported(c_in, c_out)) return PCD_ERR_UNSUPPORTED;
    if (n_rows == 0) return PCD_OK;
    if (!x || !packed_w || !nbr || !plan || !y || nbr_stride < n_rows) return PCD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (c_in == 64)
        return launch_win<Win64>(x, n_rows, packed_w, bias, nbr, nbr_stride, flip_k ? 1 : 0, n_rows_dev, plan, y, addend,
                                 bn_reduce, st);
    return launch_win<Win32>(x, n_rows, packed_w, bias, nbr, nbr_stride, flip_k ? 1 : 0, n_rows_dev, plan, y, addend,
                             bn_reduce, st);
}
