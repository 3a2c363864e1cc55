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
//   * the rulebook tile nbr[27][T] becomes a table of LDS operand slots (64 B per row, built once per rulebook; 0 = a row of
//     zeros for missing neighbours): two VALU instructions from a table entry to an operand address;
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
// The data gradient runs through the same launch with the mode-1 weight pack (transposed, offsets reversed: the k flip of the
// rulebook view lives in the pack).  The WEIGHT gradient of these layers runs over the same tiles: subm_wgrad_win_kernel below.
// Widths: 16 / 32 / 64 (and 128, slower than ggw_kernel: off) -- WinCfg instances; tile shares of equal cost: win_split_kernel.
#include <type_traits>

#include "common.h"
#include "bn_mid.h"
#include "bnred.h"
#include "colmap_common.h"      // column-map lookups: window plans built straight from a level's map (cm_win_plan_kernel)

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
// (the destination is an LDS byte ADDRESS, wave-uniform: a generic pointer would be null-checked on its way to address space 3)
__device__ __forceinline__ void win_glds16(u32x4 rsrc, unsigned lds_addr_wave_uniform, unsigned voffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr_wave_uniform)), "v"(voffset), "s"(rsrc)
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

constexpr int WIN_GRID = 256;            // full configurations (8 waves): (at most) one persistent workgroup per CU
constexpr int WIN_GRID_MAX = 512;        // half configurations (4 waves, round 6): two per CU; sizes the plan's entry table
// workgroups of a forward / data-gradient launch = shares of its plan: option "subm_window_grid" (a multiple of 8, <= WIN_GRID;
// fewer than the CU count leaves CUs to the kernels of the other streams -- a 512-thread, 230-VGPR workgroup starts only on
// an EMPTY CU)
static inline int win_grid8() {
    int g = pcd_opt(PCD_OPT_SUBM_WINDOW_GRID);
    g = g / 8 * 8;
    return g < 8 ? 8 : g > WIN_GRID ? WIN_GRID : g;
}
// workgroups of a launch of configuration C (= BatchNorm partial rows = shares of its plan)
template <class C>
static inline int win_grid() { return win_grid8() * (8 / C::NW); }
constexpr int WIN_SCRATCH = 0;           // (bnred_publish's 8 KiB of scratch at the head of the dynamic LDS are the zero rows and the
                                         //  first window: dead by then)
constexpr int WIN_PLAN_CAP = 64;         // tiles whose plans are staged in LDS at a time
// Plan buffer: [entries: WIN_GRID x 64 B][prefix: ntiles x i32, padded to 32 B][headers: ntiles x 32 B][tables: ntiles x TABB]
//   entry[w] = {first tile, end tile, 0, 0, header of the first tile (8 ints), 0 ..} of workgroup w (in XCD-major order): one
//   scalar load at kernel entry gives a workgroup its share AND what it needs to start the first DMAs.
//   wshare[s] = {first tile, end tile} of share s of the weight-gradient kernel (WinCfg::WG_SHARES shares, 2 KiB reserved)
constexpr int WIN_ENTRY_BYTES = 64;
constexpr int WIN_WG_SHARES = 80;        // weight gradient, 8-wave configurations: 8 XCDs x 10 shares, three workgroups (one per run) each
constexpr int WIN_WG_SHARES_MAX = 160;   // (4-wave configurations: twice as many, half as long)
__host__ __device__ constexpr size_t win_wshare_off() { return (size_t)WIN_GRID_MAX * WIN_ENTRY_BYTES; }
__host__ __device__ constexpr size_t win_prefix_off() { return win_wshare_off() + 2048; }
__host__ __device__ constexpr size_t win_hdr_off(int ntiles) { return win_prefix_off() + ((size_t)ntiles * 4 + 31) / 32 * 32; }
__host__ __device__ constexpr size_t win_tab_off(int ntiles) { return win_hdr_off(ntiles) + (size_t)ntiles * 32; }

// Wave roles: wave = (cb, rg, oq) -- output-channel block of 32, group of row blocks, slice of the 27 offsets.
// CQN > 1: the output channels are dealt to CQN workgroups per share of the tiles (128 channels: the weights of 32 output
// channels fill the registers of a workgroup) -- each loads the same windows and writes its COUTW columns of y.
// NW_ = waves per workgroup: 8 (one 512-thread workgroup per CU) or 4 (round 6: 256 threads, <= 80 KB of LDS -- TWO per CU, so that
// a workgroup of another stream that holds part of a CU delays half a CU's worth of this launch instead of all of it).
template <int CIN_, int NCB_, int NRG_, int NOQ_, int T_, int R_, int CQN_ = 1, int NW_ = 8>
struct WinCfg {
    static constexpr int CIN = CIN_, NCB = NCB_, NRG = NRG_, NOQ = NOQ_, T = T_, R = R_, CQN = CQN_, NW = NW_;
    static constexpr int THREADS = 64 * NW;
    static constexpr int WG_SHARES = WIN_WG_SHARES * 8 / NW;       // weight-gradient shares of the tiles
    static_assert(NCB * NRG * NOQ == NW && (NW == 8 || NW == 4), "wave roles");
    static constexpr int COUT = CIN;                     // square layers only; a wave's MFMA block spans 32 output channels (the
    static constexpr int COUTW = COUT / CQN;             //  upper 16 are zero weights at 16 channels: LDS reads bound that layer, not MFMAs)
    static_assert(32 * NCB >= COUTW && WIN_GRID % CQN == 0 && WG_SHARES <= WIN_WG_SHARES_MAX, "channel blocks");
    static constexpr int NCBP = COUT >= 32 ? COUT / 32 : 1;   // 32-channel blocks of the packed weights
    static constexpr int QN = COUTW >= 32 ? 4 : COUTW / 8;   // live 4-register groups of a lane's 32 x 32 accumulator block
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
    // Table of a tile (in the plan and, DMA'd verbatim, in LDS): 64 bytes = 32 u16 entries per tile row; offset k = oq OPW + j is
    // entry oq SLICE + j (slices padded to whole 16-byte pieces, padding = 0 = the zero row), and the four 16-byte pieces of row r
    // are stored at piece ^ ((r >> 2) & 3): the 16 rows a ds_read_b128 phase touches hit 64 distinct banks.
    static constexpr int SLICE = 32 / NOQ;               // entries per slice
    static_assert(SLICE >= OPW && (SLICE % 8 == 0 || SLICE == 4), "slices");
    static constexpr int SLP = SLICE / 8;                // 16-byte pieces per slice (0: half a piece, SLICE == 4)
    static constexpr int SLD = SLICE / 2;                // dwords per slice
    static constexpr int TABB = T * 64;                  // bytes per tile
    static_assert(TABB % 1024 == 0, "whole 1-KiB DMA instructions");
    static constexpr int NTABI = TABB / 1024;            // ... of them
    static constexpr int TSL = (NTABI + NW - 1) / NW;    // table DMA instructions per wave
    __host__ __device__ static constexpr unsigned tab_pos(unsigned r, unsigned k) {      // u16 index of (tile row r, offset k)
        const unsigned ei = (k / OPW) * SLICE + k % OPW;
        return r * 32 + (((ei >> 3) ^ ((r >> 2) & 3)) << 3) + (ei & 7);
    }
    static constexpr bool DIRECT = NOQ == 1 && NCB == 1; // a wave owns whole rows: no cross-wave sums, the epilogue runs from registers
    // what a further pass over a tile costs, in quarters of a tile's time (measured per workgroup, tools/exp_subm_win.py
    // WIN_BALANCE: 2.84 / 1.32 / 0.94 tiles at 16 / 32 / 64 channels)
    static constexpr int PASS_COST = CIN <= 16 ? 11 : CIN <= 32 ? 5 : 4;
    static constexpr int SPR = (R / RPI + NW - 1) / NW;  // window DMA instructions per run and wave
    static constexpr int NSLOT = 3 * SPR + TSL;          // prefetch instructions per wave and tile
    static constexpr int REDSTRIDE = COUTW * 4 + 16;     // bytes per (slice, row) of partial sums: +16 keeps b128 stores conflict-free
    static constexpr int REDB = NOQ * T * REDSTRIDE;
    static constexpr int EXTRA = REDB > WINB ? (REDB - WINB + 16 * ROWB - 1) / (16 * ROWB) * (16 * ROWB) : 0;   // whole swizzle periods
    static_assert(THREADS * 16 * 4 <= WINB + EXTRA, "BatchNorm column staging fits the reduction area");
    // byte offsets into the dynamic LDS
    static constexpr int ROWBASE = WIN_SCRATCH;          // row index 0 lives here
    static constexpr int WIN0 = ROWBASE + Z0 * ROWB;
    static constexpr int XTR = WIN0 + WINB;
    static constexpr int ZERO1 = XTR + EXTRA;            // Z0 rows of zeros in front of EACH window: entry 0 + BOFF is a zero row too
    static constexpr int WIN1 = ZERO1 + Z0 * ROWB;
    static constexpr int WIN1ROW = Z0 + WINROWS + EXTRA / ROWB + Z0;
    static constexpr int BOFF = WIN1ROW - Z0;            // table values are buffer-0 slots; buffer 1 = + BOFF rows (a multiple of 16: same swizzle)
    // the reduction area (partial sums of a tile, written over its window): buffer 0 = [WIN0, +REDB); buffer 1 = [WIN1, +REDB) when
    // that fits the window, else [XTR, +REDB) -- which runs over the second zero rows: they are cleared again (REZERO)
    static constexpr bool REZERO = EXTRA > 0;
    static constexpr int RED1 = REZERO ? XTR : WIN1;
    static_assert(BOFF % 16 == 0 && WINROWS % 16 == 0, "swizzle period");
    static constexpr int TAB0 = WIN1 + WINB;
    static constexpr int PLAN = TAB0 + 2 * TABB;
    static constexpr int COLS = PLAN + WIN_PLAN_CAP * 32;   // [2 COUT] floats: the workgroup's BatchNorm row
    static constexpr int LDS_BYTES = COLS + 2 * COUT * 4;
    static_assert(LDS_BYTES <= (NW == 8 ? 160 : 80) * 1024, "LDS (two 4-wave workgroups share a CU)");
    static constexpr int NL = (27 * T + THREADS - 1) / THREADS;   // rulebook entries per thread and tile (multi-pass tiles only)
    static constexpr size_t plan_bytes(int ntiles) { return win_tab_off(ntiles) + (size_t)ntiles * TABB; }
    static constexpr int CG = COUTW / 8;                 // 8-channel groups per row in the tile epilogue
    static constexpr int NEPI = T * CG;                  // epilogue threads: one (row, 8 channels) each
    static_assert(NEPI <= THREADS && NEPI % 64 == 0, "epilogue threads");
    __host__ __device__ static constexpr unsigned swz(unsigned row) { return (row / P) & (S - 1); }
    static_assert(P * S == 16 && Z0 % 16 == 0 && R % 16 == 0 && (NW * RPI) % 16 == 0, "swizzle period of 16 rows");
    static_assert(R % RPI == 0, "a run is whole DMA instructions (a partial last piece would run into the next run's rows)");
};

using Win64 = WinCfg<64, 2, 1, 4, 64, 128>;      // 64 -> 64: waves = 2 channel blocks x 4 offset slices (7 offsets each)
using Win32 = WinCfg<32, 1, 4, 2, 128, 320>;     // 32 -> 32: waves = 4 row blocks x 2 offset slices (14 offsets each)
using Win16 = WinCfg<16, 1, 8, 1, 256, 640>;     // 16 -> 16: waves = 8 row blocks, all 27 offsets each (no cross-wave sums)
// 4-wave configurations (option "subm_window_half": bit 1 = 32 channels, bit 2 = 16): the same wave roles on half the rows
using Win32h = WinCfg<32, 1, 2, 2, 64, 176, 1, 4>;
using Win16h = WinCfg<16, 1, 4, 1, 128, 320, 1, 4>;
#ifdef PCD_EXPERIMENTS      // (make EXPERIMENTS=1; 61.6 us against ggw_kernel's 49.0 at level 4 -- DESIGN.md 4.1: not in the default library)
using Win128 = WinCfg<128, 1, 1, 8, 32, 64, 4>;  // 128 -> 128: 4 workgroups x 32 output channels; waves = 8 offset slices (4 each)
#endif

// c_in -> configuration (square layers): f(Cfg{}) with the matching type, `none` otherwise
template <class F, class N>
static inline auto win_dispatch(int c_in, int c_out, F &&f, N none) -> decltype(none) {
    if (c_in != c_out) return none;
    switch (c_in) {
        case 64: return f(Win64{});
        case 32: return (pcd_opt(PCD_OPT_SUBM_WINDOW_HALF) & 2) ? f(Win32h{}) : f(Win32{});
        case 16: return (pcd_opt(PCD_OPT_SUBM_WINDOW_HALF) & 4) ? f(Win16h{}) : f(Win16{});
#ifdef PCD_EXPERIMENTS
        case 128: return f(Win128{});
#endif
        default: return none;
    }
}

// ---- plan: the three runs of every tile + its table of LDS operand slots --------------------------------------------
// header[tile] = {lo0, n0, lo1, n1, lo2, n2, passes, 0}: run g = rows [lo_g, lo_g + n_g) = min .. max of the valid entries of the
// nine rulebook rows k with (k / 3) % 3 == g (dy = g - 1) over the tile's rows; passes = max_g ceil(n_g / R), >= 1.
// table[tile] (behind the headers, TABB bytes per tile, layout WinCfg::tab_pos): where the operand of (offset k, row r) lies in
// window buffer 0, as the index of the 16-byte LDS slot that holds the row's first 8 channels:
//     row = Z0 + g R + (nbr[k][r] - lo_g),   entry = row * S + swz(row)     (S slots per row; slot c of a row is stored at c ^ swz)
// for neighbours inside the first R rows of their run, 0 (the zero row) otherwise (missing neighbour, row beyond n, the part of
// an over-long run that a later pass covers).  A reader of channels 8 c .. 8 c + 7 takes slot entry ^ c: two instructions
// between the table and the LDS address.  The kernel DMAs a tile's table straight into LDS.  The data gradient reads the SAME
// table: its k flip is in the packed weights (win_pack_one).  One wave per tile.
// `src(row, live, v)` delivers the 27 neighbour rows of `row` (-1 = none): from the rulebook table (win_plan_kernel) or straight
// from the level's column map (cm_win_plan_kernel: no table is read -- and none written, except `nbr_out` columns of the tiles
// that need a second pass, which is where the conv kernels look neighbours up again; nbr_full: the whole table).
template <class C, class Src>
__device__ __forceinline__ void win_plan_body(Src src, int n_cap, const int32_t *__restrict__ n_dev, char *__restrict__ plan_base,
                                              int ntiles_cap, int32_t *__restrict__ nbr_out, int nbr_full) {
    constexpr int T = C::T, R = C::R;
    constexpr int TPW = T >= 64 ? 1 : 64 / T;            // tiles per wave (T < 64: every T lanes one tile)
    constexpr int U = T >= 64 ? T / 64 : 1;              // rows per lane
    constexpr int LPT = T >= 64 ? 64 : T;                // lanes per tile
    int4 *plan = (int4 *)(plan_base + win_hdr_off(ntiles_cap));
    const int tile = (blockIdx.x * 4 + (threadIdx.x >> 6)) * TPW + (threadIdx.x & 63) / LPT;
    if (tile >= ntiles_cap) return;                      // (whole T-lane groups leave: the exchanges below stay inside a group)
    const int lane = (threadIdx.x & 63) % LPT;
    const int n = eff_rows(n_dev, n_cap);
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
    int v[U][27];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int row = tile * T + u * 64 + lane;
        src(row, row < n, v[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
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
        for (int d = LPT / 2; d >= 1; d >>= 1) {
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
    char *tab = plan_base + win_tab_off(ntiles_cap) + (size_t)tile * C::TABB;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned r = (unsigned)(u * 64 + lane);
        u32 w[16];                                   // the row's 32 entries
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = 0u;
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const int g = (k / 3) % 3;
            const unsigned rel = (unsigned)(v[u][k] - lo[g]);
            const unsigned row = (unsigned)(C::Z0 + g * R) + rel;
            const unsigned e = (v[u][k] >= 0 && rel < (unsigned)R) ? row * C::S + C::swz(row) : 0u;
            const int ei = (k / C::OPW) * C::SLICE + k % C::OPW;
            w[ei >> 1] |= e << (16 * (ei & 1));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *reinterpret_cast<uint4 *>(tab + r * 64 + ((c ^ ((r >> 2) & 3)) << 4)) = make_uint4(w[4 * c], w[4 * c + 1], w[4 * c + 2], w[4 * c + 3]);
    }
    if (nbr_out && (nbr_full || passes > 1)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = tile * T + u * 64 + lane;
            if (row < n) {
#pragma unroll
                for (int k = 0; k < 27; ++k) nbr_out[(size_t)k * n_cap + row] = v[u][k];
            }
        }
    }
}

struct WinSrcTable {          // neighbours from the rulebook table nbr[27][stride]
    const int32_t *__restrict__ nbr;
    int stride;
    __device__ __forceinline__ void operator()(int row, bool live, int (&v)[27]) const {
#pragma unroll
        for (int k = 0; k < 27; ++k) v[k] = live ? nbr[(size_t)k * stride + row] : -1;
    }
};

struct WinSrcColmap {         // neighbours from the level's column map: nine column lookups serve the 27 offsets (colmap.hip: cm_subm_kernel)
    const int4 *__restrict__ idx;
    const uint2 *__restrict__ cw;
    const uint4 *__restrict__ cr;
    int D, H, W, P, ncol_cap, n;
    __device__ __forceinline__ void operator()(int row, bool live, int (&v)[27]) const {
        const int4 c = live ? idx[row] : make_int4(0, 0, 0, 0);
        uint2 w[9];
        u32 key[9];
        bool in[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int y = c.z + q / 3 - 1, x = c.w + q % 3 - 1;
            in[q] = live && y >= 0 && y < H && x >= 0 && x < W;
            key[q] = in[q] ? bev_key(c.x, y, x, H, P) : 0u;
            w[q] = cw[key[q] >> 5];
        }
        uint4 r[9];
        bool hit[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int col = in[q] ? cm_col(w[q], key[q], ncol_cap) : -1;
            hit[q] = col >= 0;
            r[q] = cr[hit[q] ? col : 0];
        }
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int z = c.y + dz - 1;
                int rw = -1;
                if (hit[q] && z >= 0 && z < D) {
                    rw = cm_row((u64)r[q].x | ((u64)r[q].y << 32), (int)r[q].z, z);
                    if (rw >= n) rw = -1;
                }
                if (dz * 9 + q == 13 && live) rw = row;
                v[dz * 9 + q] = rw;
            }
    }
};

template <class C>
__global__ __launch_bounds__(256) void win_plan_kernel(const int32_t *__restrict__ nbr, int nbr_stride, int n_cap,
                                                       const int32_t *__restrict__ n_dev, char *__restrict__ plan_base, int ntiles_cap) {
    win_plan_body<C>(WinSrcTable{nbr, nbr_stride}, n_cap, n_dev, plan_base, ntiles_cap, nullptr, 0);
}

template <class C>
__global__ __launch_bounds__(256) void cm_win_plan_kernel(const int4 *__restrict__ idx, int n_cap, const int32_t *__restrict__ n_dev,
                                                          int D, int H, int W, int P, const uint2 *__restrict__ cw,
                                                          const uint4 *__restrict__ cr, int ncol_cap, char *__restrict__ plan_base,
                                                          int ntiles_cap, int32_t *__restrict__ nbr_out, int nbr_full) {
    win_plan_body<C>(WinSrcColmap{idx, cw, cr, D, H, W, P, ncol_cap, eff_rows(n_dev, n_cap)}, n_cap, n_dev, plan_base, ntiles_cap,
                     nbr_out, nbr_full);
}

// ---- shares: the tiles dealt to the WIN_GRID persistent workgroups in contiguous runs of EQUAL COST ------------------------
// A multi-pass tile (a run longer than the window: frame borders, walls) costs its workgroup about 2.5 ordinary tiles; with equal
// tile COUNTS the launch lasted as long as the few workgroups that own two of them (measured: 54 k clk against 31 k for the
// typical workgroup at 16 channels).  cost = 4 + PASS_COST (passes - 1), fitted per width; share w = tiles whose inclusive cost prefix lies in
// (total w / G, total (w + 1) / G].  Static (a function of the rulebook only): the BatchNorm partial rows stay reproducible.
// One workgroup.
template <class C>
__global__ __launch_bounds__(1024) void win_split_kernel(char *__restrict__ plan_base, int n_cap, const int32_t *__restrict__ n_dev,
                                                         int ntiles_cap, int grid) {
    __shared__ int wtot[2][16];
    constexpr int PFX_LDS = 8192;                         // the prefix also stays in LDS when it fits: the share boundaries are
    __shared__ int pfx_s[PFX_LDS];                        // binary searches -- log2(tiles) DEPENDENT reads each, L2 hits otherwise
    __shared__ int bnd[WIN_GRID_MAX + 1];
    __shared__ int bndw[WIN_WG_SHARES_MAX + 1];
    const int tid = threadIdx.x;
    const int n = eff_rows(n_dev, n_cap);
    const int nt = (n + C::T - 1) / C::T;
    const int4 *hdr = (const int4 *)(plan_base + win_hdr_off(ntiles_cap));
    int *prefix = (int *)(plan_base + win_prefix_off());
    // inclusive cost prefix in chunks of 1024 consecutive tiles, one tile per thread (a wave reads 64 consecutive headers; the
    // next chunk's header is requested before this chunk is scanned): wave scans + the 16 wave totals, ONE barrier per chunk.
    // (Until round 6: a contiguous range of tiles per thread -- 32-byte-strided loads, two serial passes over the range and a
    //  20-barrier block scan: 43 us at B = 32 for 18 k tiles; integer sums, so the prefix is the same.)
    const int wv = tid >> 6;
    int carry = 0;
    int pz = tid < nt ? hdr[(size_t)tid * 2 + 1].z : 1;
    for (int c0 = 0, it = 0; c0 < nt; c0 += 1024, ++it) {
        const int t = c0 + tid;
        const int cost = t < nt ? 4 + C::PASS_COST * (pz - 1) : 0;
        if (t + 1024 < nt) pz = hdr[(size_t)(t + 1024) * 2 + 1].z;
        const int inc = wave_inclusive_scan(cost);
        if ((tid & 63) == 63) wtot[it & 1][wv] = inc;
        __syncthreads();                                  // (double-buffered totals: the next chunk writes the other half)
        int off = carry, tot = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int s = wtot[it & 1][i];
            off += i < wv ? s : 0;
            tot += s;
        }
        if (t < nt) {
            prefix[t] = off + inc;
            if (t < PFX_LDS) pfx_s[t] = off + inc;
        }
        carry += tot;
    }
    const bool in_lds = nt <= PFX_LDS;
    const long long total = carry;
    __threadfence_block();
    __syncthreads();
    const int NSH = grid / C::CQN;                    // shares of the forward / data-gradient kernel (CQN workgroups each)
    for (int j = tid; j <= NSH; j += 1024) {
        const long long target = total * j / NSH;
        int lo = 0, hi = nt;                          // number of tiles with prefix <= target
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((long long)(in_lds ? pfx_s[mid] : prefix[mid]) <= target) lo = mid + 1; else hi = mid;
        }
        bnd[j] = j == NSH ? nt : lo;
    }
    for (int j = tid; j <= C::WG_SHARES; j += 1024) {
        const long long target = total * j / C::WG_SHARES;
        int lo = 0, hi = nt;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((long long)(in_lds ? pfx_s[mid] : prefix[mid]) <= target) lo = mid + 1; else hi = mid;
        }
        bndw[j] = j == C::WG_SHARES ? nt : lo;
    }
    __syncthreads();
    for (int w = tid; w < NSH; w += 1024) {
        const int tb = bnd[w], te = bnd[w + 1];
        int4 *e = (int4 *)(plan_base + (size_t)w * WIN_ENTRY_BYTES);
        const int4 z = make_int4(0, 0, 0, 0);
        e[0] = make_int4(tb, te, 0, 0);
        e[1] = tb < te ? hdr[(size_t)tb * 2] : z;
        e[2] = tb < te ? hdr[(size_t)tb * 2 + 1] : z;
        e[3] = z;
    }
    for (int w = tid; w < C::WG_SHARES; w += 1024)
        ((int2 *)(plan_base + win_wshare_off()))[w] = make_int2(bndw[w], bndw[w + 1]);
}

// ---- weight pack: every wave's slice contiguous, MFMA 32x32x16 A-operand order --------------------------------------
// packed[(((cb * NOQ + oq) * OPW + j) * KS + ks) * 64 + lane][e] = W_k[c_out = 32 cb + lane % 32][c_in = 16 ks + 8 (lane / 32) + e],
// k = oq * OPW + j (zeros for k >= 27).  mode 0: forward, weight [c_out][K][c_in]; mode 1: data gradient -- the contraction runs
// over the forward's OUTPUT channels and the k-flipped rulebook view, dx[i] = sum_k dy[nbr[26 - k][i]] W_k^T = sum_k' dy[nbr[k'][i]]
// W_{26-k'}^T: the flip goes into the pack, W_k[co][ci] := weight[ci][26 - k][co], and the launch reads the table as it is.
// cin_real < CIN (mode 0 only): the weight has cin_real input channels, the layer runs on rows zero-padded to CIN channels
// (conv_input of the backbones, 5 -> 16: spconv_backbone.py:191-195) -- the missing channels are packed as zeros.
template <class C>
__device__ __forceinline__ void win_pack_one(const float *__restrict__ w, int mode, size_t e, unsigned short *out, int cin_real = C::CIN) {
    const int j8 = (int)(e & 7), lane = (int)((e >> 3) & 63);
    size_t t = e >> 9;
    const int ks = (int)(t % C::KS);
    t /= C::KS;
    const int j = (int)(t % C::OPW);
    t /= C::OPW;
    const int oq = (int)(t % C::NOQ), cb = (int)(t / C::NOQ);          // cb < NCBP: (workgroup quarter, block) in the kernel's order
    const int k = oq * C::OPW + j;
    const int co = 32 * cb + (lane & 31), ci = 16 * ks + 8 * (lane >> 5) + j8;
    float v = 0.0f;
    if (k < 27 && co < C::COUT)
        v = mode == 0 ? (ci < cin_real ? w[((size_t)co * 27 + k) * cin_real + ci] : 0.0f) : w[((size_t)ci * 27 + (26 - k)) * C::COUT + co];
    out[e] = f32_to_bf16_bits(v);
}
template <class C>
constexpr size_t win_pack_elems() { return (size_t)C::NCBP * C::NOQ * C::OPW * C::KS * 512; }

__device__ __forceinline__ void win_pack_any(const float *__restrict__ w, int cin, int mode, size_t e, unsigned short *out, int cin_real) {
    if (cin_real <= 0 || cin_real > cin) cin_real = cin;
    if (cin == 64) {
        if (e < win_pack_elems<Win64>()) win_pack_one<Win64>(w, mode, e, out, cin_real);
    } else if (cin == 32) {
        if (e < win_pack_elems<Win32>()) win_pack_one<Win32>(w, mode, e, out, cin_real);
#ifdef PCD_EXPERIMENTS
    } else if (cin == 128) {
        if (e < win_pack_elems<Win128>()) win_pack_one<Win128>(w, mode, e, out);
#endif
    } else {
        if (e < win_pack_elems<Win16>()) win_pack_one<Win16>(w, mode, e, out, cin_real);
    }
}
__global__ __launch_bounds__(256) void win_pack_kernel(const float *__restrict__ w, int cin, int mode, unsigned short *out, int cin_real) {
    win_pack_any(w, cin, mode, (size_t)blockIdx.x * 256 + threadIdx.x, out, cin_real);
}

// table[i] = {weight ptr, packed ptr, c (= c_out: the kernel configuration), mode, first block, c_in of the weight (0 = c), 0, 0}
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
    win_pack_any(w, cin, mode, e, out, (int)row[5]);
}

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
unsigned long long *g_win_trace = nullptr;     // profiling aid, NULL in production (pcd_subm_window_set_trace)

// ---- the kernel ------------------------------------------------------------------------------------------------------
struct WinPlan {                 // scalars only (an array member sent the struct to scratch memory)
    int lo0, lo1, lo2, n0, n1, n2, passes;
    __device__ __forceinline__ int lo(int g) const { return g == 0 ? lo0 : g == 1 ? lo1 : lo2; }
    __device__ __forceinline__ int cnt(int g) const { return g == 0 ? n0 : g == 1 ? n1 : n2; }
};

template <class C>
__global__ __launch_bounds__(C::THREADS, C::NW == 8 ? 1 : 2) void subm_win_kernel(
    const unsigned short *__restrict__ x, const uint4 *__restrict__ wp, const float *__restrict__ bias,
    const int32_t *__restrict__ nbr, int nbr_stride, int n_cap, const int32_t *__restrict__ n_dev,
    const int4 *__restrict__ plan_g, unsigned short *__restrict__ y, unsigned x_bytes,
    const unsigned short *__restrict__ addend, BnRed bn, int dbg, unsigned long long *trace, float *__restrict__ y_f32) {
    __builtin_amdgcn_s_setprio(3);       // main-chain kernel (see spconv.hip: PCD_MAIN_PRIO)
    // profiling aid (pcd_subm_window_set_trace): shader-clock stamps of workgroup 0 / wave 0 at the phase boundaries of its tiles
    int trace_at = 0;
    auto stamp = [&]() {
        if (trace && blockIdx.x == 0 && threadIdx.x == 0 && trace_at < 256) trace[trace_at++] = __builtin_readcyclecounter();
    };
    constexpr int T = C::T, R = C::R, ROWB = C::ROWB, COUT = C::COUT, COUTW = C::COUTW, OPW = C::OPW, KS = C::KS, RBW = C::RBW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stamp();                                 // (trace slot 0: kernel entry)
    if (trace && threadIdx.x == 0 && blockIdx.x < 256) trace[256 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();   // (per-workgroup entry time, 100 MHz)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave8 % C::NCB, rg = (wave8 / C::NCB) % C::NRG, oq = wave8 / (C::NCB * C::NRG);
    const int n = eff_rows(n_dev, n_cap);
    // tiles of this workgroup: share w of win_split_kernel, w in XCD-major order (workgroup b runs on XCD b % 8 -- speed only):
    // every XCD gets a contiguous run of tiles -- neighbouring tiles' windows overlap, they share the XCD's L2
    const int ntiles_cap = (n_cap + T - 1) / T;
    const int wxm = (blockIdx.x & 7) * ((int)gridDim.x >> 3) + (blockIdx.x >> 3);          // XCD-major index
    const int cq = wxm % C::CQN;                         // this workgroup's columns of y: [cq COUTW, + COUTW) (consecutive wxm: same XCD)
    const int4 *entry = (const int4 *)((const char *)plan_g + (size_t)(wxm / C::CQN) * WIN_ENTRY_BYTES);
    const int4 ent0 = entry[0], ent1 = entry[1], ent2 = entry[2];
    const int t_begin = __builtin_amdgcn_readfirstlane(ent0.x), t_end = __builtin_amdgcn_readfirstlane(ent0.y);
    const int4 *hdr_g = (const int4 *)((const char *)plan_g + win_hdr_off(ntiles_cap));

    float *cols = (float *)(smem + C::COLS);
    if (t_begin >= t_end) {              // no tile: the BatchNorm row of this workgroup is zero
        if (bn.mode) bnred_publish(bn, blockIdx.x, COUT, [](int) { return 0.0f; }, (int)gridDim.x);
        return;
    }

    // this wave's weights: OPW offsets x KS steps, 4 VGPRs each, resident for the whole launch.  Where several waves hold the SAME
    // slice (NRG > 1: 8 copies at 16 channels, 4 at 32) the packed weights go to LDS once (into window buffer 1, free until the first
    // prefetch) and the waves fill their registers from there: 27 / 56 one-KiB instructions through the CU's texture-address unit
    // instead of 216 / 224 -- a third of a 16-channel launch's DMA work otherwise, all of it ahead of the first tile.
    bf16x8 wreg[OPW][KS];
    constexpr int NWP = C::NCB * C::NOQ * OPW * KS;      // KiB of packed weights
    // (several waves hold the same slice: the packed weights go to LDS once per workgroup -- where they fit window buffer 1; the
    //  4-wave 32-channel configuration's 56 KiB do not: its waves load their slices from global memory)
    constexpr bool STAGEW = C::NRG > 1 && NWP * 1024 <= C::WINB;
    if (!STAGEW) {
        const uint4 *wsrc = wp + (size_t)(((cq * C::NCB + cb) * C::NOQ + oq) * OPW) * KS * 64 + lane;
#pragma unroll
        for (int j = 0; j < OPW; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wreg[j][ks] = __builtin_bit_cast(bf16x8, wsrc[(j * KS + ks) * 64]);
    }
    // bias and (BatchNorm mode 2) mean of the epilogue: LDS copies (zeros without a bias) in the area of the final BatchNorm row
    for (int e = tid; e < 2 * COUTW; e += C::THREADS)
        cols[e] = e < COUTW ? (bias ? bias[cq * COUTW + e] : 0.0f) : (bn.mode == 2 ? bn.mean[cq * COUTW + e - COUTW] : 0.0f);
    // the zero rows in front of both windows
    auto clear_zero1 = [&]() {
        for (int e = tid; e < C::Z0 * ROWB / 4; e += C::THREADS) ((int *)(smem + C::ZERO1))[e] = 0;
    };
    for (int e = tid; e < C::Z0 * ROWB / 4; e += C::THREADS) ((int *)(smem + C::ROWBASE))[e] = 0;
    clear_zero1();

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
    auto make_plan = [](const int4 a, const int4 b) {
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
    const u32x4 tdma = win_rsrc((const char *)plan_g + win_tab_off(ntiles_cap), (unsigned)((size_t)ntiles_cap * C::TABB));
    typedef __attribute__((address_space(3))) char *lds_ptr_t;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_ptr_t)smem);      // LDS address of smem[0]
    if (STAGEW) {
        const u32x4 wdma = win_rsrc(wp, (unsigned)(NWP * 1024));
        for (int pc = wave8; pc < NWP; pc += C::NW)
            win_glds16(wdma, lds0 + (unsigned)(C::WIN1 + pc * 1024), (unsigned)(pc * 1024) + (unsigned)lane * 16u);
    }

    // The prefetch of a tile is NSLOT VMEM instructions per wave, issued one at a time BETWEEN the MFMA steps of the previous
    // tile (a 1-KiB instruction occupies the CU's texture-address unit for ~34 clk and the in-order wave behind it: issued as one
    // burst after the barrier, the 40-80 instructions of a tile cost every wave 1.8 k clk before its first MFMA and skewed the
    // waves by another 3 k clk -- measured with pcd_subm_window_set_trace).
    //   slots 0 .. 3 SPR - 1: window pieces -- run g = s / SPR, 1-KiB instruction wave8 + 8 (s % SPR) of the run
    //   slots 3 SPR ..      : 1-KiB piece wave8 + 8 (s - 3 SPR) of the tile's table (NTABI pieces)
    constexpr int NSLOT = C::NSLOT, SPR = C::SPR;
    // The lane part of every DMA address is computed once: a wave's window pieces all start a multiple of 16 rows (one swizzle
    // period) + (wave8 RPI) % 16 into their buffer; a table piece is 1 KiB of the tile's table as it lies in the plan.
    const unsigned wlane = (unsigned)(lane / C::S) * (unsigned)ROWB +
                           (((unsigned)(lane % C::S) ^ C::swz((unsigned)(lane / C::S + (wave8 * C::RPI) % 16))) << 4);
    auto issue_slot = [&](const WinPlan p, int tile, int buf, int pass, int slot) {
        if (dbg & 1) return;
        if (slot < 3 * SPR) {
            const int g = slot / SPR, i = wave8 + C::NW * (slot % SPR);
            const int cnt = min(R, p.cnt(g) - pass * R);
            if (i * C::RPI < cnt) {
                // (whole pieces: the rows past the end of the run are rows of x nobody refers to, or lie beyond the buffer)
                const int wrow = (buf ? C::WIN1ROW : C::Z0) + g * R + i * C::RPI;          // first LDS row of the piece (wave-uniform)
                const unsigned sbase = (unsigned)(p.lo(g) + pass * R + i * C::RPI) * (unsigned)ROWB;
                win_glds16(xdma, lds0 + (unsigned)(C::ROWBASE + wrow * ROWB), sbase + wlane);
            }
        } else if (wave8 + C::NW * (slot - 3 * SPR) < C::NTABI && pass == 0) {
            const int piece = wave8 + C::NW * (slot - 3 * SPR);
            win_glds16(tdma, lds0 + (unsigned)(C::TAB0 + buf * C::TABB + piece * 1024),
                       (unsigned)((dbg & 16) ? 0 : tile) * (unsigned)C::TABB + (unsigned)(piece * 1024) + (unsigned)lane * 16u);
            // (dbg & 16, ablation: every tile fetches tile 0's table -- always hot; wrong results, the time of a free table)
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
            const int e = tq + u * C::THREADS;
            const int k = e / T, r = e - k * T;
            const unsigned off = e < 27 * T ? ((unsigned)k * (unsigned)nbr_stride + (unsigned)(t * T + r)) * 4u : 0xFFFFFFF0u;
            nv[u] = __builtin_amdgcn_raw_buffer_load_b32(nrsrc, off, 0, 0);
        }
    };
    auto build_tab = [&](int t, const WinPlan p, const int (&nv)[C::NL], int buf, int pass) {
        unsigned short *tab = (unsigned short *)(smem + C::TAB0 + buf * C::TABB);
        const int tq = fresh(tid);
#pragma unroll
        for (int u = 0; u < C::NL; ++u) {
            const int e = tq + u * C::THREADS;
            const int k = e / T, r = e - k * T;
            const int g = (k / 3) % 3;
            const int lo = p.lo(g) + pass * R;
            const int v = nv[u];
            const unsigned rel = (unsigned)(v - lo);
            // (every loaded value is consumed unconditionally -- bitwise &, no short circuit: the compiler then places its own
            //  wait for the load here and does not carry "maybe still in flight" registers into the MFMA loop, where it would
            //  protect their reuse with an s_waitcnt vmcnt(0) that also waits for the prefetch of the next tile)
            const bool ok = (e < 27 * T) & (v >= 0) & (t * T + r < n) & (rel < (unsigned)R);
            const unsigned row = (unsigned)(C::Z0 + g * R) + rel;
            if (e < 27 * T) tab[C::tab_pos((unsigned)r, (unsigned)k)] = ok ? (unsigned short)(row * C::S + C::swz(row)) : (unsigned short)0;
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
        const unsigned half = (unsigned)lane >> 5;
        const unsigned boff = buf ? (unsigned)(C::BOFF * ROWB) : 0u;     // table entries are slots of buffer 0; (a multiple of 128)
        // a step = 4 MFMAs on 4 fragments: SO offsets are paired at narrow layers, a wide contraction (KS > 4) is cut into KH
        // parts of KSS steps (two register sets of 4 fragments either way)
        constexpr int SO = KS >= 4 ? 1 : 4 / KS;
        constexpr int KH = KS > 4 ? KS / 4 : 1, KSS = KS / KH;
        constexpr int NJ = (OPW + SO - 1) / SO;          // offset groups of this wave's slice
        constexpr int NSTEP = NJ * RBW * KH;
        constexpr int PF_FIRST = C::DIRECT ? 0 : NSTEP >= 10 ? 3 : 1;    // first step followed by prefetch slots (DIRECT: no barrier
                                                                          // to wait for -- as early as possible, the loop is short)
        constexpr int SPS = (NSLOT + (NSTEP - PF_FIRST) - 1) / (NSTEP - PF_FIRST);   // slots per step
        static_assert(NSTEP > PF_FIRST, "steps");
        bf16x8 fr[2][SO * KSS];
        // this lane's table entries (its row of every row block, the wave's slice of the offsets): SLP 16-byte reads per block
        u32 tq[RBW][C::SLD];
#pragma unroll
        for (int rbw = 0; rbw < RBW; ++rbw) {
            const unsigned r = (unsigned)((rg * RBW + rbw) * 32) + ((unsigned)lane & 31u);
            if (C::SLP > 0) {
#pragma unroll
                for (int c = 0; c < C::SLP; ++c) {
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(smem + C::TAB0 + buf * C::TABB + r * 64 +
                                                                      (((unsigned)(oq * C::SLP + c) ^ ((r >> 2) & 3u)) << 4));
                    tq[rbw][4 * c] = v[0]; tq[rbw][4 * c + 1] = v[1]; tq[rbw][4 * c + 2] = v[2]; tq[rbw][4 * c + 3] = v[3];
                }
            } else {             // SLICE == 4: half a piece (8 bytes) per slice
                const u32x2 v = *reinterpret_cast<const u32x2 *>(smem + C::TAB0 + buf * C::TABB + r * 64 +
                                                                  (((unsigned)(oq >> 1) ^ ((r >> 2) & 3u)) << 4) + (oq & 1) * 8);
                tq[rbw][0] = v[0]; tq[rbw][1] = v[1];
            }
        }
        auto fetch = [&](int step, bf16x8 (&dst)[SO * KSS]) {
            const int jg = step / (RBW * KH), rbw = (step / KH) % RBW, kh = step % KH;
#pragma unroll
            for (int o = 0; o < SO; ++o) {
                const int j = jg * SO + o;
                if (j >= OPW) continue;
                // entry = slot of the row's first 8 channels in buffer 0 (0 = the zero row in front of it); buffer 1 lies BOFF rows
                // (whole swizzle periods) further and has its own zero rows.  Slot of contraction step ks = (2 ks + half) ^
                // swz(row) = entry ^ half ^ 2 ks: one XOR with a constant per step.
                const u32 w = tq[rbw][j / 2];
                const unsigned e = (j & 1) ? w >> 16 : w & 0xffffu;
                const unsigned a0 = ((e ^ half) << 4) + boff;
#pragma unroll
                for (int ks = 0; ks < KSS; ++ks)
                    dst[o * KSS + ks] = *reinterpret_cast<const bf16x8 *>(rowbase + (a0 ^ ((unsigned)(kh * KSS + ks) << 5)));
            }
        };
        fetch(0, fr[0]);
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            if (step + 1 < NSTEP) fetch(step + 1, fr[(step + 1) & 1]);
            // (the scheduler, short of registers, sinks the reads back to one per MFMA unless told not to)
            __builtin_amdgcn_sched_barrier(0);
            const int jg = step / (RBW * KH), rbw = (step / KH) % RBW, kh = step % KH;
#pragma unroll
            for (int o = 0; o < SO; ++o)
#pragma unroll
                for (int ks = 0; ks < KSS; ++ks)
                    if (jg * SO + o < OPW)
                        acc[rbw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[jg * SO + o][kh * KSS + ks], fr[step & 1][o * KSS + ks],
                                                                           acc[rbw], 0, 0, 0);
            if (PF) {
                // B1, inside the loop: the prefetch overwrites the area the PREVIOUS tile's epilogue read its partial sums from, so
                // every wave must have left that epilogue -- but nothing before the first slot needs the barrier: the waves run
                // their first steps as they arrive and meet here (the skew of the epilogue is absorbed by MFMA work)
                // (DIRECT: no epilogue reads LDS, the barrier at the end of every tile covers the reuse of the buffers)
                if (!C::DIRECT && step == PF_FIRST - 1) {
                    WIN_BARRIER();
                    // (the previous tile's partial sums ran over the zero rows of buffer 1: clear them for the next tile)
                    if (C::REZERO && buf == 0) clear_zero1();
                }
#pragma unroll
                for (int q = 0; q < SPS; ++q)
                    if (step >= PF_FIRST && (step - PF_FIRST) * SPS + q < NSLOT) issue_slot(pn, tnext, buf ^ 1, 0, (step - PF_FIRST) * SPS + q);
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
        {   // prologue: tile chunk0 into buffer 0 -- its header came with the workgroup's entry (first chunk) or is read from the
            // plan here (wave-uniform loads): the DMAs start before the chunk's plans are staged in LDS
            if (C::REZERO) clear_zero1();
            const bool first = chunk0 == t_begin;
            const WinPlan p0 = make_plan(first ? ent1 : hdr_g[(size_t)chunk0 * 2], first ? ent2 : hdr_g[(size_t)chunk0 * 2 + 1]);
            for (int q = 0; q < NSLOT; ++q) issue_slot(p0, chunk0, 0, 0, q);
            for (int e = tid; e < (chunk1 - chunk0) * 2; e += C::THREADS) ((int4 *)(smem + C::PLAN))[e] = hdr_g[(size_t)chunk0 * 2 + e];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp();                         // (trace slot 1: weights + first window have landed)
            __syncthreads();
            if (STAGEW && first) {
                const char *wl = smem + C::WIN1 + ((cb * C::NOQ + oq) * OPW * KS) * 1024 + lane * 16;
#pragma unroll
                for (int j = 0; j < OPW; ++j)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) wreg[j][ks] = *reinterpret_cast<const bf16x8 *>(wl + (j * KS + ks) * 1024);
                __syncthreads();             // (the first prefetch overwrites the staging area)
            }
        }
        for (int t = chunk0; t < chunk1; ++t) {
            const int buf = (t - chunk0) & 1;
            const WinPlan p = get_plan(t, chunk0);
            // (window + table of tile t were published by the barriers of tile t - 1; B1 = "buffer buf ^ 1 is free" sits inside
            //  compute(), in front of the first prefetch slot)
            if (dbg & 64) {
                // COSTING ABLATION (option "win_dbg" bit 6; results are wrong with it): what a BatchNorm applied ON READ would
                // cost this launch -- one in-LDS pass relu(scale * x + shift) over the tile's three landed runs (scale / shift per
                // channel out of LDS, one bf16 rounding) + the barrier that publishes it, before the MFMA loop
                const float *aff = (const float *)(smem + C::COLS);            // [2 COUT] floats: stands in for scale | shift
                char *wbase = smem + C::ROWBASE + (buf ? C::WIN1ROW : C::Z0) * ROWB;
                for (int e = fresh(tid); e < C::WINROWS * C::S; e += C::THREADS) {
                    const unsigned row = (unsigned)e / C::S, slot = (unsigned)e % C::S;
                    const unsigned cg = slot ^ C::swz(row + (buf ? C::WIN1ROW : C::Z0));
                    uint4 v = *reinterpret_cast<uint4 *>(wbase + (size_t)e * 16);
                    const float4 s0 = *reinterpret_cast<const float4 *>(aff + cg * 8), s1 = *reinterpret_cast<const float4 *>(aff + cg * 8 + 4);
                    const float4 h0 = *reinterpret_cast<const float4 *>(aff + COUT + cg * 8), h1 = *reinterpret_cast<const float4 *>(aff + COUT + cg * 8 + 4);
                    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                    u32 w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float a = __uint_as_float(w4[j] << 16), b = __uint_as_float(w4[j] & 0xffff0000u);
                        a = fmaxf(a * sc[2 * j] + sh[2 * j], 0.0f);
                        b = fmaxf(b * sc[2 * j + 1] + sh[2 * j + 1], 0.0f);
                        w4[j] = __builtin_bit_cast(u32, __builtin_convertvector((f32x2){a, b}, bf16x2));
                    }
                    *reinterpret_cast<uint4 *>(wbase + (size_t)e * 16) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                }
                WIN_BARRIER();
            }
            stamp();
            const bool more = t + 1 < chunk1;
            // (the plan of the last tile is read twice rather than copied conditionally: a struct merged over a branch was
            //  kept in scratch memory)
            const WinPlan pn = get_plan(more ? t + 1 : t, chunk0);
#pragma unroll
            for (int rbw = 0; rbw < RBW; ++rbw)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[rbw][i] = 0.0f;
            // the epilogue's operands of this thread's (row, 8 channels).  (Rows beyond n read row n - 1: unconditional loads, no
            // select that would wait for them; the launch-uniform branches leave the registers undefined when the operand does
            // not exist -- they are only read under the same test.)
            uint4 av, xv, yv;
            int erow, ecg;
            auto load_operands = [&]() {
                const int te = fresh(tid);
                if (C::DIRECT) {         // the lane's row of the wave's block, channels 8 (lane / 32) .. + 7 (after the half swap below)
                    erow = rg * 32 + (te & 31);
                    ecg = (te & 63) >> 5;
                } else {
                    // (NEPI < 512 threads have a row: the others repeat the roles of the first NEPI -- loads of valid addresses,
                    //  nothing stored)
                    erow = (te % C::NEPI) / C::CG;
                    ecg = te % C::CG;
                }
                const int orow = t * T + erow;
                const size_t lelem = (size_t)(orow < n ? orow : n - 1) * COUT + cq * COUTW + ecg * 8;
                if (addend) av = *reinterpret_cast<const uint4 *>(addend + lelem);
                if (bn.mode == 2) {
                    xv = *reinterpret_cast<const uint4 *>(bn.x + lelem);
                    if (bn.relu) yv = *reinterpret_cast<const uint4 *>(bn.y + lelem);
                }
            };
            // v[8] = the fp32 sums of (erow, channels 8 ecg ..): + bias (+ addend), one rounding, 16 bytes out, BatchNorm sums
            auto finish = [&](float (&v)[8]) {
                const int orow = t * T + erow;
                const bool olive = orow < n;
                const unsigned ooff = olive ? (unsigned)((size_t)orow * COUT + cq * COUTW + ecg * 8) * 2u : 0xFFFFFFF0u;
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
                if (y_f32 && olive) {      // (parity aid: the fp32 sums the rounding below starts from, pcd_sparse_conv_subm_window_f32)
                    float *d = y_f32 + (size_t)orow * COUT + cq * COUTW + ecg * 8;
                    *reinterpret_cast<f32x4 *>(d) = (f32x4){v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4 *>(d + 4) = (f32x4){v[4], v[5], v[6], v[7]};
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
                        const f32x4 m0 = *reinterpret_cast<const f32x4 *>(cols + COUTW + ecg * 8),
                                    m1 = *reinterpret_cast<const f32x4 *>(cols + COUTW + ecg * 8 + 4);
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
            };
            if (C::DIRECT) load_operands();              // (in flight across the MFMA loop)
            stamp();
            if (more) compute(buf, PF1{}, pn, t + 1); else compute(buf, PF0{}, pn, t);
            stamp();
            for (int pass = 1; pass < p.passes; ++pass) {
                // a run longer than the window: next chunk of every run into the same buffer (slow path, rare)
                int nv2[C::NL];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                WIN_BARRIER();
                for (int q = 0; q < 3 * SPR; ++q) issue_slot(p, t, buf, pass, q);
                load_nbr(t, nv2);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                build_tab(t, p, nv2, buf, pass);
                WIN_BARRIER();
                compute(buf, PF0{}, p, t);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // window + table of t + 1 landed (this wave's part)
            stamp();
            if (C::DIRECT) {
                // One barrier per tile: everybody's share of tile t + 1 has landed and everybody is done reading buffer buf (the
                // prefetch of tile t + 2 may overwrite it).  The wave's 32 x 32 block holds, per lane (row n = lane & 31, h =
                // lane >> 5), channels 4 h .. + 3 in registers 0-3 and 8 + 4 h .. + 3 in 4-7 (16 live channels): the halves
                // exchange registers 4-7 / 0-3 (v_permlane32_swap), after which lane (n, h) holds channels 8 h .. 8 h + 7.
                WIN_BARRIER();
                stamp();
                stamp();
                if (!(dbg & 8)) {
                    static_assert(!C::DIRECT || (RBW == 1 && COUT == 16), "direct epilogue: one row block per wave, 16 channels");
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0][e]), __float_as_uint(acc[0][4 + e]), false, false);
                        v[e] = __uint_as_float(r[0]);
                        v[4 + e] = __uint_as_float(r[1]);
                    }
                    finish(v);
                }
                stamp();
                continue;
            }
            load_operands();                             // in flight across the two barriers below
            WIN_BARRIER();                               // B2: every wave is done reading window buf -> it becomes the reduction area
            stamp();
            char *red = smem + (buf ? C::RED1 : C::WIN0);
            if (!(dbg & 8)) {   // partial sums: lane (n = lane & 31, h = lane >> 5) holds rows (i & 3) + 8 (i >> 2) + 4 h of the 32 x 32 block
                const int h = lane >> 5;
#pragma unroll
                for (int rbw = 0; rbw < RBW; ++rbw) {
                    const int row = (rg * RBW + rbw) * 32 + (lane & 31);
                    char *dst = red + (size_t)(oq * T + row) * C::REDSTRIDE + (cb * 32 + 4 * h) * 4;
#pragma unroll
                    for (int q4 = 0; q4 < C::QN; ++q4)
                        *reinterpret_cast<f32x4 *>(dst + q4 * 32) =
                            (f32x4){acc[rbw][q4 * 4], acc[rbw][q4 * 4 + 1], acc[rbw][q4 * 4 + 2], acc[rbw][q4 * 4 + 3]};
                }
            }
            WIN_BARRIER();                               // B3
            stamp();
            if (!(dbg & 8) && (C::NEPI == C::THREADS || fresh(tid) < C::NEPI)) {   // tile epilogue: the slices summed in order
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
                finish(v);
            }
            stamp();
        }
    }
    if (bn.mode) {
        // the workgroup's BatchNorm row: column sums over its threads' rows in a fixed order -- rows of a wave by lane
        // exchanges (lanes that differ in the row bits), then the 8 waves through LDS
        // (a lane's 8-channel group: lane % CG, rows in the higher lane bits; DIRECT: lane / 32, rows in the lower five)
#pragma unroll
        for (int d = C::DIRECT ? 1 : C::CG; d < (C::DIRECT ? 32 : 64); d <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bs[e] += __shfl_xor(bs[e], d, 64);
                bq[e] += __shfl_xor(bq[e], d, 64);
            }
        __syncthreads();
        float *stage = (float *)(smem + C::WIN0);                        // [8 waves][CG][16]
        if (C::DIRECT ? (lane & 31) == 0 : lane < C::CG) {
            const int cg = C::DIRECT ? lane >> 5 : lane;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                stage[(wave8 * C::CG + cg) * 16 + e] = bs[e];
                stage[(wave8 * C::CG + cg) * 16 + 8 + e] = bq[e];
            }
        }
        __syncthreads();
        for (int e = tid; e < 2 * COUTW; e += C::THREADS) {
            const int which = e / COUTW, c = e % COUTW;
            float s = 0.0f;
            for (int w = 0; w < C::NW; ++w) s += stage[(w * C::CG + c / 8) * 16 + which * 8 + (c & 7)];
            if (which == 1 && bn.mode == 2) s *= bn.invstd[cq * COUTW + c];
            cols[e] = s;
        }
        __syncthreads();
        // (CQN > 1: the workgroup's row of the partial matrix carries its COUTW columns, zeros elsewhere -- the column sums over
        //  the rows are the same)
        bnred_publish(bn, blockIdx.x, COUT, [&](int e) {
            const int which = e / COUT, c = e % COUT - cq * COUTW;
            return (c >= 0 && c < COUTW) ? cols[which * COUTW + c] : 0.0f;
        }, (int)gridDim.x);
    }
    stamp();                                 // (last trace slot: kernel exit)
    if (trace && threadIdx.x == 0 && blockIdx.x < 256) trace[512 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}


template <class C>
static int launch_win(const void *x, int n_rows, const void *wp, const float *bias, const int32_t *nbr, int nbr_stride,
                      const int32_t *n_dev, const void *plan, void *y, const void *addend,
                      const PcdBnReduce *bnr, hipStream_t st, float *y_f32 = nullptr) {
    BnRed bn;
    const int grid = win_grid<C>();
    if (int rc = make_bnred(bnr, PCD_BF16, C::COUT, grid, &bn)) return rc;
    if ((double)n_rows * C::ROWB >= 4294967040.0) return PCD_ERR_UNSUPPORTED;
    auto k = subm_win_kernel<C>;
    // (set per call: the attribute is per device, the call idempotent)
    if (hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
        return PCD_ERR_LAUNCH;
    k<<<grid, C::THREADS, C::LDS_BYTES, st>>>((const unsigned short *)x, (const uint4 *)wp, bias, nbr, nbr_stride,
                                                    n_rows, n_dev, (const int4 *)plan, (unsigned short *)y,
                                                    (unsigned)((size_t)n_rows * C::ROWB), (const unsigned short *)addend, bn,
                                                    pcd_opt(PCD_OPT_WIN_DBG), g_win_trace, y_f32);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}


// ---- weight gradient over the same tiles ------------------------------------------------------------------------------
// dW_k[ci][co] = sum over rows r of x[nbr[k][r]][ci] dy[r][co], k = 0..26, for the SubM layers the window kernel serves.
// The 27 accumulators of a layer do not fit one workgroup at 64 channels, and the offsets of one run (dy = g - 1: the nine k
// with (k / 3) % 3 == g) only need THAT run in LDS: a workgroup = (share of the tiles, run g) keeps nine C x C accumulators in
// registers for its whole share, streams per tile the run's window (the forward's DMA, one run of three), the tile's rows of
// dy and the tile's slot table through double-buffered LDS, and builds both MFMA operands with the transposing LDS read
// (ds_read_b64_tr_b16: the contraction index = tile row is the slow axis of both row-major operands), x rows found through
// the table.  No pair lists, no gathers from global memory, dy read three times (once per run), x once.
//   waves: (block group, row-step split): C = 16: 8 waves split the 32-row steps of a tile; 32: 4 blocks x 2; 64: 8 groups of
//   2 blocks.  16x16x32 MFMA, A = x^T (ci x rows), B = dy (rows x co), as in wgrad_kernel (spconv.hip).
// Every (share, run) workgroup writes its nine offsets of slab[share][co][k][ci] (f32): WIN_WG_SHARES slabs that the batched
// fixed-order reduction (pcd_sparse_conv_wgrad_reduce_batched, `splits` = WIN_WG_SHARES) sums: deterministic, no atomics.
// Workgroup b -> XCD b % 8 takes shares 10 (b % 8) .. + 9, all three runs of a share on the same XCD (they read the same
// rows of dy and the same table through its L2); 2 of the 32 workgroups of an XCD idle.
typedef __attribute__((ext_vector_type(8))) short s16x8;
template <class C>
struct WgCfg {
    static constexpr int CIN = C::CIN, T = C::T, R = C::R, ROWB = C::ROWB, S = C::S;
    static constexpr int MB = CIN / 16, NB = CIN / 16;
    // wave = (mb, oh, rs): a 16-channel block of c_in, one of NOH parts of the run's nine offsets, one of RS interleaved subsets of
    // the tile's 32-row steps; it owns ALL output-channel blocks of those: every x operand is read from LDS by one wave only
    static constexpr int NOH = MB >= 4 ? 2 : 1;
    static constexpr int NO = (9 + NOH - 1) / NOH;                 // offsets per wave
    static constexpr int NG = MB * NOH;
    static constexpr int RS = C::NW / NG;
    static constexpr int NSTEP = T / 32;
    static_assert(NG * RS == C::NW && NSTEP % RS == 0, "wave roles");
    static constexpr int SPRW = (R / C::RPI + C::NW - 1) / C::NW;  // window DMA instructions per wave and tile
    static_assert(T * ROWB == C::NW * 1024, "dy tile: one 1-KiB DMA instruction per wave");
    // A tile's compute is short (~1 k clk) against the latency of its DMAs: a ring of NBUF stages, filled NBUF - 1 tiles ahead.
    // LDS: [Z0 zero rows][NBUF windows of R rows][NBUF dy tiles][NBUF tables][plans]
    static constexpr int NBUF = 3;
    static constexpr int WIN0 = C::Z0 * ROWB;
    static constexpr int DY0 = (C::Z0 + NBUF * R) * ROWB;
    static constexpr int TAB0 = DY0 + NBUF * T * ROWB;
    static constexpr int PLAN = TAB0 + NBUF * C::TABB;
    static constexpr int LDS_BYTES = PLAN + WIN_PLAN_CAP * 32;
    static constexpr int REDB = (RS - 1) * NG * NO * NB * 1024;    // final cross-wave sums, over the windows
    static_assert(REDB <= PLAN && LDS_BYTES <= (C::NW == 8 ? 160 : 80) * 1024, "LDS");
    static constexpr int NL = (9 * T + C::THREADS - 1) / C::THREADS;
};

template <class C, int G>
__device__ __forceinline__ void wgrad_win_body(const unsigned short *__restrict__ x, const unsigned short *__restrict__ dy,
                                               const int32_t *__restrict__ nbr, int nbr_stride, int n_cap, int n,
                                               const char *__restrict__ plan_g, float *__restrict__ slab, unsigned x_bytes,
                                               unsigned dy_bytes, int share, char *smem, unsigned long long *trace) {
    using W = WgCfg<C>;
    int trace_at = 0;
    auto stamp = [&]() {
        if (trace && blockIdx.x == 0 && threadIdx.x == 0 && trace_at < 256) trace[trace_at++] = __builtin_readcyclecounter();
    };
    stamp();
    const unsigned long long t_entry = trace ? __builtin_amdgcn_s_memrealtime() : 0ull;     // (100 MHz, the same clock on every CU)
    constexpr int T = C::T, R = C::R, ROWB = C::ROWB, S = C::S, CIN = C::CIN, NB = W::NB, NO = W::NO;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 % W::NG, rs = wave8 / W::NG;
    const int mb = grp % W::MB, oh = grp / W::MB;
    const int ntiles_cap = (n_cap + T - 1) / T;
    const int2 sh = ((const int2 *)(plan_g + win_wshare_off()))[share];
    const int t_begin = __builtin_amdgcn_readfirstlane(sh.x), t_end = __builtin_amdgcn_readfirstlane(sh.y);
    const int4 *hdr_g = (const int4 *)(plan_g + win_hdr_off(ntiles_cap));

    f32x4 acc[NO][NB];
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[o][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (tid < C::Z0 * ROWB / 4) ((int *)smem)[tid] = 0;           // the zero rows (entry 0)
    // (dy: the REAL rows only -- with a capacity above the row count the rows behind it hold whatever the allocation held, and
    //  0 x NaN is NaN: beyond num_records the DMA delivers zeros)
    (void)dy_bytes;
    const u32x4 xdma = win_rsrc(x, x_bytes), ydma = win_rsrc(dy, (unsigned)n * (unsigned)ROWB);
    const u32x4 tdma = win_rsrc(plan_g + win_tab_off(ntiles_cap), (unsigned)((size_t)ntiles_cap * C::TABB));
    typedef __attribute__((address_space(3))) char *lds_ptr_t;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_ptr_t)smem);
    // lane part of the row DMAs (window pieces and the dy piece of a wave start (wave8 RPI) % 16 rows into a swizzle period)
    const unsigned wlane = (unsigned)(lane / S) * (unsigned)ROWB +
                           (((unsigned)(lane % S) ^ C::swz((unsigned)(lane / S + (wave8 * C::RPI) % 16))) << 4);
    // MFMA operand roles of the lane (as in wgrad_kernel): contraction index 8 g4 + j <-> tile row (j < 4 ? 4 g4 + j : 16 + 4 g4
    // + j - 4) of the 32-row step; the lane supplies the address of 4 bf16 (8 bytes) of row trow, columns 4 (t & 3) .. + 3 of the
    // 16-channel block
    const int g4 = lane >> 4, t16 = lane & 15;
    const int trow = 4 * g4 + (t16 >> 2);
    const unsigned sub = (unsigned)(t16 & 1) * 8u;
    const unsigned csel = (unsigned)(2 * mb + ((t16 & 3) >> 1));               // 16-byte slot of the lane's x columns

    struct Run { int lo, cnt; };
    auto run_of = [&](const int4 a, const int4 b) {
        Run r;
        r.lo = __builtin_amdgcn_readfirstlane(G == 0 ? a.x : G == 1 ? a.z : b.x);
        r.cnt = __builtin_amdgcn_readfirstlane(G == 0 ? a.y : G == 1 ? a.w : b.y);
        return r;
    };
    const int4 *plan_s = (const int4 *)(smem + W::PLAN);
    // A stage's DMAs are NSL instructions per wave: slots 0 .. SPRW - 1 window pieces, SPRW the wave's dy piece, then its table
    // pieces.  issue_slot returns 1 if the slot had an instruction (wave-uniform: the counted waits below need the sum).
    constexpr int NSL = W::SPRW + 1 + C::TSL;
    auto issue_slot = [&](const Run p, int tile, int buf, int pass, int slot) -> int {
        if (slot < W::SPRW) {
            const int i = wave8 + C::NW * slot;
            if (i * C::RPI >= min(R, p.cnt - pass * R)) return 0;
            win_glds16(xdma, lds0 + (unsigned)(W::WIN0 + (buf * R + i * C::RPI) * ROWB),
                       (unsigned)(p.lo + pass * R + i * C::RPI) * (unsigned)ROWB + wlane);
            return 1;
        }
        if (pass != 0) return 0;
        if (slot == W::SPRW) {
            win_glds16(ydma, lds0 + (unsigned)(W::DY0 + buf * T * ROWB + wave8 * 1024),
                       (unsigned)(tile * T + wave8 * C::RPI) * (unsigned)ROWB + wlane);
            return 1;
        }
        const int piece = wave8 + C::NW * (slot - W::SPRW - 1);
        if (piece >= C::NTABI) return 0;
        win_glds16(tdma, lds0 + (unsigned)(W::TAB0 + buf * C::TABB + piece * 1024),
                   (unsigned)tile * (unsigned)C::TABB + (unsigned)(piece * 1024) + (unsigned)lane * 16u);
        return 1;
    };
    auto issue = [&](const Run p, int tile, int buf, int pass) {
        int issued = 0;
#pragma unroll
        for (int q = 0; q < NSL; ++q) issued += issue_slot(p, tile, buf, pass, q);
        return issued;
    };
    // s_waitcnt vmcnt(k), k wave-uniform: at most k of this wave's vector-memory operations still in flight
    auto wait_vm = [](int k) {
        switch (k) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        }
    };
    // OH (the wave's part of the offsets) is a template argument: the table entry of an offset is picked by compile-time indices
    // (pf: the DMAs of tile `tn` into stage `bn` are issued one at a time between the offsets of the wave's first step -- as one
    //  burst at the top of the tile the 8 waves queued ~40 instructions at the CU's texture-address unit, ~1.3 k clk in which
    //  nobody computed; *pf_count receives their number)
    auto compute_oh = [&](int buf, auto oh_tag, bool pf, const Run pn, int tn, int bn, int *pf_count) {
        constexpr int OH = decltype(oh_tag)::value;
        const char *tab = smem + W::TAB0 + buf * C::TABB;
        const char *dyb = smem + W::DY0 + buf * T * ROWB;
        // table entries are slots of the FORWARD's buffer 0 (row Z0 + G R + rel); here the run lies at row Z0 + buf R + rel
        const unsigned adj = (unsigned)((buf - G) * R * S);
        const unsigned q4 = (unsigned)(t16 & 3);
        for (int s = rs; s < W::NSTEP; s += W::RS) {
            const unsigned r_lo = (unsigned)(s * 32 + trow), r_hi = r_lo + 16u;
            // the four lanes of a quad supply addresses in the same two tile rows: each reads ONE 16-byte piece of the row's 64
            // table bytes, entries are handed round the quad by DPP (a quarter of the LDS traffic of four private copies)
            const u32x4 tl = *reinterpret_cast<const u32x4 *>(tab + r_lo * 64 + ((q4 ^ ((r_lo >> 2) & 3u)) << 4));
            const u32x4 th = *reinterpret_cast<const u32x4 *>(tab + r_hi * 64 + ((q4 ^ ((r_hi >> 2) & 3u)) << 4));
            bf16x8 bf[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const unsigned dsel = (unsigned)(2 * j) + (q4 >> 1);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(
                    dyb + r_lo * ROWB + ((dsel ^ C::swz(r_lo)) << 4) + sub));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(
                    dyb + r_hi * ROWB + ((dsel ^ C::swz(r_hi)) << 4) + sub));
                bf[j] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
            bf16x8 af[NO];
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const int og = OH * NO + o;
                if (og >= 9) continue;
                const int k = 9 * (og / 3) + 3 * G + og % 3;
                const int ei = (k / C::OPW) * C::SLICE + k % C::OPW;
                const int d = ei / 2;                                  // dword of the row: piece d / 4 (= quad lane), element d % 4
                u32 w0, w1;
                switch (d / 4) {                                       // (the DPP control is an immediate)
                    case 0: w0 = __builtin_amdgcn_mov_dpp(tl[d % 4], 0x00, 0xF, 0xF, true); w1 = __builtin_amdgcn_mov_dpp(th[d % 4], 0x00, 0xF, 0xF, true); break;
                    case 1: w0 = __builtin_amdgcn_mov_dpp(tl[d % 4], 0x55, 0xF, 0xF, true); w1 = __builtin_amdgcn_mov_dpp(th[d % 4], 0x55, 0xF, 0xF, true); break;
                    case 2: w0 = __builtin_amdgcn_mov_dpp(tl[d % 4], 0xAA, 0xF, 0xF, true); w1 = __builtin_amdgcn_mov_dpp(th[d % 4], 0xAA, 0xF, 0xF, true); break;
                    default: w0 = __builtin_amdgcn_mov_dpp(tl[d % 4], 0xFF, 0xF, 0xF, true); w1 = __builtin_amdgcn_mov_dpp(th[d % 4], 0xFF, 0xF, 0xF, true); break;
                }
                const unsigned e0 = (ei & 1) ? w0 >> 16 : w0 & 0xffffu, e1 = (ei & 1) ? w1 >> 16 : w1 & 0xffffu;
                const unsigned q0 = e0 ? e0 + adj : 0u, q1 = e1 ? e1 + adj : 0u;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3))) *)(smem + (((q0 ^ csel) << 4) + sub)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3))) *)(smem + (((q1 ^ csel) << 4) + sub)));
                af[o] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
            static_assert(NSL <= NO, "one prefetch slot per offset");
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                if (pf && s == rs && o < NSL) *pf_count += issue_slot(pn, tn, bn, 0, o);
                if (OH * NO + o >= 9) continue;
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[o][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[o], bf[j], acc[o][j], 0, 0, 0);
            }
        }
    };
    auto compute = [&](int buf, bool pf, const Run pn, int tn, int bn, int *pf_count) {
        if (W::NOH == 1 || oh == 0) compute_oh(buf, std::integral_constant<int, 0>{}, pf, pn, tn, bn, pf_count);
        else compute_oh(buf, std::integral_constant<int, W::NOH - 1>{}, pf, pn, tn, bn, pf_count);
    };

    static_assert(W::NBUF == 3, "prefetch distance 2 below");
    for (int chunk0 = t_begin; chunk0 < t_end; chunk0 += WIN_PLAN_CAP) {
        const int chunk1 = min(t_end, chunk0 + WIN_PLAN_CAP);
        __syncthreads();
        int c_next = 0;                      // DMA instructions of this wave in flight for tile t + 2 (after the wait below: t + 1)
        {
            // the first two tiles of the chunk: their headers straight from the plan (wave-uniform loads)
            const Run p0 = run_of(hdr_g[(size_t)chunk0 * 2], hdr_g[(size_t)chunk0 * 2 + 1]);
            issue(p0, chunk0, 0, 0);
            if (chunk0 + 1 < chunk1) {
                const Run p1 = run_of(hdr_g[(size_t)chunk0 * 2 + 2], hdr_g[(size_t)chunk0 * 2 + 3]);
                c_next = issue(p1, chunk0 + 1, 1, 0);
            }
            for (int e = tid; e < (chunk1 - chunk0) * 2; e += C::THREADS) ((int4 *)(smem + W::PLAN))[e] = hdr_g[(size_t)chunk0 * 2 + e];
            wait_vm(c_next);                 // (the plan loads are older than nothing here: staged below the barrier anyway)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        int buf = 0;
        for (int t = chunk0; t < chunk1; ++t) {
            const Run p = run_of(plan_s[(t - chunk0) * 2], plan_s[(t - chunk0) * 2 + 1]);
            // tile t + 2 into the stage tile t - 1 left (every wave passed the barrier behind its compute)
            int c_new = 0;
            stamp();
            const bool pf = t + 2 < chunk1;
            const int tp = pf ? t + 2 : t;
            const Run pn = run_of(plan_s[(tp - chunk0) * 2], plan_s[(tp - chunk0) * 2 + 1]);
            stamp();
            compute(buf, pf, pn, t + 2, buf == 0 ? 2 : buf - 1, &c_new);
            stamp();
            for (int pass = 1; pass * R < p.cnt; ++pass) {
                // the run is longer than the window (rare): its next R rows into the same stage, the nine table columns of
                // the run rebuilt from the rulebook for that part
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                WIN_BARRIER();
                issue(p, t, buf, pass);
                unsigned short *tabw = (unsigned short *)(smem + W::TAB0 + buf * C::TABB);
#pragma unroll
                for (int u = 0; u < W::NL; ++u) {
                    const int e = tid + u * C::THREADS;
                    const int o = e / T, r = e - o * T;
                    const int k = 9 * (o / 3) + 3 * G + o % 3;
                    if (e < 9 * T) {
                        const int v = t * T + r < n ? nbr[(size_t)k * nbr_stride + t * T + r] : -1;
                        const unsigned rel = (unsigned)(v - (p.lo + pass * R));
                        const unsigned row = (unsigned)(C::Z0 + G * R) + rel;
                        tabw[C::tab_pos((unsigned)r, (unsigned)k)] =
                            (v >= 0 && rel < (unsigned)R) ? (unsigned short)(row * S + C::swz(row)) : (unsigned short)0;
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                WIN_BARRIER();
                int none = 0;
                compute(buf, false, p, t, buf, &none);
                c_new = 0;                   // (everything has been waited for)
            }
            wait_vm(c_new);                  // tile t + 1 has landed (this wave's part): only tile t + 2's may be in flight
            stamp();
            WIN_BARRIER();                   // ... everybody's, and everybody is done with stage buf
            buf = buf == 2 ? 0 : buf + 1;
        }
    }

    // the waves of a group that split the row steps: summed in wave order through LDS
    stamp();
    __syncthreads();
    if (W::RS > 1) {
        float4 *red = (float4 *)smem;
        if (rs > 0) {
#pragma unroll
            for (int o = 0; o < NO; ++o)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    red[(((grp * (W::RS - 1) + rs - 1) * NO + o) * NB + j) * 64 + lane] =
                        make_float4(acc[o][j][0], acc[o][j][1], acc[o][j][2], acc[o][j][3]);
        }
        __syncthreads();
        if (rs == 0) {
            for (int q = 0; q < W::RS - 1; ++q)
#pragma unroll
                for (int o = 0; o < NO; ++o)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const float4 v = red[(((grp * (W::RS - 1) + q) * NO + o) * NB + j) * 64 + lane];
                        acc[o][j][0] += v.x; acc[o][j][1] += v.y; acc[o][j][2] += v.z; acc[o][j][3] += v.w;
                    }
        }
    }
    if (rs == 0) {
        // lane (g4, t16) of block (mb, nb) holds dW_k[ci = 16 mb + 4 g4 .. + 3][co = 16 nb + t16]
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const int og = oh * NO + o;
            const int k = 9 * (og / 3) + 3 * G + og % 3;
            if (og < 9) {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int co = 16 * j + t16, ci = 16 * mb + 4 * g4;
                    *reinterpret_cast<float4 *>(slab + (((size_t)share * CIN + co) * 27 + k) * CIN + ci) =
                        make_float4(acc[o][j][0], acc[o][j][1], acc[o][j][2], acc[o][j][3]);
                }
            }
        }
    }
    stamp();
    if (trace && threadIdx.x == 0 && blockIdx.x < 256) {         // per-workgroup entry / exit times (trace[256 + b], trace[512 + b])
        trace[256 + blockIdx.x] = t_entry;
        trace[512 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
}

template <class C>
__global__ __launch_bounds__(C::THREADS, C::NW == 8 ? 1 : 2) void subm_wgrad_win_kernel(
    const unsigned short *__restrict__ x, const unsigned short *__restrict__ dy, const int32_t *__restrict__ nbr, int nbr_stride,
    int n_cap, const int32_t *__restrict__ n_dev, const char *__restrict__ plan_g, float *__restrict__ slab, unsigned x_bytes,
    unsigned dy_bytes, unsigned long long *trace) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    if (j >= 3 * (C::WG_SHARES / 8)) return;
    const int share = xcd * (C::WG_SHARES / 8) + j / 3;
    const int n = eff_rows(n_dev, n_cap);
    switch (j % 3) {
        case 0: wgrad_win_body<C, 0>(x, dy, nbr, nbr_stride, n_cap, n, plan_g, slab, x_bytes, dy_bytes, share, smem, trace); break;
        case 1: wgrad_win_body<C, 1>(x, dy, nbr, nbr_stride, n_cap, n, plan_g, slab, x_bytes, dy_bytes, share, smem, trace); break;
        default: wgrad_win_body<C, 2>(x, dy, nbr, nbr_stride, n_cap, n, plan_g, slab, x_bytes, dy_bytes, share, smem, trace); break;
    }
}

template <class C>
static int launch_wgrad_win(const void *x, const void *dy, int n_rows, const int32_t *nbr, int nbr_stride, const int32_t *n_dev,
                            const void *plan, float *slab, hipStream_t st) {
    using W = WgCfg<C>;
    if ((double)n_rows * C::ROWB >= 4294967040.0) return PCD_ERR_UNSUPPORTED;
    auto k = subm_wgrad_win_kernel<C>;
    if (hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, W::LDS_BYTES) != hipSuccess)
        return PCD_ERR_LAUNCH;
    const unsigned bytes = (unsigned)((size_t)n_rows * C::ROWB);
    k<<<WIN_GRID * (8 / C::NW), C::THREADS, W::LDS_BYTES, st>>>((const unsigned short *)x, (const unsigned short *)dy, nbr, nbr_stride, n_rows,
                                                   n_dev, (const char *)plan, slab, bytes, bytes, g_win_trace);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

static bool win_supported(int c_in, int c_out) {
    return win_dispatch(c_in, c_out, [](auto) { return true; }, false);
}

}  // namespace

// =============================================================================================
extern "C" int pcd_subm_window_tile_rows(int c_in, int c_out) {
    return win_dispatch(c_in, c_out, [](auto c) { return (int)decltype(c)::T; }, 0);
}

// rows of PcdBnReduce.partial a forward / data-gradient launch of these widths writes (one per workgroup)
extern "C" int pcd_subm_window_partial_rows(int c_in, int c_out) {
    return win_dispatch(c_in, c_out, [](auto c) { return win_grid<decltype(c)>(); }, 0);
}

extern "C" int pcd_subm_window_set_trace(void *buf256_u64) {
    g_win_trace = (unsigned long long *)buf256_u64;
    return PCD_OK;
}

extern "C" size_t pcd_subm_window_plan_bytes(int n_cap, int c_in, int c_out) {
    if (n_cap < 0) return 0;
    const int nc = n_cap > 0 ? n_cap : 1;
    return win_dispatch(c_in, c_out, [&](auto c) { using C = decltype(c); return C::plan_bytes(pcd_div_up(nc, C::T)); }, (size_t)0);
}

extern "C" int pcd_subm_window_plan(const int32_t *nbr, int nbr_stride, int n_cap, const int32_t *n_dev, int c_in,
                                    int c_out, void *plan, void *stream) {
    PCD_ENTER();
    if (n_cap < 0 || !win_supported(c_in, c_out)) return PCD_ERR_INVALID_ARG;
    if (n_cap == 0) return PCD_OK;
    if (!nbr || !plan || nbr_stride < n_cap) return PCD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    win_dispatch(c_in, c_out, [&](auto c) {
        using C = decltype(c);
        const int nt = pcd_div_up(n_cap, C::T);
        win_plan_kernel<C><<<pcd_div_up(nt, 4 * (C::T >= 64 ? 1 : 64 / C::T)), 256, 0, st>>>(nbr, nbr_stride, n_cap, n_dev, (char *)plan, nt);
        win_split_kernel<C><<<1, 1024, 0, st>>>((char *)plan, n_cap, n_dev, nt, win_grid<C>());
        return 0;
    }, 0);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

// The plan of a SubM 3x3x3 level built STRAIGHT from its column map (rows z-fastest): what pcd_rulebook_subm_cm +
// pcd_subm_window_plan produce together, without the 27 x 4 B per row of the neighbour table in between.  nbr [27][n_cap]:
// nbr_full != 0 -- the whole table is written as well (for consumers that read it: generic kernels, pair lists); nbr_full == 0 --
// only the columns of tiles that need a second pass (a run longer than the window), which is all the window kernels ever read.
extern "C" int pcd_subm_window_plan_cm(const int32_t *indices, int n_cap, const int32_t *n_dev, int batch, const int *shape_host,
                                       const void *colmap, size_t colmap_bytes, int colmap_cap, int c_in, int c_out,
                                       int32_t *nbr, int nbr_full, void *plan, void *stream) {
    PCD_ENTER();
    if (n_cap < 0 || batch <= 0 || !shape_host || !win_supported(c_in, c_out)) return PCD_ERR_INVALID_ARG;
    if (shape_host[0] <= 0 || shape_host[0] > 62) return PCD_ERR_UNSUPPORTED;
    if (n_cap == 0) return PCD_OK;
    if (!indices || !colmap || !plan || !nbr) return PCD_ERR_INVALID_ARG;
    CmBuf B;
    if (!cm_carve(const_cast<void *>(colmap), colmap_bytes, batch, shape_host[1], shape_host[2], colmap_cap, B, nullptr))
        return PCD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    win_dispatch(c_in, c_out, [&](auto c) {
        using C = decltype(c);
        const int nt = pcd_div_up(n_cap, C::T);
        cm_win_plan_kernel<C><<<pcd_div_up(nt, 4 * (C::T >= 64 ? 1 : 64 / C::T)), 256, 0, st>>>(
            (const int4 *)indices, n_cap, n_dev, shape_host[0], shape_host[1], shape_host[2], B.pitch, B.cw, B.cr, B.ncol_cap,
            (char *)plan, nt, nbr, nbr_full);
        win_split_kernel<C><<<1, 1024, 0, st>>>((char *)plan, n_cap, n_dev, nt, win_grid<C>());
        return 0;
    }, 0);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" size_t pcd_subm_window_packed_weight_bytes(int c_in, int c_out) {
    return win_dispatch(c_in, c_out, [](auto c) { return win_pack_elems<decltype(c)>() * 2; }, (size_t)0);
}

extern "C" int pcd_subm_window_pack_weight(const float *weight, int c_in, int c_out, int mode, void *packed, void *stream) {
    PCD_ENTER();
    // c_in < c_out (forward pack only): a layer whose input rows are zero-padded to c_out channels (5 -> 16)
    if (!weight || !packed || (mode != 0 && mode != 1) || c_in <= 0 || c_in > c_out || (c_in < c_out && mode != 0) ||
        !win_supported(c_out, c_out))
        return PCD_ERR_INVALID_ARG;
    const size_t total = pcd_subm_window_packed_weight_bytes(c_out, c_out) / 2;
    win_pack_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(weight, c_out, mode,
                                                                                     (unsigned short *)packed, c_in);
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
                                           const int32_t *nbr, int nbr_stride, const int32_t *n_rows_dev,
                                           const void *plan, int c_out, void *y, const void *addend,
                                           const PcdBnReduce *bn_reduce, void *stream) {
    PCD_ENTER();
    if (n_rows < 0 || !win_supported(c_in, c_out)) return PCD_ERR_UNSUPPORTED;
    if (n_rows == 0) return PCD_OK;
    if (!x || !packed_w || !nbr || !plan || !y || nbr_stride < n_rows) return PCD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    return win_dispatch(c_in, c_out, [&](auto c) {
        return launch_win<decltype(c)>(x, n_rows, packed_w, bias, nbr, nbr_stride, n_rows_dev, plan, y, addend,
                                       bn_reduce, st);
    }, (int)PCD_ERR_UNSUPPORTED);
}

// the same launch, additionally writing the fp32 sums (bias and addend included) every bf16 output is rounded from
extern "C" int pcd_sparse_conv_subm_window_f32(const void *x, int n_rows, int c_in, const void *packed_w, const float *bias,
                                               const int32_t *nbr, int nbr_stride, const int32_t *n_rows_dev,
                                               const void *plan, int c_out, void *y, float *y_f32, const void *addend,
                                               void *stream) {
    PCD_ENTER();
    if (n_rows < 0 || !win_supported(c_in, c_out)) return PCD_ERR_UNSUPPORTED;
    if (n_rows == 0) return PCD_OK;
    if (!x || !packed_w || !nbr || !plan || !y || !y_f32 || nbr_stride < n_rows) return PCD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    return win_dispatch(c_in, c_out, [&](auto c) {
        return launch_win<decltype(c)>(x, n_rows, packed_w, bias, nbr, nbr_stride, n_rows_dev, plan, y, addend, nullptr,
                                       st, y_f32);
    }, (int)PCD_ERR_UNSUPPORTED);
}

// partial slabs a weight-gradient launch at c channels leaves (= `splits` of its reduction job)
extern "C" int pcd_subm_window_wgrad_splits(int c) {
    return win_dispatch(c, c, [](auto cfg) { return (int)decltype(cfg)::WG_SHARES; }, 0);
}

extern "C" int pcd_sparse_conv_subm_window_wgrad(const void *x, const void *dy, int n_rows, int c, const int32_t *nbr,
                                                 int nbr_stride, const int32_t *n_rows_dev, const void *plan, void *slab,
                                                 size_t slab_bytes, void *stream) {
    PCD_ENTER();
    if (n_rows < 0 || !win_supported(c, c) || c > 64) return PCD_ERR_UNSUPPORTED;
    const size_t need = (size_t)pcd_subm_window_wgrad_splits(c) * 27 * c * c * sizeof(float);
    if (!slab || slab_bytes < need) return PCD_ERR_WORKSPACE;
    if (n_rows == 0) {
        if (hipMemsetAsync(slab, 0, need, (hipStream_t)stream) != hipSuccess)
            return PCD_ERR_LAUNCH;
        return PCD_OK;
    }
    if (!x || !dy || !nbr || !plan || nbr_stride < n_rows) return PCD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    return win_dispatch(c, c, [&](auto cfg) {
        if constexpr (decltype(cfg)::CIN <= 64)         // (the 128-channel configuration of EXPERIMENTS builds has no weight-gradient kernel)
            return launch_wgrad_win<decltype(cfg)>(x, dy, n_rows, nbr, nbr_stride, n_rows_dev, plan, (float *)slab, st);
        else
            return (int)PCD_ERR_UNSUPPORTED;
    }, (int)PCD_ERR_UNSUPPORTED);
}
