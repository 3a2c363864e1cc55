// Rulebook construction for sparse 3D convolution on gfx950.
//
// Replaces spconv's get_indice_pairs behind spconv.SubMConv3d / spconv.SparseConv3d
// (pcdet/models/backbones_3d/spconv_backbone.py:12-15); definitions: SURVEY.md Appendix A.4.
//
// Two views of every rulebook are produced:
//   * neighbour tables  nbr[k][row]  (k-major: lanes = consecutive rows -> coalesced) which drive
//     the output-stationary gather-GEMM kernels (no atomics in the feature path);
//   * spconv's indice_pairs [K][2][P] + indice_pair_num [K] in canonical order (ascending input row
//     inside each k), used by the weight-gradient kernel and exported for API parity.
// Pair compaction is deterministic: per-wave ballot/popcount counts -> row-wise exclusive scan of
// the small [K][n/64] count matrix -> each wave writes its pairs at (wave offset + lane rank).
//
// SubM: coordinate -> row through an open-addressing hash table of packed {key32, row32} 8-byte
// slots (one CAS to insert, one 8-byte load per probe).
// Strided conv: active output cells are marked in an occupancy bitmap over the output grid; a
// popcount prefix over the bitmap words IS the sorted-unique ranking (row id = rank of the linear
// key), so neither a sort nor a hash table is needed and lookups are 2 loads.
#include "rulebook_common.h"

namespace {

constexpr u64 SLOT_EMPTY = ~0ull;

__global__ __launch_bounds__(256) void hash_insert_kernel(const int4 *__restrict__ idx, int n,
                                                          const int32_t *n_dev, int D, int H, int W,
                                                          u64 *table, u32 mask) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= eff_rows(n_dev, n)) return;
    int4 c = idx[i];
    u32 key = lin_key(c.x, c.y, c.z, c.w, D, H, W);
    u64 packed = ((u64)key << 32) | (u32)i;
    u32 h = hash_u32(key) & mask;
    for (;;) {
        u64 prev = atomicCAS(&table[h], SLOT_EMPTY, packed);
        if (prev == SLOT_EMPTY) break;
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ int hash_lookup(const u64 *__restrict__ table, u32 mask, u32 key) {
    u32 h = hash_u32(key) & mask;
    for (;;) {
        u64 s = table[h];
        if (s == SLOT_EMPTY) return -1;
        if ((u32)(s >> 32) == key) return (int)(u32)s;
        h = (h + 1) & mask;
    }
}

// One thread per output row, loop over the K offsets.  nbr[k][o] = row of (coord_o - c*dil + k*dil).
// wave_cnt[k][wave] = number of hits of this wave's 64 rows (feeds the pair compaction).
__global__ __launch_bounds__(256) void subm_probe_kernel(const int4 *__restrict__ idx, int n,
                                                         const int32_t *n_dev, ConvGeom G,
                                                         const u64 *__restrict__ table,
                                                         u32 mask, int32_t *__restrict__ nbr,
                                                         int *__restrict__ wave_cnt, int nwaves) {
    int o = blockIdx.x * 256 + threadIdx.x;
    bool live = o < eff_rows(n_dev, n);
    int4 c = live ? idx[o] : make_int4(0, 0, 0, 0);
    int wave = o >> 6;
    int k = 0;
    for (int a = 0; a < G.kd; ++a) {
        int z = c.y + (a - G.kd / 2) * G.dd;
        for (int bq = 0; bq < G.kh; ++bq) {
            int y = c.z + (bq - G.kh / 2) * G.dh;
            for (int cq = 0; cq < G.kw; ++cq, ++k) {
                int x = c.w + (cq - G.kw / 2) * G.dw;
                int r = -1;
                if (live) {
                    if (2 * k + 1 == G.K) {
                        r = o;  // centre offset: the row itself
                    } else if (z >= 0 && z < G.D && y >= 0 && y < G.H && x >= 0 && x < G.W) {
                        r = hash_lookup(table, mask, lin_key(c.x, z, y, x, G.D, G.H, G.W));
                    }
                    nbr[(size_t)k * n + o] = r;
                }
                if (wave_cnt) {
                    u64 m = __ballot(r >= 0);
                    if (lane_id() == 0 && wave < nwaves) wave_cnt[(size_t)k * nwaves + wave] = __popcll(m);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Level-1 SubM rulebook (rows in first-appearance order, no spatial order to exploit) through a hash of 4x4x4
// BLOCKS instead of a hash of cells: a 16-byte slot {block key + 1, base, 64-bit occupancy mask} per non-empty
// block and one row list (rows of a block contiguous, ordered by cell bit).  A row's 26 neighbours live in at most
// 8 blocks, one of them its own (slot remembered from the insert): (1.5^3 - 1) = 2.4 probes per row on average
// instead of 26, and the table is ~N/6 slots of 16 bytes (L2-sized at B = 4) instead of 2N slots of 8 bytes.
struct Blk {
    u32 key1;   // block key + 1; 0 = empty
    int base;   // first entry of this block in the row list
    u64 mask;   // bit (z&3)<<4 | (y&3)<<2 | (x&3)
};
static_assert(sizeof(Blk) == 16, "slot");

__device__ __forceinline__ u32 blk_key(int b, int bz, int by, int bx, int Dz, int Hy, int Wx) {
    return (((u32)b * Dz + bz) * Hy + by) * Wx + bx;
}

// Rows arrive in first-appearance order of a range-image sweep: neighbouring rows mostly fall into the same 4x4x4
// block.  Lanes that hold the same block as the lane before them form a RUN; only the run's first lane probes the
// table (one CAS) and ORs the run's combined cell mask in (one atomic) -- 2 atomics per run instead of 2 per row.
__global__ __launch_bounds__(256) void blk_insert_kernel(const int4 *__restrict__ idx, int n,
                                                         const int32_t *n_dev, int Dz, int Hy, int Wx, Blk *table,
                                                         u32 mask, int32_t *__restrict__ rowslot) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = i < eff_rows(n_dev, n);
    const int4 c = live ? idx[i] : make_int4(0, 0, 0, 0);
    const u32 key1 = live ? blk_key(c.x, c.y >> 2, c.z >> 2, c.w >> 2, Dz, Hy, Wx) + 1u : 0u;   // 0: dead lane
    u64 bits = live ? 1ull << (((c.y & 3) << 4) | ((c.z & 3) << 2) | (c.w & 3)) : 0ull;
    const u32 prev_key = (u32)__shfl_up((int)key1, 1);
    const bool head = lane == 0 || prev_key != key1;
    const u64 heads = __ballot(head);
    const int head_lane = 63 - __clzll(heads & ((2ull << lane) - 1ull));
    // suffix OR inside the run: afterwards the run's first lane holds the OR of all its members
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 t = __shfl_down(bits, d);
        const int hl = __shfl_down(head_lane, d);
        if (lane + d < 64 && hl == head_lane) bits |= t;
    }
    u32 h = 0;
    if (head && live) {
        h = hash_u32(key1) & mask;
        for (;;) {
            u32 prev = table[h].key1;
            if (prev == 0u) prev = atomicCAS(&table[h].key1, 0u, key1);
            if (prev == 0u || prev == key1) break;
            h = (h + 1) & mask;
        }
        atomicOr((unsigned long long *)&table[h].mask, bits);
    }
    h = (u32)__shfl((int)h, head_lane);
    if (live) rowslot[i] = (int)h;
}

// base of every non-empty block in the row list: popcount prefix inside a 4096-slot chunk (4 slots per thread,
// wave scan, LDS scan of the 16 wave totals) and ONE atomicAdd per chunk -- same-address atomics cost ~6 ns each,
// one per wave took longer than the probes.  The order of the blocks in the list is irrelevant: only lookups
// through `base` ever read it.
__global__ __launch_bounds__(1024) void blk_base_kernel(Blk *table, u32 cap, int *total) {
    __shared__ int wsum[16];
    __shared__ int chunk_base;
    const u32 s0 = (blockIdx.x * 1024u + threadIdx.x) * 4u;
    int cnt[4];
    int mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 sidx = s0 + j;
        cnt[j] = (sidx < cap && table[sidx].key1 != 0u) ? __popcll(table[sidx].mask) : 0;
        mine += cnt[j];
    }
    int inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(inc, d);
        if (lane_id() >= d) inc += t;
    }
    const int w = threadIdx.x >> 6;
    if (lane_id() == 63) wsum[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int q = 0; q < 16; ++q) {
            int t = wsum[q];
            wsum[q] = run;
            run += t;
        }
        chunk_base = run > 0 ? atomicAdd(total, run) : 0;
    }
    __syncthreads();
    int base = chunk_base + wsum[w] + inc - mine;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (cnt[j] > 0) table[s0 + j].base = base;
        base += cnt[j];
    }
}

__global__ __launch_bounds__(256) void blk_fill_kernel(const int4 *__restrict__ idx, int n, const int32_t *n_dev,
                                                       const Blk *__restrict__ table,
                                                       const int32_t *__restrict__ rowslot,
                                                       int32_t *__restrict__ rowlist) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= eff_rows(n_dev, n)) return;
    int4 c = idx[i];
    const uint4 raw = *reinterpret_cast<const uint4 *>(&table[rowslot[i]]);
    const u64 m = ((u64)raw.w << 32) | raw.z;
    const int bit = ((c.y & 3) << 4) | ((c.z & 3) << 2) | (c.w & 3);
    rowlist[(int)raw.y + __popcll(m & ((1ull << bit) - 1ull))] = i;
}

__global__ __launch_bounds__(256) void blk_probe_kernel(const int4 *__restrict__ idx, int n,
                                                        const int32_t *n_dev, ConvGeom G, int Dz, int Hy, int Wx,
                                                        const Blk *__restrict__ table, u32 mask,
                                                        const int32_t *__restrict__ rowslot,
                                                        const int32_t *__restrict__ rowlist,
                                                        int32_t *__restrict__ nbr, int *__restrict__ wave_cnt,
                                                        int nwaves) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    const bool live = o < eff_rows(n_dev, n);
    const int4 c = live ? idx[o] : make_int4(0, 0, 0, 0);
    const int wave = o >> 6;
    const int bz = c.y >> 2, by = c.z >> 2, bx = c.w >> 2;
    const int cz = c.y & 3, cy = c.z & 3, cx = c.w & 3;
    // per axis: side 0 = own block coordinate, side 1 = the neighbouring block a +-1 step can reach (if any)
    const int oz = cz == 0 ? bz - 1 : (cz == 3 ? bz + 1 : -1);
    const int oy = cy == 0 ? by - 1 : (cy == 3 ? by + 1 : -1);
    const int ox = cx == 0 ? bx - 1 : (cx == 3 ? bx + 1 : -1);
    const bool hz = oz >= 0 && oz < Dz, hy = oy >= 0 && oy < Hy, hx = ox >= 0 && ox < Wx;
    u64 bm[8];
    int bb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        bm[q] = 0ull;
        bb[q] = 0;
    }
    if (live) {
        const uint4 own = *reinterpret_cast<const uint4 *>(&table[rowslot[o]]);
        bm[0] = ((u64)own.w << 32) | own.z;
        bb[0] = (int)own.y;
#pragma unroll
        for (int q = 1; q < 8; ++q) {
            const bool sz = q & 4, sy = q & 2, sx = q & 1;
            if ((sz && !hz) || (sy && !hy) || (sx && !hx)) continue;
            const u32 key1 = blk_key(c.x, sz ? oz : bz, sy ? oy : by, sx ? ox : bx, Dz, Hy, Wx) + 1u;
            u32 h = hash_u32(key1) & mask;
            for (;;) {
                const uint4 sl = *reinterpret_cast<const uint4 *>(&table[h]);
                if (sl.x == 0u) break;
                if (sl.x == key1) {
                    bm[q] = ((u64)sl.w << 32) | sl.z;
                    bb[q] = (int)sl.y;
                    break;
                }
                h = (h + 1) & mask;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
        int r = -1;
        if (live) {
            const int z = cz + dz * G.dd, y = cy + dy * G.dh, x = cx + dx * G.dw;  // dilation 1 only (checked on host)
            const bool sz = z < 0 || z > 3, sy = y < 0 || y > 3, sx = x < 0 || x > 3;
            const int q = (sz ? 4 : 0) | (sy ? 2 : 0) | (sx ? 1 : 0);
            u64 m = bm[0];
            int base = bb[0];
#pragma unroll
            for (int t = 1; t < 8; ++t) {
                m = q == t ? bm[t] : m;
                base = q == t ? bb[t] : base;
            }
            const int bit = ((z & 3) << 4) | ((y & 3) << 2) | (x & 3);
            if ((m >> bit) & 1ull) r = rowlist[base + __popcll(m & ((1ull << bit) - 1ull))];
            if (k == 13) r = o;
            nbr[(size_t)k * n + o] = r;
        }
        if (wave_cnt) {
            u64 mm = __ballot(r >= 0);
            if (lane_id() == 0 && wave < nwaves) wave_cnt[(size_t)k * nwaves + wave] = __popcll(mm);
        }
    }
}

// SubM rulebook of a level whose rows ARE the bitmap ranks of the strided conv that created it (row id = rank of
// the linear key): a neighbour lookup is one bitmap word + one prefix word, both shared by the x-neighbours and
// by the neighbouring rows of the (key-sorted) wave -- no hash table is built or probed.
// PW = words per prefix entry: 1 (the strided builds' maps: one prefix per bitmap word) or 4 (the key-ordered voxeliser's
// map of the level-1 grid, pcd_voxelize_hard_sorted: one prefix per 16-byte group, the rank adds the set bits of the
// group's words in front -- a quarter of the prefix memory for a grid of 371 M cells).
template <int KD, int KH, int KW, int PW, int ORDER = PCD_ROWS_ZYX>
__global__ __launch_bounds__(256) void subm_rank_kernel(const int4 *__restrict__ idx, int n,
                                                        const int32_t *n_dev, ConvGeom G,
                                                        const u32 *__restrict__ bitmap,
                                                        const int *__restrict__ prefix,
                                                        int32_t *__restrict__ nbr, int *__restrict__ wave_cnt,
                                                        int nwaves) {
    constexpr int K = KD * KH * KW;
    constexpr int KB = PW == 1 ? K : KH * KW;      // offsets whose loads are in flight together
    const int o = blockIdx.x * 256 + threadIdx.x;
    const bool live = o < eff_rows(n_dev, n);
    const int4 c = live ? idx[o] : make_int4(0, 0, 0, 0);
    const int wave = o >> 6;
    if (ORDER == PCD_ROWS_ZYX ? (KW == 3 && G.dw == 1) : (KD == 3 && G.dd == 1)) {
        // The three neighbours along the key's FASTEST axis (x in ZYX order, z in YXZ order) of a line are three
        // CONSECUTIVE keys: one fetch of the bitmap (the word / group of key - 1 and, across a word boundary, its
        // successor) and ONE prefix serve all three -- the prefix sums are cumulative, so rank(key) = prefix(first word)
        // + set bits in front of the key within the fetched run.  A third of the probe loads of the per-offset form below
        // (which remains for dilated kernels).  Lines: (dz, dy) in ZYX order, (dy, dx) in YXZ order.
        constexpr int NL = ORDER == PCD_ROWS_ZYX ? KD * KH : KH * KW;     // lines
        constexpr int LG = ORDER == PCD_ROWS_ZYX ? KH : KW;               // lines per group of the PW = 4 form
#pragma unroll
        for (int l0 = 0; l0 < NL; l0 += (PW == 1 ? NL : LG)) {
            constexpr int LB = PW == 1 ? NL : LG;            // lines whose loads are in flight together
            u64 run[LB];
            int pre[LB];
            int p0[LB];
            bool line_ok[LB];
#pragma unroll
            for (int ll = 0; ll < LB; ++ll) {
                const int l = l0 + ll;
                bool ok;
                u32 keyc;
                if (ORDER == PCD_ROWS_ZYX) {
                    const int a = l / KH, bq = l % KH;
                    const int z = c.y + (a - KD / 2) * G.dd, y = c.z + (bq - KH / 2) * G.dh;
                    ok = live && z >= 0 && z < G.D && y >= 0 && y < G.H;
                    keyc = ok ? lin_key(c.x, z, y, c.w, G.D, G.H, G.W) : 1u;            // key of (z, y, x)
                } else {
                    const int bq = l / KW, cq = l % KW;
                    const int y = c.z + (bq - KH / 2) * G.dh, x = c.w + (cq - KW / 2) * G.dw;
                    ok = live && y >= 0 && y < G.H && x >= 0 && x < G.W;
                    keyc = ok ? ord_key(PCD_ROWS_YXZ, c.x, c.y, y, x, G.D, G.H, G.W) : 1u;   // key of (y, x, z)
                }
                line_ok[ll] = ok;
                const u32 key0 = keyc == 0u ? 0u : keyc - 1u;                           // key of the -1 step (at 0: not probed)
                const u32 w0 = key0 >> 5;
                p0[ll] = (int)(keyc - 1u - (w0 << 5));                                  // bit of the -1 step in the run (-1 at key 0)
                if (PW == 1) {
                    const u32 lo = bitmap[w0], hi = bitmap[w0 + 1];     // (the maps are padded by one word)
                    pre[ll] = prefix[w0];
                    run[ll] = (u64)lo | ((u64)hi << 32);
                } else {
                    const uint4 q = reinterpret_cast<const uint4 *>(bitmap)[w0 >> 2];
                    const u32 wi = w0 & 3u;
                    const u32 lo = wi == 0 ? q.x : wi == 1 ? q.y : wi == 2 ? q.z : q.w;
                    u32 hi = wi == 0 ? q.y : wi == 1 ? q.z : wi == 2 ? q.w : 0u;
                    if (wi == 3u && p0[ll] >= 30) hi = bitmap[w0 + 1];                  // rare: the run leaves the 16-byte group
                    pre[ll] = prefix[w0 >> 2] + (wi > 0 ? __popc(q.x) : 0) + (wi > 1 ? __popc(q.y) : 0) + (wi > 2 ? __popc(q.z) : 0);
                    run[ll] = (u64)lo | ((u64)hi << 32);
                }
            }
#pragma unroll
            for (int ll = 0; ll < LB; ++ll)
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    const int l = l0 + ll;
                    // f = index along the fastest axis; k = (a * KH + bq) * KW + cq in both orders
                    const int k = ORDER == PCD_ROWS_ZYX ? l * 3 + f : (f * KH + l / KW) * KW + l % KW;
                    const int t = (ORDER == PCD_ROWS_ZYX ? c.w : c.y) + f - 1;
                    const int tmax = ORDER == PCD_ROWS_ZYX ? G.W : G.D;
                    int r = -1;
                    if (line_ok[ll] && t >= 0 && t < tmax) {
                        const int p = p0[ll] + f;
                        if ((run[ll] >> p) & 1ull) {
                            r = pre[ll] + __popcll(run[ll] & ((1ull << p) - 1ull));
                            if (r >= n) r = -1;
                        }
                    }
                    if (2 * k + 1 == K && live) r = o;
                    if (live) nbr[(size_t)k * n + o] = r;
                    if (wave_cnt) {
                        u64 m = __ballot(r >= 0);
                        if (lane_id() == 0 && wave < nwaves) wave_cnt[(size_t)k * nwaves + wave] = __popcll(m);
                    }
                }
        }
        return;
    }
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += KB) {
        u32 below[KB];     // set bits of the prefix group in front of the probed bit
        u32 hit[KB];
        int pre[KB];
        bool inside[KB];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
            const int k = k0 + kk;
            const int a = k / (KH * KW), bq = (k / KW) % KH, cq = k % KW;
            const int z = c.y + (a - KD / 2) * G.dd, y = c.z + (bq - KH / 2) * G.dh, x = c.w + (cq - KW / 2) * G.dw;
            const bool inb = live && z >= 0 && z < G.D && y >= 0 && y < G.H && x >= 0 && x < G.W;
            const u32 key = inb ? ord_key(ORDER, c.x, z, y, x, G.D, G.H, G.W) : 0u;
            inside[kk] = inb;
            const u32 sh = key & 31;
            if (PW == 1) {
                const u32 w = bitmap[key >> 5];   // unconditional (word 0 for dead lanes): independent loads in flight
                pre[kk] = prefix[key >> 5];
                hit[kk] = (w >> sh) & 1u;
                below[kk] = (u32)__popc(w & ((1u << sh) - 1u));
            } else {
                const uint4 q = reinterpret_cast<const uint4 *>(bitmap)[key >> 7];
                pre[kk] = prefix[key >> 7];
                const u32 wi = (key >> 5) & 3u;
                const u32 w = wi == 0 ? q.x : wi == 1 ? q.y : wi == 2 ? q.z : q.w;
                hit[kk] = (w >> sh) & 1u;
                below[kk] = (u32)__popc(w & ((1u << sh) - 1u)) + (wi > 0 ? (u32)__popc(q.x) : 0u) +
                            (wi > 1 ? (u32)__popc(q.y) : 0u) + (wi > 2 ? (u32)__popc(q.z) : 0u);
            }
        }
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
            const int k = k0 + kk;
            int r = -1;
            if (inside[kk] && hit[kk]) {
                r = pre[kk] + (int)below[kk];
                if (r >= n) r = -1;  // beyond the row capacity: that row does not exist
            }
            if (2 * k + 1 == K && live) r = o;
            if (live) nbr[(size_t)k * n + o] = r;
            if (wave_cnt) {
                u64 m = __ballot(r >= 0);
                if (lane_id() == 0 && wave < nwaves) wave_cnt[(size_t)k * nwaves + wave] = __popcll(m);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// strided conv
template <int ORD>
__device__ __forceinline__ bool out_cell(const ConvGeom &G, int4 c, int a, int bq, int cq, u32 &key) {
    int tz = c.y + G.pd - a * G.dd;
    int ty = c.z + G.ph - bq * G.dh;
    int tx = c.w + G.pw - cq * G.dw;
    if (tz < 0 || ty < 0 || tx < 0) return false;
    int oz = G.sd == 2 ? (tz >> 1) : (G.sd == 1 ? tz : tz / G.sd);
    int oy = G.sh == 2 ? (ty >> 1) : (G.sh == 1 ? ty : ty / G.sh);
    int ox = G.sw == 2 ? (tx >> 1) : (G.sw == 1 ? tx : tx / G.sw);
    if (oz * G.sd != tz || oy * G.sh != ty || ox * G.sw != tx) return false;
    if (oz >= G.Do || oy >= G.Ho || ox >= G.Wo) return false;
    key = ord_key(ORD, c.x, oz, oy, ox, G.Do, G.Ho, G.Wo);
    return true;
}

// blk_cnt != NULL: the launch also counts the rows of every parity class (ncls of them) per block
template <int ORD>
__global__ __launch_bounds__(256) void conv_mark_kernel(const int4 *__restrict__ idx, int n,
                                                        const int32_t *n_dev, ConvGeom G,
                                                        unsigned char *__restrict__ bytemap, int ncls,
                                                        int *__restrict__ blk_cnt) {
    __shared__ int ccnt[CLS_MAX];
    int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < eff_rows(n_dev, n);
    int4 c = live ? idx[i] : make_int4(0, 0, 0, 0);
    if (blk_cnt) block_class_counts(live ? row_class(c, G.pd, G.ph, G.pw, G.sd, G.sh, G.sw) : -1, ncls, ccnt, blk_cnt);
    if (!live) return;
    // validity is tested once per axis level (kd + nz*kh + nz*ny*kw tests instead of 3*K)
    for (int a = 0; a < G.kd; ++a) {
        const int oz = axis_out(c.y, G.pd, G.dd, G.sd, a, G.Do);
        if (oz < 0) continue;
        for (int bq = 0; bq < G.kh; ++bq) {
            const int oy = axis_out(c.z, G.ph, G.dh, G.sh, bq, G.Ho);
            if (oy < 0) continue;
            for (int cq = 0; cq < G.kw; ++cq) {
                const int ox = axis_out(c.w, G.pw, G.dw, G.sw, cq, G.Wo);
                if (ox < 0) continue;
                // one BYTE per output cell: plain stores of the same value need no atomics (about 8 inputs mark
                // each cell; the 32-bit-word bitmap this replaced cost one memory-side atomic per mark)
                bytemap[ord_key(ORD, c.x, oz, oy, ox, G.Do, G.Ho, G.Wo)] = 1;
            }
        }
    }
}

// ---- the scan over the occupancy bitmap, two launches without a spine kernel --------------------------------
// pass 1 (conv_pack_sum_kernel): bytemap -> bitmap words, popcount of every block of PK_WORDS words -> bsums[blk]
//         and, added up by atomics (integer: order-free), of every 64 blocks -> super[blk / 64];
// pass 2 (conv_scan_emit_kernel): every block rebuilds its own base from at most (#supers + 63) of those sums --
//         a few hundred L2-resident ints -- instead of waiting for a one-block spine scan in a launch of its own.
constexpr int PK_WORDS = 1024;
constexpr int CONV_DIRECT_BLOCKS = 4096;

__device__ __forceinline__ u32 pack16(uint4 v) {   // 16 bytes of 0 / 1 -> 16 bits
    const u32 q[4] = {v.x, v.y, v.z, v.w};
    u32 bits = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u32 t = q[j] & 0x01010101u;
        t = (t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xFu;
        bits |= t << (4 * j);
    }
    return bits;
}

// one 16-byte piece (half a word) per lane and load, 8 loads in flight; lane pairs combine their halves
__global__ __launch_bounds__(256) void conv_pack_sum_kernel(const uint4 *__restrict__ bytemap, size_t nwords,
                                                            u32 *__restrict__ bitmap, int *__restrict__ bsums,
                                                            int *__restrict__ super, int direct) {
    __shared__ int lds[4];
    const size_t h0 = (size_t)blockIdx.x * (2 * PK_WORDS) + threadIdx.x, nh = 2 * nwords;
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const size_t h = h0 + (size_t)j * 256;
        v[j] = h < nh ? bytemap[h] : make_uint4(0, 0, 0, 0);
    }
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const size_t h = h0 + (size_t)j * 256;
        const u32 mine = pack16(v[j]);
        cnt += __popc(mine);
        const u32 other = (u32)__shfl_xor((int)mine, 1, 64);
        if (!(threadIdx.x & 1) && h < nh) bitmap[h >> 1] = mine | (other << 16);
    }
    const int total = block_sum(cnt, lds);
    if (threadIdx.x == 0) {
        bsums[blockIdx.x] = total;
        if (total && !direct) atomicAdd(&super[blockIdx.x >> 6], total);
    }
}

struct ScanSide {          // what the scan launch does besides the scan (all optional)
    int32_t *out_indices;  // emit the coordinates of the set cells (row = rank), rows < n_out only
    int n_out;
    void *fill_a;          // 0xFF fills (nbr_out, perm): spread over all blocks of the launch
    size_t fill_a_bytes;
    void *fill_b;
    size_t fill_b_bytes;
    int *blk_cnt;          // parity classes: the LAST block of the grid (an extra one) turns counts into offsets
    int cls_nblk, ncls, cls_tile;
    int *vstart;
};

// blocks [0, nblk): prefix[w] = number of set cells in words < w, n_out = all of them; out_indices.
__global__ __launch_bounds__(256) void conv_scan_emit_kernel(const u32 *__restrict__ bitmap, int nwords, int nblk,
                                                             const int *__restrict__ bsums,
                                                             const int *__restrict__ super, int direct,
                                                             int *__restrict__ prefix,
                                                             int *__restrict__ n_out_dev, ConvGeom G, ScanSide S) {
    __shared__ int lds[4];
    __shared__ int ctot[CLS_MAX], cstart[CLS_MAX + 1];
    const int blk = blockIdx.x;
    const bool cls_block = S.blk_cnt != nullptr && blk == (int)gridDim.x - 1;
    if (cls_block) {
        class_offsets(S.blk_cnt, S.cls_nblk, S.ncls, S.cls_tile, S.vstart, ctot, cstart);
        return;
    }
    if (blk < nblk) {
        const int w0 = blk * PK_WORDS + threadIdx.x * 4;
        u32 b[4] = {0u, 0u, 0u, 0u};
        if (w0 + 3 < nwords) {
            const uint4 q = *reinterpret_cast<const uint4 *>(bitmap + w0);
            b[0] = q.x; b[1] = q.y; b[2] = q.z; b[3] = q.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = (w0 + j < nwords) ? bitmap[w0 + j] : 0u;
        }
        // base of this block: whole supers before it + the blocks of its own super before it; up to
        // CONV_DIRECT_BLOCKS blocks (direct): all block sums before it (<= 16 L2-resident loads per thread, and the
        // pack pass needs no atomics: 64 of them on one address serialise at ~150 ns each)
        int acc = 0;
        if (direct) {
            for (int i = threadIdx.x; i < blk; i += 256) acc += bsums[i];
        } else {
            const int sb = blk >> 6;
            for (int i = threadIdx.x; i < sb; i += 256) acc += super[i];
            if ((int)threadIdx.x < (blk & 63)) acc += bsums[(sb << 6) + threadIdx.x];
        }
        const int base = block_sum(acc, lds);
        if (blk == 0 && n_out_dev) {
            int t = 0;
            if (direct) {
                for (int i = threadIdx.x; i < nblk; i += 256) t += bsums[i];
            } else {
                const int nsuper = (nblk + 63) >> 6;
                for (int i = threadIdx.x; i < nsuper; i += 256) t += super[i];
            }
            t = block_sum(t, lds);
            if (threadIdx.x == 0) *n_out_dev = t;
        }
        const int c0 = __popc(b[0]), c1 = __popc(b[1]), c2 = __popc(b[2]), c3 = __popc(b[3]);
        int total;
        const int ex = base + block_exclusive_scan(c0 + c1 + c2 + c3, lds, total);
        const int pre[4] = {ex, ex + c0, ex + c0 + c1, ex + c0 + c1 + c2};
        if (w0 + 3 < nwords) {
            *reinterpret_cast<int4 *>(prefix + w0) = make_int4(pre[0], pre[1], pre[2], pre[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (w0 + j < nwords) prefix[w0 + j] = pre[j];
        }
        if (S.out_indices) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32 bits = b[j];
                int r = pre[j];
                while (bits) {
                    const int bpos = __ffs(bits) - 1;
                    bits &= bits - 1;
                    const u32 key = ((u32)(w0 + j) << 5) + bpos;
                    int x, y, z, bq;
                    if (G.order == PCD_ROWS_YXZ) {
                        z = key % G.Do;
                        u32 t = key / G.Do;
                        x = t % G.Wo;
                        t /= G.Wo;
                        y = t % G.Ho;
                        bq = t / G.Ho;
                    } else {
                        x = key % G.Wo;
                        u32 t = key / G.Wo;
                        y = t % G.Ho;
                        t /= G.Ho;
                        z = t % G.Do;
                        bq = t / G.Do;
                    }
                    if (r < S.n_out) reinterpret_cast<int4 *>(S.out_indices)[r] = make_int4(bq, z, y, x);
                    ++r;
                }
            }
        }
    }
    const size_t nthreads = (size_t)(gridDim.x - (S.blk_cnt ? 1 : 0)) * 256, tid = (size_t)blk * 256 + threadIdx.x;
    if (S.fill_a) fill_ff(S.fill_a, S.fill_a_bytes, tid, nthreads);
    if (S.fill_b) fill_ff(S.fill_b, S.fill_b_bytes, tid, nthreads);
}

// perm != NULL: the launch also writes the parity-class permutation of the input rows (blk_off from class_offsets)
template <int ORD>
__global__ __launch_bounds__(256) void conv_fill_kernel(const int4 *__restrict__ idx, int n,
                                                        const int32_t *n_dev, ConvGeom G,
                                                        const u32 *__restrict__ bitmap,
                                                        const int *__restrict__ prefix, int n_out,
                                                        int32_t *__restrict__ nbr_in,
                                                        int32_t *__restrict__ nbr_out,
                                                        int *__restrict__ wave_cnt, int nwaves,
                                                        int *__restrict__ wsuper, int nws, int ncls,
                                                        const int *__restrict__ blk_off,
                                                        int32_t *__restrict__ perm) {
    __shared__ int wcnt[4][CLS_MAX];
    __shared__ int blk_sum[343];
    int i = blockIdx.x * 256 + threadIdx.x;
    bool live = i < eff_rows(n_dev, n);
    int4 c = live ? idx[i] : make_int4(0, 0, 0, 0);
    int wave = i >> 6;
    if (wave_cnt) {
        for (int q = threadIdx.x; q < G.K; q += 256) blk_sum[q] = 0;
        __syncthreads();
    }
    if (perm) {
        const int cls = live ? row_class(c, G.pd, G.ph, G.pw, G.sd, G.sh, G.sw) : -1;
        const int w = threadIdx.x >> 6;
        const u64 lt = (lane_id() == 0) ? 0ull : (~0ull >> (64 - lane_id()));
        int rank = 0;
        for (int q = 0; q < ncls; ++q) {
            u64 m = __ballot(cls == q);
            if (cls == q) rank = __popcll(m & lt);
            if (lane_id() == 0) wcnt[w][q] = __popcll(m);
        }
        __syncthreads();
        if (cls >= 0) {
            int before = 0;
            for (int ww = 0; ww < w; ++ww) before += wcnt[ww][cls];
            perm[blk_off[(size_t)cls * gridDim.x + blockIdx.x] + before + rank] = i;
        }
    }
    int k = 0;
    for (int a = 0; a < G.kd; ++a)
        for (int bq = 0; bq < G.kh; ++bq)
            for (int cq = 0; cq < G.kw; ++cq, ++k) {
                int o = -1;
                u32 key;
                if (live && out_cell<ORD>(G, c, a, bq, cq, key)) {
                    u32 w = key >> 5;
                    o = prefix[w] + __popc(bitmap[w] & ((1u << (key & 31)) - 1u));
                    if (o < n_out) nbr_out[(size_t)k * n_out + o] = i; else o = -1;
                }
                if (live) nbr_in[(size_t)k * n + i] = o;
                if (wave_cnt) publish_wave_count(wave_cnt, blk_sum, k, wave, nwaves, o >= 0);
            }
    if (wave_cnt) {
        __syncthreads();
        for (int q = threadIdx.x; q < G.K; q += 256)
            if (blk_sum[q]) atomicAdd(&wsuper[(size_t)q * nws + (blockIdx.x >> 4)], blk_sum[q]);
    }
}

static u32 table_capacity(int n) {
    u32 cap = 1024;
    while (cap < 2u * (u32)(n > 0 ? n : 1)) cap <<= 1;
    return cap;
}

struct ConvWs {
    unsigned char *bytemap;
    u32 *bitmap;
    int *prefix;
    int *bsums;
    int *super;      // popcount sums of 64 scan blocks each      } zeroed together with the bytemap
    int *wsuper;     // [K][nws]: pair counts of 64 waves each     } (one fill: they follow it in the workspace)
    int *wave_cnt;
    int *blk_cnt;    // [CLS_MAX][n / 256]: parity-class rows per block
    size_t nwords;
    size_t zero_bytes;   // bytemap .. end of wsuper
    int nwaves, nws, nblk, nsuper;
};

static int conv_ws_layout(void *workspace, size_t bytes, int n, int batch, const ConvGeom &G,
                          ConvWs &L, size_t *need) {
    double vol = (double)batch * G.Do * G.Ho * G.Wo;
    if (vol >= 4294967295.0 || vol <= 0) return PCD_ERR_KEYSPACE;
    L.nwords = ((size_t)vol + 31) / 32;
    L.nwaves = pcd_div_up(n > 0 ? n : 1, 64);
    L.nws = pcd_div_up(L.nwaves, 64);
    L.nblk = (int)((L.nwords + PK_WORDS - 1) / PK_WORDS);
    L.nsuper = pcd_div_up(L.nblk, 64);
    WsCarver ws(workspace, bytes);
    L.bitmap = ws.take<u32>(L.nwords);           // (bitmap, prefix) first: pcd_rulebook_conv_rank_layout
    L.prefix = ws.take<int>(L.nwords + 1);
    const size_t zero_from = ws.off;
    L.bytemap = ws.take<unsigned char>(L.nwords * 32);
    L.super = ws.take<int>(L.nsuper);
    L.wsuper = ws.take<int>((size_t)G.K * L.nws);
    L.zero_bytes = ws.off - zero_from;
    L.bsums = ws.take<int>(L.nblk);
    L.wave_cnt = ws.take<int>((size_t)G.K * L.nwaves);
    L.blk_cnt = ws.take<int>((size_t)CLS_MAX * pcd_div_up(n > 0 ? n : 1, 256));
    if (need) *need = ws.off;
    return ws.ok ? PCD_OK : PCD_ERR_WORKSPACE;
}

// The launches of a strided build.  count: fill, mark (+ class counts), pack + sums, scan (+ emit + 0xFF fills +
// class offsets when the outputs are known already); fill: neighbour tables (+ class permutation), pair lists.
struct ClsOut {
    int ncls, tile, vcap;
    int32_t *perm, *vstart;
};

static void conv_launch_mark(const int32_t *indices, int n, const int32_t *n_dev, const ConvGeom &G, const ConvWs &L,
                             const ClsOut *C, hipStream_t st) {
    pcd_fill(L.bytemap, 0, L.zero_bytes, st);
    if (n > 0) {
        if (G.order == PCD_ROWS_YXZ)
            conv_mark_kernel<PCD_ROWS_YXZ><<<pcd_div_up(n, 256), 256, 0, st>>>((const int4 *)indices, n, n_dev, G, L.bytemap,
                                                                               C ? C->ncls : 0, C ? L.blk_cnt : nullptr);
        else
            conv_mark_kernel<PCD_ROWS_ZYX><<<pcd_div_up(n, 256), 256, 0, st>>>((const int4 *)indices, n, n_dev, G, L.bytemap,
                                                                               C ? C->ncls : 0, C ? L.blk_cnt : nullptr);
    }
    conv_pack_sum_kernel<<<L.nblk, 256, 0, st>>>((const uint4 *)L.bytemap, L.nwords, L.bitmap, L.bsums, L.super,
                                                 L.nblk <= CONV_DIRECT_BLOCKS);
}

static void conv_launch_scan(int n, const ConvGeom &G, const ConvWs &L, int32_t *n_out_dev, int n_out,
                             int32_t *out_indices, int32_t *nbr_out, const ClsOut *C, hipStream_t st) {
    ScanSide S = {};
    size_t fill = 0;
    if (out_indices) {
        S.out_indices = out_indices;
        S.n_out = n_out;
        S.fill_a = nbr_out;
        S.fill_a_bytes = (size_t)G.K * n_out * sizeof(int32_t);
        fill = S.fill_a_bytes;
        if (C && n > 0) {
            S.fill_b = C->perm;
            S.fill_b_bytes = (size_t)C->vcap * sizeof(int32_t);
            S.blk_cnt = L.blk_cnt;
            S.cls_nblk = pcd_div_up(n, 256);
            S.ncls = C->ncls;
            S.cls_tile = C->tile;
            S.vstart = C->vstart;
            fill += S.fill_b_bytes;
        }
    }
    // enough threads that the fills are a handful of 16-byte stores each
    size_t blocks = fill / (256 * 16 * 8) + 1;
    if (blocks > 2048) blocks = 2048;
    if (blocks < (size_t)L.nblk) blocks = L.nblk;
    if (S.blk_cnt) ++blocks;
    conv_scan_emit_kernel<<<(unsigned)blocks, 256, 0, st>>>(L.bitmap, (int)L.nwords, L.nblk, L.bsums, L.super,
                                                            L.nblk <= CONV_DIRECT_BLOCKS, L.prefix, n_out_dev, G, S);
}

static void conv_launch_fill(const int32_t *indices, int n, const int32_t *n_dev, const ConvGeom &G, const ConvWs &L,
                             int n_out, int32_t *nbr_in, int32_t *nbr_out, int32_t *pairs, int32_t *pair_num,
                             int pad_pairs, const ClsOut *C, hipStream_t st) {
    if (G.order == PCD_ROWS_YXZ)
        conv_fill_kernel<PCD_ROWS_YXZ><<<pcd_div_up(n, 256), 256, 0, st>>>(
            (const int4 *)indices, n, n_dev, G, L.bitmap, L.prefix, n_out, nbr_in, nbr_out, pairs ? L.wave_cnt : nullptr,
            L.nwaves, L.wsuper, L.nws, C ? C->ncls : 0, L.blk_cnt, C ? C->perm : nullptr);
    else
        conv_fill_kernel<PCD_ROWS_ZYX><<<pcd_div_up(n, 256), 256, 0, st>>>(
            (const int4 *)indices, n, n_dev, G, L.bitmap, L.prefix, n_out, nbr_in, nbr_out, pairs ? L.wave_cnt : nullptr,
            L.nwaves, L.wsuper, L.nws, C ? C->ncls : 0, L.blk_cnt, C ? C->perm : nullptr);
    if (pairs) {
        if (pad_pairs) pcd_fill(pairs, 0xFF, (size_t)G.K * 2 * n * sizeof(int32_t), st);
        launch_pairs_fill_super(nbr_in, n, n_dev, G.K, 0, L.wave_cnt, L.nwaves, L.wsuper, L.nws, pairs, pair_num, st);
    }
}

}  // namespace

// =============================================================================================
extern "C" int pcd_conv_out_shape(const int *in_shape, const int *ks, const int *st, const int *pd,
                                  const int *dl, int *out_shape) {
    if (!in_shape || !ks || !st || !pd || !dl || !out_shape) return PCD_ERR_INVALID_ARG;
    for (int d = 0; d < 3; ++d) {
        int num = in_shape[d] + 2 * pd[d] - dl[d] * (ks[d] - 1) - 1;
        int q = num >= 0 ? num / st[d] : -((-num + st[d] - 1) / st[d]);
        out_shape[d] = q + 1;
    }
    return PCD_OK;
}

extern "C" size_t pcd_rulebook_subm_workspace_bytes(int n, int kvol) {
    if (n < 0 || kvol <= 0) return 0;
    int nwaves = pcd_div_up(n > 0 ? n : 1, 64);
    return ws_piece(2 * (size_t)table_capacity(n), sizeof(u64)) + 2 * ws_piece((size_t)kvol * nwaves, sizeof(int)) +
           ws_piece(kvol, sizeof(int)) + 2 * ws_piece((size_t)(n > 0 ? n : 1), sizeof(int32_t)) +
           ws_piece(1, sizeof(int));
}

extern "C" int pcd_rulebook_subm(const int32_t *indices, int n, int batch, const int *shape_host,
                                 const int *ksize_host, const int *dil_host, int32_t *nbr,
                                 int32_t *pairs, int32_t *pair_num, int pad_pairs,
                                 const int32_t *n_dev, void *workspace, size_t workspace_bytes,
                                 void *stream) {
    PCD_ENTER();
    if (n < 0 || batch <= 0 || !shape_host || !ksize_host || !dil_host) return PCD_ERR_INVALID_ARG;
    if (n > 0 && (pairs != nullptr) != (pair_num != nullptr)) return PCD_ERR_INVALID_ARG;
    const int one[3] = {1, 1, 1}, zero[3] = {0, 0, 0};
    ConvGeom G;
    int rc = make_geom(shape_host, ksize_host, one, zero, dil_host, G);
    if (rc != PCD_OK) return rc;
    if (!(G.kd & 1) || !(G.kh & 1) || !(G.kw & 1)) return PCD_ERR_UNSUPPORTED;  // SubM needs odd kernels
    if ((double)batch * G.D * G.H * G.W >= 4294967295.0) return PCD_ERR_KEYSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (pair_num) pcd_fill(pair_num, 0, G.K * sizeof(int32_t), st);
        return PCD_OK;
    }
    if (!indices || !nbr) return PCD_ERR_INVALID_ARG;
    WsCarver ws(workspace, workspace_bytes);
    u32 tcap = table_capacity(n);
    int nwaves = pcd_div_up(n, 64);
    u64 *table = ws.take<u64>(2 * (size_t)tcap);  // cell hash: tcap x 8 B; block hash: up to tcap x 16 B
    int *wave_cnt = ws.take<int>((size_t)G.K * nwaves);
    int *wave_off = ws.take<int>((size_t)G.K * nwaves);
    int *totals = ws.take<int>(G.K);
    int32_t *rowslot = ws.take<int32_t>(n);
    int32_t *rowlist = ws.take<int32_t>(n);
    int *blk_total = ws.take<int>(1);
    if (!ws.ok) return PCD_ERR_WORKSPACE;
    int nb = pcd_div_up(n, 256);
    const bool blocks = G.kd == 3 && G.kh == 3 && G.kw == 3 && G.dd == 1 && G.dh == 1 && G.dw == 1;
    if (blocks) {
        // at most n non-empty blocks (typically n/6): tcap / 2 >= n slots, strictly more unless n is a power of
        // two -- a probe for a missing block needs at least one empty slot to terminate
        const u32 bcap = (tcap / 2 > (u32)n) ? tcap / 2 : tcap;
        const int Dz = (G.D + 3) / 4, Hy = (G.H + 3) / 4, Wx = (G.W + 3) / 4;
        Blk *bt = (Blk *)table;
        pcd_fill(bt, 0, (size_t)bcap * sizeof(Blk), st);
        pcd_fill(blk_total, 0, sizeof(int), st);
        blk_insert_kernel<<<nb, 256, 0, st>>>((const int4 *)indices, n, n_dev, Dz, Hy, Wx, bt, bcap - 1, rowslot);
        blk_base_kernel<<<pcd_div_up((int)bcap, 4096), 1024, 0, st>>>(bt, bcap, blk_total);
        blk_fill_kernel<<<nb, 256, 0, st>>>((const int4 *)indices, n, n_dev, bt, rowslot, rowlist);
        blk_probe_kernel<<<nb, 256, 0, st>>>((const int4 *)indices, n, n_dev, G, Dz, Hy, Wx, bt, bcap - 1, rowslot,
                                             rowlist, nbr, pairs ? wave_cnt : nullptr, nwaves);
    } else {
        pcd_fill(table, 0xFF, (size_t)tcap * sizeof(u64), st);
        hash_insert_kernel<<<nb, 256, 0, st>>>((const int4 *)indices, n, n_dev, G.D, G.H, G.W, table, tcap - 1);
        subm_probe_kernel<<<nb, 256, 0, st>>>((const int4 *)indices, n, n_dev, G, table, tcap - 1, nbr,
                                              pairs ? wave_cnt : nullptr, nwaves);
    }
    if (pairs) {
        scan_rows_kernel<<<G.K, 256, 0, st>>>(wave_cnt, wave_off, nwaves, totals, pair_num, 1);
        if (pad_pairs) pcd_fill(pairs, 0xFF, (size_t)G.K * 2 * n * sizeof(int32_t), st);
        launch_pairs_fill(nbr, n, n_dev, G.K, 1, wave_off, nwaves, pairs, st);
    }
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

// per (offset, wave of 64 rows): number of rows with a neighbour -- what the probe kernels emit as a by-product
__global__ __launch_bounds__(256) void nbr_wave_count_kernel(const int32_t *__restrict__ tbl, int n,
                                                             const int32_t *n_dev, int K, int nwaves,
                                                             int *__restrict__ wave_cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int wave = i >> 6;
    const int nn = eff_rows(n_dev, n);
    for (int k = 0; k < K; ++k) {
        const int o = (i < nn) ? tbl[(size_t)k * n + i] : -1;
        const u64 m = __ballot(o >= 0);
        if (lane_id() == 0 && wave < nwaves) wave_cnt[(size_t)k * nwaves + wave] = __popcll(m);
    }
}

extern "C" size_t pcd_rulebook_subm_pairs_workspace_bytes(int n, int kvol) {
    if (n < 0 || kvol <= 0) return 0;
    int nwaves = pcd_div_up(n > 0 ? n : 1, 64);
    return 2 * ws_piece((size_t)kvol * nwaves, sizeof(int)) + ws_piece(kvol, sizeof(int));
}

// indice_pairs / indice_pair_num of a SubM rulebook from its neighbour table alone (for rulebooks built with
// pairs == NULL whose pairs turn out to be needed later); same result as pcd_rulebook_subm's.
static int pairs_from_table(const int32_t *nbr, int n, int kvol, int flip, int32_t *pairs, int32_t *pair_num, int pad_pairs,
                            const int32_t *n_dev, void *workspace, size_t workspace_bytes, void *stream);

extern "C" int pcd_rulebook_subm_pairs(const int32_t *nbr, int n, int kvol, int32_t *pairs, int32_t *pair_num,
                                       int pad_pairs, const int32_t *n_dev, void *workspace, size_t workspace_bytes,
                                       void *stream) {
    PCD_ENTER();
    return pairs_from_table(nbr, n, kvol, 1, pairs, pair_num, pad_pairs, n_dev, workspace, workspace_bytes, stream);
}

// The same for a STRIDED rulebook built without pair lists (the training step reads its pairs off the parity classes:
// pcd_sparse_conv_wgrad_classes): indice_pairs / indice_pair_num from nbr_in [kvol][n] alone, same result as the build's.
// Workspace: pcd_rulebook_subm_pairs_workspace_bytes(n, kvol).
extern "C" int pcd_rulebook_conv_pairs(const int32_t *nbr_in, int n, int kvol, int32_t *pairs, int32_t *pair_num,
                                       int pad_pairs, const int32_t *n_dev, void *workspace, size_t workspace_bytes,
                                       void *stream) {
    PCD_ENTER();
    return pairs_from_table(nbr_in, n, kvol, 0, pairs, pair_num, pad_pairs, n_dev, workspace, workspace_bytes, stream);
}

static int pairs_from_table(const int32_t *nbr, int n, int kvol, int flip, int32_t *pairs, int32_t *pair_num, int pad_pairs,
                            const int32_t *n_dev, void *workspace, size_t workspace_bytes, void *stream) {
    if (n < 0 || kvol <= 0 || !pair_num || (n > 0 && (!nbr || !pairs))) return PCD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        pcd_fill(pair_num, 0, kvol * sizeof(int32_t), st);
        return PCD_OK;
    }
    WsCarver ws(workspace, workspace_bytes);
    const int nwaves = pcd_div_up(n, 64);
    int *wave_cnt = ws.take<int>((size_t)kvol * nwaves);
    int *wave_off = ws.take<int>((size_t)kvol * nwaves);
    int *totals = ws.take<int>(kvol);
    if (!ws.ok) return PCD_ERR_WORKSPACE;
    nbr_wave_count_kernel<<<pcd_div_up(n, 256), 256, 0, st>>>(nbr, n, n_dev, kvol, nwaves, wave_cnt);
    scan_rows_kernel<<<kvol, 256, 0, st>>>(wave_cnt, wave_off, nwaves, totals, pair_num, flip);
    if (pad_pairs) pcd_fill(pairs, 0xFF, (size_t)kvol * 2 * n * sizeof(int32_t), st);
    launch_pairs_fill(nbr, n, n_dev, kvol, flip, wave_off, nwaves, pairs, st);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" size_t pcd_rulebook_subm_ranked_workspace_bytes(int n, int kvol) {
    if (n < 0 || kvol <= 0) return 0;
    int nwaves = pcd_div_up(n > 0 ? n : 1, 64);
    return 2 * ws_piece((size_t)kvol * nwaves, sizeof(int)) + ws_piece(kvol, sizeof(int));
}

static int subm_ranked_impl(const int32_t *indices, int n, int batch, const int *shape_host,
                           const int *ksize_host, const int *dil_host, const uint32_t *bitmap,
                           const int32_t *prefix, int32_t *nbr, int32_t *pairs, int32_t *pair_num,
                           int pad_pairs, const int32_t *n_dev, void *workspace,
                           size_t workspace_bytes, void *stream, int prefix_words, int row_order) {
    PCD_ENTER();
    if (n < 0 || batch <= 0 || !shape_host || !ksize_host || !dil_host) return PCD_ERR_INVALID_ARG;
    if (row_order != PCD_ROWS_ZYX && row_order != PCD_ROWS_YXZ) return PCD_ERR_INVALID_ARG;
    if (n > 0 && (pairs != nullptr) != (pair_num != nullptr)) return PCD_ERR_INVALID_ARG;
    const int one[3] = {1, 1, 1}, zero[3] = {0, 0, 0};
    ConvGeom G;
    int rc = make_geom(shape_host, ksize_host, one, zero, dil_host, G);
    if (rc != PCD_OK) return rc;
    if (G.kd != 3 || G.kh != 3 || G.kw != 3) return PCD_ERR_UNSUPPORTED;  // use pcd_rulebook_subm
    if ((double)batch * G.D * G.H * G.W >= 4294967295.0) return PCD_ERR_KEYSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (pair_num) pcd_fill(pair_num, 0, G.K * sizeof(int32_t), st);
        return PCD_OK;
    }
    if (!indices || !nbr || !bitmap || !prefix) return PCD_ERR_INVALID_ARG;
    WsCarver ws(workspace, workspace_bytes);
    int nwaves = pcd_div_up(n, 64);
    int *wave_cnt = ws.take<int>((size_t)G.K * nwaves);
    int *wave_off = ws.take<int>((size_t)G.K * nwaves);
    int *totals = ws.take<int>(G.K);
    if (!ws.ok) return PCD_ERR_WORKSPACE;
    int nb = pcd_div_up(n, 256);
    int *wc = pairs ? wave_cnt : nullptr;
    const int4 *idx4 = (const int4 *)indices;
    if (row_order == PCD_ROWS_YXZ) {
        if (prefix_words == 4)
            subm_rank_kernel<3, 3, 3, 4, PCD_ROWS_YXZ><<<nb, 256, 0, st>>>(idx4, n, n_dev, G, bitmap, prefix, nbr, wc, nwaves);
        else
            subm_rank_kernel<3, 3, 3, 1, PCD_ROWS_YXZ><<<nb, 256, 0, st>>>(idx4, n, n_dev, G, bitmap, prefix, nbr, wc, nwaves);
    } else {
        if (prefix_words == 4)
            subm_rank_kernel<3, 3, 3, 4><<<nb, 256, 0, st>>>(idx4, n, n_dev, G, bitmap, prefix, nbr, wc, nwaves);
        else
            subm_rank_kernel<3, 3, 3, 1><<<nb, 256, 0, st>>>(idx4, n, n_dev, G, bitmap, prefix, nbr, wc, nwaves);
    }
    if (pairs) {
        scan_rows_kernel<<<G.K, 256, 0, st>>>(wave_cnt, wave_off, nwaves, totals, pair_num, 1);
        if (pad_pairs) pcd_fill(pairs, 0xFF, (size_t)G.K * 2 * n * sizeof(int32_t), st);
        launch_pairs_fill(nbr, n, n_dev, G.K, 1, wave_off, nwaves, pairs, st);
    }
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_rulebook_subm_ranked(const int32_t *indices, int n, int batch, const int *shape_host,
                                        const int *ksize_host, const int *dil_host, const uint32_t *bitmap,
                                        const int32_t *prefix, int32_t *nbr, int32_t *pairs, int32_t *pair_num,
                                        int pad_pairs, const int32_t *n_dev, void *workspace,
                                        size_t workspace_bytes, void *stream, int row_order) {
    return subm_ranked_impl(indices, n, batch, shape_host, ksize_host, dil_host, bitmap, prefix, nbr, pairs, pair_num,
                            pad_pairs, n_dev, workspace, workspace_bytes, stream, 1, row_order);
}

extern "C" int pcd_rulebook_subm_ranked4(const int32_t *indices, int n, int batch, const int *shape_host,
                                         const int *ksize_host, const int *dil_host, const uint32_t *bitmap,
                                         const int32_t *prefix, int32_t *nbr, int32_t *pairs, int32_t *pair_num,
                                         int pad_pairs, const int32_t *n_dev, void *workspace,
                                         size_t workspace_bytes, void *stream, int row_order) {
    return subm_ranked_impl(indices, n, batch, shape_host, ksize_host, dil_host, bitmap, prefix, nbr, pairs, pair_num,
                            pad_pairs, n_dev, workspace, workspace_bytes, stream, 4, row_order);
}

extern "C" int pcd_rulebook_conv_rank_layout(int n, int batch, const int *in_shape_host, const int *ksize_host,
                                             const int *stride_host, const int *pad_host, const int *dil_host,
                                             size_t *bitmap_offset, size_t *prefix_offset, size_t *nwords) {
    ConvGeom G;
    if (n < 0 || batch <= 0 || !in_shape_host || !ksize_host || !stride_host || !pad_host || !dil_host ||
        !bitmap_offset || !prefix_offset || !nwords)
        return PCD_ERR_INVALID_ARG;
    int rc = make_geom(in_shape_host, ksize_host, stride_host, pad_host, dil_host, G);
    if (rc != PCD_OK) return rc;
    ConvWs L;
    size_t need = 0;
    conv_ws_layout(nullptr, 0, n, batch, G, L, &need);
    // conv_ws_layout takes the bitmap first, then the prefix array (each piece 256-byte aligned)
    *bitmap_offset = 0;
    *prefix_offset = ws_piece(L.nwords, sizeof(u32));
    *nwords = L.nwords;
    return PCD_OK;
}

// ---------------------------------------------------------------------------------------------
// Parity classes of the INPUT rows of a strided conv.  Input coordinate c reaches an output cell through kernel
// index k only if (c + p - k*d) is a multiple of the stride, so the residues ((c + p) mod s) of the three axes pick
// the 1..8 offsets (of 27 for k = 3, s = 2) a row can use at all.  The data gradient of a strided conv run over
// rows in their natural order executes all K offsets for every tile although 66 % of the (tile, offset) steps are
// empty; grouped by residue class every tile runs only its class's offsets.
//   perm   [vcap] : virtual row -> input row, -1 = padding; class segments start at multiples of `tile`
//   vstart [ncls + 1] (device): first virtual row of every class, vstart[ncls] = end
// The order inside a class is the input-row order (stable), so results do not depend on scheduling.
__global__ __launch_bounds__(256) void cls_count_kernel(const int4 *__restrict__ idx, int n, const int32_t *n_dev,
                                                        ConvGeom G, int ncls, int *__restrict__ blk_cnt) {
    __shared__ int cnt[CLS_MAX];
    int i = blockIdx.x * 256 + threadIdx.x;
    int cls = -1;
    if (i < eff_rows(n_dev, n)) cls = row_class(idx[i], G.pd, G.ph, G.pw, G.sd, G.sh, G.sw);
    block_class_counts(cls, ncls, cnt, blk_cnt);
}

__global__ __launch_bounds__(256) void cls_offsets_kernel(int *blk_cnt, int nblk, int ncls, int tile, int *vstart) {
    __shared__ int tot[CLS_MAX];
    __shared__ int start[CLS_MAX + 1];
    class_offsets(blk_cnt, nblk, ncls, tile, vstart, tot, start);
}

__global__ __launch_bounds__(256) void cls_fill_kernel(const int4 *__restrict__ idx, int n, const int32_t *n_dev,
                                                       ConvGeom G, int ncls, const int *__restrict__ blk_off,
                                                       int32_t *__restrict__ perm) {
    __shared__ int wcnt[4][CLS_MAX];
    int i = blockIdx.x * 256 + threadIdx.x;
    int cls = -1;
    if (i < eff_rows(n_dev, n)) cls = row_class(idx[i], G.pd, G.ph, G.pw, G.sd, G.sh, G.sw);
    const int w = threadIdx.x >> 6;
    const u64 lt = (lane_id() == 0) ? 0ull : (~0ull >> (64 - lane_id()));
    int rank = 0;
    for (int q = 0; q < ncls; ++q) {
        u64 m = __ballot(cls == q);
        if (cls == q) rank = __popcll(m & lt);
        if (lane_id() == 0) wcnt[w][q] = __popcll(m);
    }
    __syncthreads();
    if (cls >= 0) {
        int before = 0;
        for (int ww = 0; ww < w; ++ww) before += wcnt[ww][cls];
        perm[blk_off[(size_t)cls * gridDim.x + blockIdx.x] + before + rank] = i;
    }
}

extern "C" size_t pcd_rulebook_conv_classes_workspace_bytes(int n) {
    if (n < 0) return 0;
    return ws_piece((size_t)CLS_MAX * pcd_div_up(n > 0 ? n : 1, 256), sizeof(int));
}

extern "C" int pcd_rulebook_conv_classes(const int32_t *indices, int n, const int *stride_host, const int *pad_host,
                                         int tile, int32_t *perm, int vcap, int32_t *vstart_dev,
                                         const int32_t *n_dev, void *workspace, size_t workspace_bytes,
                                         void *stream) {
    PCD_ENTER();
    if (n < 0 || !stride_host || !pad_host || tile <= 0 || !perm || !vstart_dev) return PCD_ERR_INVALID_ARG;
    ConvGeom G = {};
    G.sd = stride_host[0]; G.sh = stride_host[1]; G.sw = stride_host[2];
    G.pd = pad_host[0]; G.ph = pad_host[1]; G.pw = pad_host[2];
    if (G.sd <= 0 || G.sh <= 0 || G.sw <= 0 || G.pd < 0 || G.ph < 0 || G.pw < 0) return PCD_ERR_INVALID_ARG;
    const int ncls = G.sd * G.sh * G.sw;
    if (ncls > CLS_MAX) return PCD_ERR_UNSUPPORTED;
    if (vcap < (n + tile - 1) / tile * tile + ncls * tile) return PCD_ERR_INVALID_ARG;
    if (n > 0 && !indices) return PCD_ERR_INVALID_ARG;
    const int nblk = pcd_div_up(n > 0 ? n : 1, 256);
    WsCarver ws(workspace, workspace_bytes);
    int *blk_cnt = ws.take<int>((size_t)CLS_MAX * nblk);
    if (!ws.ok) return PCD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    pcd_fill(perm, 0xFF, (size_t)vcap * sizeof(int32_t), st);
    cls_count_kernel<<<nblk, 256, 0, st>>>((const int4 *)indices, n, n_dev, G, ncls, blk_cnt);
    cls_offsets_kernel<<<1, 256, 0, st>>>(blk_cnt, nblk, ncls, tile, vstart_dev);
    cls_fill_kernel<<<nblk, 256, 0, st>>>((const int4 *)indices, n, n_dev, G, ncls, blk_cnt, perm);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" size_t pcd_rulebook_conv_workspace_bytes(int n, int batch, const int *in_shape_host,
                                                    const int *ksize_host, const int *stride_host,
                                                    const int *pad_host, const int *dil_host) {
    ConvGeom G;
    if (n < 0 || batch <= 0 || !in_shape_host || !ksize_host || !stride_host || !pad_host || !dil_host)
        return 0;
    if (make_geom(in_shape_host, ksize_host, stride_host, pad_host, dil_host, G) != PCD_OK) return 0;
    ConvWs L;
    size_t need = 0;
    conv_ws_layout(nullptr, 0, n, batch, G, L, &need);
    return need;
}

extern "C" int pcd_rulebook_conv_count(const int32_t *indices, int n, int batch,
                                       const int *in_shape_host, const int *ksize_host,
                                       const int *stride_host, const int *pad_host,
                                       const int *dil_host, int32_t *n_out_dev, const int32_t *n_dev,
                                       void *workspace, size_t workspace_bytes, void *stream, int row_order) {
    PCD_ENTER();
    if (n < 0 || batch <= 0 || !n_out_dev) return PCD_ERR_INVALID_ARG;
    if (row_order != PCD_ROWS_ZYX && row_order != PCD_ROWS_YXZ) return PCD_ERR_INVALID_ARG;
    if (!in_shape_host || !ksize_host || !stride_host || !pad_host || !dil_host)
        return PCD_ERR_INVALID_ARG;
    ConvGeom G;
    int rc = make_geom(in_shape_host, ksize_host, stride_host, pad_host, dil_host, G);
    if (rc != PCD_OK) return rc;
    if (G.Do <= 0 || G.Ho <= 0 || G.Wo <= 0) return PCD_ERR_INVALID_ARG;
    G.order = row_order;
    ConvWs L;
    rc = conv_ws_layout(workspace, workspace_bytes, n, batch, G, L, nullptr);
    if (rc != PCD_OK) return rc;
    if (n > 0 && !indices) return PCD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    conv_launch_mark(indices, n, n_dev, G, L, nullptr, st);
    conv_launch_scan(n, G, L, n_out_dev, 0, nullptr, nullptr, nullptr, st);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_rulebook_conv_fill(const int32_t *indices, int n, int batch,
                                      const int *in_shape_host, const int *ksize_host,
                                      const int *stride_host, const int *pad_host,
                                      const int *dil_host, int n_out, int32_t *out_indices,
                                      int32_t *nbr_in, int32_t *nbr_out, int32_t *pairs,
                                      int32_t *pair_num, int pad_pairs, const int32_t *n_dev,
                                      void *workspace, size_t workspace_bytes, void *stream, int row_order) {
    PCD_ENTER();
    if (n < 0 || batch <= 0 || n_out < 0) return PCD_ERR_INVALID_ARG;
    if (row_order != PCD_ROWS_ZYX && row_order != PCD_ROWS_YXZ) return PCD_ERR_INVALID_ARG;
    if (!in_shape_host || !ksize_host || !stride_host || !pad_host || !dil_host)
        return PCD_ERR_INVALID_ARG;
    if (n > 0 && (pairs != nullptr) != (pair_num != nullptr)) return PCD_ERR_INVALID_ARG;
    ConvGeom G;
    int rc = make_geom(in_shape_host, ksize_host, stride_host, pad_host, dil_host, G);
    if (rc != PCD_OK) return rc;
    G.order = row_order;
    ConvWs L;
    rc = conv_ws_layout(workspace, workspace_bytes, n, batch, G, L, nullptr);
    if (rc != PCD_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0 || n_out == 0) {
        if (pair_num) pcd_fill(pair_num, 0, G.K * sizeof(int32_t), st);
        if (n > 0 && nbr_in) pcd_fill(nbr_in, 0xFF, (size_t)G.K * n * sizeof(int32_t), st);
        return PCD_OK;
    }
    if (!indices || !out_indices || !nbr_in || !nbr_out) return PCD_ERR_INVALID_ARG;
    // the scan again, now with its outputs (prefix comes out the same); the pair-count sums restart from zero so
    // that a second fill over the same workspace stays correct
    if (pairs) pcd_fill(L.wsuper, 0, (size_t)G.K * L.nws * sizeof(int), st);
    conv_launch_scan(n, G, L, nullptr, n_out, out_indices, nbr_out, nullptr, st);
    conv_launch_fill(indices, n, n_dev, G, L, n_out, nbr_in, nbr_out, pairs, pair_num, pad_pairs, nullptr, st);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

// count + fill + parity classes in one call when the number of output rows is bounded by a capacity known on the
// host (static plans): 6 launches (5 without pair lists), nothing read back.
extern "C" int pcd_rulebook_conv_build(const int32_t *indices, int n, int batch, const int *in_shape_host,
                                       const int *ksize_host, const int *stride_host, const int *pad_host,
                                       const int *dil_host, int n_out_cap, int32_t *n_out_dev, int32_t *out_indices,
                                       int32_t *nbr_in, int32_t *nbr_out, int32_t *pairs, int32_t *pair_num,
                                       int pad_pairs, int cls_tile, int32_t *perm, int vcap, int32_t *vstart_dev,
                                       const int32_t *n_dev, void *workspace, size_t workspace_bytes, void *stream,
                                       int row_order) {
    PCD_ENTER();
    if (n <= 0 || batch <= 0 || n_out_cap <= 0 || !n_out_dev) return PCD_ERR_INVALID_ARG;
    if (row_order != PCD_ROWS_ZYX && row_order != PCD_ROWS_YXZ) return PCD_ERR_INVALID_ARG;
    if (!in_shape_host || !ksize_host || !stride_host || !pad_host || !dil_host) return PCD_ERR_INVALID_ARG;
    if (!indices || !out_indices || !nbr_in || !nbr_out) return PCD_ERR_INVALID_ARG;
    if ((pairs != nullptr) != (pair_num != nullptr)) return PCD_ERR_INVALID_ARG;
    ConvGeom G;
    int rc = make_geom(in_shape_host, ksize_host, stride_host, pad_host, dil_host, G);
    if (rc != PCD_OK) return rc;
    if (G.Do <= 0 || G.Ho <= 0 || G.Wo <= 0) return PCD_ERR_INVALID_ARG;
    G.order = row_order;
    ClsOut C = {G.sd * G.sh * G.sw, cls_tile, vcap, perm, vstart_dev};
    const bool classes = perm != nullptr;
    if (classes) {
        if (C.ncls > CLS_MAX) return PCD_ERR_UNSUPPORTED;
        if (cls_tile <= 0 || !vstart_dev || vcap < (n + cls_tile - 1) / cls_tile * cls_tile + C.ncls * cls_tile)
            return PCD_ERR_INVALID_ARG;
    }
    ConvWs L;
    rc = conv_ws_layout(workspace, workspace_bytes, n, batch, G, L, nullptr);
    if (rc != PCD_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    conv_launch_mark(indices, n, n_dev, G, L, classes ? &C : nullptr, st);
    conv_launch_scan(n, G, L, n_out_dev, n_out_cap, out_indices, nbr_out, classes ? &C : nullptr, st);
    conv_launch_fill(indices, n, n_dev, G, L, n_out_cap, nbr_in, nbr_out, pairs, pair_num, pad_pairs,
                     classes ? &C : nullptr, st);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
