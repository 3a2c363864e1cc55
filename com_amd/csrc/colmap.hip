// Column-map rulebook builds for levels whose rows are numbered z-fastest (PCD_ROWS_YXZ) -- O(rows) + O(BEV cells / 32)
// instead of O(volume of the key space).
//
// Replaces the same spconv get_indice_pairs calls as rulebook.hip (pcdet/models/backbones_3d/spconv_backbone.py:12-15,
// 199-229; definitions SURVEY.md A.4).  In (b, y, x, z) order the rows of one BEV cell -- a COLUMN -- are consecutive and
// ordered by z, so the coordinate -> row map of a level needs no bitmap over its (b, y, x, z) key space (371 M cells at
// level 1 of a 4-frame Waymo batch):
//     cw[word]  = { occupancy bits of 32 BEV cells (b, y, x), number of occupied cells in front of the word }     8 B
//     cr[col]   = { z mask lo, z mask hi, first row of the column, rows of the column }                           16 B
//     row(b, z, y, x) = cr[col].start + popcount(zmask & below(z)),  col = cw[key >> 5].prefix + popcount(bits below)
// 2.3 MB + 16 B per column at level 1 (L2-resident) against 46 MB + 11.6 MB for the flat bitmap and its prefixes.
//
// A strided conv's OUTPUT map follows from the input map alone: output column (b, oy, ox) exists iff one of its kh x kw
// input columns does; its z mask is the OR of their masks shifted by the padding, smeared over the kd kernel taps and
// (stride 2) compressed to the even bits.  BEV keys use a row pitch rounded up to 32 cells, so a word never straddles two BEV
// rows and the OCCUPANCY of 32 output cells is a few shifts over 4 input words per kernel row (one lane per output word);
// the occupied cells of a wave's 64 words (5-10 % of the cells) are then compacted so that every lane gathers the z masks of
// ONE occupied cell.  Two passes (count, emit) around per-wave sums.  No atomics, no zero-fill of a map of the output
// volume, no scatter: every table entry is written exactly once by the thread that owns its row.
#include "rulebook_common.h"
#include "colmap_common.h"
#include "cls_table.h"

namespace {

// A wave takes 2^wshift output words (32 BEV cells each), one per lane: 64 at the large levels, fewer where that would leave
// too few waves to hide the gathers' latency (level 5 of a 4-frame batch is 4.5 k words in all).
static inline int cm_wshift(int nwords_out) {
    int s = 6;
    while (s > 2 && (nwords_out >> s) < 2048) --s;
    return s;
}
// ---- level 1: the map of a row set given in (b, y, x, z) order -----------------------------------------------
__device__ __forceinline__ bool same_column(int4 a, int4 c) { return a.x == c.x && a.z == c.z && a.w == c.w; }

__global__ __launch_bounds__(256) void cm_rows_mark_kernel(const int4 *__restrict__ idx, int n, const int32_t *n_dev,
                                                           int H, int P, u32 *__restrict__ bits) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= eff_rows(n_dev, n)) return;
    const int4 c = idx[i];
    if (i > 0 && same_column(idx[i - 1], c)) return;
    const u32 key = bev_key(c.x, c.z, c.w, H, P);
    atomicOr(bits + (key >> 5), 1u << (key & 31u));      // one per column (a third of the rows), no return value
}

// head rows write their column's record: z mask and length from the rows that follow
__global__ __launch_bounds__(256) void cm_rows_fill_kernel(const int4 *__restrict__ idx, int n, const int32_t *n_dev,
                                                           int H, int P, const uint2 *__restrict__ cw,
                                                           uint4 *__restrict__ cr, int ncol_cap, int *__restrict__ ncols) {
    const int nn = eff_rows(n_dev, n);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && ncols) ncols[1] = nn;
    if (i >= nn) return;
    const int4 c = idx[i];
    if (i > 0 && same_column(idx[i - 1], c)) return;
    const u32 key = bev_key(c.x, c.z, c.w, H, P);
    const int col = cm_col(cw[key >> 5], key, ncol_cap);
    if (col < 0) return;
    u64 zm = 1ull << c.y;
    int cnt = 1;
    for (int j = i + 1; j < nn; ++j) {
        const int4 d = idx[j];
        if (!same_column(d, c)) break;
        zm |= 1ull << d.y;
        ++cnt;
    }
    cr[col] = make_uint4((u32)zm, (u32)(zm >> 32), (u32)i, (u32)cnt);
}

// ---- SubM 3x3x3 from the map ---------------------------------------------------------------------------------
// One thread per row: the nine columns (dy, dx) of its neighbourhood, each looked up once (word, then record) and serving
// the three z-neighbours.  Neighbouring rows of a wave share columns: the loads hit the same few lines.
__global__ __launch_bounds__(256) void cm_subm_kernel(const int4 *__restrict__ idx, int n, const int32_t *n_dev,
                                                      int D, int H, int W, int P, const uint2 *__restrict__ cw,
                                                      const uint4 *__restrict__ cr, int ncol_cap,
                                                      int32_t *__restrict__ nbr, int *__restrict__ wave_cnt, int nwaves) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    const int nn = eff_rows(n_dev, n);
    const bool live = o < nn;
    const int4 c = live ? idx[o] : make_int4(0, 0, 0, 0);
    const int wave = o >> 6;
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
            const int k = dz * 9 + q;
            const int z = c.y + dz - 1;
            int row = -1;
            if (hit[q] && z >= 0 && z < D) {
                row = cm_row((u64)r[q].x | ((u64)r[q].y << 32), (int)r[q].z, z);
                if (row >= nn) row = -1;
            }
            if (k == 13 && live) row = o;
            if (live) nbr[(size_t)k * n + o] = row;
            if (wave_cnt) {
                const u64 m = __ballot(row >= 0);
                if (lane_id() == 0 && wave < nwaves) wave_cnt[(size_t)k * nwaves + wave] = __popcll(m);
            }
        }
}

// ---- strided conv from the input map --------------------------------------------------------------------------
struct CmGeom {
    int D, H, W, Do, Ho, Wo;
    int kd, sd, sh, sw, pd, ph, pw;
    int P, Po;            // BEV row pitches (cells) of the input / output map
    int nwords_out;       // batch * Ho * Po / 32
    int wshift;           // log2(output words per wave)
    int ncol_cap_in, ncol_cap_out;
};

__device__ __forceinline__ u32 compress_even(u64 t) {
    t &= 0x5555555555555555ull;
    t = (t | (t >> 1)) & 0x3333333333333333ull;
    t = (t | (t >> 2)) & 0x0f0f0f0f0f0f0f0full;
    t = (t | (t >> 4)) & 0x00ff00ff00ff00ffull;
    t = (t | (t >> 8)) & 0x0000ffff0000ffffull;
    t = (t | (t >> 16)) & 0x00000000ffffffffull;
    return (u32)t;
}

// z mask of the output column fed by the OR `m` of its input columns' masks:  out bit oz = OR_kz m[oz * sd - pd + kz]
__device__ __forceinline__ u64 squash_z(u64 m, const CmGeom &G) {
    const u64 m1 = m << G.pd;
    u64 t = m1;
    for (int kz = 1; kz < G.kd; ++kz) t |= m1 >> kz;
    const u64 o = G.sd == 2 ? (u64)compress_even(t) : t;
    return o & ((1ull << G.Do) - 1ull);
}

// OR of the z masks of the KH x KW input columns of output cell (b, oy, ox) (0 = the output column does not exist)
template <int KH, int KW>
__device__ __forceinline__ u64 in_union(const CmGeom &G, const uint2 *__restrict__ cw, const uint4 *__restrict__ cr, int b,
                                        int oy, int ox) {
    uint2 w[KH * KW];
    u32 key[KH * KW];
    bool in[KH * KW];
#pragma unroll
    for (int q = 0; q < KH * KW; ++q) {
        const int y = oy * G.sh - G.ph + q / KW, x = ox * G.sw - G.pw + q % KW;
        in[q] = y >= 0 && y < G.H && x >= 0 && x < G.W;
        key[q] = in[q] ? bev_key(b, y, x, G.H, G.P) : 0u;
        w[q] = cw[key[q] >> 5];
    }
    u64 m = 0;
    bool any = false;
#pragma unroll
    for (int q = 0; q < KH * KW; ++q) {
        in[q] = in[q] && ((w[q].x >> (key[q] & 31u)) & 1u);
        any = any || in[q];
    }
    if (!any) return 0;
    uint4 r[KH * KW];
#pragma unroll
    for (int q = 0; q < KH * KW; ++q) {
        const int col = in[q] ? cm_col(w[q], key[q], G.ncol_cap_in) : -1;
        in[q] = col >= 0;
        r[q] = cr[in[q] ? col : 0];
    }
#pragma unroll
    for (int q = 0; q < KH * KW; ++q)
        if (in[q]) m |= (u64)r[q].x | ((u64)r[q].y << 32);
    return m;
}

// in_union for NB cells at once: all their word loads, then all their record loads (a wave with more than 64 occupied cells
// walks them 64 * NB at a time -- its run time is a chain of dependent round trips, one pair per batch otherwise)
template <int KH, int KW, int NB>
__device__ __forceinline__ void in_union_n(const CmGeom &G, const uint2 *__restrict__ cw, const uint4 *__restrict__ cr,
                                           const int (&b)[NB], const int (&oy)[NB], const int (&ox)[NB], const bool (&act)[NB],
                                           u64 (&m)[NB]) {
    uint2 w[NB][KH * KW];
    u32 key[NB][KH * KW];
    bool in[NB][KH * KW];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int q = 0; q < KH * KW; ++q) {
            const int y = oy[n] * G.sh - G.ph + q / KW, x = ox[n] * G.sw - G.pw + q % KW;
            in[n][q] = act[n] && y >= 0 && y < G.H && x >= 0 && x < G.W;
            key[n][q] = in[n][q] ? bev_key(b[n], y, x, G.H, G.P) : 0u;
            w[n][q] = cw[key[n][q] >> 5];
        }
    uint4 r[NB][KH * KW];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int q = 0; q < KH * KW; ++q) {
            const int col = in[n][q] ? cm_col(w[n][q], key[n][q], G.ncol_cap_in) : -1;
            in[n][q] = col >= 0;
            r[n][q] = cr[in[n][q] ? col : 0];
        }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        m[n] = 0;
#pragma unroll
        for (int q = 0; q < KH * KW; ++q)
            if (in[n][q]) m[n] |= (u64)r[n][q].x | ((u64)r[n][q].y << 32);
    }
}

constexpr int CM_NB = 1;    // batches of 64 cells a wave has in flight (count and emit passes); 2 measured: no gain (288 against 279 us, level 2 at B = 32)

struct CmConvSide {         // by-products of the count launch
    int *zero;              // words to zero (wsuper of the pair lists), spread over the cell blocks
    int zero_words;
    const int4 *idx;        // parity classes: extra blocks count the classes of 256 input rows each
    int n;
    const int32_t *n_dev;
    int ncls;
    int *blk_cnt;
    int pd, ph, pw;
};

// Occupancy of the 32 output cells (b, oy, 32 xb ..) from the input map's words alone: per kernel row the four input words
// around input word xb * sw, as a 128-bit window; out bit j = OR_kx window[32 + j * sw - pw + kx].
template <int KH, int KW>
__device__ __forceinline__ u32 out_occupancy(const CmGeom &G, const uint2 *__restrict__ cw, int b, int oy, int xb) {
    const int wpr = G.P >> 5, wb = xb * G.sw;
    u32 w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int ky = 0; ky < KH; ++ky) {
        const int y = oy * G.sh - G.ph + ky;
        if (y < 0 || y >= G.H) continue;
        const uint2 *row = cw + (size_t)(b * G.H + y) * wpr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int wi = wb - 1 + j;
            if (wi >= 0 && wi < wpr) w[j] |= row[wi].x;
        }
    }
    const u64 lo = (u64)w[0] | ((u64)w[1] << 32), hi = (u64)w[2] | ((u64)w[3] << 32);
    u64 t = 0;
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) {
        const int sh = 32 - G.pw + kx;                    // 0 < sh < 64 (pw <= 31)
        t |= (lo >> sh) | (hi << (64 - sh));
    }
    u32 occ = G.sw == 2 ? compress_even(t) : (u32)t;
    const int valid = G.Wo - (xb << 5);                   // cells of the word inside the row
    if (valid < 32) occ &= (1u << (valid > 0 ? valid : 0)) - 1u;
    return occ;
}

// position of the k-th (0-based) set bit of v (k < popcount(v))
__device__ __forceinline__ int kth_set_bit(u32 v, int k) {
    int pos = 0;
#pragma unroll
    for (int half = 16; half >= 1; half >>= 1) {
        const int c = __popc(v & ((1u << half) - 1u));
        if (k >= c) {
            k -= c;
            v >>= half;
            pos += half;
        }
    }
    return pos;
}

// The occupied cells of a wave's 64 words, one per lane and round: lane `idx - base` of round `base` takes the idx-th
// occupied cell in cell order (word = last one whose exclusive count is <= idx, bit = the (idx - count)-th set one).
struct CmWaveCells {
    u32 *occ_s;           // LDS [64]: occupancy word of every lane
    int *excl_s;          // LDS [64]: occupied cells in front of the word, inside the wave
    int total;
};
__device__ __forceinline__ CmWaveCells cm_wave_cells(u32 occ, u32 *occ_s, int *excl_s) {
    const int c = __popc(occ);
    const int inc = wave_inclusive_scan(c);
    occ_s[lane_id()] = occ;
    excl_s[lane_id()] = inc - c;
    __syncthreads();                                      // (every wave of a cell block gets here; orders the LDS writes)
    CmWaveCells Wc = {occ_s, excl_s, __shfl(inc, 63)};
    return Wc;
}
__device__ __forceinline__ void cm_wave_cell(const CmWaveCells &Wc, int idx, int &word_lane, int &bit) {
    int lo = 0;                                           // largest l with excl_s[l] <= idx
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1)
        if (Wc.excl_s[lo + step] <= idx) lo += step;      // (lo + step <= 63)
    word_lane = lo;
    bit = kth_set_bit(Wc.occ_s[lo], idx - Wc.excl_s[lo]);
}

// pass 1: { occupied cells, rows } per wave of 64 output words
template <int KH, int KW>
__global__ __launch_bounds__(256) void cm_conv_count_kernel(CmGeom G, int ncellblk, const uint2 *__restrict__ cw,
                                                            const uint4 *__restrict__ cr, int2 *__restrict__ bsums,
                                                            CmConvSide S) {
    __shared__ int ccnt[CLS_MAX];
    __shared__ u32 occ_s[4][64];
    __shared__ int excl_s[4][64];
    if ((int)blockIdx.x >= ncellblk) {
        const int i = ((int)blockIdx.x - ncellblk) * 256 + threadIdx.x;
        int cls = -1;
        if (i < eff_rows(S.n_dev, S.n)) cls = row_class(S.idx[i], S.pd, S.ph, S.pw, G.sd, G.sh, G.sw);
        if (threadIdx.x < CLS_MAX) ccnt[threadIdx.x] = 0;
        __syncthreads();
        for (int q = 0; q < S.ncls; ++q) {
            const u64 m = __ballot(cls == q);
            if (lane_id() == 0 && m) atomicAdd(&ccnt[q], __popcll(m));
        }
        __syncthreads();
        const int nclsblk = (int)gridDim.x - ncellblk;
        if ((int)threadIdx.x < S.ncls) S.blk_cnt[(size_t)threadIdx.x * nclsblk + ((int)blockIdx.x - ncellblk)] = ccnt[threadIdx.x];
        return;
    }
    if (S.zero) {
        for (int e = blockIdx.x * 256 + threadIdx.x; e < S.zero_words; e += ncellblk * 256) S.zero[e] = 0;
    }
    const int wave = threadIdx.x >> 6;
    const int word0 = (blockIdx.x * 4 + wave) << G.wshift, word = word0 + lane_id();
    const int wpr = G.Po >> 5;
    u32 occ = 0;
    if (lane_id() < (1 << G.wshift) && word < G.nwords_out) {
        const int t = word / wpr;
        occ = out_occupancy<KH, KW>(G, cw, t / G.Ho, t % G.Ho, word % wpr);
    }
    const CmWaveCells Wc = cm_wave_cells(occ, occ_s[wave], excl_s[wave]);
    int rows = 0;
    for (int base = 0; base < Wc.total; base += 64 * CM_NB) {
        int cb[CM_NB], coy[CM_NB], cox[CM_NB];
        bool act[CM_NB];
        u64 m[CM_NB];
#pragma unroll
        for (int n = 0; n < CM_NB; ++n) {
            const int idx = base + n * 64 + lane_id();
            act[n] = idx < Wc.total;
            cb[n] = coy[n] = cox[n] = 0;
            if (act[n]) {
                int wl, bit;
                cm_wave_cell(Wc, idx, wl, bit);
                const int wd = word0 + wl, t = wd / wpr;
                cb[n] = t / G.Ho;
                coy[n] = t % G.Ho;
                cox[n] = ((wd % wpr) << 5) + bit;
            }
        }
        in_union_n<KH, KW, CM_NB>(G, cw, cr, cb, coy, cox, act, m);
#pragma unroll
        for (int n = 0; n < CM_NB; ++n)
            if (act[n]) rows += __popcll(squash_z(m[n], G));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) rows += __shfl_xor(rows, d, 64);
    if (lane_id() == 0) bsums[blockIdx.x * 4 + wave] = make_int2(Wc.total, rows);
}

// exclusive scan of the wave sums (only beyond "cm_direct_blocks" entries): one block, in place, totals -> bsums[nent]
__global__ __launch_bounds__(1024) void cm_spine2_kernel(int2 *bsums, int nent) {
    // chunks of 1024 entries, one per thread, the next chunk requested before this one is scanned; wave scans + the 16 wave
    // totals (double-buffered): one barrier per chunk (until round 6: 256 threads, five barriers per 256 entries)
    __shared__ int2 wtot[2][16];
    const int tid = threadIdx.x, wv = tid >> 6;
    int2 carry = make_int2(0, 0);
    int2 nxt = tid < nent ? bsums[tid] : make_int2(0, 0);
    for (int base = 0, it = 0; base < nent; base += 1024, ++it) {
        const int i = base + tid;
        const int2 v = nxt;
        nxt = i + 1024 < nent ? bsums[i + 1024] : make_int2(0, 0);
        const int ic = wave_inclusive_scan(v.x), ir = wave_inclusive_scan(v.y);
        if ((tid & 63) == 63) wtot[it & 1][wv] = make_int2(ic, ir);
        __syncthreads();
        int2 off = carry, tot = make_int2(0, 0);
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int2 s = wtot[it & 1][w];
            off.x += w < wv ? s.x : 0;
            off.y += w < wv ? s.y : 0;
            tot.x += s.x;
            tot.y += s.y;
        }
        if (i < nent) bsums[i] = make_int2(off.x + ic - v.x, off.y + ir - v.y);
        carry.x += tot.x;
        carry.y += tot.y;
    }
    if (tid == 0) bsums[nent] = carry;
}

// sum of entries [0, ent) (by the block itself, or read from the spine's exclusive prefixes)
__device__ __forceinline__ int2 cm_base2(const int2 *__restrict__ bsums, int ent, int spined, int *lds) {
    if (spined) return bsums[ent];
    int ac = 0, ar = 0;
    for (int j = threadIdx.x; j < ent; j += 256) {
        const int2 v = bsums[j];
        ac += v.x;
        ar += v.y;
    }
    const int tc = block_sum(ac, lds);
    const int tr = block_sum(ar, lds);
    return make_int2(tc, tr);
}

// totals only (the two-phase API: the host reads n_out before it allocates the outputs)
__global__ __launch_bounds__(256) void cm_total_kernel(const int2 *__restrict__ bsums, int nent, int spined,
                                                       int *__restrict__ n_out_dev) {
    __shared__ int lds[4];
    const int2 t = cm_base2(bsums, nent, spined, lds);
    if (threadIdx.x == 0) *n_out_dev = t.y;
}

struct CmEmitSide {
    void *fill_a;           // 0xFF fill (perm of the parity classes), spread over the cell blocks
    size_t fill_a_bytes;
    int *blk_cnt;           // parity classes: `ncls` extra blocks, one per class, turn the class's counts into exclusive prefixes
    int cls_nblk, ncls, cls_tile;
    int *cls_tot;           // [CLS_MAX] rows per class (the tables launch turns them into the class starts and writes vstart)
};

// blk_cnt[cls][0 .. nblk) -> exclusive prefix in place, tot[cls] = the sum: one block of 256 threads per class, 2048 counts per
// round (8 consecutive per thread).  (Until round 6 ONE extra block did all classes, a wave per class and 512 counts per
// round: 120 us at B = 32 -- the emit launch lasted exactly as long as that block, whatever its cell blocks did.)
__device__ void class_prefix_block(int *__restrict__ row, int nblk, int *__restrict__ tot, int *lds /* [4] */) {
    int carry = 0;
    for (int base = 0; base < nblk; base += 2048) {
        const int i0 = base + threadIdx.x * 8;
        int v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = i0 + j < nblk ? row[i0 + j] : 0;
        int sum = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = v[j];
            v[j] = sum;
            sum += t;
        }
        int total;
        const int ex = block_exclusive_scan(sum, lds, total);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (i0 + j < nblk) row[i0 + j] = carry + ex + v[j];
        carry += total;
    }
    if (threadIdx.x == 0) *tot = carry;
}

// pass 2: the output map (words, column records), the output coordinates, the row count
template <int KH, int KW>
__global__ __launch_bounds__(256) void cm_conv_emit_kernel(CmGeom G, int ncellblk, const uint2 *__restrict__ cw,
                                                           const uint4 *__restrict__ cr, const int2 *__restrict__ bsums,
                                                           int spined, uint2 *__restrict__ cw_out, uint4 *__restrict__ cr_out,
                                                           int *__restrict__ ncols_out, int32_t *__restrict__ out_indices,
                                                           int n_out, int *__restrict__ n_out_dev, CmEmitSide S) {
    __shared__ int lds[4];
    __shared__ u32 occ_s[4][64];
    __shared__ int excl_s[4][64];
    if ((int)blockIdx.x >= ncellblk) {
        const int q = (int)blockIdx.x - ncellblk;
        class_prefix_block(S.blk_cnt + (size_t)q * S.cls_nblk, S.cls_nblk, S.cls_tot + q, lds);
        return;
    }
    if (S.fill_a) fill_ff(S.fill_a, S.fill_a_bytes, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)ncellblk * 256);
    const int wave = threadIdx.x >> 6;
    int2 base = cm_base2(bsums, blockIdx.x * 4, spined, lds);
    if (blockIdx.x == 0 && (n_out_dev || ncols_out)) {
        const int2 t = cm_base2(bsums, ncellblk * 4, spined, lds);
        if (threadIdx.x == 0) {
            if (n_out_dev) *n_out_dev = t.y;
            if (ncols_out) {
                ncols_out[0] = t.x;
                ncols_out[1] = t.y;
            }
        }
    }
    if (spined) {
        base = bsums[blockIdx.x * 4 + wave];
    } else {
        for (int wv = 0; wv < wave; ++wv) {
            const int2 v = bsums[blockIdx.x * 4 + wv];
            base.x += v.x;
            base.y += v.y;
        }
    }
    const int word0 = (blockIdx.x * 4 + wave) << G.wshift, word = word0 + lane_id();
    const int wpr = G.Po >> 5;
    const bool mine = lane_id() < (1 << G.wshift) && word < G.nwords_out;
    u32 occ = 0;
    if (mine) {
        const int t = word / wpr;
        occ = out_occupancy<KH, KW>(G, cw, t / G.Ho, t % G.Ho, word % wpr);
    }
    const CmWaveCells Wc = cm_wave_cells(occ, occ_s[wave], excl_s[wave]);
    if (mine) cw_out[word] = make_uint2(occ, (u32)(base.x + excl_s[wave][lane_id()]));
    int row0 = base.y;
    for (int b0 = 0; b0 < Wc.total; b0 += 64 * CM_NB) {
        int cb[CM_NB], coy[CM_NB], cox[CM_NB];
        bool act[CM_NB];
        u64 mu[CM_NB];
#pragma unroll
        for (int n = 0; n < CM_NB; ++n) {
            const int idx = b0 + n * 64 + lane_id();
            act[n] = idx < Wc.total;
            cb[n] = coy[n] = cox[n] = 0;
            if (act[n]) {
                int wl, bit;
                cm_wave_cell(Wc, idx, wl, bit);
                const int wd = word0 + wl, t = wd / wpr;
                cb[n] = t / G.Ho;
                coy[n] = t % G.Ho;
                cox[n] = ((wd % wpr) << 5) + bit;
            }
        }
        in_union_n<KH, KW, CM_NB>(G, cw, cr, cb, coy, cox, act, mu);
#pragma unroll
        for (int n = 0; n < CM_NB; ++n) {
            const int idx = b0 + n * 64 + lane_id();
            const u64 zm = act[n] ? squash_z(mu[n], G) : 0ull;
            const int cnt = __popcll(zm);
            const int inc = wave_inclusive_scan(cnt);
            const int start = row0 + inc - cnt;
            row0 += __shfl(inc, 63);
            if (act[n]) {
                const int col = base.x + idx;
                if (col < G.ncol_cap_out) cr_out[col] = make_uint4((u32)zm, (u32)(zm >> 32), (u32)start, (u32)cnt);
                if (out_indices) {
                    int r = start;
                    u64 m = zm;
                    while (m) {
                        const int z = __ffsll((unsigned long long)m) - 1;
                        m &= m - 1;
                        if (r < n_out) reinterpret_cast<int4 *>(out_indices)[r] = make_int4(cb[n], z, coy[n], cox[n]);
                        ++r;
                    }
                }
            }
        }
    }
}

struct CmTablesSide {
    int *wave_cnt;          // pair lists: hits per (offset, wave of 64 input rows) ...
    int nwaves;
    int *wsuper;            // ... and per 64 waves (zeroed by the count launch)
    int nws;
    int ncls;               // parity classes: permutation of the input rows
    const int *blk_off;     // [ncls][nclsblk] exclusive prefixes inside each class (the emit launch's class blocks)
    const int *cls_tot;     // [ncls] rows per class -> tile-aligned class starts (every block; block nb_out also writes vstart)
    int cls_tile;
    int *vstart;
    int32_t *perm;
    // compact tables (the training step's form; nbr_in / nbr_out may then be nullptr):
    //  nbr_cls [8][vcap]: entry (j, v) = output row of the input row at permutation slot v through the j-th offset its class can
    //                     use (ascending k; ClsTable::k[class][j]) -- 27 / 8 entries per row on average instead of 27;
    //  nbr_out_pk [KH * KW][n_out]: per (ky, kx) the three kz neighbours of an output row, which are CONSECUTIVE rows of one input
    //                     column: { first present row : 29 bits, presence mask of kz = 0, 1, 2 : 3 bits } (0: none)
    int32_t *nbr_cls;
    int vcap;
    u32 *nbr_out_pk;
};

// row of kz = a (0..2) from a packed word, or -1
__device__ __forceinline__ int pk_row(u32 w, int a) {
    const u32 m = w >> 29;
    return ((m >> a) & 1u) ? (int)(w & 0x1FFFFFFFu) + __popc(m & ((1u << a) - 1u)) : -1;
}

// pass 3: both neighbour tables.  Blocks [0, nb_out): one thread per OUTPUT row -> nbr_out[k][o] from the input map;
// blocks [nb_out, nb_out + nb_in): one thread per INPUT row -> nbr_in[k][i] from the output map (+ pair counts, classes).
template <int KD, int KH, int KW>
__global__ __launch_bounds__(256) void cm_conv_tables_kernel(CmGeom G, int nb_out, const int4 *__restrict__ idx, int n,
                                                             const int32_t *n_dev, const int4 *__restrict__ out_idx, int n_out,
                                                             const int32_t *n_out_dev, const uint2 *__restrict__ cw,
                                                             const uint4 *__restrict__ cr, const uint2 *__restrict__ cw_out,
                                                             const uint4 *__restrict__ cr_out, int32_t *__restrict__ nbr_in,
                                                             int32_t *__restrict__ nbr_out, CmTablesSide S) {
    constexpr int K = KD * KH * KW;
    const int nn_in = eff_rows(n_dev, n), nn_out = eff_rows(n_out_dev, n_out);
    if ((int)blockIdx.x < nb_out) {
        const int o = blockIdx.x * 256 + threadIdx.x;
        if (o >= n_out) return;
        const bool live = o < nn_out;
        const int4 c = live ? out_idx[o] : make_int4(0, 0, 0, 0);
        uint2 w[KH * KW];
        u32 key[KH * KW];
        bool in[KH * KW];
#pragma unroll
        for (int q = 0; q < KH * KW; ++q) {
            const int y = c.z * G.sh - G.ph + q / KW, x = c.w * G.sw - G.pw + q % KW;
            in[q] = live && y >= 0 && y < G.H && x >= 0 && x < G.W;
            key[q] = in[q] ? bev_key(c.x, y, x, G.H, G.P) : 0u;
            w[q] = cw[key[q] >> 5];
        }
        uint4 r[KH * KW];
#pragma unroll
        for (int q = 0; q < KH * KW; ++q) {
            const int col = in[q] ? cm_col(w[q], key[q], G.ncol_cap_in) : -1;
            in[q] = col >= 0;
            r[q] = cr[in[q] ? col : 0];
        }
        u32 pk[KH * KW];
#pragma unroll
        for (int q = 0; q < KH * KW; ++q) pk[q] = 0;
#pragma unroll
        for (int a = 0; a < KD; ++a)
#pragma unroll
            for (int q = 0; q < KH * KW; ++q) {
                const int z = c.y * G.sd - G.pd + a;
                int row = -1;
                if (in[q] && z >= 0 && z < G.D) {
                    row = cm_row((u64)r[q].x | ((u64)r[q].y << 32), (int)r[q].z, z);
                    if (row >= nn_in) row = -1;
                }
                if (nbr_out) nbr_out[(size_t)(a * KH * KW + q) * n_out + o] = row;
                if (row >= 0) pk[q] = (pk[q] >> 29) ? (pk[q] | (1u << (29 + a))) : ((u32)row | (1u << (29 + a)));
            }
        if (S.nbr_out_pk) {
#pragma unroll
            for (int q = 0; q < KH * KW; ++q) S.nbr_out_pk[(size_t)q * n_out + o] = pk[q];
        }
        return;
    }
    __shared__ int wcnt[4][CLS_MAX];
    __shared__ int blk_sum[K];
    __shared__ int cstart_s[CLS_MAX + 1];
    const int blk = (int)blockIdx.x - nb_out;
    if (S.perm && threadIdx.x == 0) {
        int vb = 0;
        for (int q = 0; q < S.ncls; ++q) {
            cstart_s[q] = vb;
            vb = (vb + S.cls_tot[q] + S.cls_tile - 1) / S.cls_tile * S.cls_tile;
        }
        cstart_s[S.ncls] = vb;
        if (blk == 0)
            for (int q = 0; q <= S.ncls; ++q) S.vstart[q] = cstart_s[q];
    }
    const int i = blk * 256 + threadIdx.x;
    const bool live = i < nn_in;
    const int4 c = live ? idx[i] : make_int4(0, 0, 0, 0);
    const int wave = i >> 6;
    if (S.wave_cnt) {
        for (int q = threadIdx.x; q < K; q += 256) blk_sum[q] = 0;
        __syncthreads();
    }
    int vslot = -1;             // permutation slot of this row (parity classes)
    if (S.perm) {
        const int cls = live ? row_class(c, G.pd, G.ph, G.pw, G.sd, G.sh, G.sw) : -1;
        const int wv = threadIdx.x >> 6;
        const u64 lt = (lane_id() == 0) ? 0ull : (~0ull >> (64 - lane_id()));
        int rank = 0;
        for (int q = 0; q < S.ncls; ++q) {
            const u64 m = __ballot(cls == q);
            if (cls == q) rank = __popcll(m & lt);
            if (lane_id() == 0) wcnt[wv][q] = __popcll(m);
        }
        __syncthreads();
        if (cls >= 0) {
            int before = 0;
            for (int ww = 0; ww < wv; ++ww) before += wcnt[ww][cls];
            const int nclsblk = (int)gridDim.x - nb_out;
            vslot = cstart_s[cls] + S.blk_off[(size_t)cls * nclsblk + blk] + before + rank;
            S.perm[vslot] = i;
        }
    }
    if (KH == 3 && KW == 3 && !nbr_in && !S.wave_cnt && S.nbr_cls && G.sh == 2 && G.sw == 2) {
        // compact table only, stride 2 in y and x: a row can use ky = 1 (y + ph odd) or ky in {0, 2} (even), the same in x -- at most
        // 2 x 2 of the 3 x 3 output columns, and only those are looked up (the class-compact table has no slot for the others)
        if (vslot < 0) return;
        const int ey = (c.z + G.ph) & 1, ex = (c.w + G.pw) & 1;
        const int nky = ey ? 1 : 2, nkx = ex ? 1 : 2;
        uint2 w4[4];
        u32 key4[4];
        bool in4[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int jy = s >> 1, jx = s & 1;
            const int ky = ey ? 1 : 2 * jy, kx = ex ? 1 : 2 * jx;
            const int oy = axis_out(c.z, G.ph, 1, 2, ky, G.Ho), ox = axis_out(c.w, G.pw, 1, 2, kx, G.Wo);
            in4[s] = jy < nky && jx < nkx && oy >= 0 && ox >= 0;
            key4[s] = in4[s] ? bev_key(c.x, oy, ox, G.Ho, G.Po) : 0u;
            w4[s] = cw_out[key4[s] >> 5];
        }
        uint4 r4[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int col = in4[s] ? cm_col(w4[s], key4[s], G.ncol_cap_out) : -1;
            in4[s] = col >= 0;
            r4[s] = cr_out[in4[s] ? col : 0];
        }
        const int nq4 = nky * nkx;
        int jz4 = 0;
#pragma unroll
        for (int a = 0; a < KD; ++a) {
            const int oz = axis_out(c.y, G.pd, 1, G.sd, a, G.Do);
            const bool pz = ((c.y + G.pd - a) % G.sd + G.sd) % G.sd == 0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int jy = s >> 1, jx = s & 1;
                if (pz && jy < nky && jx < nkx) {
                    int row = -1;
                    if (in4[s] && oz >= 0) {
                        row = cm_row((u64)r4[s].x | ((u64)r4[s].y << 32), (int)r4[s].z, oz);
                        if (row >= nn_out) row = -1;
                    }
                    S.nbr_cls[(size_t)(jz4 * nq4 + jy * nkx + jx) * S.vcap + vslot] = row;
                }
            }
            jz4 += pz ? 1 : 0;
        }
        return;
    }
    // output columns reached through (ky, kx): oy = (y + ph - ky) / sh where that is a whole, in-range number
    uint2 w[KH * KW];
    u32 key[KH * KW];
    bool in[KH * KW];
#pragma unroll
    for (int q = 0; q < KH * KW; ++q) {
        const int oy = axis_out(c.z, G.ph, 1, G.sh, q / KW, G.Ho), ox = axis_out(c.w, G.pw, 1, G.sw, q % KW, G.Wo);
        in[q] = live && oy >= 0 && ox >= 0;
        key[q] = in[q] ? bev_key(c.x, oy, ox, G.Ho, G.Po) : 0u;
        w[q] = cw_out[key[q] >> 5];
    }
    uint4 r[KH * KW];
#pragma unroll
    for (int q = 0; q < KH * KW; ++q) {
        const int col = in[q] ? cm_col(w[q], key[q], G.ncol_cap_out) : -1;
        in[q] = col >= 0;
        r[q] = cr_out[in[q] ? col : 0];
    }
    // class-compact entries: the offsets this row's class can use, numbered in ascending k = (a, q) order
    bool pq[KH * KW];
    int jq[KH * KW], nq = 0;
#pragma unroll
    for (int q = 0; q < KH * KW; ++q) {
        pq[q] = ((c.z + G.ph - q / KW) % G.sh + G.sh) % G.sh == 0 && ((c.w + G.pw - q % KW) % G.sw + G.sw) % G.sw == 0;
        jq[q] = nq;
        nq += pq[q] ? 1 : 0;
    }
    int jz = 0;
#pragma unroll
    for (int a = 0; a < KD; ++a) {
        const int oz = axis_out(c.y, G.pd, 1, G.sd, a, G.Do);
        const bool pz = ((c.y + G.pd - a) % G.sd + G.sd) % G.sd == 0;
#pragma unroll
        for (int q = 0; q < KH * KW; ++q) {
            const int k = a * KH * KW + q;
            int row = -1;
            if (in[q] && oz >= 0) {
                row = cm_row((u64)r[q].x | ((u64)r[q].y << 32), (int)r[q].z, oz);
                if (row >= nn_out) row = -1;
            }
            if (live && nbr_in) nbr_in[(size_t)k * n + i] = row;
            if (S.nbr_cls && vslot >= 0 && pz && pq[q]) S.nbr_cls[(size_t)(jz * nq + jq[q]) * S.vcap + vslot] = row;
            if (S.wave_cnt) publish_wave_count(S.wave_cnt, blk_sum, k, wave, S.nwaves, row >= 0);
        }
        jz += pz ? 1 : 0;
    }
    if (S.wave_cnt) {
        __syncthreads();
        for (int q = threadIdx.x; q < K; q += 256)
            if (blk_sum[q]) atomicAdd(&S.wsuper[(size_t)q * S.nws + (blk >> 4)], blk_sum[q]);
    }
}

// ---- host side -------------------------------------------------------------------------------------------------
struct CmConvWs {
    int2 *bsums;            // per wave of 64 output words: { occupied cells, rows }
    int *wsuper, *wave_cnt, *blk_cnt, *cls_tot;
    int ncellblk, nwaves, nws, nclsblk;
};

bool cm_geom(const int *shape, const int *ks, const int *st, const int *pd, const int *dl, int batch, ConvGeom &G, CmGeom &C) {
    if (make_geom(shape, ks, st, pd, dl, G) != PCD_OK) return false;
    if (G.dd != 1 || G.dh != 1 || G.dw != 1) return false;
    if (!((G.kd == 3 && G.kh == 3 && G.kw == 3) || (G.kd == 3 && G.kh == 1 && G.kw == 1))) return false;
    if (G.sd < 1 || G.sd > 2 || G.sh < 1 || G.sh > 2 || G.sw < 1 || G.sw > 2) return false;
    if (G.D + G.pd > 62 || G.Do > 62 || G.Do <= 0 || G.Ho <= 0 || G.Wo <= 0) return false;
    if (G.pw > 31) return false;
    C.P = cm_pitch(G.W);
    C.Po = cm_pitch(G.Wo);
    const double cells = (double)batch * G.Ho * C.Po, cells_in = (double)batch * G.H * C.P;
    if (cells >= 2147483647.0 - 4096.0 || cells_in >= 2147483647.0 - 4096.0) return false;
    C.D = G.D; C.H = G.H; C.W = G.W; C.Do = G.Do; C.Ho = G.Ho; C.Wo = G.Wo;
    C.kd = G.kd; C.sd = G.sd; C.sh = G.sh; C.sw = G.sw; C.pd = G.pd; C.ph = G.ph; C.pw = G.pw;
    C.nwords_out = (int)(cells / 32);
    C.wshift = cm_wshift(C.nwords_out);
    return true;
}

bool cm_conv_ws(void *p, size_t bytes, int n, const ConvGeom &G, const CmGeom &C, CmConvWs &L, size_t *need) {
    L.ncellblk = pcd_div_up(pcd_div_up(C.nwords_out, 1 << C.wshift), 4);
    L.nwaves = pcd_div_up(n > 0 ? n : 1, 64);
    L.nws = pcd_div_up(L.nwaves, 64);
    L.nclsblk = pcd_div_up(n > 0 ? n : 1, 256);
    WsCarver ws(p, bytes);
    L.bsums = ws.take<int2>((size_t)L.ncellblk * 4 + 2);
    L.wsuper = ws.take<int>((size_t)G.K * L.nws);
    L.wave_cnt = ws.take<int>((size_t)G.K * L.nwaves);
    L.blk_cnt = ws.take<int>((size_t)CLS_MAX * L.nclsblk);
    L.cls_tot = ws.take<int>(CLS_MAX);
    if (need) *need = ws.off;
    return p == nullptr || ws.ok;
}

template <int KH, int KW>
void cm_launch_count(const CmGeom &C, const CmConvWs &L, const CmBuf &in, const CmConvSide &S, int cls_blocks, hipStream_t st) {
    cm_conv_count_kernel<KH, KW><<<L.ncellblk + cls_blocks, 256, 0, st>>>(C, L.ncellblk, in.cw, in.cr, L.bsums, S);
    if (cm_spined(L.ncellblk * 4)) cm_spine2_kernel<<<1, 1024, 0, st>>>(L.bsums, L.ncellblk * 4);
}

template <int KH, int KW>
void cm_launch_emit(const CmGeom &C, const CmConvWs &L, const CmBuf &in, const CmBuf &out, int32_t *out_indices, int n_out,
                    int32_t *n_out_dev, const CmEmitSide &S, hipStream_t st) {
    cm_conv_emit_kernel<KH, KW><<<L.ncellblk + (S.blk_cnt ? S.ncls : 0), 256, 0, st>>>(
        C, L.ncellblk, in.cw, in.cr, L.bsums, cm_spined(L.ncellblk * 4), out.cw, out.cr, out.ncols, out_indices, n_out,
        n_out_dev, S);
}

}  // namespace

// =============================================================================================
extern "C" size_t pcd_colmap_bytes(int batch, const int *shape_host, int n_cap) {
    if (!shape_host) return 0;
    CmBuf B;
    size_t need = 0;
    if (!cm_carve(nullptr, 0, batch, shape_host[1], shape_host[2], n_cap, B, &need)) return 0;
    return need;
}

// byte offset of the map's two counters {columns, rows} (int32 each, written by the build that produced the map) inside a
// buffer of pcd_colmap_bytes(batch, shape, n_cap) bytes, and its column capacity: a caller checks columns <= capacity (a
// strided build whose output z range does not cover every input z numbers columns WITHOUT rows: more columns than rows are
// then possible, and columns beyond the capacity would be lost)
extern "C" size_t pcd_colmap_counts_offset(int batch, const int *shape_host, int n_cap, int *ncol_cap_out) {
    if (!shape_host) return 0;
    CmBuf B;
    if (!cm_carve(nullptr, 0, batch, shape_host[1], shape_host[2], n_cap, B, nullptr)) return 0;
    if (ncol_cap_out) *ncol_cap_out = B.ncol_cap;
    return ws_piece(B.nwords + 2, sizeof(uint2)) + ws_piece((size_t)B.ncol_cap + 1, sizeof(uint4));      // (cm_carve's order: cw, cr, counts)
}

extern "C" size_t pcd_colmap_from_rows_workspace_bytes(int batch, const int *shape_host) {
    if (!shape_host) return 0;
    CmBuf B;
    if (!cm_carve(nullptr, 0, batch, shape_host[1], shape_host[2], 1, B, nullptr)) return 0;
    return ws_piece(B.nwords + 4, sizeof(u32)) + ws_piece(pcd_div_up((int)B.nwords, 1024) + 2, sizeof(int));
}

extern "C" int pcd_colmap_from_rows(const int32_t *indices, int n, const int32_t *n_dev, int batch, const int *shape_host,
                                    void *colmap, size_t colmap_bytes, void *workspace, size_t workspace_bytes,
                                    void *stream) {
    PCD_ENTER();
    if (n < 0 || batch <= 0 || !shape_host || !colmap || (n > 0 && !indices)) return PCD_ERR_INVALID_ARG;
    if (shape_host[0] <= 0 || shape_host[0] > 62) return PCD_ERR_UNSUPPORTED;
    CmBuf B;
    if (!cm_carve(colmap, colmap_bytes, batch, shape_host[1], shape_host[2], n, B, nullptr)) return PCD_ERR_WORKSPACE;
    const int nwords = (int)B.nwords, nblk = pcd_div_up(nwords, 1024);
    WsCarver ws(workspace, workspace_bytes);
    u32 *bits = ws.take<u32>(B.nwords + 4);
    int *bsums = ws.take<int>(nblk + 2);
    if (!ws.ok) return PCD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int H = shape_host[1], P = B.pitch;
    pcd_fill(bits, 0, (B.nwords + 4) * sizeof(u32), st);
    const int nb = pcd_div_up(n > 0 ? n : 1, 256);
    if (n > 0) cm_rows_mark_kernel<<<nb, 256, 0, st>>>((const int4 *)indices, n, n_dev, H, P, bits);
    cm_words_count_kernel<<<nblk, 256, 0, st>>>(bits, nwords, bsums);
    const int spined = cm_spined(nblk);
    if (spined) scan_spine_kernel<<<1, 256, 0, st>>>(bsums, nblk, nullptr);
    cm_words_prefix_kernel<<<nblk, 256, 0, st>>>(bits, nwords, nblk, bsums, spined, B.cw, B.ncols);
    if (n > 0)
        cm_rows_fill_kernel<<<nb, 256, 0, st>>>((const int4 *)indices, n, n_dev, H, P, B.cw, B.cr, B.ncol_cap, B.ncols);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" size_t pcd_rulebook_subm_cm_workspace_bytes(int n) {
    if (n < 0) return 0;
    const int nwaves = pcd_div_up(n > 0 ? n : 1, 64);
    return 2 * ws_piece((size_t)27 * nwaves, sizeof(int)) + ws_piece(27, sizeof(int));
}

extern "C" int pcd_rulebook_subm_cm(const int32_t *indices, int n, int batch, const int *shape_host, const void *colmap,
                                    size_t colmap_bytes, int colmap_cap, int32_t *nbr, int32_t *pairs, int32_t *pair_num,
                                    int pad_pairs, const int32_t *n_dev, void *workspace, size_t workspace_bytes,
                                    void *stream) {
    PCD_ENTER();
    if (n < 0 || batch <= 0 || !shape_host || !colmap) return PCD_ERR_INVALID_ARG;
    if (n > 0 && (pairs != nullptr) != (pair_num != nullptr)) return PCD_ERR_INVALID_ARG;
    if (shape_host[0] <= 0 || shape_host[0] > 62) return PCD_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (pair_num) pcd_fill(pair_num, 0, 27 * sizeof(int32_t), st);
        return PCD_OK;
    }
    if (!indices || !nbr) return PCD_ERR_INVALID_ARG;
    CmBuf B;
    if (!cm_carve(const_cast<void *>(colmap), colmap_bytes, batch, shape_host[1], shape_host[2], colmap_cap, B, nullptr))
        return PCD_ERR_WORKSPACE;
    const int nwaves = pcd_div_up(n, 64);
    WsCarver ws(workspace, workspace_bytes);
    int *wave_cnt = ws.take<int>((size_t)27 * nwaves);
    int *wave_off = ws.take<int>((size_t)27 * nwaves);
    int *totals = ws.take<int>(27);
    if (pairs && !ws.ok) return PCD_ERR_WORKSPACE;
    cm_subm_kernel<<<pcd_div_up(n, 256), 256, 0, st>>>((const int4 *)indices, n, n_dev, shape_host[0], shape_host[1],
                                                       shape_host[2], B.pitch, B.cw, B.cr, B.ncol_cap, nbr,
                                                       pairs ? wave_cnt : nullptr, nwaves);
    if (pairs) {
        scan_rows_kernel<<<27, 256, 0, st>>>(wave_cnt, wave_off, nwaves, totals, pair_num, 1);
        if (pad_pairs) pcd_fill(pairs, 0xFF, (size_t)27 * 2 * n * sizeof(int32_t), st);
        launch_pairs_fill(nbr, n, n_dev, 27, 1, wave_off, nwaves, pairs, st);
    }
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" size_t pcd_rulebook_conv_cm_workspace_bytes(int n, int batch, const int *in_shape_host, const int *ksize_host,
                                                       const int *stride_host, const int *pad_host) {
    if (n < 0 || batch <= 0 || !in_shape_host || !ksize_host || !stride_host || !pad_host) return 0;
    const int one[3] = {1, 1, 1};
    ConvGeom G;
    CmGeom C;
    if (!cm_geom(in_shape_host, ksize_host, stride_host, pad_host, one, batch, G, C)) return 0;
    CmConvWs L;
    size_t need = 0;
    cm_conv_ws(nullptr, 0, n, G, C, L, &need);
    return need;
}

namespace {

struct CmConvCall {
    ConvGeom G;
    CmGeom C;
    CmConvWs L;
    CmBuf in, out;
};

int cm_conv_setup(int n, int batch, const int *in_shape_host, const int *ksize_host, const int *stride_host,
                  const int *pad_host, const void *in_colmap, size_t in_colmap_bytes, int in_cap, void *out_colmap,
                  size_t out_colmap_bytes, int out_cap, void *workspace, size_t workspace_bytes, CmConvCall &X) {
    const int one[3] = {1, 1, 1};
    if (!in_shape_host || !ksize_host || !stride_host || !pad_host || !in_colmap) return PCD_ERR_INVALID_ARG;
    if (!cm_geom(in_shape_host, ksize_host, stride_host, pad_host, one, batch, X.G, X.C)) return PCD_ERR_UNSUPPORTED;
    if (!cm_carve(const_cast<void *>(in_colmap), in_colmap_bytes, batch, X.G.H, X.G.W, in_cap, X.in, nullptr))
        return PCD_ERR_WORKSPACE;
    X.out = CmBuf{};
    if (out_colmap && !cm_carve(out_colmap, out_colmap_bytes, batch, X.G.Ho, X.G.Wo, out_cap, X.out, nullptr))
        return PCD_ERR_WORKSPACE;
    X.C.ncol_cap_in = X.in.ncol_cap;
    X.C.ncol_cap_out = out_colmap ? X.out.ncol_cap : 0;
    if (!cm_conv_ws(workspace, workspace_bytes, n, X.G, X.C, X.L, nullptr) || !workspace) return PCD_ERR_WORKSPACE;
    return PCD_OK;
}

void cm_count(const CmConvCall &X, const CmConvSide &S, int cls_blocks, hipStream_t st) {
    if (X.G.kh == 3) cm_launch_count<3, 3>(X.C, X.L, X.in, S, cls_blocks, st);
    else cm_launch_count<1, 1>(X.C, X.L, X.in, S, cls_blocks, st);
}

void cm_emit(const CmConvCall &X, int32_t *out_indices, int n_out, int32_t *n_out_dev, const CmEmitSide &S, hipStream_t st) {
    if (X.G.kh == 3) cm_launch_emit<3, 3>(X.C, X.L, X.in, X.out, out_indices, n_out, n_out_dev, S, st);
    else cm_launch_emit<1, 1>(X.C, X.L, X.in, X.out, out_indices, n_out, n_out_dev, S, st);
}

void cm_tables(const CmConvCall &X, const int32_t *indices, int n, const int32_t *n_dev, const int32_t *out_indices, int n_out,
               const int32_t *n_out_dev, int32_t *nbr_in, int32_t *nbr_out, const CmTablesSide &S, hipStream_t st) {
    const int nb_out = pcd_div_up(n_out, 256), nb_in = pcd_div_up(n, 256);
    if (X.G.kh == 3)
        cm_conv_tables_kernel<3, 3, 3><<<nb_out + nb_in, 256, 0, st>>>(X.C, nb_out, (const int4 *)indices, n, n_dev,
                                                                       (const int4 *)out_indices, n_out, n_out_dev, X.in.cw,
                                                                       X.in.cr, X.out.cw, X.out.cr, nbr_in, nbr_out, S);
    else
        cm_conv_tables_kernel<3, 1, 1><<<nb_out + nb_in, 256, 0, st>>>(X.C, nb_out, (const int4 *)indices, n, n_dev,
                                                                       (const int4 *)out_indices, n_out, n_out_dev, X.in.cw,
                                                                       X.in.cr, X.out.cw, X.out.cr, nbr_in, nbr_out, S);
}

}  // namespace

extern "C" int pcd_rulebook_conv_cm_count(int n, int batch, const int *in_shape_host, const int *ksize_host,
                                          const int *stride_host, const int *pad_host, const void *in_colmap,
                                          size_t in_colmap_bytes, int in_cap, int32_t *n_out_dev, void *workspace,
                                          size_t workspace_bytes, void *stream) {
    PCD_ENTER();
    if (n < 0 || batch <= 0 || !n_out_dev) return PCD_ERR_INVALID_ARG;
    CmConvCall X;
    int rc = cm_conv_setup(n, batch, in_shape_host, ksize_host, stride_host, pad_host, in_colmap, in_colmap_bytes, in_cap,
                           nullptr, 0, 0, workspace, workspace_bytes, X);
    if (rc != PCD_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    CmConvSide S = {};
    cm_count(X, S, 0, st);
    cm_total_kernel<<<1, 256, 0, st>>>(X.L.bsums, X.L.ncellblk * 4, cm_spined(X.L.ncellblk * 4), n_out_dev);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

// count (+ class counts) -> emit (+ class offsets) -> tables (+ class permutation) -> pair lists: four launches.
// `counted`: pcd_rulebook_conv_cm_count ran over the same workspace (the two-phase API): the count launch is skipped.
static int cm_conv_build_impl(const int32_t *indices, int n, int batch, const int *in_shape_host, const int *ksize_host,
                              const int *stride_host, const int *pad_host, const void *in_colmap, size_t in_colmap_bytes,
                              int in_cap, int n_out_cap, int32_t *n_out_dev, int32_t *out_indices, void *out_colmap,
                              size_t out_colmap_bytes, int32_t *nbr_in, int32_t *nbr_out, int32_t *pairs, int32_t *pair_num,
                              int pad_pairs, int cls_tile, int32_t *perm, int vcap, int32_t *vstart_dev, const int32_t *n_dev,
                              void *workspace, size_t workspace_bytes, void *stream, bool counted,
                              int32_t *nbr_cls = nullptr, u32 *nbr_out_pk = nullptr) {
    PCD_ENTER();
    if (n <= 0 || batch <= 0 || n_out_cap <= 0) return PCD_ERR_INVALID_ARG;
    if (!indices || !out_indices || !out_colmap) return PCD_ERR_INVALID_ARG;
    if ((!nbr_in && !nbr_cls) || (!nbr_out && !nbr_out_pk)) return PCD_ERR_INVALID_ARG;
    if (nbr_cls && !perm) return PCD_ERR_INVALID_ARG;                 // (the compact input-side table is indexed by permutation slot)
    if (pairs && !nbr_in) return PCD_ERR_INVALID_ARG;                 // (the pair lists are cut from the full table)
    if (n_out_cap >= (1 << 29) || n >= (1 << 29)) return PCD_ERR_UNSUPPORTED;
    if ((pairs != nullptr) != (pair_num != nullptr)) return PCD_ERR_INVALID_ARG;
    CmConvCall X;
    int rc = cm_conv_setup(n, batch, in_shape_host, ksize_host, stride_host, pad_host, in_colmap, in_colmap_bytes, in_cap,
                           out_colmap, out_colmap_bytes, n_out_cap, workspace, workspace_bytes, X);
    if (rc != PCD_OK) return rc;
    const int ncls = X.G.sd * X.G.sh * X.G.sw;
    const bool classes = perm != nullptr;
    if (classes) {
        if (ncls > CLS_MAX) return PCD_ERR_UNSUPPORTED;
        if (cls_tile <= 0 || !vstart_dev || vcap < (n + cls_tile - 1) / cls_tile * cls_tile + ncls * cls_tile)
            return PCD_ERR_INVALID_ARG;
    }
    if ((nbr_cls || nbr_out_pk) && X.G.kd != 3) return PCD_ERR_UNSUPPORTED;
    if (nbr_cls) {                         // 8 table rows: every class must get by with at most 8 usable offsets
        const int one[3] = {1, 1, 1};
        ClsTable CT;
        if (int rc2 = make_cls_table(ksize_host, stride_host, one, CT)) return rc2;
    }
    hipStream_t st = (hipStream_t)stream;
    const CmConvWs &L = X.L;
    if (!counted) {
        CmConvSide S = {};
        if (pairs) {
            S.zero = L.wsuper;
            S.zero_words = X.G.K * L.nws;
        }
        if (classes) {
            S.idx = (const int4 *)indices;
            S.n = n;
            S.n_dev = n_dev;
            S.ncls = ncls;
            S.blk_cnt = L.blk_cnt;
            S.pd = X.G.pd; S.ph = X.G.ph; S.pw = X.G.pw;
        }
        cm_count(X, S, classes ? L.nclsblk : 0, st);
    } else {
        if (pairs) pcd_fill(L.wsuper, 0, (size_t)X.G.K * L.nws * sizeof(int), st);
        if (classes) return PCD_ERR_INVALID_ARG;      // (the two-phase API builds its classes with pcd_rulebook_conv_classes)
    }
    CmEmitSide E = {};
    if (classes) {
        E.fill_a = perm;
        E.fill_a_bytes = (size_t)vcap * sizeof(int32_t);
        E.blk_cnt = L.blk_cnt;
        E.cls_nblk = L.nclsblk;
        E.ncls = ncls;
        E.cls_tile = cls_tile;
        E.cls_tot = L.cls_tot;
    }
    cm_emit(X, out_indices, n_out_cap, n_out_dev, E, st);
    CmTablesSide T = {};
    if (pairs) {
        T.wave_cnt = L.wave_cnt;
        T.nwaves = L.nwaves;
        T.wsuper = L.wsuper;
        T.nws = L.nws;
    }
    if (classes) {
        T.ncls = ncls;
        T.blk_off = L.blk_cnt;
        T.cls_tot = L.cls_tot;
        T.cls_tile = cls_tile;
        T.vstart = vstart_dev;
        T.perm = perm;
    }
    T.nbr_cls = nbr_cls;
    T.vcap = vcap;
    T.nbr_out_pk = nbr_out_pk;
    cm_tables(X, indices, n, n_dev, out_indices, n_out_cap, n_out_dev, nbr_in, nbr_out, T, st);
    if (pairs) {
        if (pad_pairs) pcd_fill(pairs, 0xFF, (size_t)X.G.K * 2 * n * sizeof(int32_t), st);
        launch_pairs_fill_super(nbr_in, n, n_dev, X.G.K, 0, L.wave_cnt, L.nwaves, L.wsuper, L.nws, pairs, pair_num, st);
    }
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_rulebook_conv_cm_fill(const int32_t *indices, int n, int batch, const int *in_shape_host,
                                         const int *ksize_host, const int *stride_host, const int *pad_host,
                                         const void *in_colmap, size_t in_colmap_bytes, int in_cap, int n_out,
                                         int32_t *out_indices, void *out_colmap, size_t out_colmap_bytes, int32_t *nbr_in,
                                         int32_t *nbr_out, int32_t *pairs, int32_t *pair_num, int pad_pairs,
                                         const int32_t *n_dev, void *workspace, size_t workspace_bytes, void *stream) {
    return cm_conv_build_impl(indices, n, batch, in_shape_host, ksize_host, stride_host, pad_host, in_colmap, in_colmap_bytes,
                              in_cap, n_out, nullptr, out_indices, out_colmap, out_colmap_bytes, nbr_in, nbr_out, pairs,
                              pair_num, pad_pairs, 0, nullptr, 0, nullptr, n_dev, workspace, workspace_bytes, stream, true);
}

extern "C" int pcd_rulebook_conv_cm_build(const int32_t *indices, int n, int batch, const int *in_shape_host,
                                          const int *ksize_host, const int *stride_host, const int *pad_host,
                                          const void *in_colmap, size_t in_colmap_bytes, int in_cap, int n_out_cap,
                                          int32_t *n_out_dev, int32_t *out_indices, void *out_colmap, size_t out_colmap_bytes,
                                          int32_t *nbr_in, int32_t *nbr_out, int32_t *pairs, int32_t *pair_num, int pad_pairs,
                                          int cls_tile, int32_t *perm, int vcap, int32_t *vstart_dev, const int32_t *n_dev,
                                          void *workspace, size_t workspace_bytes, void *stream) {
    if (!n_out_dev) return PCD_ERR_INVALID_ARG;
    return cm_conv_build_impl(indices, n, batch, in_shape_host, ksize_host, stride_host, pad_host, in_colmap, in_colmap_bytes,
                              in_cap, n_out_cap, n_out_dev, out_indices, out_colmap, out_colmap_bytes, nbr_in, nbr_out, pairs,
                              pair_num, pad_pairs, cls_tile, perm, vcap, vstart_dev, n_dev, workspace, workspace_bytes, stream,
                              false);
}

// The same build with COMPACT neighbour tables (the training step's form): no 27-wide nbr_in / nbr_out, no pair lists.
//   nbr_out_packed [kh * kw][n_out_cap] u32: per (ky, kx) { first present input row : 29, presence of kz = 0, 1, 2 : 3 } -- the three
//       kz neighbours of an output row are consecutive rows of one input column (rows are z-fastest);
//   nbr_cls [8][vcap] i32: entry (j, v) = output row reached from input row perm[v] through the j-th kernel offset its stride-parity
//       class can use (ascending k), or -1 -- 27 / 8 entries per row on average.
// Read by pcd_sparse_conv_gather_gemm_packed, pcd_sparse_conv_dgrad_classes_v2 and pcd_sparse_conv_wgrad_classes (nbr_compact = 1);
// pcd_rulebook_conv_expand_nbr_out / _nbr_in give the 27-wide tables back.  Kernel depth 3, at most 8 classes.
extern "C" int pcd_rulebook_conv_cm_build_compact(const int32_t *indices, int n, int batch, const int *in_shape_host,
                                                  const int *ksize_host, const int *stride_host, const int *pad_host,
                                                  const void *in_colmap, size_t in_colmap_bytes, int in_cap, int n_out_cap,
                                                  int32_t *n_out_dev, int32_t *out_indices, void *out_colmap,
                                                  size_t out_colmap_bytes, uint32_t *nbr_out_packed, int32_t *nbr_cls, int cls_tile,
                                                  int32_t *perm, int vcap, int32_t *vstart_dev, const int32_t *n_dev,
                                                  void *workspace, size_t workspace_bytes, void *stream) {
    if (!n_out_dev || !nbr_out_packed || !nbr_cls || !perm) return PCD_ERR_INVALID_ARG;
    return cm_conv_build_impl(indices, n, batch, in_shape_host, ksize_host, stride_host, pad_host, in_colmap, in_colmap_bytes,
                              in_cap, n_out_cap, n_out_dev, out_indices, out_colmap, out_colmap_bytes, nullptr, nullptr, nullptr,
                              nullptr, 0, cls_tile, perm, vcap, vstart_dev, n_dev, workspace, workspace_bytes, stream, false,
                              nbr_cls, nbr_out_packed);
}

namespace {

__global__ __launch_bounds__(256) void pk_expand_out_kernel(const u32 *__restrict__ pk, int kq, int n_out,
                                                            const int32_t *n_out_dev, int32_t *__restrict__ nbr_out) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    const bool live = o < eff_rows(n_out_dev, n_out);
    for (int q = 0; q < kq; ++q) {
        const u32 w = live ? pk[(size_t)q * n_out + o] : 0u;
        for (int a = 0; a < 3; ++a) nbr_out[(size_t)(a * kq + q) * n_out + o] = pk_row(w, a);
    }
}

__global__ __launch_bounds__(256) void cls_expand_in_kernel(const int32_t *__restrict__ nbr_cls, int vcap,
                                                            const int32_t *__restrict__ perm, const int32_t *__restrict__ vstart,
                                                            ClsTable T, int n, int32_t *__restrict__ nbr_in) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= vstart[T.ncls] || v >= vcap) return;
    const int i = perm[v];
    if (i < 0 || i >= n) return;
    int cls = 0;
    for (int q = 1; q < T.ncls; ++q)
        if (vstart[q] <= v) cls = q;
    for (int c = 0; c < 8; ++c)            // (compile-time indices into the by-value table)
        if (c == cls)
            for (int j = 0; j < 8; ++j)
                if (j < T.nk[c]) nbr_in[(size_t)T.k[c][j] * n + i] = nbr_cls[(size_t)j * vcap + v];
}

}  // namespace

extern "C" int pcd_rulebook_conv_expand_nbr_out(const uint32_t *nbr_out_packed, int kq, int n_out, const int32_t *n_out_dev,
                                                int32_t *nbr_out, void *stream) {
    PCD_ENTER();
    if (kq <= 0 || n_out < 0 || (n_out > 0 && (!nbr_out_packed || !nbr_out))) return PCD_ERR_INVALID_ARG;
    if (n_out == 0) return PCD_OK;
    pk_expand_out_kernel<<<pcd_div_up(n_out, 256), 256, 0, (hipStream_t)stream>>>(nbr_out_packed, kq, n_out, n_out_dev, nbr_out);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_rulebook_conv_expand_nbr_in(const int32_t *nbr_cls, int vcap, const int32_t *perm, const int32_t *vstart_dev,
                                               const int *ksize_host, const int *stride_host, int n, int32_t *nbr_in,
                                               void *stream) {
    PCD_ENTER();
    if (!ksize_host || !stride_host || n < 0 || vcap < 0) return PCD_ERR_INVALID_ARG;
    if (n == 0) return PCD_OK;
    if (!nbr_cls || !perm || !vstart_dev || !nbr_in) return PCD_ERR_INVALID_ARG;
    const int one[3] = {1, 1, 1};
    ClsTable T;
    if (int rc = make_cls_table(ksize_host, stride_host, one, T)) return rc;
    const int K = ksize_host[0] * ksize_host[1] * ksize_host[2];
    hipStream_t st = (hipStream_t)stream;
    pcd_fill(nbr_in, 0xFF, (size_t)K * n * sizeof(int32_t), st);
    cls_expand_in_kernel<<<pcd_div_up(vcap > 0 ? vcap : 1, 256), 256, 0, st>>>(nbr_cls, vcap, perm, vstart_dev, T, n, nbr_in);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
