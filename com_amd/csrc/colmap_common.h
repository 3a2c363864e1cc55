// Column maps (include/pcd_ops.h "Column maps"): layout of the caller-owned buffer, lookups and the scan over the BEV occupancy
// words -- shared by colmap.hip (rulebook builds) and voxelize.hip (the z-fastest voxeliser emits level 1's map itself).
#pragma once
#include "rulebook_common.h"

namespace {

// block sums are added up by the consuming block up to "cm_direct_blocks" blocks (4096: 16 loads per thread), a spine launch beyond
static inline bool cm_spined(int nblk) { return nblk > pcd_opt(PCD_OPT_CM_DIRECT_BLOCKS); }

struct CmBuf {            // a level's column map inside ONE caller-owned buffer (pcd_colmap_bytes)
    uint2 *cw;            // [nwords + 2]
    uint4 *cr;            // [ncol_cap + 1]
    int *ncols;           // [4]: columns, rows (diagnostics)
    size_t nwords;        // batch * H * pitch / 32
    int ncol_cap;
    int pitch;            // BEV row pitch in cells: W rounded up to 32
};

static inline int cm_pitch(int W) { return (W + 31) & ~31; }

bool cm_carve(void *p, size_t bytes, int batch, int H, int W, int n_cap, CmBuf &B, size_t *need) {
    if (batch <= 0 || H <= 0 || W <= 0) return false;
    B.pitch = cm_pitch(W);
    const double cells = (double)batch * H * B.pitch;
    if (cells >= 2147483647.0 - 4096.0) return false;
    B.nwords = (size_t)cells / 32;
    const int real = batch * H * W;
    B.ncol_cap = n_cap < real ? (n_cap > 0 ? n_cap : 1) : real;
    WsCarver ws(p, bytes);
    B.cw = ws.take<uint2>(B.nwords + 2);
    B.cr = ws.take<uint4>((size_t)B.ncol_cap + 1);
    B.ncols = ws.take<int>(4);
    if (need) *need = ws.off;
    return p == nullptr || ws.ok;
}

// ---- lookups ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 bev_key(int b, int y, int x, int H, int pitch) { return ((u32)b * H + y) * pitch + x; }

// column index of a BEV cell given its word, or -1
__device__ __forceinline__ int cm_col(uint2 w, u32 key, int ncol_cap) {
    const u32 bit = key & 31u;
    if (!((w.x >> bit) & 1u)) return -1;
    const int col = (int)w.y + __popc(w.x & ((1u << bit) - 1u));
    return col < ncol_cap ? col : -1;
}

__device__ __forceinline__ int cm_row(u64 zm, int start, int z) {
    return ((zm >> z) & 1ull) ? start + __popcll(zm & ((1ull << z) - 1ull)) : -1;
}

// words -> block sums of set bits (1024 words per block)
__global__ __launch_bounds__(256) void cm_words_count_kernel(const u32 *__restrict__ bits, int nwords, int *__restrict__ bsums) {
    __shared__ int lds[4];
    const int w0 = blockIdx.x * 1024 + threadIdx.x * 4;
    int c = 0;
    if (w0 + 3 < nwords) {
        const uint4 q = *reinterpret_cast<const uint4 *>(bits + w0);
        c = __popc(q.x) + __popc(q.y) + __popc(q.z) + __popc(q.w);
    } else {
        for (int j = 0; j < 4; ++j)
            if (w0 + j < nwords) c += __popc(bits[w0 + j]);
    }
    const int t = block_sum(c, lds);
    if (threadIdx.x == 0) bsums[blockIdx.x] = t;
}

// sum of bsums[0 .. blk) by the block itself, or bsums[blk] when a spine launch left exclusive prefixes there
__device__ __forceinline__ int cm_base(const int *__restrict__ bsums, int blk, int spined, int *lds) {
    if (spined) return bsums[blk];
    int acc = 0;
    for (int j = threadIdx.x; j < blk; j += 256) acc += bsums[j];
    return block_sum(acc, lds);
}

__global__ __launch_bounds__(256) void cm_words_prefix_kernel(const u32 *__restrict__ bits, int nwords, int nblk,
                                                              const int *__restrict__ bsums, int spined,
                                                              uint2 *__restrict__ cw, int *__restrict__ ncols,
                                                              u32 *__restrict__ colkey = nullptr, int ncol_cap = 0) {
    __shared__ int lds[4];
    const int base = cm_base(bsums, blockIdx.x, spined, lds);
    if (blockIdx.x == 0 && ncols) {
        int t = 0;
        if (spined) t = bsums[nblk];
        else {
            for (int j = threadIdx.x; j < nblk; j += 256) t += bsums[j];
            t = block_sum(t, lds);
        }
        if (threadIdx.x == 0) ncols[0] = t;
    }
    const int w0 = blockIdx.x * 1024 + threadIdx.x * 4;
    u32 b[4] = {0u, 0u, 0u, 0u};
    for (int j = 0; j < 4; ++j)
        if (w0 + j < nwords) b[j] = bits[w0 + j];
    const int c0 = __popc(b[0]), c1 = __popc(b[1]), c2 = __popc(b[2]), c3 = __popc(b[3]);
    int total;
    const int ex = base + block_exclusive_scan(c0 + c1 + c2 + c3, lds, total);
    const int pre[4] = {ex, ex + c0, ex + c0 + c1, ex + c0 + c1 + c2};
    for (int j = 0; j < 4; ++j)
        if (w0 + j < nwords) cw[w0 + j] = make_uint2(b[j], (u32)pre[j]);
    if (colkey) {          // the inverse map: column -> BEV key (consecutive columns from consecutive threads)
        for (int j = 0; j < 4; ++j) {
            u32 m = b[j];
            int c = pre[j];
            while (m) {
                const int bit = __ffs(m) - 1;
                m &= m - 1u;
                if (c < ncol_cap) colkey[c] = (u32)(w0 + j) * 32u + (u32)bit;
                ++c;
            }
        }
    }
}

}  // namespace
