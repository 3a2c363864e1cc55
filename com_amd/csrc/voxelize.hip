// Point -> voxel scatter for gfx950.
//
// Hard voxelisation (reference call site pcdet/datasets/processor/data_processor.py:44-60, algorithm
// SURVEY.md A.1) is order dependent on the CPU: voxel ids are first-appearance ranks and each voxel
// keeps its first T points.  The GPU formulation is order free and deterministic:
//   1. every point hashes its (b,z,y,x) key into an open-addressing table (64-bit keys, CAS) and
//      pushes its own index through a T-deep "k smallest" cascade of atomicMin's attached to the
//      slot.  Whatever the interleaving, slot.best[t] ends as the (t+1)-th smallest point index of
//      the voxel == the reference's t-th kept point; best[0] is the voxel's first point.
//   2. flag = "I am the first point of my voxel"; an exclusive scan of the flags over the point
//      buffer (wave ballot/popcount + block scan) gives the first-appearance rank == voxel id.
//   3. per-frame caps (max_voxels) and the compaction over frames are resolved by a tiny kernel,
//      then one thread per voxel gathers its <= T points, writes coords / num_points / voxels and
//      the fused MeanVFE row.
// Point rows are read with coalesced scalar loads (5 floats/point); index math uses __fsub_rn /
// __fdiv_rn so no reciprocal or contraction can change floor((p - min) / vsize).
//
// Dynamic voxelisation + mean (pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:53-72) needs the
// voxels sorted by key: an occupancy bitmap over the key space + popcount prefix gives every key
// its rank without sorting.
#include "colmap_common.h"

namespace {

struct VoxGeom {
    float r0, r1, r2;
    float v0, v1, v2;
    int gx, gy, gz;
    int kz;   // z extent of the KEY space of the key-ordered form (>= gz: the backbones' sparse_shape has gz + 1 planes)
    int order;   // key-ordered form: PCD_ROWS_ZYX = rows by (b, z, y, x); PCD_ROWS_YXZ = rows by (b, y, x, z), z fastest
};

// (__host__ too: pcd_voxelize_hard_host runs the same arithmetic on the CPU -- IEEE float32 subtract and divide, no contraction)
__host__ __device__ __forceinline__ bool voxel_coord(const float *p, const VoxGeom &G, int &cx, int &cy,
                                                     int &cz) {
#if defined(__HIP_DEVICE_COMPILE__)
    float fx = floorf(__fdiv_rn(__fsub_rn(p[0], G.r0), G.v0));
    float fy = floorf(__fdiv_rn(__fsub_rn(p[1], G.r1), G.v1));
    float fz = floorf(__fdiv_rn(__fsub_rn(p[2], G.r2), G.v2));
#else
    volatile float dx = p[0] - G.r0, dy = p[1] - G.r1, dz = p[2] - G.r2;      // (volatile: one rounding per operation)
    volatile float qx = dx / G.v0, qy = dy / G.v1, qz = dz / G.v2;
    float fx = floorf(qx), fy = floorf(qy), fz = floorf(qz);
#endif
    bool ok = (fx >= 0.0f) && (fx < (float)G.gx) && (fy >= 0.0f) && (fy < (float)G.gy) &&
              (fz >= 0.0f) && (fz < (float)G.gz);
    cx = (int)fx;
    cy = (int)fy;
    cz = (int)fz;
    return ok;
}

__device__ __forceinline__ int frame_of(const int32_t *offs, int batch, int i) {
    // largest b with offs[b] <= i
    int lo = 0, hi = batch;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (offs[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

constexpr u64 KEY_EMPTY = ~0ull;
constexpr u32 IDX_NONE = ~0u;

// LiDAR sweeps arrive in range-image order, so neighbouring points mostly fall into the same voxel.  Lanes of a
// wave that hold the same voxel as the lane before them form a RUN: only the run's first lane probes the key
// table; the others take its slot, and start the candidate cascade at their rank within the run.
//
// Candidate cascade (per voxel T words, all-ones = none): a value visits slot t with atomicMin, stays if the slot
// was empty, otherwise carries max(old, v) to slot t + 1.  The values visiting slot t + 1 are exactly those that
// visited slot t minus the minimum that stays, so slot t ends up holding the (t+1)-th smallest index -- for any
// arrival order, and also when a value starts at a slot j <= (number of smaller indices of the same voxel), which
// the rank within a run guarantees.  Members of rank >= T can never be kept and skip the table altogether.
__global__ __launch_bounds__(256) void vox_insert_kernel(const float *__restrict__ pts, int n,
                                                         int stride, int feat_off,
                                                         const int32_t *__restrict__ offs, int batch,
                                                         VoxGeom G, int T, int L, u64 *keys, u32 *best,
                                                         u32 mask, int32_t *pt_slot) {
  // (a workgroup takes the 256-point chunks blockIdx.x, + gridDim.x, ...: option "vox_grid" sizes the grid)
  for (int chunk = blockIdx.x; chunk * 256 < n; chunk += gridDim.x) {
    const int i = chunk * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    u64 key = KEY_EMPTY;
    if (i < n) {
        const float *p = pts + (size_t)i * stride + feat_off;
        float xyz[3] = {p[0], p[1], p[2]};
        int cx, cy, cz;
        if (voxel_coord(xyz, G, cx, cy, cz)) {
            int b = frame_of(offs, batch, i);
            key = (((u64)b * G.gz + cz) * G.gy + cy) * G.gx + cx;
        }
    }
    const bool valid = key != KEY_EMPTY;
    u64 prev_key = __shfl_up(key, 1);
    const bool head = lane == 0 || prev_key != key;
    const u64 heads = __ballot(head);
    const int head_lane = 63 - __clzll(heads & ((2ull << lane) - 1ull));
    const int run_rank = lane - head_lane;

    u32 h = 0;
    u32 last = IDX_NONE;
    if (head && valid) {
        h = hash_u64(key) & mask;
        for (;;) {
            u64 *kp = keys + (size_t)h * (L / 2);      // key and candidates of a slot share one 32-byte record
            u64 prev = *kp;
            // same sector as the key, so it comes back with it: the last kept candidate.  Values only ever
            // decrease -- a stale read can only be too large, never cause a wrong early exit below.
            last = __hip_atomic_load(best + (size_t)h * L + (T - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == key) break;
            if (prev == KEY_EMPTY) prev = atomicCAS(kp, KEY_EMPTY, key);
            if (prev == KEY_EMPTY || prev == key) break;
            h = (h + 1) & mask;
        }
    }
    h = __shfl(h, head_lane);
    if (i >= n) continue;
    if (!valid) {
        pt_slot[i] = -1;
        continue;
    }
    pt_slot[i] = (int32_t)h;
    u32 v = (u32)i;
    // T smaller indices are already kept (or precede us in the run): we can never enter
    if (run_rank >= T || last < v) continue;
    u32 *slot = best + (size_t)h * L;
    for (int t = run_rank; t < T; ++t) {
        u32 old = atomicMin(&slot[t], v);
        if (old == IDX_NONE) break;
        v = old > v ? old : v;
    }
  }
}

// "point i is the first point of its voxel": one random read of the candidate table per point.  The flag is
// evaluated ONCE, by the reduce pass of the scan, which leaves it in `flag` (the scan's output array, overwritten
// in place by the down pass); the down pass and the emit kernel read it back coalesced.
struct FirstFlag {
    const int32_t *pt_slot;
    const u32 *best;
    int L;   // slot stride in 32-bit words
    int *flag;
    __device__ int operator()(int i) const {
        int s = pt_slot[i];
        int v = (s >= 0 && best[(size_t)s * L] == (u32)i) ? 1 : 0;
        flag[i] = v;
        return v;
    }
};
struct StoredFlag {
    const int *flag;
    __device__ int operator()(int i) const { return flag[i]; }
};

// one block: per-frame counts, caps, output bases
__global__ void vox_frames_kernel(const int32_t *offs, int batch, const int *rank, int max_voxels,
                                  int cap, int *frame_rank0, int *frame_base, int32_t *voxel_counts) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int base = 0;
    for (int b = 0; b < batch; ++b) {
        int r0 = rank[offs[b]];
        int r1 = rank[offs[b + 1]];
        int m = r1 - r0;
        if (m > max_voxels) m = max_voxels;
        if (base + m > cap) m = cap - base;
        frame_rank0[b] = r0;
        frame_base[b] = base;
        voxel_counts[b] = m;
        base += m;
    }
    voxel_counts[batch] = base;
}

// Key-sorted row order (pcd_voxelize_hard_sorted): the KEPT voxels (first-appearance rank below the per-frame cap,
// exactly the reference's set) are numbered by their (b, z, y, x) key instead -- the order spconv's strided convs and
// torch.unique produce anyway -- so that the neighbours a level-1 conv gathers lie in nearby rows.  No sort: an
// occupancy bitmap over the key space, population counts per 16-byte GROUP (128 keys) taken by one coalesced pass
// over the bitmap, an exclusive scan of the group counts; a voxel's row = its group's prefix + the set bits below its
// own inside the group (16 + 4 bytes per lookup).  Bitmap + group prefixes ARE the coordinate -> row map of level 1:
// handed to the caller (rank_bitmap / rank_prefix) they replace the hash table of the level-1 SubM rulebook
// (pcd_rulebook_subm_ranked4).  Marking costs one memory-side atomic per voxel -- the only one: ~13 G
// atomics/s device-wide measured, and a second atomic per voxel on a per-chunk counter (contended: neighbouring
// voxels share chunks) took the mark kernel from 18 to 62 us.

__device__ __forceinline__ u32 vox_key_u32(const VoxGeom &G, int b, int cz, int cy, int cx) {
    if (G.order == PCD_ROWS_YXZ) return (((u32)b * G.gy + cy) * G.gx + cx) * G.kz + cz;
    return (((u32)b * G.kz + cz) * G.gy + cy) * G.gx + cx;
}

__global__ __launch_bounds__(256) void vox_sorted_mark_wide_kernel(
    const float *__restrict__ pts, int n, int stride, int feat_off, const int32_t *__restrict__ offs, int batch,
    VoxGeom G, const int *__restrict__ rank, const int *frame_rank0, const int32_t *voxel_counts, u32 *bitmap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int rk = rank[i];
    if (rank[i + 1] == rk) return;                 // not the first point of a voxel
    const int b = frame_of(offs, batch, i);
    if (rk - frame_rank0[b] >= voxel_counts[b]) return;
    const float *p0 = pts + (size_t)i * stride + feat_off;
    float xyz[3] = {p0[0], p0[1], p0[2]};
    int cx, cy, cz;
    voxel_coord(xyz, G, cx, cy, cz);
    const u32 key = vox_key_u32(G, b, cz, cy, cx);
    atomicOr(bitmap + (key >> 5), 1u << (key & 31));
}

// (also the per-frame table of vox_frames_kernel: every block derives it from the ranks at the frame boundaries --
//  batch + 1 loads and a serial loop over the frames by one thread -- and block 0 publishes it for the emit kernel)
constexpr int VOX_FOLD_FRAMES = 256;
__global__ __launch_bounds__(256) void vox_sorted_mark_kernel(
    const float *__restrict__ pts, int n, int stride, int feat_off, const int32_t *__restrict__ offs, int batch,
    VoxGeom G, const int *__restrict__ rank, int max_voxels, int cap, int *frame_rank0, int *frame_base,
    int32_t *voxel_counts, u32 *bitmap) {
    __shared__ int r_s[VOX_FOLD_FRAMES + 1];
    __shared__ int m_s[VOX_FOLD_FRAMES];
    for (int b = threadIdx.x; b <= batch; b += 256) r_s[b] = rank[offs[b]];
    __syncthreads();
    if (threadIdx.x == 0) {
        int base = 0;
        for (int b = 0; b < batch; ++b) {
            int m = r_s[b + 1] - r_s[b];
            if (m > max_voxels) m = max_voxels;
            if (base + m > cap) m = cap - base;
            m_s[b] = m;
            if (blockIdx.x == 0) {
                frame_rank0[b] = r_s[b];
                frame_base[b] = base;
                voxel_counts[b] = m;
            }
            base += m;
        }
        if (blockIdx.x == 0) voxel_counts[batch] = base;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int rk = rank[i];
    if (rank[i + 1] == rk) return;                 // not the first point of a voxel
    const int b = frame_of(offs, batch, i);
    if (rk - r_s[b] >= m_s[b]) return;
    const float *p0 = pts + (size_t)i * stride + feat_off;
    float xyz[3] = {p0[0], p0[1], p0[2]};
    int cx, cy, cz;
    voxel_coord(xyz, G, cx, cy, cz);
    const u32 key = vox_key_u32(G, b, cz, cy, cx);
    atomicOr(bitmap + (key >> 5), 1u << (key & 31));
}

// Sum of the block sums in front of block `blk`, taken by the block itself (no spine launch; up to VOX_DIRECT_BLOCKS
// blocks = 16 loads per thread -- beyond that the caller runs scan_spine_kernel and passes spined = 1: bsums then
// hold the exclusive prefixes already).  (Sums per 64 blocks gathered with atomicAdd by the reduce pass were slower
// than the spine launch they replaced: 64 memory-side atomics on one address serialise at ~150 ns each.)
constexpr int VOX_DIRECT_BLOCKS = 4096;
__device__ __forceinline__ int blocks_before(int blk, const int *__restrict__ bsums, int spined, int *wsum /*[4] LDS*/,
                                             int *base_s /*LDS*/) {
    if (spined) return bsums[blk];
    int part = 0;
    for (int j = threadIdx.x; j < blk; j += 256) part += bsums[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) *base_s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return *base_s;
}

// flags -> exclusive ranks in place, rank[n] = number of voxels
__global__ __launch_bounds__(256) void vox_flag_down_kernel(int *rank, int n, const int *__restrict__ bsums,
                                                            int spined) {
    __shared__ int lds[4];
    __shared__ int base_s;
    const int base = blocks_before(blockIdx.x, bsums, spined, lds, &base_s);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int v = (i < n) ? rank[i] : 0;
    int total;
    const int ex = block_exclusive_scan(v, lds, total);
    if (i < n) rank[i] = base + ex;
    if (i == n - 1) rank[n] = base + ex + v;
}

// ---- z-fastest rows: the level's COLUMN MAP instead of a bitmap over the key space ---------------------------------------
// (b, y, x, z) order: a voxel's row = first row of its BEV column + the set z bits below its own.  Two marks (BEV occupancy
// bit, then the z bit in the column's 64-bit mask, both order-free ORs), two scans (words -> column ranks, columns -> first
// rows): 1.1 MB + 16 bytes per column instead of a 46 MB bitmap (zero-filled, scanned and probed at random) + 11.6 MB of
// prefixes -- and the result IS level 1's coordinate -> row map for the rulebook builds (colmap.hip).
__global__ __launch_bounds__(256) void vox_cm_mark_kernel(
    const float *__restrict__ pts, int n, int stride, int feat_off, const int32_t *__restrict__ offs, int batch,
    VoxGeom G, const int *__restrict__ rank, int max_voxels, int cap, int *frame_rank0, int *frame_base,
    int32_t *voxel_counts, u32 *cbits, int pitch) {
    __shared__ int r_s[VOX_FOLD_FRAMES + 1];
    __shared__ int m_s[VOX_FOLD_FRAMES];
    for (int b = threadIdx.x; b <= batch; b += 256) r_s[b] = rank[offs[b]];
    __syncthreads();
    if (threadIdx.x == 0) {
        int base = 0;
        for (int b = 0; b < batch; ++b) {
            int m = r_s[b + 1] - r_s[b];
            if (m > max_voxels) m = max_voxels;
            if (base + m > cap) m = cap - base;
            m_s[b] = m;
            if (blockIdx.x == 0) {
                frame_rank0[b] = r_s[b];
                frame_base[b] = base;
                voxel_counts[b] = m;
            }
            base += m;
        }
        if (blockIdx.x == 0) voxel_counts[batch] = base;
    }
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int rk = rank[i];
        if (rank[i + 1] == rk) continue;               // not the first point of a voxel
        const int b = frame_of(offs, batch, i);
        if (rk - r_s[b] >= m_s[b]) continue;
        const float *p0 = pts + (size_t)i * stride + feat_off;
        float xyz[3] = {p0[0], p0[1], p0[2]};
        int cx, cy, cz;
        voxel_coord(xyz, G, cx, cy, cz);
        const u32 key = bev_key(b, cy, cx, G.gy, pitch);
        atomicOr(cbits + (key >> 5), 1u << (key & 31u));
    }
}

__global__ __launch_bounds__(256) void vox_cm_zmark_kernel(
    const float *__restrict__ pts, int n, int stride, int feat_off, const int32_t *__restrict__ offs, int batch,
    VoxGeom G, const int *__restrict__ rank, const int *__restrict__ frame_rank0, const int32_t *__restrict__ voxel_counts,
    const uint2 *__restrict__ cw, int ncol_cap, u64 *__restrict__ zm, int pitch) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int rk = rank[i];
        if (rank[i + 1] == rk) continue;
        const int b = frame_of(offs, batch, i);
        if (rk - frame_rank0[b] >= voxel_counts[b]) continue;
        const float *p0 = pts + (size_t)i * stride + feat_off;
        float xyz[3] = {p0[0], p0[1], p0[2]};
        int cx, cy, cz;
        voxel_coord(xyz, G, cx, cy, cz);
        const u32 key = bev_key(b, cy, cx, G.gy, pitch);
        const int col = cm_col(cw[key >> 5], key, ncol_cap);
        if (col >= 0) atomicOr((unsigned long long *)(zm + col), 1ull << cz);
    }
}

struct ZmCount {
    const u64 *zm;
    __device__ int operator()(int i) const { return __popcll(zm[i]); }
};

// z masks -> column records {mask, first row, rows}: the down pass of the scan over the columns
__global__ __launch_bounds__(256) void vox_cm_cols_kernel(const u64 *__restrict__ zm, int ncols_cap, const int *__restrict__ bsums,
                                                          int spined, uint4 *__restrict__ cr, int *__restrict__ ncols) {
    __shared__ int lds[4];
    __shared__ int base_s;
    const int base = blocks_before(blockIdx.x, bsums, spined, lds, &base_s);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const u64 m = i < ncols_cap ? zm[i] : 0ull;
    const int v = __popcll(m);
    int total;
    const int ex = block_exclusive_scan(v, lds, total);
    if (i < ncols_cap) cr[i] = make_uint4((u32)m, (u32)(m >> 32), (u32)(base + ex), (u32)v);
    if (i == ncols_cap - 1 && ncols) ncols[1] = base + ex + v;     // (rows; ncols[0] = columns, written by the word scan)
}

// Prefix GROUPS of 4 bitmap words (16 bytes, 128 keys): population count of every group.  A block covers 2048 groups
// in 8 coalesced rounds (whole 4-KiB block loads); also block sums.
constexpr int VOX_GROUPS_PER_BLOCK = 2048;
__global__ __launch_bounds__(256) void vox_group_count_kernel(const uint4 *__restrict__ bitmap4, int ngroups,
                                                              int *__restrict__ cnt, int *bsums) {
    __shared__ int lds[4];
    const int g0 = blockIdx.x * VOX_GROUPS_PER_BLOCK + threadIdx.x;
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = g0 + j * 256;
        v[j] = g < ngroups ? bitmap4[g] : make_uint4(0u, 0u, 0u, 0u);
    }
    int total = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int p = __popc(v[j].x) + __popc(v[j].y) + __popc(v[j].z) + __popc(v[j].w);
        total += p;
        const int g = g0 + j * 256;
        if (g < ngroups) cnt[g] = p;
    }
    int block_total;
    block_exclusive_scan(total, lds, block_total);
    if (threadIdx.x == 0) bsums[blockIdx.x] = block_total;
}

// group counts -> exclusive prefix over all groups, in place (8 consecutive groups per thread)
__global__ __launch_bounds__(256) void vox_group_prefix_kernel(int *cnt, int ngroups, const int *__restrict__ bsums,
                                                               int spined) {
    __shared__ int lds[4];
    __shared__ int base_s;
    const int base0 = blocks_before(blockIdx.x, bsums, spined, lds, &base_s);
    const int g0 = blockIdx.x * VOX_GROUPS_PER_BLOCK + threadIdx.x * 8;
    int v[8];
    if (g0 + 7 < ngroups) {
        const int4 a = *reinterpret_cast<const int4 *>(cnt + g0), b = *reinterpret_cast<const int4 *>(cnt + g0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = g0 + j < ngroups ? cnt[g0 + j] : 0;
    }
    int sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int t = v[j];
        v[j] = sum;
        sum += t;
    }
    int block_total;
    const int ex = base0 + block_exclusive_scan(sum, lds, block_total);
    if (g0 + 7 < ngroups) {
        *reinterpret_cast<int4 *>(cnt + g0) = make_int4(ex + v[0], ex + v[1], ex + v[2], ex + v[3]);
        *reinterpret_cast<int4 *>(cnt + g0 + 4) = make_int4(ex + v[4], ex + v[5], ex + v[6], ex + v[7]);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (g0 + j < ngroups) cnt[g0 + j] = ex + v[j];
    }
}

__device__ __forceinline__ int vox_sorted_row(const u32 *__restrict__ bitmap, const int *__restrict__ group_prefix,
                                              u32 key) {
    const uint4 q = reinterpret_cast<const uint4 *>(bitmap)[key >> 7];
    const u32 wi = (key >> 5) & 3u, sh = key & 31u;
    const u32 w = wi == 0 ? q.x : wi == 1 ? q.y : wi == 2 ? q.z : q.w;
    return group_prefix[key >> 7] + __popc(w & ((1u << sh) - 1u)) + (wi > 0 ? __popc(q.x) : 0) +
           (wi > 1 ? __popc(q.y) : 0) + (wi > 2 ? __popc(q.z) : 0);
}

__global__ __launch_bounds__(256) void vox_emit_kernel(
    const float *__restrict__ pts, int n, int stride, int feat_off, int C,
    const int32_t *__restrict__ offs, int batch, VoxGeom G, int T, int L, const u32 *__restrict__ best,
    const int32_t *__restrict__ pt_slot, const int *__restrict__ rank, const int *frame_rank0,
    const int *frame_base, const int32_t *voxel_counts, float *voxels, int32_t *coords,
    int32_t *num_points, float *mean_f32, unsigned short *mean_bf16, int bf16_stride, int cap_rows,
    const u32 *__restrict__ bitmap, const int *__restrict__ chunk_prefix, const uint2 *__restrict__ cw,
    const uint4 *__restrict__ cr, int ncol_cap, int pitch) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int rk = rank[i];
    if (rank[i + 1] == rk) return;                 // not the first point of a voxel (exclusive ranks: no step)
    const u32 *slot = best + (size_t)pt_slot[i] * L;
    int b = frame_of(offs, batch, i);
    int vid = rk - frame_rank0[b];
    if (vid >= voxel_counts[b]) return;  // beyond max_voxels (or capacity)
    int row = frame_base[b] + vid;
    const float *p0 = pts + (size_t)i * stride + feat_off;
    float xyz[3] = {p0[0], p0[1], p0[2]};
    int cx, cy, cz;
    voxel_coord(xyz, G, cx, cy, cz);
    if (cw) {      // z-fastest rows through the column map: first row of the BEV column + the set z bits below
        const u32 key = bev_key(b, cy, cx, G.gy, pitch);
        const int col = cm_col(cw[key >> 5], key, ncol_cap);
        if (col < 0) return;                       // (column beyond the capacity: its rows are dropped like any row beyond it)
        const uint4 r = cr[col];
        row = cm_row((u64)r.x | ((u64)r.y << 32), (int)r.z, cz);
        if (row < 0 || row >= cap_rows) return;
    } else if (bitmap)    // key order (frames are the key's major digit: frame b still owns rows frame_base[b] ..)
        row = vox_sorted_row(bitmap, chunk_prefix, vox_key_u32(G, b, cz, cy, cx));
    reinterpret_cast<int4 *>(coords)[row] = make_int4(b, cz, cy, cx);
    float sum[16];
    for (int c = 0; c < C; ++c) sum[c] = 0.0f;
    int np = 0;
    for (int t = 0; t < T; ++t) {
        u32 j = slot[t];
        if (j == IDX_NONE) {
            if (voxels)
                for (int c = 0; c < C; ++c) voxels[((size_t)row * T + t) * C + c] = 0.0f;
            continue;
        }
        ++np;
        const float *p = pts + (size_t)j * stride + feat_off;
        for (int c = 0; c < C; ++c) {
            float v = p[c];
            sum[c] += v;  // same order as voxels.sum(dim=1): t ascending
            if (voxels) voxels[((size_t)row * T + t) * C + c] = v;
        }
    }
    num_points[row] = np;
    float norm = np > 0 ? (float)np : 1.0f;
    if (mean_f32)
        for (int c = 0; c < C; ++c) mean_f32[(size_t)row * C + c] = __fdiv_rn(sum[c], norm);
    if (mean_bf16) {
        for (int c = 0; c < bf16_stride; ++c)
            mean_bf16[(size_t)row * bf16_stride + c] =
                c < C ? f32_to_bf16_bits(__fdiv_rn(sum[c], norm)) : (unsigned short)0;
    }
}

// z-fastest rows, emitted IN ROW ORDER (round 5): a workgroup takes 256 consecutive columns of the map -- their rows are one
// contiguous range -- and its threads walk that range: row -> column by a binary search over the 256 first rows in LDS, z =
// the k-th set bit of the column's mask, the voxel's record by a probe of the key table.  Every output row is written by the
// thread next to its neighbours' (whole lines instead of 16 / 4 / 32-byte pieces scattered over the arrays: the per-point form
// above moved 304 MB for ~50 MB of results, most of it lines fetched to be partially overwritten).  Same records, same
// arithmetic -> the same outputs bit for bit.
__device__ __forceinline__ int kth_set_bit64(u64 v, int k) {
    int pos = 0;
#pragma unroll
    for (int half = 32; half >= 1; half >>= 1) {
        const int c = __popcll(v & ((1ull << half) - 1ull));
        if (k >= c) {
            k -= c;
            v >>= half;
            pos += half;
        }
    }
    return pos;
}

template <int COLS>
__global__ __launch_bounds__(256) void vox_emit_rows_kernel(
    const float *__restrict__ pts, int stride, int feat_off, int C, VoxGeom G, int T, int L, const u64 *__restrict__ keys,
    const u32 *__restrict__ best, u32 mask, float *voxels, int32_t *coords, int32_t *num_points, float *mean_f32,
    unsigned short *mean_bf16, int bf16_stride, int cap_rows, const uint4 *__restrict__ cr, const u32 *__restrict__ colkey,
    const int *__restrict__ ncols, int ncol_cap, int pitch) {
    __shared__ int start_s[COLS + 1];
    __shared__ u32 zlo_s[COLS], zhi_s[COLS], key_s[COLS];
    const int nc = min(ncols[0], ncol_cap);
    const int c0 = blockIdx.x * COLS;
    if (c0 >= nc) return;
    const int c = c0 + threadIdx.x;
    if (threadIdx.x < COLS) {
        uint4 r = make_uint4(0u, 0u, 0u, 0u);
        if (c < nc) r = cr[c];
        start_s[threadIdx.x] = c < nc ? (int)r.z : 0x7fffffff;
        zlo_s[threadIdx.x] = r.x;
        zhi_s[threadIdx.x] = r.y;
        key_s[threadIdx.x] = c < nc ? colkey[c] : 0u;
        if (threadIdx.x == min(COLS - 1, nc - 1 - c0)) start_s[COLS] = (int)(r.z + r.w);    // (end of the last live column)
    }
    __syncthreads();
    const int nlive = min(COLS, nc - c0);
    const int r_begin = start_s[0], r_end = min(start_s[COLS], cap_rows);
    for (int row = r_begin + threadIdx.x; row < r_end; row += 256) {
        int lo = 0;                                           // last column with start <= row
#pragma unroll
        for (int step = COLS / 2; step >= 1; step >>= 1)
            if (lo + step < nlive && start_s[lo + step] <= row) lo += step;
        const u64 zm = (u64)zlo_s[lo] | ((u64)zhi_s[lo] << 32);
        const int cz = kth_set_bit64(zm, row - start_s[lo]);
        const u32 bk = key_s[lo];
        const int cx = (int)(bk % (u32)pitch), by = (int)(bk / (u32)pitch), cy = by % G.gy, b = by / G.gy;
        const u64 key = (((u64)b * G.gz + cz) * G.gy + cy) * G.gx + cx;
        u32 h = hash_u64(key) & mask;
        // (the voxel is in the table -- it was marked from there; the probe count is bounded all the same)
        int probes = 0;
        while (keys[(size_t)h * (L / 2)] != key && probes < 4096) {
            h = (h + 1) & mask;
            ++probes;
        }
        if (probes >= 4096) continue;
        const u32 *slot = best + (size_t)h * L;
        reinterpret_cast<int4 *>(coords)[row] = make_int4(b, cz, cy, cx);
        float sum[16];
        for (int ch = 0; ch < C; ++ch) sum[ch] = 0.0f;
        int np = 0;
        for (int t = 0; t < T; ++t) {
            const u32 j = slot[t];
            if (j == IDX_NONE) {
                if (voxels)
                    for (int ch = 0; ch < C; ++ch) voxels[((size_t)row * T + t) * C + ch] = 0.0f;
                continue;
            }
            ++np;
            const float *p = pts + (size_t)j * stride + feat_off;
            for (int ch = 0; ch < C; ++ch) {
                const float v = p[ch];
                sum[ch] += v;  // same order as voxels.sum(dim=1): t ascending
                if (voxels) voxels[((size_t)row * T + t) * C + ch] = v;
            }
        }
        num_points[row] = np;
        const float norm = np > 0 ? (float)np : 1.0f;
        if (mean_f32)
            for (int ch = 0; ch < C; ++ch) mean_f32[(size_t)row * C + ch] = __fdiv_rn(sum[ch], norm);
        if (mean_bf16) {
            for (int ch = 0; ch < bf16_stride; ++ch)
                mean_bf16[(size_t)row * bf16_stride + ch] =
                    ch < C ? f32_to_bf16_bits(__fdiv_rn(sum[ch], norm)) : (unsigned short)0;
        }
    }
}

__global__ __launch_bounds__(256) void mean_vfe_kernel(const float *__restrict__ voxels,
                                                       const int32_t *__restrict__ nump, int m,
                                                       int T, int C, float *out) {
    int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= m * C) return;
    int r = e / C, c = e - r * C;
    float s = 0.0f;
    for (int t = 0; t < T; ++t) s += voxels[((size_t)r * T + t) * C + c];
    float nrm = (float)nump[r];
    nrm = nrm < 1.0f ? 1.0f : nrm;
    out[e] = __fdiv_rn(s, nrm);
}

// ---------------------------------------------------------------------------------------------
// dynamic voxelisation
__global__ __launch_bounds__(256) void dyn_mark_kernel(const float *__restrict__ pts, int n, int C,
                                                       VoxGeom G, int batch, u32 *bitmap,
                                                       u32 *pt_key) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *p = pts + (size_t)i * (C + 1);
    float xyz[3] = {p[1], p[2], p[3]};
    int cx, cy, cz;
    int b = (int)p[0];
    if (!voxel_coord(xyz, G, cx, cy, cz) || b < 0 || b >= batch) {
        pt_key[i] = IDX_NONE;
        return;
    }
    // reference key order: b, x, y, z   (dynamic_mean_vfe.py:57-60)
    u32 key = (((u32)b * G.gx + cx) * G.gy + cy) * G.gz + cz;
    pt_key[i] = key;
    u32 bit = 1u << (key & 31);
    u32 *w = bitmap + (key >> 5);
    if (!(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(w, bit);
}

struct PopcWord {
    const u32 *bitmap;
    __device__ int operator()(int i) const { return __popc(bitmap[i]); }
};

__global__ __launch_bounds__(256) void dyn_emit_coords_kernel(const u32 *__restrict__ bitmap,
                                                              const int *__restrict__ prefix,
                                                              int nwords, VoxGeom G, int cap,
                                                              int32_t *coords) {
    int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nwords) return;
    u32 bits = bitmap[w];
    int r = prefix[w];
    while (bits) {
        int bpos = __ffs(bits) - 1;
        bits &= bits - 1;
        u32 key = ((u32)w << 5) + bpos;
        int cz = key % G.gz;
        u32 t = key / G.gz;
        int cy = t % G.gy;
        t /= G.gy;
        int cx = t % G.gx;
        int b = t / G.gx;
        if (r < cap) reinterpret_cast<int4 *>(coords)[r] = make_int4(b, cz, cy, cx);
        ++r;
    }
}

__global__ __launch_bounds__(256) void dyn_accum_kernel(const float *__restrict__ pts, int n, int C,
                                                        const u32 *__restrict__ pt_key,
                                                        const u32 *__restrict__ bitmap,
                                                        const int *__restrict__ prefix, int cap,
                                                        float *sums, int *cnt, int32_t *point_voxel) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32 key = pt_key[i];
    if (key == IDX_NONE) {
        if (point_voxel) point_voxel[i] = -1;
        return;
    }
    u32 w = key >> 5;
    int r = prefix[w] + __popc(bitmap[w] & ((1u << (key & 31)) - 1u));
    if (point_voxel) point_voxel[i] = r < cap ? r : -1;    // = unq_inv of torch.unique(sorted keys)
    if (r >= cap) return;
    const float *p = pts + (size_t)i * (C + 1) + 1;
    for (int c = 0; c < C; ++c) atomicAdd(&sums[(size_t)r * C + c], p[c]);
    atomicAdd(&cnt[r], 1);
}

__global__ __launch_bounds__(256) void dyn_finalize_kernel(float *feat, const int *cnt,
                                                           const int *num_voxels, int cap, int C) {
    int e = blockIdx.x * 256 + threadIdx.x;
    int m = *num_voxels;
    if (m > cap) m = cap;
    if (e >= m * C) return;
    int r = e / C;
    feat[e] = __fdiv_rn(feat[e], (float)cnt[r]);
}

static VoxGeom make_geom(const float *range, const float *vs) {
    VoxGeom G;
    G.r0 = range[0]; G.r1 = range[1]; G.r2 = range[2];
    G.v0 = vs[0]; G.v1 = vs[1]; G.v2 = vs[2];
    int g[3];
    for (int j = 0; j < 3; ++j) {
        double d = ((double)range[j + 3] - (double)range[j]) / (double)vs[j];
        g[j] = (int)llround(d);  // data_processor.py:127-128
    }
    G.gx = g[0]; G.gy = g[1]; G.gz = g[2];
    G.kz = G.gz;
    G.order = PCD_ROWS_ZYX;
    return G;
}

// power of two >= 1.5 x points: at most one voxel per point, so the load factor is <= 2/3 in the worst case and
// ~1/3 on LiDAR frames (about two points per voxel); every halving keeps more of the key / candidate tables in L2
// 32-bit words per hash record: 64-bit key + T candidate indices, padded to an even count (8-byte aligned keys)
static int slot_words(int max_points) { return (2 + max_points + 1) & ~1; }

static u32 table_capacity(int n) {
    u32 cap = 1024;
    const u32 need = (u32)(n > 0 ? n : 1);
    while (cap < need + need / 2) cap <<= 1;
    return cap;
}

}  // namespace

// =============================================================================================
static bool sorted_words(int batch, const VoxGeom &G, size_t *nwords, size_t *nchunks) {
    const double vol = (double)batch * G.gx * G.gy * G.kz;
    if (vol >= 4294967295.0 - 1024.0) return false;
    *nchunks = ((size_t)vol + 127) / 128;            // prefix groups of 4 words
    *nwords = *nchunks * 4;
    return true;
}

static size_t hard_workspace_bytes(int n_points, int max_points, int batch, const VoxGeom *G, int cm_cap = 0) {
    if (n_points < 0 || max_points <= 0 || batch <= 0) return 0;
    u32 cap = table_capacity(n_points);
    size_t b = 0;
    if (G && cm_cap > 0) {         // z-fastest rows through the column map (no key-space bitmap)
        CmBuf B;
        if (!cm_carve(nullptr, 0, batch, G->gy, G->gx, cm_cap, B, nullptr)) return 0;
        b += ws_piece(B.nwords + 4, sizeof(u32));                          // BEV occupancy bits
        b += ws_piece(pcd_div_up((int)B.nwords, 1024) + 2, sizeof(int));   // their block sums
        b += ws_piece((size_t)B.ncol_cap + 1, sizeof(u64));                // z masks by column
        b += ws_piece(pcd_div_up(B.ncol_cap, 256) + 2, sizeof(int));       // block sums of the column scan
        b += ws_piece((size_t)B.ncol_cap + 1, sizeof(u32));                // column -> BEV key
    } else if (G) {
        size_t nw, nc;
        if (!sorted_words(batch, *G, &nw, &nc)) return 0;
        b += ws_piece(nw, sizeof(u32));                    // occupancy bitmap of the kept voxels (unless the caller's)
        b += ws_piece(nc + 8, sizeof(int));                // group counts -> prefix, in place (unless the caller's)
        b += ws_piece(pcd_div_up((int)nc, VOX_GROUPS_PER_BLOCK) + 2, sizeof(int));        // block sums
    }
    b += ws_piece((size_t)cap * slot_words(max_points), sizeof(u32));   // {key u64, best u32[T]} records
    b += ws_piece(n_points + 1, sizeof(int32_t));          // pt_slot
    b += ws_piece(n_points + 1, sizeof(int));              // rank
    b += ws_piece(pcd_div_up(n_points, 256) + 2, sizeof(int));
    b += ws_piece(batch + 1, sizeof(int)) * 2;
    return b;
}

extern "C" size_t pcd_voxelize_hard_workspace_bytes(int n_points, int max_points, int batch) {
    return hard_workspace_bytes(n_points, max_points, batch, nullptr);
}

extern "C" size_t pcd_voxelize_hard_sorted_workspace_bytes(int n_points, int max_points, int batch,
                                                           const float *range_host, const float *vsize_host,
                                                           int key_depth) {
    // (the key space has the same size in both row orders)
    if (!range_host || !vsize_host) return 0;
    VoxGeom G = make_geom(range_host, vsize_host);
    if (key_depth > 0 && key_depth < G.gz) return 0;
    if (key_depth > 0) G.kz = key_depth;
    return hard_workspace_bytes(n_points, max_points, batch, &G);
}

static int voxelize_hard_impl(const float *points, int n_points, int point_stride,
                              int feat_offset, int num_features, const int32_t *frame_offsets,
                              int batch, const float *range_host, const float *vsize_host,
                              int max_points, int max_voxels, int cap, float *voxels,
                              int32_t *coords, int32_t *num_points, float *mean_f32,
                              void *mean_bf16, int mean_bf16_stride, int32_t *voxel_counts,
                              void *workspace, size_t workspace_bytes, void *stream, bool key_order,
                              uint32_t *rank_bitmap, int32_t *rank_prefix, int key_depth, int row_order,
                              void *colmap = nullptr, size_t colmap_bytes = 0) {
    PCD_ENTER();
    if (row_order != PCD_ROWS_ZYX && row_order != PCD_ROWS_YXZ) return PCD_ERR_INVALID_ARG;
    if (n_points < 0 || batch <= 0 || max_points <= 0 || max_voxels < 0 || cap < 0 ||
        !frame_offsets || !range_host || !vsize_host || !coords || !num_points || !voxel_counts)
        return PCD_ERR_INVALID_ARG;
    if (n_points > 0 && !points) return PCD_ERR_INVALID_ARG;
    if (num_features < 3 || num_features > 16 || point_stride < feat_offset + num_features)
        return PCD_ERR_UNSUPPORTED;
    if (mean_bf16 && mean_bf16_stride < num_features) return PCD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    VoxGeom G = make_geom(range_host, vsize_host);
    if (key_depth > 0) {
        if (key_depth < G.gz) return PCD_ERR_INVALID_ARG;
        G.kz = key_depth;
    }
    G.order = row_order;
    if ((double)batch * G.gx * G.gy * G.gz >= 1.8e19) return PCD_ERR_KEYSPACE;
    size_t nw = 0, nc = 0;
    const bool cm = colmap != nullptr;             // z-fastest rows through the column map, which the caller keeps
    if (cm && (!key_order || row_order != PCD_ROWS_YXZ || G.kz > 62 || batch > VOX_FOLD_FRAMES)) return PCD_ERR_UNSUPPORTED;
    if (key_order && !cm && !sorted_words(batch, G, &nw, &nc)) return PCD_ERR_KEYSPACE;
    if (workspace_bytes < hard_workspace_bytes(n_points, max_points, batch, key_order ? &G : nullptr, cm ? (cap > 0 ? cap : 1) : 0))
        return PCD_ERR_WORKSPACE;
    WsCarver ws(workspace, workspace_bytes);
    u32 *bitmap = nullptr;
    int *chunk_bsums = nullptr, *chunk_prefix = nullptr;
    CmBuf CB = {};
    u32 *cbits = nullptr;
    int *cbsums = nullptr, *colsums = nullptr;
    u64 *zm = nullptr;
    u32 *colkey = nullptr;
    if (cm) {
        if (!cm_carve(colmap, colmap_bytes, batch, G.gy, G.gx, cap > 0 ? cap : 1, CB, nullptr)) return PCD_ERR_WORKSPACE;
        cbits = ws.take<u32>(CB.nwords + 4);
        cbsums = ws.take<int>(pcd_div_up((int)CB.nwords, 1024) + 2);
        zm = ws.take<u64>((size_t)CB.ncol_cap + 1);
        colsums = ws.take<int>(pcd_div_up(CB.ncol_cap, 256) + 2);
        colkey = ws.take<u32>((size_t)CB.ncol_cap + 1);
    } else if (key_order) {
        bitmap = ws.take<u32>(nw);
        chunk_prefix = ws.take<int>(nc + 8);
        chunk_bsums = ws.take<int>(pcd_div_up((int)nc, VOX_GROUPS_PER_BLOCK) + 2);
        if (rank_bitmap && rank_prefix) {               // the caller keeps the coordinate -> row map
            bitmap = rank_bitmap;
            chunk_prefix = rank_prefix;
        }
    }
    u32 tcap = table_capacity(n_points);
    const int L = slot_words(max_points);
    u32 *tab = ws.take<u32>((size_t)tcap * L);
    u64 *keys = (u64 *)tab;     // record h: key at word h*L, candidates at words h*L + 2 ..
    u32 *best = tab + 2;
    int32_t *pt_slot = ws.take<int32_t>(n_points + 1);
    int *rank = ws.take<int>(n_points + 1);
    int *bsums = ws.take<int>(pcd_div_up(n_points, 256) + 2);
    int *frame_rank0 = ws.take<int>(batch + 1);
    int *frame_base = ws.take<int>(batch + 1);
    if (!ws.ok) return PCD_ERR_WORKSPACE;
    int nb = pcd_div_up(n_points, 256);
    const bool fast_sorted = key_order && !cm && n_points > 0 && batch <= VOX_FOLD_FRAMES;
    if (cm) {
        // (the bits, their block sums and the z masks lie next to each other in the workspace: one fill)
        pcd_fill(cbits, 0, (size_t)((char *)(zm + CB.ncol_cap + 1) - (char *)cbits), st);
    } else if (key_order && n_points > 0)
        pcd_fill(bitmap, 0, nw * sizeof(u32), st);
    // (forking this fill onto a helper stream beside the insert pass -- event fork / join inside the call -- crashed
    //  the HIP runtime when the call was captured into a graph from a stream that had itself joined the capture)
    // one memset to 0xFF sets both sentinels (empty key, no candidate)
    pcd_fill(tab, 0xFF, (size_t)tcap * L * sizeof(u32), st);
    if (n_points > 0) {
        const int vg = pcd_opt(PCD_OPT_VOX_GRID);
        vox_insert_kernel<<<vg > 0 && vg < nb ? vg : nb, 256, 0, st>>>(points, n_points, point_stride, feat_offset,
                                              frame_offsets, batch, G, max_points, L, keys, best,
                                              tcap - 1, pt_slot);
        PCD_RETURN_IF_LAUNCH_FAILED();
    }
    int rc = PCD_OK;
    auto chunk_scan = [&]() {
        const int ncb = pcd_div_up((int)nc, VOX_GROUPS_PER_BLOCK);
        const int spined = ncb > VOX_DIRECT_BLOCKS;
        vox_group_count_kernel<<<ncb, 256, 0, st>>>((const uint4 *)bitmap, (int)nc, chunk_prefix, chunk_bsums);
        if (spined) scan_spine_kernel<<<1, 256, 0, st>>>(chunk_bsums, ncb, nullptr);
        vox_group_prefix_kernel<<<ncb, 256, 0, st>>>(chunk_prefix, (int)nc, chunk_bsums, spined);
    };
    if (cm && n_points > 0) {
        FirstFlag ff{pt_slot, best, L, rank};
        const int spined = nb > VOX_DIRECT_BLOCKS;
        scan_reduce_kernel<FirstFlag><<<nb, 256, 0, st>>>(ff, n_points, bsums);
        if (spined) scan_spine_kernel<<<1, 256, 0, st>>>(bsums, nb, nullptr);
        vox_flag_down_kernel<<<nb, 256, 0, st>>>(rank, n_points, bsums, spined);
        const int vgm = pcd_opt(PCD_OPT_VOX_GRID) > 0 && pcd_opt(PCD_OPT_VOX_GRID) < nb ? pcd_opt(PCD_OPT_VOX_GRID) : nb;
        vox_cm_mark_kernel<<<vgm, 256, 0, st>>>(points, n_points, point_stride, feat_offset, frame_offsets, batch, G, rank,
                                               max_voxels, cap, frame_rank0, frame_base, voxel_counts, cbits, CB.pitch);
        const int nwords = (int)CB.nwords, nwb = pcd_div_up(nwords, 1024);
        cm_words_count_kernel<<<nwb, 256, 0, st>>>(cbits, nwords, cbsums);
        const int wsp = cm_spined(nwb);
        if (wsp) scan_spine_kernel<<<1, 256, 0, st>>>(cbsums, nwb, nullptr);
        cm_words_prefix_kernel<<<nwb, 256, 0, st>>>(cbits, nwords, nwb, cbsums, wsp, CB.cw, CB.ncols, colkey, CB.ncol_cap);
        vox_cm_zmark_kernel<<<vgm, 256, 0, st>>>(points, n_points, point_stride, feat_offset, frame_offsets, batch, G, rank,
                                                frame_rank0, voxel_counts, CB.cw, CB.ncol_cap, zm, CB.pitch);
        const int ncb = pcd_div_up(CB.ncol_cap, 256);
        const int csp = ncb > VOX_DIRECT_BLOCKS;
        scan_reduce_kernel<ZmCount><<<ncb, 256, 0, st>>>(ZmCount{zm}, CB.ncol_cap, colsums);
        if (csp) scan_spine_kernel<<<1, 256, 0, st>>>(colsums, ncb, nullptr);
        vox_cm_cols_kernel<<<ncb, 256, 0, st>>>(zm, CB.ncol_cap, colsums, csp, CB.cr, CB.ncols);
        PCD_RETURN_IF_LAUNCH_FAILED();
    } else if (cm) {
        // no points: an empty map (all-zero words with zero prefixes), zero frame counts
        const int nwords = (int)CB.nwords, nwb = pcd_div_up(nwords, 1024);
        cm_words_count_kernel<<<nwb, 256, 0, st>>>(cbits, nwords, cbsums);
        cm_words_prefix_kernel<<<nwb, 256, 0, st>>>(cbits, nwords, nwb, cbsums, 0, CB.cw, CB.ncols);
        rc = scan_exclusive(StoredFlag{rank}, 0, rank, bsums, nullptr, st);
        if (rc != PCD_OK) return rc;
        vox_frames_kernel<<<1, 64, 0, st>>>(frame_offsets, batch, rank, max_voxels, cap, frame_rank0, frame_base, voxel_counts);
    } else if (fast_sorted) {
        // 6 launches behind the insert pass (the generic form below: 10): block sums added up by the consuming blocks
        // instead of scan spines, the per-frame table folded into the mark kernel
        FirstFlag ff{pt_slot, best, L, rank};
        const int spined = nb > VOX_DIRECT_BLOCKS;
        scan_reduce_kernel<FirstFlag><<<nb, 256, 0, st>>>(ff, n_points, bsums);
        if (spined) scan_spine_kernel<<<1, 256, 0, st>>>(bsums, nb, nullptr);
        vox_flag_down_kernel<<<nb, 256, 0, st>>>(rank, n_points, bsums, spined);
        vox_sorted_mark_kernel<<<nb, 256, 0, st>>>(points, n_points, point_stride, feat_offset, frame_offsets, batch,
                                                   G, rank, max_voxels, cap, frame_rank0, frame_base, voxel_counts,
                                                   bitmap);
        chunk_scan();
        PCD_RETURN_IF_LAUNCH_FAILED();
    } else {
        if (n_points > 0) {
            FirstFlag ff{pt_slot, best, L, rank};
            StoredFlag sf{rank};
            scan_reduce_kernel<FirstFlag><<<nb, 256, 0, st>>>(ff, n_points, bsums);
            scan_spine_kernel<<<1, 256, 0, st>>>(bsums, nb, nullptr);
            scan_down_kernel<StoredFlag><<<nb, 256, 0, st>>>(sf, n_points, bsums, rank);   // in place
            PCD_RETURN_IF_LAUNCH_FAILED();
        } else {
            rc = scan_exclusive(StoredFlag{rank}, 0, rank, bsums, nullptr, st);
            if (rc != PCD_OK) return rc;
        }
        vox_frames_kernel<<<1, 64, 0, st>>>(frame_offsets, batch, rank, max_voxels, cap, frame_rank0,
                                            frame_base, voxel_counts);
        if (n_points > 0 && key_order) {     // (more than VOX_FOLD_FRAMES frames)
            vox_sorted_mark_wide_kernel<<<nb, 256, 0, st>>>(points, n_points, point_stride, feat_offset, frame_offsets,
                                                            batch, G, rank, frame_rank0, voxel_counts, bitmap);
            chunk_scan();
            PCD_RETURN_IF_LAUNCH_FAILED();
        }
    }
    if (n_points > 0 && cm && pcd_opt(PCD_OPT_VOX_EMIT_ROWS)) {
        // (256 columns ~ 800 rows per workgroup, three rows per thread one after the other: 118 us in the graph and a step of
        //  3.07 ms; 64 columns per workgroup -- one row per thread, four times the workgroups -- 74 us and 3.11 ms: the kernel runs
        //  beside level 2 of the forward pass, and what it costs the step is the CUs it takes from the window kernels there, not
        //  its own duration.  The per-point form: 66 us, 3.09 ms.)
#define VOX_EMIT_ROWS(COLS)                                                                                                   \
    vox_emit_rows_kernel<COLS><<<pcd_div_up(CB.ncol_cap, COLS), 256, 0, st>>>(                                                 \
        points, point_stride, feat_offset, num_features, G, max_points, L, keys, best, tcap - 1, voxels, coords, num_points,   \
        mean_f32, (unsigned short *)mean_bf16, mean_bf16_stride, cap, CB.cr, colkey, CB.ncols, CB.ncol_cap, CB.pitch)
        switch (pcd_opt(PCD_OPT_VOX_EMIT_ROWS)) {        // (option value = columns per workgroup; 1 = 256)
            case 64: VOX_EMIT_ROWS(64); break;
            case 128: VOX_EMIT_ROWS(128); break;
            default: VOX_EMIT_ROWS(256); break;
        }
#undef VOX_EMIT_ROWS
    } else if (n_points > 0) {
        vox_emit_kernel<<<nb, 256, 0, st>>>(points, n_points, point_stride, feat_offset,
                                            num_features, frame_offsets, batch, G, max_points, L, best,
                                            pt_slot, rank, frame_rank0, frame_base, voxel_counts,
                                            voxels, coords, num_points, mean_f32,
                                            (unsigned short *)mean_bf16, mean_bf16_stride, cap, bitmap, chunk_prefix,
                                            cm ? CB.cw : nullptr, cm ? CB.cr : nullptr, cm ? CB.ncol_cap : 0, cm ? CB.pitch : 0);
    }
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_voxelize_hard(const float *points, int n_points, int point_stride,
                                 int feat_offset, int num_features, const int32_t *frame_offsets,
                                 int batch, const float *range_host, const float *vsize_host,
                                 int max_points, int max_voxels, int cap, float *voxels,
                                 int32_t *coords, int32_t *num_points, float *mean_f32,
                                 void *mean_bf16, int mean_bf16_stride, int32_t *voxel_counts,
                                 void *workspace, size_t workspace_bytes, void *stream) {
    return voxelize_hard_impl(points, n_points, point_stride, feat_offset, num_features, frame_offsets, batch,
                              range_host, vsize_host, max_points, max_voxels, cap, voxels, coords, num_points,
                              mean_f32, mean_bf16, mean_bf16_stride, voxel_counts, workspace, workspace_bytes, stream,
                              false, nullptr, nullptr, 0, PCD_ROWS_ZYX);
}

extern "C" int pcd_voxelize_hard_sorted_rank_words(int batch, const float *range_host, const float *vsize_host,
                                                   int key_depth, size_t *bitmap_words, size_t *prefix_words) {
    if (batch <= 0 || !range_host || !vsize_host || !bitmap_words || !prefix_words) return PCD_ERR_INVALID_ARG;
    VoxGeom G = make_geom(range_host, vsize_host);
    if (key_depth > 0 && key_depth < G.gz) return PCD_ERR_INVALID_ARG;
    if (key_depth > 0) G.kz = key_depth;
    size_t nw, nc;
    if (!sorted_words(batch, G, &nw, &nc)) return PCD_ERR_KEYSPACE;
    *bitmap_words = nw;
    *prefix_words = nc + 8;
    return PCD_OK;
}

extern "C" int pcd_voxelize_hard_sorted(const float *points, int n_points, int point_stride,
                                        int feat_offset, int num_features, const int32_t *frame_offsets,
                                        int batch, const float *range_host, const float *vsize_host,
                                        int max_points, int max_voxels, int cap, float *voxels,
                                        int32_t *coords, int32_t *num_points, float *mean_f32,
                                        void *mean_bf16, int mean_bf16_stride, int32_t *voxel_counts,
                                        int key_depth, int row_order, uint32_t *rank_bitmap,
                                        int32_t *rank_prefix, void *workspace, size_t workspace_bytes,
                                        void *stream) {
    if ((rank_bitmap != nullptr) != (rank_prefix != nullptr) || key_depth < 0) return PCD_ERR_INVALID_ARG;
    return voxelize_hard_impl(points, n_points, point_stride, feat_offset, num_features, frame_offsets, batch,
                              range_host, vsize_host, max_points, max_voxels, cap, voxels, coords, num_points,
                              mean_f32, mean_bf16, mean_bf16_stride, voxel_counts, workspace, workspace_bytes, stream,
                              true, rank_bitmap, rank_prefix, key_depth, row_order);
}

// pcd_voxelize_hard_sorted with row_order PCD_ROWS_YXZ whose coordinate -> row map is the level's COLUMN MAP (colmap: a
// buffer of pcd_colmap_bytes(batch, (key_depth or gz, gy, gx), cap) bytes the caller keeps for the rulebook builds), built
// by the voxeliser itself from two order-free marks and two small scans -- no bitmap over the (b, y, x, z) key space.
extern "C" size_t pcd_voxelize_hard_yxz_workspace_bytes(int n_points, int max_points, int batch, const float *range_host,
                                                        const float *vsize_host, int key_depth, int cap) {
    if (!range_host || !vsize_host) return 0;
    VoxGeom G = make_geom(range_host, vsize_host);
    if (key_depth > 0 && key_depth < G.gz) return 0;
    if (key_depth > 0) G.kz = key_depth;
    if (G.kz > 62 || batch > VOX_FOLD_FRAMES) return 0;
    return hard_workspace_bytes(n_points, max_points, batch, &G, cap > 0 ? cap : 1);
}

extern "C" int pcd_voxelize_hard_yxz(const float *points, int n_points, int point_stride, int feat_offset,
                                     int num_features, const int32_t *frame_offsets, int batch, const float *range_host,
                                     const float *vsize_host, int max_points, int max_voxels, int cap, float *voxels,
                                     int32_t *coords, int32_t *num_points, float *mean_f32, void *mean_bf16,
                                     int mean_bf16_stride, int32_t *voxel_counts, int key_depth, void *colmap,
                                     size_t colmap_bytes, void *workspace, size_t workspace_bytes, void *stream) {
    if (!colmap || key_depth < 0) return PCD_ERR_INVALID_ARG;
    return voxelize_hard_impl(points, n_points, point_stride, feat_offset, num_features, frame_offsets, batch,
                              range_host, vsize_host, max_points, max_voxels, cap, voxels, coords, num_points,
                              mean_f32, mean_bf16, mean_bf16_stride, voxel_counts, workspace, workspace_bytes, stream,
                              true, nullptr, nullptr, key_depth, PCD_ROWS_YXZ, colmap, colmap_bytes);
}

// Host-side hard voxelisation of ONE frame (the loop spconv's CPU voxel generator runs): what the reference
// calls inside forked DataLoader workers (pcdet/datasets/processor/data_processor.py:44-60,130-141), where no HIP call is
// possible.  HOST pointers, no stream, no GPU work; the same coordinate arithmetic as the kernels (voxel_coord), the
// reference's first-appearance ids, per-voxel first `max_points` points and `max_voxels` cut.  voxels [max_voxels][T][C]
// (zero padded), coords [max_voxels][3] (z, y, x), num_points [max_voxels]; *num_voxels_out = M.
#include <unordered_map>
extern "C" int pcd_voxelize_hard_host(const float *points_host, int n_points, int point_stride, int num_features,
                                      const float *range_host, const float *vsize_host, int max_points, int max_voxels,
                                      float *voxels_host, int32_t *coords_host, int32_t *num_points_host,
                                      int32_t *num_voxels_out) {
    if (n_points < 0 || max_points <= 0 || max_voxels < 0 || !range_host || !vsize_host || !num_voxels_out)
        return PCD_ERR_INVALID_ARG;
    if (num_features < 3 || point_stride < num_features) return PCD_ERR_UNSUPPORTED;
    if (n_points > 0 && (!points_host || !voxels_host || !coords_host || !num_points_host)) return PCD_ERR_INVALID_ARG;
    const VoxGeom G = make_geom(range_host, vsize_host);
    std::unordered_map<unsigned long long, int> cell;
    cell.reserve((size_t)(n_points > 0 ? n_points : 1));
    int m = 0;
    for (int i = 0; i < n_points; ++i) {
        const float *p = points_host + (size_t)i * point_stride;
        int cx, cy, cz;
        if (!voxel_coord(p, G, cx, cy, cz)) continue;
        const unsigned long long key = ((unsigned long long)cz * G.gy + cy) * G.gx + cx;
        auto it = cell.find(key);
        int v;
        if (it == cell.end()) {
            if (m >= max_voxels) continue;         // never registered: its later points are skipped too
            v = m++;
            cell.emplace(key, v);
            coords_host[(size_t)v * 3 + 0] = cz;
            coords_host[(size_t)v * 3 + 1] = cy;
            coords_host[(size_t)v * 3 + 2] = cx;
            num_points_host[v] = 0;
            for (size_t e = 0; e < (size_t)max_points * num_features; ++e) voxels_host[(size_t)v * max_points * num_features + e] = 0.0f;
        } else {
            v = it->second;
        }
        const int np = num_points_host[v];
        if (np < max_points) {
            for (int c = 0; c < num_features; ++c) voxels_host[((size_t)v * max_points + np) * num_features + c] = p[c];
            num_points_host[v] = np + 1;
        }
    }
    *num_voxels_out = m;
    return PCD_OK;
}

extern "C" int pcd_mean_vfe(const float *voxels, const int32_t *num_points, int m, int max_points,
                            int num_features, float *out, void *stream) {
    PCD_ENTER();
    if (m < 0 || max_points <= 0 || num_features <= 0) return PCD_ERR_INVALID_ARG;
    if (m == 0) return PCD_OK;
    if (!voxels || !num_points || !out) return PCD_ERR_INVALID_ARG;
    mean_vfe_kernel<<<pcd_div_up(m * num_features, 256), 256, 0, (hipStream_t)stream>>>(
        voxels, num_points, m, max_points, num_features, out);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

// ---------------------------------------------------------------------------------------------
static int dyn_words(int batch, const VoxGeom &G, size_t *nwords) {
    double vol = (double)batch * G.gx * G.gy * G.gz;
    if (vol >= 4294967295.0) return PCD_ERR_KEYSPACE;
    *nwords = ((size_t)vol + 31) / 32;
    return PCD_OK;
}

extern "C" size_t pcd_voxelize_dynamic_workspace_bytes(int n_points, int num_features, int batch,
                                                       const float *range_host,
                                                       const float *vsize_host) {
    if (n_points < 0 || batch <= 0 || !range_host || !vsize_host) return 0;
    (void)num_features;
    VoxGeom G = make_geom(range_host, vsize_host);
    size_t nw;
    if (dyn_words(batch, G, &nw) != PCD_OK) return 0;
    size_t b = 0;
    b += ws_piece(nw, sizeof(u32));
    b += ws_piece(nw + 1, sizeof(int));
    b += ws_piece(pcd_div_up((int)nw, 256) + 2, sizeof(int));
    b += ws_piece(n_points + 1, sizeof(u32));
    b += ws_piece(n_points + 1, sizeof(int));
    return b;
}

extern "C" int pcd_voxelize_dynamic_mean(const float *points_b, int n_points, int num_features,
                                         int batch, const float *range_host,
                                         const float *vsize_host, int cap, float *features,
                                         int32_t *coords, int32_t *counts, int32_t *num_voxels,
                                         int32_t *point_voxel, void *workspace, size_t workspace_bytes,
                                         void *stream) {
    PCD_ENTER();
    if (n_points < 0 || batch <= 0 || cap < 0 || !range_host || !vsize_host || !num_voxels)
        return PCD_ERR_INVALID_ARG;
    if (num_features < 3 || num_features > 16) return PCD_ERR_UNSUPPORTED;
    if (n_points > 0 && (!points_b || !features || !coords)) return PCD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    VoxGeom G = make_geom(range_host, vsize_host);
    size_t nw;
    int rc = dyn_words(batch, G, &nw);
    if (rc != PCD_OK) return rc;
    WsCarver ws(workspace, workspace_bytes);
    u32 *bitmap = ws.take<u32>(nw);
    int *prefix = ws.take<int>(nw + 1);
    int *bsums = ws.take<int>(pcd_div_up((int)nw, 256) + 2);
    u32 *pt_key = ws.take<u32>(n_points + 1);
    int *cnt_ws = ws.take<int>(n_points + 1);
    if (!ws.ok) return PCD_ERR_WORKSPACE;
    int *cnt = counts ? counts : cnt_ws;
    int rows = cap < n_points ? cap : n_points;
    pcd_fill(bitmap, 0, nw * sizeof(u32), st);
    if (rows > 0) {
        pcd_fill(features, 0, (size_t)rows * num_features * sizeof(float), st);
        pcd_fill(cnt, 0, (size_t)rows * sizeof(int), st);
    }
    int nb = pcd_div_up(n_points, 256);
    if (n_points > 0)
        dyn_mark_kernel<<<nb, 256, 0, st>>>(points_b, n_points, num_features, G, batch, bitmap,
                                            pt_key);
    PopcWord pw{bitmap};
    rc = scan_exclusive(pw, (int)nw, prefix, bsums, num_voxels, st);
    if (rc != PCD_OK) return rc;
    if (n_points > 0) {
        dyn_emit_coords_kernel<<<pcd_div_up((int)nw, 256), 256, 0, st>>>(bitmap, prefix, (int)nw, G,
                                                                        cap, coords);
        dyn_accum_kernel<<<nb, 256, 0, st>>>(points_b, n_points, num_features, pt_key, bitmap,
                                             prefix, cap, features, cnt, point_voxel);
        dyn_finalize_kernel<<<pcd_div_up(rows * num_features, 256), 256, 0, st>>>(
            features, cnt, num_voxels, cap, num_features);
    }
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
