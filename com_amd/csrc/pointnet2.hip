// PV-RCNN stage-2 natives (BASELINE config 4) -- the "stack" PointNet++ ops of
// pcdet/ops/pointnet2/pointnet2_stack/src: ball_query_gpu.cu:16-83, group_points_gpu.cu:15-118,
// sampling_gpu.cu:188-348, interpolate_gpu.cu:16-186, voxel_query_gpu.cu:10-105 (binder pointnet2_api.cpp).
// Results are defined by the reference kernels' sequential semantics (first `nsample` hits in ascending point index,
// strict `<` / `>` tie rules); the MI355X forms below reproduce those semantics with 64-lane waves instead of one
// thread walking all points:
//   * ball query: one WAVE per query centre; 64 points are tested per step, a ballot + prefix popcount keeps the
//     hits in index order, the loop stops once nsample are found (the reference: one thread, N serial iterations);
//   * three_nn: one wave per unknown point; every lane keeps its own 3 best of a strided subset, three rounds of a
//     wave-wide (distance, index) minimum merge them -- the same 3 neighbours in the same order as the serial scan;
//   * stack FPS: one 1024-thread workgroup per batch element (the sequential dependence over the samples is
//     inherent), argmax through wave shuffles + one LDS hop, with the tie rule of the reference's reduction tree;
//   * grouping / interpolation forward are gathers; their gradients are scatter-adds with fp32 atomics as in the
//     reference (order-dependent rounding; everything else here is deterministic).
#include "common.h"

namespace {

__device__ __forceinline__ int batch_of(const int32_t *cnt, int B, int i, int *start_other, const int32_t *other_cnt) {
    int b = 0, acc = cnt[0];
    for (int k = 1; k < B; ++k) {
        if (i < acc) break;
        acc += cnt[k];
        b = k;
    }
    int s = 0;
    for (int k = 0; k < b; ++k) s += other_cnt[k];
    *start_other = s;
    return b;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ball_query_kernel(int B, int M, float radius, int nsample,
                                                         const float *__restrict__ new_xyz,
                                                         const int32_t *__restrict__ new_cnt, const float *__restrict__ xyz,
                                                         const int32_t *__restrict__ xyz_cnt, int32_t *__restrict__ idx) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= M) return;
    const int lane = lane_id();
    int start;
    const int b = batch_of(new_cnt, B, q, &start, xyz_cnt);
    const int n = xyz_cnt[b];
    const float *pts = xyz + (size_t)start * 3;
    const float r2 = radius * radius;
    const float cx = new_xyz[(size_t)q * 3], cy = new_xyz[(size_t)q * 3 + 1], cz = new_xyz[(size_t)q * 3 + 2];
    int32_t *out = idx + (size_t)q * nsample;
    int cnt = 0, first = -1;
    for (int base = 0; base < n && cnt < nsample; base += 64) {
        const int k = base + lane;
        bool hit = false;
        if (k < n) {
            const float x = pts[(size_t)k * 3], y = pts[(size_t)k * 3 + 1], z = pts[(size_t)k * 3 + 2];
            const float d2 = (cx - x) * (cx - x) + (cy - y) * (cy - y) + (cz - z) * (cz - z);
            hit = d2 < r2;
        }
        int total;
        const int rank = wave_rank(hit, total);
        if (total) {
            if (first < 0) first = base + __builtin_ctzll(__ballot(hit));
            if (hit && cnt + rank < nsample) out[cnt + rank] = k;
            cnt += total;
        }
    }
    if (cnt == 0) {
        if (lane == 0) out[0] = -1;                        // (ball_query_gpu.cu:65: the caller zeroes empty balls)
    } else {
        cnt = min(cnt, nsample);
        for (int l = cnt + lane; l < nsample; l += 64) out[l] = first;   // unfilled slots repeat the first hit (:55-59)
    }
}

// out[m][c][s] = features[start_b + idx[m][s]][c]
__global__ __launch_bounds__(256) void group_points_kernel(int B, int M, int C, int nsample,
                                                           const float *__restrict__ features,
                                                           const int32_t *__restrict__ feat_cnt,
                                                           const int32_t *__restrict__ idx,
                                                           const int32_t *__restrict__ idx_cnt, float *__restrict__ out) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)M * C * nsample) return;
    const int s = (int)(e % nsample), c = (int)((e / nsample) % C), m = (int)(e / nsample / C);
    int start;
    batch_of(idx_cnt, B, m, &start, feat_cnt);
    out[e] = features[((size_t)start + idx[(size_t)m * nsample + s]) * C + c];
}

__global__ __launch_bounds__(256) void group_points_grad_kernel(int B, int M, int C, int nsample,
                                                                const float *__restrict__ grad_out,
                                                                const int32_t *__restrict__ idx,
                                                                const int32_t *__restrict__ idx_cnt,
                                                                const int32_t *__restrict__ feat_cnt,
                                                                float *__restrict__ grad_features) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)M * C * nsample) return;
    const int s = (int)(e % nsample), c = (int)((e / nsample) % C), m = (int)(e / nsample / C);
    int start;
    batch_of(idx_cnt, B, m, &start, feat_cnt);
    atomicAdd(grad_features + ((size_t)start + idx[(size_t)m * nsample + s]) * C + c, grad_out[e]);
}

// The same scatter with the CHANNEL as the fastest thread index: the 64 lanes of a wave add into 256 contiguous bytes of one
// point's gradient row instead of into 64 different rows (one memory-side atomic per cache line touched: the kernel above ran
// at 1.5 G atomics/s, 146 ms per launch of the RoI-grid pooling's backward pass).  grad_out is [M][C][nsample] (sample fastest,
// group_points_grad_gpu in the reference): a workgroup takes one m, reads its C x nsample tile coalesced into LDS (rows padded by
// one float) and scatters it transposed.  Sums are float atomics as in the reference (order not fixed).
__global__ __launch_bounds__(256) void group_points_grad_tiled_kernel(int B, int M, int C, int nsample,
                                                                      const float *__restrict__ grad_out,
                                                                      const int32_t *__restrict__ idx,
                                                                      const int32_t *__restrict__ idx_cnt,
                                                                      const int32_t *__restrict__ feat_cnt,
                                                                      float *__restrict__ grad_features) {
    extern __shared__ float gp_tile[];           // [C][nsample + 1]
    __shared__ int rows[64];
    const int m = blockIdx.x;
    int start;
    batch_of(idx_cnt, B, m, &start, feat_cnt);
    const float *src = grad_out + (size_t)m * C * nsample;
    for (int e = threadIdx.x; e < C * nsample; e += 256) gp_tile[(e / nsample) * (nsample + 1) + e % nsample] = src[e];
    __shared__ int first[64];
    if ((int)threadIdx.x < nsample) rows[threadIdx.x] = start + idx[(size_t)m * nsample + threadIdx.x];
    __syncthreads();
    // a ball with fewer than nsample neighbours repeats its first index in the remaining slots (ball_query): the gradients of
    // equal indices are added up in LDS first, one atomic per DISTINCT point and channel leaves the workgroup
    if ((int)threadIdx.x < nsample) {
        int f = threadIdx.x;
        for (int s2 = 0; s2 < (int)threadIdx.x; ++s2)
            if (rows[s2] == rows[threadIdx.x]) {
                f = s2;
                break;
            }
        first[threadIdx.x] = f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float *row = gp_tile + c * (nsample + 1);
        for (int s = 1; s < nsample; ++s) {
            const int f = first[s];
            if (f != s) row[f] += row[s];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * nsample; e += 256) {
        const int s = e / C, c = e - s * C;
        if (first[s] == s) atomicAdd(grad_features + (size_t)rows[s] * C + c, gp_tile[c * (nsample + 1) + s]);
    }
}

// ---------------------------------------------------------------------------------------------
// farthest point sampling, one workgroup per batch element.  key order of the argmax: larger distance, then the tie
// rule of the reference's 1024-slot reduction tree (sampling_gpu.cu:14-19,252-322: slot t merges with slot t + off,
// the LEFT slot wins ties, off = 512 .. 1): after the step `off` a slot holds the winner of the threads congruent to
// it mod off, so among equal distances the thread whose id has the smaller BIT-REVERSED value wins; inside a
// thread the first (smallest) index wins (strict >).
struct FpsBest {
    float d;
    int k;
};
__device__ __forceinline__ bool fps_better(const FpsBest &a, const FpsBest &b) {   // a beats b
    if (a.d != b.d) return a.d > b.d;
    const unsigned ta = __builtin_bitreverse32((unsigned)(a.k & 1023)), tb = __builtin_bitreverse32((unsigned)(b.k & 1023));
    if (ta != tb) return ta < tb;
    return a.k < b.k;
}

__global__ __launch_bounds__(1024) void stack_fps_kernel(const float *__restrict__ xyz, float *__restrict__ temp,
                                                         const int32_t *__restrict__ xyz_cnt, int32_t *__restrict__ idxs,
                                                         const int32_t *__restrict__ num_sampled) {
    __shared__ float wd[16];
    __shared__ int wk[16];
    __shared__ int old_s;
    const int b = blockIdx.x;
    int start = 0, ostart = 0;
    for (int k = 0; k < b; ++k) {
        start += xyz_cnt[k];
        ostart += num_sampled[k];
    }
    const float *pts = xyz + (size_t)start * 3;
    float *tmp = temp + start;
    int32_t *out = idxs + ostart;
    const int n = xyz_cnt[b], m = num_sampled[b];
    const int tid = threadIdx.x, wave = tid >> 6;
    if (tid == 0 && m > 0) out[0] = start;
    int old = 0;
    for (int j = 1; j < m; ++j) {
        const float x1 = pts[(size_t)old * 3], y1 = pts[(size_t)old * 3 + 1], z1 = pts[(size_t)old * 3 + 2];
        FpsBest best = {-1.0f, 0};
        for (int k = tid; k < n; k += 1024) {
            const float x2 = pts[(size_t)k * 3], y2 = pts[(size_t)k * 3 + 1], z2 = pts[(size_t)k * 3 + 2];
            const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
            const float d2 = fminf(d, tmp[k]);
            tmp[k] = d2;
            if (d2 > best.d) {
                best.d = d2;
                best.k = k;
            }
        }
        // threads without a point keep (-1, 0) with thread id = tid: give them their own tid as tie key
        if (best.d < 0.0f) best.k = tid;               // never wins (every real distance is >= 0)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            FpsBest o;
            o.d = __shfl_xor(best.d, off);
            o.k = __shfl_xor(best.k, off);
            if (fps_better(o, best)) best = o;
        }
        if ((tid & 63) == 0) {
            wd[wave] = best.d;
            wk[wave] = best.k;
        }
        __syncthreads();
        if (tid < 64) {
            FpsBest v = {tid < 16 ? wd[tid] : -2.0f, tid < 16 ? wk[tid] : 0x7fffffff};
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                FpsBest o;
                o.d = __shfl_xor(v.d, off);
                o.k = __shfl_xor(v.k, off);
                if (fps_better(o, v)) v = o;
            }
            if (tid == 0) {
                old_s = v.k;
                out[j] = v.k + start;
            }
        }
        __syncthreads();
        old = old_s;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// COOPERATIVE farthest point sampling for large frames (VoxelSetAbstraction samples 4096 keypoints from ~160 k raw
// points per frame, voxel_set_abstraction.py:236-263): the reference -- and stack_fps_kernel above -- give ONE
// workgroup per frame 4095 dependent passes over all of the frame's points (~30 us each at 160 k points: 120 ms).
// Here G workgroups share a frame: each keeps a slice of the points and their running distances in LDS, finds its
// local farthest point, publishes (distance, index) and meets the others at a per-frame barrier (a monotonic
// arrival counter, agent scope); every workgroup then reduces the G candidates itself.  The argmax key is the TOTAL
// order fps_better (distance, then the reference tree's bit-reversed thread id, then the index), so the selected points
// are the reference's whatever the partition.  All B x G workgroups must be resident at once (the host sizes G from the
// device's CU count and the occupancy query); the spin is bounded and raises `err` instead of hanging the device.
constexpr int FPS_COOP_MAXG = 64;

__global__ __launch_bounds__(256) void stack_fps_coop_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ xyz_cnt,
                                                             int32_t *__restrict__ idxs,
                                                             const int32_t *__restrict__ num_sampled, int G, int slice_cap,
                                                             unsigned long long *__restrict__ cand, unsigned *__restrict__ counters,
                                                             int *__restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) float fps_lds[];
    float *px = fps_lds, *py = px + slice_cap, *pz = py + slice_cap, *tmp = pz + slice_cap;
    __shared__ float wd[4];
    __shared__ int wk[4];
    __shared__ int old_s;
    // frame -> workgroups: blockIdx = gi * B + b.  Workgroup i runs on XCD i % 8 (observed; speed only), so a frame's G
    // workgroups sit on the XCDs {b, b + B, ..} mod 8 -- two of the eight at B = 4, one at B = 8 -- instead of all eight:
    // most of the per-iteration candidate exchange stays inside one XCD's L2.
    const int Bf = (int)gridDim.x / G;
    const int b = blockIdx.x % Bf, gi = blockIdx.x / Bf;
    int start = 0, ostart = 0;
    for (int k = 0; k < b; ++k) {
        start += xyz_cnt[k];
        ostart += num_sampled[k];
    }
    const int n = xyz_cnt[b], m = num_sampled[b];
    const float *pts = xyz + (size_t)start * 3;
    int32_t *out = idxs + ostart;
    const int per = (n + G - 1) / G;
    const int lo = min(n, gi * per), hi = min(n, lo + per);
    const int cnt = hi - lo;                                // <= slice_cap (checked by the host against max_cnt)
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int k = tid; k < cnt; k += 256) {
        px[k] = pts[(size_t)(lo + k) * 3];
        py[k] = pts[(size_t)(lo + k) * 3 + 1];
        pz[k] = pts[(size_t)(lo + k) * 3 + 2];
        tmp[k] = 1e10f;
    }
    if (gi == 0 && tid == 0 && m > 0) out[0] = start;
    __syncthreads();
    unsigned long long *cb = cand + (size_t)b * 2 * FPS_COOP_MAXG;
    (void)counters;
    int old = 0;
    for (int j = 1; j < m; ++j) {
        const float x1 = pts[(size_t)old * 3], y1 = pts[(size_t)old * 3 + 1], z1 = pts[(size_t)old * 3 + 2];
        FpsBest best = {-1.0f, tid};                       // (never wins: every real distance is >= 0)
        for (int k = tid; k < cnt; k += 256) {
            const float x2 = px[k], y2 = py[k], z2 = pz[k];
            const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
            const float d2 = fminf(d, tmp[k]);
            tmp[k] = d2;
            const FpsBest c = {d2, lo + k};
            if (fps_better(c, best)) best = c;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            FpsBest o;
            o.d = __shfl_xor(best.d, off);
            o.k = __shfl_xor(best.k, off);
            if (fps_better(o, best)) best = o;
        }
        if ((tid & 63) == 0) {
            wd[wave] = best.d;
            wk[wave] = best.k;
        }
        __syncthreads();
        if (tid == 0) {
            FpsBest v = {wd[0], wk[0]};
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const FpsBest o = {wd[w], wk[w]};
                if (fps_better(o, v)) v = o;
            }
            // one 64-bit word per candidate: [63:52] iteration tag, [51:20] distance bits, [19:0] frame-local index
            const unsigned long long packed = ((unsigned long long)(j & 0xFFF) << 52) |
                                              ((unsigned long long)__float_as_uint(v.d) << 20) | (unsigned)(v.k & 0xFFFFF);
            __hip_atomic_store(cb + (j & 1) * FPS_COOP_MAXG + gi, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // every workgroup polls the G slots of this iteration's buffer until all carry the tag j (ONE round trip per
        // iteration: no arrival counter).  Buffer parity: a slot is rewritten at iteration j + 2 at the earliest, which
        // needs every workgroup's candidate j + 1, i.e. every workgroup has finished reading iteration j.
        if (tid < 64) {
            FpsBest v = {-2.0f, 0x7fffffff};
            int spins = 0;
            for (;;) {
                bool ok = true;
                if (tid < G) {
                    const unsigned long long pk = __hip_atomic_load(cb + (j & 1) * FPS_COOP_MAXG + tid, __ATOMIC_RELAXED,
                                                                    __HIP_MEMORY_SCOPE_AGENT);
                    ok = (int)(pk >> 52) == (j & 0xFFF);
                    v.d = __uint_as_float((unsigned)(pk >> 20));
                    v.k = (int)(pk & 0xFFFFFull);
                }
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) {                               // ~ seconds: a workgroup of the frame is not resident
                    if (tid == 0) *err = 1;
                    break;
                }
            }
            if (tid >= G) v = {-2.0f, 0x7fffffff};
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                FpsBest o;
                o.d = __shfl_xor(v.d, off);
                o.k = __shfl_xor(v.k, off);
                if (fps_better(o, v)) v = o;
            }
            if (tid == 0) {
                old_s = v.k;
                if (gi == 0) out[j] = v.k + start;
            }
        }
        __syncthreads();
        old = old_s;
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // (everyone leaves within an iteration)
    }
}

// ---------------------------------------------------------------------------------------------
struct Nn3 {
    float d[3];
    int k[3];
};
__device__ __forceinline__ bool nn_less(float d1, int k1, float d2, int k2) { return d1 < d2 || (d1 == d2 && k1 < k2); }

__global__ __launch_bounds__(256) void three_nn_kernel(int B, int N, const float *__restrict__ unknown,
                                                       const int32_t *__restrict__ unknown_cnt,
                                                       const float *__restrict__ known,
                                                       const int32_t *__restrict__ known_cnt, float *__restrict__ dist2,
                                                       int32_t *__restrict__ idx) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= N) return;
    const int lane = lane_id();
    int start;
    const int b = batch_of(unknown_cnt, B, q, &start, known_cnt);
    const int m = known_cnt[b];
    const float *kn = known + (size_t)start * 3;
    const float ux = unknown[(size_t)q * 3], uy = unknown[(size_t)q * 3 + 1], uz = unknown[(size_t)q * 3 + 2];
    const float INF = 3.0e38f;                                 // (the reference starts from 1e40 in double)
    Nn3 t = {{INF, INF, INF}, {0x7fffffff, 0x7fffffff, 0x7fffffff}};
    for (int k = lane; k < m; k += 64) {
        const float x = kn[(size_t)k * 3], y = kn[(size_t)k * 3 + 1], z = kn[(size_t)k * 3 + 2];
        const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
        if (nn_less(d, k, t.d[0], t.k[0])) {
            t.d[2] = t.d[1]; t.k[2] = t.k[1];
            t.d[1] = t.d[0]; t.k[1] = t.k[0];
            t.d[0] = d; t.k[0] = k;
        } else if (nn_less(d, k, t.d[1], t.k[1])) {
            t.d[2] = t.d[1]; t.k[2] = t.k[1];
            t.d[1] = d; t.k[1] = k;
        } else if (nn_less(d, k, t.d[2], t.k[2])) {
            t.d[2] = d; t.k[2] = k;
        }
    }
    float od[3];
    int ok[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {          // r-th smallest (distance, index) over the wave; its owner pops it
        float bd = t.d[0];
        int bk = t.k[0];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(bd, off);
            const int oi = __shfl_xor(bk, off);
            if (nn_less(o, oi, bd, bk)) {
                bd = o;
                bk = oi;
            }
        }
        od[r] = bd;
        ok[r] = bk;
        if (t.k[0] == bk && t.d[0] == bd) {
            t.d[0] = t.d[1]; t.k[0] = t.k[1];
            t.d[1] = t.d[2]; t.k[1] = t.k[2];
            t.d[2] = INF; t.k[2] = 0x7fffffff;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            // fewer than 3 known points: the reference leaves (1e40 -> inf as float, index 0)
            const bool none = ok[r] == 0x7fffffff;
            dist2[(size_t)q * 3 + r] = none ? __builtin_inff() : od[r];
            idx[(size_t)q * 3 + r] = (none ? 0 : ok[r]) + start;
        }
    }
}

__global__ __launch_bounds__(256) void three_interpolate_kernel(int N, int C, const float *__restrict__ features,
                                                                const int32_t *__restrict__ idx,
                                                                const float *__restrict__ weight, float *__restrict__ out) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)N * C) return;
    const int p = (int)(e / C), c = (int)(e - (size_t)p * C);
    const int32_t *id = idx + (size_t)p * 3;
    const float *w = weight + (size_t)p * 3;
    out[e] = w[0] * features[(size_t)id[0] * C + c] + w[1] * features[(size_t)id[1] * C + c] +
             w[2] * features[(size_t)id[2] * C + c];
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(int N, int C, const float *__restrict__ grad_out,
                                                                     const int32_t *__restrict__ idx,
                                                                     const float *__restrict__ weight,
                                                                     float *__restrict__ grad_features) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)N * C) return;
    const int p = (int)(e / C), c = (int)(e - (size_t)p * C);
    const int32_t *id = idx + (size_t)p * 3;
    const float *w = weight + (size_t)p * 3;
    const float g = grad_out[e];
#pragma unroll
    for (int r = 0; r < 3; ++r) atomicAdd(grad_features + (size_t)id[r] * C + c, g * w[r]);
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void voxel_query_kernel(int M, int R1, int R2, int R3, int nsample, float radius,
                                                          int zr, int yr, int xr, const float *__restrict__ new_xyz,
                                                          const float *__restrict__ xyz,
                                                          const int32_t *__restrict__ new_coords,
                                                          const int32_t *__restrict__ point_indices,
                                                          int32_t *__restrict__ idx) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= M) return;
    const float r2 = radius * radius;
    const float nx = new_xyz[(size_t)q * 3], ny = new_xyz[(size_t)q * 3 + 1], nz = new_xyz[(size_t)q * 3 + 2];
    const int32_t *cd = new_coords + (size_t)q * 4;
    int32_t *out = idx + (size_t)q * nsample;
    int cnt = 0;
    for (int dz = -zr; dz <= zr; ++dz) {
        const int z = cd[1] + dz;
        if (z < 0 || z >= R1) continue;
        for (int dy = -yr; dy <= yr; ++dy) {
            const int y = cd[2] + dy;
            if (y < 0 || y >= R2) continue;
            for (int dx = -xr; dx <= xr; ++dx) {
                const int x = cd[3] + dx;
                if (x < 0 || x >= R3) continue;
                const int nb = point_indices[(((size_t)cd[0] * R1 + z) * R2 + y) * R3 + x];
                if (nb < 0) continue;
                const float px = xyz[(size_t)nb * 3], py = xyz[(size_t)nb * 3 + 1], pz = xyz[(size_t)nb * 3 + 2];
                const float d2 = (px - nx) * (px - nx) + (py - ny) * (py - ny) + (pz - nz) * (pz - nz);
                if (d2 > r2) continue;
                if (cnt < nsample) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) out[l] = nb;
                    out[cnt] = nb;
                    ++cnt;
                }
            }
        }
    }
    if (cnt == 0) out[0] = -1;
}

}  // namespace

#define PN2_CHECK_LAUNCH()             \
    PCD_RETURN_IF_LAUNCH_FAILED();     \
    return PCD_OK

extern "C" int pcd_ball_query_stack(int B, int M, float radius, int nsample, const float *new_xyz,
                                    const int32_t *new_xyz_batch_cnt, const float *xyz, const int32_t *xyz_batch_cnt,
                                    int32_t *idx, void *stream) {
    PCD_ENTER();
    if (B <= 0 || M < 0 || nsample <= 0) return PCD_ERR_INVALID_ARG;
    if (M == 0) return PCD_OK;
    if (!new_xyz || !new_xyz_batch_cnt || !xyz || !xyz_batch_cnt || !idx) return PCD_ERR_INVALID_ARG;
    ball_query_kernel<<<pcd_div_up(M, 4), 256, 0, (hipStream_t)stream>>>(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt,
                                                                        xyz, xyz_batch_cnt, idx);
    PN2_CHECK_LAUNCH();
}

extern "C" int pcd_group_points_stack(int B, int M, int C, int nsample, const float *features,
                                      const int32_t *features_batch_cnt, const int32_t *idx, const int32_t *idx_batch_cnt,
                                      float *out, void *stream) {
    PCD_ENTER();
    if (B <= 0 || M < 0 || C <= 0 || nsample <= 0) return PCD_ERR_INVALID_ARG;
    if (M == 0) return PCD_OK;
    if (!features || !features_batch_cnt || !idx || !idx_batch_cnt || !out) return PCD_ERR_INVALID_ARG;
    const size_t n = (size_t)M * C * nsample;
    group_points_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(B, M, C, nsample, features,
                                                                                    features_batch_cnt, idx, idx_batch_cnt, out);
    PN2_CHECK_LAUNCH();
}

extern "C" int pcd_group_points_stack_grad(int B, int M, int C, int nsample, const float *grad_out, const int32_t *idx,
                                           const int32_t *idx_batch_cnt, const int32_t *features_batch_cnt,
                                           float *grad_features_zeroed, void *stream) {
    PCD_ENTER();
    if (B <= 0 || M < 0 || C <= 0 || nsample <= 0) return PCD_ERR_INVALID_ARG;
    if (M == 0) return PCD_OK;
    if (!grad_out || !idx || !idx_batch_cnt || !features_batch_cnt || !grad_features_zeroed) return PCD_ERR_INVALID_ARG;
    const size_t n = (size_t)M * C * nsample;
    const size_t tile_bytes = (size_t)C * (nsample + 1) * sizeof(float);
    if (nsample <= 64 && tile_bytes <= 48 * 1024) {
        group_points_grad_tiled_kernel<<<(unsigned)M, 256, tile_bytes, (hipStream_t)stream>>>(
            B, M, C, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features_zeroed);
    } else {          // (balls of more than 64 samples / tiles beyond 48 KB: one atomic per element)
        group_points_grad_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(
            B, M, C, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features_zeroed);
    }
    PN2_CHECK_LAUNCH();
}

extern "C" int pcd_stack_farthest_point_sampling(int B, const float *xyz, float *temp_1e10, const int32_t *xyz_batch_cnt,
                                                 int32_t *idxs, const int32_t *num_sampled_points, void *stream) {
    PCD_ENTER();
    if (B <= 0 || !xyz || !temp_1e10 || !xyz_batch_cnt || !idxs || !num_sampled_points) return PCD_ERR_INVALID_ARG;
    stack_fps_kernel<<<B, 1024, 0, (hipStream_t)stream>>>(xyz, temp_1e10, xyz_batch_cnt, idxs, num_sampled_points);
    PN2_CHECK_LAUNCH();
}

extern "C" size_t pcd_stack_fps_coop_workspace_bytes(int B) {
    if (B <= 0) return 0;
    return ws_piece((size_t)B * 2 * FPS_COOP_MAXG, sizeof(unsigned long long)) + ws_piece((size_t)B * 32, sizeof(unsigned)) + 256;
}

// max_cnt_host: an upper bound of every frame's point count (the host knows it: the counts come from its collate step).
// Returns PCD_ERR_UNSUPPORTED when the cooperative form does not apply (too many frames for one workgroup per CU):
// call pcd_stack_farthest_point_sampling then.
extern "C" int pcd_stack_farthest_point_sampling_coop(int B, const float *xyz, const int32_t *xyz_batch_cnt, int32_t *idxs,
                                                      const int32_t *num_sampled_points, int max_cnt_host, void *workspace,
                                                      size_t workspace_bytes, void *stream) {
    PCD_ENTER();
    if (B <= 0 || !xyz || !xyz_batch_cnt || !idxs || !num_sampled_points || max_cnt_host <= 0) return PCD_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < pcd_stack_fps_coop_workspace_bytes(B)) return PCD_ERR_WORKSPACE;
    // All B x G workgroups spin on each other: they must be CO-RESIDENT.  G comes from the device the call runs on -- its
    // CU count and what the occupancy query admits per CU for this kernel's LDS footprint -- never from a constant; one
    // workgroup per CU is the design point (the slice of a frame lives in LDS), and a quarter of the CUs is left to whatever
    // else the caller has in flight on other streams.  (The spin is bounded: if residency still fails, `err` is raised and
    // the caller re-runs the one-workgroup kernel.)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return PCD_ERR_LAUNCH;
    int G = (cus - cus / 4) / B;                 // three quarters of the CUs
    if (G > FPS_COOP_MAXG) G = FPS_COOP_MAXG;
    const int g_opt = pcd_opt(PCD_OPT_FPS_G);   // (experiments: workgroups per frame)
    if (g_opt >= 2 && g_opt <= G) G = g_opt;
    if (G < 2) return PCD_ERR_UNSUPPORTED;
    const int slice_cap = pcd_div_up(max_cnt_host, G);
    const size_t lds = (size_t)slice_cap * 16;
    if (lds > 96 * 1024 || max_cnt_host > (1 << 20)) return PCD_ERR_UNSUPPORTED;   // (one workgroup per CU; 20-bit indices)
    if (lds > 64 * 1024 &&      // per call: the attribute is per device and the call is idempotent (no process-wide cache)
        hipFuncSetAttribute((const void *)stack_fps_coop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PCD_ERR_LAUNCH;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stack_fps_coop_kernel, 256, lds) != hipSuccess)
        return PCD_ERR_LAUNCH;
    if (per_cu < 1 || (long long)B * G > (long long)cus * (per_cu < 8 ? per_cu : 8))
        return PCD_ERR_UNSUPPORTED;            // the grid cannot be resident at once on this device
    hipStream_t st = (hipStream_t)stream;
    char *w = (char *)workspace;
    unsigned long long *cand = (unsigned long long *)w;
    unsigned *counters = (unsigned *)(w + ws_piece((size_t)B * 2 * FPS_COOP_MAXG, sizeof(unsigned long long)));
    int *err = (int *)((char *)counters + ws_piece((size_t)B * 32, sizeof(unsigned)));
    pcd_fill(workspace, 0, pcd_stack_fps_coop_workspace_bytes(B), st);   // candidate tags 0 (no iteration has tag 0 before j = 4096), err = 0
    stack_fps_coop_kernel<<<B * G, 256, lds, st>>>(xyz, xyz_batch_cnt, idxs, num_sampled_points, G, slice_cap, cand, counters,
                                                  err);
    PN2_CHECK_LAUNCH();
}

extern "C" int pcd_three_nn_stack(int B, int N, const float *unknown, const int32_t *unknown_batch_cnt, const float *known,
                                  const int32_t *known_batch_cnt, float *dist2, int32_t *idx, void *stream) {
    PCD_ENTER();
    if (B <= 0 || N < 0) return PCD_ERR_INVALID_ARG;
    if (N == 0) return PCD_OK;
    if (!unknown || !unknown_batch_cnt || !known || !known_batch_cnt || !dist2 || !idx) return PCD_ERR_INVALID_ARG;
    three_nn_kernel<<<pcd_div_up(N, 4), 256, 0, (hipStream_t)stream>>>(B, N, unknown, unknown_batch_cnt, known,
                                                                      known_batch_cnt, dist2, idx);
    PN2_CHECK_LAUNCH();
}

extern "C" int pcd_three_interpolate_stack(int N, int C, const float *features, const int32_t *idx, const float *weight,
                                           float *out, void *stream) {
    PCD_ENTER();
    if (N < 0 || C <= 0) return PCD_ERR_INVALID_ARG;
    if (N == 0) return PCD_OK;
    if (!features || !idx || !weight || !out) return PCD_ERR_INVALID_ARG;
    const size_t n = (size_t)N * C;
    three_interpolate_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(N, C, features, idx, weight, out);
    PN2_CHECK_LAUNCH();
}

extern "C" int pcd_three_interpolate_stack_grad(int N, int C, const float *grad_out, const int32_t *idx,
                                                const float *weight, float *grad_features_zeroed, void *stream) {
    PCD_ENTER();
    if (N < 0 || C <= 0) return PCD_ERR_INVALID_ARG;
    if (N == 0) return PCD_OK;
    if (!grad_out || !idx || !weight || !grad_features_zeroed) return PCD_ERR_INVALID_ARG;
    const size_t n = (size_t)N * C;
    three_interpolate_grad_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(N, C, grad_out, idx, weight,
                                                                                              grad_features_zeroed);
    PN2_CHECK_LAUNCH();
}

extern "C" int pcd_voxel_query_stack(int M, int R1, int R2, int R3, int nsample, float radius, int z_range, int y_range,
                                     int x_range, const float *new_xyz, const float *xyz, const int32_t *new_coords,
                                     const int32_t *point_indices, int32_t *idx, void *stream) {
    PCD_ENTER();
    if (M < 0 || R1 <= 0 || R2 <= 0 || R3 <= 0 || nsample <= 0) return PCD_ERR_INVALID_ARG;
    if (M == 0) return PCD_OK;
    if (!new_xyz || !xyz || !new_coords || !point_indices || !idx) return PCD_ERR_INVALID_ARG;
    voxel_query_kernel<<<pcd_div_up(M, 256), 256, 0, (hipStream_t)stream>>>(M, R1, R2, R3, nsample, radius, z_range, y_range,
                                                                           x_range, new_xyz, xyz, new_coords,
                                                                           point_indices, idx);
    PN2_CHECK_LAUNCH();
}
