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
// BUCKET-PRUNED farthest point sampling for large frames (round 6) -- exact, one workgroup per frame, nothing device-scope
// inside the loop.  The sequential definition (sampling_gpu.cu:188-348) updates EVERY point's running distance against the new
// centre in each of the 4095 iterations; here the points of a frame are sorted once along a Z-curve (Morton code of a 128 x 128
// grid over the frame's x-y extent) and cut into BUCKETS of CH consecutive points (CH = 64 .. 256, <= FPS_NB buckets): equal
// point counts whatever the density (a LiDAR frame holds thousands of points per square metre near the sensor and a handful far
// out), compact in space.  Every bucket keeps the tight bounding box of its points and its current farthest point (running
// distance + the reference's tie key + coordinates).  A new centre c can only change a bucket whose box lies closer to c than
// the bucket's largest running distance:
//     dbox(c) >= maxd(bucket)  =>  d(p, c) >= dbox(c) >= maxd >= tmp[p]  for every p in it  =>  fminf(d, tmp[p]) = tmp[p].
// The inequality d(p, c) >= dbox(c) holds IN FLOAT ARITHMETIC: dbox is computed with the reference's own expression order
// ((ex ex + ey ey) + ez ez, no contraction) from per-axis gaps with |fl(b - c)| <= |fl(p - c)| -- rounding is monotone.  After
// the first few dozen samples a new centre touches the handful of buckets around it; the argmax is the reduction of the bucket
// maxima under the SAME total order as stack_fps_kernel (distance, bit-reversed thread id of the reference's tree, index), so
// the selected points are the reference's, ties included.  An iteration is one or two global round trips (the touched buckets'
// points, L2-resident) + LDS; the cooperative kernel below needs two device-scope hops (~5.4 us).
constexpr int FPS_NB = 2048;             // buckets per frame at most (two per thread of the 1024-thread workgroup)
constexpr int FPS_FINE = 128;            // Z-curve grid (FPS_FINE^2 = 16384 counters in LDS)
constexpr int FPS_CHMAX = 256;
constexpr int FPS_BT = 3;                // touched buckets a wave loads at a time

struct FpsBucketWs {            // per call: sorted point arrays over all frames + per-frame bucket boxes
    float4 *sp;                 // (x, y, z, running distance) of the sorted points: one 16-byte load / store per point
    int *sk;                    // original index inside the frame
    float *bbox;                // [B][6][FPS_NB]
    long long *stats;           // [B][8] profiling aid: touched buckets, rounds, clocks of the three phases (tools/exp_fps.py)
};

__host__ __device__ inline int fps_chunk(int n) {          // points per bucket: a multiple of 64, <= FPS_NB buckets
    int ch = (n + FPS_NB - 1) / FPS_NB;
    ch = (ch + 63) / 64 * 64;
    return ch < 64 ? 64 : ch;
}

__device__ __forceinline__ float block_minmax(float v, bool is_max, float *lds16) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off);
        v = is_max ? fmaxf(v, o) : fminf(v, o);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds16[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = lds16[0];
    for (int w = 1; w < 16; ++w) r = is_max ? fmaxf(r, lds16[w]) : fminf(r, lds16[w]);
    return r;
}

__device__ __forceinline__ unsigned fps_spread7(unsigned v) {      // 7 bits -> every other bit
    v = (v | (v << 4)) & 0x070fu;
    v = (v | (v << 2)) & 0x1333u;
    v = (v | (v << 1)) & 0x1555u;
    return v;
}

// binning: one 1024-thread workgroup per frame; dynamic LDS: FPS_FINE^2 counters
__global__ __launch_bounds__(1024) void fps_bin_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ xyz_cnt,
                                                       FpsBucketWs W) {
    extern __shared__ int fps_hist[];                     // [FPS_FINE * FPS_FINE]: counts, then cursors
    __shared__ float red[16];
    __shared__ int wsum[16];
    constexpr int NC = FPS_FINE * FPS_FINE, PER = NC / 1024;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int start = 0;
    for (int k = 0; k < b; ++k) start += xyz_cnt[k];
    const int n = xyz_cnt[b];
    const float *pts = xyz + (size_t)start * 3;
    float lox = 3.0e38f, loy = 3.0e38f, hix = -3.0e38f, hiy = -3.0e38f;
    for (int k = tid; k < n; k += 1024) {
        const float x = pts[(size_t)k * 3], y = pts[(size_t)k * 3 + 1];
        lox = fminf(lox, x); hix = fmaxf(hix, x);
        loy = fminf(loy, y); hiy = fmaxf(hiy, y);
    }
    lox = block_minmax(lox, false, red);
    hix = block_minmax(hix, true, red);
    loy = block_minmax(loy, false, red);
    hiy = block_minmax(hiy, true, red);
    const float ivx = hix > lox ? (float)FPS_FINE / (hix - lox) : 0.0f, ivy = hiy > loy ? (float)FPS_FINE / (hiy - loy) : 0.0f;
    auto cell = [&](float x, float y) {
        int cx = (int)((x - lox) * ivx), cy = (int)((y - loy) * ivy);
        cx = cx < 0 ? 0 : cx > FPS_FINE - 1 ? FPS_FINE - 1 : cx;
        cy = cy < 0 ? 0 : cy > FPS_FINE - 1 ? FPS_FINE - 1 : cy;
        return (int)(fps_spread7((unsigned)cx) | (fps_spread7((unsigned)cy) << 1));
    };
    for (int e = tid; e < NC; e += 1024) fps_hist[e] = 0;
    __syncthreads();
    for (int k = tid; k < n; k += 1024) atomicAdd(&fps_hist[cell(pts[(size_t)k * 3], pts[(size_t)k * 3 + 1])], 1);
    __syncthreads();
    // exclusive scan of the NC counters: PER consecutive counters per thread
    int loc[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        loc[i] = fps_hist[tid * PER + i];
        sum += loc[i];
    }
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int run = inc - sum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        fps_hist[tid * PER + i] = run;                    // cursor of the cell
        run += loc[i];
    }
    __syncthreads();
    // scatter (the order inside a cell does not matter: min-updates are per point, the argmax is a total order)
    for (int k = tid; k < n; k += 1024) {
        const float x = pts[(size_t)k * 3], y = pts[(size_t)k * 3 + 1], z = pts[(size_t)k * 3 + 2];
        const int pos = start + atomicAdd(&fps_hist[cell(x, y)], 1);
        W.sp[pos] = make_float4(x, y, z, 1e10f);
        W.sk[pos] = k;
    }
    __syncthreads();
    // tight boxes of the buckets (CH consecutive sorted points each): a wave per bucket
    const int ch = fps_chunk(n), nb = (n + ch - 1) / ch;
    float *bb = W.bbox + (size_t)b * 6 * FPS_NB;
    for (int q = wave; q < nb; q += 16) {
        const int s0 = start + q * ch, cn = min(ch, n - q * ch);
        float mnx = 3.0e38f, mny = 3.0e38f, mnz = 3.0e38f, mxx = -3.0e38f, mxy = -3.0e38f, mxz = -3.0e38f;
        for (int p = lane; p < cn; p += 64) {
            const float4 pt = W.sp[s0 + p];
            const float x = pt.x, y = pt.y, z = pt.z;
            mnx = fminf(mnx, x); mxx = fmaxf(mxx, x);
            mny = fminf(mny, y); mxy = fmaxf(mxy, y);
            mnz = fminf(mnz, z); mxz = fmaxf(mxz, z);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mnx = fminf(mnx, __shfl_xor(mnx, off)); mxx = fmaxf(mxx, __shfl_xor(mxx, off));
            mny = fminf(mny, __shfl_xor(mny, off)); mxy = fmaxf(mxy, __shfl_xor(mxy, off));
            mnz = fminf(mnz, __shfl_xor(mnz, off)); mxz = fmaxf(mxz, __shfl_xor(mxz, off));
        }
        if (lane == 0) {
            bb[0 * FPS_NB + q] = mnx; bb[1 * FPS_NB + q] = mny; bb[2 * FPS_NB + q] = mnz;
            bb[3 * FPS_NB + q] = mxx; bb[4 * FPS_NB + q] = mxy; bb[5 * FPS_NB + q] = mxz;
        }
    }
}

// The argmax key of a point in two words, compared lexicographically (larger = better): the running distance itself (a float
// >= 0; -1 = "no point"), then a 32-bit tie word -- the reference tree's rule (smaller bit-reversed thread id k & 1023 first,
// then the smaller index): fps_better's total order.
__device__ __forceinline__ unsigned fps_tie(int k) {
    const unsigned br = __builtin_bitreverse32((unsigned)(k & 1023)) >> 22;
    return ((1023u - br) << 22) | (0x3fffffu - (unsigned)k);
}
__device__ __forceinline__ int fps_tie_index(unsigned t) { return (int)(0x3fffffu - (t & 0x3fffffu)); }

// wave-wide maxima without the LDS crossbar: four DPP steps inside the rows of 16 lanes (one v_max with a DPP operand each), the
// four row maxima through scalar registers.  (A __shfl_xor of five values per step cost this kernel 5 k clocks per sample.)
#define FPS_DPP_ALL(OP, V)                                                                       \
    V = OP(V, __builtin_bit_cast(decltype(V), __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), 0xB1, 0xf, 0xf, false)));  \
    V = OP(V, __builtin_bit_cast(decltype(V), __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), 0x4E, 0xf, 0xf, false)));  \
    V = OP(V, __builtin_bit_cast(decltype(V), __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), 0x141, 0xf, 0xf, false))); \
    V = OP(V, __builtin_bit_cast(decltype(V), __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), 0x140, 0xf, 0xf, false)));
__device__ __forceinline__ float fps_wave_max_f32(float v) {
    FPS_DPP_ALL(fmaxf, v)
    float r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
#pragma unroll
    for (int row = 1; row < 4; ++row) r = fmaxf(r, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), row * 16)));
    return r;
}
__device__ __forceinline__ unsigned fps_umax(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned fps_wave_max_u32(unsigned v) {
    FPS_DPP_ALL(fps_umax, v)
    unsigned r = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
#pragma unroll
    for (int row = 1; row < 4; ++row) r = fps_umax(r, (unsigned)__builtin_amdgcn_readlane((int)v, row * 16));
    return r;
}
// true on the ONE lane that holds the wave's best (d, tie); dmax / tmax: that key (wave-uniform)
__device__ __forceinline__ bool fps_wave_best(float d, unsigned tie, float &dmax, unsigned &tmax) {
    dmax = fps_wave_max_f32(d);
    tmax = fps_wave_max_u32(d == dmax ? tie : 0u);
    return d == dmax && tie == tmax && dmax >= 0.0f;
}

// LDS per bucket: box minimum + largest running distance (float4), box maximum + the tie word of its farthest point (float4),
// that point (float4)
constexpr int FPS_LDS_BYTES = FPS_NB * (16 + 16 + 16);

struct FpsWaveBest {
    float d;
    unsigned tie;
    float x, y, z;
    float pad[3];
};

// Iteration = ONE workgroup barrier.  Bucket q belongs to wave q % 16 (consecutive buckets are neighbours on the Z-curve: the
// dozen buckets a centre touches spread over the waves), lane (q / 16) % 64, slot q / 1024: a wave tests its own 128 buckets,
// updates the touched ones itself (nobody else reads or writes them), reduces their keys and publishes its best; after the
// barrier 16 lanes of every wave reduce the 16 wave results (double-buffered by the parity of the sample).
__global__ __launch_bounds__(1024) void fps_bucket_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ xyz_cnt,
                                                          int32_t *__restrict__ idxs, const int32_t *__restrict__ num_sampled,
                                                          FpsBucketWs W, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char fpsb_lds[];
    float4 *bmin = (float4 *)fpsb_lds;                                 // (mnx, mny, mnz, maxd)
    float4 *bmax = bmin + FPS_NB;                                      // (mxx, mxy, mxz, tie word of the farthest point)
    float4 *bpt = bmax + FPS_NB;                                       // the farthest point of the bucket
    __shared__ FpsWaveBest wbest[2][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int start = 0, ostart = 0;
    for (int k = 0; k < b; ++k) {
        start += xyz_cnt[k];
        ostart += num_sampled[k];
    }
    const int n = xyz_cnt[b], m = num_sampled[b];
    int32_t *out = idxs + ostart;
    if (m <= 0 || n <= 0) return;
    const int ch = fps_chunk(n), nb = (n + ch - 1) / ch;
    const float *bb = W.bbox + (size_t)b * 6 * FPS_NB;
    float4 *__restrict__ sp = W.sp + start;
    const int *__restrict__ sk = W.sk + start;
    const int nu = ch / 64;                                            // 64-point pieces of a bucket (1 .. FPS_CHMAX / 64)
    const int myq[2] = {lane * 16 + wave, (lane + 64) * 16 + wave};    // this thread's buckets
    // LDS slot of bucket q: the 128 buckets of a wave side by side, a lane's two 64 apart (bucket order itself would put the
    // float4s of a wave's lanes 256 bytes apart: all on the same four banks)
    auto slot = [](int q) { return (q & 15) * 128 + (q >> 4); };
    const int mys[2] = {wave * 128 + lane, wave * 128 + lane + 64};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = myq[h];
        const bool live = q < nb;
        bmin[mys[h]] = make_float4(live ? bb[0 * FPS_NB + q] : 0.0f, live ? bb[1 * FPS_NB + q] : 0.0f, live ? bb[2 * FPS_NB + q] : 0.0f,
                                   live ? 1e10f : -1.0f);                // (a bucket without points never gets touched, never wins)
        bmax[mys[h]] = make_float4(live ? bb[3 * FPS_NB + q] : 0.0f, live ? bb[4 * FPS_NB + q] : 0.0f, live ? bb[5 * FPS_NB + q] : 0.0f, 0.0f);
        bpt[mys[h]] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (tid == 0) out[0] = start;
    float x1 = xyz[(size_t)start * 3], y1 = xyz[(size_t)start * 3 + 1], z1 = xyz[(size_t)start * 3 + 2];
    long long st_dirty = 0, st_rounds = 0, st_t1 = 0, st_t2 = 0, st_t3 = 0;
    for (int j = 1; j < m; ++j) {
        const long long c0 = __builtin_readcyclecounter();
        // (1) which of this wave's buckets can change?  gap of the box to the centre per axis (max3: the positive one of the two
        //     differences, or 0 inside), then the reference's expression order
        bool dt[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 lo = bmin[mys[h]], hi = bmax[mys[h]];
            const float ex = fmaxf(fmaxf(lo.x - x1, x1 - hi.x), 0.0f);
            const float ey = fmaxf(fmaxf(lo.y - y1, y1 - hi.y), 0.0f);
            const float ez = fmaxf(fmaxf(lo.z - z1, z1 - hi.z), 0.0f);
            const float dbox = ex * ex + ey * ey + ez * ez;
            dt[h] = dbox < lo.w;                                  // (lo.w = -1 for a bucket without points)
        }
        unsigned long long todo[2] = {__ballot(dt[0]), __ballot(dt[1])};
        const long long c1 = __builtin_readcyclecounter();
        st_dirty += __popcll(todo[0]) + __popcll(todo[1]);
        st_rounds += (todo[0] | todo[1]) != 0ull;
        // (2) the touched buckets of this wave, FPS_BT at a time: ALL their loads are issued before the first use
        while (todo[0] | todo[1]) {
            int qs[FPS_BT];
#pragma unroll
            for (int i = 0; i < FPS_BT; ++i) {
                const int h = todo[0] ? 0 : 1;
                if (todo[h]) {
                    const int l = __builtin_ctzll(todo[h]);
                    todo[h] &= todo[h] - 1ull;
                    qs[i] = (l + 64 * h) * 16 + wave;
                } else {
                    qs[i] = -1;
                }
            }
            constexpr int U = FPS_CHMAX / 64;
            float4 pp[FPS_BT][U];
            int pk[FPS_BT][U];
#pragma unroll
            for (int i = 0; i < FPS_BT; ++i) {
                const int s0 = qs[i] * ch, cn = qs[i] >= 0 ? min(ch, n - s0) : 0;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    pp[i][u] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
                    pk[i][u] = 0;
                    if (u < nu && qs[i] >= 0) {                 // (wave-uniform: no instruction for pieces the bucket does not have)
                        const int p = lane + 64 * u;
                        if (p < cn && !(dbg & 1)) {
                            pp[i][u] = sp[s0 + p];
                            pk[i][u] = sk[s0 + p];
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < FPS_BT; ++i) {
                if (qs[i] < 0) continue;                        // (wave-uniform)
                const int q = qs[i], s0 = q * ch, cn = min(ch, n - s0);
                float bd = -1.0f, bxv = 0.0f, byv = 0.0f, bzv = 0.0f;
                unsigned bt = 0u;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = lane + 64 * u;
                    if (u < nu && p < cn) {
                        const float px = pp[i][u].x, py = pp[i][u].y, pz = pp[i][u].z;
                        const float d = (px - x1) * (px - x1) + (py - y1) * (py - y1) + (pz - z1) * (pz - z1);
                        const float d2 = fminf(d, pp[i][u].w);
                        if (d2 != pp[i][u].w && !(dbg & 2)) sp[s0 + p].w = d2;
                        const unsigned t = fps_tie(pk[i][u]);
                        if (d2 > bd || (d2 == bd && t > bt)) {
                            bd = d2; bt = t; bxv = px; byv = py; bzv = pz;
                        }
                    }
                }
                float dmax;
                unsigned tmax;
                if (!(dbg & 4) && fps_wave_best(bd, bt, dmax, tmax)) {          // the one lane that holds the bucket's farthest point
                    bmin[slot(q)].w = dmax;
                    bmax[slot(q)].w = __uint_as_float(tmax);
                    bpt[slot(q)] = make_float4(bxv, byv, bzv, 0.0f);
                }
            }
        }
        const long long c2 = __builtin_readcyclecounter();
        // (3) this wave's best bucket (its own LDS writes above are ordered before these reads: same wave, in-order LDS)
        __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): the LDS stores of the owner lanes have landed
        const float d0 = bmin[mys[0]].w, d1 = bmin[mys[1]].w;
        const unsigned t0 = __float_as_uint(bmax[mys[0]].w), t1 = __float_as_uint(bmax[mys[1]].w);
        const bool second = d1 > d0 || (d1 == d0 && t1 > t0);
        const float md = second ? d1 : d0;
        const unsigned mt = second ? t1 : t0;
        float dmax;
        unsigned tmax;
        if (fps_wave_best(md, mt, dmax, tmax)) {
            const float4 pt = bpt[second ? mys[1] : mys[0]];
            FpsWaveBest e;
            e.d = dmax; e.tie = tmax; e.x = pt.x; e.y = pt.y; e.z = pt.z; e.pad[0] = e.pad[1] = e.pad[2] = 0.0f;
            wbest[j & 1][wave] = e;
        } else if (dmax < 0.0f && lane == 0) {
            wbest[j & 1][wave].d = -1.0f;
            wbest[j & 1][wave].tie = 0u;
        }
        __syncthreads();
        // the best of the 16 waves: lanes 0..15 of every wave take one entry each, reduce inside their row of 16
        {
            const FpsWaveBest e = wbest[j & 1][lane & 15];
            float gd = e.d;
            FPS_DPP_ALL(fmaxf, gd)
            unsigned gt = e.d == gd ? e.tie : 0u;
            FPS_DPP_ALL(fps_umax, gt)
            const bool win = e.d == gd && e.tie == gt;                      // one lane of every row of 16
            const unsigned long long wm = __ballot(win);
            const int wl = __builtin_ctzll(wm);                             // (lanes 0..15 hold the same 16 entries in every row)
            x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e.x), wl));
            y1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e.y), wl));
            z1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e.z), wl));
            if (tid == 0) out[j] = fps_tie_index((unsigned)__builtin_amdgcn_readlane((int)e.tie, wl)) + start;
        }
        const long long c3 = __builtin_readcyclecounter();
        st_t1 += c1 - c0; st_t2 += c2 - c1; st_t3 += c3 - c2;
    }
    if (tid == 0 && W.stats) {
        long long *st = W.stats + (size_t)b * 8;
        st[0] = st_dirty; st[1] = st_rounds; st[2] = st_t1; st[3] = st_t2; st[4] = st_t3; st[5] = nb; st[6] = ch; st[7] = m;
    }
}
#undef FPS_DPP_ALL

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

static void fps_bucket_carve(void *workspace, int B, size_t total, FpsBucketWs &W, size_t *need) {
    char *w = (char *)workspace;
    size_t off = 0;
    auto take = [&](size_t count, size_t elem) {
        char *p = w ? w + off : nullptr;
        off += ws_piece(count, elem);
        return p;
    };
    W.sp = (float4 *)take(total, 16);
    W.sk = (int *)take(total, 4);
    W.bbox = (float *)take((size_t)B * 6 * FPS_NB, 4);
    W.stats = (long long *)take((size_t)B * 8, 8);
    *need = off;
}

extern "C" size_t pcd_stack_fps_buckets_workspace_bytes(int B, int total_points) {
    if (B <= 0 || total_points < 0) return 0;
    FpsBucketWs W;
    size_t need = 0;
    fps_bucket_carve(nullptr, B, (size_t)total_points, W, &need);
    return need;
}

// Bucket-pruned exact farthest point sampling (see fps_bucket_kernel): same selected points as
// pcd_stack_farthest_point_sampling, one workgroup per frame, no inter-workgroup traffic.  total_points = sum of xyz_batch_cnt
// (the host knows it: the rows of xyz).
extern "C" int pcd_stack_farthest_point_sampling_buckets(int B, const float *xyz, const int32_t *xyz_batch_cnt, int32_t *idxs,
                                                         const int32_t *num_sampled_points, int total_points, int max_cnt_host,
                                                         void *workspace, size_t workspace_bytes, void *stream) {
    PCD_ENTER();
    if (B <= 0 || !xyz || !xyz_batch_cnt || !idxs || !num_sampled_points || total_points < 0) return PCD_ERR_INVALID_ARG;
    FpsBucketWs W;
    size_t need = 0;
    fps_bucket_carve(workspace, B, (size_t)total_points, W, &need);
    if (!workspace || workspace_bytes < need) return PCD_ERR_WORKSPACE;
    // a frame of more than FPS_NB x FPS_CHMAX points does not fit the bucket tables: the caller takes another form
    if (max_cnt_host <= 0 || max_cnt_host > FPS_NB * FPS_CHMAX) return PCD_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int bin_lds = FPS_FINE * FPS_FINE * (int)sizeof(int);
    if (hipFuncSetAttribute((const void *)fps_bin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bin_lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)fps_bucket_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FPS_LDS_BYTES) != hipSuccess)
        return PCD_ERR_LAUNCH;
    fps_bin_kernel<<<B, 1024, bin_lds, st>>>(xyz, xyz_batch_cnt, W);
    const int dbg = pcd_opt(PCD_OPT_FPS_G) >= 1000 ? pcd_opt(PCD_OPT_FPS_G) - 1000 : 0;     // (ablation bits of tools/exp_fps.py; 0 in production)
    fps_bucket_kernel<<<B, 1024, FPS_LDS_BYTES, st>>>(xyz, xyz_batch_cnt, idxs, num_sampled_points, W, dbg);
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
