// Dense 3x3 convolution (stride 1, padding 1) over channels-last bf16 maps on gfx950: the BaseBEVBackbone / CenterHead
// convs behind the sparse hot path (pcdet/models/backbones_2d/base_bev_backbone.py:30-112, dense_heads/center_head.py:
// 11-46 -- nn.Conv2d(k=3, padding=1) through MIOpen in the reference).  Implicit GEMM on v_mfma_f32_16x16x32_bf16:
//   D[cout][pixel] += W_tap[cout][cin] * X[pixel + tap][cin]      (A operand = weights, B operand = input pixels)
// Workgroup = 16 x 16 output pixels x 64 output channels, 4 waves x (4 pixel rows x 4 cout blocks) = 16 accumulators
// each.  Per 32-channel chunk of the input the workgroup stages the 18 x 18 halo tile (20.7 KB: every input pixel is
// read from HBM ONCE and used by 9 taps x 64 channels) and that chunk's weights of all 9 taps (36 KB) in LDS, double
// buffered: global loads of chunk c + 1 are in flight while chunk c is multiplied.  Both operand reads are lane-linear
// 16-byte ds_reads (a B fragment = 16 consecutive pixels x 64 B = 1 KiB contiguous), conflict free.  Weights are
// packed once per step into fragment order with the channel interleave of the sparse kernels (a lane ends up with 16
// CONSECUTIVE output channels of its pixel: two 16-byte stores).  The data gradient is the same kernel on weights
// packed transposed and rotated (mode 1).
#include "common.h"
#include "bn_mid.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int TP = 16;                 // tile: TP x TP output pixels
constexpr int HT = TP + 2;             // halo tile side
constexpr int IN_BYTES = HT * HT * 64; // one 32-channel chunk of the halo tile
constexpr int W_BYTES = 9 * 4 * 64 * 16;   // one chunk's weights: [tap][cout block][lane][16 B]

// channel of row i (0..15) of cout block mb (0..3) inside a group of 64: lane group g = i / 4 owns channels
// [16 g, 16 g + 16) across the 4 blocks
__host__ __device__ inline int chan_of(int mb, int i) { return (i >> 2) * 16 + mb * 4 + (i & 3); }

// weight [cout][cin][3][3] f32 (OIHW) -> packed [cout_pad / 64][cin / 32][tap][mb][lane][8] bf16; output channels
// in [cout, cout_pad) are zero (small heads run padded).
// mode 1 (data gradient): the roles of cin / cout swap and the taps rotate by 180 degrees
__device__ __forceinline__ void conv2d_pack_piece(const float *__restrict__ w, int cin, int cout, int cout_pad, int mode,
                                                  unsigned short *__restrict__ packed, size_t e) {
    const int n_out = mode == 0 ? cout : cin, n_in = mode == 0 ? cin : cout;
    const int ncc = (mode == 0 ? cin : cout_pad) / 32;
    const int lane = (int)(e & 63);
    size_t q = e >> 6;
    const int mb = (int)(q & 3); q >>= 2;
    const int tap = (int)(q % 9); q /= 9;
    const int cc = (int)(q % ncc);
    const int cg = (int)(q / ncc);
    const int o = cg * 64 + chan_of(mb, lane & 15);
    const int k0 = cc * 32 + (lane >> 4) * 8;
    unsigned short v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = k0 + j;
        float f = 0.0f;
        if (o < n_out && c < n_in)
            f = mode == 0 ? w[((size_t)o * cin + c) * 9 + tap] : w[((size_t)c * cin + o) * 9 + (8 - tap)];
        v[j] = f32_to_bf16_bits(f);
    }
    uint4 out;
    out.x = v[0] | ((u32)v[1] << 16); out.y = v[2] | ((u32)v[3] << 16);
    out.z = v[4] | ((u32)v[5] << 16); out.w = v[6] | ((u32)v[7] << 16);
    reinterpret_cast<uint4 *>(packed)[e] = out;
}

__global__ __launch_bounds__(256) void conv2d_pack_kernel(const float *__restrict__ w, int cin, int cout, int cout_pad,
                                                          int mode, unsigned short *__restrict__ packed, size_t total) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;   // one 16-byte lane piece per thread
    if (e < total) conv2d_pack_piece(w, cin, cout, cout_pad, mode, packed, e);
}

__device__ __forceinline__ void planes_pack_dispatch(const float *w, int cin, int cout, int mode, unsigned short *packed,
                                                     size_t e);   // (modes 2..7, below)

// all packs of a model in ONE launch: table rows {weight, packed, cin, cout, cout_pad, mode, first block, pieces}
__global__ __launch_bounds__(256) void conv2d_pack_batched_kernel(const long long *__restrict__ table, int n) {
    int j = 0;
    for (int q = 1; q < n; ++q)
        if (table[(size_t)q * 8 + 6] <= (long long)blockIdx.x) j = q;
    const long long *row = table + (size_t)j * 8;
    const size_t e = (size_t)(blockIdx.x - row[6]) * 256 + threadIdx.x;
    if (e >= (size_t)row[7]) return;
    if (row[5] >= 2)
        planes_pack_dispatch((const float *)row[0], (int)row[2], (int)row[3], (int)row[5], (unsigned short *)row[1], e);
    else
        conv2d_pack_piece((const float *)row[0], (int)row[2], (int)row[3], (int)row[4], (int)row[5],
                          (unsigned short *)row[1], e);
}


// BatchNorm sums in the conv's epilogue (PcdBnReduce, see spconv.hip: the same two modes for the dense maps).  A
// workgroup covers 256 pixels x 64 channels: partial row `tile` (= blockIdx.x) has 2 x c_total columns, of which this
// workgroup writes [cg * 64, cg * 64 + 64) of both halves; with `mid` the workgroup that completes a group of rows (every
// row needs all gridDim.y channel groups: expect = rows of the group x gridDim.y arrivals) folds it.
struct DenseBn {
    int mode, relu;
    const unsigned short *x, *y;     // mode 2: BatchNorm input / output (= this conv's forward input), [pixels][c_total]
    const float *mean, *invstd;
    float *partial;
    double *mid;
    int *counters;
};

// sums s[j] / q[j] of the lane's 16 channels (c0 + j) over its pixels -> partial row; all threads of the workgroup
__device__ __forceinline__ void dense_bn_publish(const DenseBn &bn, float (&sv)[16], float (&qv)[16], int c_total, int cg,
                                                 int g, int m, int wave, char *smem) {
    float *red = reinterpret_cast<float *>(smem);          // [4 waves][2][64]; the tile buffers are free now
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        sv[j] = row16_sum(sv[j]);
        qv[j] = row16_sum(qv[j]);
    }
    __syncthreads();
    if (m == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            red[(wave * 2 + 0) * 64 + g * 16 + j] = sv[j];
            red[(wave * 2 + 1) * 64 + g * 16 + j] = qv[j];
        }
    }
    __syncthreads();
    const int tile = blockIdx.x, nrows = gridDim.x;
    float *dst = bn.partial + (size_t)tile * 2 * c_total;
    if (threadIdx.x < 128) {
        const int half = threadIdx.x >> 6, ch = threadIdx.x & 63;
        const float v = ((red[(0 * 2 + half) * 64 + ch] + red[(1 * 2 + half) * 64 + ch]) + red[(2 * 2 + half) * 64 + ch]) +
                        red[(3 * 2 + half) * 64 + ch];
        if (cg * 64 + ch < c_total) {
            float *d = dst + half * c_total + cg * 64 + ch;
            if (bn.mid) __hip_atomic_store(d, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *d = v;
        }
    }
    if (!bn.mid) return;
    __shared__ int last_s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the row piece has reached the coherence point ...
    __syncthreads();
    const int r = tile & (BN_MID_ROWS - 1);
    if (tile == 0 && cg == 0)                              // fewer tiles than groups: the empty groups' rows are zero
        for (int e = nrows * 2 * c_total + threadIdx.x; e < BN_MID_ROWS * 2 * c_total; e += blockDim.x) bn.mid[e] = 0.0;
    if (threadIdx.x == 0) {                                // ... before this workgroup counts as arrived
        int *cnt = bn.counters + r * BN_COUNTER_STRIDE;
        const int expect = ((nrows - r + BN_MID_ROWS - 1) / BN_MID_ROWS) * (int)gridDim.y;
        const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = old == expect - 1;
        if (last_s) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (last_s) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        bn_mid_row(bn.partial, nrows, c_total, r, bn.mid, reinterpret_cast<double *>(smem));
    }
}

template <int WB>   // weight stage buffers: 2 = double buffered (113 KB LDS, 1 workgroup / CU); 1 = single (77 KB, 2 / CU)
__global__ __launch_bounds__(256, WB == 2 ? 1 : 2) void conv2d_3x3_kernel(const unsigned short *__restrict__ x, int B, int H, int W,
                                                             int cin, const uint4 *__restrict__ wp, int cout,
                                                             const float *__restrict__ bias,
                                                             unsigned short *__restrict__ y, unsigned x_bytes,
                                                             unsigned w_bytes, int x_cs, int y_cs, DenseBn bn) {
    // x_cs / y_cs: channels per pixel of the buffers x / y live in (>= cin / cout: a channel block of a wider map)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *in_s = smem;                       // [2][IN_BYTES]
    char *w_s = smem + 2 * IN_BYTES;         // [WB][W_BYTES]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = lane & 15, g = lane >> 4;
    const int tiles_x = (W + TP - 1) / TP, tiles_y = (H + TP - 1) / TP;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int cg = blockIdx.y;
    const int x0 = tx * TP, y0 = ty * TP;
    const int ncc = cin / 32;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, (int)w_bytes, 0x00020000);

    // staging assignment: halo tile = 324 pixels x 4 pieces = 1296 pieces (6 per thread, last partial);
    // weights = 2304 pieces (9 per thread)
    unsigned in_off[6];
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int p = it * 256 + threadIdx.x;
        const int pix = p >> 2, piece = p & 3;
        const int py = pix / HT, px = pix - py * HT;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        const bool ok = p < HT * HT * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        in_off[it] = ok ? (unsigned)((((size_t)b * H + gy) * W + gx) * x_cs * 2 + piece * 16) : 0xFFFFFFF0u;
    }
    u32x4 in_r[6], w_r[9];
    auto load_chunk = [&](int cc) {
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const unsigned off = in_off[it] == 0xFFFFFFF0u ? 0xFFFFFFF0u : in_off[it] + (unsigned)cc * 64u;
            in_r[it] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
        }
        const unsigned wbase = (unsigned)(((size_t)cg * ncc + cc) * W_BYTES);
#pragma unroll
        for (int it = 0; it < 9; ++it)
            w_r[it] = __builtin_amdgcn_raw_buffer_load_b128(wrs, wbase + (unsigned)(it * 256 + threadIdx.x) * 16u, 0, 0);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int p = it * 256 + threadIdx.x;
            if (p < HT * HT * 4) *reinterpret_cast<u32x4 *>(in_s + buf * IN_BYTES + p * 16) = in_r[it];
        }
#pragma unroll
        for (int it = 0; it < 9; ++it)
            *reinterpret_cast<u32x4 *>(w_s + (WB == 2 ? buf : 0) * W_BYTES + (it * 256 + threadIdx.x) * 16) = w_r[it];
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) acc[mb][pb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int cc = 0; cc < ncc; ++cc) {
        const int buf = cc & 1;
        if (cc + 1 < ncc) load_chunk(cc + 1);
        const char *ins = in_s + buf * IN_BYTES, *ws = w_s + (WB == 2 ? buf : 0) * W_BYTES;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                af[mb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(ws + ((tap * 4 + mb) * 64 + lane) * 16));
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int r = wave * 4 + pb;
                bf[pb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(
                                                        ins + ((r + dy) * HT + (m + dx)) * 64 + g * 16));
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int pb = 0; pb < 4; ++pb)
                    acc[mb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mb], bf[pb], acc[mb][pb], 0, 0, 0);
        }
        if (WB == 1) __syncthreads();                      // single weight stage: everyone is done reading it
        if (cc + 1 < ncc) store_chunk(buf ^ 1);            // (read in the previous iteration, before its barrier)
        __syncthreads();
    }
    // lane (pixel m, group g) holds channels cg*64 + g*16 + mb*4 + r of the pixels (row wave*4 + pb, column m)
    const int c0 = cg * 64 + g * 16;
    float bv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bv[j] = (bias && c0 + j < cout) ? bias[c0 + j] : 0.0f;
    float sv[16], qv[16], mu[16], is[16];
    if (bn.mode) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            sv[j] = qv[j] = 0.0f;
            mu[j] = (bn.mode == 2 && c0 + j < cout) ? bn.mean[c0 + j] : 0.0f;
            is[j] = (bn.mode == 2 && c0 + j < cout) ? bn.invstd[c0 + j] : 0.0f;
        }
    }
    // (with BatchNorm sums: all four rows are rounded and summed FIRST, the partial row is published, and only then the
    //  tile is stored -- the publication waits for its own stores (s_waitcnt vmcnt(0)), which must not include the 32 KB
    //  of output stores: that cost 8 us per launch)
    u32 o[4][8];
    bool live[4];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
        const int gy = y0 + wave * 4 + pb, gx = x0 + m;
        live[pb] = !(gy >= H || gx >= W || c0 >= cout);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const u32 lo0 = f32_to_bf16_bits(acc[mb][pb][0] + bv[mb * 4 + 0]);
            const u32 hi0 = f32_to_bf16_bits(acc[mb][pb][1] + bv[mb * 4 + 1]);
            const u32 lo1 = f32_to_bf16_bits(acc[mb][pb][2] + bv[mb * 4 + 2]);
            const u32 hi1 = f32_to_bf16_bits(acc[mb][pb][3] + bv[mb * 4 + 3]);
            o[pb][mb * 2] = lo0 | (hi0 << 16);
            o[pb][mb * 2 + 1] = lo1 | (hi1 << 16);
        }
        if (!live[pb] || !bn.mode) continue;
        const size_t pix = ((size_t)b * H + gy) * W + gx;
        if (bn.mode == 1) {            // statistics of the outputs AS STORED (bf16)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float v = __uint_as_float((j & 1) ? (o[pb][j >> 1] & 0xffff0000u) : (o[pb][j >> 1] << 16));
                sv[j] += v;
                qv[j] += v * v;
            }
        } else {                       // this output is dy of a BatchNorm(+ReLU): sum dz, sum dz * xhat
            const uint4 *xp = reinterpret_cast<const uint4 *>(bn.x + pix * cout + c0);
            const uint4 x0v = xp[0], x1v = xp[1];
            const u32 xw[8] = {x0v.x, x0v.y, x0v.z, x0v.w, x1v.x, x1v.y, x1v.z, x1v.w};
            u32 yw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (bn.relu) {
                const uint4 *yp = reinterpret_cast<const uint4 *>(bn.y + pix * cout + c0);
                const uint4 y0v = yp[0], y1v = yp[1];
                yw[0] = y0v.x; yw[1] = y0v.y; yw[2] = y0v.z; yw[3] = y0v.w;
                yw[4] = y1v.x; yw[5] = y1v.y; yw[6] = y1v.z; yw[7] = y1v.w;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float v = __uint_as_float((j & 1) ? (o[pb][j >> 1] & 0xffff0000u) : (o[pb][j >> 1] << 16));
                const float xv = __uint_as_float((j & 1) ? (xw[j >> 1] & 0xffff0000u) : (xw[j >> 1] << 16));
                const float yv = __uint_as_float((j & 1) ? (yw[j >> 1] & 0xffff0000u) : (yw[j >> 1] << 16));
                const float dz = (bn.relu && !(yv > 0.0f)) ? 0.0f : v;
                sv[j] += dz;
                qv[j] += dz * (xv - mu[j]) * is[j];
            }
        }
    }
    if (bn.mode) dense_bn_publish(bn, sv, qv, cout, cg, g, m, wave, smem);
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
        if (!live[pb]) continue;
        const int gy = y0 + wave * 4 + pb, gx = x0 + m;
        uint4 *dst = reinterpret_cast<uint4 *>(y + (((size_t)b * H + gy) * W + gx) * y_cs + c0);
        dst[0] = make_uint4(o[pb][0], o[pb][1], o[pb][2], o[pb][3]);
        dst[1] = make_uint4(o[pb][4], o[pb][5], o[pb][6], o[pb][7]);
    }
}


// ---------------------------------------------------------------------------------------------
// The three layers of BaseBEVBackbone the 3x3 kernel does not cover (base_bev_backbone.py:36-75): the stride-2 3x3 conv
// that opens block 2 and the ConvTranspose2d(k = stride) of the two deblocks, forward and data gradient, on the same
// tile machinery.  A stride-2 map splits into 4 PARITY PLANES (y & 1, x & 1); per plane each of these operators is a
// stride-1 stencil with 1, 2 or 4 taps on the half-resolution ("coarse") grid:
//   GATHER  (coarse OUTPUT): y[oy][ox] += W[tap] x[2 (oy + d) + py][2 (ox + d') + px]  over all planes and their taps
//           = Conv2d(3, stride 2, pad 1) forward, and the data gradient of ConvTranspose2d(2, stride 2)
//   SCATTER (coarse INPUT) : y[2 cy + py][2 cx + px] = sum over the plane's taps W[tap] x[cy + d][cx + d']
//           = data gradient of that Conv2d, forward of that ConvTranspose2d; blockIdx.z = output plane
//   K1: ConvTranspose2d(1, stride 1) = a 1x1 conv (one plane, one tap), both directions through SCATTER.
// Workgroup = 16 x 16 coarse pixels x 64 output channels as above; per (plane, 32-channel chunk) stage the 18 x 18 tile
// of that plane (double buffered) and the plane's 1-4 weight taps (single buffered): 57 KB of LDS, two workgroups / CU.
enum { K_C3S2 = 0, K_K2S2 = 1, K_K1 = 2 };

template <int KIND> __host__ __device__ constexpr int pl_count() { return KIND == K_K1 ? 1 : 4; }
template <int KIND> __host__ __device__ constexpr int pl_step() { return KIND == K_K1 ? 1 : 2; }
template <int KIND> __host__ __device__ constexpr int ax_n(int par) { return (KIND == K_C3S2 && par) ? 2 : 1; }
// kernel index / coarse offset of tap t of an axis with parity par
template <int KIND> __host__ __device__ constexpr int ax_k(int par, int t) {
    return KIND == K_C3S2 ? (par ? (t ? 2 : 0) : 1) : (KIND == K_K2S2 ? par : 0);
}
template <int KIND, bool GATHER> __host__ __device__ constexpr int ax_d(int par, int t) {
    return (KIND == K_C3S2 && par && t == 0) ? (GATHER ? -1 : 1) : 0;
}
template <int KIND> __host__ __device__ constexpr int pl_taps(int plane) {
    return ax_n<KIND>(plane >> 1) * ax_n<KIND>(plane & 1);
}
template <int KIND> __host__ __device__ constexpr int pl_base(int plane) {
    int s = 0;
    for (int p = 0; p < plane; ++p) s += pl_taps<KIND>(p);
    return s;
}
template <int KIND> __host__ __device__ constexpr int taps_total() { return pl_base<KIND>(pl_count<KIND>()); }
template <int KIND> __host__ __device__ constexpr int taps_max() { return KIND == K_C3S2 ? 4 : 1; }
template <int KIND> __host__ __device__ constexpr int kernel_w() { return KIND == K_C3S2 ? 3 : (KIND == K_K2S2 ? 2 : 1); }

// packs of the plane kernels: [n_out_pad / 64][n_in / 32][tap, plane-major][mb][lane][8] bf16.
// Element (o, c, ky, kx) of the operator sits at w[o * so + c * sc + ky * KW + kx] (so / sc: see pcd_conv2d_pack_* modes)
template <int KIND>
__device__ __forceinline__ void planes_pack_piece(const float *__restrict__ w, int n_out, int n_in, long long so, long long sc,
                                                  unsigned short *__restrict__ packed, size_t e) {
    constexpr int NT = taps_total<KIND>(), KW = kernel_w<KIND>();
    const int ncc = n_in / 32;
    const int lane = (int)(e & 63);
    size_t q = e >> 6;
    const int mb = (int)(q & 3); q >>= 2;
    const int tap = (int)(q % NT); q /= NT;
    const int cc = (int)(q % ncc);
    const int cg = (int)(q / ncc);
    int plane = 0;
#pragma unroll
    for (int p = 1; p < pl_count<KIND>(); ++p)
        if (tap >= pl_base<KIND>(p)) plane = p;
    const int t = tap - pl_base<KIND>(plane);
    const int py = plane >> 1, px = plane & 1;
    const int nx = ax_n<KIND>(px);
    const int ky = ax_k<KIND>(py, t / nx), kx = ax_k<KIND>(px, t % nx);
    const int o = cg * 64 + chan_of(mb, lane & 15);
    const int k0 = cc * 32 + (lane >> 4) * 8;
    unsigned short v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = k0 + j;
        const float f = (o < n_out && c < n_in) ? w[(size_t)o * so + (size_t)c * sc + ky * KW + kx] : 0.0f;
        v[j] = f32_to_bf16_bits(f);
    }
    uint4 out;
    out.x = v[0] | ((u32)v[1] << 16); out.y = v[2] | ((u32)v[3] << 16);
    out.z = v[4] | ((u32)v[5] << 16); out.w = v[6] | ((u32)v[7] << 16);
    reinterpret_cast<uint4 *>(packed)[e] = out;
}

// pack modes 2..7 -> (kind, operator strides); cin / cout are the LAYER's channel counts
struct PlanePack { int kind, n_out, n_in; long long so, sc; };
__host__ __device__ inline PlanePack plane_pack_of(int mode, int cin, int cout) {
    switch (mode) {
    case 2: return PlanePack{K_C3S2, cout, cin, (long long)cin * 9, 9};          // Conv2d(3, s 2) forward
    case 3: return PlanePack{K_C3S2, cin, cout, 9, (long long)cin * 9};          //                data gradient
    case 4: return PlanePack{K_K2S2, cout, cin, 4, (long long)cout * 4};         // ConvTranspose2d(2, s 2) forward
    case 5: return PlanePack{K_K2S2, cin, cout, (long long)cout * 4, 4};         //                data gradient
    case 6: return PlanePack{K_K1, cout, cin, 1, (long long)cout};               // ConvTranspose2d(1, s 1) forward
    default: return PlanePack{K_K1, cin, cout, (long long)cout, 1};              // (7)            data gradient
    }
}
__device__ __forceinline__ void planes_pack_dispatch(const float *w, int cin, int cout, int mode, unsigned short *packed,
                                                     size_t e) {
    const PlanePack P = plane_pack_of(mode, cin, cout);
    if (P.kind == K_C3S2) planes_pack_piece<K_C3S2>(w, P.n_out, P.n_in, P.so, P.sc, packed, e);
    else if (P.kind == K_K2S2) planes_pack_piece<K_K2S2>(w, P.n_out, P.n_in, P.so, P.sc, packed, e);
    else planes_pack_piece<K_K1>(w, P.n_out, P.n_in, P.so, P.sc, packed, e);
}

__global__ __launch_bounds__(256) void planes_pack_kernel(const float *__restrict__ w, int cin, int cout, int mode,
                                                          unsigned short *__restrict__ packed, size_t total) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e < total) planes_pack_dispatch(w, cin, cout, mode, packed, e);
}

template <int KIND, bool GATHER>
__global__ __launch_bounds__(256, 2) void conv2d_planes_kernel(const unsigned short *__restrict__ x, int B, int Hi, int Wi,
                                                               int cin, const uint4 *__restrict__ wp, int cout,
                                                               const float *__restrict__ bias,
                                                               unsigned short *__restrict__ y, int Ho, int Wo,
                                                               unsigned x_bytes, unsigned w_bytes) {
    constexpr int NP = pl_count<KIND>(), ST = pl_step<KIND>(), NT = taps_total<KIND>(), MAXT = taps_max<KIND>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *in_s = smem;                       // [2][IN_BYTES]
    char *w_s = smem + 2 * IN_BYTES;         // [MAXT][4 KB]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = lane & 15, g = lane >> 4;
    const int Hc = GATHER ? Ho : Hi, Wc = GATHER ? Wo : Wi;          // the coarse grid the tiles cover
    const int tiles_x = (Wc + TP - 1) / TP, tiles_y = (Hc + TP - 1) / TP;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int cg = blockIdx.y;
    const int zplane = GATHER ? 0 : (int)blockIdx.z;
    const int x0 = tx * TP, y0 = ty * TP;
    const int ncc = cin / 32;
    const int stages = (GATHER ? NP : 1) * ncc;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, (int)w_bytes, 0x00020000);

    u32x4 in_r[6], w_r[MAXT];
    auto load_stage = [&](int s) {
        const int plane = GATHER ? s / ncc : zplane;
        const int cc = GATHER ? s - plane * ncc : s;
        const int py = plane >> 1, px = plane & 1;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int p = it * 256 + threadIdx.x;
            const int pix = p >> 2, piece = p & 3;
            const int ry = pix / HT, rx = pix - ry * HT;
            int gy = y0 + ry - 1, gx = x0 + rx - 1;                  // coarse position of this tile pixel
            if (GATHER) { gy = gy * ST + py; gx = gx * ST + px; }    // -> the plane's pixel of the fine input
            const bool ok = p < HT * HT * 4 && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
            const unsigned off = ok ? (unsigned)((((size_t)b * Hi + gy) * Wi + gx) * cin * 2 + piece * 16 + cc * 64)
                                    : 0xFFFFFFF0u;
            in_r[it] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
        }
        const int nt = pl_taps<KIND>(plane);
        const unsigned wbase = (unsigned)((((size_t)cg * ncc + cc) * NT + pl_base<KIND>(plane)) * 4096);
#pragma unroll
        for (int it = 0; it < MAXT; ++it)
            w_r[it] = __builtin_amdgcn_raw_buffer_load_b128(
                wrs, it < nt ? wbase + (unsigned)(it * 256 + threadIdx.x) * 16u : 0xFFFFFFF0u, 0, 0);
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int p = it * 256 + threadIdx.x;
            if (p < HT * HT * 4) *reinterpret_cast<u32x4 *>(in_s + buf * IN_BYTES + p * 16) = in_r[it];
        }
#pragma unroll
        for (int it = 0; it < MAXT; ++it)
            *reinterpret_cast<u32x4 *>(w_s + (it * 256 + threadIdx.x) * 16) = w_r[it];
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) acc[mb][pb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto mma_plane = [&](auto plane_c, const char *ins) {
        constexpr int P = decltype(plane_c)::value;
        constexpr int py = P >> 1, px = P & 1, ny = ax_n<KIND>(py), nx = ax_n<KIND>(px);
#pragma unroll
        for (int ty_ = 0; ty_ < ny; ++ty_)
#pragma unroll
            for (int tx_ = 0; tx_ < nx; ++tx_) {
                const int tt = ty_ * nx + tx_;
                const int dy = ax_d<KIND, GATHER>(py, ty_), dx = ax_d<KIND, GATHER>(px, tx_);
                bf16x8 af[4], bf[4];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    af[mb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(w_s + ((tt * 4 + mb) * 64 + lane) * 16));
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    const int r = wave * 4 + pb;
                    bf[pb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(
                                                            ins + ((r + 1 + dy) * HT + (m + 1 + dx)) * 64 + g * 16));
                }
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb)
                        acc[mb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mb], bf[pb], acc[mb][pb], 0, 0, 0);
            }
    };

    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int s = 0; s < stages; ++s) {
        const int buf = s & 1;
        if (s + 1 < stages) load_stage(s + 1);
        const char *ins = in_s + buf * IN_BYTES;
        const int plane = GATHER ? s / ncc : zplane;
        if (NP == 1 || plane == 0) mma_plane(std::integral_constant<int, 0>{}, ins);
        else if (plane == 1) mma_plane(std::integral_constant<int, 1>{}, ins);
        else if (plane == 2) mma_plane(std::integral_constant<int, 2>{}, ins);
        else mma_plane(std::integral_constant<int, 3>{}, ins);
        __syncthreads();                                   // the weight stage (and this tile buffer) have been read
        if (s + 1 < stages) store_stage(buf ^ 1);
        __syncthreads();
    }
    const int c0 = cg * 64 + g * 16;
    float bv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bv[j] = (bias && c0 + j < cout) ? bias[c0 + j] : 0.0f;
    const int opy = zplane >> 1, opx = zplane & 1;
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
        int gy = y0 + wave * 4 + pb, gx = x0 + m;
        if (gy >= Hc || gx >= Wc) continue;
        if (!GATHER) { gy = gy * ST + opy; gx = gx * ST + opx; }
        if (gy >= Ho || gx >= Wo || c0 >= cout) continue;
        u32 o[8];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const u32 lo0 = f32_to_bf16_bits(acc[mb][pb][0] + bv[mb * 4 + 0]);
            const u32 hi0 = f32_to_bf16_bits(acc[mb][pb][1] + bv[mb * 4 + 1]);
            const u32 lo1 = f32_to_bf16_bits(acc[mb][pb][2] + bv[mb * 4 + 2]);
            const u32 hi1 = f32_to_bf16_bits(acc[mb][pb][3] + bv[mb * 4 + 3]);
            o[mb * 2] = lo0 | (hi0 << 16);
            o[mb * 2 + 1] = lo1 | (hi1 << 16);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(y + (((size_t)b * Ho + gy) * Wo + gx) * cout + c0);
        dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
        dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
}


// ---------------------------------------------------------------------------------------------
// Weight gradient of the dense 3x3 conv: dW[co][ky][kx][ci] = sum over pixels dy[p][co] * x[p + (ky - 1, kx - 1)][ci].
// (The sparse pair kernels of spconv.hip did this over dense pair lists: 9 gathers of the same x rows + two index loads
//  per pair.)  Here a workgroup owns a 64 (cin) x CO (cout) chunk of ALL 9 taps for its share of 4 x 32 pixel tiles:
// the x halo tile (6 x 34 pixels) and the dy tile are staged ONCE per tile, the contraction index of
// v_mfma_f32_16x16x32_bf16 is the pixel (32 consecutive pixels of a tile row per step), so both operands come out of LDS
// through the transposing ds_read_b64_tr_b16 exactly as in wgrad_kernel.  Wave w owns cin block w (16 channels) x all CO
// x 9 taps = 9 x NBW accumulators that live in registers over all of the workgroup's tiles; per step it reads 9 A
// fragments (the tap shifts are LDS row offsets) + NBW B fragments for 9 NBW MFMAs.  The next tile's global loads are
// in flight during the MFMAs.  Output: one slab [cout][9][cin] per split (fixed order -> deterministic), reduced by the
// sparse path's batched slab reduction (PcdWgradReduceJob.splits / layout / cout_write).
constexpr int WG_TH = 4, WG_TW = 32, WG_HH = WG_TH + 2, WG_HW = WG_TW + 2;
constexpr int WG_XS = 80;                                     // LDS pixel stride of x (elements): 64 channels + 16 pad
template <int CO> struct WgYs { static constexpr int value = ((CO / 16) % 2 == 0) ? CO + 16 : CO; };

typedef __attribute__((ext_vector_type(4))) short s16x4_;
typedef __attribute__((ext_vector_type(8))) short s16x8_;

template <int NBW>
__global__ __launch_bounds__(256, 2) void conv2d_wgrad_kernel(const unsigned short *__restrict__ x, int x_cs,
                                                              const unsigned short *__restrict__ dy, int B, int H, int W,
                                                              int cin, int cout, int n_splits, int n_chunks,
                                                              int n_co_chunks, float *__restrict__ slab,
                                                              unsigned x_bytes, unsigned dy_bytes) {
    constexpr int CO = NBW * 16, YS = WgYs<CO>::value;
    constexpr int XP = WG_HH * WG_HW * 8, YP = WG_TH * WG_TW * (CO / 8);        // 16-byte pieces per tile
    constexpr int XI = (XP + 255) / 256, YI = (YP + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned short *xs = (unsigned short *)smem;                               // [HH * HW][XS]
    unsigned short *ys = xs + WG_HH * WG_HW * WG_XS;                            // [TH * TW][YS]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, t = lane & 15;
    const int trow = 4 * g + (t >> 2);                       // contraction row this lane addresses (and trow + 16)
    // (split, chunk) items in contiguous runs per XCD, the chunks of one split next to each other: the workgroups that
    // read the same pixels share an L2
    const int items = n_splits * n_chunks;
    const int per_xcd = gridDim.x >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= items) return;
    const int split = item / n_chunks, chunk = item - split * n_chunks;
    const int ci0 = (chunk / n_co_chunks) * 64, co0 = (chunk % n_co_chunks) * CO;
    const int tiles_x = (W + WG_TW - 1) / WG_TW, tiles_y = (H + WG_TH - 1) / WG_TH;
    const int n_tiles = B * tiles_y * tiles_x;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, (int)dy_bytes, 0x00020000);

    f32x4 acc[9][NBW];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[k][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    u32x4 xr[XI], yr[YI];
    auto load_tile = [&](int tile) {
        int q = tile;
        const int tx = q % tiles_x; q /= tiles_x;
        const int ty = q % tiles_y;
        const int b = q / tiles_y;
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const int p = it * 256 + threadIdx.x;
            const int pix = p >> 3, piece = p & 7;
            const int hy = pix / WG_HW, hx = pix - hy * WG_HW;
            const int gy = ty * WG_TH + hy - 1, gx = tx * WG_TW + hx - 1;
            const bool ok = p < XP && tile < n_tiles && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const unsigned off = ok ? (unsigned)((((size_t)b * H + gy) * W + gx) * x_cs * 2 + (ci0 + piece * 8) * 2)
                                    : 0xFFFFFFF0u;
            xr[it] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < YI; ++it) {
            const int p = it * 256 + threadIdx.x;
            const int pix = p / (CO / 8), piece = p - pix * (CO / 8);
            const int py = pix / WG_TW, px = pix - py * WG_TW;
            const int gy = ty * WG_TH + py, gx = tx * WG_TW + px;
            const bool ok = p < YP && tile < n_tiles && gy < H && gx < W;
            const unsigned off = ok ? (unsigned)((((size_t)b * H + gy) * W + gx) * cout * 2 + (co0 + piece * 8) * 2)
                                    : 0xFFFFFFF0u;
            yr[it] = __builtin_amdgcn_raw_buffer_load_b128(yrs, off, 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const int p = it * 256 + threadIdx.x;
            if (p < XP) *reinterpret_cast<u32x4 *>(xs + (p >> 3) * WG_XS + (p & 7) * 8) = xr[it];
        }
#pragma unroll
        for (int it = 0; it < YI; ++it) {
            const int p = it * 256 + threadIdx.x;
            const int pix = p / (CO / 8), piece = p - pix * (CO / 8);
            if (p < YP) *reinterpret_cast<u32x4 *>(ys + pix * YS + piece * 8) = yr[it];
        }
    };
    typedef s16x4_ __attribute__((address_space(3))) * lds_tr_ptr;
    auto frag = [&](const unsigned short *a0, int hi_off) -> bf16x8 {
        s16x4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(a0));
        s16x4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(a0 + hi_off));
        s16x8_ cat = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, cat);
    };

    int tile = split;
    load_tile(tile);
    for (; tile < n_tiles; tile += n_splits) {
        store_tile();
        __syncthreads();
        load_tile(tile + n_splits);                          // (beyond the last tile: every offset out of range, no traffic)
#pragma unroll 1
        for (int s = 0; s < WG_TH; ++s) {
            bf16x8 bfr[NBW];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
                bfr[nb] = frag(ys + (s * WG_TW + trow) * YS + nb * 16 + (t & 3) * 4, 16 * YS);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int ky = k / 3, kx = k % 3;
                const bf16x8 af = frag(xs + ((s + ky) * WG_HW + trow + kx) * WG_XS + wave * 16 + (t & 3) * 4, 16 * WG_XS);
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
                    acc[k][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[nb], acc[k][nb], 0, 0, 0);
            }
        }
        __syncthreads();                                     // everyone is done reading this tile
    }
    // lane (g, t) of block nb holds dW[co = nb * 16 + t][ci = 16 wave + 4 g .. + 3] of every tap
    float *sl = slab + (size_t)split * cout * 9 * cin;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int co = co0 + nb * 16 + t, ci = ci0 + wave * 16 + 4 * g;
            *reinterpret_cast<float4 *>(sl + ((size_t)co * 9 + k) * cin + ci) =
                make_float4(acc[k][nb][0], acc[k][nb][1], acc[k][nb][2], acc[k][nb][3]);
        }
}


// Weight gradients of the plane operators (pcd_conv2d_planes_nhwc): the same tile / transposing-read machinery as
// conv2d_wgrad_kernel with the A operand taken from the FINE map plane by plane (stride-2 pixel addressing, the plane's
// 1 / 2 / 4 taps as LDS row offsets) and the B operand from the coarse map, staged once per tile:
//   C3S2  A = x (fine), B = dy (coarse)               -> slabs [cout][9][cin]
//   K2S2  A = dy (fine), B = x (coarse)                -> slabs [cin][4][cout]   (the ConvTranspose2d parameter's layout)
//   K1    A = dy, B = x, one plane, one tap            -> slabs [cin][1][cout]
template <int KIND, int NBW>
__global__ __launch_bounds__(256, KIND == K_C3S2 ? 1 : 2) void conv2d_wgrad_planes_kernel(
    const unsigned short *__restrict__ xa, int Ha, int Wa, int ca, const unsigned short *__restrict__ xb, int B, int Hc,
    int Wc, int cb, int n_splits, int n_chunks, int n_cb_chunks, float *__restrict__ slab, unsigned a_bytes,
    unsigned b_bytes) {
    constexpr int NP = pl_count<KIND>(), ST = pl_step<KIND>(), NT = taps_total<KIND>(), KW = kernel_w<KIND>();
    constexpr int HL = KIND == K_C3S2 ? 1 : 0;                                   // halo (low and high side)
    constexpr int AH = WG_TH + 2 * HL, AW = WG_TW + 2 * HL;
    constexpr int CO = NBW * 16, YS = WgYs<CO>::value;
    constexpr int XP = AH * AW * 8, YP = WG_TH * WG_TW * (CO / 8);
    constexpr int XI = (XP + 255) / 256, YI = (YP + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned short *xs = (unsigned short *)smem;                               // [AH * AW][XS]   one plane of the A tile
    unsigned short *ys = xs + AH * AW * WG_XS;                                  // [TH * TW][YS]   the B tile
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, t = lane & 15;
    const int trow = 4 * g + (t >> 2);
    const int items = n_splits * n_chunks;
    const int per_xcd = gridDim.x >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= items) return;
    const int split = item / n_chunks, chunk = item - split * n_chunks;
    const int ca0 = (chunk / n_cb_chunks) * 64, cb0 = (chunk % n_cb_chunks) * CO;
    const int tiles_x = (Wc + WG_TW - 1) / WG_TW, tiles_y = (Hc + WG_TH - 1) / WG_TH;
    const int n_tiles = B * tiles_y * tiles_x;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void *)xa, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, (int)b_bytes, 0x00020000);

    f32x4 acc[NT][NBW];
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[k][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    u32x4 xr[XI], yr[YI];
    auto load_a = [&](int tile, int plane) {
        int q = tile;
        const int tx = q % tiles_x; q /= tiles_x;
        const int ty = q % tiles_y;
        const int b = q / tiles_y;
        const int py = plane >> 1, px = plane & 1;
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const int p = it * 256 + threadIdx.x;
            const int pix = p >> 3, piece = p & 7;
            const int hy = pix / AW, hx = pix - hy * AW;
            const int gy = (ty * WG_TH + hy - HL) * ST + py, gx = (tx * WG_TW + hx - HL) * ST + px;
            const bool ok = p < XP && tile < n_tiles && gy >= 0 && gy < Ha && gx >= 0 && gx < Wa;
            const unsigned off = ok ? (unsigned)((((size_t)b * Ha + gy) * Wa + gx) * ca * 2 + (ca0 + piece * 8) * 2)
                                    : 0xFFFFFFF0u;
            xr[it] = __builtin_amdgcn_raw_buffer_load_b128(ars, off, 0, 0);
        }
    };
    auto load_b = [&](int tile) {
        int q = tile;
        const int tx = q % tiles_x; q /= tiles_x;
        const int ty = q % tiles_y;
        const int b = q / tiles_y;
#pragma unroll
        for (int it = 0; it < YI; ++it) {
            const int p = it * 256 + threadIdx.x;
            const int pix = p / (CO / 8), piece = p - pix * (CO / 8);
            const int qy = pix / WG_TW, qx = pix - qy * WG_TW;
            const int gy = ty * WG_TH + qy, gx = tx * WG_TW + qx;
            const bool ok = p < YP && tile < n_tiles && gy < Hc && gx < Wc;
            const unsigned off = ok ? (unsigned)((((size_t)b * Hc + gy) * Wc + gx) * cb * 2 + (cb0 + piece * 8) * 2)
                                    : 0xFFFFFFF0u;
            yr[it] = __builtin_amdgcn_raw_buffer_load_b128(brs, off, 0, 0);
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const int p = it * 256 + threadIdx.x;
            if (p < XP) *reinterpret_cast<u32x4 *>(xs + (p >> 3) * WG_XS + (p & 7) * 8) = xr[it];
        }
    };
    auto store_b = [&]() {
#pragma unroll
        for (int it = 0; it < YI; ++it) {
            const int p = it * 256 + threadIdx.x;
            const int pix = p / (CO / 8), piece = p - pix * (CO / 8);
            if (p < YP) *reinterpret_cast<u32x4 *>(ys + pix * YS + piece * 8) = yr[it];
        }
    };
    typedef s16x4_ __attribute__((address_space(3))) * lds_tr_ptr;
    auto frag = [&](const unsigned short *a0, int hi_off) -> bf16x8 {
        s16x4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(a0));
        s16x4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(a0 + hi_off));
        s16x8_ cat = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, cat);
    };
    auto mma_plane = [&](auto plane_c) {
        constexpr int P = decltype(plane_c)::value;
        constexpr int py = P >> 1, px = P & 1, ny = ax_n<KIND>(py), nx = ax_n<KIND>(px), base = pl_base<KIND>(P);
#pragma unroll 1
        for (int s = 0; s < WG_TH; ++s) {
            bf16x8 bfr[NBW];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
                bfr[nb] = frag(ys + (s * WG_TW + trow) * YS + nb * 16 + (t & 3) * 4, 16 * YS);
#pragma unroll
            for (int ty_ = 0; ty_ < ny; ++ty_)
#pragma unroll
                for (int tx_ = 0; tx_ < nx; ++tx_) {
                    const int dy = ax_d<KIND, true>(py, ty_), dx = ax_d<KIND, true>(px, tx_);
                    const bf16x8 af = frag(xs + ((s + HL + dy) * AW + trow + HL + dx) * WG_XS + wave * 16 + (t & 3) * 4,
                                           16 * WG_XS);
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb)
                        acc[base + ty_ * nx + tx_][nb] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[nb], acc[base + ty_ * nx + tx_][nb], 0, 0, 0);
                }
        }
    };

    int tile = split;
    load_a(tile, 0);
    load_b(tile);
    for (; tile < n_tiles; tile += n_splits) {
#pragma unroll 1
        for (int plane = 0; plane < NP; ++plane) {
            store_a();
            if (plane == 0) store_b();
            __syncthreads();
            if (plane + 1 < NP) {
                load_a(tile, plane + 1);
            } else {
                load_a(tile + n_splits, 0);                  // (beyond the last tile: all offsets out of range)
                load_b(tile + n_splits);
            }
            if (NP == 1 || plane == 0) mma_plane(std::integral_constant<int, 0>{});
            else if (plane == 1) mma_plane(std::integral_constant<int, 1>{});
            else if (plane == 2) mma_plane(std::integral_constant<int, 2>{});
            else mma_plane(std::integral_constant<int, 3>{});
            __syncthreads();
        }
    }
    // slab [cb][KH * KW][ca]: lane (g, t) of block nb holds rows cb0 + nb * 16 + t, columns ca0 + 16 wave + 4 g .. + 3
    float *sl = slab + (size_t)split * cb * (KW * KW) * ca;
#pragma unroll
    for (int P = 0; P < NP; ++P) {
        const int py = P >> 1, px = P & 1, nx = ax_n<KIND>(px);
#pragma unroll
        for (int tt = 0; tt < pl_taps<KIND>(P); ++tt) {
            const int kop = ax_k<KIND>(py, tt / nx) * KW + ax_k<KIND>(px, tt % nx);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const int rb = cb0 + nb * 16 + t, cacol = ca0 + wave * 16 + 4 * g;
                const f32x4 v = acc[pl_base<KIND>(P) + tt][nb];
                *reinterpret_cast<float4 *>(sl + ((size_t)rb * (KW * KW) + kop) * ca + cacol) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

}  // namespace

static int pad32(int c) { return (c + 31) / 32 * 32; }

// (output channels are padded to a multiple of 32: the data gradient contracts over them in steps of 32)
static size_t planes_pack_bytes(int cin, int cout, int mode) {
    const PlanePack P = plane_pack_of(mode, cin, cout);
    if (P.n_in % 32) return 0;
    const int nt = P.kind == K_C3S2 ? taps_total<K_C3S2>() : (P.kind == K_K2S2 ? taps_total<K_K2S2>() : taps_total<K_K1>());
    return (size_t)((P.n_out + 63) / 64) * (P.n_in / 32) * nt * 4096;
}

extern "C" size_t pcd_conv2d_packed_weight_bytes(int cin, int cout, int mode) {
    if (cin <= 0 || cout <= 0 || mode < 0 || mode > 7) return 0;
    if (mode >= 2) return planes_pack_bytes(cin, cout, mode);
    const int cp = pad32(cout);
    const int n_out = mode == 0 ? cp : cin, n_in = mode == 0 ? cin : cp;
    return (size_t)((n_out + 63) / 64) * (n_in / 32) * W_BYTES;
}

extern "C" int pcd_conv2d_pack_weight(const float *weight, int cin, int cout, int mode, void *packed, void *stream) {
    PCD_ENTER();
    if (!weight || !packed || cin <= 0 || cout <= 0 || mode < 0 || mode > 7) return PCD_ERR_INVALID_ARG;
    if (mode >= 2) {
        const size_t pieces = planes_pack_bytes(cin, cout, mode) / 16;
        if (pieces == 0) return PCD_ERR_UNSUPPORTED;
        planes_pack_kernel<<<(unsigned)((pieces + 255) / 256), 256, 0, (hipStream_t)stream>>>(
            weight, cin, cout, mode, (unsigned short *)packed, pieces);
        PCD_RETURN_IF_LAUNCH_FAILED();
        return PCD_OK;
    }
    if (cin % 32) return PCD_ERR_UNSUPPORTED;
    const size_t total = pcd_conv2d_packed_weight_bytes(cin, cout, mode) / 16;
    conv2d_pack_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        weight, cin, cout, pad32(cout), mode, (unsigned short *)packed, total);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_conv2d_pack_weights_batched(const void *table, int n, int total_blocks, void *stream) {
    PCD_ENTER();
    if (n < 0 || total_blocks < 0 || (n > 0 && !table)) return PCD_ERR_INVALID_ARG;
    if (n == 0 || total_blocks == 0) return PCD_OK;
    conv2d_pack_batched_kernel<<<(unsigned)total_blocks, 256, 0, (hipStream_t)stream>>>((const long long *)table, n);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_conv2d_3x3_tiles(int batch, int height, int width) {
    if (batch <= 0 || height <= 0 || width <= 0) return PCD_ERR_INVALID_ARG;
    return batch * ((height + TP - 1) / TP) * ((width + TP - 1) / TP);
}

extern "C" int pcd_conv2d_3x3_nhwc_bn(const void *x, int x_cs, int batch, int height, int width, int cin,
                                      const void *packed_w, int cout, const float *bias, void *y, int y_cs,
                                      const PcdBnReduce *bnr, void *stream) {
    PCD_ENTER();
    DenseBn bn = {};
    if (bnr && bnr->mode) {
        if (bnr->mode != 1 && bnr->mode != 2) return PCD_ERR_INVALID_ARG;
        if (!bnr->partial || bnr->partial_rows != pcd_conv2d_3x3_tiles(batch, height, width) || y_cs != cout)
            return PCD_ERR_INVALID_ARG;
        if (bnr->mode == 2 && (!bnr->x || !bnr->mean || !bnr->invstd || (bnr->relu && !bnr->y))) return PCD_ERR_INVALID_ARG;
        if ((bnr->mid != nullptr) != (bnr->counters != nullptr)) return PCD_ERR_INVALID_ARG;
        bn.mode = bnr->mode;
        bn.relu = bnr->relu;
        bn.x = (const unsigned short *)bnr->x;
        bn.y = (const unsigned short *)bnr->y;
        bn.mean = bnr->mean;
        bn.invstd = bnr->invstd;
        bn.partial = bnr->partial;
        bn.mid = bnr->mid;
        bn.counters = bnr->counters;
    }
    if (batch <= 0 || height <= 0 || width <= 0 || cin <= 0 || cout <= 0) return PCD_ERR_INVALID_ARG;
    if (!x || !packed_w || !y || x_cs < cin || y_cs < cout || x_cs % 8 || y_cs % 8) return PCD_ERR_INVALID_ARG;
    if (cin % 32 || cout % 16) return PCD_ERR_UNSUPPORTED;
    const double xb = ((double)batch * height * width - 1) * x_cs * 2 + (double)cin * 2;
    if (xb >= 4294966000.0) return PCD_ERR_UNSUPPORTED;
    const size_t wb_bytes = (size_t)((cout + 63) / 64) * (cin / 32) * W_BYTES;   // bytes of the pack the kernel reads
    const int wb = pcd_opt(PCD_OPT_CONV2D_WB);   // (1: 28-35 % of the MFMA peak, 2: 21-27 %)
    const size_t lds = 2 * (size_t)IN_BYTES + (size_t)(wb == 1 ? 1 : 2) * (size_t)W_BYTES;
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute((const void *)conv2d_3x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(2 * IN_BYTES + 2 * W_BYTES)) != hipSuccess ||
            hipFuncSetAttribute((const void *)conv2d_3x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(2 * IN_BYTES + W_BYTES)) != hipSuccess)
            return PCD_ERR_LAUNCH;
        raised = true;
    }
    const int tiles = batch * ((height + TP - 1) / TP) * ((width + TP - 1) / TP);
    dim3 grid((unsigned)tiles, (unsigned)((cout + 63) / 64));
    if (wb == 1)
        conv2d_3x3_kernel<1><<<grid, 256, lds, (hipStream_t)stream>>>((const unsigned short *)x, batch, height, width, cin,
                                                                     (const uint4 *)packed_w, cout, bias,
                                                                     (unsigned short *)y, (unsigned)xb, (unsigned)wb_bytes,
                                                                     x_cs, y_cs, bn);
    else
        conv2d_3x3_kernel<2><<<grid, 256, lds, (hipStream_t)stream>>>((const unsigned short *)x, batch, height, width, cin,
                                                                     (const uint4 *)packed_w, cout, bias,
                                                                     (unsigned short *)y, (unsigned)xb, (unsigned)wb_bytes,
                                                                     x_cs, y_cs, bn);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_conv2d_3x3_nhwc_ld(const void *x, int x_cs, int batch, int height, int width, int cin,
                                      const void *packed_w, int cout, const float *bias, void *y, int y_cs, void *stream) {
    return pcd_conv2d_3x3_nhwc_bn(x, x_cs, batch, height, width, cin, packed_w, cout, bias, y, y_cs, nullptr, stream);
}

extern "C" int pcd_conv2d_3x3_nhwc(const void *x, int batch, int height, int width, int cin, const void *packed_w,
                                   int cout, const float *bias, void *y, void *stream) {
    return pcd_conv2d_3x3_nhwc_bn(x, cin, batch, height, width, cin, packed_w, cout, bias, y, cout, nullptr, stream);
}


template <int KIND, bool GATHER>
static int launch_planes(const void *x, int batch, int hi, int wi, int cin, const void *packed_w, int cout,
                         const float *bias, void *y, int ho, int wo, hipStream_t stream) {
    constexpr int NT = taps_total<KIND>();
    const size_t lds = 2 * (size_t)IN_BYTES + (size_t)taps_max<KIND>() * 4096;
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute((const void *)conv2d_planes_kernel<KIND, GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return PCD_ERR_LAUNCH;
        raised = true;
    }
    const int hc = GATHER ? ho : hi, wc = GATHER ? wo : wi;
    const int tiles = batch * ((hc + TP - 1) / TP) * ((wc + TP - 1) / TP);
    const size_t wb_bytes = (size_t)((cout + 63) / 64) * (cin / 32) * NT * 4096;
    dim3 grid((unsigned)tiles, (unsigned)((cout + 63) / 64), GATHER ? 1u : (unsigned)pl_count<KIND>());
    conv2d_planes_kernel<KIND, GATHER><<<grid, 256, lds, stream>>>(
        (const unsigned short *)x, batch, hi, wi, cin, (const uint4 *)packed_w, cout, bias, (unsigned short *)y, ho, wo,
        (unsigned)((size_t)batch * hi * wi * cin * 2), (unsigned)wb_bytes);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_conv2d_planes_nhwc(int pack_mode, const void *x, int batch, int hi, int wi, int cin,
                                      const void *packed_w, int cout, const float *bias, void *y, int ho, int wo,
                                      void *stream) {
    PCD_ENTER();
    if (batch <= 0 || hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0 || cin <= 0 || cout <= 0) return PCD_ERR_INVALID_ARG;
    if (!x || !packed_w || !y || pack_mode < 2 || pack_mode > 7) return PCD_ERR_INVALID_ARG;
    if (cin % 32 || cout % 16) return PCD_ERR_UNSUPPORTED;
    if ((double)batch * hi * wi * cin * 2 >= 4294966000.0 || (double)batch * ho * wo * cout * 2 >= 1.7e10)
        return PCD_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    switch (pack_mode) {
    case 2:   // Conv2d(3, stride 2, padding 1) forward: x fine [hi, wi] -> y coarse
        if (ho != (hi - 1) / 2 + 1 || wo != (wi - 1) / 2 + 1) return PCD_ERR_INVALID_ARG;
        return launch_planes<K_C3S2, true>(x, batch, hi, wi, cin, packed_w, cout, bias, y, ho, wo, st);
    case 3:   // its data gradient: x = dy coarse [hi, wi] -> y = dx fine [ho, wo]
        if (hi != (ho - 1) / 2 + 1 || wi != (wo - 1) / 2 + 1) return PCD_ERR_INVALID_ARG;
        return launch_planes<K_C3S2, false>(x, batch, hi, wi, cin, packed_w, cout, bias, y, ho, wo, st);
    case 4:   // ConvTranspose2d(2, stride 2) forward: coarse -> fine
        if (ho != 2 * hi || wo != 2 * wi) return PCD_ERR_INVALID_ARG;
        return launch_planes<K_K2S2, false>(x, batch, hi, wi, cin, packed_w, cout, bias, y, ho, wo, st);
    case 5:   // its data gradient: fine -> coarse
        if (hi != 2 * ho || wi != 2 * wo) return PCD_ERR_INVALID_ARG;
        return launch_planes<K_K2S2, true>(x, batch, hi, wi, cin, packed_w, cout, bias, y, ho, wo, st);
    default:  // 6 / 7: ConvTranspose2d(1, stride 1) forward / data gradient
        if (ho != hi || wo != wi) return PCD_ERR_INVALID_ARG;
        return launch_planes<K_K1, false>(x, batch, hi, wi, cin, packed_w, cout, bias, y, ho, wo, st);
    }
}


// splits (= slabs of [cout][9][cin] f32 the reduction has to sum) pcd_conv2d_wgrad_3x3_nhwc writes; 0 = shape not covered
extern "C" int pcd_conv2d_wgrad_3x3_splits(int batch, int height, int width, int cin, int cout) {
    if (batch <= 0 || height <= 0 || width <= 0 || cin <= 0 || cout <= 0 || cin % 64 || cout % 32) return 0;
    const int tiles = batch * ((height + WG_TH - 1) / WG_TH) * ((width + WG_TW - 1) / WG_TW);
    const int chunks = (cin / 64) * (cout % 64 == 0 ? cout / 64 : cout / 32);
    const int target = pcd_opt(PCD_OPT_CONV2D_WG_BLOCKS);
    // about 128 workgroups per launch: measured in the full step (tools/exp_wgblocks.sh) 64 / 96 / 128 / 192 / 256 / 512 /
    // 1024 -> 8.75 / 8.11 / 7.86 / 7.89 / 7.96 / 8.21 / 8.50 ms -- the kernel runs BESIDE the data-gradient chain: fewer, longer
    // workgroups leave that chain half of the CUs and write fewer slabs
    int splits = (target + chunks - 1) / chunks;
    if (splits > tiles) splits = tiles;
    if (splits < 1) splits = 1;
    // ... with an equal number of tiles each where that is close (188 x 188 x 4: 1152 tiles)
    const int per = (tiles + splits - 1) / splits;
    return (tiles + per - 1) / per;
}

extern "C" int pcd_conv2d_wgrad_3x3_nhwc(const void *x, int x_cs, const void *dy, int batch, int height, int width, int cin,
                                         int cout, void *slabs, size_t slab_bytes, void *stream) {
    PCD_ENTER();
    const int splits = pcd_conv2d_wgrad_3x3_splits(batch, height, width, cin, cout);
    if (splits <= 0) return PCD_ERR_UNSUPPORTED;
    if (!x || !dy || !slabs || x_cs < cin || x_cs % 8) return PCD_ERR_INVALID_ARG;
    if (slab_bytes < (size_t)splits * cout * 9 * cin * sizeof(float)) return PCD_ERR_WORKSPACE;
    const double xb = ((double)batch * height * width - 1) * x_cs * 2 + (double)cin * 2;
    const double yb = (double)batch * height * width * cout * 2;
    if (xb >= 4294966000.0 || yb >= 4294966000.0) return PCD_ERR_UNSUPPORTED;
    const bool wide = cout % 64 == 0;
    const int n_co = wide ? cout / 64 : cout / 32, chunks = (cin / 64) * n_co;
    const int grid = (splits * chunks + 7) / 8 * 8;
    hipStream_t st = (hipStream_t)stream;
    if (wide) {
        const size_t lds = (size_t)(WG_HH * WG_HW * WG_XS + WG_TH * WG_TW * WgYs<64>::value) * 2;
        conv2d_wgrad_kernel<4><<<grid, 256, lds, st>>>((const unsigned short *)x, x_cs, (const unsigned short *)dy, batch,
                                                     height, width, cin, cout, splits, chunks, n_co, (float *)slabs,
                                                     (unsigned)xb, (unsigned)yb);
    } else {
        const size_t lds = (size_t)(WG_HH * WG_HW * WG_XS + WG_TH * WG_TW * WgYs<32>::value) * 2;
        conv2d_wgrad_kernel<2><<<grid, 256, lds, st>>>((const unsigned short *)x, x_cs, (const unsigned short *)dy, batch,
                                                     height, width, cin, cout, splits, chunks, n_co, (float *)slabs,
                                                     (unsigned)xb, (unsigned)yb);
    }
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}


// Weight gradient of the plane operators (forward pack modes 2 / 4 / 6 of pcd_conv2d_planes_nhwc).  `fine` / `coarse`: the
// two maps of the layer ([b][hf][wf][cf] and [b][hc][wc][cc] bf16, contiguous) -- mode 2: x / dy; modes 4, 6: dy / x.
// Slabs [cc][k * k][cf] f32 (finish with pcd_sparse_conv_wgrad_reduce_batched: kvol = k * k, cin = cf, cout = cc, layout 1
// gives the torch parameter's layout for all three: Conv2d [cout][cin][3][3], ConvTranspose2d [cin][cout][k][k]).
extern "C" int pcd_conv2d_wgrad_planes_splits(int mode, int batch, int hc, int wc, int cf, int cc) {
    if ((mode != 2 && mode != 4 && mode != 6) || batch <= 0 || hc <= 0 || wc <= 0 || cf <= 0 || cc <= 0 || cf % 64 || cc % 32)
        return 0;
    // mode 2 (stride-2 conv: 9 accumulator sets + 4 plane stagings per tile, one workgroup per CU) measures SLOWER than the
    // pair kernels over dense pair lists (99-152 vs 80 us, tools/exp_wgrad_planes.py): not offered unless asked for
    const int mode2 = pcd_opt(PCD_OPT_CONV2D_WGP_MODE2);
    if (mode == 2 && !mode2) return 0;
    const int tiles = batch * ((hc + WG_TH - 1) / WG_TH) * ((wc + WG_TW - 1) / WG_TW);
    const int chunks = (cf / 64) * (cc % 64 == 0 ? cc / 64 : cc / 32);
    const int target = pcd_opt(PCD_OPT_CONV2D_WGP_BLOCKS);
    int splits = (target + chunks - 1) / chunks;
    if (splits > tiles) splits = tiles;
    if (splits < 1) splits = 1;
    const int per = (tiles + splits - 1) / splits;
    return (tiles + per - 1) / per;
}

template <int KIND, int NBW>
static void launch_wgrad_planes(const void *fine, int hf, int wf, int cf, const void *coarse, int batch, int hc, int wc, int cc,
                                int splits, void *slabs, hipStream_t st) {
    constexpr int HL = KIND == K_C3S2 ? 1 : 0;
    const int n_cc = cc / (NBW * 16), chunks = (cf / 64) * n_cc;
    const int grid = (splits * chunks + 7) / 8 * 8;
    const size_t lds = (size_t)((WG_TH + 2 * HL) * (WG_TW + 2 * HL) * WG_XS + WG_TH * WG_TW * WgYs<NBW * 16>::value) * 2;
    conv2d_wgrad_planes_kernel<KIND, NBW><<<grid, 256, lds, st>>>(
        (const unsigned short *)fine, hf, wf, cf, (const unsigned short *)coarse, batch, hc, wc, cc, splits, chunks, n_cc,
        (float *)slabs, (unsigned)((size_t)batch * hf * wf * cf * 2), (unsigned)((size_t)batch * hc * wc * cc * 2));
}

extern "C" int pcd_conv2d_wgrad_planes_nhwc(int mode, const void *fine, int hf, int wf, int cf, const void *coarse, int batch,
                                            int hc, int wc, int cc, void *slabs, size_t slab_bytes, void *stream) {
    PCD_ENTER();
    const int splits = pcd_conv2d_wgrad_planes_splits(mode, batch, hc, wc, cf, cc);
    if (splits <= 0) return PCD_ERR_UNSUPPORTED;
    if (!fine || !coarse || !slabs) return PCD_ERR_INVALID_ARG;
    const int kk = mode == 2 ? 9 : (mode == 4 ? 4 : 1);
    if (mode == 2 ? (hc != (hf - 1) / 2 + 1 || wc != (wf - 1) / 2 + 1)
                  : (mode == 4 ? (hf != 2 * hc || wf != 2 * wc) : (hf != hc || wf != wc)))
        return PCD_ERR_INVALID_ARG;
    if (slab_bytes < (size_t)splits * cc * kk * cf * sizeof(float)) return PCD_ERR_WORKSPACE;
    if ((double)batch * hf * wf * cf * 2 >= 4294966000.0 || (double)batch * hc * wc * cc * 2 >= 4294966000.0)
        return PCD_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const bool wide = cc % 64 == 0;
#define PCD_WGP(K)                                                                                                      \
    if (wide) launch_wgrad_planes<K, 4>(fine, hf, wf, cf, coarse, batch, hc, wc, cc, splits, slabs, st);                \
    else launch_wgrad_planes<K, 2>(fine, hf, wf, cf, coarse, batch, hc, wc, cc, splits, slabs, st);
    if (mode == 2) { PCD_WGP(K_C3S2) } else if (mode == 4) { PCD_WGP(K_K2S2) } else { PCD_WGP(K_K1) }
#undef PCD_WGP
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
