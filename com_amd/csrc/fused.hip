// Fused sparse epilogues for gfx950: BatchNorm1d (+ residual add) (+ ReLU) over a [n][c] feature matrix
// (nn.BatchNorm1d(eps=1e-3, momentum=0.01) + nn.ReLU between every sparse conv and the residual add of
// SparseBasicBlock: pcdet/models/backbones_3d/spconv_backbone.py:21-25,50-66,73).
//
// All HBM-bound streaming kernels: 16-byte loads/stores, a lane always owns the same 16-byte channel piece
// (grid strides are multiples of the pieces per row) so per-channel parameters stay in registers.
//   forward (training): pass 1  per-block partial (sum, sum of squares) -> workspace;
//                       finalize mean / invstd (+ running-stat update) in one small block (fixed order);
//                       pass 2  y = relu(x * scale + shift + residual).
//   backward:           pass 1  dbeta = sum dz, dgamma = sum dz * xhat, dz = dy * (y > 0);
//                       pass 2  dx = gamma * invstd * (dz - dbeta/n - xhat * dgamma/n), dresidual = dz.
// Deterministic (no atomics): partials are reduced in a fixed order.
#include "common.h"

namespace {

constexpr int MAX_BLOCKS = 512;       // blocks of the reduction passes (= rows of the partial buffer)
constexpr int MAX_APPLY_BLOCKS = 2048; // blocks of the streaming apply passes
// 16-byte pieces every thread of an apply pass requests BEFORE anything else (the grid gives a thread >= BN_PPT = 4 pieces).
// Measured in the training step, same box, alternating builds: 2 in flight 3.138 ms, all 4 in flight 3.148, twice the
// workgroups with 2 pieces each 3.22 -- these passes run beside the weight-gradient stream, more of them in flight at once
// takes from it what it gives them.
#ifndef BN_PF_VALUE
#define BN_PF_VALUE 2
#endif
constexpr int BN_PF = BN_PF_VALUE;

template <typename T>
struct Piece;  // 16-byte piece of a row
template <>
struct Piece<float> {
    static constexpr int N = 4;
    typedef float4 Raw;
    __device__ static Raw raw(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    __device__ static void decode(const Raw &r, float (&v)[4]) { v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w; }
    __device__ static void load(const float *p, float (&v)[4]) {
        float4 r = *reinterpret_cast<const float4 *>(p);
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    }
    __device__ static void store(float *p, const float (&v)[4]) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __device__ static float stored(float v) { return v; }   // the value a store leaves in memory
};
template <>
struct Piece<unsigned short> {
    static constexpr int N = 8;
    typedef uint4 Raw;
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    __device__ static Raw raw(const unsigned short *p) { return *reinterpret_cast<const uint4 *>(p); }
    __device__ static void decode(const Raw &r, float (&v)[8]) {
        u32 w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[2 * j] = __uint_as_float(w[j] << 16);
            v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        }
    }
    __device__ static void load(const unsigned short *p, float (&v)[8]) { decode(raw(p), v); }
    // v_cvt_pk_bf16_f32: round to nearest even, two values per instruction (= f32_to_bf16_bits for every finite value)
    __device__ static u32 pack2(float a, float b) {
        return __builtin_bit_cast(u32, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
    }
    __device__ static void store(unsigned short *p, const float (&v)[8]) {
        u32 w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = pack2(v[2 * j], v[2 * j + 1]);
        *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __device__ static float stored(float v) { return __uint_as_float(pack2(v, 0.0f) << 16); }
};

// A thread's N per-channel parameters.  16-byte loads when the arrays allow it (`vec`, decided on the host from the
// pointer alignment): with one 4-byte load per channel the prologue of a streaming kernel was up to 48 wave-wide
// load instructions (~32 clk each in the CU's single texture-address unit) -- a third of the kernel's run time.
template <int N>
__device__ __forceinline__ void load_params(const float *__restrict__ p, int ch0, bool vec, float (&v)[N]) {
    if (vec) {
#pragma unroll
        for (int j = 0; j < N; j += 4) {
            float4 t = *reinterpret_cast<const float4 *>(p + ch0 + j);
            v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = p[ch0 + j];
    }
}
template <int N>
__device__ __forceinline__ void fill_params(float val, float (&v)[N]) {
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = val;
}
static bool aligned16(const void *a, const void *b = nullptr, const void *c = nullptr, const void *d = nullptr,
                      const void *e = nullptr, const void *f = nullptr) {
    const void *ps[6] = {a, b, c, d, e, f};
    for (const void *q : ps)
        if (q && ((uintptr_t)q & 15u)) return false;
    return true;
}

// block-level reduction of per-thread piece accumulators -> partial[blockIdx][which][c].
// pcs <= 64: the lanes of a wave that own the same piece (lane % pcs) are summed with a shuffle butterfly, then
// the 4 waves through 4 * NACC * c floats of LDS (4 KB at c = 128) -- the previous form staged all 256 threads
// (16 KB), which kept these blocks from being scheduled beside the weight-gradient workgroups that fill the LDS.
// Fixed summation tree -> deterministic.  `lds` is dynamic shared memory of bn_reduce_lds_bytes(c, N) bytes.
template <int N, int NACC>
__device__ __forceinline__ void block_reduce_store(float (&acc)[NACC][N], int c, int pcs,
                                                   float *partial /*[gridDim.x][NACC][c]*/, float *lds) {
    if (pcs <= 64 && (pcs & (pcs - 1)) == 0 && blockDim.x == 256) {
        for (int off = pcs; off < 64; off <<= 1) {
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int j = 0; j < N; ++j) acc[a][j] += __shfl_xor(acc[a][j], off);
        }
        const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
        if (l < pcs) {
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int j = 0; j < N; ++j) lds[(w * NACC + a) * c + l * N + j] = acc[a][j];
        }
        __syncthreads();
        for (int item = threadIdx.x; item < NACC * c; item += 256)
            partial[(size_t)blockIdx.x * NACC * c + item] =
                ((lds[item] + lds[NACC * c + item]) + lds[2 * NACC * c + item]) + lds[3 * NACC * c + item];
        return;
    }
    // general form (pcs > 64, or a piece count that is no power of two: the block then has (256 / pcs) * pcs threads)
    // lds: [blockDim][NACC*N]
    const int BT = (int)blockDim.x;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int j = 0; j < N; ++j) lds[threadIdx.x * (NACC * N) + a * N + j] = acc[a][j];
    __syncthreads();
    for (int item = threadIdx.x; item < NACC * c; item += BT) {
        int a = item / c, ch = item - a * c;
        int piece = ch / N, j = ch - piece * N;
        float s = 0.0f;
        for (int t = piece; t < BT; t += pcs) s += lds[t * (NACC * N) + a * N + j];
        partial[((size_t)blockIdx.x * NACC + a) * c + ch] = s;
    }
}
static size_t bn_reduce_lds_bytes(int c, int N) {
    const int pcs = c / N;
    return ((pcs <= 64 && (pcs & (pcs - 1)) == 0) ? (size_t)4 * 2 * c : (size_t)256 * 2 * N) * sizeof(float);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T *__restrict__ x, int n_cap,
                                                       const int32_t *n_dev, int c,
                                                       float *__restrict__ partial) {
    constexpr int N = Piece<T>::N;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int pcs = c / N;
    const size_t total = (size_t)eff_rows(n_dev, n_cap) * pcs;
    float acc[2][N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[0][j] = acc[1][j] = 0.0f;
    // 4 pieces in flight per thread: with 2 workgroups per CU a thread's 4-5 pieces otherwise cost one full memory
    // latency EACH (the same accumulation order as the plain loop, so results do not change)
    const size_t S = (size_t)gridDim.x * blockDim.x;
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; e + 3 * S < total; e += 4 * S) {
        float v[4][N];
#pragma unroll
        for (int u = 0; u < 4; ++u) Piece<T>::load(x + (e + u * S) * N, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                acc[0][j] += v[u][j];
                acc[1][j] += v[u][j] * v[u][j];
            }
    }
    for (; e < total; e += S) {
        float v[N];
        Piece<T>::load(x + e * N, v);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            acc[0][j] += v[j];
            acc[1][j] += v[j] * v[j];
        }
    }
    block_reduce_store<N, 2>(acc, c, pcs, partial, lds);
}

// Reduce the per-block partials [nblocks][2][c] with ONE 1024-thread workgroup: thread -> (slice, channel),
// 1024/cp slices split the block axis (cp = c rounded up to a power of two, coalesced over channels), then a
// fixed-order LDS reduction over the slices.  Requires c <= 1024.
__device__ __forceinline__ void reduce_partials(const float *__restrict__ partial, int nblocks, int c, int &ch,
                                                double &s, double &ss, double *lds /*[2][1024]*/) {
    int cp = 1;
    while (cp < c) cp <<= 1;
    const int slices = 1024 / cp;
    const int slice = threadIdx.x / cp;
    ch = threadIdx.x - slice * cp;
    double a = 0.0, b = 0.0;
    if (ch < c) {
        // 8 independent loads in flight per thread (the rows were just written: L2 hits, latency bound)
        int blk = slice;
        for (; blk + 7 * slices < nblocks; blk += 8 * slices) {
            float va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                va[u] = partial[((size_t)(blk + u * slices) * 2 + 0) * c + ch];
                vb[u] = partial[((size_t)(blk + u * slices) * 2 + 1) * c + ch];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a += (double)va[u];
                b += (double)vb[u];
            }
        }
        for (; blk < nblocks; blk += slices) {
            a += (double)partial[((size_t)blk * 2 + 0) * c + ch];
            b += (double)partial[((size_t)blk * 2 + 1) * c + ch];
        }
    }
    lds[threadIdx.x] = a;
    lds[1024 + threadIdx.x] = b;
    __syncthreads();
    s = 0.0;
    ss = 0.0;
    if (slice == 0) {
        for (int q = 0; q < slices; ++q) {
            s += lds[q * cp + ch];
            ss += lds[1024 + q * cp + ch];
        }
    }
    if (slice != 0) ch = c;  // only slice 0 carries the result
}

// Two-stage finalisation of the partial sums (used instead of the single-workgroup kernels above when the apply
// pass follows): MID_ROWS workgroups each sum an interleaved subset of the partial rows in double (coalesced over
// the 2c columns, fixed order) -> mid[MID_ROWS][2c]; every workgroup of the apply kernel then adds the MID_ROWS
// rows itself (16 loads per column, L2 hits) instead of waiting for one workgroup to chew through up to 5000 rows
// (conv tiles deliver one partial row each): 10-12 us -> ~3 us + a 1 us prologue.
constexpr int MID_ROWS = PCD_BN_MID_ROWS;   // (spconv.hip's fused form of this reduction relies on the value)

__global__ __launch_bounds__(1024) void bn_mid_kernel(const float *__restrict__ partial, int nblocks, int c,
                                                      double *__restrict__ mid) {
    __shared__ double lds[1024];
    const int cols = 2 * c;
    int cp = 1;
    while (cp < cols && cp < 1024) cp <<= 1;
    const int slices = 1024 / cp;                      // cols > 1024: one slice, columns looped
    const int slice = threadIdx.x / cp;
    for (int col0 = 0; col0 < cols; col0 += cp) {
        const int col = col0 + (threadIdx.x - slice * cp);
        double a = 0.0;
        if (col < cols) {
            // latency bound (a few dozen rows per slice): 8 independent loads in flight per thread
            int blk = blockIdx.x + MID_ROWS * slice;
            const int step = MID_ROWS * slices;
            for (; blk + 7 * step < nblocks; blk += 8 * step) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(blk + u * step) * cols + col];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += (double)v[u];
            }
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int bq = blk + u * step;
                v[u] = bq < nblocks ? partial[(size_t)bq * cols + col] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) a += (double)v[u];
        }
        lds[threadIdx.x] = a;
        __syncthreads();
        if (slice == 0 && col < cols) {
            double s = 0.0;
            for (int q = 0; q < slices; ++q) s += lds[q * cp + threadIdx.x];
            mid[(size_t)blockIdx.x * cols + col] = s;
        }
        __syncthreads();
    }
}

// tot[0..2c) = column totals of mid (every workgroup of an apply kernel; ends with a barrier)
__device__ __forceinline__ void mid_totals(const double *__restrict__ mid, int c, double *tot) {
    const int cols = 2 * c;
    for (int t = threadIdx.x; t < cols; t += blockDim.x) {
        double v[MID_ROWS];
#pragma unroll
        for (int r = 0; r < MID_ROWS; ++r) v[r] = mid[(size_t)r * cols + t];
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < MID_ROWS; ++r) s += v[r];
        tot[t] = s;
    }
    __syncthreads();
}
static size_t bn_mid_lds_bytes(int c) { return (size_t)2 * c * sizeof(double) + (size_t)2 * c * sizeof(float); }

// grid = 1, block = 1024: mean / invstd / running stats; scale/shift for the apply pass
__global__ __launch_bounds__(1024) void bn_finalize_kernel(
    const float *__restrict__ partial, int nblocks, int n_cap, const int32_t *n_dev, int c,
    const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float momentum,
    float *running_mean, float *running_var, float *save_mean, float *save_invstd, float *scale,
    float *shift) {
    __shared__ double lds[2 * 1024];
    const int n = eff_rows(n_dev, n_cap);
    int ch;
    double s, ss;
    reduce_partials(partial, nblocks, c, ch, s, ss, lds);
    if (ch >= c) return;
    double mean = n > 0 ? s / n : 0.0;
    double var = n > 0 ? ss / n - mean * mean : 0.0;
    if (var < 0.0) var = 0.0;
    float invstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[ch] = (float)mean;
    save_invstd[ch] = invstd;
    if (running_mean) running_mean[ch] = (1.0f - momentum) * running_mean[ch] + momentum * (float)mean;
    if (running_var) {
        double unbiased = n > 1 ? var * (double)n / (double)(n - 1) : var;
        running_var[ch] = (1.0f - momentum) * running_var[ch] + momentum * (float)unbiased;
    }
    float g = gamma ? gamma[ch] : 1.0f, b = beta ? beta[ch] : 0.0f;
    scale[ch] = g * invstd;
    shift[ch] = b - (float)mean * g * invstd;
}

__global__ void bn_eval_coeff_kernel(int c, const float *gamma, const float *beta, const float *rm,
                                     const float *rv, float eps, float *scale, float *shift) {
    int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    float invstd = 1.0f / sqrtf(rv[ch] + eps);
    float g = gamma ? gamma[ch] : 1.0f, b = beta ? beta[ch] : 0.0f;
    scale[ch] = g * invstd;
    shift[ch] = b - rm[ch] * g * invstd;
}

// training-mode finalisation done inside bn_apply_kernel (mid == nullptr: scale / shift come from memory)
struct BnFin {
    const double *mid;
    const float *gamma, *beta;
    float eps, momentum;
    float *running_mean, *running_var, *save_mean, *save_invstd;
};

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T *__restrict__ x, const T *__restrict__ res,
                                                       int n_cap, const int32_t *n_dev, int c,
                                                       const float *__restrict__ scale,
                                                       const float *__restrict__ shift, int relu,
                                                       T *__restrict__ y, int vec, BnFin fin, int y_ld) {
    __builtin_amdgcn_s_setprio(3);   // critical chain: ahead of the weight-gradient waves sharing the SIMD (spconv.hip)
    constexpr int N = Piece<T>::N;
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    const int pcs = c / N;
    const size_t total = (size_t)eff_rows(n_dev, n_cap) * pcs;
    const int piece = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) % pcs);  // fixed: strides are multiples of pcs
    float sc[N], sh[N];
    // The first BN_PF pieces of this thread are requested BEFORE the statistics prologue below (16 dependent L2 reads per
    // column + a barrier: ~2 us in which nothing streamed -- a fifth of the launch at 4 pieces per thread).
    const size_t e0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, S = (size_t)gridDim.x * blockDim.x;
    typename Piece<T>::Raw px[BN_PF], pr[BN_PF];
#pragma unroll
    for (int u = 0; u < BN_PF; ++u) {
        const size_t e = e0 + u * S < total ? e0 + u * S : 0;          // (beyond the end: any valid piece, unused)
        px[u] = Piece<T>::raw(x + e * N);
        if (res) pr[u] = Piece<T>::raw(res + e * N);
    }
    if (fin.mid) {
        // training: finish the batch statistics here (see bn_mid_kernel); workgroup 0 also publishes them
        double *tot = (double *)dyn_lds;
        float *sc_s = (float *)(tot + 2 * c), *sh_s = sc_s + c;
        mid_totals(fin.mid, c, tot);
        const int n = eff_rows(n_dev, n_cap);
        for (int ch = threadIdx.x; ch < c; ch += (int)blockDim.x) {
            const double s = tot[ch], ss = tot[c + ch];
            double mean = n > 0 ? s / n : 0.0;
            double var = n > 0 ? ss / n - mean * mean : 0.0;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)fin.eps));
            const float g = fin.gamma ? fin.gamma[ch] : 1.0f, b = fin.beta ? fin.beta[ch] : 0.0f;
            sc_s[ch] = g * invstd;
            sh_s[ch] = b - (float)mean * g * invstd;
            if (blockIdx.x == 0) {
                fin.save_mean[ch] = (float)mean;
                fin.save_invstd[ch] = invstd;
                if (fin.running_mean)
                    fin.running_mean[ch] = (1.0f - fin.momentum) * fin.running_mean[ch] + fin.momentum * (float)mean;
                if (fin.running_var) {
                    const double unbiased = n > 1 ? var * (double)n / (double)(n - 1) : var;
                    fin.running_var[ch] = (1.0f - fin.momentum) * fin.running_var[ch] + fin.momentum * (float)unbiased;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < N; ++j) {
            sc[j] = sc_s[piece * N + j];
            sh[j] = sh_s[piece * N + j];
        }
    } else {
        load_params<N>(scale, piece * N, vec, sc);
        load_params<N>(shift, piece * N, vec, sh);
    }
    // y may be a column block of a wider matrix (row stride y_ld elements): row = e / pcs advances by a whole number
    // of rows per grid stride
    size_t ye = (e0 / pcs) * (size_t)(y_ld / N) + piece;
    const size_t ystep = (S / pcs) * (size_t)(y_ld / N);
    auto finish = [&](const typename Piece<T>::Raw &xr, const typename Piece<T>::Raw &rr, size_t yat) {
        float v[N], r[N];
        Piece<T>::decode(xr, v);
        if (res) Piece<T>::decode(rr, r);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float o = v[j] * sc[j] + sh[j];
            if (res) o += r[j];
            if (relu) o = o > 0.0f ? o : 0.0f;
            v[j] = o;
        }
        Piece<T>::store(y + yat * N, v);
    };
    size_t e = e0;
#pragma unroll
    for (int u = 0; u < BN_PF; ++u, e += S, ye += ystep)
        if (e < total) finish(px[u], pr[u], ye);
    for (; e < total; e += S, ye += ystep) {
        const typename Piece<T>::Raw xr = Piece<T>::raw(x + e * N);
        typename Piece<T>::Raw rr = xr;
        if (res) rr = Piece<T>::raw(res + e * N);
        finish(xr, rr, ye);
    }
}

// ReLU mask: from the saved output y when it is given; with y == nullptr (no residual was added in the forward)
// from the sign of x * scale + shift recomputed exactly as bn_apply_kernel computed it.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                                                            const T *__restrict__ y, int n_cap,
                                                            const int32_t *n_dev, int c,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ invstd, int relu,
                                                            float *__restrict__ partial, int vec, int dy_ld) {
    constexpr int N = Piece<T>::N;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int pcs = c / N;
    const size_t total = (size_t)eff_rows(n_dev, n_cap) * pcs;
    const int piece = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) % pcs);
    float mu[N], is[N], sc[N], sh[N], gmv[N], btv[N];
    load_params<N>(mean, piece * N, vec, mu);
    load_params<N>(invstd, piece * N, vec, is);
    if (gamma) load_params<N>(gamma, piece * N, vec, gmv); else fill_params<N>(1.0f, gmv);
    if (beta) load_params<N>(beta, piece * N, vec, btv); else fill_params<N>(0.0f, btv);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        sc[j] = gmv[j] * is[j];
        sh[j] = btv[j] - mu[j] * gmv[j] * is[j];
    }
    const bool mask_from_x = relu && y == nullptr;
    float acc[2][N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[0][j] = acc[1][j] = 0.0f;
    auto accumulate = [&](const float (&g)[N], const float (&xv)[N], const float (&yv)[N]) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float t = mask_from_x ? xv[j] * sc[j] + sh[j] : yv[j];
            const float dz = (relu && !(t > 0.0f)) ? 0.0f : g[j];
            acc[0][j] += dz;
            acc[1][j] += dz * (xv[j] - mu[j]) * is[j];
        }
    };
    // two rows of loads (4-6 x 16 bytes) in flight per thread, same accumulation order as the plain loop
    const size_t S = (size_t)gridDim.x * blockDim.x;
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t de = (e / pcs) * (size_t)(dy_ld / N) + piece;               // dy: row stride dy_ld elements
    const size_t dstep = (S / pcs) * (size_t)(dy_ld / N);
    for (; e + S < total; e += 2 * S, de += 2 * dstep) {
        float g[2][N], xv[2][N], yv[2][N];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            Piece<T>::load(dy + (de + u * dstep) * N, g[u]);
            Piece<T>::load(x + (e + u * S) * N, xv[u]);
            if (relu && !mask_from_x) Piece<T>::load(y + (e + u * S) * N, yv[u]);
        }
        accumulate(g[0], xv[0], yv[0]);
        accumulate(g[1], xv[1], yv[1]);
    }
    for (; e < total; e += S, de += dstep) {
        float g[N], xv[N], yv[N];
        Piece<T>::load(dy + de * N, g);
        Piece<T>::load(x + e * N, xv);
        if (relu && !mask_from_x) Piece<T>::load(y + e * N, yv);
        accumulate(g, xv, yv);
    }
    block_reduce_store<N, 2>(acc, c, pcs, partial, lds);
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float *__restrict__ partial,
                                                               int nblocks, int c, float *dgamma,
                                                               float *dbeta) {
    __shared__ double lds[2 * 1024];
    int ch;
    double s, ss;
    reduce_partials(partial, nblocks, c, ch, s, ss, lds);
    if (ch >= c) return;
    dbeta[ch] = (float)s;
    dgamma[ch] = (float)ss;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                                                           const T *__restrict__ y, int n_cap,
                                                           const int32_t *n_dev, int c,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ beta,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ invstd,
                                                           float *__restrict__ dgamma,
                                                           float *__restrict__ dbeta, int relu,
                                                           int training, T *__restrict__ dx,
                                                           T *__restrict__ dres, int vec, const double *mid,
                                                           float *__restrict__ colsum_partial, int dy_ld) {
    __builtin_amdgcn_s_setprio(3);   // critical chain: ahead of the weight-gradient waves sharing the SIMD (spconv.hip)
    constexpr int N = Piece<T>::N;
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    const int pcs = c / N;
    const int n = eff_rows(n_dev, n_cap);
    const size_t total = (size_t)n * pcs;
    const int piece = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) % pcs);
    float mu[N], is[N], gm[N], sh[N], k1[N], k2[N], gmv[N], btv[N];
    const float inv_n = n > 0 ? 1.0f / (float)n : 0.0f;
    const bool mask_from_x = relu && y == nullptr;
    // the first BN_PF pieces of this thread are requested BEFORE the reduction prologue below (see bn_apply_kernel)
    const size_t e0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, S = (size_t)gridDim.x * blockDim.x;
    size_t de = (e0 / pcs) * (size_t)(dy_ld / N) + piece;              // dy: row stride dy_ld elements
    const size_t dstep = (S / pcs) * (size_t)(dy_ld / N);
    typename Piece<T>::Raw pg[BN_PF], px[BN_PF], py[BN_PF];
#pragma unroll
    for (int u = 0; u < BN_PF; ++u) {
        const bool in = e0 + u * S < total;
        pg[u] = Piece<T>::raw(dy + (in ? de + u * dstep : 0) * N);
        px[u] = Piece<T>::raw(x + (in ? e0 + u * S : 0) * N);
        if (relu && !mask_from_x) py[u] = Piece<T>::raw(y + (in ? e0 + u * S : 0) * N);
    }
    load_params<N>(mean, piece * N, vec, mu);
    load_params<N>(invstd, piece * N, vec, is);
    if (gamma) load_params<N>(gamma, piece * N, vec, gmv); else fill_params<N>(1.0f, gmv);
    if (beta) load_params<N>(beta, piece * N, vec, btv); else fill_params<N>(0.0f, btv);
    if (training && mid) {
        // finish the two reductions here (see bn_mid_kernel); workgroup 0 publishes dgamma / dbeta
        double *tot = (double *)dyn_lds;
        float *fs = (float *)(tot + 2 * c);
        mid_totals(mid, c, tot);
        for (int t = threadIdx.x; t < 2 * c; t += (int)blockDim.x) {
            const float v = (float)tot[t];
            fs[t] = v;
            if (blockIdx.x == 0) (t < c ? dbeta[t] : dgamma[t - c]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < N; ++j) {
            k1[j] = fs[piece * N + j];
            k2[j] = fs[c + piece * N + j];
        }
    } else if (training) {
        load_params<N>(dbeta, piece * N, vec, k1);
        load_params<N>(dgamma, piece * N, vec, k2);
    } else {
        fill_params<N>(0.0f, k1);
        fill_params<N>(0.0f, k2);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        gm[j] = gmv[j] * is[j];
        sh[j] = btv[j] - mu[j] * gmv[j] * is[j];
        k1[j] *= inv_n;
        k2[j] *= inv_n;
    }
    // colsum_partial: column sums of dx AS STORED -- dx is dy of the conv in front of this BatchNorm and its column
    // sum that conv's bias gradient (replaces a separate pass over dx; one partial row per workgroup)
    float cs[1][N];
#pragma unroll
    for (int j = 0; j < N; ++j) cs[0][j] = 0.0f;
    auto finish = [&](const typename Piece<T>::Raw &gr, const typename Piece<T>::Raw &xr, const typename Piece<T>::Raw &yr,
                      size_t e) {
        float g[N], xv[N], yv[N], o[N];
        Piece<T>::decode(gr, g);
        Piece<T>::decode(xr, xv);
        if (relu && !mask_from_x) Piece<T>::decode(yr, yv);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float t = mask_from_x ? xv[j] * gm[j] + sh[j] : yv[j];
            float dz = (relu && !(t > 0.0f)) ? 0.0f : g[j];
            g[j] = dz;
            float xhat = (xv[j] - mu[j]) * is[j];
            o[j] = gm[j] * (dz - k1[j] - xhat * k2[j]);
            cs[0][j] += Piece<T>::stored(o[j]);
        }
        Piece<T>::store(dx + e * N, o);
        if (dres) Piece<T>::store(dres + e * N, g);
    };
    size_t e = e0;
#pragma unroll
    for (int u = 0; u < BN_PF; ++u, e += S, de += dstep)
        if (e < total) finish(pg[u], px[u], py[u], e);
    for (; e < total; e += S, de += dstep) {
        const typename Piece<T>::Raw gr = Piece<T>::raw(dy + de * N), xr = Piece<T>::raw(x + e * N);
        typename Piece<T>::Raw yr = xr;
        if (relu && !mask_from_x) yr = Piece<T>::raw(y + e * N);
        finish(gr, xr, yr, e);
    }
    if (colsum_partial) {
        __syncthreads();      // the prologue's use of the dynamic LDS is over
        block_reduce_store<N, 1>(cs, c, pcs, colsum_partial, (float *)dyn_lds);
    }
}

// out[ch] = sum over the rows of partial[nblocks][c] (fixed order): finishes the column sums bn_bwd_apply_kernel took.
// One workgroup per job; the jobs ride in the kernel arguments (no device-side table to upload, graph-capturable).
struct ColJobs {
    PcdColsumJob job[PCD_COLSUM_MAX_JOBS];
};
__global__ __launch_bounds__(1024) void col_rows_finalize_kernel(ColJobs jobs) {
    __shared__ double lds[1024];
    const float *__restrict__ partial = jobs.job[blockIdx.x].partial;
    const int nblocks = jobs.job[blockIdx.x].rows, c = jobs.job[blockIdx.x].c;
    float *out = jobs.job[blockIdx.x].out;
    int cp = 1;
    while (cp < c) cp <<= 1;
    const int slices = 1024 / cp;
    const int slice = threadIdx.x / cp, ch = threadIdx.x - slice * cp;
    double a = 0.0;
    if (ch < c) {
        int blk = slice;
        // (a chain of load latencies on the main stream at the end of the backward pass: 16 loads in flight, same order of sums)
        for (; blk + 15 * slices < nblocks; blk += 16 * slices) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = partial[(size_t)(blk + u * slices) * c + ch];
#pragma unroll
            for (int u = 0; u < 16; ++u) a += (double)v[u];
        }
        for (; blk + 7 * slices < nblocks; blk += 8 * slices) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(blk + u * slices) * c + ch];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += (double)v[u];
        }
        for (; blk < nblocks; blk += slices) a += (double)partial[(size_t)blk * c + ch];
    }
    lds[threadIdx.x] = a;
    __syncthreads();
    if (slice == 0 && ch < c) {
        double s = 0.0;
        for (int q = 0; q < slices; ++q) s += lds[q * cp + ch];
        out[ch] = (float)s;
    }
}

// Reduction passes ALWAYS use MAX_BLOCKS workgroups: which elements a (block, thread) accumulates then depends
// only on the real row count (host n or *n_dev) and never on the capacity the buffers were allocated with, so
// eager and static-shape (hipGraph) execution produce bit-identical statistics.
static int grid_for(size_t pieces, int pcs, int max_blocks = MAX_BLOCKS) {
    (void)pcs;
    if (max_blocks == MAX_BLOCKS) return MAX_BLOCKS;
#ifndef BN_PPT
#define BN_PPT 4
#endif
    size_t blocks = (pieces + 256 * BN_PPT - 1) / (256 * BN_PPT);  // streaming apply passes: >= BN_PPT pieces per thread
    if (blocks > (size_t)max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// threads per block of the streaming passes: a multiple of the pieces per row, so that a thread keeps ONE piece
// (column group) over its whole grid-stride loop
static int bn_threads(int pcs) { return 256 / pcs * pcs; }

static bool shape_ok(int c, int dtype) {
    int N = dtype == PCD_F32 ? 4 : 8;
    if (c % N) return false;
    int pcs = c / N;
    return pcs >= 1 && pcs <= 256 && c <= 1024;   // (256 % pcs != 0: blocks of (256 / pcs) * pcs threads, see bn_threads)
}

struct BnWs {
    float *partial, *scale, *shift;
    double *mid;
};
static bool bn_ws(void *ws, size_t bytes, int c, BnWs &L) {
    WsCarver w(ws, bytes);
    L.partial = w.take<float>((size_t)MAX_BLOCKS * 2 * c);
    L.scale = w.take<float>(c);
    L.shift = w.take<float>(c);
    L.mid = w.take<double>((size_t)MID_ROWS * 2 * c);
    return w.ok;
}

}  // namespace

extern "C" size_t pcd_bn_workspace_bytes(int c) {
    if (c <= 0) return 0;
    return ws_piece((size_t)MAX_BLOCKS * 2 * c, sizeof(float)) + 2 * ws_piece(c, sizeof(float)) +
           ws_piece((size_t)MID_ROWS * 2 * c, sizeof(double));
}

extern "C" int pcd_bn_forward_ld(const void *x, const void *residual, int dtype, int n, int c,
                                 const float *gamma, const float *beta, float eps, float momentum,
                                 int training, float *running_mean, float *running_var, int relu, void *y, int y_ld,
                                 float *save_mean, float *save_invstd, const int32_t *n_dev,
                                 const float *ext_partial, int ext_rows, void *workspace, size_t workspace_bytes,
                                 void *stream) {
    PCD_ENTER();
    if (n < 0 || c <= 0 || (dtype != PCD_F32 && dtype != PCD_BF16)) return PCD_ERR_INVALID_ARG;
    if (y_ld < c || y_ld % (dtype == PCD_F32 ? 4 : 8)) return PCD_ERR_INVALID_ARG;
    if (!shape_ok(c, dtype)) return PCD_ERR_UNSUPPORTED;
    if (!training && (!running_mean || !running_var)) return PCD_ERR_INVALID_ARG;
    if (training && (!save_mean || !save_invstd)) return PCD_ERR_INVALID_ARG;
    if (n > 0 && (!x || !y)) return PCD_ERR_INVALID_ARG;
    BnWs L;
    if (!bn_ws(workspace, workspace_bytes, c, L)) return PCD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int N = dtype == PCD_F32 ? 4 : 8;
    const int pcs = c / N;
    int grid = grid_for((size_t)n * pcs, pcs);
    int agrid = grid_for((size_t)n * pcs, pcs, MAX_APPLY_BLOCKS);
    const bool ext_mid = ext_partial && ext_rows == PCD_BN_EXT_MID;   // the conv launch folded its rows already
    if (ext_partial && ext_rows < 0 && !ext_mid) return PCD_ERR_INVALID_ARG;
    if (ext_mid && !(training && n > 0)) return PCD_ERR_INVALID_ARG;
    if (ext_mid) L.mid = (double *)ext_partial;
    if (training) {
        if (ext_partial)   // the conv epilogue already took the sums (PcdBnReduce mode 1)
            ;
        else if (dtype == PCD_F32)
            bn_stats_kernel<float><<<grid, bn_threads(c / N), bn_reduce_lds_bytes(c, 4), st>>>((const float *)x, n, n_dev, c, L.partial);
        else
            bn_stats_kernel<unsigned short><<<grid, bn_threads(c / N), bn_reduce_lds_bytes(c, 8), st>>>((const unsigned short *)x, n, n_dev, c,
                                                                  L.partial);
        const float *part = ext_partial ? ext_partial : L.partial;
        const int prow = ext_partial ? ext_rows : grid;
        if (ext_mid)
            ;
        else if (n > 0)   // two-stage: the apply kernel finishes the statistics itself
            bn_mid_kernel<<<MID_ROWS, 1024, 0, st>>>(part, prow, c, L.mid);
        else
            bn_finalize_kernel<<<1, 1024, 0, st>>>(part, prow, n, n_dev, c, gamma, beta, eps, momentum, running_mean,
                                                   running_var, save_mean, save_invstd, L.scale, L.shift);
    } else {
        bn_eval_coeff_kernel<<<pcd_div_up(c, 128), 128, 0, st>>>(c, gamma, beta, running_mean, running_var,
                                                                 eps, L.scale, L.shift);
    }
    if (n > 0) {
        BnFin fin = {};
        if (training) {
            fin.mid = L.mid;
            fin.gamma = gamma;
            fin.beta = beta;
            fin.eps = eps;
            fin.momentum = momentum;
            fin.running_mean = running_mean;
            fin.running_var = running_var;
            fin.save_mean = save_mean;
            fin.save_invstd = save_invstd;
        }
        const size_t lds = training ? bn_mid_lds_bytes(c) : 0;
        if (dtype == PCD_F32)
            bn_apply_kernel<float><<<agrid, bn_threads(pcs), lds, st>>>((const float *)x, (const float *)residual, n, n_dev, c,
                                                           L.scale, L.shift, relu, (float *)y, 1, fin, y_ld);
        else
            bn_apply_kernel<unsigned short><<<agrid, bn_threads(pcs), lds, st>>>(
                (const unsigned short *)x, (const unsigned short *)residual, n, n_dev, c, L.scale, L.shift, relu,
                (unsigned short *)y, 1, fin, y_ld);
    }
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_bn_forward(const void *x, const void *residual, int dtype, int n, int c,
                              const float *gamma, const float *beta, float eps, float momentum,
                              int training, float *running_mean, float *running_var, int relu, void *y,
                              float *save_mean, float *save_invstd, const int32_t *n_dev,
                              const float *ext_partial, int ext_rows, void *workspace, size_t workspace_bytes,
                              void *stream) {
    return pcd_bn_forward_ld(x, residual, dtype, n, c, gamma, beta, eps, momentum, training, running_mean, running_var,
                             relu, y, c, save_mean, save_invstd, n_dev, ext_partial, ext_rows, workspace,
                             workspace_bytes, stream);
}

__global__ __launch_bounds__(1024) void col_sum_finalize_kernel(const float *__restrict__ partial, int nblocks,
                                                                int c, float *out) {
    __shared__ double lds[2 * 1024];
    int ch;
    double s, ss;
    reduce_partials(partial, nblocks, c, ch, s, ss, lds);
    if (ch >= c) return;
    out[ch] = (float)s;
}

extern "C" int pcd_col_sum(const void *x, int dtype, int n, int c, float *out, const int32_t *n_dev,
                           void *workspace, size_t workspace_bytes, void *stream) {
    PCD_ENTER();
    if (n < 0 || c <= 0 || !out || (dtype != PCD_F32 && dtype != PCD_BF16)) return PCD_ERR_INVALID_ARG;
    if (!shape_ok(c, dtype)) return PCD_ERR_UNSUPPORTED;
    if (n > 0 && !x) return PCD_ERR_INVALID_ARG;
    BnWs L;
    if (!bn_ws(workspace, workspace_bytes, c, L)) return PCD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int N = dtype == PCD_F32 ? 4 : 8;
    int grid = grid_for((size_t)n * (c / N), c / N);
    if (dtype == PCD_F32)
        bn_stats_kernel<float><<<grid, bn_threads(c / N), bn_reduce_lds_bytes(c, 4), st>>>((const float *)x, n, n_dev, c, L.partial);
    else
        bn_stats_kernel<unsigned short><<<grid, bn_threads(c / N), bn_reduce_lds_bytes(c, 8), st>>>((const unsigned short *)x, n, n_dev, c, L.partial);
    col_sum_finalize_kernel<<<1, 1024, 0, st>>>(L.partial, grid, c, out);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_bn_backward_ld(const void *dy, int dy_ld, const void *x, const void *y, int dtype, int n, int c,
                                  const float *gamma, const float *beta, const float *save_mean,
                                  const float *save_invstd,
                                  int relu, int training, void *dx, void *dresidual, float *dgamma,
                                  float *dbeta, const int32_t *n_dev, const float *ext_partial, int ext_rows,
                                  float *colsum_partial, void *workspace, size_t workspace_bytes, void *stream) {
    PCD_ENTER();
    if (n < 0 || c <= 0 || (dtype != PCD_F32 && dtype != PCD_BF16)) return PCD_ERR_INVALID_ARG;
    if (dy_ld < c || dy_ld % (dtype == PCD_F32 ? 4 : 8)) return PCD_ERR_INVALID_ARG;
    if (!shape_ok(c, dtype)) return PCD_ERR_UNSUPPORTED;
    if (!save_mean || !save_invstd || !dgamma || !dbeta) return PCD_ERR_INVALID_ARG;
    if (n > 0 && (!dy || !x || !dx)) return PCD_ERR_INVALID_ARG;
    BnWs L;
    if (!bn_ws(workspace, workspace_bytes, c, L)) return PCD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int N = dtype == PCD_F32 ? 4 : 8;
    const int pcs = c / N;
    int grid = grid_for((size_t)n * pcs, pcs);
    int agrid = grid_for((size_t)n * pcs, pcs, MAX_APPLY_BLOCKS);
    const int vec = aligned16(gamma, beta, save_mean, save_invstd, dgamma, dbeta) ? 1 : 0;
    const bool ext_mid = ext_partial && ext_rows == PCD_BN_EXT_MID;   // the dgrad launch folded its rows already
    if (ext_partial && ext_rows < 0 && !ext_mid) return PCD_ERR_INVALID_ARG;
    if (ext_mid && !(training && n > 0)) return PCD_ERR_INVALID_ARG;
    if (ext_mid) L.mid = (double *)ext_partial;
    const float *part = ext_partial ? ext_partial : L.partial;     // PcdBnReduce mode 2: sums taken by the dgrad
    const int prow = ext_partial ? ext_rows : grid;
    const bool two_stage = training && n > 0;      // the apply kernel finishes the reductions itself
    const double *mid = two_stage ? L.mid : nullptr;
    size_t alds = two_stage ? bn_mid_lds_bytes(c) : 0;
    if (colsum_partial && alds < bn_reduce_lds_bytes(c, N)) alds = bn_reduce_lds_bytes(c, N);
    auto finalize = [&]() {
        if (ext_mid)
            ;
        else if (two_stage)
            bn_mid_kernel<<<MID_ROWS, 1024, 0, st>>>(part, prow, c, L.mid);
        else
            bn_bwd_finalize_kernel<<<1, 1024, 0, st>>>(part, prow, c, dgamma, dbeta);
    };
    if (dtype == PCD_F32) {
        if (!ext_partial)
            bn_bwd_reduce_kernel<float><<<grid, bn_threads(pcs), bn_reduce_lds_bytes(c, 4), st>>>((const float *)dy, (const float *)x,
                                                          (const float *)y, n, n_dev, c, gamma, beta, save_mean,
                                                          save_invstd, relu, L.partial, vec, dy_ld);
        finalize();
        if (n > 0)
            bn_bwd_apply_kernel<float><<<agrid, bn_threads(pcs), alds, st>>>(
                (const float *)dy, (const float *)x, (const float *)y, n, n_dev, c, gamma, beta, save_mean,
                save_invstd, dgamma, dbeta, relu, training, (float *)dx, (float *)dresidual, vec, mid, colsum_partial,
                dy_ld);
    } else {
        typedef unsigned short B;
        if (!ext_partial)
            bn_bwd_reduce_kernel<B><<<grid, bn_threads(pcs), bn_reduce_lds_bytes(c, 8), st>>>((const B *)dy, (const B *)x, (const B *)y, n, n_dev, c,
                                                      gamma, beta, save_mean, save_invstd, relu, L.partial, vec, dy_ld);
        finalize();
        if (n > 0)
            bn_bwd_apply_kernel<B><<<agrid, bn_threads(pcs), alds, st>>>((const B *)dy, (const B *)x, (const B *)y, n, n_dev, c,
                                                            gamma, beta, save_mean, save_invstd, dgamma, dbeta,
                                                            relu, training, (B *)dx, (B *)dresidual, vec, mid, colsum_partial,
                                                            dy_ld);
    }
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_bn_backward(const void *dy, const void *x, const void *y, int dtype, int n, int c,
                               const float *gamma, const float *beta, const float *save_mean,
                               const float *save_invstd,
                               int relu, int training, void *dx, void *dresidual, float *dgamma,
                               float *dbeta, const int32_t *n_dev, const float *ext_partial, int ext_rows,
                               float *colsum_partial, void *workspace, size_t workspace_bytes, void *stream) {
    return pcd_bn_backward_ld(dy, c, x, y, dtype, n, c, gamma, beta, save_mean, save_invstd, relu, training, dx, dresidual,
                              dgamma, dbeta, n_dev, ext_partial, ext_rows, colsum_partial, workspace, workspace_bytes,
                              stream);
}

// rows of the colsum_partial buffer pcd_bn_backward fills ([rows][c] f32): the grid of its apply pass
extern "C" int pcd_bn_backward_colsum_rows(int dtype, int n, int c) {
    if (n < 0 || c <= 0 || (dtype != PCD_F32 && dtype != PCD_BF16)) return PCD_ERR_INVALID_ARG;
    if (!shape_ok(c, dtype)) return PCD_ERR_UNSUPPORTED;
    const int pcs = c / (dtype == PCD_F32 ? 4 : 8);
    return n > 0 ? grid_for((size_t)n * pcs, pcs, MAX_APPLY_BLOCKS) : 0;
}

extern "C" int pcd_col_sum_finalize(const PcdColsumJob *jobs_host, int n_jobs, void *stream) {
    PCD_ENTER();
    if (n_jobs < 0 || n_jobs > PCD_COLSUM_MAX_JOBS || (n_jobs > 0 && !jobs_host)) return PCD_ERR_INVALID_ARG;
    if (n_jobs == 0) return PCD_OK;
    ColJobs J = {};
    for (int i = 0; i < n_jobs; ++i) {
        const PcdColsumJob &j = jobs_host[i];
        if (j.rows < 0 || j.c <= 0 || j.c > 1024 || !j.out || (j.rows > 0 && !j.partial)) return PCD_ERR_INVALID_ARG;
        J.job[i] = j;
    }
    col_rows_finalize_kernel<<<n_jobs, 1024, 0, (hipStream_t)stream>>>(J);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
