// fp8 (OCP e4m3) feature path of the sparse convolution for gfx950 -- BASELINE config 5 ("SECOND/VoxelNet fp8 features
// on CDNA4 fp8 MFMA, 300k-pt dense clouds"; the reference itself is fp32: this is a build-side precision, SURVEY.md
// section 7 step 9).  INFERENCE form of a post_act_block (pcdet/models/backbones_3d/spconv_backbone.py:8-27):
//     y = quant( relu( conv(x8, w8) * alpha[c] + beta[c] ) )
// with x8 / w8 in e4m3, per-tensor scales folded together with the eval-mode BatchNorm into the per-channel
// alpha / beta, fp32 accumulation in v_mfma_f32_16x16x32_fp8_fp8, and the next layer's quantisation in the
// epilogue: ONE kernel per conv + BatchNorm + ReLU, and half the gathered bytes of the bf16 path (a 128-channel
// row is ONE 128-byte line) -- these kernels are priced per gathered line / per vector-memory instruction.
//
// Structure = the output-stationary gather-GEMM of spconv.hip: a wave owns MI x 16 output rows and all output
// channels; a lane's 16-byte gather now carries 16 fp8 channels and feeds TWO MFMAs (contraction step = 64
// channels); packed weights stream through a double-buffered LDS stage shared by the 4 waves.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__host__ __device__ constexpr int f8_quad(int nb) { return nb % 4 == 0 ? 4 : (nb % 2 == 0 ? 2 : 1); }

__device__ __forceinline__ float sat448(float v) { return __builtin_amdgcn_fmed3f(v, -448.0f, 448.0f); }

__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
    // v_cvt_pk_fp8_f32: OCP e4m3 on gfx950, round to nearest even; out-of-range inputs convert to NaN unless the
    // FP8 overflow mode bit is set, so saturate explicitly (e4m3fn has no infinity: |x| > 448 -> +-448)
    a = sat448(a); b = sat448(b); c = sat448(c); d = sat448(d);
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}

// weight [Cout][K][Cin] f32 -> e4m3 fragments: element (dstep, nb, lane, byte) -- see the header comment of
// pcd_fp8_pack_weight for the index maps
__global__ __launch_bounds__(256) void fp8_pack_weight_kernel(const float *__restrict__ w, int K, int cin, int cout,
                                                              int cshift, int NB, float scale_inv, size_t total_words,
                                                              unsigned *__restrict__ out) {
    const size_t wd = (size_t)blockIdx.x * 256 + threadIdx.x;       // one 32-bit word = 4 consecutive bytes
    if (wd >= total_words) return;
    const int j4 = (int)(wd & 3) * 4;                                // first byte of the word inside the lane's 16
    const int lane = (int)((wd >> 2) & 63);
    const size_t t = wd >> 8;
    const int nb = (int)(t % NB);
    const int s = (int)(t / NB);
    const int Q = f8_quad(NB), m = lane & 15, g = lane >> 4;
    const int col = (nb / Q) * 16 * Q + (m >> 2) * 4 * Q + (nb % Q) * 4 + (m & 3);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = s * 64 + g * 16 + j4 + j;
        const int k = q >> cshift, c = q & ((1 << cshift) - 1);
        v[j] = (k < K && c < cin && col < cout) ? w[((size_t)col * K + k) * cin + c] * scale_inv : 0.0f;
    }
    out[wd] = pack4_fp8(v[0], v[1], v[2], v[3]);
}

// x [n][c] (f32 or bf16, row stride c_stride) -> e4m3 [n][cb], zero padded channels; one thread per 4 output bytes
template <typename T>
__global__ __launch_bounds__(256) void fp8_quantize_kernel(const T *__restrict__ x, int n_cap, const int32_t *n_dev, int c,
                                                           int c_stride, int cb, float scale_inv,
                                                           unsigned *__restrict__ out) {
    const int n = eff_rows(n_dev, n_cap);
    const size_t wd = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int wpr = cb / 4;
    if (wd >= (size_t)n * wpr) return;
    const int r = (int)(wd / wpr), c0 = (int)(wd - (size_t)r * wpr) * 4;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float f = 0.0f;
        if (c0 + j < c) {
            if (sizeof(T) == 2)
                f = bf16_bits_to_f32(((const unsigned short *)x)[(size_t)r * c_stride + c0 + j]);
            else
                f = ((const float *)x)[(size_t)r * c_stride + c0 + j];
        }
        v[j] = f * scale_inv;
    }
    out[wd] = pack4_fp8(v[0], v[1], v[2], v[3]);
}

__global__ __launch_bounds__(256) void fp8_dequantize_kernel(const unsigned char *__restrict__ x, size_t n, float scale,
                                                             float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_f32_fp8((int)x[i], 0) * scale;
}

enum { F8_OUT_F32 = 0, F8_OUT_BF16 = 1, F8_OUT_FP8 = 2 };

template <int NB, int MI, int OUT>
__global__ __launch_bounds__(256) void gg_fp8_kernel(const unsigned char *__restrict__ x, int cshift,
                                                     const uint4 *__restrict__ wp, const int32_t *__restrict__ nbr,
                                                     int nbr_stride, int K, int flip, int n_out_cap,
                                                     const int32_t *__restrict__ n_out_dev,
                                                     const float *__restrict__ alpha, const float *__restrict__ beta,
                                                     int relu, float out_scale_inv, void *__restrict__ yv, int y_stride,
                                                     int nsteps, unsigned x_bytes) {
    constexpr int ROWS = 4 * MI * 16;
    constexpr int VEC = NB * 64;                            // uint4 per weight stage (one 64-channel step)
    constexpr int WPT = (VEC + 255) / 256;
    constexpr int Q = f8_quad(NB);
    const int n_out = eff_rows(n_out_dev, n_out_cap);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4 *wbuf = (uint4 *)smem;                            // [2][VEC]
    int *nbr_s = (int *)(smem + (size_t)2 * VEC * sizeof(uint4));   // [K + 1][ROWS]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rl = lane & 15, g = lane >> 4;
    // contiguous runs of the REAL tiles per XCD (see xcd_tile in spconv.hip)
    const int nt = (n_out + ROWS - 1) / ROWS;
    const int tpx = min((nt + 7) >> 3, (int)(gridDim.x >> 3));
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tile = jb < tpx ? xcd * tpx + jb : 8 * tpx + (jb - tpx) * 8 + xcd;
    const int r0wg = tile * ROWS;
    if (r0wg >= n_out) return;
    const __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)nbr, 0, (int)((unsigned)K * (unsigned)nbr_stride * 4u), 0x00020000);
    {
        const int total = K * ROWS;
        for (int base = threadIdx.x; base < total + ROWS; base += 4 * 256) {
            int v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256;
                const int k = idx / ROWS, r = idx - k * ROWS;
                const int row = r0wg + r;
                const int krow = flip ? (K - 1 - k) : k;
                const bool ok = idx < total && row < n_out;
                v[u] = __builtin_amdgcn_raw_buffer_load_b32(
                    nrsrc, ok ? ((unsigned)krow * (unsigned)nbr_stride + (unsigned)row) * 4u : 0xFFFFFFF0u, 0, 0);
                if (!ok) v[u] = -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256;
                if (idx < total + ROWS) nbr_s[idx] = v[u];
            }
        }
    }
    const unsigned wtotal = (unsigned)nsteps * VEC;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, (int)(wtotal * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    u32x4 wreg[WPT];
    auto load_w = [&](int s) {
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const unsigned e = (unsigned)(j * 256) + threadIdx.x;
            wreg[j] = __builtin_amdgcn_raw_buffer_load_b128(
                wrsrc, e < (unsigned)VEC ? ((unsigned)s * VEC + e) * 16u : 0xFFFFFFF0u, 0, 0);
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const unsigned e = (unsigned)(j * 256) + threadIdx.x;
            if (e < (unsigned)VEC) reinterpret_cast<u32x4 *>(wbuf)[buf * VEC + e] = wreg[j];
        }
    };
    load_w(0);
    store_w(0);
    __syncthreads();

    f32x4 acc[MI][NB];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mi][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tile_row = wave * (MI * 16) + rl;
    const int cmask = (1 << cshift) - 1;
    auto gather = [&](int s, u32x4(&a)[MI], bool &valid) {
        const int q0 = s * 64 + g * 16;
        int k = q0 >> cshift;
        k = k < K ? k : K;
        const unsigned c0 = (unsigned)(q0 & cmask);
        valid = false;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int i = nbr_s[k * ROWS + tile_row + mi * 16];
            a[mi] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ((unsigned)i << cshift) + c0, 0, 0);   // -1 -> zeros
            valid |= (i >= 0);
        }
    };
    auto compute = [&](int cur, const u32x4(&a)[MI], bool valid) {
        if (__any(valid)) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const uint4 w = wbuf[cur * VEC + nb * 64 + lane];
                const long w0 = (long)(((unsigned long long)w.y << 32) | w.x), w1 = (long)(((unsigned long long)w.w << 32) | w.z);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const long a0 = (long)(((unsigned long long)a[mi][1] << 32) | a[mi][0]);
                    const long a1 = (long)(((unsigned long long)a[mi][3] << 32) | a[mi][2]);
                    acc[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w0, a0, acc[mi][nb], 0, 0, 0);
                    acc[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w1, a1, acc[mi][nb], 0, 0, 0);
                }
            }
        }
    };
    u32x4 a0[MI], a1[MI];
    bool v0, v1;
    gather(0, a0, v0);
    for (int s = 0; s < nsteps; s += 2) {
        load_w(s + 1);
        gather(s + 1, a1, v1);
        compute(0, a0, v0);
        store_w(1);
        __syncthreads();
        load_w(s + 2);
        gather(s + 2, a0, v0);
        compute(1, a1, v1);                                  // step nsteps (odd count): all -1 -> skipped
        store_w(0);
        __syncthreads();
    }

    // epilogue: y = quant(relu(acc * alpha + beta)); lane (g, rl) owns 4 Q consecutive channels of its row per
    // interleave group (the pack's channel order), i.e. 16 bytes of fp8 at Q = 4
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = r0wg + tile_row + mi * 16;
        if (row >= n_out) continue;
#pragma unroll
        for (int qd = 0; qd < NB / Q; ++qd) {
            const int col = qd * 16 * Q + g * 4 * Q;
            float v[4 * Q];
#pragma unroll
            for (int j = 0; j < 4 * Q; ++j) {
                float t = acc[mi][qd * Q + j / 4][j % 4] * alpha[col + j] + beta[col + j];
                v[j] = relu ? fmaxf(t, 0.0f) : t;
            }
            const size_t at = (size_t)row * y_stride + col;
            if (OUT == F8_OUT_FP8) {
                unsigned o[Q];
#pragma unroll
                for (int u = 0; u < Q; ++u)
                    o[u] = pack4_fp8(v[4 * u] * out_scale_inv, v[4 * u + 1] * out_scale_inv, v[4 * u + 2] * out_scale_inv,
                                     v[4 * u + 3] * out_scale_inv);
                unsigned *y = (unsigned *)((unsigned char *)yv + at);
#pragma unroll
                for (int u = 0; u < Q; ++u) y[u] = o[u];
            } else if (OUT == F8_OUT_BF16) {
                unsigned short *y = (unsigned short *)yv + at;
#pragma unroll
                for (int j = 0; j < 4 * Q; ++j) y[j] = f32_to_bf16_bits(v[j]);
            } else {
                float *y = (float *)yv + at;
#pragma unroll
                for (int j = 0; j < 4 * Q; ++j) y[j] = v[j];
            }
        }
    }
}

static int log2_exact_f8(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return (1 << s) == v ? s : -1;
}

template <int NB, int MI>
static int launch_f8(const void *x, int cshift, const void *wp, const int32_t *nbr, int nbr_stride, int K, int flip,
                     int n_out, const int32_t *n_out_dev, const float *alpha, const float *beta, int relu,
                     float out_scale_inv, void *y, int y_kind, int y_stride, int nsteps, unsigned x_bytes,
                     hipStream_t st) {
    constexpr int ROWS = 4 * MI * 16;
    const int grid = pcd_div_up(pcd_div_up(n_out, ROWS), 8) * 8;
    const size_t lds = (size_t)2 * NB * 64 * sizeof(uint4) + (size_t)(K + 1) * ROWS * sizeof(int);
    if (lds > 64 * 1024) return PCD_ERR_UNSUPPORTED;
#define F8_GO(OUTK)                                                                                              \
    gg_fp8_kernel<NB, MI, OUTK><<<grid, 256, lds, st>>>((const unsigned char *)x, cshift, (const uint4 *)wp, nbr, \
                                                         nbr_stride, K, flip, n_out, n_out_dev, alpha, beta, relu, \
                                                         out_scale_inv, y, y_stride, nsteps, x_bytes)
    if (y_kind == F8_OUT_FP8) F8_GO(F8_OUT_FP8);
    else if (y_kind == F8_OUT_BF16) F8_GO(F8_OUT_BF16);
    else F8_GO(F8_OUT_F32);
#undef F8_GO
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

}  // namespace

extern "C" size_t pcd_fp8_packed_weight_bytes(int kvol, int cin_pad, int cout) {
    if (kvol <= 0 || cin_pad < 16 || log2_exact_f8(cin_pad) < 0 || cout <= 0) return 0;
    const size_t nsteps = ((size_t)kvol * cin_pad + 63) / 64;
    return nsteps * (size_t)((cout + 15) / 16) * 64 * 16;
}

extern "C" int pcd_fp8_pack_weight(const float *weight, int kvol, int cin, int cin_pad, int cout, float scale_inv,
                                   void *packed, void *stream) {
    PCD_ENTER();
    const int cshift = log2_exact_f8(cin_pad);
    if (!weight || !packed || kvol <= 0 || cin <= 0 || cin > cin_pad || cin_pad < 16 || cshift < 0 || cout <= 0)
        return PCD_ERR_INVALID_ARG;
    const int NB = (cout + 15) / 16;
    const size_t words = pcd_fp8_packed_weight_bytes(kvol, cin_pad, cout) / 4;
    fp8_pack_weight_kernel<<<(unsigned)((words + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        weight, kvol, cin, cout, cshift, NB, scale_inv, words, (unsigned *)packed);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_fp8_quantize(const void *x, int dtype, int n, const int32_t *n_dev, int c, int c_stride, int cb,
                                float scale_inv, void *out, void *stream) {
    PCD_ENTER();
    if (n < 0 || c <= 0 || c_stride < c || cb < c || (cb & 3)) return PCD_ERR_INVALID_ARG;
    if (n == 0) return PCD_OK;
    if (!x || !out) return PCD_ERR_INVALID_ARG;
    const size_t words = (size_t)n * (cb / 4);
    const unsigned grid = (unsigned)((words + 255) / 256);
    if (dtype == PCD_BF16)
        fp8_quantize_kernel<unsigned short><<<grid, 256, 0, (hipStream_t)stream>>>((const unsigned short *)x, n, n_dev, c,
                                                                                  c_stride, cb, scale_inv, (unsigned *)out);
    else if (dtype == PCD_F32)
        fp8_quantize_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float *)x, n, n_dev, c, c_stride, cb,
                                                                         scale_inv, (unsigned *)out);
    else
        return PCD_ERR_INVALID_ARG;
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_fp8_dequantize(const void *x8, size_t count, float scale, float *out, void *stream) {
    PCD_ENTER();
    if (count == 0) return PCD_OK;
    if (!x8 || !out) return PCD_ERR_INVALID_ARG;
    fp8_dequantize_kernel<<<(unsigned)((count + 255) / 256), 256, 0, (hipStream_t)stream>>>((const unsigned char *)x8, count,
                                                                                            scale, out);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_sparse_conv_gather_gemm_fp8(const void *x8, int n_rows_in, int cin_pad, const void *packed_w,
                                               const int32_t *nbr, int nbr_stride, int kvol, int flip_k, int n_rows_out,
                                               const int32_t *n_rows_out_dev, int c_out, const float *alpha,
                                               const float *beta, int relu, float out_scale_inv, void *y, int y_kind,
                                               int y_stride, void *stream) {
    PCD_ENTER();
    if (n_rows_out < 0 || n_rows_in < 0 || kvol <= 0 || c_out <= 0 || (c_out % 16)) return PCD_ERR_INVALID_ARG;
    if (y_kind < 0 || y_kind > 2 || y_stride < c_out) return PCD_ERR_INVALID_ARG;
    if (n_rows_out == 0) return PCD_OK;
    const int cshift = log2_exact_f8(cin_pad);
    if (cshift < 4) return PCD_ERR_UNSUPPORTED;
    if (!x8 || !packed_w || !nbr || !alpha || !beta || !y || nbr_stride < n_rows_out) return PCD_ERR_INVALID_ARG;
    if ((double)n_rows_in * cin_pad >= 4294900000.0) return PCD_ERR_UNSUPPORTED;
    if (y_kind == F8_OUT_FP8 && (y_stride & 3)) return PCD_ERR_INVALID_ARG;
    const unsigned x_bytes = (unsigned)((size_t)n_rows_in * cin_pad);
    const int nsteps = (kvol * cin_pad + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
#define F8_ARGS x8, cshift, packed_w, nbr, nbr_stride, kvol, flip_k, n_rows_out, n_rows_out_dev, alpha, beta, relu, out_scale_inv, y, y_kind, y_stride, nsteps, x_bytes, st
    switch (c_out / 16) {
        case 1: return launch_f8<1, 1>(F8_ARGS);
        case 2: return launch_f8<2, 2>(F8_ARGS);
        case 4: return launch_f8<4, 2>(F8_ARGS);
        case 8: return launch_f8<8, 2>(F8_ARGS);
        default: return PCD_ERR_UNSUPPORTED;
    }
#undef F8_ARGS
}
