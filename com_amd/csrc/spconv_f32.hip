// fp32-exact sparse convolution: features, weights and results in fp32, v_mfma_f32_16x16x4_f32.
//
// The reference runs fp32 end to end (spconv's default algo with fp32 weights); the
// production kernels (spconv.hip) multiply bf16 operands.  These forms exist for PARITY work -- reproducing a
// reference checkpoint's activations to fp32 accuracy, and end-to-end checks of a whole backbone at the 1e-3
// tolerance without bf16 rounding noise in the way -- not for speed: no LDS staging, no software pipeline, every
// wave reads the weights it needs from L2.  Replaces the same spconv ops as pcd_sparse_conv_gather_gemm / _wgrad
// (pcdet/utils/spconv_utils.py:3-6, spconv_backbone.py:12-15), arithmetic of SURVEY.md Appendix A.5.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MFMA 16x16x4 f32 layout: A[i][k], B[k][j] are supplied by lane (i or j = lane % 16, k = lane / 16); D[i][j] comes
// back in lane (j = lane % 16), rows i = 4 * (lane / 16) + r.  The contraction index of step (t, u) is channel
// 16 t + 4 g + u for lane group g: any order works as long as A and B agree, and this one lets every lane use the
// four elements of ONE 16-byte load over four MFMA steps.
//
// One wave = 16 output rows x all output channels (NB blocks of 16); workgroup = 4 waves = 64 rows.
template <int NB>
__global__ __launch_bounds__(256) void gg_f32_kernel(const float *__restrict__ x, int c_in, const float *__restrict__ w,
                                                     const float *__restrict__ bias, const int32_t *__restrict__ nbr,
                                                     int nbr_stride, int K, int flip, int n_out_cap,
                                                     const int32_t *__restrict__ n_out_dev, int c_out,
                                                     float *__restrict__ y, const float *__restrict__ addend) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rl = lane & 15, g = lane >> 4;
    const int n_out = eff_rows(n_out_dev, n_out_cap);
    const int row0 = (blockIdx.x * 4 + wave) * 16;
    if (row0 >= n_out) return;
    const int row = row0 + rl;
    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int c_in4 = (c_in + 15) & ~15;
    for (int k = 0; k < K; ++k) {
        const int krow = flip ? (K - 1 - k) : k;
        const int i = row < n_out ? nbr[(size_t)krow * nbr_stride + row] : -1;
        if (__builtin_amdgcn_ballot_w64(i >= 0) == 0ull) continue;          // no neighbour in this 16-row tile
        const float *xr = x + (size_t)(i >= 0 ? i : 0) * c_in;
        for (int t = 0; t < c_in4; t += 16) {
            const int c = t + 4 * g;
            float a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = (i >= 0 && c + u < c_in) ? xr[c + u] : 0.0f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int co = nb * 16 + rl;
                const float *wr = w + ((size_t)co * K + k) * c_in;
                float b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) b[u] = (co < c_out && c + u < c_in) ? wr[c + u] : 0.0f;
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc[nb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int co = nb * 16 + rl;
        if (co >= c_out) continue;
        const float bv = bias ? bias[co] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = row0 + 4 * g + r;
            if (o < n_out) {
                float v = acc[nb][r] + bv;
                if (addend) v += addend[(size_t)o * c_out + co];
                y[(size_t)o * c_out + co] = v;
            }
        }
    }
}

// dW[co][k][ci] = sum over the pairs (i, o) of offset k, in list order:  A[i = co][k = pair] = dy[o][co],
// B[k = pair][j = ci] = x[i][ci].  One WAVE per (offset, 16 x 16 block of dW) walks the whole pair list, 4 pairs per
// MFMA step, lane group g taking pair 4 s + g: a fixed order, no atomics, no cross-wave reduction.
__global__ __launch_bounds__(64) void wgrad_f32_kernel(const float *__restrict__ x, int c_in, const float *__restrict__ dy,
                                                       int c_out, const int32_t *__restrict__ pairs,
                                                       const int32_t *__restrict__ pair_num, int K, int pmax,
                                                       float *__restrict__ dw) {
    const int lane = threadIdx.x & 63, rl = lane & 15, g = lane >> 4;
    const int nci = (c_in + 15) / 16, nco = (c_out + 15) / 16;
    const int k = blockIdx.x / (nci * nco), rem = blockIdx.x % (nci * nco);
    const int cob = rem / nci, cib = rem % nci;
    const int P = pair_num[k];
    const int32_t *pin = pairs + ((size_t)k * 2 + 0) * pmax;
    const int32_t *pout = pairs + ((size_t)k * 2 + 1) * pmax;
    const int co = cob * 16 + rl, ci = cib * 16 + rl;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int p0 = 0; p0 < P; p0 += 16) {                  // 4 MFMA steps = 16 pairs per iteration, loads first
        float a[4], b[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int p = p0 + 4 * s + g;
            const bool ok = p < P;
            const int i = ok ? pin[p] : 0, o = ok ? pout[p] : 0;
            a[s] = (ok && co < c_out) ? dy[(size_t)o * c_out + co] : 0.0f;
            b[s] = (ok && ci < c_in) ? x[(size_t)i * c_in + ci] : 0.0f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cor = cob * 16 + 4 * g + r;             // D[i = co row][j = ci column]
        if (cor < c_out && ci < c_in) dw[((size_t)cor * K + k) * c_in + ci] = acc[r];
    }
}

}  // namespace

extern "C" int pcd_sparse_conv_gather_gemm_f32(const float *x, int n_rows_in, int c_in, const float *weight,
                                               const float *bias, const int32_t *nbr, int nbr_stride, int kvol,
                                               int flip_k, int n_rows_out, const int32_t *n_rows_out_dev, int c_out,
                                               float *y, const float *addend, void *stream) {
    PCD_ENTER();
    if (n_rows_in < 0 || n_rows_out < 0 || kvol <= 0 || c_in <= 0 || c_out <= 0) return PCD_ERR_INVALID_ARG;
    if (n_rows_out == 0) return PCD_OK;
    if (!x || !weight || !nbr || !y || nbr_stride < n_rows_out) return PCD_ERR_INVALID_ARG;
    const int nb = (c_out + 15) / 16;
    const unsigned grid = (unsigned)pcd_div_up(n_rows_out, 64);
    hipStream_t st = (hipStream_t)stream;
#define GF(N)                                                                                                         \
    case N:                                                                                                           \
        gg_f32_kernel<N><<<grid, 256, 0, st>>>(x, c_in, weight, bias, nbr, nbr_stride, kvol, flip_k, n_rows_out,      \
                                               n_rows_out_dev, c_out, y, addend);                                     \
        break;
    switch (nb) {
        GF(1) GF(2) GF(3) GF(4) GF(5) GF(6) GF(7) GF(8)
        default: return PCD_ERR_UNSUPPORTED;              // more than 128 output channels
    }
#undef GF
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_sparse_conv_wgrad_f32(const float *x, int n_x, int c_in, const float *dy, int n_dy, int c_out,
                                         const int32_t *pairs, const int32_t *pair_num, int kvol, int pmax,
                                         float *dweight, void *stream) {
    PCD_ENTER();
    if (kvol <= 0 || c_in <= 0 || c_out <= 0 || pmax < 0 || n_x < 0 || n_dy < 0 || !dweight) return PCD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (pmax == 0) {
        pcd_fill(dweight, 0, (size_t)c_out * kvol * c_in * sizeof(float), st);
        return PCD_OK;
    }
    if (!x || !dy || !pairs || !pair_num) return PCD_ERR_INVALID_ARG;
    const unsigned grid = (unsigned)kvol * (unsigned)((c_in + 15) / 16) * (unsigned)((c_out + 15) / 16);
    wgrad_f32_kernel<<<grid, 64, 0, st>>>(x, c_in, dy, c_out, pairs, pair_num, kvol, pmax, dweight);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
