// Hard-voxel pillar encoder pieces (pcdet/models/backbones_3d/vfe/pillar_vfe.py:94-123 and the pooling half of its
// PFNLayer, :29-49), the parts the reference runs as chains of elementwise torch kernels:
//   * pcd_pillar_decorate: [M][T][C] padded pillar points -> [M][T][C + 6 (+1)] decorated, masked features in ONE
//     pass (pillar mean over the T slots, offset to the mean, offset to the pillar centre, optional range, padding
//     slots zeroed) -- the reference materialises points_mean, f_cluster, f_center, the concatenation and the mask
//     product as separate tensors (7 kernels, ~5 passes over the [M][T][C'] tensor);
//   * pcd_pfn_relu_pool (+ backward): ReLU, max over the T points of a pillar and -- for a non-final PFN stage --
//     the concatenation [h, broadcast max] written directly (reference: relu, max, repeat, cat = 4 passes).
// Linear and BatchNorm1d between them are library GEMM / normalisation and stay in torch (as the reference's own
// DynamicPillarVFE split does).  float32 throughout, like the reference.
#include "common.h"

namespace {

struct PillarGeom {
    float vs[3];       // voxel size x, y, z
    float off[3];      // voxel_size / 2 + range_min  (pillar_vfe.py:76-81)
};

// one thread per (pillar, slot); the T slots of a pillar sit in consecutive threads, the mean is a T-step loop over
// the pillar's xyz (3 T floats, L1-resident)
__global__ __launch_bounds__(256) void pillar_decorate_kernel(const float *__restrict__ vox, const int32_t *__restrict__ nump,
                                                              const int32_t *__restrict__ coords, int m, int T, int C,
                                                              int use_abs, int with_dist, PillarGeom G,
                                                              float *__restrict__ out, int c_out) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)m * T) return;
    const int p = (int)(e / T), t = (int)(e - (size_t)p * T);
    const float *pv = vox + (size_t)p * T * C;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int j = 0; j < T; ++j) {           // sum over ALL slots (padding is zero), in slot order (pillar_vfe.py:97)
        sx += pv[(size_t)j * C + 0];
        sy += pv[(size_t)j * C + 1];
        sz += pv[(size_t)j * C + 2];
    }
    const int n = nump[p];
    const float cnt = (float)n;             // NOT clamped, as in the reference
    const float mx = sx / cnt, my = sy / cnt, mz = sz / cnt;
    const float *q = pv + (size_t)t * C;
    const float x = q[0], y = q[1], z = q[2];
    const int32_t *cd = coords + (size_t)p * 4;      // (b, z, y, x)
    const float cx = (float)cd[3] * G.vs[0] + G.off[0];
    const float cy = (float)cd[2] * G.vs[1] + G.off[1];
    const float cz = (float)cd[1] * G.vs[2] + G.off[2];
    const float keep = t < n ? 1.0f : 0.0f;          // get_paddings_indicator, pillar_vfe.py:86-92,115-118
    float *o = out + e * c_out;
    int w = 0;
    for (int j = use_abs ? 0 : 3; j < C; ++j) o[w++] = q[j] * keep;
    o[w++] = (x - mx) * keep;
    o[w++] = (y - my) * keep;
    o[w++] = (z - mz) * keep;
    o[w++] = (x - cx) * keep;
    o[w++] = (y - cy) * keep;
    o[w++] = (z - cz) * keep;
    if (with_dist) o[w++] = sqrtf(x * x + y * y + z * z) * keep;
}

// thread per (pillar, channel): relu + max over t (first maximal slot wins), optional [h, max] concatenation
__global__ __launch_bounds__(256) void pfn_relu_pool_kernel(const float *__restrict__ x, int m, int T, int C, int last,
                                                            float *__restrict__ out, int32_t *__restrict__ arg) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)m * C) return;
    const int p = (int)(e / C), c = (int)(e - (size_t)p * C);
    const float *px = x + (size_t)p * T * C + c;
    float best = -1.0f;
    int bi = 0;
    const int oc = last ? C : 2 * C;
    for (int t = 0; t < T; ++t) {
        const float v = fmaxf(px[(size_t)t * C], 0.0f);
        if (!last) out[((size_t)p * T + t) * oc + c] = v;
        if (v > best) {
            best = v;
            bi = t;
        }
    }
    arg[e] = bi;
    if (last) {
        out[e] = best;
    } else {
        for (int t = 0; t < T; ++t) out[((size_t)p * T + t) * oc + C + c] = best;
    }
}

__global__ __launch_bounds__(256) void pfn_relu_pool_bwd_kernel(const float *__restrict__ g, const float *__restrict__ x,
                                                                const int32_t *__restrict__ arg, int m, int T, int C,
                                                                int last, float *__restrict__ gx) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)m * C) return;
    const int p = (int)(e / C), c = (int)(e - (size_t)p * C);
    const int oc = last ? C : 2 * C;
    const int bi = arg[e];
    float gmax;
    if (last) {
        gmax = g[e];
    } else {
        gmax = 0.f;
        for (int t = 0; t < T; ++t) gmax += g[((size_t)p * T + t) * oc + C + c];   // broadcast -> sum, slot order
    }
    for (int t = 0; t < T; ++t) {
        const size_t at = ((size_t)p * T + t) * C + c;
        float v = last ? 0.f : g[((size_t)p * T + t) * oc + c];
        if (t == bi) v += gmax;
        gx[at] = x[at] > 0.0f ? v : 0.0f;
    }
}

}  // namespace

extern "C" int pcd_pillar_decorate(const float *voxels, const int32_t *num_points, const int32_t *coords, int m, int T,
                                   int C, int use_absolute_xyz, int with_distance, const float *voxel_size_host,
                                   const float *offset_host, float *out, void *stream) {
    PCD_ENTER();
    if (m < 0 || T <= 0 || C < 3 || !voxel_size_host || !offset_host) return PCD_ERR_INVALID_ARG;
    if (m == 0) return PCD_OK;
    if (!voxels || !num_points || !coords || !out) return PCD_ERR_INVALID_ARG;
    PillarGeom G;
    for (int j = 0; j < 3; ++j) {
        G.vs[j] = voxel_size_host[j];
        G.off[j] = offset_host[j];
    }
    const int c_out = (use_absolute_xyz ? C : C - 3) + 6 + (with_distance ? 1 : 0);
    const size_t n = (size_t)m * T;
    pillar_decorate_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        voxels, num_points, coords, m, T, C, use_absolute_xyz, with_distance, G, out, c_out);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_pfn_relu_pool(const float *x, int m, int T, int C, int last_layer, float *out, int32_t *arg,
                                 void *stream) {
    PCD_ENTER();
    if (m < 0 || T <= 0 || C <= 0) return PCD_ERR_INVALID_ARG;
    if (m == 0) return PCD_OK;
    if (!x || !out || !arg) return PCD_ERR_INVALID_ARG;
    const size_t n = (size_t)m * C;
    pfn_relu_pool_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, m, T, C, last_layer, out, arg);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

extern "C" int pcd_pfn_relu_pool_backward(const float *grad_out, const float *x, const int32_t *arg, int m, int T, int C,
                                          int last_layer, float *grad_x, void *stream) {
    PCD_ENTER();
    if (m < 0 || T <= 0 || C <= 0) return PCD_ERR_INVALID_ARG;
    if (m == 0) return PCD_OK;
    if (!grad_out || !x || !arg || !grad_x) return PCD_ERR_INVALID_ARG;
    const size_t n = (size_t)m * C;
    pfn_relu_pool_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(grad_out, x, arg, m, T, C,
                                                                                          last_layer, grad_x);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
