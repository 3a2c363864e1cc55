// CenterHead target assignment on the GPU (SURVEY.md 8f #2 "host-side step overheads"): the reference builds the
// training targets with a Python loop over ground-truth boxes, on CPU tensors, with one `.item()` and one numpy ->
// torch -> device copy per box (pcdet/models/dense_heads/center_head.py:104-161, :163-225 and
// pcdet/models/model_utils/centernet_utils.py:46-107) -- a per-step host stall that grows with the number of
// objects.  Here: one wave per batch element compacts the boxes of the head's classes (ballot prefix keeps their
// order), computes centre / Gaussian radius / regression targets with the reference's float32 formulas, and a second
// kernel draws every Gaussian with an order-free atomic max (heat-map values are >= 0, so their float bits order like
// integers) -- deterministic, no host round trip.
#include "common.h"

namespace {

struct AssignGeom {
    float range_x, range_y, vs_x, vs_y;
    int stride, W, H, num_max, min_radius, code, head_classes;
    float overlap;
    int cls_map[16];       // dataset class id (1-based; 0 = padding) -> 1-based id inside this head, 0 = not in the head
};

// centernet_utils.py:46-72 in float32, operation by operation
__device__ __forceinline__ float gaussian_radius_f32(float height, float width, float min_overlap) {
    const float a1 = 1.0f;
    const float b1 = height + width;
    const float c1 = width * height * (1.0f - min_overlap) / (1.0f + min_overlap);
    const float sq1 = sqrtf(b1 * b1 - 4.0f * a1 * c1);
    const float r1 = (b1 + sq1) / 2.0f;
    const float a2 = 4.0f;
    const float b2 = 2.0f * (height + width);
    const float c2 = (1.0f - min_overlap) * width * height;
    const float sq2 = sqrtf(b2 * b2 - 4.0f * a2 * c2);
    const float r2 = (b2 + sq2) / 2.0f;
    const float a3 = 4.0f * min_overlap;
    const float b3 = -2.0f * min_overlap * (height + width);
    const float c3 = (min_overlap - 1.0f) * width * height;
    const float sq3 = sqrtf(b3 * b3 - 4.0f * a3 * c3);
    const float r3 = (b3 + sq3) / 2.0f;
    return fminf(fminf(r1, r2), r3);
}

__global__ __launch_bounds__(64) void assign_rows_kernel(const float *__restrict__ gt, int n, AssignGeom G,
                                                         float *__restrict__ ret_boxes, long long *__restrict__ inds,
                                                         long long *__restrict__ mask, int4 *__restrict__ draw) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float *rows = gt + (size_t)b * n * G.code;
    const int out_code = G.code;                              // ret_boxes rows: code - 1 + 1 values
    int count = 0;
    for (int base = 0; base < n && count < G.num_max; base += 64) {
        const int r = base + lane;
        int local = 0;
        if (r < n) {
            const int cls = (int)rows[(size_t)r * G.code + G.code - 1];
            local = (cls >= 0 && cls < 16) ? G.cls_map[cls] : 0;
        }
        int total;
        const int k = count + wave_rank(local > 0, total);
        count += total;
        if (local <= 0 || k >= G.num_max) continue;
        const float *q = rows + (size_t)r * G.code;
        float cx = (q[0] - G.range_x) / G.vs_x / (float)G.stride;
        float cy = (q[1] - G.range_y) / G.vs_y / (float)G.stride;
        cx = fminf(fmaxf(cx, 0.0f), (float)G.W - 0.5f);
        cy = fminf(fmaxf(cy, 0.0f), (float)G.H - 0.5f);
        const int ix = (int)cx, iy = (int)cy;
        const float dx = q[3] / G.vs_x / (float)G.stride, dy = q[4] / G.vs_y / (float)G.stride;
        int radius = (int)gaussian_radius_f32(dx, dy, G.overlap);
        radius = radius < G.min_radius ? G.min_radius : radius;
        const size_t at = (size_t)b * G.num_max + k;
        int4 d = make_int4(-1, 0, 0, 0);
        if (dx > 0.0f && dy > 0.0f && ix >= 0 && ix <= G.W && iy >= 0 && iy <= G.H) {
            inds[at] = (long long)iy * G.W + ix;
            mask[at] = 1;
            float *o = ret_boxes + at * out_code;
            o[0] = cx - (float)ix;
            o[1] = cy - (float)iy;
            o[2] = q[2];
            o[3] = logf(q[3]);
            o[4] = logf(q[4]);
            o[5] = logf(q[5]);
            o[6] = cosf(q[6]);
            o[7] = sinf(q[6]);
            for (int j = 8; j < out_code; ++j) o[j] = q[j - 1];
            d = make_int4(local - 1, ix, iy, radius);
        }
        draw[at] = d;
    }
}

// one wave per (batch, object): max the object's Gaussian into its class plane (centernet_utils.py:75-107)
__global__ __launch_bounds__(64) void draw_gaussian_kernel(const int4 *__restrict__ draw, AssignGeom G,
                                                           float *__restrict__ heatmap) {
    const size_t at = blockIdx.x;
    const int4 d = draw[at];
    if (d.x < 0) return;
    const int b = (int)(at / G.num_max);
    const int radius = d.w, x = d.y, y = d.z;
    const int left = min(x, radius), right = min(G.W - x, radius + 1);
    const int top = min(y, radius), bottom = min(G.H - y, radius + 1);
    const int w = left + right, h = top + bottom;
    if (w <= 0 || h <= 0) return;
    const double sigma = (double)(2 * radius + 1) / 6.0;
    const double eps_cut = 2.220446049250313e-16;             // np.finfo(float64).eps * h.max(), h.max() = 1
    int *plane = reinterpret_cast<int *>(heatmap + ((size_t)b * G.head_classes + d.x) * G.H * G.W);
    for (int p = threadIdx.x; p < w * h; p += 64) {
        const int py = p / w, px = p - py * w;
        const int gx = px - left, gy = py - top;               // offset from the centre
        double v = exp(-(double)(gx * gx + gy * gy) / (2.0 * sigma * sigma));
        if (v < eps_cut) v = 0.0;
        const float f = (float)v;
        atomicMax(plane + (size_t)(y + gy) * G.W + (x + gx), __float_as_int(f));
    }
}

}  // namespace

extern "C" size_t pcd_centerhead_assign_workspace_bytes(int batch, int num_max_objs) {
    if (batch <= 0 || num_max_objs <= 0) return 0;
    return ws_piece((size_t)batch * num_max_objs, sizeof(int4));
}

extern "C" int pcd_centerhead_assign_targets(const float *gt_boxes, int batch, int n_boxes, int code_size,
                                             const int *class_map_host, int n_class_map, int head_classes, int fm_w, int fm_h,
                                             int feature_map_stride, const float *voxel_size_xy_host,
                                             const float *range_xy_host, int num_max_objs, float gaussian_overlap,
                                             int min_radius, float *heatmap, float *ret_boxes, long long *inds,
                                             long long *mask, void *workspace, size_t workspace_bytes, void *stream) {
    PCD_ENTER();
    if (batch <= 0 || n_boxes < 0 || code_size < 8 || !class_map_host || n_class_map <= 0 || n_class_map > 16 ||
        head_classes <= 0 || fm_w <= 0 || fm_h <= 0 || feature_map_stride <= 0 || num_max_objs <= 0 || !voxel_size_xy_host ||
        !range_xy_host)
        return PCD_ERR_INVALID_ARG;
    if (!heatmap || !ret_boxes || !inds || !mask) return PCD_ERR_INVALID_ARG;
    if (n_boxes > 0 && !gt_boxes) return PCD_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < pcd_centerhead_assign_workspace_bytes(batch, num_max_objs)) return PCD_ERR_WORKSPACE;
    AssignGeom G = {};
    G.range_x = range_xy_host[0]; G.range_y = range_xy_host[1];
    G.vs_x = voxel_size_xy_host[0]; G.vs_y = voxel_size_xy_host[1];
    G.stride = feature_map_stride; G.W = fm_w; G.H = fm_h; G.num_max = num_max_objs; G.min_radius = min_radius;
    G.code = code_size; G.head_classes = head_classes; G.overlap = gaussian_overlap;
    for (int i = 0; i < n_class_map; ++i) G.cls_map[i] = class_map_host[i];
    hipStream_t st = (hipStream_t)stream;
    int4 *draw = (int4 *)workspace;
    pcd_fill(heatmap, 0, (size_t)batch * head_classes * fm_h * fm_w * sizeof(float), st);
    pcd_fill(ret_boxes, 0, (size_t)batch * num_max_objs * code_size * sizeof(float), st);
    pcd_fill(inds, 0, (size_t)batch * num_max_objs * sizeof(long long), st);
    pcd_fill(mask, 0, (size_t)batch * num_max_objs * sizeof(long long), st);
    pcd_fill(draw, 0xFF, (size_t)batch * num_max_objs * sizeof(int4), st);
    assign_rows_kernel<<<batch, 64, 0, st>>>(gt_boxes, n_boxes, G, ret_boxes, inds, mask, draw);
    draw_gaussian_kernel<<<(unsigned)((size_t)batch * num_max_objs), 64, 0, st>>>(draw, G, heatmap);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
