// Rotated BEV overlap / IoU and bitmask NMS for gfx950 (replaces pcdet/ops/iou3d_nms: iou3d_nms_kernel.cu:236-413,
// host side iou3d_nms.cpp:60-188; used by eval post-processing, pcdet/models/model_utils/model_nms_utils.py:15-20,
// and recall statistics, detector3d_template.py:308).
//
// Arithmetic follows the reference's float formulas step by step (corner rotation, segment intersection with its
// two solution branches, the 1e-2 margin of the point-in-box test, the centroid / atan2 ordering of the polygon
// vertices, the bubble sort whose comparison is NOT a strict weak order for equal angles, the fan triangulation) so
// that results agree with it to the last bits the device's cos / sin / atan2 allow.
//
// MI355X-first differences:
//   * suppression mask: one 64-lane WAVE per 64 x 64 tile of (row boxes, column boxes); lane = row box, the 64
//     column boxes sit in LDS, the lane's 64 comparisons fill exactly one 64-bit word (the reference uses 64-thread
//     blocks for the same reason).  Tiles below the diagonal are never read by the reduction and are skipped.
//   * the sequential reduction (which box survives) runs ON THE DEVICE in one wave: lane j owns the j-th 64-bit
//     word(s) of the "removed" set, rows of the mask are streamed 8 ahead; the keep list and its length stay in
//     device memory -- no N x N / 64 mask copied to the host, no host loop, no allocation inside the call (the
//     reference cudaMallocs / cudaMemcpys / cudaFrees per call and loops over the boxes on the CPU).
#include "common.h"

namespace {

constexpr float IOU_EPS = 1e-8f;        // iou3d_nms_kernel.cu:14

// (the geometry is __host__ __device__: pcd_boxes_iou_bev_host below runs the SAME code on the host, where cosf / sinf /
//  atan2f are the C library's -- the functions the reference's iou3d_cpu.cpp calls)
struct P2 {
    float x, y;
};
__host__ __device__ __forceinline__ P2 mk(float x, float y) { P2 p; p.x = x; p.y = y; return p; }
__host__ __device__ __forceinline__ float cross2(const P2 &a, const P2 &b) { return a.x * b.y - a.y * b.x; }
// (p1 - p0) x (p2 - p0)
__host__ __device__ __forceinline__ float cross3(const P2 &p1, const P2 &p2, const P2 &p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

// bounding boxes of the two segments overlap (iou3d_nms_kernel.cu:43-49)
__host__ __device__ __forceinline__ bool seg_boxes_touch(const P2 &p1, const P2 &p2, const P2 &q1, const P2 &q2) {
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

// point inside the rotated rectangle, with the reference's 1e-2 margin (iou3d_nms_kernel.cu:51-61)
__host__ __device__ __forceinline__ bool in_box2d(const float *box, const P2 &p) {
    const float MARGIN = 1e-2f;
    const float cx = box[0], cy = box[1];
    const float c = cosf(-box[6]), s = sinf(-box[6]);
    const float rx = (p.x - cx) * c + (p.y - cy) * (-s);
    const float ry = (p.x - cx) * s + (p.y - cy) * c;
    return fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN;
}

// proper intersection of segments p0-p1 and q0-q1 (iou3d_nms_kernel.cu:63-92)
__host__ __device__ __forceinline__ bool seg_intersection(const P2 &p1, const P2 &p0, const P2 &q1, const P2 &q0, P2 &ans) {
    if (!seg_boxes_touch(p0, p1, q0, q1)) return false;
    const float s1 = cross3(q0, p1, p0);
    const float s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0);
    const float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > IOU_EPS) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

__host__ __device__ __forceinline__ void corners_of(const float *box, P2 (&c)[5]) {
    const float hx = box[3] / 2, hy = box[4] / 2;
    const float x1 = box[0] - hx, y1 = box[1] - hy, x2 = box[0] + hx, y2 = box[1] + hy;
    const float ca = cosf(box[6]), sa = sinf(box[6]);
    const float px[4] = {x1, x2, x2, x1}, py[4] = {y1, y1, y2, y2};
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // rotate around the centre (iou3d_nms_kernel.cu:94-98)
        c[k].x = (px[k] - box[0]) * ca + (py[k] - box[1]) * (-sa) + box[0];
        c[k].y = (px[k] - box[0]) * sa + (py[k] - box[1]) * ca + box[1];
    }
    c[4] = c[0];
}

// area of the intersection polygon of two rotated rectangles (iou3d_nms_kernel.cu:104-223)
__host__ __device__ inline float overlap_bev(const float *a, const float *b) {
    P2 ca[5], cb[5];
    corners_of(a, ca);
    corners_of(b, cb);
    P2 pts[16];
    P2 centre = mk(0.f, 0.f);
    int cnt = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            P2 x;
            if (seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], x)) {
                centre.x += x.x;
                centre.y += x.y;
                pts[cnt++] = x;
            }
        }
    for (int k = 0; k < 4; ++k) {
        if (in_box2d(a, cb[k])) {
            centre.x += cb[k].x;
            centre.y += cb[k].y;
            pts[cnt++] = cb[k];
        }
        if (in_box2d(b, ca[k])) {
            centre.x += ca[k].x;
            centre.y += ca[k].y;
            pts[cnt++] = ca[k];
        }
    }
    centre.x /= cnt;      // (cnt == 0: inf / nan, never used -- the loops below do not run)
    centre.y /= cnt;
    // bubble sort by polar angle around the centroid, exactly the reference's passes (its predicate is `>` on
    // atan2 values, so ties are left in place)
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i) {
            const float ai = atan2f(pts[i].y - centre.y, pts[i].x - centre.x);
            const float an = atan2f(pts[i + 1].y - centre.y, pts[i + 1].x - centre.x);
            if (ai > an) {
                const P2 t = pts[i];
                pts[i] = pts[i + 1];
                pts[i + 1] = t;
            }
        }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k)
        area += cross2(mk(pts[k].x - pts[0].x, pts[k].y - pts[0].y), mk(pts[k + 1].x - pts[0].x, pts[k + 1].y - pts[0].y));
    return fabsf(area) / 2.0f;
}

__host__ __device__ __forceinline__ float iou_bev_dev(const float *a, const float *b) {     // iou3d_nms_kernel.cu:225-234
    const float sa = a[3] * a[4], sb = b[3] * b[4];
    const float so = overlap_bev(a, b);
    return so / fmaxf(sa + sb - so, IOU_EPS);
}

__device__ __forceinline__ float iou_normal_dev(const float *a, const float *b) {  // iou3d_nms_kernel.cu:312-324
    const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    const float inter = w * h;
    return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, IOU_EPS);
}

// one thread per (a, b) pair; 64 b-boxes per wave row so that the a-box loads are wave-uniform broadcasts
template <bool IOU>
__global__ __launch_bounds__(256) void pairwise_kernel(const float *__restrict__ boxes_a, int na,
                                                       const float *__restrict__ boxes_b, int nb,
                                                       float *__restrict__ out) {
    const int bi = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ai = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ai >= na || bi >= nb) return;
    float a[7], b[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        a[j] = boxes_a[(size_t)ai * 7 + j];
        b[j] = boxes_b[(size_t)bi * 7 + j];
    }
    out[(size_t)ai * nb + bi] = IOU ? iou_bev_dev(a, b) : overlap_bev(a, b);
}

// suppression mask: grid (col_blocks, row_blocks), one wave per tile; tiles below the diagonal exit
template <bool NORMAL>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float *__restrict__ boxes, int n, float thresh,
                                                      u64 *__restrict__ mask, int col_blocks) {
    const int col = blockIdx.x, row = blockIdx.y;
    if (col < row) return;
    __shared__ float cb[64 * 7];
    const int lane = threadIdx.x;
    const int ncol = min(n - col * 64, 64), nrow = min(n - row * 64, 64);
    for (int e = lane; e < ncol * 7; e += 64) cb[e] = boxes[(size_t)col * 64 * 7 + e];
    __syncthreads();
    if (lane >= nrow) return;
    const int me = row * 64 + lane;
    float a[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) a[j] = boxes[(size_t)me * 7 + j];
    u64 t = 0;
    for (int i = (row == col) ? lane + 1 : 0; i < ncol; ++i) {
        const float v = NORMAL ? iou_normal_dev(a, cb + i * 7) : iou_bev_dev(a, cb + i * 7);
        if (v > thresh) t |= 1ull << i;
    }
    mask[(size_t)me * col_blocks + col] = t;
}

// greedy reduction (iou3d_nms.cpp:100-130) in ONE wave: lane l owns words l, l + 64, ... of the removed set
__global__ __launch_bounds__(64) void nms_reduce_kernel(const u64 *__restrict__ mask, int n, int col_blocks,
                                                        long long *__restrict__ keep, int32_t *__restrict__ num_keep) {
    extern __shared__ u64 remv[];                      // [col_blocks]
    const int lane = threadIdx.x;
    for (int w = lane; w < col_blocks; w += 64) remv[w] = 0;
    __syncthreads();
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        const int blk = i >> 6, bit = i & 63;
        const u64 word = remv[blk];                    // wave-uniform read (same address in every lane)
        if (!((word >> bit) & 1ull)) {
            if (lane == 0) keep[kept] = i;
            ++kept;
            // only words >= blk matter from here on (upper triangle); coalesced 8-byte loads
            for (int w = blk + lane; w < col_blocks; w += 64) remv[w] |= mask[(size_t)i * col_blocks + w];
            __syncthreads();
        }
    }
    if (lane == 0) *num_keep = kept;
}

}  // namespace

extern "C" int pcd_boxes_overlap_bev(const float *boxes_a, int num_a, const float *boxes_b, int num_b, float *out,
                                     int want_iou, void *stream) {
    PCD_ENTER();
    if (num_a < 0 || num_b < 0) return PCD_ERR_INVALID_ARG;
    if (num_a == 0 || num_b == 0) return PCD_OK;
    if (!boxes_a || !boxes_b || !out) return PCD_ERR_INVALID_ARG;
    dim3 grid((unsigned)pcd_div_up(num_b, 64), (unsigned)pcd_div_up(num_a, 4));
    if (want_iou)
        pairwise_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(boxes_a, num_a, boxes_b, num_b, out);
    else
        pairwise_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(boxes_a, num_a, boxes_b, num_b, out);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}

// Host-side variant (boxes_bev_iou_cpu, iou3d_nms_utils.py:12-28 -> src/iou3d_cpu.cpp:232-252): what COMAug's database
// sampler calls per frame for its collision test (datasets/augmentor/database_sampler_v2.py:600-601) inside DataLoader
// workers.  HOST pointers, no stream, no GPU work: the same geometry code as the kernels above, compiled for the host.
extern "C" int pcd_boxes_iou_bev_host(const float *boxes_a_host, int num_a, const float *boxes_b_host, int num_b,
                                      float *out_host) {
    if (num_a < 0 || num_b < 0) return PCD_ERR_INVALID_ARG;
    if (num_a == 0 || num_b == 0) return PCD_OK;
    if (!boxes_a_host || !boxes_b_host || !out_host) return PCD_ERR_INVALID_ARG;
    for (int i = 0; i < num_a; ++i)
        for (int j = 0; j < num_b; ++j)
            out_host[(size_t)i * num_b + j] = iou_bev_dev(boxes_a_host + (size_t)i * 7, boxes_b_host + (size_t)j * 7);
    return PCD_OK;
}

extern "C" size_t pcd_nms_workspace_bytes(int num_boxes) {
    if (num_boxes <= 0) return 256;
    const size_t cb = (size_t)pcd_div_up(num_boxes, 64);
    return ws_piece((size_t)num_boxes * cb, sizeof(u64));
}

extern "C" int pcd_nms_bev(const float *boxes, int num_boxes, float thresh, int normal, long long *keep,
                           int32_t *num_keep_dev, void *workspace, size_t workspace_bytes, void *stream) {
    PCD_ENTER();
    if (num_boxes < 0 || !num_keep_dev) return PCD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (num_boxes == 0) {
        pcd_fill(num_keep_dev, 0, sizeof(int32_t), st);
        PCD_RETURN_IF_LAUNCH_FAILED();
        return PCD_OK;
    }
    if (!boxes || !keep) return PCD_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < pcd_nms_workspace_bytes(num_boxes)) return PCD_ERR_WORKSPACE;
    const int cb = pcd_div_up(num_boxes, 64);
    if ((size_t)cb * sizeof(u64) > 64 * 1024) return PCD_ERR_UNSUPPORTED;     // > 524 288 boxes
    u64 *mask = (u64 *)workspace;
    dim3 grid((unsigned)cb, (unsigned)cb);
    if (normal)
        nms_mask_kernel<true><<<grid, 64, 0, st>>>(boxes, num_boxes, thresh, mask, cb);
    else
        nms_mask_kernel<false><<<grid, 64, 0, st>>>(boxes, num_boxes, thresh, mask, cb);
    nms_reduce_kernel<<<1, 64, (size_t)cb * sizeof(u64), st>>>(mask, num_boxes, cb, keep, num_keep_dev);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
