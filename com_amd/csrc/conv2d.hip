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

// all packs of a model in ONE launch: table rows {weight, packed, cin, cout, cout_pad, mode, first block, pieces}
__global__ __launch_bounds__(256) void conv2d_pack_batched_kernel(const long long *__restrict__ table, int n) {
    int j = 0;
    for (int q = 1; q < n; ++q)
        if (table[(size_t)q * 8 + 6] <= (long long)blockIdx.x) j = q;
    const long long *row = table + (size_t)j * 8;
    const size_t e = (size_t)(blockIdx.x - row[6]) * 256 + threadIdx.x;
    if (e < (size_t)row[7])
        conv2d_pack_piece((const float *)row[0], (int)row[2], (int)row[3], (int)row[4], (int)row[5],
                          (unsigned short *)row[1], e);
}

template <int WB>   // weight stage buffers: 2 = double buffered (113 KB LDS, 1 workgroup / CU); 1 = single (77 KB, 2 / CU)
__global__ __launch_bounds__(256, WB == 2 ? 1 : 2) void conv2d_3x3_kernel(const unsigned short *__restrict__ x, int B, int H, int W,
                                                             int cin, const uint4 *__restrict__ wp, int cout,
                                                             const float *__restrict__ bias,
                                                             unsigned short *__restrict__ y, unsigned x_bytes,
                                                             unsigned w_bytes) {
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
        in_off[it] = ok ? (unsigned)((((size_t)b * H + gy) * W + gx) * cin * 2 + piece * 16) : 0xFFFFFFF0u;
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
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
        const int gy = y0 + wave * 4 + pb, gx = x0 + m;
        if (gy >= H || gx >= W || c0 >= cout) continue;
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
        uint4 *dst = reinterpret_cast<uint4 *>(y + (((size_t)b * H + gy) * W + gx) * cout + c0);
        dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
        dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
}

}  // namespace

static int pad32(int c) { return (c + 31) / 32 * 32; }

// (output channels are padded to a multiple of 32: the data gradient contracts over them in steps of 32)
extern "C" size_t pcd_conv2d_packed_weight_bytes(int cin, int cout, int mode) {
    if (cin <= 0 || cout <= 0 || (mode != 0 && mode != 1)) return 0;
    const int cp = pad32(cout);
    const int n_out = mode == 0 ? cp : cin, n_in = mode == 0 ? cin : cp;
    return (size_t)((n_out + 63) / 64) * (n_in / 32) * W_BYTES;
}

extern "C" int pcd_conv2d_pack_weight(const float *weight, int cin, int cout, int mode, void *packed, void *stream) {
    PCD_ENTER();
    if (!weight || !packed || cin <= 0 || cout <= 0 || (mode != 0 && mode != 1)) return PCD_ERR_INVALID_ARG;
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

extern "C" int pcd_conv2d_3x3_nhwc(const void *x, int batch, int height, int width, int cin, const void *packed_w,
                                   int cout, const float *bias, void *y, void *stream) {
    PCD_ENTER();
    if (batch <= 0 || height <= 0 || width <= 0 || cin <= 0 || cout <= 0) return PCD_ERR_INVALID_ARG;
    if (!x || !packed_w || !y) return PCD_ERR_INVALID_ARG;
    if (cin % 32 || cout % 16) return PCD_ERR_UNSUPPORTED;
    const double xb = (double)batch * height * width * cin * 2;
    if (xb >= 4294966000.0) return PCD_ERR_UNSUPPORTED;
    const size_t wb_bytes = (size_t)((cout + 63) / 64) * (cin / 32) * W_BYTES;   // bytes of the pack the kernel reads
    static const int wb = getenv("PCD_CONV2D_WB") ? atoi(getenv("PCD_CONV2D_WB")) : 1;   // (1: 28-35 % of the MFMA peak, 2: 21-27 %)
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
                                                                     (unsigned short *)y, (unsigned)xb, (unsigned)wb_bytes);
    else
        conv2d_3x3_kernel<2><<<grid, 256, lds, (hipStream_t)stream>>>((const unsigned short *)x, batch, height, width, cin,
                                                                     (const uint4 *)packed_w, cout, bias,
                                                                     (unsigned short *)y, (unsigned)xb, (unsigned)wb_bytes);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
