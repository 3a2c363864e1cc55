// Library info entry points of libpcdops_hip.so.
#include "common.h"

extern "C" int pcd_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char *pcd_build_arch(void) { return "gfx950"; }

extern "C" const char *pcd_error_string(int code) {
    switch (code) {
        case PCD_OK: return "ok";
        case PCD_ERR_INVALID_ARG: return "invalid argument";
        case PCD_ERR_UNSUPPORTED: return "unsupported shape / dtype for this build";
        case PCD_ERR_KEYSPACE: return "batch * D * H * W does not fit the 32-bit coordinate key";
        case PCD_ERR_WORKSPACE: return "workspace too small";
        case PCD_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}

static thread_local int g_last_hip_error = 0;
extern "C" void pcd_set_last_hip_error(int code) { g_last_hip_error = code; }
extern "C" const char *pcd_last_hip_error_string(void) {
    return hipGetErrorString((hipError_t)g_last_hip_error);
}

// 0 when `stream` is not capturing, else the id of the capture it belongs to (hipStreamGetCaptureInfo): lets the host
// side key per-graph resources (e.g. the counter blocks of the fused BatchNorm reductions) by the capture they were
// allocated in, so that two graphs -- or a graph and eager launches -- never share a counter.
extern "C" int pcd_stream_capture_id(void *stream, unsigned long long *id_out) {
    if (!id_out) return PCD_ERR_INVALID_ARG;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipError_t e = hipStreamGetCaptureInfo((hipStream_t)stream, &st, &id);
    if (e != hipSuccess) {
        pcd_set_last_hip_error((int)e);
        return PCD_ERR_LAUNCH;
    }
    *id_out = st == hipStreamCaptureStatusActive ? (id ? id : 1ull) : 0ull;
    return PCD_OK;
}

// ---- tuning options (pcd_ops.h) ----------------------------------------------------------------------------------
#include <string.h>
namespace {
struct OptRow {
    const char *key;
    int value;
};
OptRow g_opts[PCD_OPT_COUNT] = {
    {"gg_resident_kb", 32}, {"ggw", 1}, {"gg1", 1}, {"subm_window", 23}, {"subm_window_wgrad", 6}, {"wg128", 1}, {"wg128_chunks", 512},
    {"wg_rows", 6144}, {"conv2d_wb", 1}, {"conv2d_wg_blocks", 128}, {"conv2d_wgp_mode2", 0}, {"conv2d_wgp_blocks", 512},
    {"fps_g", 0}, {"gg_dbg", 0}, {"ggw_dbg", 0}, {"win_dbg", 0}, {"cm_direct_blocks", 4096}, {"subm_window_grid", 256}, {"ggwin", 0}, {"gg2", 0}, {"ggw_mi", 0}, {"ggw_cw", 2}, {"vox_emit_rows", 1}, {"vox_grid", 0}, {"subm_window_half", 0},
};
}  // namespace
int pcd_opt(int which) { return g_opts[which].value; }
extern "C" int pcd_set_option(const char *key, int value) {
    if (!key) return PCD_ERR_INVALID_ARG;
    for (int i = 0; i < PCD_OPT_COUNT; ++i)
        if (strcmp(key, g_opts[i].key) == 0) {
            g_opts[i].value = value;
            return PCD_OK;
        }
    return PCD_ERR_INVALID_ARG;
}
extern "C" int pcd_get_option(const char *key, int *value_out) {
    if (!key || !value_out) return PCD_ERR_INVALID_ARG;
    for (int i = 0; i < PCD_OPT_COUNT; ++i)
        if (strcmp(key, g_opts[i].key) == 0) {
            *value_out = g_opts[i].value;
            return PCD_OK;
        }
    return PCD_ERR_INVALID_ARG;
}

// ---- diagnostics for tools/exp_cu_mask.py -----------------------------------------------------------------------------
// A stream restricted to the compute units whose bits are set in cu_mask (hipExtStreamCreateWithCUMask), and a kernel that
// keeps `blocks` workgroups of 1024 threads busy for `ticks` ticks of the 100 MHz device clock: how long a launch of 256 of
// them takes says how many CUs it was given -- eagerly, and when the launch is replayed as a hipGraph kernel node.
static __global__ __launch_bounds__(1024) void debug_spin_kernel(unsigned long long ticks, unsigned *xcc_seen) {
    const unsigned long long t0 = wall_clock64();
    if (xcc_seen && threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        atomicOr(xcc_seen, 1u << (xcc & 15u));
    }
    while (wall_clock64() - t0 < ticks) {
    }
}
extern "C" int pcd_debug_stream_create_cu_mask(const uint32_t *cu_mask, int words, void **stream_out) {
    if (!cu_mask || words <= 0 || !stream_out) return PCD_ERR_INVALID_ARG;
    hipStream_t st = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, cu_mask);
    if (e != hipSuccess) {
        pcd_set_last_hip_error((int)e);
        return PCD_ERR_LAUNCH;
    }
    *stream_out = (void *)st;
    return PCD_OK;
}
extern "C" int pcd_debug_spin(int blocks, unsigned long long ticks, uint32_t *xcc_seen, void *stream) {
    PCD_ENTER();
    if (blocks <= 0) return PCD_ERR_INVALID_ARG;
    debug_spin_kernel<<<blocks, 1024, 0, (hipStream_t)stream>>>(ticks, xcc_seen);
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
// the same with `threads` (a multiple of 64, <= 1024) per workgroup, `lds_bytes` of dynamic LDS and ~`vgprs` live vector
// registers per lane (24 / 40 / 56 / 72; 0 = minimal): a co-tenant of chosen weight for the residency experiments
// (tools/exp_coresident.py)
template <int NV>
static __global__ __launch_bounds__(1024) void debug_spin_regs_kernel(unsigned long long ticks, float *sink) {
    const unsigned long long t0 = wall_clock64();
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (float)(threadIdx.x + i);
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[i]));
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    if (sink && s == 12345.678f) *sink = s;
}
extern "C" int pcd_debug_spin_shape(int blocks, int threads, int lds_bytes, int vgprs, unsigned long long ticks, void *stream) {
    PCD_ENTER();
    if (blocks <= 0 || threads <= 0 || threads > 1024 || threads % 64 || lds_bytes < 0 || lds_bytes > 64 * 1024) return PCD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (vgprs) {
        case 0: debug_spin_kernel<<<blocks, threads, lds_bytes, st>>>(ticks, nullptr); break;
        case 24: debug_spin_regs_kernel<16><<<blocks, threads, lds_bytes, st>>>(ticks, nullptr); break;
        case 40: debug_spin_regs_kernel<32><<<blocks, threads, lds_bytes, st>>>(ticks, nullptr); break;
        case 56: debug_spin_regs_kernel<48><<<blocks, threads, lds_bytes, st>>>(ticks, nullptr); break;
        case 72: debug_spin_regs_kernel<64><<<blocks, threads, lds_bytes, st>>>(ticks, nullptr); break;
        default: return PCD_ERR_INVALID_ARG;
    }
    PCD_RETURN_IF_LAUNCH_FAILED();
    return PCD_OK;
}
