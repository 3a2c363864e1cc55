// Probe: how long does a device radix sort of one batch's (voxel key, point index) pairs take on gfx950?
// hipcc --offload-arch=gfx950 -O3 -o sort_probe sort_probe.hip ; ./sort_probe
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    for (int n : {160000, 640000, 1280000, 5120000}) {
        std::vector<unsigned> hk(n), hv(n);
        for (int i = 0; i < n; ++i) { hk[i] = (unsigned)(((unsigned long long)rand() * 2654435761ull) % (1u << 29)); hv[i] = i; }
        unsigned *k0, *k1, *v0, *v1;
        CK(hipMalloc(&k0, n * 4)); CK(hipMalloc(&k1, n * 4)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
        CK(hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice));
        for (int bits : {24, 29, 32}) {
            size_t tb = 0;
            CK(rocprim::radix_sort_pairs(nullptr, tb, k0, k1, v0, v1, n, 0, bits, 0));
            void *tmp; CK(hipMalloc(&tmp, tb));
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            for (int w = 0; w < 3; ++w) CK(rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, n, 0, bits, 0));
            CK(hipEventRecord(a, 0));
            const int R = 20;
            for (int r = 0; r < R; ++r) CK(rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, n, 0, bits, 0));
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("pairs u32/u32 n=%d bits=%d: %.1f us (temp %zu B)\n", n, bits, ms * 1000 / R, tb);
            size_t tb2 = 0;
            CK(rocprim::radix_sort_keys(nullptr, tb2, k0, k1, n, 0, bits, 0));
            void *tmp2; CK(hipMalloc(&tmp2, tb2));
            for (int w = 0; w < 3; ++w) CK(rocprim::radix_sort_keys(tmp2, tb2, k0, k1, n, 0, bits, 0));
            CK(hipEventRecord(a, 0));
            for (int r = 0; r < R; ++r) CK(rocprim::radix_sort_keys(tmp2, tb2, k0, k1, n, 0, bits, 0));
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b));
            printf("keys  u32     n=%d bits=%d: %.1f us\n", n, bits, ms * 1000 / R);
            CK(hipFree(tmp)); CK(hipFree(tmp2));
        }
        CK(hipFree(k0)); CK(hipFree(k1)); CK(hipFree(v0)); CK(hipFree(v1));
    }
    return 0;
}
