// Box-measured stream rate: a float4 (16 B per lane) copy kernel over two 2 GiB buffers - far beyond the 256 MiB Infinity Cache, so the
// figure is HBM read + write bandwidth as a plain streaming kernel achieves it on this box (MI355X_MICROARCH.md: 6.29 TB/s, 79 % of the
// 8 TB/s specification).  The bandwidth tables of profiles/ quote kernels against this, not against an ATen copy_.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy4(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    float4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    for (int wgs : {2048, 4096, 8192, 16384}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(ev0));
            copy4<<<wgs, 256>>>(a, b, n);
            CK(hipEventRecord(ev1)); CK(hipEventSynchronize(ev1));
            float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
            if (ms < best) best = ms;
        }
        printf("float4 copy 2 GiB -> 2 GiB, %5d workgroups: %.3f ms = %.2f TB/s (read + write)\n", wgs, best, 2.0 * bytes / best / 1e9);
    }
    return 0;
}
