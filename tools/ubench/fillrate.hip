// Micro-benchmark: chip-wide fill rate of (a) global_load_lds 16 B/lane (LDS-DMA) and (b) global_load_dwordx4 into
// registers, for an L2-resident and an HBM-resident source, at several resident-workgroup counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int LDS_KB>
__global__ __launch_bounds__(256) void dma_fill(const char *__restrict__ src, size_t span_mask, int iters, int *sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6;
    size_t off = ((size_t)blockIdx.x * 65536 + (size_t)tid * 16);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {          // 4 x 4 KiB per workgroup per iteration (16 KiB stage)
            const char *g = src + ((off + (size_t)j * 4096) & span_mask);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(lds + ((it & 1) * 16384 + j * 4096 + wave * 1024)), 16, 0, 0);
        }
        off += 16384;
        if ((it & 1) == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lds[tid] == 123 && sink) sink[0] = 1;
}

__global__ __launch_bounds__(256) void reg_fill(const uint4 *__restrict__ src, size_t span_mask, int iters, int *sink) {
    const int tid = threadIdx.x;
    size_t off = ((size_t)blockIdx.x * 65536 + (size_t)tid * 16);
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 v = *(const uint4 *)((const char *)src + ((off + (size_t)j * 4096) & span_mask));
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
        off += 16384;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345 && sink) sink[0] = 1;
}

int main() {
    const size_t big = (size_t)1 << 31;      // 2 GiB source
    char *src; int *sink;
    CK(hipMalloc(&src, big)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(src, 1, big));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 512;
    for (size_t span : {(size_t)1 << 20, (size_t)1 << 24, (size_t)1 << 27, big}) {
        for (int wgs : {256, 512, 1024, 2048}) {
            for (int kind = 0; kind < 2; ++kind) {
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(a));
                    if (kind == 0) dma_fill<32><<<wgs, 256, 32768>>>(src, span - 1, iters, sink);
                    else reg_fill<<<wgs, 256>>>((const uint4 *)src, span - 1, iters, sink);
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    if (ms < best) best = ms;
                }
                const double bytes = (double)wgs * iters * 16384.0;
                printf("span %6zu MiB  wgs %5d  %s  %.3f ms  %.2f TB/s\n", span >> 20, wgs, kind == 0 ? "lds-dma " : "to-vgpr ", best, bytes / best / 1e9);
            }
        }
    }
    return 0;
}
