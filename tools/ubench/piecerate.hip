// Micro-benchmark: chip-wide LDS-DMA fill rate (global_load_lds, 16 B per lane) as a function of the PIECE size - the number of
// consecutive bytes a group of lanes fetches from one row (64 B = a 32-channel bf16 K tile of one pixel, 128 B = 64 channels = one
// L2 line, ...), rows ROW_STRIDE bytes apart, source resident in L2 / the Infinity Cache.  What an implicit-GEMM K tile of BK channels
// costs to stage: the igemm kernels fetch 64-byte pieces (BK = 32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int PIECE_LANES>
__global__ __launch_bounds__(256) void piece_fill(const char *__restrict__ src, size_t span_mask, int row_stride, int iters, int *sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int PPI = 64 / PIECE_LANES;                 // pieces per wave instruction
    // piece p of this workgroup's instruction stream reads row (base_row + p), byte offset (lane % PIECE_LANES) * 16 in the row
    size_t row = (size_t)blockIdx.x * 4096 + (size_t)(wave * PPI + lane / PIECE_LANES);
    const int in_row = (lane % PIECE_LANES) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char *g = src + (((row + (size_t)j * 4 * PPI) * (size_t)row_stride + in_row) & span_mask);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(lds + ((it & 1) * 16384 + j * 4096 + wave * 1024)), 16, 0, 0);
        }
        row += 16 * PPI;
        if ((it & 1) == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lds[tid] == 123 && sink) sink[0] = 1;
}

template <int PL>
static void run(const char *src, size_t span, int row_stride, int wgs, int *sink) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 512;
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        piece_fill<PL><<<wgs, 256, 32768>>>(src, span - 1, row_stride, iters, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("span %5zu MiB  row stride %5d B  piece %4d B  wgs %5d  %.3f ms  %.2f TB/s\n", span >> 20, row_stride, PL * 16, wgs, best,
           (double)wgs * iters * 16384.0 / best / 1e9);
}

int main() {
    const size_t big = (size_t)1 << 30;
    char *src; int *sink;
    CK(hipMalloc(&src, big)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(src, 1, big));
    for (size_t span : {(size_t)1 << 24, (size_t)1 << 27}) {
        for (int stride : {320, 2048}) {
            for (int wgs : {512, 1024}) {
                run<4>(src, span, stride, wgs, sink);
                run<8>(src, span, stride, wgs, sink);
                run<16>(src, span, stride, wgs, sink);
                if (stride >= 1024) run<64>(src, span, stride, wgs, sink);
            }
        }
    }
    return 0;
}
