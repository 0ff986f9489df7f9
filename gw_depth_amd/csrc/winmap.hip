// Window partition / reverse for the Swin-style stages as ONE index-remapping copy each way:
//   gather : (B, H, W, C) token map -> (B * nWin, 49, C) windows, including zero padding to a multiple of 7 and the
//            cyclic shift (F.pad + torch.roll + view/permute/contiguous of multiscale_transformerr.py:667-676,705-707)
//   scatter: the inverse (window_reverse + un-shift + crop, :730-747).
// Each is the other's backward.  16-byte channel vectors; one thread per (window token, channel vector).
#include "common.h"

namespace {

// 16 bytes of T plus 16 bytes of T, element-wise in fp32
template <typename T> __device__ __forceinline__ uint4 add_vec(const uint4 &a, const uint4 &b) {
    constexpr int N = 16 / sizeof(T);
    uint4 r;
    const T *pa = (const T *)&a, *pb = (const T *)&b;
    T *pr = (T *)&r;
#pragma unroll
    for (int e = 0; e < N; ++e) pr[e] = from_f32<T>(to_f32(pa[e]) + to_f32(pb[e]));
    return r;
}

template <bool GATHER, typename T>
__global__ void winmap_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, const uint4 *__restrict__ res, int B, int H, int W,
                              int CV, int shift) {
    const int Hp = (H + 6) / 7 * 7, Wp = (W + 6) / 7 * 7, nwx = Wp / 7, nwy = Hp / 7;
    const int64_t total = (int64_t)B * Hp * Wp * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;                           // window-major token index
        const int t = (int)(r % 49);
        r /= 49;
        const int wx = (int)(r % nwx);
        r /= nwx;
        const int wy = (int)(r % nwy);
        const int b = (int)(r / nwy);
        int py = wy * 7 + t / 7 + shift, px = wx * 7 + t % 7 + shift;      // position in the padded, un-shifted map
        if (py >= Hp) py -= Hp;
        if (px >= Wp) px -= Wp;
        const bool inside = py < H && px < W;
        const int64_t map_idx = (((int64_t)b * H + py) * W + px) * CV + cv;
        if (GATHER) {
            dst[i] = inside ? src[map_idx] : make_uint4(0u, 0u, 0u, 0u);
        } else if (inside) {
            dst[map_idx] = res ? add_vec<T>(src[i], res[map_idx]) : src[i];     // window reverse (+ the block's residual stream)
        }
    }
}

// up to GWD_WINMAP_JOBS maps of the same geometry (different channel counts) in ONE launch: job = blockIdx.y.  A Swin block with class
// tokens partitions / reverses three maps (features, depth tokens, seg tokens) at every hand-over; as separate launches each of the
// small ones cost a full dependent-launch slot (~6 us) for a few hundred KB.
struct WinJobs {
    const uint4 *src[GWD_WINMAP_JOBS];
    uint4 *dst[GWD_WINMAP_JOBS];
    const uint4 *res[GWD_WINMAP_JOBS];
    int cv[GWD_WINMAP_JOBS];
};
template <bool GATHER, typename T>
__global__ void winmap_multi_kernel(const WinJobs j, int B, int H, int W, int shift) {
    const int job = blockIdx.y, CV = j.cv[job];
    const uint4 *__restrict__ src = j.src[job];
    uint4 *__restrict__ dst = j.dst[job];
    const uint4 *__restrict__ res = j.res[job];
    const int Hp = (H + 6) / 7 * 7, Wp = (W + 6) / 7 * 7, nwx = Wp / 7, nwy = Hp / 7;
    const int64_t total = (int64_t)B * Hp * Wp * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;
        const int t = (int)(r % 49);
        r /= 49;
        const int wx = (int)(r % nwx);
        r /= nwx;
        const int wy = (int)(r % nwy);
        const int b = (int)(r / nwy);
        int py = wy * 7 + t / 7 + shift, px = wx * 7 + t % 7 + shift;
        if (py >= Hp) py -= Hp;
        if (px >= Wp) px -= Wp;
        const bool inside = py < H && px < W;
        const int64_t map_idx = (((int64_t)b * H + py) * W + px) * CV + cv;
        if (GATHER) {
            dst[i] = inside ? src[map_idx] : make_uint4(0u, 0u, 0u, 0u);
        } else if (inside) {
            dst[map_idx] = res ? add_vec<T>(src[i], res[map_idx]) : src[i];
        }
    }
}

}  // namespace

extern "C" int gwd_window_map_multi(const void *const *src, void *const *dst, const void *const *residual, const int32_t *C, int32_t n,
                                    int32_t B, int32_t H, int32_t W, int32_t shift, int32_t gather, int32_t dtype, void *stream) {
    if (!src || !dst || !C || n <= 0 || n > GWD_WINMAP_JOBS || B <= 0 || H <= 0 || W <= 0 || shift < 0 || shift >= 7) return -1;
    const int esz = dtype == GWD_BF16 ? 2 : (dtype == GWD_F32 ? 4 : 0);
    if (!esz) return -2;
    const int Hp = (H + 6) / 7 * 7, Wp = (W + 6) / 7 * 7;
    WinJobs j;
    int64_t most = 0;
    for (int i = 0; i < n; ++i) {
        if (!src[i] || !dst[i] || C[i] <= 0) return -1;
        if (gather && residual && residual[i]) return -1;
        if ((C[i] * esz) % 16) return -4;
        j.src[i] = (const uint4 *)src[i];
        j.dst[i] = (uint4 *)dst[i];
        j.res[i] = residual ? (const uint4 *)residual[i] : nullptr;
        j.cv[i] = C[i] * esz / 16;
        const int64_t total = (int64_t)B * Hp * Wp * j.cv[i];
        most = total > most ? total : most;
    }
    const int64_t nb = (most + 255) / 256;
    const dim3 grid((unsigned)(nb > 8192 ? 8192 : nb), (unsigned)n);
    if (gather) winmap_multi_kernel<true, float><<<grid, 256, 0, (hipStream_t)stream>>>(j, B, H, W, shift);
    else if (dtype == GWD_BF16) winmap_multi_kernel<false, __bf16><<<grid, 256, 0, (hipStream_t)stream>>>(j, B, H, W, shift);
    else winmap_multi_kernel<false, float><<<grid, 256, 0, (hipStream_t)stream>>>(j, B, H, W, shift);
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_window_map(const void *src, void *dst, const void *residual, int32_t B, int32_t H, int32_t W, int32_t C, int32_t shift,
                              int32_t gather, int32_t dtype, void *stream) {
    if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || C <= 0 || shift < 0 || shift >= 7) return -1;
    if (gather && residual) return -1;
    const int esz = dtype == GWD_BF16 ? 2 : (dtype == GWD_F32 ? 4 : 0);
    if (!esz) return -2;
    if ((C * esz) % 16) return -4;
    const int CV = C * esz / 16;
    const int Hp = (H + 6) / 7 * 7, Wp = (W + 6) / 7 * 7;
    const int64_t total = (int64_t)B * Hp * Wp * CV;
    int64_t nb = (total + 255) / 256;
    const int grid = (int)(nb > 8192 ? 8192 : nb);
    const uint4 *r4 = (const uint4 *)residual;
    if (gather)
        winmap_kernel<true, float><<<grid, 256, 0, (hipStream_t)stream>>>((const uint4 *)src, (uint4 *)dst, nullptr, B, H, W, CV, shift);
    else if (dtype == GWD_BF16)
        winmap_kernel<false, __bf16><<<grid, 256, 0, (hipStream_t)stream>>>((const uint4 *)src, (uint4 *)dst, r4, B, H, W, CV, shift);
    else
        winmap_kernel<false, float><<<grid, 256, 0, (hipStream_t)stream>>>((const uint4 *)src, (uint4 *)dst, r4, B, H, W, CV, shift);
    GWD_CHECK_LAUNCH();
    return 0;
}
