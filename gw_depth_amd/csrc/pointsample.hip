// Sampling a pixel-major feature map (B, H, W, C) at S normalised points per image - torch.nn.functional.grid_sample
// (align_corners=False, padding_mode='zeros') for a grid of shape (B, S, 1, 2), as the reference uses it to read
// reference-point features / anchor depths (points_sample.py:262-268, multiscale_transformerr.py:688-691).
//   forward : out (B, S, C) fp32 = bilinear (mode 0) or nearest (mode 1) sample, arithmetic in fp32 in ATen's order
//   backward: gmap (B, H, W, C) (pre-zeroed by the caller, map dtype) += scatter of gout; one thread owns one
//             (image, channel) and walks the S points in order, so colliding points need no atomics.
// No gradient w.r.t. the coordinates (the callers' points are index-valued / detached; nearest has none).
#include "common.h"

namespace {

__device__ __forceinline__ float unnormalize(float coord, int size) { return ((coord + 1.f) * size - 1.f) / 2.f; }

template <typename T>
__global__ void point_sample_fwd_kernel(const T *__restrict__ map, const float *__restrict__ coords, float *__restrict__ out,
                                        int B, int H, int W, int C, int S, int mode) {
    const int64_t total = (int64_t)B * S * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t bs = i / C;
        const int b = (int)(bs / S);
        const float ix = unnormalize(coords[bs * 2], W), iy = unnormalize(coords[bs * 2 + 1], H);
        const T *img = map + (int64_t)b * H * W * C + c;
        float r = 0.f;
        if (mode == 1) {
            const int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
            if (xn >= 0 && xn < W && yn >= 0 && yn < H) r = to_f32(img[((int64_t)yn * W + xn) * C]);
        } else {
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
            const float nw = ((float)x1 - ix) * ((float)y1 - iy), ne = (ix - (float)x0) * ((float)y1 - iy);
            const float sw = ((float)x1 - ix) * (iy - (float)y0), se = (ix - (float)x0) * (iy - (float)y0);
            const bool x0ok = x0 >= 0 && x0 < W, x1ok = x1 >= 0 && x1 < W, y0ok = y0 >= 0 && y0 < H, y1ok = y1 >= 0 && y1 < H;
            if (y0ok && x0ok) r += to_f32(img[((int64_t)y0 * W + x0) * C]) * nw;
            if (y0ok && x1ok) r += to_f32(img[((int64_t)y0 * W + x1) * C]) * ne;
            if (y1ok && x0ok) r += to_f32(img[((int64_t)y1 * W + x0) * C]) * sw;
            if (y1ok && x1ok) r += to_f32(img[((int64_t)y1 * W + x1) * C]) * se;
        }
        out[i] = r;
    }
}

template <typename T>
__global__ void point_sample_bwd_kernel(const float *__restrict__ gout, const float *__restrict__ coords, T *__restrict__ gmap,
                                        int B, int H, int W, int C, int S, int mode) {
    const int64_t total = (int64_t)B * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), b = (int)(i / C);
        T *img = gmap + (int64_t)b * H * W * C + c;
        for (int s = 0; s < S; ++s) {
            const int64_t bs = (int64_t)b * S + s;
            const float g = gout[bs * C + c];
            const float ix = unnormalize(coords[bs * 2], W), iy = unnormalize(coords[bs * 2 + 1], H);
            if (mode == 1) {
                const int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
                if (xn >= 0 && xn < W && yn >= 0 && yn < H) {
                    T *p = img + ((int64_t)yn * W + xn) * C;
                    *p = from_f32<T>(to_f32(*p) + g);
                }
            } else {
                const float fx = floorf(ix), fy = floorf(iy);
                const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
                const float wx1 = ix - (float)x0, wx0 = (float)x1 - ix, wy1 = iy - (float)y0, wy0 = (float)y1 - iy;
                const bool x0ok = x0 >= 0 && x0 < W, x1ok = x1 >= 0 && x1 < W, y0ok = y0 >= 0 && y0 < H, y1ok = y1 >= 0 && y1 < H;
                if (y0ok && x0ok) { T *p = img + ((int64_t)y0 * W + x0) * C; *p = from_f32<T>(to_f32(*p) + g * (wx0 * wy0)); }
                if (y0ok && x1ok) { T *p = img + ((int64_t)y0 * W + x1) * C; *p = from_f32<T>(to_f32(*p) + g * (wx1 * wy0)); }
                if (y1ok && x0ok) { T *p = img + ((int64_t)y1 * W + x0) * C; *p = from_f32<T>(to_f32(*p) + g * (wx0 * wy1)); }
                if (y1ok && x1ok) { T *p = img + ((int64_t)y1 * W + x1) * C; *p = from_f32<T>(to_f32(*p) + g * (wx1 * wy1)); }
            }
        }
    }
}

}  // namespace

extern "C" int gwd_point_sample_forward(const void *map, const float *coords, float *out, int32_t B, int32_t H, int32_t W,
                                        int32_t C, int32_t S, int32_t mode, int32_t dtype, void *stream) {
    if (!map || !coords || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || S <= 0 || (mode != 0 && mode != 1)) return -1;
    const int64_t total = (int64_t)B * S * C;
    int64_t nb = (total + 255) / 256;
    const int grid = (int)(nb > 4096 ? 4096 : nb);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GWD_BF16) point_sample_fwd_kernel<__bf16><<<grid, 256, 0, st>>>((const __bf16 *)map, coords, out, B, H, W, C, S, mode);
    else if (dtype == GWD_F32) point_sample_fwd_kernel<float><<<grid, 256, 0, st>>>((const float *)map, coords, out, B, H, W, C, S, mode);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_point_sample_backward(const float *gout, const float *coords, void *gmap, int32_t B, int32_t H, int32_t W,
                                         int32_t C, int32_t S, int32_t mode, int32_t dtype, void *stream) {
    if (!gout || !coords || !gmap || B <= 0 || H <= 0 || W <= 0 || C <= 0 || S <= 0 || (mode != 0 && mode != 1)) return -1;
    const int64_t total = (int64_t)B * C;
    int64_t nb = (total + 63) / 64;
    const int grid = (int)(nb > 4096 ? 4096 : nb);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GWD_BF16) point_sample_bwd_kernel<__bf16><<<grid, 64, 0, st>>>(gout, coords, (__bf16 *)gmap, B, H, W, C, S, mode);
    else if (dtype == GWD_F32) point_sample_bwd_kernel<float><<<grid, 64, 0, st>>>(gout, coords, (float *)gmap, B, H, W, C, S, mode);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}
