// Pixel-major resampling: bilinear (align_corners=True) and legacy-nearest up-sampling, k x k average
// pooling — forward and backward.  All kernels are gathers (no atomics): the backward of an up-sample
// loops over the footprint of each SOURCE pixel, so results are bitwise reproducible and the channel dim
// stays the coalesced one.  HBM-bound; one thread per (pixel, channel).
#include "common.h"

namespace {

enum { MODE_BILINEAR_AC = 0, MODE_NEAREST = 1 };

__device__ __forceinline__ float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

template <typename T>
__global__ void resample_fwd_kernel(const T *__restrict__ x, T *__restrict__ y, int B, int Hs, int Ws, int Ho, int Wo, int C,
                                    int mode) {
    const int64_t total = (int64_t)B * Ho * Wo * C;
    const float sh = mode == MODE_BILINEAR_AC ? ac_scale(Hs, Ho) : (float)Hs / (float)Ho;
    const float sw = mode == MODE_BILINEAR_AC ? ac_scale(Ws, Wo) : (float)Ws / (float)Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const T *xb = x + (size_t)b * Hs * Ws * C + c;
        float v;
        if (mode == MODE_NEAREST) {
            const int iy = min((int)floorf(oy * sh), Hs - 1), ix = min((int)floorf(ox * sw), Ws - 1);
            v = to_f32(xb[((size_t)iy * Ws + ix) * C]);
        } else {
            const float fy = sh * oy, fx = sw * ox;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < Hs - 1), x1 = x0 + (x0 < Ws - 1);
            const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
            v = hy * (hx * to_f32(xb[((size_t)y0 * Ws + x0) * C]) + lx * to_f32(xb[((size_t)y0 * Ws + x1) * C])) +
                ly * (hx * to_f32(xb[((size_t)y1 * Ws + x0) * C]) + lx * to_f32(xb[((size_t)y1 * Ws + x1) * C]));
        }
        y[i] = from_f32<T>(v);
    }
}

// weight with which output coordinate o (of `out`) reads source coordinate s (of `in`)
__device__ __forceinline__ float tap_weight(int o, int s, int in, float scale, int mode) {
    if (mode == MODE_NEAREST) return min((int)floorf(o * scale), in - 1) == s ? 1.f : 0.f;
    const float f = scale * o;
    const int i0 = (int)f, i1 = i0 + (i0 < in - 1);
    const float l = f - i0;
    return (i0 == s ? 1.f - l : 0.f) + (i1 == s ? l : 0.f);
}

template <typename T>
__global__ void resample_bwd_kernel(const T *__restrict__ gy, T *__restrict__ gx, int B, int Hs, int Ws, int Ho, int Wo, int C,
                                    int mode) {
    const int64_t total = (int64_t)B * Hs * Ws * C;
    const float sh = mode == MODE_BILINEAR_AC ? ac_scale(Hs, Ho) : (float)Hs / (float)Ho;
    const float sw = mode == MODE_BILINEAR_AC ? ac_scale(Ws, Wo) : (float)Ws / (float)Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int sx = (int)(r % Ws);
        r /= Ws;
        const int sy = (int)(r % Hs);
        const int b = (int)(r / Hs);
        // conservative footprint: outputs whose source coordinate lies in (s-1, s+1) [bilinear] / [s, s+1) [nearest]
        int ylo, yhi, xlo, xhi;
        if (sh > 0.f) {
            ylo = max(0, (int)floorf((sy - 1) / sh) - 1);
            yhi = min(Ho - 1, (int)ceilf((sy + 1) / sh) + 1);
        } else { ylo = 0; yhi = Ho - 1; }
        if (sw > 0.f) {
            xlo = max(0, (int)floorf((sx - 1) / sw) - 1);
            xhi = min(Wo - 1, (int)ceilf((sx + 1) / sw) + 1);
        } else { xlo = 0; xhi = Wo - 1; }
        const T *gb = gy + (size_t)b * Ho * Wo * C + c;
        float acc = 0.f;
        for (int oy = ylo; oy <= yhi; ++oy) {
            const float wy = tap_weight(oy, sy, Hs, sh, mode);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int ox = xlo; ox <= xhi; ++ox) {
                const float wx = tap_weight(ox, sx, Ws, sw, mode);
                if (wx != 0.f) row += wx * to_f32(gb[((size_t)oy * Wo + ox) * C]);
            }
            acc += wy * row;
        }
        gx[i] = from_f32<T>(acc);
    }
}

template <typename T>
__global__ void avgpool_fwd_kernel(const T *__restrict__ x, T *__restrict__ y, int B, int H, int W, int C, int k) {
    const int Ho = H / k, Wo = W / k;
    const int64_t total = (int64_t)B * Ho * Wo * C;
    const float inv = 1.0f / (float)(k * k);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const T *xb = x + (((size_t)b * H + (size_t)oy * k) * W + (size_t)ox * k) * C + c;
        float s = 0.f;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) s += to_f32(xb[((size_t)dy * W + dx) * C]);
        y[i] = from_f32<T>(s * inv);
    }
}

template <typename T>
__global__ void avgpool_bwd_kernel(const T *__restrict__ gy, T *__restrict__ gx, int B, int H, int W, int C, int k) {
    const int Ho = H / k, Wo = W / k;
    const int64_t total = (int64_t)B * H * W * C;
    const float inv = 1.0f / (float)(k * k);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int x = (int)(r % W);
        r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        const int oy = y / k, ox = x / k;
        float v = 0.f;
        if (oy < Ho && ox < Wo) v = to_f32(gy[(((size_t)b * Ho + oy) * Wo + ox) * C + c]) * inv;
        gx[i] = from_f32<T>(v);
    }
}

inline int flat_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

#define RS_DISPATCH(KERNEL, total, ...)                                                                     \
    if (dtype == GWD_BF16) KERNEL<__bf16><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>(__VA_ARGS__);  \
    else if (dtype == GWD_F32) KERNEL<float><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>(__VA_ARGS__); \
    else return -2;                                                                                         \
    GWD_CHECK_LAUNCH();                                                                                     \
    return 0;

extern "C" int gwd_resample_forward(const void *x, void *y, int32_t B, int32_t Hs, int32_t Ws, int32_t Ho, int32_t Wo,
                                    int32_t C, int32_t mode, int32_t dtype, void *stream) {
    if (!x || !y || B <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || mode < 0 || mode > 1) return -1;
    const int64_t total = (int64_t)B * Ho * Wo * C;
    if (dtype == GWD_BF16) resample_fwd_kernel<__bf16><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const __bf16 *)x, (__bf16 *)y, B, Hs, Ws, Ho, Wo, C, mode);
    else if (dtype == GWD_F32) resample_fwd_kernel<float><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const float *)x, (float *)y, B, Hs, Ws, Ho, Wo, C, mode);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_resample_backward(const void *gy, void *gx, int32_t B, int32_t Hs, int32_t Ws, int32_t Ho, int32_t Wo,
                                     int32_t C, int32_t mode, int32_t dtype, void *stream) {
    if (!gy || !gx || B <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || mode < 0 || mode > 1) return -1;
    const int64_t total = (int64_t)B * Hs * Ws * C;
    if (dtype == GWD_BF16) resample_bwd_kernel<__bf16><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const __bf16 *)gy, (__bf16 *)gx, B, Hs, Ws, Ho, Wo, C, mode);
    else if (dtype == GWD_F32) resample_bwd_kernel<float><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const float *)gy, (float *)gx, B, Hs, Ws, Ho, Wo, C, mode);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_avgpool_forward(const void *x, void *y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                                   int32_t dtype, void *stream) {
    if (!x || !y || B <= 0 || H < k || W < k || C <= 0 || k <= 0) return -1;
    const int64_t total = (int64_t)B * (H / k) * (W / k) * C;
    if (dtype == GWD_BF16) avgpool_fwd_kernel<__bf16><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const __bf16 *)x, (__bf16 *)y, B, H, W, C, k);
    else if (dtype == GWD_F32) avgpool_fwd_kernel<float><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const float *)x, (float *)y, B, H, W, C, k);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_avgpool_backward(const void *gy, void *gx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                                    int32_t dtype, void *stream) {
    if (!gy || !gx || B <= 0 || H < k || W < k || C <= 0 || k <= 0) return -1;
    const int64_t total = (int64_t)B * H * W * C;
    if (dtype == GWD_BF16) avgpool_bwd_kernel<__bf16><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const __bf16 *)gy, (__bf16 *)gx, B, H, W, C, k);
    else if (dtype == GWD_F32) avgpool_bwd_kernel<float><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const float *)gy, (float *)gx, B, H, W, C, k);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}
