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

// ---- 16-byte vector versions (C a multiple of the vector width) --------------------------------------------------
template <int VB> struct RsRaw;
template <> struct RsRaw<16> { typedef uint4 type; };
template <> struct RsRaw<8> { typedef uint2 type; };

// Nearest-mode backward, 16/8-byte channel vectors: the footprint of source pixel (sy, sx) is the set of outputs whose
// legacy-nearest source is (sy, sx) - at most ceil(Ho/Hs)+1 candidates per axis, tested with the forward's own rule.
// (gradient of the virtually up-sampled convolution inputs of the decoder: 314 MB per call at full resolution)
template <typename T, int VB>
__global__ void resample_nearest_bwd_vec_kernel(const T *__restrict__ gy, T *__restrict__ gx, int B, int Hs, int Ws, int Ho, int Wo, int C) {
    constexpr int VEC = VB / (int)sizeof(T);
    typedef typename RsRaw<VB>::type Raw;
    const int CV = C / VEC;
    const int64_t total = (int64_t)B * Hs * Ws * CV;
    const float sh = (float)Hs / (float)Ho, sw = (float)Ws / (float)Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;
        const int sx = (int)(r % Ws);
        r /= Ws;
        const int sy = (int)(r % Hs);
        const int b = (int)(r / Hs);
        const int ylo = max(0, (int)floorf(sy / sh) - 1), yhi = min(Ho - 1, (int)ceilf((sy + 1) / sh) + 1);
        const int xlo = max(0, (int)floorf(sx / sw) - 1), xhi = min(Wo - 1, (int)ceilf((sx + 1) / sw) + 1);
        const T *gb = gy + (size_t)b * Ho * Wo * C + cv * VEC;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int oy = ylo; oy <= yhi; ++oy) {
            if (tap_weight(oy, sy, Hs, sh, MODE_NEAREST) == 0.f) continue;
            for (int ox = xlo; ox <= xhi; ++ox) {
                if (tap_weight(ox, sx, Ws, sw, MODE_NEAREST) == 0.f) continue;
                const Raw raw = *(const Raw *)(gb + ((size_t)oy * Wo + ox) * C);
                const T *p = (const T *)&raw;
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += to_f32(p[e]);
            }
        }
        alignas(16) T out[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) out[e] = from_f32<T>(acc[e]);
        *(Raw *)(gx + (size_t)i * VEC) = *(const Raw *)out;
    }
}

template <typename T, int VB>
__global__ void resample_fwd_vec_kernel(const T *__restrict__ x, T *__restrict__ y, int B, int Hs, int Ws, int Ho, int Wo, int C,
                                        int mode, int ldy) {
    constexpr int VEC = VB / (int)sizeof(T);
    typedef typename RsRaw<VB>::type Raw;
    const int CV = C / VEC;
    const int64_t total = (int64_t)B * Ho * Wo * CV;
    const float sh = mode == MODE_BILINEAR_AC ? ac_scale(Hs, Ho) : (float)Hs / (float)Ho;
    const float sw = mode == MODE_BILINEAR_AC ? ac_scale(Ws, Wo) : (float)Ws / (float)Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const T *xb = x + (size_t)b * Hs * Ws * C + cv * VEC;
        alignas(16) T out[VEC];
        if (mode == MODE_NEAREST) {
            const int iy = min((int)floorf(oy * sh), Hs - 1), ix = min((int)floorf(ox * sw), Ws - 1);
            *(Raw *)out = *(const Raw *)(xb + ((size_t)iy * Ws + ix) * C);
        } else {
            const float fy = sh * oy, fx = sw * ox;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < Hs - 1), x1 = x0 + (x0 < Ws - 1);
            const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
            const Raw r00 = *(const Raw *)(xb + ((size_t)y0 * Ws + x0) * C), r01 = *(const Raw *)(xb + ((size_t)y0 * Ws + x1) * C);
            const Raw r10 = *(const Raw *)(xb + ((size_t)y1 * Ws + x0) * C), r11 = *(const Raw *)(xb + ((size_t)y1 * Ws + x1) * C);
            const T *p00 = (const T *)&r00, *p01 = (const T *)&r01, *p10 = (const T *)&r10, *p11 = (const T *)&r11;
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                out[e] = from_f32<T>(hy * (hx * to_f32(p00[e]) + lx * to_f32(p01[e])) + ly * (hx * to_f32(p10[e]) + lx * to_f32(p11[e])));
        }
        *(Raw *)(y + (((size_t)b * Ho + oy) * Wo + ox) * ldy + cv * VEC) = *(const Raw *)out;       // ldy >= C: a channel slice of a wider map
    }
}

// Backward of an up-sample, separable: pass 1 reduces the output gradient along x into tmp[b][oy][sx][c] (fp32),
// pass 2 reduces tmp along y.  Same summation order as the single-pass gather (rows, then the weighted row sums), but
// a source pixel's footprint of (2 ry + 2)(2 rx + 2) taps becomes two loops of 2 r + 2 - at the 16x pyramid branches
// that is 1 300 taps per thread over 90 K threads versus 35 taps over 190 K wide threads.
template <typename T, int VB>
__global__ void resample_bwd_x_kernel(const T *__restrict__ gy, float *__restrict__ tmp, int B, int Ws, int Ho, int Wo, int C, int mode, int ldg) {
    constexpr int VEC = VB / (int)sizeof(T);
    typedef typename RsRaw<VB>::type Raw;
    const int CV = C / VEC;
    const int64_t total = (int64_t)B * Ho * Ws * CV;
    const float sw = mode == MODE_BILINEAR_AC ? ac_scale(Ws, Wo) : (float)Ws / (float)Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;
        const int sx = (int)(r % Ws);
        r /= Ws;                                            // r = b * Ho + oy
        int xlo, xhi;
        if (sw > 0.f) {
            xlo = max(0, (int)floorf((sx - 1) / sw) - 1);
            xhi = min(Wo - 1, (int)ceilf((sx + 1) / sw) + 1);
        } else { xlo = 0; xhi = Wo - 1; }
        const T *row = gy + (size_t)r * Wo * ldg + cv * VEC;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int ox = xlo; ox <= xhi; ++ox) {
            const float wx = tap_weight(ox, sx, Ws, sw, mode);
            if (wx != 0.f) {
                const Raw raw = *(const Raw *)(row + (size_t)ox * ldg);
                const T *p = (const T *)&raw;
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += wx * to_f32(p[e]);
            }
        }
        float *o = tmp + ((size_t)r * Ws + sx) * C + cv * VEC;
#pragma unroll
        for (int e = 0; e < VEC; e += 4) *(float4 *)(o + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
    }
}

template <typename T>
__global__ void resample_bwd_y_kernel(const float *__restrict__ tmp, T *__restrict__ gx, int B, int Hs, int Ws, int Ho, int C, int mode) {
    const int C4 = C / 4;
    const int64_t total = (int64_t)B * Hs * Ws * C4;
    const float sh = mode == MODE_BILINEAR_AC ? ac_scale(Hs, Ho) : (float)Hs / (float)Ho;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        int64_t r = i / C4;
        const int sx = (int)(r % Ws);
        r /= Ws;
        const int sy = (int)(r % Hs);
        const int b = (int)(r / Hs);
        int ylo, yhi;
        if (sh > 0.f) {
            ylo = max(0, (int)floorf((sy - 1) / sh) - 1);
            yhi = min(Ho - 1, (int)ceilf((sy + 1) / sh) + 1);
        } else { ylo = 0; yhi = Ho - 1; }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oy = ylo; oy <= yhi; ++oy) {
            const float wy = tap_weight(oy, sy, Hs, sh, mode);
            if (wy != 0.f) {
                const float4 v = *(const float4 *)(tmp + (((size_t)b * Ho + oy) * Ws + sx) * C + c4 * 4);
                acc.x += wy * v.x; acc.y += wy * v.y; acc.z += wy * v.z; acc.w += wy * v.w;
            }
        }
        T *o = gx + (((size_t)b * Hs + sy) * Ws + sx) * C + c4 * 4;
        o[0] = from_f32<T>(acc.x); o[1] = from_f32<T>(acc.y); o[2] = from_f32<T>(acc.z); o[3] = from_f32<T>(acc.w);
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

// vector forms of the k x k average pool (C a multiple of VB bytes)
template <typename T, int VB>
__global__ void avgpool_fwd_vec_kernel(const T *__restrict__ x, T *__restrict__ y, int B, int H, int W, int C, int k) {
    constexpr int VEC = VB / (int)sizeof(T);
    typedef typename RsRaw<VB>::type Raw;
    const int Ho = H / k, Wo = W / k, CV = C / VEC;
    const int64_t total = (int64_t)B * Ho * Wo * CV;
    const float inv = 1.0f / (float)(k * k);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const T *xb = x + (((size_t)b * H + (size_t)oy * k) * W + (size_t)ox * k) * C + cv * VEC;
        float s[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] = 0.f;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) {
                const Raw raw = *(const Raw *)(xb + ((size_t)dy * W + dx) * C);
                const T *p = (const T *)&raw;
#pragma unroll
                for (int e = 0; e < VEC; ++e) s[e] += to_f32(p[e]);
            }
        alignas(16) T out[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) out[e] = from_f32<T>(s[e] * inv);
        *(Raw *)(y + i * VEC) = *(const Raw *)out;
    }
}

template <typename T, int VB>
__global__ void avgpool_bwd_vec_kernel(const T *__restrict__ gy, T *__restrict__ gx, int B, int H, int W, int C, int k) {
    constexpr int VEC = VB / (int)sizeof(T);
    typedef typename RsRaw<VB>::type Raw;
    const int Ho = H / k, Wo = W / k, CV = C / VEC;
    const int64_t total = (int64_t)B * H * W * CV;
    const float inv = 1.0f / (float)(k * k);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;
        const int xx = (int)(r % W);
        r /= W;
        const int yy = (int)(r % H);
        const int b = (int)(r / H);
        const int oy = yy / k, ox = xx / k;
        alignas(16) T out[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) out[e] = from_f32<T>(0.f);
        if (oy < Ho && ox < Wo) {
            const Raw raw = *(const Raw *)(gy + (((size_t)b * Ho + oy) * Wo + ox) * C + cv * VEC);
            const T *p = (const T *)&raw;
#pragma unroll
            for (int e = 0; e < VEC; ++e) out[e] = from_f32<T>(to_f32(p[e]) * inv);
        }
        *(Raw *)(gx + i * VEC) = *(const Raw *)out;
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
                                    int32_t C, int32_t mode, int32_t ldy, int32_t dtype, void *stream) {
    if (!x || !y || B <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || mode < 0 || mode > 1) return -1;
    if (ldy == 0) ldy = C;
    if (ldy < C) return -1;
    if (ldy != C && (C % 4 || ldy % 4 || (dtype == GWD_BF16 && C % 8 == 0 && ldy % 8))) return -4;     // pitched output: vector kernels only
    const int64_t total = (int64_t)B * Ho * Wo * C;
    if (dtype == GWD_BF16 && C % 4 == 0) {
        if (C % 8 == 0) resample_fwd_vec_kernel<__bf16, 16><<<flat_grid(total / 8), 256, 0, (hipStream_t)stream>>>((const __bf16 *)x, (__bf16 *)y, B, Hs, Ws, Ho, Wo, C, mode, ldy);
        else resample_fwd_vec_kernel<__bf16, 8><<<flat_grid(total / 4), 256, 0, (hipStream_t)stream>>>((const __bf16 *)x, (__bf16 *)y, B, Hs, Ws, Ho, Wo, C, mode, ldy);
        GWD_CHECK_LAUNCH();
        return 0;
    }
    if (dtype == GWD_F32 && C % 4 == 0) {
        resample_fwd_vec_kernel<float, 16><<<flat_grid(total / 4), 256, 0, (hipStream_t)stream>>>((const float *)x, (float *)y, B, Hs, Ws, Ho, Wo, C, mode, ldy);
        GWD_CHECK_LAUNCH();
        return 0;
    }
    if (ldy != C) return -4;
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
    if (mode == MODE_NEAREST && dtype == GWD_BF16 && C % 4 == 0) {
        hipStream_t s = (hipStream_t)stream;
        if (C % 8 == 0) resample_nearest_bwd_vec_kernel<__bf16, 16><<<flat_grid(total / 8), 256, 0, s>>>((const __bf16 *)gy, (__bf16 *)gx, B, Hs, Ws, Ho, Wo, C);
        else resample_nearest_bwd_vec_kernel<__bf16, 8><<<flat_grid(total / 4), 256, 0, s>>>((const __bf16 *)gy, (__bf16 *)gx, B, Hs, Ws, Ho, Wo, C);
        GWD_CHECK_LAUNCH();
        return 0;
    }
    if (mode == MODE_NEAREST && dtype == GWD_F32 && C % 4 == 0) {
        resample_nearest_bwd_vec_kernel<float, 16><<<flat_grid(total / 4), 256, 0, (hipStream_t)stream>>>((const float *)gy, (float *)gx, B, Hs, Ws, Ho, Wo, C);
        GWD_CHECK_LAUNCH();
        return 0;
    }
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
    if (dtype == GWD_BF16 && C % 4 == 0) {
        if (C % 8 == 0) avgpool_fwd_vec_kernel<__bf16, 16><<<flat_grid(total / 8), 256, 0, (hipStream_t)stream>>>((const __bf16 *)x, (__bf16 *)y, B, H, W, C, k);
        else avgpool_fwd_vec_kernel<__bf16, 8><<<flat_grid(total / 4), 256, 0, (hipStream_t)stream>>>((const __bf16 *)x, (__bf16 *)y, B, H, W, C, k);
        GWD_CHECK_LAUNCH();
        return 0;
    }
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
    if (dtype == GWD_BF16 && C % 4 == 0) {
        if (C % 8 == 0) avgpool_bwd_vec_kernel<__bf16, 16><<<flat_grid(total / 8), 256, 0, (hipStream_t)stream>>>((const __bf16 *)gy, (__bf16 *)gx, B, H, W, C, k);
        else avgpool_bwd_vec_kernel<__bf16, 8><<<flat_grid(total / 4), 256, 0, (hipStream_t)stream>>>((const __bf16 *)gy, (__bf16 *)gx, B, H, W, C, k);
        GWD_CHECK_LAUNCH();
        return 0;
    }
    if (dtype == GWD_BF16) avgpool_bwd_kernel<__bf16><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const __bf16 *)gy, (__bf16 *)gx, B, H, W, C, k);
    else if (dtype == GWD_F32) avgpool_bwd_kernel<float><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const float *)gy, (float *)gx, B, H, W, C, k);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_resample_backward_sep(const void *gy, float *tmp, void *gx, int32_t B, int32_t Hs, int32_t Ws, int32_t Ho,
                                         int32_t Wo, int32_t C, int32_t mode, int32_t ldg, int32_t dtype, void *stream) {
    if (!gy || !tmp || !gx || B <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || mode < 0 || mode > 1) return -1;
    if (dtype != GWD_BF16 && dtype != GWD_F32) return -2;
    if (ldg == 0) ldg = C;
    if (ldg < C) return -1;
    if (C % 4 || ldg % 4 || (dtype == GWD_BF16 && C % 8 == 0 && ldg % 8)) return -4;       // caller uses gwd_resample_backward
    hipStream_t st = (hipStream_t)stream;
    const int vec = (dtype == GWD_BF16 && C % 8 == 0) ? 8 : 4;
    const int64_t t1 = (int64_t)B * Ho * Ws * (C / vec), t2 = (int64_t)B * Hs * Ws * (C / 4);
    if (dtype == GWD_BF16) {
        if (vec == 8) resample_bwd_x_kernel<__bf16, 16><<<flat_grid(t1), 256, 0, st>>>((const __bf16 *)gy, tmp, B, Ws, Ho, Wo, C, mode, ldg);
        else resample_bwd_x_kernel<__bf16, 8><<<flat_grid(t1), 256, 0, st>>>((const __bf16 *)gy, tmp, B, Ws, Ho, Wo, C, mode, ldg);
        resample_bwd_y_kernel<__bf16><<<flat_grid(t2), 256, 0, st>>>(tmp, (__bf16 *)gx, B, Hs, Ws, Ho, C, mode);
    } else {
        resample_bwd_x_kernel<float, 16><<<flat_grid(t1), 256, 0, st>>>((const float *)gy, tmp, B, Ws, Ho, Wo, C, mode, ldg);
        resample_bwd_y_kernel<float><<<flat_grid(t2), 256, 0, st>>>(tmp, (float *)gx, B, Hs, Ws, Ho, C, mode);
    }
    GWD_CHECK_LAUNCH();
    return 0;
}
