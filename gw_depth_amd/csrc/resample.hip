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
__global__ void resample_nearest_bwd_vec_kernel(const T *__restrict__ gy, T *__restrict__ gx, int B, int Hs, int Ws, int Ho, int Wo, int C,
                                                const T *__restrict__ gate, int gate_act) {
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
        if (gate) {                                      // the source map is an activation output whose backward runs here (conv2d defer)
            const Raw rg = *(const Raw *)(gate + (size_t)i * VEC);
            const T *pg = (const T *)&rg;
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = gate_grad(acc[e], to_f32(pg[e]), gate_act);
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

// ---------------------------------------------------------------------------------------------------------------------------
// The four average pools of the PSP module (k = 16, 8, 4, 2; points_sample.py:107-113) from ONE pass over the map: a workgroup owns a
// 16 x 16 pixel block, a thread sums a 2 x 2 patch, the coarser levels are sums of four finer sums (fp32, through LDS) - the same
// means as four separate F.avg_pool2d calls (floor semantics: an output exists only where its whole k x k window does).
template <typename T>
__global__ __launch_bounds__(256) void psp_pool_fwd_kernel(const T *__restrict__ x, T *__restrict__ p16, T *__restrict__ p8, T *__restrict__ p4,
                                                           T *__restrict__ p2, int B, int H, int W, int C, int bx_n, int by_n) {
    constexpr int VEC = 16 / (int)sizeof(T);
    __shared__ float s2[64][4][VEC], s4[16][4][VEC], s8[4][4][VEC];
    const int t = threadIdx.x, q = t & 63, cg = t >> 6;       // q = 2x2 patch of the block (8 x 8 of them), cg = channel-vector lane
    int blk = blockIdx.x;
    const int bx = blk % bx_n;
    blk /= bx_n;
    const int by = blk % by_n, b = blk / by_n;
    const int CV = C / VEC;
    const int qy = q >> 3, qx = q & 7;
    const int y0 = by * 16 + qy * 2, x0 = bx * 16 + qx * 2;
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8, H16 = H / 16, W16 = W / 16;
    for (int cv0 = 0; cv0 < CV; cv0 += 4) {
        const int cv = cv0 + cg;
        float a[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] = 0.f;
        if (cv < CV) {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
                    if (y0 + dy < H && x0 + dx < W) {
                        const uint4 raw = *(const uint4 *)(x + (((size_t)b * H + y0 + dy) * W + x0 + dx) * C + cv * VEC);
                        const T *pv = (const T *)&raw;
#pragma unroll
                        for (int e = 0; e < VEC; ++e) a[e] += to_f32(pv[e]);
                    }
            const int oy = y0 >> 1, ox = x0 >> 1;
            if (oy < H2 && ox < W2) {
                alignas(16) T o[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(a[e] * 0.25f);
                *(uint4 *)(p2 + (((size_t)b * H2 + oy) * W2 + ox) * C + cv * VEC) = *(const uint4 *)o;
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) s2[q][cg][e] = a[e];
        __syncthreads();
        if (q < 16) {                                          // 4 x 4 outputs of the 4-pool
            const int ry = q >> 2, rx = q & 3;
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                a[e] = s2[(2 * ry) * 8 + 2 * rx][cg][e] + s2[(2 * ry) * 8 + 2 * rx + 1][cg][e] + s2[(2 * ry + 1) * 8 + 2 * rx][cg][e] +
                       s2[(2 * ry + 1) * 8 + 2 * rx + 1][cg][e];
            const int oy = by * 4 + ry, ox = bx * 4 + rx;
            if (cv < CV && oy < H4 && ox < W4) {
                alignas(16) T o[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(a[e] * (1.f / 16.f));
                *(uint4 *)(p4 + (((size_t)b * H4 + oy) * W4 + ox) * C + cv * VEC) = *(const uint4 *)o;
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) s4[q][cg][e] = a[e];
        }
        __syncthreads();
        if (q < 4) {
            const int ry = q >> 1, rx = q & 1;
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                a[e] = s4[(2 * ry) * 4 + 2 * rx][cg][e] + s4[(2 * ry) * 4 + 2 * rx + 1][cg][e] + s4[(2 * ry + 1) * 4 + 2 * rx][cg][e] +
                       s4[(2 * ry + 1) * 4 + 2 * rx + 1][cg][e];
            const int oy = by * 2 + ry, ox = bx * 2 + rx;
            if (cv < CV && oy < H8 && ox < W8) {
                alignas(16) T o[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(a[e] * (1.f / 64.f));
                *(uint4 *)(p8 + (((size_t)b * H8 + oy) * W8 + ox) * C + cv * VEC) = *(const uint4 *)o;
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) s8[q][cg][e] = a[e];
        }
        __syncthreads();
        if (q == 0 && cv < CV && by < H16 && bx < W16) {
            alignas(16) T o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>((s8[0][cg][e] + s8[1][cg][e] + s8[2][cg][e] + s8[3][cg][e]) * (1.f / 256.f));
            *(uint4 *)(p16 + (((size_t)b * H16 + by) * W16 + bx) * C + cv * VEC) = *(const uint4 *)o;
        }
        __syncthreads();
    }
}

// gx = g_pass (the gradient of the map itself, pixel pitch ldg; may be NULL) + sum over the four levels of g_k[y / k][x / k] / k^2
template <typename T>
__global__ void psp_pool_bwd_kernel(const T *__restrict__ gpass, const T *__restrict__ g16, const T *__restrict__ g8, const T *__restrict__ g4,
                                    const T *__restrict__ g2, T *__restrict__ gx, int B, int H, int W, int C, int ldg) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int CV = C / VEC;
    const int64_t total = (int64_t)B * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;
        const int xx = (int)(r % W);
        r /= W;
        const int yy = (int)(r % H);
        const int b = (int)(r / H);
        float a[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] = 0.f;
        if (gpass) {
            const uint4 raw = *(const uint4 *)(gpass + (((size_t)b * H + yy) * W + xx) * ldg + cv * VEC);
            const T *pv = (const T *)&raw;
#pragma unroll
            for (int e = 0; e < VEC; ++e) a[e] = to_f32(pv[e]);
        }
        auto level = [&](const T *g, int k, float inv) {
            const int Hk = H / k, Wk = W / k, oy = yy / k, ox = xx / k;
            if (g && oy < Hk && ox < Wk) {
                const uint4 raw = *(const uint4 *)(g + (((size_t)b * Hk + oy) * Wk + ox) * C + cv * VEC);
                const T *pv = (const T *)&raw;
#pragma unroll
                for (int e = 0; e < VEC; ++e) a[e] += to_f32(pv[e]) * inv;
            }
        };
        level(g2, 2, 0.25f);
        level(g4, 4, 1.f / 16.f);
        level(g8, 8, 1.f / 64.f);
        level(g16, 16, 1.f / 256.f);
        alignas(16) T o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(a[e]);
        *(uint4 *)(gx + i * VEC) = *(const uint4 *)o;
    }
}

// dst (B,H,W,C) = [residual +] src (B,Ho,Wo,C) placed on the pixels (s*i, s*j), zeros elsewhere: the data gradient of a 1x1 convolution
// with stride s is a plain GEMM over the OUTPUT pixels followed by this placement (3/4 of the input pixels of a stride-2 layer get
// no gradient at all; the transposed-gather implicit GEMM computed them as zero-page products).
template <typename T>
__global__ void stride_place_kernel(const T *__restrict__ src, const T *__restrict__ residual, T *__restrict__ dst, int B, int H, int W,
                                    int Ho, int Wo, int C, int s) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int CV = C / VEC;
    const int64_t total = (int64_t)B * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;
        const int x = (int)(r % W);
        r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        const int oy = y / s, ox = x / s;
        const bool live = (y % s == 0) && (x % s == 0) && oy < Ho && ox < Wo;
        if (live) v = *(const uint4 *)(src + (((size_t)b * Ho + oy) * Wo + ox) * C + cv * VEC);
        if (residual) {
            const uint4 rr = *(const uint4 *)(residual + i * VEC);
            if (live) {
                const T *a = (const T *)&v, *c = (const T *)&rr;
                alignas(16) T o[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(to_f32(a[e]) + to_f32(c[e]));
                v = *(const uint4 *)o;
            } else {
                v = rr;
            }
        }
        *(uint4 *)(dst + i * VEC) = v;
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
                                     int32_t C, int32_t mode, const void *gate, int32_t gate_act, int32_t dtype, void *stream) {
    if (!gy || !gx || B <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || mode < 0 || mode > 1) return -1;
    if (gate && (gate_act != GWD_ACT_RELU && gate_act != GWD_ACT_ELU)) return -1;
    if (gate && !(mode == MODE_NEAREST && C % 4 == 0)) return -4;           // the vector kernels of the nearest mode only
    const int64_t total = (int64_t)B * Hs * Ws * C;
    if (mode == MODE_NEAREST && dtype == GWD_BF16 && C % 4 == 0) {
        hipStream_t s = (hipStream_t)stream;
        if (C % 8 == 0) resample_nearest_bwd_vec_kernel<__bf16, 16><<<flat_grid(total / 8), 256, 0, s>>>((const __bf16 *)gy, (__bf16 *)gx, B, Hs, Ws, Ho, Wo, C, (const __bf16 *)gate, gate_act);
        else resample_nearest_bwd_vec_kernel<__bf16, 8><<<flat_grid(total / 4), 256, 0, s>>>((const __bf16 *)gy, (__bf16 *)gx, B, Hs, Ws, Ho, Wo, C, (const __bf16 *)gate, gate_act);
        GWD_CHECK_LAUNCH();
        return 0;
    }
    if (mode == MODE_NEAREST && dtype == GWD_F32 && C % 4 == 0) {
        resample_nearest_bwd_vec_kernel<float, 16><<<flat_grid(total / 4), 256, 0, (hipStream_t)stream>>>((const float *)gy, (float *)gx, B, Hs, Ws, Ho, Wo, C, (const float *)gate, gate_act);
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

extern "C" int gwd_psp_pool_forward(const void *x, void *p16, void *p8, void *p4, void *p2, int32_t B, int32_t H, int32_t W, int32_t C,
                                    int32_t dtype, void *stream) {
    if (!x || !p16 || !p8 || !p4 || !p2 || B <= 0 || H < 16 || W < 16 || C <= 0) return -1;
    const int vec = dtype == GWD_BF16 ? 8 : (dtype == GWD_F32 ? 4 : 0);
    if (!vec) return -2;
    if (C % vec) return -4;
    const int bx = (W + 15) / 16, by = (H + 15) / 16;
    if ((int64_t)B * bx * by >= (1LL << 31)) return -7;
    const int grid = B * bx * by;
    if (dtype == GWD_BF16)
        psp_pool_fwd_kernel<__bf16><<<grid, 256, 0, (hipStream_t)stream>>>((const __bf16 *)x, (__bf16 *)p16, (__bf16 *)p8, (__bf16 *)p4, (__bf16 *)p2, B, H, W, C, bx, by);
    else
        psp_pool_fwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float *)x, (float *)p16, (float *)p8, (float *)p4, (float *)p2, B, H, W, C, bx, by);
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_psp_pool_backward(const void *gpass, const void *g16, const void *g8, const void *g4, const void *g2, void *gx,
                                     int32_t B, int32_t H, int32_t W, int32_t C, int32_t ldg, int32_t dtype, void *stream) {
    if (!gx || B <= 0 || H < 16 || W < 16 || C <= 0) return -1;
    const int vec = dtype == GWD_BF16 ? 8 : (dtype == GWD_F32 ? 4 : 0);
    if (!vec) return -2;
    if (ldg == 0) ldg = C;
    if (C % vec || ldg % vec || ldg < C) return -4;
    const int64_t total = (int64_t)B * H * W * (C / vec);
    if (dtype == GWD_BF16)
        psp_pool_bwd_kernel<__bf16><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const __bf16 *)gpass, (const __bf16 *)g16, (const __bf16 *)g8, (const __bf16 *)g4, (const __bf16 *)g2, (__bf16 *)gx, B, H, W, C, ldg);
    else
        psp_pool_bwd_kernel<float><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const float *)gpass, (const float *)g16, (const float *)g8, (const float *)g4, (const float *)g2, (float *)gx, B, H, W, C, ldg);
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_stride_place(const void *src, const void *residual, void *dst, int32_t B, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                int32_t C, int32_t stride, int32_t dtype, void *stream) {
    if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || stride <= 0) return -1;
    const int vec = dtype == GWD_BF16 ? 8 : (dtype == GWD_F32 ? 4 : 0);
    if (!vec) return -2;
    if (C % vec) return -4;
    const int64_t total = (int64_t)B * H * W * (C / vec);
    if (dtype == GWD_BF16)
        stride_place_kernel<__bf16><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const __bf16 *)src, (const __bf16 *)residual, (__bf16 *)dst, B, H, W, Ho, Wo, C, stride);
    else
        stride_place_kernel<float><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const float *)src, (const float *)residual, (float *)dst, B, H, W, Ho, Wo, C, stride);
    GWD_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 convolution over a 2x nearest-upsampled map == 4x4 / stride 2 / pad 1 taps on the low-resolution side (ops._upsampled_dgrad_weight:
// the 2 x 2 children of a low-res pixel commute into the taps).  Two tiny layout kernels, one launch each instead of ~8 element-wise ones:
//   collapse: w (Cout,3,3,Cin) fp32 -> wk (Cin,4,4,Cout), wk[ci][t][s][co] = sum_{kh in G(t), kw in G(s)} w[co][kh][kw][ci], G(t) = [max(0, 2-t), min(2, 3-t)]
//   fold:     D (Cin,4,4,Cout) fp32 -> dw (Cout,3,3,Cin) += sum_{t in T(kh), s in T(kw)} D[ci][t][s][co],                   T(k) = [2-k, 3-k]
namespace {

template <typename T>
__global__ void ups_taps_collapse_kernel(const float *__restrict__ w, T *__restrict__ wk, int Cout, int Cin) {
    const int total = Cin * 16 * Cout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % Cout, ts = (i / Cout) & 15, ci = i / (16 * Cout);
        const int t = ts >> 2, s = ts & 3;
        float acc = 0.f;
        for (int kh = max(0, 2 - t); kh <= min(2, 3 - t); ++kh)
            for (int kw = max(0, 2 - s); kw <= min(2, 3 - s); ++kw) acc += w[((size_t)(co * 3 + kh) * 3 + kw) * Cin + ci];
        wk[i] = from_f32<T>(acc);
    }
}

__global__ void ups_taps_fold_kernel(const float *__restrict__ D, float *__restrict__ dw, int Cout, int Cin) {
    const int total = Cout * 9 * Cin;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ci = i % Cin, k = (i / Cin) % 9, co = i / (9 * Cin);
        const int kh = k / 3, kw = k - 3 * kh;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) acc += D[((size_t)(ci * 4 + 2 - kh + t) * 4 + 2 - kw + s) * Cout + co];
        dw[i] += acc;
    }
}

}  // namespace

extern "C" int gwd_upsample_taps_collapse(const float *w, void *wk, int32_t Cout, int32_t Cin, int32_t dtype, void *stream) {
    if (!w || !wk || Cout <= 0 || Cin <= 0 || (int64_t)Cout * Cin * 16 > INT32_MAX) return -1;
    const int64_t total = (int64_t)Cin * 16 * Cout;
    if (dtype == GWD_BF16)
        ups_taps_collapse_kernel<__bf16><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>(w, (__bf16 *)wk, Cout, Cin);
    else if (dtype == GWD_F32)
        ups_taps_collapse_kernel<float><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>(w, (float *)wk, Cout, Cin);
    else
        return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_upsample_taps_fold(const float *D, float *dw, int32_t Cout, int32_t Cin, void *stream) {
    if (!D || !dw || Cout <= 0 || Cin <= 0 || (int64_t)Cout * Cin * 16 > INT32_MAX) return -1;
    ups_taps_fold_kernel<<<flat_grid((int64_t)Cout * 9 * Cin), 256, 0, (hipStream_t)stream>>>(D, dw, Cout, Cin);
    GWD_CHECK_LAUNCH();
    return 0;
}
