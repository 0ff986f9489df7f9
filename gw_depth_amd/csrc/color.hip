// Photometric part of the input pipeline on the device (third slice): the four adjustments of the reference's
// ColorJitter (/root/reference/src/datasets/transforms_depth.py:551-600), bit-exact with what torchvision obtains from Pillow on a
// uint8 RGB image:
//   brightness / contrast / saturation = Image.blend(degenerate, image, factor) with degenerate = black / the rounded mean of the
//   luma image / the luma of the pixel (ImageEnhance.Brightness / Contrast / Color); luma = (19595 R + 38470 G + 7471 B + 2^15) >> 16;
//   blend in SINGLE precision, one rounding per operation (no fused multiply-add), truncated, clipped when the factor is outside [0,1];
//   hue = RGB -> HSV -> H + shift (uint8 wrap-around) -> RGB with Pillow's Convert.c arithmetic (float / double mix reproduced).
// The contrast mean is a device-side reduction (64-bit integer sum of the luma) read by the blend kernel: no host round trip.
#include "common.h"

// The reference arithmetic (Pillow built for x86-64) rounds after every operation: no fused multiply-add anywhere in this file.
// hipcc contracts a * b + c by default - even through __fmul_rn / __fadd_rn - which changed 1.7 % of the contrast outputs by one.
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

__device__ __forceinline__ uint8_t blend1(int d, int i, float a, bool inside) {
    const float prod = a * (float)(i - d);        // plain operators: the _rn intrinsics are header code compiled with contraction on
    const float t = (float)d + prod;
    if (inside) return (uint8_t)(int)t;
    if (t <= 0.0f) return 0;
    if (t >= 255.0f) return 255;
    return (uint8_t)(int)t;
}

__global__ void luma_sum_kernel(const uint8_t *__restrict__ img, unsigned long long *__restrict__ sum, int64_t npix) {
    unsigned long long s = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x)
        s += (unsigned)luma(img[p * 3], img[p * 3 + 1], img[p * 3 + 2]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(sum, s);
}

// mode 0 brightness, 1 contrast (mean from *sum / npix), 2 saturation
__global__ void blend_kernel(const uint8_t *__restrict__ img, uint8_t *__restrict__ out, const unsigned long long *__restrict__ sum, int64_t npix,
                             int mode, float factor) {
    const bool inside = factor >= 0.0f && factor <= 1.0f;
    int mean = 0;
    if (mode == 1) mean = (int)((double)(*sum) / (double)npix + 0.5);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        const int r = img[p * 3], g = img[p * 3 + 1], b = img[p * 3 + 2];
        const int d = mode == 0 ? 0 : (mode == 1 ? mean : luma(r, g, b));
        out[p * 3] = blend1(d, r, factor, inside);
        out[p * 3 + 1] = blend1(d, g, factor, inside);
        out[p * 3 + 2] = blend1(d, b, factor, inside);
    }
}

__device__ __forceinline__ int clip8(long v) { return v < 0 ? 0 : (v > 255 ? 255 : (int)v); }

__global__ void hue_kernel(const uint8_t *__restrict__ img, uint8_t *__restrict__ out, int64_t npix, int shift) {
    for (int64_t px = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; px < npix; px += (int64_t)gridDim.x * blockDim.x) {
        const int r = img[px * 3], g = img[px * 3 + 1], b = img[px * 3 + 2];
        const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
        int uh = 0, us = 0;
        const int uv = maxc;
        if (minc != maxc) {                                   // rgb2hsv_row
            const float cr = (float)(maxc - minc);
            const float s = cr / (float)maxc;
            const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
            float h;
            if (r == maxc) h = bc - gc;
            else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
            else h = (float)(4.0 + (double)gc - (double)rc);
            h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
            uh = clip8((long)((double)h * 255.0));
            us = clip8((long)((double)s * 255.0));
        }
        uh = (uh + shift) & 255;                              // np.uint8 addition wraps
        int ro, go, bo;
        if (us == 0) {
            ro = go = bo = uv;
        } else {                                              // hsv2rgb
            const double hf = (double)(float)uh * 6.0 / 255.0;
            const int i = (int)floor(hf);
            const float f = (float)(hf - (double)(float)i);
            const float fs = (float)((double)(float)us / 255.0);
            const double vf = (double)(float)uv;
            const int p = clip8((long)round(vf * (1.0 - (double)fs)));
            const int q = clip8((long)round(vf * (1.0 - (double)fs * (double)f)));
            const int t = clip8((long)round(vf * (1.0 - (double)fs * (1.0 - (double)f))));
            switch (i % 6) {
                case 0: ro = uv; go = t; bo = p; break;
                case 1: ro = q; go = uv; bo = p; break;
                case 2: ro = p; go = uv; bo = t; break;
                case 3: ro = p; go = q; bo = uv; break;
                case 4: ro = t; go = p; bo = uv; break;
                default: ro = uv; go = p; bo = q; break;
            }
        }
        out[px * 3] = (uint8_t)ro;
        out[px * 3 + 1] = (uint8_t)go;
        out[px * 3 + 2] = (uint8_t)bo;
    }
}

inline int flat_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int gwd_color_adjust(const uint8_t *rgb, uint8_t *out, uint64_t *scratch, int64_t npix, int32_t mode, float factor, void *stream) {
    if (!rgb || !out || npix <= 0 || mode < 0 || mode > 3) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (mode == 3) {                                          // hue: `factor` carries the uint8 shift itself (0..255), see gwdepth.h
        if (factor < 0.0f || factor > 255.0f) return -1;
        const int shift = (int)factor;
        hue_kernel<<<flat_grid(npix), 256, 0, s>>>(rgb, out, npix, shift);
    } else {
        if (mode == 1) {
            if (!scratch) return -1;
            hipError_t e = hipMemsetAsync(scratch, 0, sizeof(uint64_t), s);
            if (e != hipSuccess) return (int)e;
            luma_sum_kernel<<<flat_grid(npix), 256, 0, s>>>(rgb, (unsigned long long *)scratch, npix);
        }
        blend_kernel<<<flat_grid(npix), 256, 0, s>>>(rgb, out, (const unsigned long long *)scratch, npix, mode, factor);
    }
    GWD_CHECK_LAUNCH();
    return 0;
}
