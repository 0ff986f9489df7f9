// Geometric part of the input pipeline on decoded images (second slice): the resizes, flips and crops of
// /root/reference/src/datasets/transforms_depth.py:59-372 as table-driven integer kernels, bit-exact with what the reference gets from
// Pillow through torchvision:
//   * RGB images: Image.resize(BILINEAR) = two separable passes over uint8 data with Pillow's fixed-point coefficients (22 fractional
//     bits, support scaled by max(1, in / out): the built-in antialiasing), each pass rounding to uint8 - gwd_resample_u8_pass.  The
//     coefficient and bound tables come from the caller (a few hundred doubles per image side, computed on the host exactly as
//     Pillow's precompute_coeffs / normalize_coeffs_8bpc do); a flip or a crop before the resize is an index map of the SOURCE
//     (base + step * i), so flip -> crop -> resize is one read of the original image.
//   * depth (int32 mm) and label (uint8) maps: Image.resize(NEAREST), flips, crops = one gather through per-axis index tables
//     (Pillow accumulates the source coordinate in double precision; the caller reproduces that on the host) - gwd_gather2d.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

// dst[o][j][c] (axis 0) or dst[j][o][c] (axis 1): o = output index along the resampled axis, j = index along the other axis
__global__ void resample_u8_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const int32_t *__restrict__ bounds,
                                   const int32_t *__restrict__ kk, int ksize, int axis, int n_out, int other, int C, int64_t row_stride,
                                   int base0, int step0, int base1, int step1) {
    const int64_t total = (int64_t)n_out * other * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        int o, j;
        if (axis == 1) {                        // horizontal: dst [other rows][n_out][C]
            o = (int)(r % n_out);
            j = (int)(r / n_out);
        } else {                                // vertical: dst [n_out][other cols][C]
            j = (int)(r % other);
            o = (int)(r / other);
        }
        const int first = bounds[2 * o], cnt = bounds[2 * o + 1];
        const int32_t *k = kk + (size_t)o * ksize;
        int ss = 1 << (PRECISION_BITS - 1);
        const int64_t jo = (int64_t)base1 + (int64_t)step1 * j;
        for (int t = 0; t < cnt; ++t) {
            const int64_t a = (int64_t)base0 + (int64_t)step0 * (first + t);
            const int64_t off = axis == 1 ? (jo * row_stride + a * C + c) : (a * row_stride + jo * C + c);
            ss += (int)src[off] * k[t];
        }
        ss >>= PRECISION_BITS;
        dst[i] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
    }
}

template <int EB>
__global__ void gather2d_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const int32_t *__restrict__ ytab,
                                const int32_t *__restrict__ xtab, int oh, int ow, int64_t row_stride_bytes) {
    const int64_t total = (int64_t)oh * ow;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % ow), y = (int)(i / ow);
        const uint8_t *p = src + (int64_t)ytab[y] * row_stride_bytes + (int64_t)xtab[x] * EB;
        uint8_t *q = dst + i * EB;
#pragma unroll
        for (int b = 0; b < EB; ++b) q[b] = p[b];
    }
}

inline int flat_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int gwd_resample_u8_pass(const uint8_t *src, uint8_t *dst, const int32_t *bounds, const int32_t *kk, int32_t ksize, int32_t axis,
                                    int32_t n_out, int32_t other, int32_t C, int64_t src_row_stride, int32_t base0, int32_t step0,
                                    int32_t base1, int32_t step1, void *stream) {
    if (!src || !dst || !bounds || !kk || ksize <= 0 || n_out <= 0 || other <= 0 || C <= 0 || (axis != 0 && axis != 1)) return -1;
    if ((step0 != 1 && step0 != -1) || (step1 != 1 && step1 != -1)) return -1;
    const int64_t total = (int64_t)n_out * other * C;
    resample_u8_kernel<<<flat_grid(total), 256, 0, (hipStream_t)stream>>>(src, dst, bounds, kk, ksize, axis, n_out, other, C, src_row_stride, base0,
                                                                          step0, base1, step1);
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_gather2d(const void *src, void *dst, const int32_t *ytab, const int32_t *xtab, int32_t oh, int32_t ow,
                            int64_t src_row_stride_bytes, int32_t elem_bytes, void *stream) {
    if (!src || !dst || !ytab || !xtab || oh <= 0 || ow <= 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    const int g = flat_grid((int64_t)oh * ow);
    const uint8_t *sp = (const uint8_t *)src;
    uint8_t *dp = (uint8_t *)dst;
    switch (elem_bytes) {
        case 1: gather2d_kernel<1><<<g, 256, 0, s>>>(sp, dp, ytab, xtab, oh, ow, src_row_stride_bytes); break;
        case 2: gather2d_kernel<2><<<g, 256, 0, s>>>(sp, dp, ytab, xtab, oh, ow, src_row_stride_bytes); break;
        case 3: gather2d_kernel<3><<<g, 256, 0, s>>>(sp, dp, ytab, xtab, oh, ow, src_row_stride_bytes); break;
        case 4: gather2d_kernel<4><<<g, 256, 0, s>>>(sp, dp, ytab, xtab, oh, ow, src_row_stride_bytes); break;
        default: return -4;
    }
    GWD_CHECK_LAUNCH();
    return 0;
}
