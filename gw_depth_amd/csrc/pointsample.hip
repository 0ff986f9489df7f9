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

// Nearest sampling of the map as torch.roll(F.pad(map, to (Hf, Wf)), (-shift, -shift)) would present it (the shifted-window frame of the 1/32
// stage, multiscale_transformerr.py:662-691) without building that tensor: the coordinates address the (Hf, Wf) frame, frame pixel (y, x) is map
// pixel ((y + shift) mod Hf, (x + shift) mod Wf), zero where that lies in the padding.  Hf = 0: no frame.
struct Frame {
    int Hf, Wf, shift;
};
// frame pixel -> map pixel; false: outside the frame or in its padding
__device__ __forceinline__ bool frame_pixel(const Frame &f, int H, int W, int &xn, int &yn) {
    if (f.Hf > 0) {
        if (xn < 0 || xn >= f.Wf || yn < 0 || yn >= f.Hf) return false;
        xn += f.shift;
        yn += f.shift;
        xn -= xn >= f.Wf ? f.Wf : 0;
        yn -= yn >= f.Hf ? f.Hf : 0;
    }
    return xn >= 0 && xn < W && yn >= 0 && yn < H;
}

template <typename T>
__global__ void point_sample_fwd_kernel(const T *__restrict__ map, const float *__restrict__ coords, float *__restrict__ out,
                                        int B, int H, int W, int C, int S, int mode, Frame fr) {
    const int64_t total = (int64_t)B * S * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t bs = i / C;
        const int b = (int)(bs / S);
        const float ix = unnormalize(coords[bs * 2], fr.Hf > 0 ? fr.Wf : W), iy = unnormalize(coords[bs * 2 + 1], fr.Hf > 0 ? fr.Hf : H);
        const T *img = map + (int64_t)b * H * W * C + c;
        float r = 0.f;
        if (mode == 1) {
            int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
            if (frame_pixel(fr, H, W, xn, yn)) r = to_f32(img[((int64_t)yn * W + xn) * C]);
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

// The same gradient as a GATHER: a thread owns one pixel x 16 bytes of channels of image b and walks the S points (their taps and
// weights staged once per workgroup in LDS), adding the points that touch its pixel; every element of gmap is written exactly once -
// no pre-zeroing pass, no read-modify-write chain of S x 4 dependent updates per thread (31 us per call on the 21 x 21 x 512 maps of
// the 1/32 stage, where the scatter form has only B x C = 4 096 threads).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void point_sample_bwd_gather_kernel(const float *__restrict__ gout, const float *__restrict__ coords,
                                                                      T *__restrict__ gmap, int H, int W, int C, int S, int mode, Frame fr) {
    constexpr int SMAX = 256;                    // VEC = 16 bytes of channels per thread, or 1 channel (any C: the 1-channel depth maps)
    __shared__ int px[SMAX], py[SMAX];
    __shared__ float wgt[SMAX][4];
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int s = tid; s < S; s += 256) {
        const int64_t bs = (int64_t)b * S + s;
        const float ix = unnormalize(coords[bs * 2], fr.Hf > 0 ? fr.Wf : W), iy = unnormalize(coords[bs * 2 + 1], fr.Hf > 0 ? fr.Hf : H);
        if (mode == 1) {
            int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
            if (fr.Hf > 0 && !frame_pixel(fr, H, W, xn, yn)) xn = yn = -(1 << 20);       // meets no pixel
            px[s] = xn;
            py[s] = yn;
            wgt[s][0] = 1.f;
            wgt[s][1] = wgt[s][2] = wgt[s][3] = 0.f;
        } else {
            const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
            const float wx1 = ix - (float)x0, wx0 = (float)(x0 + 1) - ix, wy1 = iy - (float)y0, wy0 = (float)(y0 + 1) - iy;
            px[s] = x0;
            py[s] = y0;
            wgt[s][0] = wx0 * wy0;
            wgt[s][1] = wx1 * wy0;
            wgt[s][2] = wx0 * wy1;
            wgt[s][3] = wx1 * wy1;
        }
    }
    __syncthreads();
    const int CV = C / VEC;
    const int64_t i = (int64_t)blockIdx.x * 256 + tid;
    if (i >= (int64_t)H * W * CV) return;
    const int cv = (int)(i % CV);
    const int pix = (int)(i / CV), y = pix / W, x = pix - y * W;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    const int span = mode == 1 ? 0 : 1;
    for (int s = 0; s < S; ++s) {
        const int dx = x - px[s], dy = y - py[s];
        if ((unsigned)dx > (unsigned)span || (unsigned)dy > (unsigned)span) continue;
        const float w = wgt[s][dy * 2 + dx];
        const float *g = gout + ((int64_t)b * S + s) * C + cv * VEC;
        if constexpr (VEC == 1) {
            acc[0] += g[0] * w;
        } else {
#pragma unroll
            for (int e = 0; e < VEC; e += 4) {
                const float4 v = *(const float4 *)(g + e);
                acc[e] += v.x * w; acc[e + 1] += v.y * w; acc[e + 2] += v.z * w; acc[e + 3] += v.w * w;
            }
        }
    }
    T *dst = gmap + ((int64_t)b * H * W + pix) * C + cv * VEC;
    if constexpr (VEC == 1) {
        dst[0] = from_f32<T>(acc[0]);
    } else {
        alignas(16) T o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(acc[e]);
        *(uint4 *)dst = *(const uint4 *)o;
    }
}

}  // namespace

static int point_sample_backward_gather(const float *gout, const float *coords, void *gmap, int32_t B, int32_t H, int32_t W,
                                        int32_t C, int32_t S, int32_t mode, int32_t dtype, void *stream, Frame fr) {
    if (!gout || !coords || !gmap || B <= 0 || H <= 0 || W <= 0 || C <= 0 || S <= 0 || (mode != 0 && mode != 1)) return -1;
    int vec = dtype == GWD_BF16 ? 8 : (dtype == GWD_F32 ? 4 : 0);
    if (!vec) return -2;
    if (S > 256 || B > 65535) return -4;                       // caller: zero gmap, gwd_point_sample_backward
    if (C % vec || ((uintptr_t)gmap % 16)) vec = 1;
    const int64_t per = (int64_t)H * W * (C / vec);
    if (per >= (1LL << 31)) return -7;
    dim3 grid((unsigned)((per + 255) / 256), (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GWD_BF16) {
        if (vec == 1) point_sample_bwd_gather_kernel<__bf16, 1><<<grid, 256, 0, st>>>(gout, coords, (__bf16 *)gmap, H, W, C, S, mode, fr);
        else point_sample_bwd_gather_kernel<__bf16, 8><<<grid, 256, 0, st>>>(gout, coords, (__bf16 *)gmap, H, W, C, S, mode, fr);
    } else {
        if (vec == 1) point_sample_bwd_gather_kernel<float, 1><<<grid, 256, 0, st>>>(gout, coords, (float *)gmap, H, W, C, S, mode, fr);
        else point_sample_bwd_gather_kernel<float, 4><<<grid, 256, 0, st>>>(gout, coords, (float *)gmap, H, W, C, S, mode, fr);
    }
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_point_sample_backward_gather(const float *gout, const float *coords, void *gmap, int32_t B, int32_t H, int32_t W,
                                                int32_t C, int32_t S, int32_t mode, int32_t dtype, void *stream) {
    return point_sample_backward_gather(gout, coords, gmap, B, H, W, C, S, mode, dtype, stream, Frame{0, 0, 0});
}

static int point_sample_forward(const void *map, const float *coords, float *out, int32_t B, int32_t H, int32_t W,
                                int32_t C, int32_t S, int32_t mode, int32_t dtype, void *stream, Frame fr) {
    if (!map || !coords || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || S <= 0 || (mode != 0 && mode != 1)) return -1;
    const int64_t total = (int64_t)B * S * C;
    int64_t nb = (total + 255) / 256;
    const int grid = (int)(nb > 4096 ? 4096 : nb);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GWD_BF16) point_sample_fwd_kernel<__bf16><<<grid, 256, 0, st>>>((const __bf16 *)map, coords, out, B, H, W, C, S, mode, fr);
    else if (dtype == GWD_F32) point_sample_fwd_kernel<float><<<grid, 256, 0, st>>>((const float *)map, coords, out, B, H, W, C, S, mode, fr);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_point_sample_forward(const void *map, const float *coords, float *out, int32_t B, int32_t H, int32_t W,
                                        int32_t C, int32_t S, int32_t mode, int32_t dtype, void *stream) {
    return point_sample_forward(map, coords, out, B, H, W, C, S, mode, dtype, stream, Frame{0, 0, 0});
}

// Nearest sampling in the padded / rolled frame (see Frame): Hf >= H, Wf >= W, 0 <= shift < min(Hf, Wf).  The backward is the gather form
// (every element of gmap written; -4 when S > 256: build the frame with pad + roll and use the plain entry points).
extern "C" int gwd_point_sample_framed_forward(const void *map, const float *coords, float *out, int32_t B, int32_t H, int32_t W, int32_t C,
                                               int32_t S, int32_t Hf, int32_t Wf, int32_t shift, int32_t dtype, void *stream) {
    if (Hf < H || Wf < W || shift < 0 || shift >= Hf || shift >= Wf) return -1;
    return point_sample_forward(map, coords, out, B, H, W, C, S, 1, dtype, stream, Frame{Hf, Wf, shift});
}
extern "C" int gwd_point_sample_framed_backward(const float *gout, const float *coords, void *gmap, int32_t B, int32_t H, int32_t W, int32_t C,
                                                int32_t S, int32_t Hf, int32_t Wf, int32_t shift, int32_t dtype, void *stream) {
    if (Hf < H || Wf < W || shift < 0 || shift >= Hf || shift >= Wf) return -1;
    return point_sample_backward_gather(gout, coords, gmap, B, H, W, C, S, 1, dtype, stream, Frame{Hf, Wf, shift});
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
