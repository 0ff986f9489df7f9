// Padding masks of the feature levels and the sine position embedding in two launches per level.
// Reference: the mask pyramid is F.interpolate(mask[None].float(), size).to(bool) (/root/reference/src/models/backbone.py:81-88,
// nearest), the embedding is PositionEmbeddingSine (/root/reference/src/models/position_encoding.py:28-48): cumulative counts of
// the un-masked pixels along y and x, optionally normalised to 2 pi, divided by temperature^(2 (c / 2) / F), sin on even and cos on
// odd channels, (y | x) concatenated.  As torch ops that is ~15 small launches per level (cumsum x2, div, pow, arange, sin, cos,
// stack, flatten, cat, casts) x 4 levels + 12 for the masks, every step, for ~1 MB of output.
#include "common.h"

namespace {

// one workgroup per image: thread t < w scans column t (y counts), thread w + i scans row i (x counts); also writes the level mask
__global__ __launch_bounds__(1024) void pos_counts_kernel(const uint8_t *__restrict__ full, uint8_t *__restrict__ lvl, int16_t *__restrict__ counts,
                                                         int H, int W, int h, int w) {
    __shared__ uint8_t sm[32768];                       // the level mask of one image when it fits (every level of the 480 x 640 step)
    const int b = blockIdx.x;
    const uint8_t *src = full + (size_t)b * H * W;
    uint8_t *m = lvl + (size_t)b * h * w;
    const bool in_lds = h * w <= 32768;
    int16_t *cnt = counts + (size_t)b * h * w * 2;
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;         // ATen nearest: src = min(floor(dst * in / out), in - 1)
    for (int p = threadIdx.x; p < h * w; p += blockDim.x) {
        const int i = p / w, j = p - i * w;
        int yi = (int)floorf((float)i * sy), xj = (int)floorf((float)j * sx);
        yi = yi < H - 1 ? yi : H - 1;
        xj = xj < W - 1 ? xj : W - 1;
        const uint8_t v = src[(size_t)yi * W + xj] ? 1 : 0;
        m[p] = v;
        if (in_lds) sm[p] = v;
    }
    __syncthreads();
    const uint8_t *rd = in_lds ? sm : m;
    for (int t = threadIdx.x; t < w + h; t += blockDim.x) {
        if (t < w) {
            int c = 0;
            for (int i = 0; i < h; ++i) {
                c += rd[i * w + t] ? 0 : 1;
                cnt[(i * w + t) * 2] = (int16_t)c;
            }
        } else {
            const int i = t - w;
            int c = 0;
            for (int j = 0; j < w; ++j) {
                c += rd[i * w + j] ? 0 : 1;
                cnt[(i * w + j) * 2 + 1] = (int16_t)c;
            }
        }
    }
}

// out[b][i][j][c]: c < F from the y count, c >= F from the x count
__global__ __launch_bounds__(256) void pos_emit_kernel(const int16_t *__restrict__ counts, const float *__restrict__ dim_t, float *__restrict__ out,
                                                       int64_t total, int h, int w, int F, int normalize) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        // 32-bit index arithmetic (the host checks total < 2^31): four 64-bit divisions per element were most of this kernel
        const unsigned u = (unsigned)idx, F2 = 2u * (unsigned)F;
        const int c = (int)(u % F2);
        const unsigned p = u / F2;                             // (b, i, j)
        const int j = (int)(p % (unsigned)w), i = (int)((p / (unsigned)w) % (unsigned)h);
        const int64_t b = p / ((unsigned)w * (unsigned)h);
        const int16_t *cb = counts + b * h * w * 2;
        const bool isx = c >= F;
        float v = (float)cb[((size_t)i * w + j) * 2 + (isx ? 1 : 0)];
        if (normalize) {
            const float last = (float)(isx ? cb[((size_t)i * w + (w - 1)) * 2 + 1] : cb[((size_t)(h - 1) * w + j) * 2]);
            v = __fmul_rn(__fdiv_rn(v, __fadd_rn(last, 1e-6f)), 6.283185307179586f);
        }
        const int k = isx ? c - F : c;
        v = __fdiv_rn(v, dim_t[k]);
        out[idx] = (k & 1) ? cosf(v) : sinf(v);
    }
}

}  // namespace

extern "C" int gwd_pos_sine(const uint8_t *mask_full, uint8_t *mask_level, int16_t *counts, const float *dim_t, float *out, int32_t B,
                            int32_t H, int32_t W, int32_t h, int32_t w, int32_t F, int32_t normalize, void *stream) {
    if (!counts || B <= 0 || h <= 0 || w <= 0) return -1;
    if (mask_full && (H <= 0 || W <= 0)) return -1;
    if (h > 32767 || w > 32767) return -7;
    hipStream_t s = (hipStream_t)stream;
    if (mask_full) {                                   // first call for a level: mask + counts
        if (!mask_level) return -1;
        pos_counts_kernel<<<B, 1024, 0, s>>>(mask_full, mask_level, counts, H, W, h, w);       // one workgroup per image: 16 waves hide the gather latency
        GWD_CHECK_LAUNCH();
    }
    if (out) {                                         // an embedding of F + F channels from the level's counts
        if (!dim_t || F <= 0) return -1;
        const int64_t total = (int64_t)B * h * w * 2 * F;
        if (total >= (1LL << 31)) return -7;
        int64_t nb = (total + 255) / 256;
        pos_emit_kernel<<<(int)(nb > 4096 ? 4096 : nb), 256, 0, s>>>(counts, dim_t, out, total, h, w, F, normalize);
        GWD_CHECK_LAUNCH();
    }
    return 0;
}
