// ResNet stem in one kernel: conv 7x7 / stride 2 / pad 3 (3 -> 64) + folded FrozenBatchNorm + ReLU + max-pool 3x3 / stride 2 / pad 1
// (torchvision resnet50 conv1 / bn1 / relu / maxpool as the reference's backbone runs them, /root/reference/src/models/backbone.py:65-92;
// the stem is frozen there - backbone.py:62-64 - so this is a forward-only kernel).
//
// As separate launches the 3-channel convolution sat on the scalar-gather implicit GEMM (6-byte pixels: no vector loads, no LDS-DMA;
// 221 us for 17 GFLOP), wrote 79 MB of conv output and the pool read it back (49 us).  Here a workgroup owns 8 x 15 POOLED pixels:
// their 17 x 31 conv outputs need a 39 x 69 input patch, staged once in LDS as [row][col][4] bf16 (channel 3 = 0), so that TWO taps
// (kw, kw+1) x 4 channels of a pixel are one aligned 16-byte ds_read = 8 consecutive k of the reduction k = (kh, kw padded to 8, c
// padded to 4) = 224 = 14 MFMA steps.  Y^T = W . X^T on v_mfma_f32_32x32x16_bf16: A = the weights (rows = output channels, held in
// REGISTERS for the whole launch: 14 steps x 2 channel tiles), B = 32 consecutive conv columns of one conv row; the accumulator has
// the conv COLUMN on the lane, so the pool's three rows are three accumulators of the same lane (max in registers) and its three
// columns are the neighbouring lanes (two wave shifts).  Post-ReLU values are >= 0, so out-of-range conv positions enter the max as
// 0 - the same result as the pool's -inf padding.  Only the pooled map (1/4 of the conv output) is written.
#include "common.h"

namespace stem {

constexpr int PW = 70;                 // patch columns (2 * 31 + 8), even -> every 2-pixel read is 16-byte aligned
constexpr int PR = 39;                 // patch rows: 2 * 16 + 7
constexpr int PXS = 15, PYS = 8;       // pooled columns / rows per workgroup
constexpr int NT = 256;

__device__ __forceinline__ f32x16 mma(const bf16x8 &a, const bf16x8 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

__global__ __launch_bounds__(NT, 2) void stem_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w, const float *__restrict__ shift,
                                                  __bf16 *__restrict__ y, int B, int H, int W, int Hc, int Wc, int Hp, int Wp, int tiles_y,
                                                  int tiles_x, int ntiles) {
    __shared__ __attribute__((aligned(16))) __bf16 patch[PR * PW * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < PR * PW * 4 / 8; i += NT) ((uint4 *)patch)[i] = uint4{0u, 0u, 0u, 0u};      // channel 3 and column 69 stay zero
    // weights in MFMA A-operand order (gwd_stem_pack): lane (r = channel in the tile, h) holds k = 16 s + 8 h + j
    bf16x8 wa[14][2];
#pragma unroll
    for (int s = 0; s < 14; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) wa[s][mt] = *(const bf16x8 *)(w + (size_t)((s * 2 + mt) * 64 + lane) * 8);
    float sh[2][16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) sh[mt][i] = shift ? shift[32 * mt + acc_row(i, h)] : 0.f;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int per = tiles_y * tiles_x;
        const int b = tile / per, rem = tile - b * per;
        const int py0 = (rem / tiles_x) * PYS, px0 = (rem % tiles_x) * PXS;
        const int R0 = 2 * py0 - 1, C0 = 2 * px0 - 1;          // conv row / column of local (0, 0)
        const int iy0 = 2 * R0 - 3, ix0 = 2 * C0 - 3;          // input pixel of patch (0, 0)
        __syncthreads();                                        // previous tile's reads are done (first time: the zero fill)
        for (int idx = tid; idx < PR * 69 * 3; idx += NT) {
            const int prow = idx / 207, e = idx - prow * 207;
            const int pcol = e / 3, c = e - pcol * 3;
            const int iy = iy0 + prow, ix = ix0 + pcol;
            __bf16 v = (__bf16)0.f;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[(((size_t)b * H + iy) * W + ix) * 3 + c];
            patch[(prow * PW + pcol) * 4 + c] = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int pr = 0; pr < 2; ++pr) {
            const int pyl = 2 * wave + pr, py = py0 + pyl;
            if (py >= Hp) break;                                // wave-uniform
            f32x16 vmax[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) vmax[mt][i] = 0.f;
#pragma unroll 1
            for (int dr = 0; dr < 3; ++dr) {
                const int lr = 2 * pyl + dr, cr = R0 + lr;
                if ((unsigned)cr >= (unsigned)Hc) continue;     // pool padding row (wave-uniform)
                f32x16 acc[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
#pragma unroll
                for (int s = 0; s < 14; ++s) {
                    const int kh = s >> 1, kw0 = 4 * (s & 1) + 2 * h;
                    const bf16x8 bf = *(const bf16x8 *)(patch + ((2 * lr + kh) * PW + 2 * r + kw0) * 4);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[mt] = mma(wa[s][mt], bf, acc[mt]);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) vmax[mt][i] = fmaxf(vmax[mt][i], acc[mt][i] + sh[mt][i]);       // relu = the max with 0
            }
            const bool colok = (unsigned)(C0 + r) < (unsigned)Wc;
            const int px = px0 + ((r - 1) >> 1);
            const bool store = (r & 1) && r <= 29 && px < Wp;
            __bf16 *dst = y + (((size_t)b * Hp + py) * Wp + px) * 64;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    alignas(8) __bf16 o[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v = colok ? vmax[mt][4 * g + q] : 0.f;
                        const float l = __shfl_up(v, 1, 64), rr = __shfl_down(v, 1, 64);
                        o[q] = (__bf16)fmaxf(v, fmaxf(l, rr));
                    }
                    if (store) *(uint2 *)(dst + 32 * mt + 8 * g + 4 * h) = *(const uint2 *)o;
                }
        }
    }
}

// w fp32 [64][7][7][3] (* scale[64]) -> bf16 in the order the lanes read it: [step s][channel tile mt][lane][j],
// k = 16 s + 8 h + j  ->  kh = s >> 1, kw = 4 (s & 1) + 2 h + (j >> 2), c = j & 3; kw = 7 and c = 3 are zero padding
__global__ void stem_pack_kernel(const float *__restrict__ w, const float *__restrict__ scale, __bf16 *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 14 * 2 * 64 * 8) return;
    const int j = i & 7, lane = (i >> 3) & 63, mt = (i >> 9) & 1, s = i >> 10;
    const int r = lane & 31, h = lane >> 5, co = 32 * mt + r;
    const int kh = s >> 1, kw = 4 * (s & 1) + 2 * h + (j >> 2), c = j & 3;
    float v = 0.f;
    if (kw < 7 && c < 3) v = w[((co * 7 + kh) * 7 + kw) * 3 + c] * (scale ? scale[co] : 1.f);
    out[i] = (__bf16)v;
}

}  // namespace stem

extern "C" int gwd_stem_pack(const float *w, const float *scale, void *packed, void *stream) {
    if (!w || !packed) return -1;
    stem::stem_pack_kernel<<<(GWD_STEM_PACKED_ELEMS + 255) / 256, 256, 0, (hipStream_t)stream>>>(w, scale, (__bf16 *)packed);
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_stem_forward(const void *x, const void *w, const float *shift, void *y, int32_t B, int32_t H, int32_t W, int32_t dtype,
                                void *stream) {
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0) return -1;
    if (dtype != GWD_BF16) return -2;
    if ((int64_t)B * H * W * 3 >= (1LL << 31)) return -7;
    const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1;      // (H + 2*3 - 7) / 2 + 1
    const int Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;    // (Hc + 2*1 - 3) / 2 + 1
    const int tiles_y = (Hp + stem::PYS - 1) / stem::PYS, tiles_x = (Wp + stem::PXS - 1) / stem::PXS;
    const int64_t ntiles = (int64_t)B * tiles_y * tiles_x;
    if (ntiles >= (1LL << 31)) return -7;
    const int rounds = (int)((ntiles + 511) / 512);            // persistent workgroups (the weights live in their registers), 2 per CU
    const int grid = (int)((ntiles + rounds - 1) / rounds);
    stem::stem_kernel<<<grid, stem::NT, 0, (hipStream_t)stream>>>((const __bf16 *)x, (const __bf16 *)w, shift, (__bf16 *)y, B, H, W, Hc, Wc, Hp,
                                                                   Wp, tiles_y, tiles_x, (int)ntiles);
    GWD_CHECK_LAUNCH();
    return 0;
}
