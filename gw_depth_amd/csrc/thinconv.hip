// 3x3 / stride 1 / pad 1 convolutions with ONE OR TWO output channels over a 32-channel bf16 map - the full-resolution
// depth and segmentation heads (src/models/dense_upsample.py:175-181: get_depth 32->1, get_seg 32->2 at 480x640).
// On the implicit-GEMM kernels these cost 0.2-0.6 ms each (a 32-column MFMA tile for 1-2 real columns, and a weight
// gradient whose 2.4 M-pixel reduction ends in atomics); they are pure streaming problems (157 MB of activations, 576
// weights), so here they are: 16x16 pixel tiles with a halo in LDS, VALU math, everything else in registers.
//   forward : thread = output pixel, 288 MACs per output channel, fused scale/shift/activation
//   dgrad   : thread = input pixel, 32 output channels from the 9 x NO neighbouring gradients
//   wgrad   : persistent workgroups (one round), thread = (tap, channel), sums over the pixels of its tiles in
//             registers, ONE set of 288 x NO atomics per workgroup at the end
#include "common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int C = 32, TILE = 16, HALO = TILE + 2, PIX = 80;     // PIX: LDS bytes per pixel (64 + 16: conflict-free 16-byte reads)

// x tile with halo -> LDS (zero outside the image); tile origin (h0, w0) of image b
__device__ __forceinline__ void load_x_tile(char *xs, const __bf16 *x, int b, int h0, int w0, int H, int W, int nthreads) {
    for (int i = threadIdx.x; i < HALO * HALO * 4; i += nthreads) {
        const int v = i & 3, p = i >> 2, py = p / HALO, px = p - py * HALO;
        const int ih = h0 + py - 1, iw = w0 + px - 1;
        uint4 val = make_uint4(0u, 0u, 0u, 0u);
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
            val = *(const uint4 *)(x + (((size_t)b * H + ih) * W + iw) * C + v * 8);
        *(uint4 *)(xs + p * PIX + v * 16) = val;
    }
}

template <int NO>
__global__ __launch_bounds__(256) void thin_fwd_kernel(const gwd_conv_desc d) {
    __shared__ __attribute__((aligned(16))) char xs[HALO * HALO * PIX];
    const int H = d.Hi, W = d.Wi;
    const int tw = (W + TILE - 1) / TILE, th = (H + TILE - 1) / TILE;
    int t = blockIdx.x;
    const int w0 = (t % tw) * TILE;
    t /= tw;
    const int h0 = (t % th) * TILE, b = t / th;
    // the 9 x 32 x NO weights are wave-uniform: read through the scalar cache as packed bf16 pairs (one dword = channels 2j, 2j + 1)
    // and used as SGPR operands - as fp32 copies in LDS they were 2/3 (NO = 1) to 4/5 (NO = 2) of the kernel's LDS traffic
    const uint32_t *__restrict__ wq = (const uint32_t *)d.w;
    load_x_tile(xs, (const __bf16 *)d.x, b, h0, w0, H, W, 256);
    __syncthreads();
    const int ty = threadIdx.x / TILE, tx = threadIdx.x % TILE;
    float acc[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) acc[n] = 0.f;
    // the tap loop stays rolled: fully unrolled, the compiler hoists all 36 LDS vectors and spills
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap - kh * 3;
        const char *px = xs + ((ty + kh) * HALO + tx + kw) * PIX;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const bf16x8 xv = *(const bf16x8 *)(px + v * 16);
            const uint32_t *xw = (const uint32_t *)&xv;
#pragma unroll
            for (int n = 0; n < NO; ++n) {
                const uint32_t *wp = wq + ((n * 9 + tap) * C + v * 8) / 2;
#pragma unroll
                for (int j = 0; j < 4; ++j)         // v_dot2c_f32_bf16: two multiply-adds per instruction straight from the packed pairs (was: two conversions + two FMAs)
                    acc[n] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, xw[j]), __builtin_bit_cast(bf16x2, wp[j]), acc[n], false);
            }
        }
    }
    const int oh = h0 + ty, ow = w0 + tx;
    if (oh < H && ow < W) {
        const size_t o = (((size_t)b * H + oh) * W + ow) * NO;
        __bf16 *y = (__bf16 *)d.y;
        __bf16 *z = (__bf16 *)d.z;
#pragma unroll
        for (int n = 0; n < NO; ++n) {
            float v = acc[n] * (d.scale ? d.scale[n] : 1.0f) + (d.shift ? d.shift[n] : 0.0f);
            if (z) z[o + n] = (__bf16)v;
            y[o + n] = (__bf16)(apply_act(v, d.act) * d.act_scale);
        }
    }
}

// data gradient: desc is the transposed-gather GEMM (x = dy (B,H,W,NO), w = [32][3][3][NO], y = gx (B,H,W,32))
template <int NO>
__global__ __launch_bounds__(256) void thin_dgrad_kernel(const gwd_conv_desc d) {
    __shared__ float gs[HALO * HALO * NO];
    __shared__ __attribute__((aligned(16))) float ws[9 * NO * C];           // [tap][n][c]
    const int H = d.Hi, W = d.Wi;
    const int tw = (W + TILE - 1) / TILE, th = (H + TILE - 1) / TILE;
    int t = blockIdx.x;
    const int w0 = (t % tw) * TILE;
    t /= tw;
    const int h0 = (t % th) * TILE, b = t / th;
    const __bf16 *wgt = (const __bf16 *)d.w;
    for (int i = threadIdx.x; i < 9 * NO * C; i += 256) {
        const int c = i % C, n = (i / C) % NO, tap = i / (C * NO);
        ws[i] = (float)wgt[(c * 9 + tap) * NO + n];
    }
    const __bf16 *gy = (const __bf16 *)d.x;
    for (int i = threadIdx.x; i < HALO * HALO; i += 256) {
        const int py = i / HALO, px = i - py * HALO;
        const int ih = h0 + py - 1, iw = w0 + px - 1;
        const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
#pragma unroll
        for (int n = 0; n < NO; ++n) gs[i * NO + n] = ok ? (float)gy[(((size_t)b * H + ih) * W + iw) * NO + n] : 0.f;
    }
    __syncthreads();
    const int ty = threadIdx.x / TILE, tx = threadIdx.x % TILE;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    // gx[p][c] = sum_{kh,kw,n} dy[p + (1 - kh, 1 - kw)][n] * w[n][kh][kw][c]
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int n = 0; n < NO; ++n) {
                const float g = gs[((ty + 2 - kh) * HALO + tx + 2 - kw) * NO + n];
                const float *wp = ws + ((kh * 3 + kw) * NO + n) * C;
#pragma unroll
                for (int c4 = 0; c4 < C; c4 += 4) {
                    const f32x4 wv = *(const f32x4 *)(wp + c4);
                    acc[c4] += g * wv[0];
                    acc[c4 + 1] += g * wv[1];
                    acc[c4 + 2] += g * wv[2];
                    acc[c4 + 3] += g * wv[3];
                }
            }
    const int oh = h0 + ty, ow = w0 + tx;
    if (oh < H && ow < W) {
        __bf16 *gx = (__bf16 *)d.y + (((size_t)b * H + oh) * W + ow) * C;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)acc[v * 8 + e];
            *(bf16x8 *)(gx + v * 8) = o;
        }
    }
}

template <int NO>
__global__ __launch_bounds__(320) void thin_wgrad_kernel(const gwd_conv_desc d, float *__restrict__ dw, int n_tiles) {
    __shared__ __attribute__((aligned(16))) char xs[HALO * HALO * PIX];
    __shared__ float gs[TILE * TILE * NO];
    const int H = d.Hi, W = d.Wi;
    const int tw = (W + TILE - 1) / TILE, th = (H + TILE - 1) / TILE;
    const int tap = threadIdx.x / C, c = threadIdx.x % C;              // threads 288..319 only help with the loads
    const int kh = tap / 3, kw = tap - kh * 3;
    const bool worker = threadIdx.x < 9 * C;
    float acc[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) acc[n] = 0.f;
    const __bf16 *gy = (const __bf16 *)d.y;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int t = tile;
        const int w0 = (t % tw) * TILE;
        t /= tw;
        const int h0 = (t % th) * TILE, b = t / th;
        __syncthreads();
        load_x_tile(xs, (const __bf16 *)d.x, b, h0, w0, H, W, 320);
        for (int i = threadIdx.x; i < TILE * TILE; i += 320) {
            const int oh = h0 + i / TILE, ow = w0 + i % TILE;
            const bool ok = oh < H && ow < W;
#pragma unroll
            for (int n = 0; n < NO; ++n) gs[i * NO + n] = ok ? (float)gy[(((size_t)b * H + oh) * W + ow) * NO + n] : 0.f;
        }
        __syncthreads();
        if (worker) {
            // one tile row per trip, its 16 pixels unrolled: every LDS address of the row is the row's base plus a constant.  The flat pixel
            // loop spent ~6 vector instructions per multiply-add on index arithmetic (SQ_INSTS_VALU 64 M per launch against 11 M useful
            // ones): the kernel is bound by instruction issue, not by its 157 MB of activations - 167 / 148 -> 136 / 102 us.  (Channel pairs
            // per thread with 16-byte gradient reads: fewer instructions still, but 144 busy threads of 320: 178 / 133 us.)
            for (int py = 0; py < TILE; ++py) {
                const char *xrow = xs + ((py + kh) * HALO + kw) * PIX + c * 2;
                const float *grow = gs + py * TILE * NO;
#pragma unroll
                for (int px = 0; px < TILE; ++px) {
                    const float xv = (float)*(const __bf16 *)(xrow + px * PIX);
#pragma unroll
                    for (int n = 0; n < NO; ++n) acc[n] += grow[px * NO + n] * xv;
                }
            }
        }
    }
    if (worker) {
#pragma unroll
        for (int n = 0; n < NO; ++n) unsafeAtomicAdd(dw + (n * 9 + tap) * C + c, acc[n]);
    }
}

bool thin_common(const gwd_conv_desc *d) {
    return d->dtype == GWD_BF16 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Ho == d->Hi && d->Wo == d->Wi &&
           (int64_t)d->B * ((d->Hi + TILE - 1) / TILE) * ((d->Wi + TILE - 1) / TILE) < (1LL << 31);
}

}  // namespace

// 1 = launched, 0 = not a thin problem (caller continues with the implicit-GEMM kernels)
int gwd_thin_conv_forward(const gwd_conv_desc *d, hipStream_t s) {
    if (!thin_common(d) || d->mult) return 0;
    const unsigned tiles = (unsigned)(d->B * ((d->Hi + TILE - 1) / TILE) * ((d->Wi + TILE - 1) / TILE));
    // desc.gate: not here - any per-element gate in thin_dgrad_kernel's epilogue tips the compiler into a 256-VGPR schedule of the
    // whole kernel (occupancy 8 -> 1, 78 -> 176 us even for launches WITHOUT a gate; operand prefetch, raw-word extraction, launch bounds
    // and a scheduling barrier all tried); a gated thin problem runs on the implicit GEMM, and the model keeps conv2's own ELU pass
    if (d->gate) return 0;
    if (d->gather == GWD_GATHER_CONV && d->Cin == C && (d->Cout == 1 || d->Cout == 2) && !d->residual) {
        if (d->Cout == 1) thin_fwd_kernel<1><<<tiles, 256, 0, s>>>(*d);
        else thin_fwd_kernel<2><<<tiles, 256, 0, s>>>(*d);
        return 1;
    }
    if (d->gather == GWD_GATHER_TRANSPOSED && d->Cout == C && (d->Cin == 1 || d->Cin == 2) && !d->residual && !d->z && !d->scale &&
        !d->shift && d->act == GWD_ACT_NONE && d->act_scale == 1.0f) {
        if (d->Cin == 1) thin_dgrad_kernel<1><<<tiles, 256, 0, s>>>(*d);
        else thin_dgrad_kernel<2><<<tiles, 256, 0, s>>>(*d);
        return 1;
    }
    return 0;
}

int gwd_thin_conv_wgrad(const gwd_conv_desc *d, float *dw, hipStream_t s) {
    if (!thin_common(d) || d->gather != GWD_GATHER_CONV || d->Cin != C || (d->Cout != 1 && d->Cout != 2) || d->scale) return 0;   // scaled gradients: igemm
    const int tiles = d->B * ((d->Hi + TILE - 1) / TILE) * ((d->Wi + TILE - 1) / TILE);
    constexpr int cap = 1024;                                          // every workgroup flushes its 288 / 576 sums with atomics onto the same addresses
    const int grid = tiles < cap ? tiles : cap;                        // <= 4 workgroups (34 KiB LDS each) per CU, one round
    if (d->Cout == 1) thin_wgrad_kernel<1><<<grid, 320, 0, s>>>(*d, dw, tiles);
    else thin_wgrad_kernel<2><<<grid, 320, 0, s>>>(*d, dw, tiles);
    return 1;
}
