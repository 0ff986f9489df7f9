// 3x3 stride-1 convolutions with 32 / 64 channels on large maps (the full- and half-resolution decoder of
// /root/reference/src/models/dense_upsample.py:160-182 and ResNet layer1's 3x3: 0.15 - 2.5 M output pixels, 288 - 576 MACs each).
// These layers are HBM-bound by arithmetic intensity, but the tap-by-tap implicit GEMM re-reads every input pixel nine times
// from L2 (1.4 GB of L2 -> LDS traffic for a 157 MB map): 165 - 340 us per launch against a 60 - 70 us HBM floor.
//
// Here an 8 x 32 output tile's (8+2) x (32+2) input HALO is staged in LDS once and all nine taps read it there; the weights
// (<= 83 KB) sit in LDS for the whole launch (persistent workgroups walk the tiles, the next tile's halo is prefetched into
// registers during the MFMAs).  Products run as Y^T = W . X^T on v_mfma_f32_32x32x16_bf16: A = one tap's weights (rows =
// output channels), B = 32 consecutive pixels of a halo row (lane = pixel, 8 channels per k-slot) - the accumulator has the
// PIXEL on the lane and 4 consecutive channels per register quad, so the epilogue (shift, activation) stores 8 bytes per lane.
// Gather modes: plain, data gradient (taps mirrored, transposed weights prepared by gwd_weight_prep), and the exact 2x nearest
// up-sample of the up-convolutions (source pixel = virtual pixel >> 1), fused into the halo load.
// The weight gradient uses the same halo: dW[co][tap][ci] += gy^T . x_shifted per tap, both operands gathered by column with
// ds_read_b64_tr_b16, accumulated in registers over all tiles a workgroup visits, one set of fp32 atomics at the end.
#include "common.h"
#include <stdlib.h>

namespace tconv {

constexpr int TW = 32, HW = TW + 2;            // output tile width, halo width; tile height TH = 8 or 16 rows (2 rows per wave)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ f32x16 mma(const bf16x8 &a, const bf16x8 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

template <int CIN, int TH, int RPW = 2> struct Halo {
    static constexpr int HH = TH + 2, HPIX = HH * HW;       // halo rows / pixels
    static constexpr int NT = 64 * TH / RPW;                // threads: one wave per RPW output rows
    static constexpr int PS = CIN + 8;                      // pixel stride in elements (80 / 144 bytes: spreads the banks)
    static constexpr int CHUNKS = HPIX * (CIN / 8);         // 16-byte pieces of one halo
    static constexpr int NCH = (CHUNKS + NT - 1) / NT;      // per thread
};

// one 16-byte piece of the halo of tile (b, oy0, ox0): chunk -> (halo pixel, channel group); zero outside the (virtual) image
template <int CIN, int TH, bool UP>
__device__ __forceinline__ u32x4 halo_fetch(const __bf16 *__restrict__ x, int chunk, int b, int oy0, int ox0, int Hs, int Ws, int Hv, int Wv) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    if (chunk >= Halo<CIN, TH>::CHUNKS) return z;
    const int hp = chunk / (CIN / 8), part = chunk - hp * (CIN / 8);
    const int hy = hp / HW, hx = hp - hy * HW;
    const int vy = oy0 - 1 + hy, vx = ox0 - 1 + hx;
    if ((unsigned)vy >= (unsigned)Hv || (unsigned)vx >= (unsigned)Wv) return z;
    const int sy = UP ? (vy >> 1) : vy, sx = UP ? (vx >> 1) : vx;
    return *(const u32x4 *)(x + (((size_t)b * Hs + sy) * Ws + sx) * CIN + part * 8);
}

template <int CIN, int TH>
__device__ __forceinline__ void halo_put(__bf16 *halo, int chunk, const u32x4 &v) {
    if (chunk < Halo<CIN, TH>::CHUNKS) {
        const int hp = chunk / (CIN / 8), part = chunk - hp * (CIN / 8);
        *(u32x4 *)(halo + hp * Halo<CIN, TH>::PS + part * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------------ forward / data gradient
// x: [B][Hs][Ws][CIN] (Hs = Hv / 2 when UP), w: [COUT][3][3][CIN] bf16 (already scaled / transposed by gwd_weight_prep),
// y: [B][Hv][Wv][COUT].  FLIP: the data-gradient gather (source = pixel + 1 - tap).
// ACTK: the epilogue's activation as a compile-time constant (GWD_ACT_NONE: data gradients, GWD_ACT_ELU: the decoder) or -1 = decided at
// run time - the four-way switch with its erf / expm1 / exp paths, inlined once per output register, was most of the epilogue's code.
template <int CIN, int COUT, int TH, bool UP, bool FLIP, int RPW = 2, int ACTK = -1>
__global__ __launch_bounds__(64 * TH / RPW) void tconv_fwd_kernel(const gwd_conv_desc d, int tiles_y, int tiles_x, int ntiles) {
    // RPW = output rows per wave: 2 halves the weight-fragment reads per MFMA; 1 doubles the waves of a workgroup - for the layers whose
    // LDS footprint (weights + halo) allows ONE workgroup per CU, that is the difference between one and two waves per SIMD
    using H = Halo<CIN, TH, RPW>;
    constexpr int NT = H::NT;
    constexpr int NTL = (COUT + 31) / 32, COUTP = 32 * NTL;  // output-channel tiles; rows COUT..COUTP-1 of the weight image are zero
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    __bf16 *halo = lds, *wl = lds + H::HPIX * H::PS;        // weights [tap][COUTP][PS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const __bf16 *x = (const __bf16 *)d.x, *w = (const __bf16 *)d.w;
    __bf16 *y = (__bf16 *)d.y;
    const int Hv = d.Ho, Wv = d.Wo, Hs = d.Hi, Ws = d.Wi;   // same-size convolution on the (virtual) grid Hv x Wv
    if constexpr (COUTP != COUT)
        for (int c = tid; c < 9 * COUTP * (CIN / 8); c += NT) *(u32x4 *)(wl + (c / (CIN / 8)) * H::PS + (c % (CIN / 8)) * 8) = u32x4{0u, 0u, 0u, 0u};
    if constexpr (COUTP != COUT) __syncthreads();
    for (int c = tid; c < 9 * COUT * (CIN / 8); c += NT) {
        const int part = c % (CIN / 8), row = c / (CIN / 8), co = row / 9, tap = row - co * 9;      // source row = (co, tap)
        *(u32x4 *)(wl + (tap * COUTP + co) * H::PS + part * 8) = *(const u32x4 *)(w + (size_t)row * CIN + part * 8);
    }
    float sh[NTL][16];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) sh[nt][i] = (d.shift && 32 * nt + acc_row(i, h) < COUT) ? d.shift[32 * nt + acc_row(i, h)] : 0.f;

    u32x4 pre[H::NCH];
    int tile = blockIdx.x;
    auto where = [&](int t, int &b, int &oy0, int &ox0) {
        const int per = tiles_y * tiles_x;
        b = t / per;
        const int rem = t - b * per;
        oy0 = (rem / tiles_x) * TH;
        ox0 = (rem % tiles_x) * TW;
    };
    if (tile < ntiles) {
        int b, oy0, ox0;
        where(tile, b, oy0, ox0);
#pragma unroll
        for (int i = 0; i < H::NCH; ++i) pre[i] = halo_fetch<CIN, TH, UP>(x, tid + NT * i, b, oy0, ox0, Hs, Ws, Hv, Wv);
    }
    for (; tile < ntiles; tile += gridDim.x) {
        int b, oy0, ox0;
        where(tile, b, oy0, ox0);
        __syncthreads();                                     // every wave is done with the previous halo (and, first time, the weights are in)
#pragma unroll
        for (int i = 0; i < H::NCH; ++i) halo_put<CIN, TH>(halo, tid + NT * i, pre[i]);
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) {               // the next tile's halo travels during the MFMAs
            int nb, ny, nx;
            where(tile + gridDim.x, nb, ny, nx);
#pragma unroll
            for (int i = 0; i < H::NCH; ++i) pre[i] = halo_fetch<CIN, TH, UP>(x, tid + NT * i, nb, ny, nx, Hs, Ws, Hv, Wv);
        }
        f32x16 acc[RPW][NTL];
#pragma unroll
        for (int mt = 0; mt < RPW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - 3 * kh;
            const int dy = FLIP ? 2 - kh : kh, dx = FLIP ? 2 - kw : kw;      // halo offset of the tap (0..2)
#pragma unroll
            for (int s = 0; s < CIN / 16; ++s) {
                bf16x8 bf[RPW];
#pragma unroll
                for (int mt = 0; mt < RPW; ++mt)
                    bf[mt] = *(const bf16x8 *)(halo + ((RPW * wave + mt + dy) * HW + r + dx) * H::PS + 16 * s + 8 * h);
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) {
                    const bf16x8 af = *(const bf16x8 *)(wl + (tap * COUTP + 32 * nt + r) * H::PS + 16 * s + 8 * h);
#pragma unroll
                    for (int mt = 0; mt < RPW; ++mt) acc[mt][nt] = mma(af, bf[mt], acc[mt][nt]);
                }
            }
        }
        // epilogue: lane = pixel, registers 4g..4g+3 = channels 32 nt + 8 g + 4 h + 0..3
#pragma unroll
        for (int mt = 0; mt < RPW; ++mt) {
            const int oy = oy0 + RPW * wave + mt, ox = ox0 + r;
            if (oy < Hv && ox < Wv) {
                __bf16 *dst = y + (((size_t)b * Hv + oy) * Wv + ox) * COUT;
                const __bf16 *rsd = d.residual ? (const __bf16 *)d.residual + (((size_t)b * Hv + oy) * Wv + ox) * COUT : nullptr;
                const __bf16 *gte = d.gate ? (const __bf16 *)d.gate + (((size_t)b * Hv + oy) * Wv + ox) * COUT : nullptr;
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (32 * nt + 8 * g + 4 * h >= COUT) continue;         // COUT = 16: the upper half of the channel tile is padding
                        union { uint2 u; __bf16 e[4]; } v, rr;
                        rr.u = uint2{0u, 0u};
                        if (rsd) rr.u = *(const uint2 *)(rsd + 32 * nt + 8 * g + 4 * h);         // added before the activation, as the implicit GEMM does
                        float f[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) f[j] = apply_act(acc[mt][nt][4 * g + j] + sh[nt][4 * g + j] + (float)rr.e[j], ACTK < 0 ? d.act : ACTK) * d.act_scale;
                        if (gte) {                                             // backward of the producer's activation (desc.gate), last
                            union { uint2 u; __bf16 e[4]; } gg;
                            gg.u = *(const uint2 *)(gte + 32 * nt + 8 * g + 4 * h);
#pragma unroll
                            for (int j = 0; j < 4; ++j) f[j] = gate_grad(f[j], (float)gg.e[j], d.gate_act);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) v.e[j] = (__bf16)f[j];
                        *(uint2 *)(dst + 32 * nt + 8 * g + 4 * h) = v.u;
                    }
            }
        }
    }
}

template <int CIN, int COUT, int TH>
size_t fwd_lds() { return (size_t)(Halo<CIN, TH>::HPIX * Halo<CIN, TH>::PS + 9 * ((COUT + 31) / 32 * 32) * Halo<CIN, TH>::PS) * 2; }

static bool enabled() { return true; }       // (against the tap-by-tap implicit GEMM for these layers: -0.5 ms per step, round 2)

template <int CIN, int COUT, bool UP, bool FLIP>
int launch_fwd(const gwd_conv_desc *d, hipStream_t s) {
    // 64 -> 32 (90 KB of LDS at 8 rows = one 4-wave workgroup per CU): 16-row tiles with 8 waves double the waves per CU for the same
    // weights in LDS (277 -> 162 us on the up-convolution); the 32-channel-input variants fit 2-3 workgroups per CU at 8 rows and are
    // faster there (104 vs 124 us), 64 -> 64 has no room for a 16-row halo beside its 83 KB of weights
    constexpr int TH = (CIN == 64 && COUT == 32) ? 16 : 8;
    const int ty = (d->Ho + TH - 1) / TH, tx = (d->Wo + TW - 1) / TW;
    const long ntiles = (long)d->B * ty * tx;
    const size_t lds = fwd_lds<CIN, COUT, TH>();
    const int per_cu = lds > 80 * 1024 ? 1 : (lds > 53 * 1024 ? 2 : 3);
    long grid = 256L * per_cu;
    if (grid > ntiles) grid = ntiles;
    constexpr int rpw1 = 1;                                     // bit mask of the layers that run one row per wave: 1: 64 -> 64 (measured), 2: 64 -> 32, 4: 32 -> 64, 8: 32 -> 32
    constexpr int bit = (CIN == 64 && COUT == 64) ? 1 : (CIN == 64 && COUT == 32) ? 2 : (CIN == 32 && COUT == 64) ? 4 : (CIN == 32 && COUT == 32) ? 8 : 0;
#define TC_GO(RPW_, ACTK_)                                                                                                       \
    {                                                                                                                            \
        static bool attr = false;                                                                                                \
        if (!attr) {                                            /* dynamic LDS beyond 64 KiB has to be requested once per kernel */ \
            (void)hipFuncSetAttribute((const void *)tconv_fwd_kernel<CIN, COUT, TH, UP, FLIP, RPW_, ACTK_>,                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                     \
            attr = true;                                                                                                         \
        }                                                                                                                        \
        tconv_fwd_kernel<CIN, COUT, TH, UP, FLIP, RPW_, ACTK_><<<(unsigned)grid, 64 * TH / RPW_, lds, s>>>(*d, ty, tx, (int)ntiles); \
        return 1;                                                                                                                \
    }
#define TC_ACT(RPW_)                                                                                                             \
    if (d->act == GWD_ACT_NONE) TC_GO(RPW_, GWD_ACT_NONE)                                                                        \
    if (d->act == GWD_ACT_ELU) TC_GO(RPW_, GWD_ACT_ELU)                                                                          \
    TC_GO(RPW_, -1)
    if constexpr (bit != 0) {
        if (rpw1 & bit) { TC_ACT(1) }
    }
    TC_ACT(2)
#undef TC_ACT
#undef TC_GO
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[cg][tap][cx] += sum over pixels of gy[p][cg] * x[p + tap - 1][cx]  (x through the same halo, optionally 2x up-sampled).
// 3 waves, 3 taps each; per tile 16 k-steps of 16 consecutive pixels; D^T-free orientation: A = gy^T (rows = gy channels),
// B = shifted x (columns = x channels), both column gathers (ds_read_b64_tr_b16, natural k order) from [pixel][channel] images.
__device__ __forceinline__ bf16x8 gather16(const __bf16 *img, int rs, int pix0, int col0, int lane) {
    // lane (r, h) gets img[pix0 + 8 h + j][col0 + r], j = 0..7 (32 columns from col0)
    const int g = lane >> 4, t = lane & 15, q = t >> 2, p = t & 3;
    const __bf16 *a = img + (pix0 + 8 * (g >> 1) + q) * rs + col0 + 16 * (g & 1) + 4 * p;
    union { s16x4 h[2]; bf16x8 v; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)a);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(a + 4 * rs));
    return u.v;
}

template <int CX, int CG, bool UP, int TPW = 3>
__global__ __launch_bounds__(64 * 9 / TPW) void tconv_wgrad_kernel(const gwd_conv_desc d, float *__restrict__ dw, int tiles_y, int tiles_x, int ntiles) {
    constexpr int TH = 8, HPIX = (TH + 2) * HW, NTW = 64 * 9 / TPW;     // TPW taps per wave: 3 waves x 3 taps, or 9 waves x 1 tap
    using H = Halo<CX, TH>;
    constexpr int GS = CG + 8;                              // gy image pixel stride
    constexpr int GCH = TH * TW * (CG / 8), NG = (GCH + NTW - 1) / NTW, NX = (H::CHUNKS + NTW - 1) / NTW;
    constexpr int TX = (CX + 31) / 32, TG = (CG + 31) / 32;   // 16 channels: the column gathers of a 32-wide tile run on into the next
                                                             // pixel's bytes - those rows / columns of the product are never flushed
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    __bf16 *halo = lds, *gimg = lds + HPIX * H::PS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __bf16 *x = (const __bf16 *)d.x, *gy = (const __bf16 *)d.y;      // gwd_conv_wgrad: the desc's y slot carries the output gradient
    const int Hv = d.Ho, Wv = d.Wo, Hs = d.Hi, Ws = d.Wi;
    f32x16 acc[TPW][TG][TX];
#pragma unroll
    for (int a = 0; a < TPW; ++a)
#pragma unroll
        for (int b = 0; b < TG; ++b)
#pragma unroll
            for (int c = 0; c < TX; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][c][i] = 0.f;
    auto where = [&](int t, int &b, int &oy0, int &ox0) {
        const int per = tiles_y * tiles_x;
        b = t / per;
        const int rem = t - b * per;
        oy0 = (rem / tiles_x) * TH;
        ox0 = (rem % tiles_x) * TW;
    };
    auto gy_fetch = [&](int chunk, int b, int oy0, int ox0) -> u32x4 {
        const u32x4 z = {0u, 0u, 0u, 0u};
        if (chunk >= GCH) return z;
        const int pix = chunk / (CG / 8), part = chunk - pix * (CG / 8);
        const int oy = oy0 + pix / TW, ox = ox0 + pix % TW;
        if (oy >= Hv || ox >= Wv) return z;
        return *(const u32x4 *)(gy + (((size_t)b * Hv + oy) * Wv + ox) * CG + part * 8);
    };
    u32x4 px[NX], pg[NG];
    int tile = blockIdx.x;
    if (tile < ntiles) {
        int b, oy0, ox0;
        where(tile, b, oy0, ox0);
#pragma unroll
        for (int i = 0; i < NX; ++i) px[i] = halo_fetch<CX, 8, UP>(x, tid + NTW * i, b, oy0, ox0, Hs, Ws, Hv, Wv);
#pragma unroll
        for (int i = 0; i < NG; ++i) pg[i] = gy_fetch(tid + NTW * i, b, oy0, ox0);
    }
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NX; ++i) halo_put<CX, 8>(halo, tid + NTW * i, px[i]);
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int chunk = tid + NTW * i;
            if (chunk < GCH) *(u32x4 *)(gimg + (chunk / (CG / 8)) * GS + (chunk % (CG / 8)) * 8) = pg[i];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) {
            int nb, ny, nx;
            where(tile + gridDim.x, nb, ny, nx);
#pragma unroll
            for (int i = 0; i < NX; ++i) px[i] = halo_fetch<CX, 8, UP>(x, tid + NTW * i, nb, ny, nx, Hs, Ws, Hv, Wv);
#pragma unroll
            for (int i = 0; i < NG; ++i) pg[i] = gy_fetch(tid + NTW * i, nb, ny, nx);
        }
#pragma unroll 2
        for (int ks = 0; ks < 16; ++ks) {                   // 16 consecutive pixels of tile row ks / 2
            const int row = ks >> 1, x0 = (ks & 1) * 16;
            bf16x8 af[TG];
#pragma unroll
            for (int b = 0; b < TG; ++b) af[b] = gather16(gimg, GS, row * TW + x0, 32 * b, lane);
#pragma unroll
            for (int a = 0; a < TPW; ++a) {
                const int tap = TPW * wave + a, kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
                for (int c = 0; c < TX; ++c) {
                    const bf16x8 bf = gather16(halo, H::PS, (row + kh) * HW + x0 + kw, 32 * c, lane);
#pragma unroll
                    for (int b = 0; b < TG; ++b) acc[a][b][c] = mma(af[b], bf, acc[a][b][c]);
                }
            }
        }
    }
    // flush: D[cg][cx] of tap -> dw[cg][tap][cx] (fp32 atomics, lanes = consecutive cx)
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int a = 0; a < TPW; ++a) {
        const int tap = TPW * wave + a;
#pragma unroll
        for (int b = 0; b < TG; ++b)
#pragma unroll
            for (int c = 0; c < TX; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int cg = 32 * b + acc_row(i, h), cx = 32 * c + r;
                    if (cg >= CG || cx >= CX) continue;
                    const float sc = d.scale ? d.scale[cg] : 1.0f;
                    unsafeAtomicAdd(dw + ((size_t)cg * 9 + tap) * CX + cx, acc[a][b][c][i] * sc);
                }
    }
}

template <int CX, int CG, bool UP>
int launch_wgrad(const gwd_conv_desc *d, float *dw, hipStream_t s) {
    constexpr int TH = 8;
    const int ty = (d->Ho + TH - 1) / TH, tx = (d->Wo + TW - 1) / TW;
    const long ntiles = (long)d->B * ty * tx;
    const size_t lds = (size_t)(Halo<CX, TH>::HPIX * Halo<CX, TH>::PS + TH * TW * (CG + 8) + 64) * 2;      // + slack for the run-on gathers of the 16-channel case
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)tconv_wgrad_kernel<CX, CG, UP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int per_cu = lds > 53 * 1024 ? 2 : 3;
    long grid = 256L * per_cu;
    // every workgroup ends with CG x 9 x CX fp32 atomics onto the SAME few thousand addresses: with 768 workgroups that flush
    // (768-way contention per address) outweighs the tile loop for the small weights; cap the workgroup count by the weight size
    // measured (tools/convbench.py, grid sweep): 32 x 32 at 480 x 640: 154 us with 768 workgroups, 116 with 512;
    // 64 x 32 at 240 x 320: 125 -> 86 us with 256
    const long cap = CX * CG <= 1024 ? 512 : 256;
    if (grid > cap) grid = cap;
    if (grid > ntiles) grid = ntiles;
    // nine waves with one filter tap each instead of three with three: 64 x 32 at 240 x 320 85 -> 52 us (more waves per CU for the
    // same LDS), 32 x 32 at 480 x 640 120 -> 126 us - so only the 64-channel input runs that way
    if (CX == 64) {
        static bool attr1 = false;
        if (!attr1) {
            (void)hipFuncSetAttribute((const void *)tconv_wgrad_kernel<CX, CG, UP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr1 = true;
        }
        tconv_wgrad_kernel<CX, CG, UP, 1><<<(unsigned)grid, 576, lds, s>>>(*d, dw, ty, tx, (int)ntiles);
        return 1;
    }
    tconv_wgrad_kernel<CX, CG, UP><<<(unsigned)grid, 192, lds, s>>>(*d, dw, ty, tx, (int)ntiles);
    return 1;
}

}  // namespace tconv

// (Round 2 also tried the same register-resident-weights idea for the 1x1 / Linear layers with K <= 256 on 153 600 rows - each wave
//  streaming 32-row tiles of x straight from global memory as B fragments, lane = row: correct, but 23 -> 28 us (64 -> 128) and
//  33 -> 72 us (64 -> 256) against the LDS-DMA kernel: a lane-per-row 16-byte load touches 32 different 128-byte lines per wave
//  instruction, a quarter of the coalesced rate.  Removed; those layers stay on igemm_dma_kernel.)

// 1 = launched, 0 = not one of these layers (the caller continues with the generic kernels)
int gwd_tile_conv_forward(const gwd_conv_desc *d, hipStream_t s) {
    using namespace tconv;
    if (!enabled() || d->dtype != GWD_BF16 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1) return 0;
    if (d->scale || d->z || d->mult || !d->w) return 0;
    if (d->residual && ((uintptr_t)d->residual % 8)) return 0;
    if (d->gate && (((uintptr_t)d->gate % 8) || d->gate_act == GWD_ACT_GELU)) return 0;
    if ((long)d->B * d->Ho * d->Wo < 131072 || (long)d->B * d->Ho * d->Wo >= (1L << 31)) return 0;
    if (((uintptr_t)d->x | (uintptr_t)d->w | (uintptr_t)d->y) % 16) return 0;
    const bool up = d->gather == GWD_GATHER_UPSAMPLED, flip = d->gather == GWD_GATHER_TRANSPOSED;
    if (up) {
        if (d->Hv != 2 * d->Hi || d->Wv != 2 * d->Wi || d->Ho != d->Hv || d->Wo != d->Wv) return 0;
    } else if (d->Ho != d->Hi || d->Wo != d->Wi) return 0;
    const int ci = d->Cin, co = d->Cout;
#define TC(CI, CO)                                                                                  \
    if (ci == CI && co == CO) {                                                                     \
        if (up) return launch_fwd<CI, CO, true, false>(d, s);                                       \
        if (flip) return launch_fwd<CI, CO, false, true>(d, s);                                     \
        return launch_fwd<CI, CO, false, false>(d, s);                                              \
    }
    TC(32, 32) TC(32, 64) TC(64, 32) TC(64, 64) TC(16, 16)
#undef TC
    return 0;
}

// gwd_conv_wgrad's thin-channel case (desc.x = layer input, desc.y = output gradient, Cin = x channels, Cout = gy channels)
int gwd_tile_conv_wgrad(const gwd_conv_desc *d, float *dw, hipStream_t s) {
    using namespace tconv;
    if (!enabled() || d->dtype != GWD_BF16 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1) return 0;
    if ((long)d->B * d->Ho * d->Wo < 131072 || (long)d->B * d->Ho * d->Wo >= (1L << 31)) return 0;
    if (((uintptr_t)d->x | (uintptr_t)d->y) % 16) return 0;
    const bool up = d->gather == GWD_GATHER_UPSAMPLED;
    if (up) {
        if (d->Hv != 2 * d->Hi || d->Wv != 2 * d->Wi || d->Ho != d->Hv || d->Wo != d->Wv) return 0;
    } else if (d->gather != GWD_GATHER_CONV || d->Ho != d->Hi || d->Wo != d->Wi) return 0;
    const int cx = d->Cin, cg = d->Cout;
#define TW_(CX_, CG_)                                                                               \
    if (cx == CX_ && cg == CG_) return up ? launch_wgrad<CX_, CG_, true>(d, dw, s) : launch_wgrad<CX_, CG_, false>(d, dw, s);
    TW_(32, 32) TW_(64, 32)        // (16, 16) stays on the generic kernel: 19 us there, 21-35 us here (its 2 304 weights are all atomics)
#undef TW_
    return 0;
}
