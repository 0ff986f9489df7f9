// Implicit-GEMM convolution / linear layers on the CDNA4 matrix cores.
//
//   forward / data-gradient :  Y[m][n] = act( scale[n] * sum_k A[m][k] * W[n][k] + shift[n] + R[m][n] )
//   weight-gradient         :  dW[n][k] += sum_m dY[m][n] * A[m][k]
//
// with m = (b, oh, ow) an output pixel, k = (kh, kw, c) and A the im2col view of an NHWC input that
// is never materialised: every thread gathers 16-byte channel vectors straight from HBM, the gather
// rule (plain / transposed for strided data gradients / nearest-upsampled input) is a scalar switch.
// Tiles are staged through LDS (register-staged, double-buffered, one barrier per K step) and
// consumed by 64-wide waves with v_mfma_f32_32x32x16_bf16 (bf16 storage, fp32 accumulate) or the
// exact-fp32 v_mfma_f32_32x32x2_f32 (parity mode).  The epilogue fuses FrozenBN affine / bias,
// residual add and ReLU / GELU / ELU / sigmoid so a Bottleneck conv writes its activation once.
#include "common.h"
#include <cmath>

namespace {

template <typename T> struct Cfg;
template <> struct Cfg<float> { static constexpr int VEC = 4, BK = 16; };
template <> struct Cfg<__bf16> { static constexpr int VEC = 8, BK = 32; };

__device__ __forceinline__ bool src_pixel(const gwd_conv_desc &d, int oh, int ow, int kh, int kw, int &ih, int &iw) {
    if (d.gather == GWD_GATHER_CONV) {
        ih = oh * d.stride - d.pad + kh;
        iw = ow * d.stride - d.pad + kw;
        return (unsigned)ih < (unsigned)d.Hi && (unsigned)iw < (unsigned)d.Wi;
    } else if (d.gather == GWD_GATHER_TRANSPOSED) {
        const int th = oh + d.pad - kh, tw = ow + d.pad - kw;
        if (th < 0 || tw < 0) return false;
        if (d.stride == 1) {
            ih = th;
            iw = tw;
        } else {
            ih = th / d.stride;
            iw = tw / d.stride;
            if (ih * d.stride != th || iw * d.stride != tw) return false;
        }
        return ih < d.Hi && iw < d.Wi;
    } else {  // GWD_GATHER_UPSAMPLED: legacy 'nearest' rule floor(dst * in/out), as aten upsample_nearest2d
        const int vh = oh - d.pad + kh, vw = ow - d.pad + kw;
        if ((unsigned)vh >= (unsigned)d.Hv || (unsigned)vw >= (unsigned)d.Wv) return false;
        ih = min((int)floorf((float)vh * ((float)d.Hi / (float)d.Hv)), d.Hi - 1);
        iw = min((int)floorf((float)vw * ((float)d.Wi / (float)d.Wv)), d.Wi - 1);
        return true;
    }
}

// One 16-byte vector of the im2col row of output pixel (b,oh,ow) starting at flattened k0.
template <typename T>
__device__ __forceinline__ uint4 gather_vec(const gwd_conv_desc &d, bool row_ok, int b, int oh, int ow, int k0, int K,
                                            bool fast, bool half) {
    constexpr int VEC = Cfg<T>::VEC;
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
    if (!row_ok || k0 >= K) return r;
    const T *x = (const T *)d.x;
    if (half) {                 // Cin % (VEC/2) == 0 (e.g. the 60 / 300 channel pyramids): two 8-byte pieces, one tap each
        uint2 h[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = k0 + e * (VEC / 2);
            if (k < K) {
                const int tap = k / d.Cin, c = k - tap * d.Cin;
                const int kh = tap / d.KW, kw = tap - kh * d.KW;
                int ih, iw;
                if (src_pixel(d, oh, ow, kh, kw, ih, iw))
                    h[e] = *(const uint2 *)(x + ((size_t)(b * d.Hi + ih) * d.Wi + iw) * d.Cin + c);
            }
        }
        return make_uint4(h[0].x, h[0].y, h[1].x, h[1].y);
    }
    if (fast) {
        const int tap = k0 / d.Cin, c = k0 - tap * d.Cin;
        const int kh = tap / d.KW, kw = tap - kh * d.KW;
        int ih, iw;
        if (src_pixel(d, oh, ow, kh, kw, ih, iw))
            r = *(const uint4 *)(x + ((size_t)(b * d.Hi + ih) * d.Wi + iw) * d.Cin + c);
    } else {
        T tmp[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int k = k0 + e;
            T v = from_f32<T>(0.f);
            if (k < K) {
                const int tap = k / d.Cin, c = k - tap * d.Cin;
                const int kh = tap / d.KW, kw = tap - kh * d.KW;
                int ih, iw;
                if (src_pixel(d, oh, ow, kh, kw, ih, iw)) v = x[((size_t)(b * d.Hi + ih) * d.Wi + iw) * d.Cin + c];
            }
            tmp[e] = v;
        }
        r = *(uint4 *)tmp;
    }
    return r;
}

// 16 bytes of row `n` of a row-major [rows][K] matrix starting at column k0 (zero outside).
template <typename T>
__device__ __forceinline__ uint4 row_vec(const T *base, int n, int N, int k0, int K, bool fast, bool half = false) {
    constexpr int VEC = Cfg<T>::VEC;
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
    if (n >= N || k0 >= K) return r;
    const T *p = base + (size_t)n * K + k0;
    if (fast) return *(const uint4 *)p;
    if (half) {                 // K % (VEC/2) == 0: the row is 8-byte aligned, the tile's last vector may be half full
        const uint2 lo = *(const uint2 *)p;
        const uint2 hi = (k0 + VEC / 2 < K) ? *(const uint2 *)(p + VEC / 2) : make_uint2(0u, 0u);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    T tmp[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) tmp[e] = (k0 + e < K) ? p[e] : from_f32<T>(0.f);
    return *(uint4 *)tmp;
}

__device__ __forceinline__ f32x16 mma(const bf16x8 &a, const bf16x8 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma(float a, float b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ----------------------------------------------------------------------------------------------
// forward / data gradient
// ----------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, int BK, bool MULT = false>
__global__ __launch_bounds__(256) void igemm_fwd_kernel(const gwd_conv_desc d) {
    constexpr int VEC = Cfg<T>::VEC;
    constexpr int KV = BK / VEC;        // 16-byte vectors per tile row (4)
    constexpr int LDK = BK + VEC;       // padded LDS row: 80 bytes, conflict-free ds_read_b128
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int ROWS_PER_PASS = 256 / KV;  // 64
    constexpr int A_IT = BM / ROWS_PER_PASS;
    constexpr int B_IT = (BN + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
    constexpr int TILE_ELEMS = 2 * (BM + BN) * LDK;
    constexpr int STAGE_ELEMS = 4 * 32 * 36 * (int)(sizeof(float) / sizeof(T));   // epilogue staging (bf16 only)
    static_assert(WM * WN == 4, "4 waves");

    __shared__ __attribute__((aligned(16))) T smem[TILE_ELEMS > STAGE_ELEMS ? TILE_ELEMS : STAGE_ELEMS];
    T *As = smem;
    T *Bs = smem + 2 * BM * LDK;

    const int M = d.B * d.Ho * d.Wo, N = d.Cout, K = d.KH * d.KW * d.Cin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const bool fastA = (d.Cin % VEC) == 0, fastB = (K % VEC) == 0;
    const bool halfA = !fastA && (d.Cin % (VEC / 2)) == 0, halfB = !fastB && (K % (VEC / 2)) == 0;
    // When Cin is a multiple of BK every K tile lies inside ONE filter tap: (kh, kw, c0) are then
    // workgroup-uniform (scalar registers, advanced incrementally) and the per-thread gather costs a
    // bounds test and one multiply-add chain instead of two integer divisions per 16-byte vector.
    const bool uni = (d.Cin % BK) == 0;

    const int lv = tid % KV, lr = tid / KV;
    int a_b[A_IT], a_oh[A_IT], a_ow[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + lr + i * ROWS_PER_PASS;
        a_ok[i] = m < M;
        const int mm = a_ok[i] ? m : 0;
        a_b[i] = mm / (d.Ho * d.Wo);
        const int rem = mm - a_b[i] * (d.Ho * d.Wo);
        a_oh[i] = rem / d.Wo;
        a_ow[i] = rem - a_oh[i] * d.Wo;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int KT = (K + BK - 1) / BK;
    uint4 ra[A_IT], rb[B_IT];
    int u_kh = 0, u_kw = 0, u_c0 = 0;      // tap of the NEXT tile to load (uniform path)

    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK + lv * VEC;
        if (uni) {
            const T *x = (const T *)d.x;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                int ih, iw;
                uint4 r = make_uint4(0u, 0u, 0u, 0u);
                if (a_ok[i] && src_pixel(d, a_oh[i], a_ow[i], u_kh, u_kw, ih, iw))
                    r = *(const uint4 *)(x + ((size_t)(a_b[i] * d.Hi + ih) * d.Wi + iw) * d.Cin + u_c0 + lv * VEC);
                ra[i] = r;
            }
            u_c0 += BK;
            if (u_c0 >= d.Cin) {
                u_c0 = 0;
                if (++u_kw == d.KW) {
                    u_kw = 0;
                    ++u_kh;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) ra[i] = gather_vec<T>(d, a_ok[i], a_b[i], a_oh[i], a_ow[i], k0, K, fastA, halfA);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int row = lr + i * ROWS_PER_PASS;
            rb[i] = (row < BN) ? row_vec<T>((const T *)d.w, n0 + row, N, k0, K, fastB, halfB) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            *(uint4 *)(As + (size_t)buf * BM * LDK + (lr + i * ROWS_PER_PASS) * LDK + lv * VEC) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int row = lr + i * ROWS_PER_PASS;
            if (row < BN) *(uint4 *)(Bs + (size_t)buf * BN * LDK + row * LDK + lv * VEC) = rb[i];
        }
    };
    const int fr = lane & 31, fh = lane >> 5;
    auto compute = [&](int cur) {
        const T *Ab = As + (size_t)cur * BM * LDK + (wm * (BM / WM) + fr) * LDK;
        const T *Bb = Bs + (size_t)cur * BN * LDK + (wn * (BN / WN) + fr) * LDK;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *(const bf16x8 *)(Ab + i * 32 * LDK + ks * 16 + fh * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = *(const bf16x8 *)(Bb + j * 32 * LDK + ks * 16 + fh * 8);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mma(af[i], bfr[j], acc[i][j]);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                float af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = Ab[i * 32 * LDK + ks * 2 + fh];
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = Bb[j * 32 * LDK + ks * 2 + fh];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mma(af[i], bfr[j], acc[i][j]);
            }
        }
    };

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) load_tiles(kt + 1);
        compute(cur);
        if (kt + 1 < KT) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    T *y = (T *)d.y;
    T *z = (T *)d.z;
    const T *res = (const T *)d.residual;
    const T *mul = MULT ? (const T *)d.mult : nullptr;
    const T *gate = MULT ? nullptr : (const T *)d.gate;        // never together with a multiplier (check_desc)
    if constexpr (sizeof(T) == 2) {
        if ((N & 7) == 0) {
            // each wave transposes one 32x32 accumulator tile at a time through a private LDS patch, then every
            // lane emits 16-byte stores of 8 consecutive channels (8x fewer store instructions than the
            // column-per-lane layout; scale/shift/residual are read as vectors too)
            float *stage = (float *)smem + wave * (32 * 36);
            const int vr = lane >> 2, vc = (lane & 3) * 8;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * fh) * 36 + fr] = acc[i][j][r];
                    __builtin_amdgcn_wave_barrier();
                    const int nb = n0 + wn * (BN / WN) + j * 32 + vc;
                    if (nb < N) {
                        float sc[8], sh[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            sc[e] = d.scale ? d.scale[nb + e] : 1.0f;
                            sh[e] = d.shift ? d.shift[nb + e] : 0.0f;
                        }
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            const int row = vr + 16 * half;
                            const int m = m0 + wm * (BM / WM) + i * 32 + row;
                            if (m >= M) continue;
                            const f32x4 lo = *(const f32x4 *)(stage + row * 36 + vc);
                            const f32x4 hi = *(const f32x4 *)(stage + row * 36 + vc + 4);
                            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            const size_t o = (size_t)m * N + nb;
                            if (res && !mul) {
                                const bf16x8 rv = *(const bf16x8 *)(res + o);
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e] + (float)rv[e];
                            } else {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
                            }
                            bf16x8 out;
                            if (z) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) out[e] = (__bf16)v[e];
                                *(bf16x8 *)(z + o) = out;
                            }
                            if (mul) {                   // y = act_scale * act(v) * mult + residual (dropout, then the skip)
                                const bf16x8 mv = *(const bf16x8 *)(mul + o);
                                bf16x8 rv;
#pragma unroll
                                for (int e = 0; e < 8; ++e) rv[e] = (__bf16)0.0f;
                                if (res) rv = *(const bf16x8 *)(res + o);
#pragma unroll
                                for (int e = 0; e < 8; ++e) out[e] = (__bf16)(apply_act(v[e], d.act) * d.act_scale * (float)mv[e] + (float)rv[e]);
                            } else {
                                float gm[8];                 // desc.gate: backward of the producer's activation, from its output
#pragma unroll
                                for (int e = 0; e < 8; ++e) gm[e] = d.act_scale;
                                if (gate) {
                                    const bf16x8 gv = *(const bf16x8 *)(gate + o);
#pragma unroll
                                    for (int e = 0; e < 8; ++e) gm[e] = gate_grad(d.act_scale, (float)gv[e], d.gate_act);
                                }
#pragma unroll
                                for (int e = 0; e < 8; ++e) out[e] = (__bf16)(apply_act(v[e], d.act) * gm[e]);
                            }
                            *(bf16x8 *)(y + o) = out;
                        }
                    }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WN) + j * 32 + fr;
        if (n >= N) continue;
        const float sc = d.scale ? d.scale[n] : 1.0f;
        const float sh = d.shift ? d.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m >= M) continue;
                const size_t o = (size_t)m * N + n;
                float v = acc[i][j][r] * sc + sh;
                if (res && !mul) v += to_f32(res[o]);
                if (z) z[o] = from_f32<T>(v);
                v = apply_act(v, d.act) * d.act_scale;
                if (mul) v = v * to_f32(mul[o]) + (res ? to_f32(res[o]) : 0.f);
                if (gate) v = gate_grad(v, to_f32(gate[o]), d.gate_act);
                y[o] = from_f32<T>(v);
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// forward / data gradient, LDS-DMA pipeline (bf16, Cin % 32 == 0)
// ----------------------------------------------------------------------------------------------
// Same tiling as igemm_fwd_kernel, but tiles travel HBM -> LDS with global_load_lds_dwordx4 (no VGPR staging, no
// ds_write) through a ring of STAGES buffers: the gathers of tiles t+1 .. t+STAGES-1 are in flight while tile t
// feeds the MFMAs, retired by a COUNTED s_waitcnt vmcnt(N) in front of one raw s_barrier per K step.  An LDS-DMA
// wave instruction writes 64 x 16 B linearly (16 rows of 64 B), so the bank-conflict swizzle is applied to the
// per-lane SOURCE address (chunk c of row r lands at position c ^ ((r >> 2) & 3)) and again on the fragment read.
// Out-of-image taps / rows are fetched from a caller-provided zero page, which keeps every lane active.
// GM: 0 = plain conv gather, 1 = transposed gather with stride 1 (data gradient of a stride-1 layer), 2 = any
// (decided at run time): the two hot cases get a straight-line address path.
// Workgroups are dispatched round-robin over the 8 XCDs (each with a private L2): remap the hardware workgroup id so
// that every XCD works on one CONTIGUOUS band of the logical tile order; tiles that re-read the same operand bytes
// (filter-tap halos of neighbouring pixel strips, the column tiles of one row tile, the tiles of one reduction split)
// are then neighbours in one L2 instead of eight HBM readers.
__device__ __forceinline__ int xcd_band(int id, int total) {
    const int xcd = id & 7, idx = id >> 3, per = total >> 3, rem = total & 7;
    return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

// TAIL: Cin is a multiple of 8 but not of 32 (the 80-channel pyramid layers): the last 32-channel K tile of every tap is part data,
// part zero page - compile-time variant, like MULT, so that the common kernels pay nothing for the per-lane channel bound.
// GATE (0 none | 1 ReLU / ELU from the producer's output | 2 GELU from its pre-activation value): desc.gate in the epilogue, compile-time as well - as a run-time branch it cost EVERY launch ~3 % (0.122 -> 0.126 ms on the
// 160 -> 160 layer, 0.2 ms per step), and only the data gradients of the ResNet blocks carry one (never the 160-wide tiles).
// LEAN: no activation, no pre-activation copy (every data gradient, every conv in front of a LayerNorm, every plain Linear) - the
// activation switch (erf / expm1 / exp paths, inlined 8 x TM x TN x 2 times) is most of the epilogue's code: without it the 160 -> 160
// layer runs 0.122 -> 0.108 ms (same-box, 3 runs each).  Gated launches are always lean (the dispatcher sends the rest elsewhere).
// (ACTK: -1 = any activation, decided at run time, + the pre-activation copy; 0 = none = "lean"; 1 = ReLU, a single v_max - the
// Bottleneck convolutions.)
#ifndef GWD_DBG_ZERO
#define GWD_DBG_ZERO 0      // development ablation builds only (tools/ab_build.sh): bit 0 / 1 = stage the A / B tile from the zero page
#endif
template <int BM, int BN, int WM, int WN, int STAGES, int BK = 32>
struct DmaTileCfg {
    static constexpr int NW = WM * WN;
    static constexpr int RPI = 1024 / (BK * 2);          // rows one 1 KiB DMA wave-instruction moves: 16 rows of 64 B, or 8 of 128 B
    static constexpr int B_INSTR = (BN + RPI - 1) / RPI;
    static constexpr int BROWS = B_INSTR * RPI;
    static constexpr int STAGE_BYTES = (BM + BROWS) * BK * 2;
    static constexpr int EPI_BYTES = NW * 32 * 36 * 4 + NW * (BM / WM / 32) * 64 * 4;      // transposition patches + the ConvLn row statistics
    static constexpr int SMEM = STAGES * STAGE_BYTES > EPI_BYTES ? STAGES * STAGE_BYTES : EPI_BYTES;
};

// One output tile [m0, m0 + BM) x [n0, n0 + BN).
// (Round 3, measured and dropped: a persistent launch of one workgroup per resident slot - whole tiles, then the rows of the
// mostly empty last round dealt out 44 per workgroup as a tile in which only two wave rows compute.  0.112 -> 0.122 ms on the
// 160 -> 160 layer, with or without the idle waves' tile traffic: a K step of such a tail costs ~1.1 us whatever it computes, 45
// of them are more than the 0.035 ms the half-empty round costs.  DESIGN.md section 4.)
// BK = 64 (round 3): a K tile of 64 channels makes every staged row piece a whole 128-byte line.  tools/ubench/piecerate.hip, L2-resident
// source, 512 workgroups: 64-byte pieces fill LDS at 17.6 TB/s chip-wide, 128-byte pieces at 33.4 TB/s - and the small tiles of the mid-
// size GEMMs (64 x 64: 32 FLOP per staged byte) are bound by exactly that rate (340 TF/s = 10.5 TB/s of fill on ResNet layer3).  Needs
// Cin % 64 == 0; half as many barriers per reduction as a bonus.  Swizzle: 16-byte chunk c of row r sits at position c ^ ((r >> 1) & 7)
// (the 16 lanes one ds_read_b128 cycle serves then hit 16 different 4-bank groups); BK = 32 keeps c ^ ((r >> 2) & 3).
// HALO (round 3): 3x3 / stride 1 / pad 1 layers on maps that tile exactly into 8 x 32 pixel patches.  The row tile is such a patch
// (wave w = patch row w, accumulator row = x), and the activations are staged ONCE per 32-channel block as the 10 x 34 pixel halo
// patch (21.25 KB) instead of once per tap: a K step then stages the 10 KB weight tile of its (channel block, tap) and a ninth of a
// halo patch - 12.4 KB and ~200 distinct 128-byte lines instead of 26.6 KB and 416 lines.  The lines are what the fill costs
// (tools/ubench/piecerate.hip: ~0.4 lines per clock and CU whatever the piece size; ablation with the A or B tile staged from the
// zero page, 160 -> 160 forward: 0.113 ms -> A 0.084, B 0.094, both 0.075; profiles/r03_ablations.txt 11).  The nine taps read
// their A fragments from the same patch at a pixel offset; the transposed gather (stride-1 data gradient) is the same walk with the
// taps mirrored.  Ring: 2 halo patches (22 KB each: channel block cb + 1 lands while the nine taps of cb compute) + 3 weight tiles.
struct HaloCfg {
#if GWD_DBG_ZERO & 128          // timing experiment only (wrong results): ONE halo patch buffer, five weight tiles
    static constexpr int A_INSTR = 22, A_BYTES = A_INSTR * 1024, B_BYTES = 160 * 64, B_STAGES = 5, A_BUFS = 1;
#else
    static constexpr int A_INSTR = 22, A_BYTES = A_INSTR * 1024, B_BYTES = 160 * 64, B_STAGES = 3, A_BUFS = 2;
#endif
    static constexpr int RING = A_BUFS * A_BYTES + B_STAGES * B_BYTES;
};

// PAR (round 3): data gradient of a 3x3 / stride 2 / pad 1 convolution, the transposed gather without its structural zeros.  Output pixel
// (oh, ow) only meets the taps with kh = oh + 1 and kw = ow + 1 (mod 2): 1, 2, 2 or 4 of the nine, by the parity class of the pixel - walked
// in image order three of every four staged A rows were the zero page (435 'TFLOP/s' of which 109 were arithmetic).  Here the rows are
// dealt out by class (tile -> class (py, px), then pixels (b, y', x') with oh = 2 y' + py, ow = 2 x' + px), the reduction of a tile
// visits only its class's taps (2.25 instead of 9 on average), and the epilogue stores each row at its own pixel.
template <int BM, int BN, int WM, int WN, int STAGES, int GM, bool MULT, bool TAIL, int GATE, int ACTK, bool LN = false, int KPB = 1, int BK = 32, bool HALO = false, bool PAR = false>
__device__ __forceinline__ void dma_tile(const gwd_conv_desc &d, char *smem, const int m0, const int n0) {
    typedef __bf16 T;
    static_assert(!HALO || (BM == 256 && BN == 160 && WM == 8 && WN == 1 && BK == 32 && KPB == 1 && !TAIL && GM <= 1), "halo tiles: 8 x 32 pixel patches, 160 columns");
    static_assert(!PAR || (GM == 2 && !HALO && !LN && !TAIL && !MULT && KPB == 1), "parity classes: general transposed gather only");
    constexpr int NW = WM * WN;                          // 4 or 8 waves
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int RPI = 1024 / (BK * 2), CPR = BK / 8;   // rows per 1 KiB DMA wave-instruction (16 | 8), 16-byte chunks per row (4 | 8)
    constexpr int A_IT = (BM / RPI) / NW;
    constexpr int B_INSTR = (BN + RPI - 1) / RPI;        // wave w issues B instructions w, w+NW, ...
    constexpr int B_IT = (B_INSTR + NW - 1) / NW;
    constexpr int B_FULL = B_INSTR % NW;                 // waves below this index issue B_IT, the others B_IT-1 (0: all B_IT)
    constexpr int STAGE_BYTES = DmaTileCfg<BM, BN, WM, WN, STAGES, BK>::STAGE_BYTES;
    static_assert((BM / RPI) % NW == 0 && TM >= 1 && TN >= 1 && (BK == 32 || (BK == 64 && !TAIL && KPB == 1)), "tile / wave layout");
    auto swz = [](int row) { return BK == 32 ? (row >> 2) & 3 : (row >> 1) & 7; };

    const int M = d.B * d.Ho * d.Wo, N = d.Cout, K = d.KH * d.KW * d.Cin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const T *x = (const T *)d.x;
    const T *wgt = (const T *)d.w;
    const char *zero = (const char *)d.zero_page;

    // this lane's rows of the A tile (one per DMA instruction it issues) and its swizzled source chunk.
    // p_h / p_w: the tap-independent part of the source coordinate (see src_pixel): ih = p_h + kh (plain and
    // up-sampled gathers) or p_h - kh (transposed); a_pix = first pixel of the row's image.
    int p_h[A_IT], p_w[A_IT], a_ck[A_IT];
    size_t a_pix[A_IT];
    bool a_ok[A_IT];
    const int gmode = GM == 0 ? (int)GWD_GATHER_CONV : (GM == 1 ? (int)GWD_GATHER_TRANSPOSED : (PAR ? (int)GWD_GATHER_TRANSPOSED : d.gather));
    const int gstride = GM == 1 ? 1 : d.stride;
    // PAR: the launch covers 4 classes x ceil(Mq / BM) row tiles, Mq = B * (Ho / 2) * (Wo / 2) pixels per class
    const int par_hh = d.Ho >> 1, par_wh = d.Wo >> 1, par_mq = d.B * par_hh * par_wh;
    // classes interleaved over the row tiles (tile t -> class 3 - t % 4, rows t / 4): the XCD bands of the launch order get the same mix of
    // four-tap and one-tap tiles (class-major order gave two XCDs all the four-tap tiles)
    const int par_cls = PAR ? 3 - ((m0 / BM) & 3) : 0;
    const int par_r0 = PAR ? ((m0 / BM) >> 2) * BM : 0, par_py = par_cls >> 1, par_px = par_cls & 1;
    auto par_pixel = [&](int r, int &b, int &oh, int &ow) {          // row r of the class -> output pixel
        b = r / (par_hh * par_wh);
        const int rem = r - b * (par_hh * par_wh);
        const int yq = rem / par_wh;
        oh = 2 * yq + par_py;
        ow = 2 * (rem - yq * par_wh) + par_px;
    };
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int row = RPI * (wave * A_IT + i) + lane / CPR;
        a_ck[i] = ((lane % CPR) ^ swz(row)) * 8;
        const int m = m0 + row;
        a_ok[i] = PAR ? (par_r0 + row < par_mq) : (m < M);
        const int mm = a_ok[i] ? m : 0;
        int b = mm / (d.Ho * d.Wo);
        const int rem = mm - b * (d.Ho * d.Wo);
        int oh = rem / d.Wo, ow = rem - oh * d.Wo;
        if constexpr (PAR) par_pixel(a_ok[i] ? par_r0 + row : 0, b, oh, ow);
        a_pix[i] = (size_t)b * d.Hi * d.Wi;
        if (gmode == GWD_GATHER_CONV) {
            p_h[i] = oh * d.stride - d.pad;
            p_w[i] = ow * d.stride - d.pad;
        } else if (gmode == GWD_GATHER_TRANSPOSED) {
            p_h[i] = oh + d.pad;
            p_w[i] = ow + d.pad;
        } else {
            p_h[i] = oh - d.pad;
            p_w[i] = ow - d.pad;
        }
    }
    const float up_sh = (float)d.Hi / (float)(d.Hv > 0 ? d.Hv : 1), up_sw = (float)d.Wi / (float)(d.Wv > 0 ? d.Wv : 1);
    const char *b_src[B_IT];
    bool b_ok[B_IT];
    int b_ck[B_IT];
    const int my_b_loads = (B_FULL == 0 || wave < B_FULL) ? B_IT : B_IT - 1;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int row = RPI * (wave + i * NW) + lane / CPR;
        const int ck = ((lane % CPR) ^ swz(row)) * 8;
        b_ck[i] = ck;
        const int n = n0 + row;
        b_ok[i] = row < BN && n < N;
        b_src[i] = (const char *)(wgt + (size_t)(b_ok[i] ? n : 0) * K + ck);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // PAR: taps kh0, kh0 + 2 (< KH) and kw0, kw0 + 2 only
    const int t_step = PAR ? 2 : 1, kh0 = PAR ? (par_py ^ 1) : 0, kw0 = PAR ? (par_px ^ 1) : 0;
    const int KT = TAIL ? d.KH * d.KW * ((d.Cin + BK - 1) / BK) : (PAR ? (par_py + 1) * (par_px + 1) * (d.Cin / BK) : K / BK);
    // reduction order: 64-channel group (one 128-byte line per pixel) outermost, then the filter taps, then the two
    // 32-channel halves of the group - a pixel's line is re-read for the next tap / half one or two tiles later, while
    // it is still in L2 (tap-outermost order has a reuse distance of Cin/32 tiles x every resident workgroup).
    int u_kh = kh0, u_kw = kw0, u_cb = 0, u_sub = 0, u_c0 = 0;  // workgroup-uniform state of the next tile to issue
    int u_grp = (BK == 64) ? 1 : (d.Cin > 32 ? 2 : 1);   // K tiles in the current 64-channel group (BK 32: two halves, the last may be a partial tile: TAIL)
    auto issue = [&](int stage) {
        char *sb = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int ih, iw;
            bool ok;
            if (gmode == GWD_GATHER_CONV) {                       // uniform branches, straight-line lanes
                ih = p_h[i] + u_kh;
                iw = p_w[i] + u_kw;
                ok = ((unsigned)ih < (unsigned)d.Hi) & ((unsigned)iw < (unsigned)d.Wi);
            } else if (gmode == GWD_GATHER_TRANSPOSED) {
                const int th = p_h[i] - u_kh, tw = p_w[i] - u_kw;
                if (gstride == 1) {
                    ih = th;
                    iw = tw;
                    ok = ((unsigned)ih < (unsigned)d.Hi) & ((unsigned)iw < (unsigned)d.Wi);
                } else {
                    const int thc = th < 0 ? 0 : th, twc = tw < 0 ? 0 : tw;
                    ih = thc / d.stride;
                    iw = twc / d.stride;
                    ok = (th >= 0) & (tw >= 0) & (ih * d.stride == th) & (iw * d.stride == tw) & (ih < d.Hi) & (iw < d.Wi);
                }
            } else {
                const int vh = p_h[i] + u_kh, vw = p_w[i] + u_kw;
                ok = ((unsigned)vh < (unsigned)d.Hv) & ((unsigned)vw < (unsigned)d.Wv);
                ih = min((int)floorf((float)vh * up_sh), d.Hi - 1);
                iw = min((int)floorf((float)vw * up_sw), d.Wi - 1);
            }
            ok = ok & a_ok[i];
            if constexpr (TAIL) ok = ok & (u_c0 + a_ck[i] < d.Cin);
            const size_t off = (a_pix[i] + (size_t)(ok ? ih : 0) * d.Wi + (ok ? iw : 0)) * d.Cin + u_c0 + a_ck[i];
            const char *src = (ok && !(GWD_DBG_ZERO & 1)) ? (const char *)(x + off) : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(sb + (wave * A_IT + i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            if (i < my_b_loads) {                                 // wave-uniform
                bool bok = b_ok[i];
                if constexpr (TAIL) bok = bok & (u_c0 + b_ck[i] < d.Cin);
                const char *src = (bok && !(GWD_DBG_ZERO & 2)) ? b_src[i] + (size_t)((u_kh * d.KW + u_kw) * d.Cin + u_c0) * 2 : zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(sb + BM * BK * 2 + (wave + i * NW) * 1024), 16, 0, 0);
            }
        }
        if (++u_sub == u_grp) {
            u_sub = 0;
            if ((u_kw += t_step) >= d.KW) {
                u_kw = kw0;
                if ((u_kh += t_step) >= d.KH) {
                    u_kh = kh0;
                    u_cb += 64;
                    u_grp = (BK == 64) ? 1 : (d.Cin - u_cb > 32 ? 2 : 1);
                }
            }
        }
        u_c0 = u_cb + u_sub * BK;
    };
    const int fr = lane & 31, fh = lane >> 5;
    auto compute = [&](int stage) {
        const T *As = (const T *)(smem + stage * STAGE_BYTES);
        const T *Bs = (const T *)(smem + stage * STAGE_BYTES + BM * BK * 2);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (BM / WM) + i * 32 + fr;
                af[i] = *(const bf16x8 *)(As + row * BK + (((ks * 2 + fh) ^ swz(row)) * 8));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * (BN / WN) + j * 32 + fr;
                bfr[j] = *(const bf16x8 *)(Bs + row * BK + (((ks * 2 + fh) ^ swz(row)) * 8));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mma(af[i], bfr[j], acc[i][j]);
        }
    };

    // row -> output pixel of accumulator block i, row r (0..31) of this wave
    int halo_m0 = 0;
    if constexpr (HALO) {
        const int txs = d.Wo >> 5, tys = d.Ho >> 3;
        const int mt = m0 / BM;
        const int tb = mt / (txs * tys), tr = mt - tb * (txs * tys);
        const int ty = tr / txs, tx = tr - ty * txs;
        halo_m0 = (tb * d.Ho + ty * 8 + wave) * d.Wo + tx * 32;
        constexpr int A_IT_H = 3;                         // DMA instructions wave + 8 i (< 22) of a halo patch
        const char *ha_src[A_IT_H];
        bool ha_ok[A_IT_H];
#pragma unroll
        for (int i = 0; i < A_IT_H; ++i) {
            const int p = 16 * (wave + 8 * i) + (lane >> 2);          // halo pixel (row-major in the 10 x 34 patch)
            const int hy = p / 34, hx = p - hy * 34;
            const int iy = ty * 8 + hy - 1, ix = tx * 32 + hx - 1;
            ha_ok[i] = (p < 340) & ((unsigned)iy < (unsigned)d.Hi) & ((unsigned)ix < (unsigned)d.Wi);
            const int ck = ((lane & 3) ^ ((p >> 2) & 3)) * 8;
            ha_src[i] = (const char *)(x + ((size_t)(tb * d.Hi + (ha_ok[i] ? iy : 0)) * d.Wi + (ha_ok[i] ? ix : 0)) * d.Cin + ck);
        }
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // scalar: the wait switch below is a chain of scalar branches
        const int nA = wave_u < HaloCfg::A_INSTR - 16 ? 3 : 2, nB = (B_FULL == 0 || wave_u < B_FULL) ? B_IT : B_IT - 1;
        const int NCB = d.Cin >> 5, S = NCB * 9;
        auto issue_a = [&](int cb) {
            char *ab = smem + (cb & (HaloCfg::A_BUFS - 1)) * HaloCfg::A_BYTES;
#pragma unroll
            for (int i = 0; i < A_IT_H; ++i) {
                if (wave + 8 * i < HaloCfg::A_INSTR) {            // wave-uniform
                    const char *src = (ha_ok[i] && !(GWD_DBG_ZERO & 1)) ? ha_src[i] + cb * 64 : zero;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(ab + (wave + 8 * i) * 1024), 16, 0, 0);
                }
            }
        };
        int i_cb = 0, i_tap = 0;                          // (channel block, tap) of the next weight tile to issue
        auto issue_b = [&](int stage) {
            char *sb = smem + HaloCfg::A_BUFS * HaloCfg::A_BYTES + stage * HaloCfg::B_BYTES;
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                if (i < my_b_loads) {
                    const char *src = (b_ok[i] && !(GWD_DBG_ZERO & 2)) ? b_src[i] + (size_t)(i_tap * d.Cin + i_cb * 32) * 2 : zero;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(sb + (wave + i * NW) * 1024), 16, 0, 0);
                }
            }
            if (++i_tap == 9) {
                i_tap = 0;
                ++i_cb;
            }
        };
        const int fr_h = lane & 31, fh_h = lane >> 5;
        const int hp0 = wave * 34 + fr_h;
        // Fragment reads and MFMAs of a step in a fixed issue order, the reads as inline assembly with counted lgkmcnt waits: six reads up
        // front, then every MFMA of the first 16-channel half is followed by one read of the second half, so four or five reads stay in
        // flight behind the matrix pipe.  (The compiler's own order was two reads, s_waitcnt lgkmcnt(0), one or two MFMAs - a full LDS
        // round trip exposed five times per half - and with the reads it can see it drains the counter to zero in this loop whatever
        // the order.)  7 fragment slots + 80 accumulator registers stay inside the 128 of four waves per SIMD.
        typedef __attribute__((address_space(3))) const char *lds_cptr;
        auto compute_h = [&](int cb, int toff, int stage) {
            const int hp = hp0 + toff;
            const uint32_t ap = (uint32_t)(uintptr_t)(lds_cptr)(smem + (cb & (HaloCfg::A_BUFS - 1)) * HaloCfg::A_BYTES + hp * 64);
            const int asw = (hp >> 2) & 3;
            const uint32_t a0 = ap + (((fh_h) ^ asw) << 4), a1 = ap + (((2 + fh_h) ^ asw) << 4);
            const uint32_t bp = (uint32_t)(uintptr_t)(lds_cptr)(smem + HaloCfg::A_BUFS * HaloCfg::A_BYTES + stage * HaloCfg::B_BYTES) + fr_h * 64;
            const uint32_t b0 = bp + (((fh_h) ^ swz(fr_h)) << 4), b1 = bp + (((2 + fh_h) ^ swz(fr_h)) << 4);     // swz(j * 32 + fr) == swz(fr)
            bf16x8 fa0, fa1, f0[TN], f1[TN];
            static_assert(TN == 5, "schedule written for five column blocks");
#define HALO_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define HALO_WAIT(n, x, y) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(n))
            HALO_RD(fa0, a0, 0);
            HALO_RD(f0[0], b0, 0);
            HALO_RD(f0[1], b0, 2048);
            HALO_RD(f0[2], b0, 4096);
            HALO_RD(f0[3], b0, 6144);
            HALO_RD(f0[4], b0, 8192);
            HALO_WAIT(4, fa0, f0[0]);
            acc[0][0] = mma(fa0, f0[0], acc[0][0]);
            HALO_RD(fa1, a1, 0);
            HALO_WAIT(4, fa0, f0[1]);
            acc[0][1] = mma(fa0, f0[1], acc[0][1]);
            HALO_RD(f1[0], b1, 0);
            HALO_WAIT(4, fa0, f0[2]);
            acc[0][2] = mma(fa0, f0[2], acc[0][2]);
            HALO_RD(f1[1], b1, 2048);
            HALO_WAIT(4, fa0, f0[3]);
            acc[0][3] = mma(fa0, f0[3], acc[0][3]);
            HALO_RD(f1[2], b1, 4096);
            HALO_WAIT(4, fa0, f0[4]);
            acc[0][4] = mma(fa0, f0[4], acc[0][4]);
            HALO_RD(f1[3], b1, 6144);
            HALO_WAIT(3, fa1, f1[0]);
            acc[0][0] = mma(fa1, f1[0], acc[0][0]);
            HALO_RD(f1[4], b1, 8192);
            HALO_WAIT(3, fa1, f1[1]);
            acc[0][1] = mma(fa1, f1[1], acc[0][1]);
            HALO_WAIT(2, fa1, f1[2]);
            acc[0][2] = mma(fa1, f1[2], acc[0][2]);
            HALO_WAIT(1, fa1, f1[3]);
            acc[0][3] = mma(fa1, f1[3], acc[0][3]);
            HALO_WAIT(0, fa1, f1[4]);
            acc[0][4] = mma(fa1, f1[4], acc[0][4]);
#undef HALO_RD
#undef HALO_WAIT
        };
        constexpr int NB = HaloCfg::B_STAGES;             // weight tiles st + 1 .. st + NB - 2 stay in flight behind the one step st waits for
        issue_a(0);
#pragma unroll
        for (int t = 0; t < NB - 1; ++t) issue_b(t);
        int cb = 0, kh = 0, kw = 0;
        int a_age = NB;                                   // steps since a halo patch was issued (it is in flight behind tile st while a_age <= NB - 2)
        for (int st = 0; st < S; ++st) {
            // the weight tile of this step has landed once only the newer tiles (and a halo patch issued after it) are in flight
            const int newer = min(NB - 2, S - 1 - st);
            const int allow = newer * nB + ((a_age <= NB - 2 && a_age <= newer) ? nA : 0);         // wave-uniform
            switch (allow) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            }
#if !(GWD_DBG_ZERO & 64)
            __builtin_amdgcn_s_barrier();
#endif
            const int tap = kh * 3 + kw;
            ++a_age;
#if !(GWD_DBG_ZERO & 32)
            if (tap == 0 && cb + 1 < NCB) {
                issue_a(cb + 1);
                a_age = 1;
            }
            if (st + NB - 1 < S) issue_b((st + NB - 1) % NB);
#endif
            compute_h(cb, GM == 1 ? (2 - kh) * 34 + (2 - kw) : kh * 34 + kw, st % NB);
            if (++kw == 3) {
                kw = 0;
                if (++kh == 3) {
                    kh = 0;
                    ++cb;
                }
            }
        }
    } else if constexpr (KPB == 1) {
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
            if (t < KT) issue(t);
        for (int kt = 0; kt < KT; ++kt) {
            // tile kt has landed once at most (STAGES-2) newer tiles of this wave are still in flight
            if (kt + STAGES - 2 < KT) {
                if (my_b_loads == B_IT)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A_IT + B_IT) * (STAGES - 2)) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A_IT + B_IT - 1) * (STAGES - 2)) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                 // everybody's part of tile kt landed; compute(kt-1) is finished
            if (kt + STAGES - 1 < KT) issue((kt + STAGES - 1) % STAGES);
            compute(kt % STAGES);
        }
    } else {
        // KPB K tiles per barrier (launch precondition: KT % KPB == 0): the small tiles of the mid-size GEMMs (ResNet layer2-4, the
        // transformer linears) spend a K step on one barrier, one LDS round trip and two dependent MFMAs per wave - 0.34 us whatever
        // the ring depth.  Same stages, same DMA, same fragment reads; a GROUP of KPB consecutive stages is waited for, fenced and
        // consumed together, so the fixed cost of a step is paid per 32 * KPB channels of the reduction.
        constexpr int SS = STAGES / KPB;                  // groups in the ring
        static_assert(STAGES % KPB == 0 && SS >= 2, "ring = whole groups");
        const int KG = KT / KPB;
#pragma unroll
        for (int t = 0; t < (SS - 1) * KPB; ++t)
            if (t < KT) issue(t);
        for (int kg = 0; kg < KG; ++kg) {
            if (kg + SS - 2 < KG) {
                if (my_b_loads == B_IT)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A_IT + B_IT) * KPB * (SS - 2)) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A_IT + B_IT - 1) * KPB * (SS - 2)) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (kg + SS - 1 < KG) {
#pragma unroll
                for (int u = 0; u < KPB; ++u) issue(((kg + SS - 1) * KPB + u) % STAGES);
            }
#pragma unroll
            for (int u = 0; u < KPB; ++u) compute((kg * KPB + u) % STAGES);
        }
    }
    __syncthreads();
    const int wbase = HALO ? halo_m0 : m0 + wm * (BM / WM);      // output pixel of this wave's accumulator row 0

    if constexpr (LN) {
        // ---- ConvLn epilogue (points_sample.py:12-25): LayerNorm over the row's C = d.ln_C real channels (eps 1e-5, biased variance)
        // straight from the fp32 accumulators, then * gamma + beta (d.scale / d.shift, C entries), [GELU: ACTK 2], [+ residual].
        // WN == 1 and one column tile: a wave holds complete rows - row r of accumulator tile i lives in the 32 lanes of one wave
        // half (column = lane & 31) across the TN tiles, so mean and variance are TN adds and one 32-lane segment sum each.  Two
        // passes (mean, then centred squares) as nn.LayerNorm computes them.  The statistics go to d.ln_mean / d.ln_rstd (the
        // LayerNorm backward kernel reads them) and, through a wave-private LDS table, to the lanes that own the row after the
        // transposition; d.z gets the conv output itself (what the backward normalises again).  Columns >= C are zero padding
        // (ops._PadConvFn): they do not count and come out as zeros.
        static_assert(WN == 1 && !MULT && !GATE, "ConvLn epilogue: complete rows per wave");
        T *y = (T *)d.y;
        T *z = (T *)d.z;
        const T *res = (const T *)d.residual;
        const int C = d.ln_C;
        const float inv_c = 1.0f / (float)C;
        float *stage = (float *)smem + wave * (32 * 36);
        float *stats = (float *)smem + NW * (32 * 36) + wave * (TM * 64);
        const int vr = lane >> 2, vc = (lane & 3) * 8;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float mu[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) sum += acc[i][j][r];
                mu[r] = segment_sum<32>(sum) * inv_c;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sq = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float dv = acc[i][j][r] - mu[r];
                    sq += (j * 32 + fr < C) ? dv * dv : 0.f;
                }
                const float rs = rsqrtf(segment_sum<32>(sq) * inv_c + 1e-5f);
                if (fr == 0) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
                    stats[(i * 32 + row) * 2] = mu[r];
                    stats[(i * 32 + row) * 2 + 1] = rs;
                    const int mr = wbase + i * 32 + row;
                    if (mr < M) {
                        d.ln_mean[mr] = mu[r];
                        d.ln_rstd[mr] = rs;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * fh) * 36 + fr] = acc[i][j][r];
                __builtin_amdgcn_wave_barrier();
                const int nb = n0 + j * 32 + vc;
                if (nb < N) {
                    float sc[8], sh[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        sc[e] = nb + e < C ? d.scale[nb + e] : 0.f;
                        sh[e] = nb + e < C ? d.shift[nb + e] : 0.f;
                    }
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int row = vr + 16 * half;
                        const int mr = wbase + i * 32 + row;
                        if (mr >= M) continue;
                        const f32x4 lo = *(const f32x4 *)(stage + row * 36 + vc);
                        const f32x4 hi = *(const f32x4 *)(stage + row * 36 + vc + 4);
                        const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        const float m_ = stats[(i * 32 + row) * 2], rs = stats[(i * 32 + row) * 2 + 1];
                        const size_t o = (size_t)mr * N + nb;
                        bf16x8 out;
                        if (z) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) out[e] = (__bf16)v[e];
                            *(bf16x8 *)(z + o) = out;
                        }
                        float t[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            t[e] = (v[e] - m_) * rs * sc[e] + sh[e];
                            if (ACTK == 2) t[e] = gelu_fast(t[e]);
                        }
                        if (res) {
                            const bf16x8 rv = *(const bf16x8 *)(res + o);
#pragma unroll
                            for (int e = 0; e < 8; ++e) t[e] += (float)rv[e];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) out[e] = (__bf16)t[e];
                        *(bf16x8 *)(y + o) = out;
                    }
                }
            }
        }
        return;
    }
    // ---- epilogue (identical to igemm_fwd_kernel's vector path; N % 8 == 0 is a launch precondition)
    T *y = (T *)d.y;
    T *z = (ACTK == 0 || ACTK == 1) ? nullptr : (T *)d.z;       // ACTK 2 = GELU keeps the pre-activation copy its backward needs
    const T *res = (const T *)d.residual;
    const T *mul = MULT ? (const T *)d.mult : nullptr;       // compile-time: the multiplier path costs the plain kernels registers
    const T *gate = (GATE && !MULT) ? (const T *)d.gate : nullptr;      // never together with a multiplier (check_desc)
    float *stage = (float *)smem + wave * (32 * 36);
    const int vr = lane >> 2, vc = (lane & 3) * 8;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * fh) * 36 + fr] = acc[i][j][r];
            __builtin_amdgcn_wave_barrier();
            const int nb = n0 + wn * (BN / WN) + j * 32 + vc;
            if (nb < N) {
                float sc[8], sh[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sc[e] = d.scale ? d.scale[nb + e] : 1.0f;
                    sh[e] = d.shift ? d.shift[nb + e] : 0.0f;
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int row = vr + 16 * half;
                    int m = wbase + i * 32 + row;
                    if constexpr (PAR) {
                        const int r = par_r0 + wm * (BM / WM) + i * 32 + row;
                        if (r >= par_mq) continue;
                        int pb, poh, pow_;
                        par_pixel(r, pb, poh, pow_);
                        m = (pb * d.Ho + poh) * d.Wo + pow_;
                    } else if (m >= M) continue;
                    const f32x4 lo = *(const f32x4 *)(stage + row * 36 + vc);
                    const f32x4 hi = *(const f32x4 *)(stage + row * 36 + vc + 4);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const size_t o = (size_t)m * N + nb;
                    if (res && !mul) {
                        const bf16x8 rv = *(const bf16x8 *)(res + o);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e] + (float)rv[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
                    }
                    bf16x8 out;
                    if (z) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) out[e] = (__bf16)v[e];
                        *(bf16x8 *)(z + o) = out;
                    }
                    if (mul) {                           // y = act_scale * act(v) * mult + residual (dropout, then the skip)
                        const bf16x8 mv = *(const bf16x8 *)(mul + o);
                        bf16x8 rv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) rv[e] = (__bf16)0.0f;
                        if (res) rv = *(const bf16x8 *)(res + o);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            out[e] = (__bf16)((ACTK == 0 ? v[e] : (ACTK == 1 ? (v[e] > 0.f ? v[e] : 0.f) : apply_act(v[e], d.act))) * d.act_scale * (float)mv[e] + (float)rv[e]);
                    } else {
                        float gm[8];                         // desc.gate: backward of the producer's activation, from its output
#pragma unroll
                        for (int e = 0; e < 8; ++e) gm[e] = d.act_scale;
                        if (gate) {
                            const bf16x8 gv = *(const bf16x8 *)(gate + o);
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                gm[e] = GATE == 2 ? d.act_scale * gelu_grad_fast((float)gv[e])       // the producer's GELU, from its PRE-activation value
                                                  : gate_grad(d.act_scale, (float)gv[e], d.gate_act);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            out[e] = (__bf16)((ACTK == 0 ? v[e] : (ACTK == 1 ? (v[e] > 0.f ? v[e] : 0.f) : (ACTK == 2 ? gelu_fast(v[e]) : apply_act(v[e], d.act)))) * gm[e]);
                    }
                    *(bf16x8 *)(y + o) = out;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int STAGES, int GM, bool MULT = false, bool TAIL = false, int GATE = 0, int ACTK = -1, bool LN = false, int KPB = 1, int BK = 32, bool HALO = false, bool PAR = false>
__global__ __launch_bounds__(WM *WN * 64) __attribute__((amdgpu_waves_per_eu(HALO ? 4 : 1, HALO ? 4 : 8))) void igemm_dma_kernel(const gwd_conv_desc d, const int tile_base, const int tile_count) {
    constexpr int SM_ = DmaTileCfg<BM, BN, WM, WN, STAGES, BK>::SMEM;
    __shared__ __attribute__((aligned(1024))) char smem[HALO ? (HaloCfg::RING > DmaTileCfg<BM, BN, WM, WN, STAGES, BK>::EPI_BYTES ? HaloCfg::RING : DmaTileCfg<BM, BN, WM, WN, STAGES, BK>::EPI_BYTES) : SM_];
    const int n_tiles = (d.Cout + BN - 1) / BN;
    // this launch covers tiles tile_base .. tile_base + tile_count - 1 of the logical order.  (Cutting a big problem into a body of
    // whole rounds of 256-row tiles and a tail of 128-row tiles was measured in round 2: 0.125 -> 0.134 ms on the 160 -> 160
    // layer, the second launch and the lone waves of the tail cost more than the half-empty round they replace.)
    const int tile = tile_base + xcd_band(blockIdx.x, tile_count);             // column tiles of a row tile are adjacent
    dma_tile<BM, BN, WM, WN, STAGES, GM, MULT, TAIL, GATE, ACTK, LN, KPB, BK, HALO, PAR>(d, smem, (tile / n_tiles) * BM, (tile % n_tiles) * BN);
}

// ----------------------------------------------------------------------------------------------
// weight gradient
// ----------------------------------------------------------------------------------------------
template <typename T, int BNW, int BKW>
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(const gwd_conv_desc d, float *__restrict__ dw, int m_per_block) {
    constexpr int VEC = Cfg<T>::VEC, RM = Cfg<T>::BK;  // reduction rows per step
    // row strides chosen so the transposed bf16 reads (ds_read_b64_tr_b16) of one 32-lane half hit
    // 64 distinct banks: stride(words) mod 64 == 16.  fp32 reads are single words: any stride works.
    constexpr int LDY = (sizeof(T) == 2) ? (BNW == 64 ? 96 : 160) : BNW + 4;
    constexpr int LDX = (sizeof(T) == 2) ? (BKW == 64 ? 96 : 160) : BKW + 4;
    constexpr int TN = BNW / 64, TK = BKW / 64;  // 32x32 tiles per wave (2x2 waves)
    constexpr int YV = BNW / VEC, XV = BKW / VEC;  // vectors per row
    constexpr int Y_IT = RM * YV / 256, X_IT = RM * XV / 256;
    static_assert(Y_IT >= 1 && X_IT >= 1, "tile too small");

    __shared__ __attribute__((aligned(16))) T smem[2 * RM * (LDY + LDX)];
    T *Ys = smem;
    T *Xs = smem + 2 * RM * LDY;

    const int M = d.B * d.Ho * d.Wo, N = d.Cout, K = d.KH * d.KW * d.Cin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int kb0 = blockIdx.x * BKW, n0 = blockIdx.y * BNW;
    const int m_begin = blockIdx.z * m_per_block;
    const int m_end = min(M, m_begin + m_per_block);
    const bool fastA = (d.Cin % VEC) == 0, fastY = (N % VEC) == 0;
    const bool halfA = !fastA && (d.Cin % (VEC / 2)) == 0, halfY = !fastY && (N % (VEC / 2)) == 0;
    const T *gy = (const T *)d.y;

    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 ry[Y_IT], rx[X_IT];
    // Per-thread constants of the im2col gather: the k range of a workgroup is fixed, so the filter tap of each of
    // this thread's vectors is decoded once; the output pixel of its row advances by RM per step (no divisions).
    int x_kh[X_IT], x_kw[X_IT], x_c[X_IT], x_b[X_IT], x_oh[X_IT], x_ow[X_IT];
    bool x_kok[X_IT];
    const int HoWo = d.Ho * d.Wo;
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
        const int idx = tid + i * 256, row = idx / XV, v = idx % XV;
        const int k0 = kb0 + v * VEC;
        x_kok[i] = k0 < K;
        const int kk = x_kok[i] ? k0 : 0;
        const int tap = kk / d.Cin;
        x_c[i] = kk - tap * d.Cin;
        x_kh[i] = tap / d.KW;
        x_kw[i] = tap - x_kh[i] * d.KW;
        const int m = m_begin + row;
        x_b[i] = m / HoWo;
        const int rem = m - x_b[i] * HoWo;
        x_oh[i] = rem / d.Wo;
        x_ow[i] = rem - x_oh[i] * d.Wo;
    }
    auto load_tiles = [&](int mbase) {
#pragma unroll
        for (int i = 0; i < Y_IT; ++i) {
            const int idx = tid + i * 256, row = idx / YV, v = idx % YV;
            const int m = mbase + row;
            ry[i] = (m < m_end) ? row_vec<T>(gy, m, M, n0 + v * VEC, N, fastY, halfY) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < X_IT; ++i) {
            const int idx = tid + i * 256, row = idx / XV, v = idx % XV;
            const bool ok = (mbase + row) < m_end;
            if (fastA) {
                uint4 r = make_uint4(0u, 0u, 0u, 0u);
                int ih, iw;
                if (ok && x_kok[i] && src_pixel(d, x_oh[i], x_ow[i], x_kh[i], x_kw[i], ih, iw))
                    r = *(const uint4 *)((const T *)d.x + ((size_t)(x_b[i] * d.Hi + ih) * d.Wi + iw) * d.Cin + x_c[i]);
                rx[i] = r;
            } else {
                rx[i] = gather_vec<T>(d, ok, x_b[i], x_oh[i], x_ow[i], kb0 + v * VEC, K, false, halfA);
            }
            // advance this row's output pixel by RM for the next step
            if (HoWo == 1) {
                x_b[i] += RM;
            } else {
                x_ow[i] += RM;
                while (x_ow[i] >= d.Wo) {
                    x_ow[i] -= d.Wo;
                    if (++x_oh[i] == d.Ho) {
                        x_oh[i] = 0;
                        ++x_b[i];
                    }
                }
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < Y_IT; ++i) {
            const int idx = tid + i * 256, row = idx / YV, v = idx % YV;
            *(uint4 *)(Ys + (size_t)buf * RM * LDY + row * LDY + v * VEC) = ry[i];
        }
#pragma unroll
        for (int i = 0; i < X_IT; ++i) {
            const int idx = tid + i * 256, row = idx / XV, v = idx % XV;
            *(uint4 *)(Xs + (size_t)buf * RM * LDX + row * LDX + v * VEC) = rx[i];
        }
    };

    const int steps = (m_end - m_begin + RM - 1) / RM;
    if (steps > 0) {
        load_tiles(m_begin);
        store_tiles(0);
    }
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    // transposed-read geometry: 16-lane group g reads the 4x16 block rows 8*(g>>1)+4*rd.., cols 16*(g&1)..
    const int g = lane >> 4, t = lane & 15;
    for (int s = 0; s < steps; ++s) {
        const int cur = s & 1;
        if (s + 1 < steps) load_tiles(m_begin + (s + 1) * RM);
        const T *Yb = Ys + (size_t)cur * RM * LDY + wn * (BNW / 2);
        const T *Xb = Xs + (size_t)cur * RM * LDX + wk * (BKW / 2);
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < RM / 16; ++ks) {
                bf16x8 af[TN], bfr[TK];
                const int row = ks * 16 + 8 * (g >> 1) + (t >> 2);
                const int col = 16 * (g & 1) + 4 * (t & 3);
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    const T *p = Yb + row * LDY + i * 32 + col;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + 4 * LDY));
                    union { s16x4 h[2]; bf16x8 v; } u;
                    u.h[0] = lo;
                    u.h[1] = hi;
                    af[i] = u.v;
                }
#pragma unroll
                for (int j = 0; j < TK; ++j) {
                    const T *p = Xb + row * LDX + j * 32 + col;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + 4 * LDX));
                    union { s16x4 h[2]; bf16x8 v; } u;
                    u.h[0] = lo;
                    u.h[1] = hi;
                    bfr[j] = u.v;
                }
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TK; ++j) acc[i][j] = mma(af[i], bfr[j], acc[i][j]);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < RM / 2; ++ks) {
                float af[TN], bfr[TK];
#pragma unroll
                for (int i = 0; i < TN; ++i) af[i] = Yb[(ks * 2 + fh) * LDY + i * 32 + fr];
#pragma unroll
                for (int j = 0; j < TK; ++j) bfr[j] = Xb[(ks * 2 + fh) * LDX + j * 32 + fr];
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TK; ++j) acc[i][j] = mma(af[i], bfr[j], acc[i][j]);
            }
        }
        if (s + 1 < steps) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // the per-row multiplier first, in a pass of its own: a load between two atomics makes the compiler wait for vmcnt(0) - i.e. for
    // every atomic issued so far - in front of each multiply, and the flush runs at one atomic per memory round trip
    if (d.scale) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * (BNW / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const float sc = n < N ? d.scale[n] : 0.f;
#pragma unroll
                for (int j = 0; j < TK; ++j) acc[i][j][r] *= sc;
            }
    }
#pragma unroll
    for (int j = 0; j < TK; ++j) {
        const int k = kb0 + wk * (BKW / 2) + j * 32 + fr;
        if (k >= K) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * (BNW / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (n < N) unsafeAtomicAdd(dw + (size_t)n * K + k, acc[i][j][r]);
            }
    }
}

// ----------------------------------------------------------------------------------------------
// weight gradient, LDS-DMA pipeline (bf16, Cin % 8 == 0, Cout % 8 == 0)
// ----------------------------------------------------------------------------------------------
// Same math as igemm_wgrad_kernel; the dY tile [32 pixels][BNW channels] and the im2col tile [32 pixels][BKW] go
// HBM -> LDS by global_load_lds_dwordx4 through a ring of STAGES buffers.  An LDS image is a linear array of 16-byte
// chunks (row-major, unpadded: a DMA wave instruction writes 64 consecutive chunks), every lane derives the
// (row, chunk) it fetches from its linear chunk index.  The transposed reads (ds_read_b64_tr_b16) touch 4 consecutive
// rows per 16-lane group: rows of 320 B (BNW = 160) are 16 words apart mod 64 and need nothing, rows of 256/512 B
// (128 B) alias and get an XOR of the chunk index with row bits, applied to the SOURCE address and to the read.
// Wave grid WN x WK over the [BNW x BKW] output tile; BNW = 160 serves the 160/320-channel pyramids exactly.
template <int ROWB> __device__ __forceinline__ int wg_swz(int row) {
    if constexpr ((ROWB / 4) % 64 == 0) return (row & 3) << 2;
    else if constexpr ((ROWB / 4) % 64 == 32) return ((row >> 1) & 1) << 2;
    else return 0;
}

// FAST: 0 = any gather (per-step coordinate walk and bounds tests with divergent lanes); 1 = stride-1 'same'
// convolution (Ho == Hi, Wo == Wi >= 11): the source pixel of output pixel m under tap (kh, kw) is m + const, so a
// lane's address advances by a constant per step and only the in-image test needs (oh, ow), kept by branch-free
// conditional subtractions; 2 = 1x1 / stride 1 / no padding (every Linear): purely linear, always in range.
// The body takes its workgroup index and count as arguments: igemm_wgrad_dma_kernel passes the hardware ids,
// igemm_wgrad_group_kernel (several layers' weight gradients in one launch) the ids inside the layer's own block range.
template <int BNW, int BKW, int WN, int WK, int STAGES, int FAST>
__device__ __forceinline__ void wgrad_dma_body(const gwd_conv_desc &d, float *__restrict__ dw, int m_per_block, int wg_id, int wg_count) {
    typedef __bf16 T;
    constexpr int RM = 32;
    constexpr int TN = BNW / WN / 32, TK = BKW / WK / 32;
    constexpr int YB = BNW * 2, XB = BKW * 2;                  // row bytes
    constexpr int YC = BNW / 8, XC = BKW / 8;                  // 16-byte chunks per row
    constexpr int YI = RM * YC / 64, XI = RM * XC / 64;        // DMA wave-instructions per tile
    constexpr int TI = YI + XI;
    constexpr int IT = (TI + 3) / 4, FULL = TI % 4;            // wave w issues instructions w, w+4, ...; waves < FULL issue IT
    constexpr int STAGE_BYTES = RM * (YB + XB);
    static_assert(WN * WK == 4 && (RM * YC) % 64 == 0 && (RM * XC) % 64 == 0, "tile");
    __shared__ __attribute__((aligned(1024))) char smem[STAGES * STAGE_BYTES];

    const int M = d.B * d.Ho * d.Wo, N = d.Cout, K = d.KH * d.KW * d.Cin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WK, wk = wave % WK;
    // one XCD owns a contiguous run of (split, tile) pairs: all tiles of a reduction split re-read the same pixels
    const int k_tiles = (K + BKW - 1) / BKW, tiles = k_tiles * ((N + BNW - 1) / BNW);
    const int band_id = xcd_band(wg_id, wg_count);
    const int tile = band_id % tiles;
    const int kb0 = (tile % k_tiles) * BKW, n0 = (tile / k_tiles) * BNW;
    const int m_begin = (band_id / tiles) * m_per_block;
    const int m_end = min(M, m_begin + m_per_block);
    const T *gy = (const T *)d.y;
    const T *x = (const T *)d.x;
    const char *zero = (const char *)d.zero_page;
    const int my_loads = (FULL == 0 || wave < FULL) ? IT : IT - 1;
    const int HoWo = d.Ho * d.Wo;

    // ---- per-lane DMA assignments: instruction j = wave + 4*i of the combined (Y then X) list
    bool is_y[IT], col_ok[IT];
    int row[IT], x_kh[IT], x_kw[IT], x_c[IT], x_b[IT], x_oh[IT], x_ow[IT];
    const T *y_src[IT];
    const T *x_lin[IT];          // FAST: address of the lane's 16 bytes for output pixel 0 (advanced by pixel index)
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int j = wave + 4 * i;
        is_y[i] = j < YI;
        y_src[i] = gy;
        x_kh[i] = x_kw[i] = x_c[i] = x_b[i] = x_oh[i] = x_ow[i] = 0;
        if (is_y[i]) {
            const int q = j * 64 + lane;
            row[i] = q / YC;
            const int c = (q - row[i] * YC) ^ wg_swz<YB>(row[i]);
            col_ok[i] = n0 + c * 8 < N;
            y_src[i] = gy + (size_t)(m_begin + row[i]) * N + n0 + c * 8;
        } else {
            const int q = (j - YI) * 64 + lane;
            row[i] = q / XC;
            const int c = (q - row[i] * XC) ^ wg_swz<XB>(row[i]);
            const int k0 = kb0 + c * 8;
            col_ok[i] = (j < TI) && k0 < K;
            const int kk = col_ok[i] ? k0 : 0;
            const int tap = kk / d.Cin;
            x_c[i] = kk - tap * d.Cin;
            x_kh[i] = tap / d.KW;
            x_kw[i] = tap - x_kh[i] * d.KW;
            const int m = m_begin + row[i];
            x_b[i] = m / HoWo;
            const int rem = m - x_b[i] * HoWo;
            x_oh[i] = rem / d.Wo;
            x_ow[i] = rem - x_oh[i] * d.Wo;
        }
        x_lin[i] = x + ((ptrdiff_t)(x_kh[i] - d.pad) * d.Wi + (x_kw[i] - d.pad)) * d.Cin + x_c[i];
    }

    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int steps = (m_end - m_begin + RM - 1) / RM;
    int issued = 0;
    auto issue = [&](int stage) {
        char *sb = smem + stage * STAGE_BYTES;
        const int mbase = m_begin + issued * RM;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            if (i >= my_loads) continue;                          // wave-uniform
            const int j = wave + 4 * i;
            const bool rok = col_ok[i] && (mbase + row[i]) < m_end;
            const char *src = zero;
            if (is_y[i]) {                                        // wave-uniform (an instruction is all-Y or all-X)
                if (rok) src = (const char *)(y_src[i] + (size_t)issued * RM * N);
            } else if (FAST == 2) {
                if (rok) src = (const char *)(x_lin[i] + (size_t)(mbase + row[i]) * d.Cin);
            } else if (FAST == 1) {
                const int ih = x_oh[i] + x_kh[i] - d.pad, iw = x_ow[i] + x_kw[i] - d.pad;
                const bool ok = rok & ((unsigned)ih < (unsigned)d.Hi) & ((unsigned)iw < (unsigned)d.Wi);
                if (ok) src = (const char *)(x_lin[i] + (ptrdiff_t)(mbase + row[i]) * d.Cin);
                x_ow[i] += RM;                                    // (oh, ow) of the lane's next pixel, image index not needed
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const bool wrap = x_ow[i] >= d.Wo;
                    x_ow[i] -= wrap ? d.Wo : 0;
                    x_oh[i] += wrap ? 1 : 0;
                    x_oh[i] -= x_oh[i] >= d.Ho ? d.Ho : 0;
                }
            } else {
                int ih, iw;
                if (rok && src_pixel(d, x_oh[i], x_ow[i], x_kh[i], x_kw[i], ih, iw))
                    src = (const char *)(x + ((size_t)(x_b[i] * d.Hi + ih) * d.Wi + iw) * d.Cin + x_c[i]);
                if (HoWo == 1) {
                    x_b[i] += RM;
                } else {
                    x_ow[i] += RM;
                    while (x_ow[i] >= d.Wo) {
                        x_ow[i] -= d.Wo;
                        if (++x_oh[i] == d.Ho) {
                            x_oh[i] = 0;
                            ++x_b[i];
                        }
                    }
                }
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(sb + j * 1024), 16, 0, 0);
        }
        ++issued;
    };

    const int g = lane >> 4, t = lane & 15;
    auto compute = [&](int stage) {
        const char *Yb = smem + stage * STAGE_BYTES;
        const char *Xb = Yb + RM * YB;                            // == Yb + YI * 1024
#pragma unroll
        for (int ks = 0; ks < RM / 16; ++ks) {
            bf16x8 af[TN], bfr[TK];
            const int r0 = ks * 16 + 8 * (g >> 1) + (t >> 2);           // rows r0 (lo) and r0 + 4 (hi)
            const int cw = 16 * (g & 1) + 4 * (t & 3);                    // column inside the 32-wide tile
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int col = wn * (BNW / WN) + i * 32 + cw;
                const char *plo = Yb + r0 * YB + (((col >> 3) ^ wg_swz<YB>(r0)) << 4) + ((col & 7) << 1);
                const char *phi = Yb + (r0 + 4) * YB + (((col >> 3) ^ wg_swz<YB>(r0 + 4)) << 4) + ((col & 7) << 1);
                union { s16x4 h[2]; bf16x8 v; } u;
                u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)plo);
                u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)phi);
                af[i] = u.v;
            }
#pragma unroll
            for (int j = 0; j < TK; ++j) {
                const int col = wk * (BKW / WK) + j * 32 + cw;
                const char *plo = Xb + r0 * XB + (((col >> 3) ^ wg_swz<XB>(r0)) << 4) + ((col & 7) << 1);
                const char *phi = Xb + (r0 + 4) * XB + (((col >> 3) ^ wg_swz<XB>(r0 + 4)) << 4) + ((col & 7) << 1);
                union { s16x4 h[2]; bf16x8 v; } u;
                u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)plo);
                u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)phi);
                bfr[j] = u.v;
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TK; ++j) acc[i][j] = mma(af[i], bfr[j], acc[i][j]);
        }
    };

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < steps) issue(s);
    for (int s = 0; s < steps; ++s) {
        if (s + STAGES - 2 < steps) {
            if (my_loads == IT)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IT * (STAGES - 2)) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((IT - 1) * (STAGES - 2)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (s + STAGES - 1 < steps) issue((s + STAGES - 1) % STAGES);
        compute(s % STAGES);
    }

    const int fr = lane & 31, fh = lane >> 5;
    // the per-row multiplier (folded FrozenBN) in a pass of its own, BEFORE the first atomic: a load between two atomics puts an
    // s_waitcnt vmcnt(0) - a wait for every atomic issued so far - in front of each multiply, and the flush runs at one atomic per
    // memory round trip (seen in the ISA in round 3; every BN-folded ResNet layer paid it)
    if (d.scale) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * (BNW / WN) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const float sc = n < N ? d.scale[n] : 0.f;
#pragma unroll
                for (int j = 0; j < TK; ++j) acc[i][j][r] *= sc;
            }
    }
#pragma unroll
    for (int j = 0; j < TK; ++j) {
        const int kcol = kb0 + wk * (BKW / WK) + j * 32 + fr;
        if (kcol >= K) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * (BNW / WN) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (n < N) unsafeAtomicAdd(dw + (size_t)n * K + kcol, acc[i][j][r]);
            }
    }
}

template <int BNW, int BKW, int WN, int WK, int STAGES, int FAST>
__global__ __launch_bounds__(256) void igemm_wgrad_dma_kernel(const gwd_conv_desc d, float *__restrict__ dw, int m_per_block) {
    wgrad_dma_body<BNW, BKW, WN, WK, STAGES, FAST>(d, dw, m_per_block, (int)blockIdx.x, (int)gridDim.x);
}

// Weight gradients of up to WG_GROUP layers in ONE launch.  The step has ~230 weight gradients of small plain GEMMs
// (every Linear / 1x1 layer of the transformers) whose best split count gives 16-400 workgroups for 10-60 us: alone
// each leaves most of the 512 resident slots empty, and consecutive launches of one stream do not overlap.  The job
// records travel by value in the kernel arguments; every job's block range starts at a multiple of 8 so that the
// hardware's round-robin XCD assignment of the GLOBAL workgroup id is also the XCD of the id inside the job (xcd_band).
constexpr int WG_GROUP = 16;
struct WgradGroup {
    gwd_conv_desc d[WG_GROUP];
    float *dw[WG_GROUP];
    int m_per_block[WG_GROUP], block0[WG_GROUP], blocks[WG_GROUP];
    int n;
};
template <int BNW, int BKW, int WN, int WK, int STAGES, int FAST>
__global__ __launch_bounds__(256) void igemm_wgrad_group_kernel(const WgradGroup g) {
    int ji = 0;
#pragma unroll 1
    for (int k = 1; k < g.n; ++k)
        if ((int)blockIdx.x >= g.block0[k]) ji = k;
    const int id = (int)blockIdx.x - g.block0[ji];
    if (id >= g.blocks[ji]) return;                       // padding up to the next multiple of 8 (whole workgroup: no barrier is skipped)
    const gwd_conv_desc d = g.d[ji];                      // workgroup-uniform: lives in SGPRs
    wgrad_dma_body<BNW, BKW, WN, WK, STAGES, FAST>(d, g.dw[ji], g.m_per_block[ji], id, g.blocks[ji]);
}

// ----------------------------------------------------------------------------------------------
// weight gradient of 3x3 / stride 1 / pad 1 layers with Cout % 160 == 0: one staged pixel strip for all nine taps
// ----------------------------------------------------------------------------------------------
// igemm_wgrad_dma_kernel<160,128> stages 18 KB per 32-pixel step for 40 MFMAs (4 waves x 10) - 450 bytes of LDS-DMA fill per MFMA,
// every input pixel's channels fetched once per tap by different workgroups - and runs at ~6 TB/s of fill = 21 % of the MFMA peak
// (DESIGN.md section 4).  Here a workgroup owns ONE 32-channel slice of the input for ALL nine taps and a 160-wide slice of the
// outputs: 45 accumulator tiles (9 taps x five 32-column tiles) over nine or fifteen waves (layout note below).  A step is 32 consecutive pixels of one image row:
// their dY rows [32][160] (10 KB) and ONE halo patch of x - rows oh-1 .. oh+1, columns ow0-1 .. ow0+32, 32 channels: 102 x 64 B -
// land in LDS once; tap (kh, kw) reads its 32 pixels at image row kh * 34 + kw of the patch.  16.5 KB per 90 MFMAs = 183 B / MFMA.
// Both operands are needed pixel-major (the reduction runs over pixels): ds_read_b64_tr_b16, as in wgrad_dma_body; pitches of 320 B
// and 64 B put the four rows of a transposed read 16 banks apart - conflict free without a swizzle.  One workgroup per CU (80 or 48
// accumulator registers per wave), a ring of WT_STAGES steps in dynamic LDS, counted vmcnt + one barrier per step.
// The reduction over pixels is split over workgroups (tiles x splits ~ one per CU), fp32 atomics into the flat gradient.
constexpr int WT_STAGES = 8, WT_YB = 10240, WT_XB = 7168, WT_STAGE = WT_YB + WT_XB;
// Wave layout.  9: wave = tap, its five 32-column tiles (1 x fragment + 5 dY fragments per 5 MFMAs) - nine waves sit 3-2-2-2 on the four SIMDs.
// 15: wave = (kh, column tile), its three kw taps (3 x fragments at patch offsets kw + 1 dY fragment per 3 MFMAs) - 4-4-4-3.
// Same box, 9 / 15 (tools/convbench.py): 160 -> 160 0.130-0.133 / 0.126-0.131 ms, 800 -> 320 0.834 / 0.806, 320 -> 320 0.338 / 0.328-0.342; whole step
// 31.35 31.29 / 31.23 31.23 ms.  The small gain says the step is not bound by the busiest SIMD's MFMA stream (DMA fill + barrier: section 4b).
#ifndef GWD_WT_WAVES
#define GWD_WT_WAVES 15
#endif
constexpr int WT_W = GWD_WT_WAVES, WT_NX = WT_W == 9 ? 1 : 3, WT_NY = WT_W == 9 ? 5 : 1, WT_ACC = WT_NX * WT_NY;
static_assert(WT_W == 9 || WT_W == 15, "wgrad_taps wave layout");
// NB the body is a function of its own with a __restrict__ parameter ON PURPOSE: inlining it gives every memory access of the body
// alias-scope metadata, and only with that does the compiler's wait-count pass accept that the transposed LDS reads do not alias the
// LDS-DMA writes still in flight.  Without it an `s_waitcnt vmcnt(0)` lands in front of the first ds_read_b64_tr_b16 of every step and
// the ring degenerates to ONE step in flight (read off the ISA after the first version ran at 1.35 us per step; wgrad_dma_body has
// had this shape all along, by accident - plain LDS loads, as in dma_tile, are not affected).
__device__ __forceinline__ void wgrad_taps_body(const gwd_conv_desc &d, float *__restrict__ dw, char *wt_smem, const int chunks_per_split, const int wg_count) {
    typedef __bf16 T;
    const int N = d.Cout, Cin = d.Cin, H = d.Ho, W = d.Wo, K = 9 * Cin;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;          // WT_W waves: see the layout note above
    const int tiles_c = (Cin + 31) / 32, tiles = tiles_c * (N / 160);
    const int band = xcd_band((int)blockIdx.x, wg_count);                // the tiles of one pixel range are neighbours on one XCD
    const int tile = band % tiles, split = band / tiles;
    const int c0 = (tile % tiles_c) * 32, n0 = (tile / tiles_c) * 160;
    const int cpr = (W + 31) / 32, total = d.B * H * cpr;
    const int s0 = split * chunks_per_split, s1 = min(total, s0 + chunks_per_split);
    const int steps = s1 - s0;
    if (steps <= 0) return;                                               // workgroup-uniform
    const T *gy = (const T *)d.y;
    const T *x = (const T *)d.x;
    const char *zero = (const char *)d.zero_page;

    // ---- this lane's part of the 17 DMA instructions of a step: wave w issues instructions w and w + 9 (wave 8: one only)
    // j < 10: dY, 16-byte piece q = j * 64 + lane of the [32 pixels][160 channels] block; j >= 10: x patch, piece q of [102][4].
    // Everything that does not depend on the step is folded into per-lane element offsets here; the step's own coordinates
    // (image, row, first column) live in scalar registers and advance by additions - the first version recomputed them with two
    // integer divisions per wave and step, ~0.5 us of address arithmetic in front of every barrier (ablation: DMA + barrier alone
    // 0.62 us per step, reads + MFMAs alone 0.62 us, together 1.05).
    int l_off[2], l_dh[2], l_dw[2];
    bool is_y[2], l_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = wave + WT_W * i;
        is_y[i] = j < 10;
        const int q = (is_y[i] ? j : j - 10) * 64 + lane;
        if (is_y[i]) {
            const int p = q / 20;
            l_dh[i] = 0;
            l_dw[i] = p;                                                  // valid while ow0 + p < W
            l_off[i] = p * N + n0 + (q - p * 20) * 8;
            l_ok[i] = true;
        } else {
            const int pr = q >> 2, kr = pr / 34, px = pr - kr * 34, co = c0 + (q & 3) * 8;
            l_dh[i] = kr - 1;
            l_dw[i] = px - 1;
            l_off[i] = ((kr - 1) * W + (px - 1)) * Cin + co;
            l_ok[i] = q < 408 && co < Cin;
        }
    }
    const int my_loads = wave + WT_W < 17 ? 2 : 1;
    int issued = 0;
    // coordinates of the next step to request (workgroup-uniform -> scalar registers)
    int n_b = __builtin_amdgcn_readfirstlane(s0 / (H * cpr));
    int n_oh = __builtin_amdgcn_readfirstlane((s0 - n_b * (H * cpr)) / cpr);
    int n_ow = __builtin_amdgcn_readfirstlane((s0 - n_b * (H * cpr) - n_oh * cpr) * 32);
    auto issue = [&](int stage) {
        char *sb = wt_smem + stage * WT_STAGE;
        const size_t pix = (size_t)(n_b * H + n_oh) * W + n_ow;           // scalar
        const T *y_row = gy + pix * N;
        const T *x_row = x + pix * Cin;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i >= my_loads) break;                                     // wave-uniform
            const int j = wave + WT_W * i;
            const int ih = n_oh + l_dh[i], iw = n_ow + l_dw[i];
            const bool ok = l_ok[i] & ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
            const char *src = ok ? (const char *)((is_y[i] ? y_row : x_row) + l_off[i]) : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(sb + j * 1024), 16, 0, 0);
        }
        ++issued;
        n_ow += 32;
        if (n_ow >= W) {
            n_ow = 0;
            if (++n_oh == H) {
                n_oh = 0;
                ++n_b;
            }
        }
    };

    f32x16 acc[WT_ACC];
#pragma unroll
    for (int i = 0; i < WT_ACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int g = lane >> 4, t = lane & 15;
    const int kh = WT_W == 9 ? wave / 3 : wave / 5, kw = WT_W == 9 ? wave - kh * 3 : 0, ct = WT_W == 9 ? 0 : wave - kh * 5;
    const int r_lo = 8 * (g >> 1) + (t >> 2), cw = 16 * (g & 1) + 4 * (t & 3);
    const int x_base = ((kh * 34 + kw + r_lo) * 64) + cw * 2;             // byte offset of this lane's first transposed read in the patch (tap kw + xi: + xi * 64)
    const int y_base = r_lo * 320 + cw * 2 + ct * 64;

    // Software pipeline over the two 16-pixel halves of a step: while the MFMAs of one half run, the transposed reads of the next
    // half (the second half of this step, or the first half of the NEXT step) are in flight - with 2-3 waves per SIMD nothing else
    // covers the ds_read -> MFMA latency.  The barrier that certifies step s + 1 therefore sits in the MIDDLE of iteration s; the
    // ring slot refilled behind it is the one of step s - 1, whose reads every wave finished before it got here.
    typedef union { s16x4 h[2]; bf16x8 v; } Frag;
    auto read_half = [&](int slot, int ks, Frag (&bx)[WT_NX], Frag (&ay)[WT_NY]) {
        const char *Yb = wt_smem + slot * WT_STAGE;
        const char *Xb = Yb + WT_YB;
#pragma unroll
        for (int i = 0; i < WT_NX; ++i) {
            bx[i].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(Xb + x_base + i * 64 + ks * 16 * 64));
            bx[i].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(Xb + x_base + i * 64 + (ks * 16 + 4) * 64));
        }
#pragma unroll
        for (int i = 0; i < WT_NY; ++i) {
            ay[i].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(Yb + y_base + ks * 16 * 320 + i * 64));
            ay[i].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(Yb + y_base + (ks * 16 + 4) * 320 + i * 64));
        }
    };
    auto certify = [&](int c) {                         // every wave's part of step c has landed (then: barrier)
        // loads of steps c + 1 .. (last issued) may stay in flight; `issued` steps have been requested so far
        const int younger = issued - 1 - c;
        if (younger >= WT_STAGES - 3 && my_loads == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (WT_STAGES - 3)) : "memory");
        else if (younger >= WT_STAGES - 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WT_STAGES - 3) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
#pragma unroll 1
    for (int s = 0; s < WT_STAGES - 2; ++s)
        if (s < steps) issue(s);
    certify(0);
    if (WT_STAGES - 2 < steps) issue(WT_STAGES - 2);
    Frag bx0[WT_NX], ay0[WT_NY], bx1[WT_NX], ay1[WT_NY];
    read_half(0, 0, bx0, ay0);
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
        read_half(s % WT_STAGES, 1, bx1, ay1);
#pragma unroll
        for (int i = 0; i < WT_ACC; ++i) acc[i] = mma(ay0[i % WT_NY].v, bx0[i / WT_NY].v, acc[i]);
        if (s + 1 < steps) {
            certify(s + 1);
            if (s + WT_STAGES - 1 < steps) issue((s + WT_STAGES - 1) % WT_STAGES);
            read_half((s + 1) % WT_STAGES, 0, bx0, ay0);
        }
#pragma unroll
        for (int i = 0; i < WT_ACC; ++i) acc[i] = mma(ay1[i % WT_NY].v, bx1[i / WT_NY].v, acc[i]);
    }

    // ---- flush: D[row = n][col = c]; one register of a tile = two 128-byte row segments per wave instruction (full atomic rate)
    const int fr = lane & 31, fh = lane >> 5;
    const int c = c0 + fr;
    if (d.scale) {                                                        // before the first atomic: see wgrad_dma_body
#pragma unroll
        for (int i = 0; i < WT_ACC; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] *= d.scale[n0 + (ct + i % WT_NY) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh];
    }
    if (c < Cin) {
#pragma unroll
        for (int i = 0; i < WT_ACC; ++i) {
            const size_t kcol = (size_t)(kh * 3 + kw + i / WT_NY) * Cin + c;          // accumulator i: tap kw + i / NY, column tile ct + i % NY
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + (ct + i % WT_NY) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                unsafeAtomicAdd(dw + (size_t)n * K + kcol, acc[i][r]);
            }
        }
    }
}
__global__ __launch_bounds__(64 * WT_W) void wgrad_taps_kernel(const gwd_conv_desc d, float *__restrict__ dw, const int chunks_per_split, const int wg_count) {
    extern __shared__ __attribute__((aligned(1024))) char wt_dyn_smem[];
    wgrad_taps_body(d, dw, wt_dyn_smem, chunks_per_split, wg_count);
}

template <typename T>
__global__ void weight_prep_kernel(const float *__restrict__ w, const float *__restrict__ rs, T *__restrict__ wf,
                                   T *__restrict__ wd, int N, int taps, int C) {
    const size_t total = (size_t)N * taps * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t r = i / C;
        const int tap = (int)(r % taps), n = (int)(r / taps);
        const float v = rs ? w[i] * rs[n] : w[i];
        if (wf) wf[i] = from_f32<T>(v);
        if (wd) wd[((size_t)c * taps + tap) * N + n] = from_f32<T>(v);
    }
}

// Every registered weight in ONE launch: block b belongs to the job whose [block0, next block0) range holds it
// (binary search over the table, workgroup-uniform).  A block owns one 32(n) x 32(c) tile of one filter tap and
// transposes it through LDS: reads run along c (128-byte fp32 rows), the [C][taps][N] data-gradient copy is written
// along n (64-byte bf16 rows) - the naive element-per-thread version wrote 2 bytes per cache line (10x off HBM speed).
__global__ __launch_bounds__(256) void weight_prep_batch_kernel(const gwd_prep_job *__restrict__ jobs, int n_jobs,
                                                                const int32_t *__restrict__ block_job) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n_jobs - 1;
    if (block_job) {
        lo = block_job[blockIdx.x];           // one load instead of ~9 dependent ones: the search was most of a block's life
    } else {
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
    }
    const gwd_prep_job j = jobs[lo];
    const int ct = (j.C + 31) / 32, nt = (j.N + 31) / 32;
    int t = (int)blockIdx.x - j.block0;
    const int c0 = (t % ct) * 32;
    t /= ct;
    const int n0 = (t % nt) * 32, tap = t / nt;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    __bf16 *wf = (__bf16 *)j.w_fwd, *wd = (__bf16 *)j.w_dgrad;
    // zero-padded destinations (Np > 0): rows N -> Np, every group of Cg input channels -> Cgp; the padding is never written
    // (the caller zeroes the copies once), only the index map changes
    const bool pad = j.Np > 0;
    const int Np = pad ? j.Np : j.N, Cp = pad ? (j.C / j.Cg) * j.Cgp : j.C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int n = n0 + ty + 8 * k, c = c0 + tx;
        float v = 0.f;
        if (n < j.N && c < j.C) {
            const size_t i = ((size_t)n * j.taps + tap) * j.C + c;
            v = j.row_scale ? j.w[i] * j.row_scale[n] : j.w[i];
            if (wf) wf[pad ? ((size_t)n * j.taps + tap) * Cp + (c / j.Cg) * j.Cgp + c % j.Cg : i] = (__bf16)v;
        }
        tile[ty + 8 * k][tx] = v;
    }
    if (!wd) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, n = n0 + tx;
        if (n < j.N && c < j.C) {
            const int cp = pad ? (c / j.Cg) * j.Cgp + c % j.Cg : c;
            wd[((size_t)cp * j.taps + tap) * Np + n] = (__bf16)tile[tx][ty + 8 * k];
        }
    }
}

// dst (N, taps, G*Cg) += src (Np, taps, G*Cgp): the weight gradient of a layer that ran on zero-padded channel counts, folded back
// onto the parameter's own shape.  Jobs by value in the kernel arguments (capturable, nothing to upload), one 256-thread block
// per 256 destination elements.
struct UnpadBatch {
    gwd_unpad_job j[GWD_UNPAD_BATCH];
    int n;
};
__global__ __launch_bounds__(256) void unpad_add_batch_kernel(const UnpadBatch b) {
    int ji = 0;
#pragma unroll 1
    for (int k = 1; k < b.n; ++k)
        if ((int)blockIdx.x >= b.j[k].block0) ji = k;
    const gwd_unpad_job j = b.j[ji];
    const int i = ((int)blockIdx.x - j.block0) * 256 + threadIdx.x;
    const int C = j.G * j.Cg;
    if (i >= j.N * j.taps * C) return;
    const int c = i % C, r = i / C;                      // r = n * taps + tap
    j.dst[i] += j.src[(size_t)r * (j.G * j.Cgp) + (c / j.Cg) * j.Cgp + c % j.Cg];
}

int check_desc(const gwd_conv_desc *d) {
    if (!d || !d->x || !d->y) return -1;
    if (d->dtype != GWD_F32 && d->dtype != GWD_BF16) return -2;
    if (d->B <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Cin <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0) return -3;
    if (d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->pad < 0) return -4;
    if (d->gather < 0 || d->gather > 2) return -5;
    if (d->gather == GWD_GATHER_UPSAMPLED && (d->stride != 1 || d->Hv <= 0 || d->Wv <= 0)) return -6;
    if ((int64_t)d->B * d->Ho * d->Wo >= (1LL << 31) || (int64_t)d->KH * d->KW * d->Cin >= (1LL << 31)) return -7;
    if ((int64_t)d->B * d->Hi * d->Wi * d->Cin >= (1LL << 40)) return -7;
    if (d->gate && ((d->gate_act != GWD_ACT_RELU && d->gate_act != GWD_ACT_ELU && d->gate_act != GWD_ACT_GELU) || d->mult)) return -4;
    return 0;
}

// Fixed choices (each was an A/B switch while it was being measured; the measurements are in DESIGN.md section 4):
constexpr bool big_tiles_enabled() { return true; }      // 256-row tiles from M >= 131 072 on
constexpr bool tail_enabled() { return true; }           // zero-page channel tail for Cin % 32 != 0 (the 80-channel layers)
constexpr int small_tile_threshold() { return 320; }     // fewer 128 x 128 tiles than this: 64 x 64 tiles


// ----------------------------------------------------------------------------------------------
// long reductions on few rows: K split over the waves of a workgroup
// ----------------------------------------------------------------------------------------------
// The FFN 2048 -> 256 GEMMs of the DETR layers (800 / 2 400 rows) and the 1 024- / 2 048-deep 1x1 layers of ResNet layer3 / 4 fill
// 50-600 tiles of 64 x 64 - at most a workgroup or two per CU - and then walk K / 32 = 32-64 steps, each one a barrier, an LDS round
// trip and two dependent MFMAs per wave: 0.34 us per step whatever the ring depth (8 stages measured: no change), 22 us for 0.4 GFLOP.
// Here every WAVE owns the whole 64 x 64 tile (4 independent accumulators: 8 MFMAs per step instead of 2) over every fourth K step,
// with a private LDS-DMA ring (no workgroup barrier in the loop); the four partial tiles meet in LDS and the usual epilogue runs once.
// The 3x3 layers of ResNet layer4 and of the 1/32- and 1/64-resolution pyramid levels (K = 1 440 ... 4 608 on 560 / 2 400 pixels) are the
// same problem with a gather: plain gather with any stride or transposed gather with stride 1, bf16, Cin % 32 == 0, N % 8 == 0.
template <int STAGES, bool MULT, bool LEAN = false>
__global__ __launch_bounds__(256) void gemm_ksplit_kernel(const gwd_conv_desc d, const int tile_count) {
    typedef __bf16 T;
    constexpr int BM = 64, BN = 64, STAGE_BYTES = (BM + BN) * 64, WAVE_BYTES = STAGES * STAGE_BYTES, RP = 68;    // RP: fp32 pitch of the partial tiles
    static_assert(4 * BM * RP * 4 <= 4 * WAVE_BYTES, "the partial tiles reuse the rings");
    extern __shared__ __attribute__((aligned(1024))) char ks_smem[];
    const int M = d.B * d.Ho * d.Wo, N = d.Cout, K = d.KH * d.KW * d.Cin;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_tiles = (N + BN - 1) / BN;
    const int tile = xcd_band(blockIdx.x, tile_count);
    const int m0 = (tile / n_tiles) * BM, n0 = (tile % n_tiles) * BN;
    const T *x = (const T *)d.x;
    const T *wgt = (const T *)d.w;
    const char *zero = (const char *)d.zero_page;
    char *ring = ks_smem + wave * WAVE_BYTES;

    // one DMA instruction moves 16 rows x 64 B; lane (r, c) fetches source chunk c ^ swizzle(row) so that the fragment reads below
    // are bank-conflict free (same scheme as igemm_dma_kernel).  K tile kt = (filter tap, 32-channel slice): kt * 32 is the offset in
    // a weight row ([Cout][KH][KW][Cin]); the A row of an output pixel comes from the tap's source pixel or from the zero page.
    const bool transposed = d.gather == GWD_GATHER_TRANSPOSED;
    const int cpt = d.Cin / 32;                           // K tiles per filter tap
    int a_ph[4], a_pw[4], a_ck[4];
    size_t a_pix[4];
    bool a_ok[4];
    const char *b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 16 * i + (lane >> 2);
        const int ck = ((lane & 3) ^ ((row >> 2) & 3)) * 8;
        const int m = m0 + row, n = n0 + row;
        a_ok[i] = m < M;
        const int mm = a_ok[i] ? m : 0;
        const int b = mm / (d.Ho * d.Wo);
        const int rem = mm - b * (d.Ho * d.Wo);
        const int oh = rem / d.Wo, ow = rem - oh * d.Wo;
        a_pix[i] = (size_t)b * d.Hi * d.Wi;
        a_ph[i] = transposed ? oh + d.pad : oh * d.stride - d.pad;
        a_pw[i] = transposed ? ow + d.pad : ow * d.stride - d.pad;
        a_ck[i] = ck;
        b_src[i] = n < N ? (const char *)(wgt + (size_t)n * K + ck) : nullptr;
    }
    const int KT = K / 32;
    const int steps = (KT - wave + 3) / 4;                // this wave's K tiles: wave, wave + 4, ... (the four waves read 256 contiguous bytes of a row)
    auto issue = [&](int s, int stage) {
        char *sb = ring + stage * STAGE_BYTES;
        const int kt = wave + 4 * s;
        const int tap = kt / cpt, c0 = (kt - tap * cpt) * 32;
        const int kh = tap / d.KW, kw = tap - kh * d.KW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ih = transposed ? a_ph[i] - kh : a_ph[i] + kh, iw = transposed ? a_pw[i] - kw : a_pw[i] + kw;
            const bool ok = a_ok[i] & ((unsigned)ih < (unsigned)d.Hi) & ((unsigned)iw < (unsigned)d.Wi);
            const char *sa = ok ? (const char *)(x + (a_pix[i] + (size_t)ih * d.Wi + iw) * d.Cin + c0 + a_ck[i]) : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)sa,
                                             (__attribute__((address_space(3))) void *)(sb + i * 1024), 16, 0, 0);
        }
        const size_t koff = (size_t)kt * 64;              // bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char *sw = b_src[i] ? b_src[i] + koff : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)sw,
                                             (__attribute__((address_space(3))) void *)(sb + BM * 64 + i * 1024), 16, 0, 0);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < steps) issue(t, t);
    for (int s = 0; s < steps; ++s) {
        if (s + STAGES - 2 < steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (STAGES - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // the buffer refilled now was read by the MFMAs of the previous step, which have issued (their operands had arrived)
        if (s + STAGES - 1 < steps) issue(s + STAGES - 1, (s + STAGES - 1) % STAGES);
        const T *As = (const T *)(ring + (s % STAGES) * STAGE_BYTES);
        const T *Bs = As + BM * 32;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = i * 32 + fr;
                af[i] = *(const bf16x8 *)(As + row * 32 + (((ks * 2 + fh) ^ ((row >> 2) & 3)) * 8));
                bfr[i] = *(const bf16x8 *)(Bs + row * 32 + (((ks * 2 + fh) ^ ((row >> 2) & 3)) * 8));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mma(af[i], bfr[j], acc[i][j]);
        }
    }
    __syncthreads();                                      // every ring is drained: the partial tiles take their place
    float *red = (float *)ks_smem;                        // [wave][64][RP]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[(wave * BM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * RP + j * 32 + fr] = acc[i][j][r];
    __syncthreads();
    // epilogue: thread = (row, 16 consecutive columns) = two 16-byte vectors; same arithmetic and order as igemm_dma_kernel
    const int row = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 16;
    const int m = m0 + row;
    if (m >= M) return;
    T *y = (T *)d.y;
    T *z = LEAN ? nullptr : (T *)d.z;
    const T *res = (const T *)d.residual;
    const T *mul = MULT ? (const T *)d.mult : nullptr;
    const T *gate = MULT ? nullptr : (const T *)d.gate;
#pragma unroll
    for (int hv = 0; hv < 2; ++hv) {
        const int nb = n0 + c0 + 8 * hv;
        if (nb >= N) continue;
        float v[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x4 sum = *(const f32x4 *)(red + (size_t)row * RP + c0 + 8 * hv + 4 * q);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 p = *(const f32x4 *)(red + (size_t)(w * BM + row) * RP + c0 + 8 * hv + 4 * q);
                sum[0] += p[0]; sum[1] += p[1]; sum[2] += p[2]; sum[3] += p[3];
            }
            v[4 * q] = sum[0]; v[4 * q + 1] = sum[1]; v[4 * q + 2] = sum[2]; v[4 * q + 3] = sum[3];
        }
        const size_t o = (size_t)m * N + nb;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * (d.scale ? d.scale[nb + e] : 1.0f) + (d.shift ? d.shift[nb + e] : 0.0f);
        if (res && !mul) {
            const bf16x8 rv = *(const bf16x8 *)(res + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
        }
        bf16x8 out;
        if (z) {
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = (__bf16)v[e];
            *(bf16x8 *)(z + o) = out;
        }
        if (mul) {                                        // y = act_scale * act(v) * mult + residual (dropout, then the skip)
            const bf16x8 mv = *(const bf16x8 *)(mul + o);
            bf16x8 rv;
#pragma unroll
            for (int e = 0; e < 8; ++e) rv[e] = (__bf16)0.0f;
            if (res) rv = *(const bf16x8 *)(res + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = (__bf16)((LEAN ? v[e] : apply_act(v[e], d.act)) * d.act_scale * (float)mv[e] + (float)rv[e]);
        } else {
            float gm[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) gm[e] = d.act_scale;
            if (gate) {
                const bf16x8 gv = *(const bf16x8 *)(gate + o);
#pragma unroll
                for (int e = 0; e < 8; ++e) gm[e] = gate_grad(d.act_scale, (float)gv[e], d.gate_act);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = (__bf16)((LEAN ? v[e] : apply_act(v[e], d.act)) * gm[e]);
        }
        *(bf16x8 *)(y + o) = out;
    }
}

constexpr int ksplit_min_k() { return 1024; }

// 1 = launched
static int launch_ksplit(const gwd_conv_desc *d, hipStream_t s) {
    const int M = d->B * d->Ho * d->Wo, N = d->Cout, K = d->KH * d->KW * d->Cin;
    if (ksplit_min_k() <= 0 || K < ksplit_min_k() || d->dtype != GWD_BF16 || !d->zero_page) return 0;
    if (d->gather == GWD_GATHER_UPSAMPLED || (d->gather == GWD_GATHER_TRANSPOSED && d->stride != 1)) return 0;
    if ((d->Cin % 32) || (N % 8)) return 0;
    if (d->gate && d->gate_act == GWD_ACT_GELU) return 0;         // the GELU gate lives in the LDS-DMA tile kernels only
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    if (tiles > 256) return 0;                            // one workgroup (128 KiB of LDS) per CU: beyond one round the ordinary tiles win (304 tiles, K = 4 608: 54 -> 58 us; 600 tiles: 17 -> 21 us)
    constexpr int ST = 4, LDS = 4 * ST * (64 + 64) * 64;  // 128 KiB: one workgroup per CU
    const bool lean = d->act == GWD_ACT_NONE && !d->z;
#define KS_LAUNCH(MULT_, LEAN_)                                                                                                    \
    {                                                                                                                              \
        static bool attr = false;                                                                                                  \
        if (!attr) {                                                                                                               \
            (void)hipFuncSetAttribute((const void *)gemm_ksplit_kernel<ST, MULT_, LEAN_>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
            attr = true;                                                                                                           \
        }                                                                                                                          \
        gemm_ksplit_kernel<ST, MULT_, LEAN_><<<(unsigned)tiles, 256, LDS, s>>>(*d, (int)tiles);                                     \
    }
    if (d->mult) {
        if (lean) KS_LAUNCH(true, true) else KS_LAUNCH(true, false)
    } else {
        if (lean) KS_LAUNCH(false, true) else KS_LAUNCH(false, false)
    }
#undef KS_LAUNCH
    return 1;
}

static bool dma_enabled() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("GWD_IGEMM_DMA");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

// gwd_conv_desc.ln_mean != NULL: convolution with the ConvLn epilogue (dma_tile<..., LN>).  0 = launched, -4 = no fused kernel for the shape.
// 3x3 / stride 1 / pad 1 on a map of whole 8 x 32 pixel patches, whole 32-channel blocks: the halo-patch variant of the 256 x 160 tile
static bool halo_ok(const gwd_conv_desc *d) {
#ifdef GWD_NO_HALO
    return false;
#endif
    return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Hi == d->Ho && d->Wi == d->Wo && (d->Wo % 32) == 0 && (d->Ho % 8) == 0 &&
           (d->Cin % 32) == 0 && (d->Cout % 160) == 0 && (d->gather == GWD_GATHER_CONV || d->gather == GWD_GATHER_TRANSPOSED);
}

// data gradient of a 3x3 / stride 2 / pad 1 convolution onto a map of exactly twice the size: the parity-class variant
static bool par_ok(const gwd_conv_desc *d) {
#ifdef GWD_NO_PAR
    return false;
#endif
    return d->gather == GWD_GATHER_TRANSPOSED && d->stride == 2 && d->KH == 3 && d->KW == 3 && d->pad == 1 && d->Ho == 2 * d->Hi && d->Wo == 2 * d->Wi &&
           (d->Cin % 32) == 0;
}

static int launch_convln(const gwd_conv_desc *d, hipStream_t s) {
    const int M = d->B * d->Ho * d->Wo, N = d->Cout;
    if (d->dtype != GWD_BF16 || !dma_enabled() || !d->zero_page || !d->ln_rstd || !d->scale || !d->shift || d->mult || d->gate) return -4;
    if (d->gather != GWD_GATHER_CONV || (d->act != GWD_ACT_NONE && d->act != GWD_ACT_GELU) || d->act_scale != 1.0f) return -4;
    if (d->ln_C <= 0 || d->ln_C > N || (N % 8) || (d->Cin % 8)) return -4;
    const bool tail = (d->Cin % 32) != 0;
    if (tail && d->Cin < 32) return -4;
    const bool gelu = d->act == GWD_ACT_GELU;
#define LN_LAUNCH(BM_, BN_, WM_, WN_, ST_, TAIL_, GRID)                                                                           \
    {                                                                                                                             \
        const dim3 g_(GRID);                                                                                                      \
        if (gelu) igemm_dma_kernel<BM_, BN_, WM_, WN_, ST_, 0, false, TAIL_, false, 2, true><<<g_, WM_ * WN_ * 64, 0, s>>>(*d, 0, (int)g_.x); \
        else igemm_dma_kernel<BM_, BN_, WM_, WN_, ST_, 0, false, TAIL_, false, 0, true><<<g_, WM_ * WN_ * 64, 0, s>>>(*d, 0, (int)g_.x);      \
    }
    if (N == 160) {
        if (big_tiles_enabled() && M >= 256 * 512) {
            if (tail) LN_LAUNCH(256, 160, 8, 1, 3, true, (M + 255) / 256)
            else if (halo_ok(d)) {
                const dim3 g_((M + 255) / 256);
                if (gelu) igemm_dma_kernel<256, 160, 8, 1, 3, 0, false, false, false, 2, true, 1, 32, true><<<g_, 512, 0, s>>>(*d, 0, (int)g_.x);
                else igemm_dma_kernel<256, 160, 8, 1, 3, 0, false, false, false, 0, true, 1, 32, true><<<g_, 512, 0, s>>>(*d, 0, (int)g_.x);
            } else LN_LAUNCH(256, 160, 8, 1, 3, false, (M + 255) / 256)
            return 0;
        }
        // smaller maps (the PSP branches on pooled maps) run 128 x 160 tiles at two waves per SIMD: the longer epilogue costs more than
        // the separate LayerNorm launch there (8x30x40: 45.3 us as two launches, 49.5 us fused) - not fused
        return -4;
    }
    if (tail) return -4;
    if (N <= 32) LN_LAUNCH(128, 32, 4, 1, 4, false, (M + 127) / 128)
    else if (N <= 64) LN_LAUNCH(128, 64, 4, 1, 4, false, (M + 127) / 128)
    else return -4;
#undef LN_LAUNCH
    return 0;
}

template <typename T>
int launch_fwd(const gwd_conv_desc *d, hipStream_t s) {
    const int M = d->B * d->Ho * d->Wo, N = d->Cout;
    constexpr int BK = Cfg<T>::BK;
    const unsigned gm = (M + 127) / 128;
    if (d->ln_mean) {                         // ConvLn epilogue: a fused kernel or -4 (the caller runs convolution and LayerNorm apart)
        const int rc = launch_convln(d, s);
        if (rc) return rc;
        GWD_CHECK_LAUNCH();
        return 0;
    }
    if constexpr (sizeof(T) == 2) {
        if (launch_ksplit(d, s)) {
            GWD_CHECK_LAUNCH();
            return 0;
        }
    }
    if (d->mult) {
        // element-wise multiplier in the epilogue (dropout + skip of the DETR sub-layers: GEMMs with M <= a few thousand rows):
        // dedicated instantiations of the 64x64 tiles, so that the multiplier path costs the other kernels nothing
        if constexpr (sizeof(T) == 2) {
            if (dma_enabled() && d->zero_page && (d->Cin % 32) == 0 && (N % 8) == 0 && d->gather == GWD_GATHER_CONV) {
                const dim3 g(((M + 63) / 64) * ((N + 63) / 64));
                if (!d->z && d->act == GWD_ACT_NONE) igemm_dma_kernel<64, 64, 2, 2, 4, 0, true, false, false, 0><<<g, 256, 0, s>>>(*d, 0, (int)g.x);
                else if (!d->z && d->act == GWD_ACT_RELU) igemm_dma_kernel<64, 64, 2, 2, 4, 0, true, false, false, 1><<<g, 256, 0, s>>>(*d, 0, (int)g.x);
                else igemm_dma_kernel<64, 64, 2, 2, 4, 0, true><<<g, 256, 0, s>>>(*d, 0, (int)g.x);
                GWD_CHECK_LAUNCH();
                return 0;
            }
        }
        igemm_fwd_kernel<T, 64, 64, 2, 2, BK, true><<<dim3((M + 63) / 64, (N + 63) / 64), 256, 0, s>>>(*d);
        GWD_CHECK_LAUNCH();
        return 0;
    }
    if constexpr (sizeof(T) == 2) {
        // Cin = 8 (mod 32) multiples such as the 80-channel pyramid: the LDS-DMA kernels with a zero-page channel tail (big maps, the two
        // hot tile shapes, plain and stride-1 transposed gathers) instead of the register-staged kernel (112-166 us per launch)
        const int actk = d->act == GWD_ACT_GELU ? 2 : (d->z ? -1 : (d->act == GWD_ACT_NONE ? 0 : (d->act == GWD_ACT_RELU ? 1 : -1)));
        const bool lean = actk == 0;
        if (dma_enabled() && tail_enabled() && !d->gate && d->zero_page && (d->Cin % 32) != 0 && (d->Cin % 8) == 0 && d->Cin > 32 && (N % 8) == 0 && M >= 256 * 512 &&
            (d->gather == GWD_GATHER_CONV || (d->gather == GWD_GATHER_TRANSPOSED && d->stride == 1))) {
            const bool tr = d->gather != GWD_GATHER_CONV;
            const unsigned gm2 = (M + 255) / 256;
            if (N % 160 == 0) {
                const dim3 g(gm2 * (N / 160));
                if (lean) {
                    if (tr) igemm_dma_kernel<256, 160, 8, 1, 3, 1, false, true, false, 0><<<g, 512, 0, s>>>(*d, 0, (int)g.x);
                    else igemm_dma_kernel<256, 160, 8, 1, 3, 0, false, true, false, 0><<<g, 512, 0, s>>>(*d, 0, (int)g.x);
                } else if (tr) igemm_dma_kernel<256, 160, 8, 1, 3, 1, false, true><<<g, 512, 0, s>>>(*d, 0, (int)g.x);
                else igemm_dma_kernel<256, 160, 8, 1, 3, 0, false, true><<<g, 512, 0, s>>>(*d, 0, (int)g.x);
                GWD_CHECK_LAUNCH();
                return 0;
            }
            if (N > 64) {
                const dim3 g(gm2 * ((N + 127) / 128));
                if (lean) {
                    if (tr) igemm_dma_kernel<256, 128, 4, 2, 3, 1, false, true, false, 0><<<g, 512, 0, s>>>(*d, 0, (int)g.x);
                    else igemm_dma_kernel<256, 128, 4, 2, 3, 0, false, true, false, 0><<<g, 512, 0, s>>>(*d, 0, (int)g.x);
                } else if (tr) igemm_dma_kernel<256, 128, 4, 2, 3, 1, false, true><<<g, 512, 0, s>>>(*d, 0, (int)g.x);
                else igemm_dma_kernel<256, 128, 4, 2, 3, 0, false, true><<<g, 512, 0, s>>>(*d, 0, (int)g.x);
                GWD_CHECK_LAUNCH();
                return 0;
            }
        }
        // a gate on a 160-wide layer or together with an activation / a pre-activation copy (none in the model) goes to the
        // register-staged kernel below: the 160-wide tiles stay gate-free, the gated variants lean
        if (dma_enabled() && d->zero_page && (d->Cin % 32) == 0 && (N % 8) == 0 && !(d->gate && (N % 160 == 0 || !lean))) {
            const int gmk = d->gather == GWD_GATHER_CONV ? 0 : ((d->gather == GWD_GATHER_TRANSPOSED && d->stride == 1) ? 1 : 2);
#define DMA_LAUNCH_G(BM_, BN_, WM_, WN_, ST_, GRID, G_, L_, KPB_)                                              \
    switch (gmk) {                                                                                              \
        case 0: igemm_dma_kernel<BM_, BN_, WM_, WN_, ST_, 0, false, false, G_, L_, false, KPB_><<<GRID, WM_ * WN_ * 64, 0, s>>>(*d, 0, (int)(GRID).x); break;   \
        case 1: igemm_dma_kernel<BM_, BN_, WM_, WN_, ST_, 1, false, false, G_, L_, false, KPB_><<<GRID, WM_ * WN_ * 64, 0, s>>>(*d, 0, (int)(GRID).x); break;   \
        default: igemm_dma_kernel<BM_, BN_, WM_, WN_, ST_, 2, false, false, G_, L_, false, KPB_><<<GRID, WM_ * WN_ * 64, 0, s>>>(*d, 0, (int)(GRID).x); break;  \
    }
#define DMA_LAUNCH_K(BM_, BN_, WM_, WN_, ST_, GRID, KPB_)                                                      \
    if (BN_ % 160 != 0 && d->gate && d->gate_act == GWD_ACT_GELU) { DMA_LAUNCH_G(BM_, BN_, WM_, WN_, ST_, GRID, (BN_ % 160 != 0 ? 2 : 0), 0, KPB_) }   \
    else if (BN_ % 160 != 0 && d->gate) { DMA_LAUNCH_G(BM_, BN_, WM_, WN_, ST_, GRID, (BN_ % 160 != 0 ? 1 : 0), 0, KPB_) }   \
    else if (actk == 0) { DMA_LAUNCH_G(BM_, BN_, WM_, WN_, ST_, GRID, 0, 0, KPB_) }                         \
    else if (actk == 1 && BN_ % 160 != 0) { DMA_LAUNCH_G(BM_, BN_, WM_, WN_, ST_, GRID, 0, (BN_ % 160 != 0 ? 1 : -1), KPB_) } \
    else if (actk == 2 && BN_ % 160 != 0) { DMA_LAUNCH_G(BM_, BN_, WM_, WN_, ST_, GRID, 0, (BN_ % 160 != 0 ? 2 : -1), KPB_) } \
    else { DMA_LAUNCH_G(BM_, BN_, WM_, WN_, ST_, GRID, 0, -1, KPB_) }
#define DMA_LAUNCH(BM_, BN_, WM_, WN_, ST_, GRID) DMA_LAUNCH_K(BM_, BN_, WM_, WN_, ST_, GRID, 1)
#define DMA_LAUNCH_G64(BM_, BN_, WM_, WN_, ST_, GRID, G_, L_)                                                   \
    switch (gmk) {                                                                                              \
        case 0: igemm_dma_kernel<BM_, BN_, WM_, WN_, ST_, 0, false, false, G_, L_, false, 1, 64><<<GRID, WM_ * WN_ * 64, 0, s>>>(*d, 0, (int)(GRID).x); break;   \
        case 1: igemm_dma_kernel<BM_, BN_, WM_, WN_, ST_, 1, false, false, G_, L_, false, 1, 64><<<GRID, WM_ * WN_ * 64, 0, s>>>(*d, 0, (int)(GRID).x); break;   \
        default: igemm_dma_kernel<BM_, BN_, WM_, WN_, ST_, 2, false, false, G_, L_, false, 1, 64><<<GRID, WM_ * WN_ * 64, 0, s>>>(*d, 0, (int)(GRID).x); break;  \
    }
#define DMA_LAUNCH_64(BM_, BN_, WM_, WN_, ST_, GRID)                                                           \
    if (d->gate && d->gate_act == GWD_ACT_GELU) { DMA_LAUNCH_G64(BM_, BN_, WM_, WN_, ST_, GRID, 2, 0) }          \
    else if (d->gate) { DMA_LAUNCH_G64(BM_, BN_, WM_, WN_, ST_, GRID, 1, 0) }                                   \
    else if (actk == 0) { DMA_LAUNCH_G64(BM_, BN_, WM_, WN_, ST_, GRID, 0, 0) }                             \
    else if (actk == 1) { DMA_LAUNCH_G64(BM_, BN_, WM_, WN_, ST_, GRID, 0, 1) }                             \
    else if (actk == 2) { DMA_LAUNCH_G64(BM_, BN_, WM_, WN_, ST_, GRID, 0, 2) }                             \
    else { DMA_LAUNCH_G64(BM_, BN_, WM_, WN_, ST_, GRID, 0, -1) }
            // 64-channel K tiles (whole 128-byte lines per staged row piece) where every tap is a whole number of them; for the 64 x 64
            // tiles only (128 x 64 and 128 x 128 with two stages: no gain, measured)
            const bool bk64 = (d->Cin % 64) == 0;
            // otherwise two 32-channel K tiles per barrier for them (needs an even number of K tiles)
            const bool kpb2 = ((d->KH * d->KW * (d->Cin / 32)) % 2) == 0;
            const bool big = big_tiles_enabled() && M >= 256 * 512;      // >= 2 workgroups per CU with 256-row tiles
            const unsigned gm2 = (M + 255) / 256;
            if (gmk == 2 && par_ok(d) && N > 64 && lean && !(d->gate && d->gate_act == GWD_ACT_GELU)) {
                // stride-2 3x3 data gradient by parity class (dma_tile<..., PAR>): 4 x ceil(M / 4 / BM) row tiles
                const int mq = M / 4;
                const unsigned t128 = 4u * ((mq + 127) / 128) * ((N + 127) / 128);
                if (t128 >= 512 || !bk64) {
                    const dim3 g(t128);
                    if (d->gate) igemm_dma_kernel<128, 128, 2, 2, 3, 2, false, false, true, 0, false, 1, 32, false, true><<<g, 256, 0, s>>>(*d, 0, (int)g.x);
                    else igemm_dma_kernel<128, 128, 2, 2, 3, 2, false, false, false, 0, false, 1, 32, false, true><<<g, 256, 0, s>>>(*d, 0, (int)g.x);
                } else {
                    const dim3 g(4u * ((mq + 63) / 64) * ((N + 63) / 64));
                    if (d->gate) igemm_dma_kernel<64, 64, 2, 2, 3, 2, false, false, true, 0, false, 1, 64, false, true><<<g, 256, 0, s>>>(*d, 0, (int)g.x);
                    else igemm_dma_kernel<64, 64, 2, 2, 3, 2, false, false, false, 0, false, 1, 64, false, true><<<g, 256, 0, s>>>(*d, 0, (int)g.x);
                }
            } else if (N % 160 == 0) {
                if (big && gmk <= 1 && halo_ok(d)) {
                    const dim3 g(gm2 * (N / 160));
                    if (actk == 0) {
                        if (gmk == 0) igemm_dma_kernel<256, 160, 8, 1, 3, 0, false, false, false, 0, false, 1, 32, true><<<g, 512, 0, s>>>(*d, 0, (int)g.x);
                        else igemm_dma_kernel<256, 160, 8, 1, 3, 1, false, false, false, 0, false, 1, 32, true><<<g, 512, 0, s>>>(*d, 0, (int)g.x);
                    } else {
                        if (gmk == 0) igemm_dma_kernel<256, 160, 8, 1, 3, 0, false, false, false, -1, false, 1, 32, true><<<g, 512, 0, s>>>(*d, 0, (int)g.x);
                        else igemm_dma_kernel<256, 160, 8, 1, 3, 1, false, false, false, -1, false, 1, 32, true><<<g, 512, 0, s>>>(*d, 0, (int)g.x);
                    }
                } else if (big) { DMA_LAUNCH(256, 160, 8, 1, 3, dim3(gm2 * (N / 160))) } else { DMA_LAUNCH(128, 160, 4, 1, 3, dim3(gm * (N / 160))) }
            } else if (N > 64) {
                const unsigned t128 = gm * ((N + 127) / 128);
                const int small_thr = small_tile_threshold();
                if (big) { DMA_LAUNCH(256, 128, 4, 2, 3, dim3(gm2 * ((N + 127) / 128))) }
                else if ((int)t128 < small_thr) {
                    // fewer 128x128 tiles than CUs: quarter tiles put four times as many workgroups on the chip
                    if (bk64) { DMA_LAUNCH_64(64, 64, 2, 2, 3, dim3(((M + 63) / 64) * ((N + 63) / 64))) }
                    else if (kpb2) { DMA_LAUNCH_K(64, 64, 2, 2, 6, dim3(((M + 63) / 64) * ((N + 63) / 64)), 2) }
                    else { DMA_LAUNCH(64, 64, 2, 2, 4, dim3(((M + 63) / 64) * ((N + 63) / 64))) }
                } else { DMA_LAUNCH(128, 128, 2, 2, 3, dim3(t128)) }
            } else if (N > 32) {
                // fewer 128-row tiles than two per CU: 64-row tiles (three workgroups per CU fit) run the map in one round instead of
                // a round and a tail (conv3x3 64 -> 64 @ 8x60x80: 300 workgroups of 128 x 64 = 1.17 rounds)
                if (bk64 && gm < 512 && d->KH * d->KW > 1) { DMA_LAUNCH_64(64, 64, 2, 2, 3, dim3((M + 63) / 64)) }      // same box: 18 -> 12 us, step -0.1 ms; 1x1: no gain
                else DMA_LAUNCH(128, 64, 2, 2, 4, dim3(gm))
            } else {
                DMA_LAUNCH(128, 32, 4, 1, 4, dim3(gm))
            }
#undef DMA_LAUNCH
#undef DMA_LAUNCH_64
#undef DMA_LAUNCH_G64
#undef DMA_LAUNCH_K
#undef DMA_LAUNCH_G
            GWD_CHECK_LAUNCH();
            return 0;
        }
    }
    if (d->gate && d->gate_act == GWD_ACT_GELU) return -4;       // GELU gate: LDS-DMA tile kernels only; the caller applies it in a pass of its own
    // fewer than two tiles per CU (the 60/120-channel pyramid at 1/8 resolution: 300 tiles of 128 rows = one full round plus
    // a tail round of 44): quarter tiles give four times as many workgroups
    const bool quarter = N > 32 && gm * ((N + 127) / 128) < 512;
    if (N % 160 == 0) {                       // 160 / 320 channel pyramids: exact tiles, no padded columns
        igemm_fwd_kernel<T, 128, 160, 4, 1, BK><<<dim3(gm, N / 160), 256, 0, s>>>(*d);
    } else if (quarter) {
        igemm_fwd_kernel<T, 64, 64, 2, 2, BK><<<dim3((M + 63) / 64, (N + 63) / 64), 256, 0, s>>>(*d);
    } else if (N > 64) {
        igemm_fwd_kernel<T, 128, 128, 2, 2, BK><<<dim3(gm, (N + 127) / 128), 256, 0, s>>>(*d);
    } else if (N > 32) {
        igemm_fwd_kernel<T, 128, 64, 2, 2, BK><<<dim3(gm, 1), 256, 0, s>>>(*d);
    } else {
        igemm_fwd_kernel<T, 128, 32, 4, 1, BK><<<dim3(gm, 1), 256, 0, s>>>(*d);
    }
    GWD_CHECK_LAUNCH();
    return 0;
}

#ifndef GWD_WG_BAL1
#define GWD_WG_BAL1 1.3e12     // nominal flush rate (bytes/s) behind the split count of grouped 3x3 / strided weight gradients (0 = fill the chip with each layer: +0.15 ms per step)
#endif
constexpr int wgrad_target_blocks(int resident) { return resident; }     // exactly one full round of resident workgroups (no tail round)
constexpr int wgrad_per_cu_128() { return 2; }                            // resident 128 x 128 weight-gradient workgroups per CU the split count aims at

static void wgrad_split(int M, int tiles, int rm, int &splits, int &m_per_block, int resident = 768, double balance = 0.0) {
    // M-splits so that tiles x splits fills the chip's resident workgroup slots once; >= 8 reduction steps each
    splits = wgrad_target_blocks(resident) / tiles;
    if (splits * tiles < wgrad_target_blocks(resident) * 3 / 4) ++splits;     // far below a full round: round up instead
    if (balance > 0.0) {
        // every split flushes tiles x (tile bytes) of fp32 atomics (~1.3 TB/s chip-wide) but shortens each workgroup's
        // serial chain of (M / 32 / splits) steps: time ~ steps * t_step / S + S * flush  ->  S* = sqrt(steps * t_step / flush),
        // passed in as balance = t_step / (flush time of one split)
        const int s_opt = (int)(sqrt((double)(M / rm) * balance) + 0.5);
        if (s_opt >= 1 && s_opt < splits) splits = s_opt;
    }
    const int max_splits = (M + 8 * rm - 1) / (8 * rm);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    m_per_block = (M + splits - 1) / splits;
    m_per_block = ((m_per_block + rm - 1) / rm) * rm;
    splits = (M + m_per_block - 1) / m_per_block;
}

// gwd_conv_wgrad_batch: weight gradients on the two hot tile shapes (128 x 128, 64 x 64; every gather form) are collected here instead
// of being launched one by one; a full group, and flush() at the end of the batch, runs as ONE igemm_wgrad_group_kernel launch: the
// layers of a group fill the chip together, so each needs fewer, longer workgroups (fewer prologues and atomic flushes), and one
// layer's flush runs beside another's reduction.
struct WgradCollector {
    static constexpr int NT_ = 3;
    WgradGroup g[NT_][3];                                 // [128 x 128 | 64 x 64 | 32 x 128][gather form: 0 general, 1 same-size stride 1, 2 plain GEMM]
    int total[NT_][3];
    hipStream_t s;
    explicit WgradCollector(hipStream_t st) : s(st) {
        for (int t = 0; t < NT_; ++t)
            for (int f = 0; f < 3; ++f) g[t][f].n = total[t][f] = 0;
    }
    void launch(int t, int f) {
        WgradGroup &gr = g[t][f];
        if (!gr.n) return;
        const int tot = total[t][f];
        if (t == 0) {
            if (f == 2) igemm_wgrad_group_kernel<128, 128, 2, 2, 3, 2><<<tot, 256, 0, s>>>(gr);
            else if (f == 1) igemm_wgrad_group_kernel<128, 128, 2, 2, 3, 1><<<tot, 256, 0, s>>>(gr);
            else igemm_wgrad_group_kernel<128, 128, 2, 2, 3, 0><<<tot, 256, 0, s>>>(gr);
        } else if (t == 1) {
            if (f == 2) igemm_wgrad_group_kernel<64, 64, 2, 2, 4, 2><<<tot, 256, 0, s>>>(gr);
            else if (f == 1) igemm_wgrad_group_kernel<64, 64, 2, 2, 4, 1><<<tot, 256, 0, s>>>(gr);
            else igemm_wgrad_group_kernel<64, 64, 2, 2, 4, 0><<<tot, 256, 0, s>>>(gr);
        } else {
            if (f == 2) igemm_wgrad_group_kernel<32, 128, 1, 4, 4, 2><<<tot, 256, 0, s>>>(gr);
            else if (f == 1) igemm_wgrad_group_kernel<32, 128, 1, 4, 4, 1><<<tot, 256, 0, s>>>(gr);
            else igemm_wgrad_group_kernel<32, 128, 1, 4, 4, 0><<<tot, 256, 0, s>>>(gr);
        }
        gr.n = total[t][f] = 0;
    }
    bool take(int bn, int bk, int fast, const gwd_conv_desc &d, float *dw, int m_per_block, int blocks) {
        if (fast < 0 || fast > 2 || !((bn == 128 && bk == 128) || (bn == 64 && bk == 64) || (bn == 32 && bk == 128))) return false;
        const int t = bn == 128 ? 0 : (bn == 64 ? 1 : 2);
        if (g[t][fast].n >= WG_GROUP) launch(t, fast);
        WgradGroup &gr = g[t][fast];
        const int i = gr.n++;
        gr.d[i] = d;
        gr.dw[i] = dw;
        gr.m_per_block[i] = m_per_block;
        gr.block0[i] = total[t][fast];
        gr.blocks[i] = blocks;
        total[t][fast] += (blocks + 7) & ~7;
        return true;
    }
    void flush() {
        for (int t = 0; t < NT_; ++t)
            for (int f = 0; f < 3; ++f) launch(t, f);
    }
};

// 1 = launched wgrad_taps_kernel
static int launch_wgrad_taps(const gwd_conv_desc *d, float *dw, hipStream_t s) {
    if (d->dtype != GWD_BF16 || !d->zero_page) return 0;
    if (d->gather != GWD_GATHER_CONV || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->Ho != d->Hi || d->Wo != d->Wi) return 0;
    // measured against igemm_wgrad_dma_kernel<160,128>, same box (tools/convbench.py): 800 -> 320 at 8x120x160 1.17 -> 0.81 ms, 160 -> 160
    // 0.133 -> 0.114 ms; 80 -> 160 0.081 -> 0.085 and 160 -> 160 on a pooled 8x60x80 map 0.057 -> 0.070 (the nine-tap flush of a
    // workgroup - 184 KB of atomics - needs enough steps to pay for itself): those stay where they were
    if ((d->Cout % 160) || (d->Cin % 8) || d->Cin < 160) return 0;
    const long M = (long)d->B * d->Ho * d->Wo;
    if (M < 131072) return 0;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    static bool attr = false;
    constexpr int LDS = WT_STAGES * WT_STAGE;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)wgrad_taps_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr = true;
    }
    const int tiles = ((d->Cin + 31) / 32) * (d->Cout / 160);
    const int total = d->B * d->Ho * ((d->Wo + 31) / 32);
    int splits = cus / tiles;
    if (splits < 1) splits = 1;
    int cps = (total + splits - 1) / splits;
    if (cps < 8) cps = 8;
    splits = (total + cps - 1) / cps;
    const int wgs = tiles * splits;
    wgrad_taps_kernel<<<wgs, 64 * WT_W, LDS, s>>>(*d, dw, cps, wgs);
    return 1;
}

template <typename T>
int launch_wgrad(const gwd_conv_desc *d, float *dw, hipStream_t s, WgradCollector *coll = nullptr) {
    constexpr int RM = Cfg<T>::BK;
    const int M = d->B * d->Ho * d->Wo, N = d->Cout, K = d->KH * d->KW * d->Cin;
    int splits, m_per_block;
    if constexpr (sizeof(T) == 2) {
        if (dma_enabled() && d->zero_page && (d->Cin % 8) == 0 && (N % 8) == 0) {
#define WG_LAUNCH(BN_, BK_, WN_, WK_, ST_)                                                                   \
    {                                                                                                        \
        const int tiles = ((N + BN_ - 1) / BN_) * ((K + BK_ - 1) / BK_);                                     \
        const int lds = ST_ * 32 * (BN_ + BK_) * 2;                                                          \
        int per_cu = (160 * 1024) / lds;                  /* LDS-limited; the >= 128-wide tiles hold ~200 VGPRs */      \
        if (BN_ * BK_ > 128 * 128 && per_cu > 2) per_cu = 2;   /* -> 2 waves per SIMD = 2 workgroups per CU */          \
        if (BN_ * BK_ == 128 * 128 && per_cu > wgrad_per_cu_128()) per_cu = wgrad_per_cu_128();  /* 138 registers: 3 fit */      \
        /* plain GEMMs: a step costs ~0.45 us, one split's flush tiles * BN * BK * 4 bytes at a NOMINAL 0.25 TB/s: the atomics themselves run */  \
        /* at ~1.3 TB/s, but these layers run grouped (WgradCollector), the group fills the chip, and fewer, longer workgroups per layer */   \
        /* pay fewer prologues and flushes - whole step, same box: rate 4e12 35.15 ms | 1.3e12 34.49 | 0.5e12 34.12 | 0.25e12 34.05 | 0.12e12 34.04; */ \
        /* the same balance for the 3x3 layers (not grouped): +0.1 ... +0.4 ms, not applied */ \
        const double bal = fast == 2 ? 0.45e-6 * 0.25e12 / ((double)tiles * BN_ * BK_ * 4) : (coll ? GWD_WG_BAL1 * 0.6e-6 / ((double)tiles * BN_ * BK_ * 4) : 0.0);                         \
        wgrad_split(M, tiles, 32, splits, m_per_block, 256 * (per_cu > 4 ? 4 : per_cu), bal);                \
        dim3 grid((unsigned)tiles * splits);                                                                 \
        if (coll && coll->take(BN_, BK_, fast, *d, dw, m_per_block, tiles * splits)) { /* runs at flush() */ }             \
        else if (fast == 2) igemm_wgrad_dma_kernel<BN_, BK_, WN_, WK_, ST_, 2><<<grid, 256, 0, s>>>(*d, dw, m_per_block);      \
        else if (fast == 1) igemm_wgrad_dma_kernel<BN_, BK_, WN_, WK_, ST_, 1><<<grid, 256, 0, s>>>(*d, dw, m_per_block); \
        else igemm_wgrad_dma_kernel<BN_, BK_, WN_, WK_, ST_, 0><<<grid, 256, 0, s>>>(*d, dw, m_per_block);   \
    }
            if (launch_wgrad_taps(d, dw, s)) {
                GWD_CHECK_LAUNCH();
                return 0;
            }
            int fast = 0;
            if (d->gather == GWD_GATHER_CONV && d->stride == 1) {
                if (d->KH == 1 && d->KW == 1 && d->pad == 0) fast = 2;
                else if (d->Ho == d->Hi && d->Wo == d->Wi && d->Wo >= 11) fast = 1;
            }
            // tile shapes measured and dropped: 160 x 256 / 128 x 256 (one workgroup per CU: -1 ms per step in all), the 3-stage ring for
            // 160 x 128 (+3-4 %), 64 x 64 for the narrow layers
            if (N % 160 == 0 && K >= 128) WG_LAUNCH(160, 128, 1, 4, 4)      // 4 stages = 72 KB, still 2 per CU
            else if (N > 64 && K > 64) WG_LAUNCH(128, 128, 2, 2, 3)
            else if (N <= 32 && K >= 128) WG_LAUNCH(32, 128, 1, 4, 4)      // narrow layers: no half-empty 64-row tile
            else WG_LAUNCH(64, 64, 2, 2, 4)
#undef WG_LAUNCH
            GWD_CHECK_LAUNCH();
            return 0;
        }
    }
    const bool big = (N > 64 && K > 64);
    const int bn = big ? 128 : 64, bk = big ? 128 : 64;
    wgrad_split(M, ((N + bn - 1) / bn) * ((K + bk - 1) / bk), RM, splits, m_per_block);
    dim3 grid((K + bk - 1) / bk, (N + bn - 1) / bn, splits);
    if (grid.y > 65535 || grid.z > 65535) return -8;
    if (big)
        igemm_wgrad_kernel<T, 128, 128><<<grid, 256, 0, s>>>(*d, dw, m_per_block);
    else
        igemm_wgrad_kernel<T, 64, 64><<<grid, 256, 0, s>>>(*d, dw, m_per_block);
    GWD_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// GWD_TRACE_CONV=1: one stderr line per conv / weight-gradient call (shape census of a step; tools/convcensus.py)
static bool trace_conv() {
    static int v = -1;
    if (v < 0) v = getenv("GWD_TRACE_CONV") ? 1 : 0;
    return v == 1;
}
static void trace_line(const char *what, const gwd_conv_desc *d) {
    fprintf(stderr, "GWDCONV %s B=%d Hi=%d Wi=%d Cin=%d Ho=%d Wo=%d Cout=%d k=%d s=%d g=%d dt=%d\n", what, d->B, d->Hi, d->Wi, d->Cin, d->Ho,
            d->Wo, d->Cout, d->KH, d->stride, d->gather, d->dtype);
}

// thinconv.hip: streaming kernels for the 1-2 channel heads; return 1 when they took the problem
int gwd_thin_conv_forward(const gwd_conv_desc *d, hipStream_t s);
int gwd_thin_conv_wgrad(const gwd_conv_desc *d, float *dw, hipStream_t s);
// tileconv.hip: halo-tiled 3x3 kernels for 32 / 64-channel layers on large maps; return 1 when they took the problem
int gwd_tile_conv_forward(const gwd_conv_desc *d, hipStream_t s);
int gwd_tile_conv_wgrad(const gwd_conv_desc *d, float *dw, hipStream_t s);

extern "C" int gwd_conv_forward(const gwd_conv_desc *d, void *stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!d->w) return -1;
    if ((int64_t)d->B * d->Ho * d->Wo * d->Cout >= (1LL << 40)) return -7;
    if (!d->ln_mean && (gwd_thin_conv_forward(d, (hipStream_t)stream) || gwd_tile_conv_forward(d, (hipStream_t)stream))) {
        if (trace_conv()) trace_line("fwd", d);
        GWD_CHECK_LAUNCH();
        return 0;
    }
    rc = d->dtype == GWD_BF16 ? launch_fwd<__bf16>(d, (hipStream_t)stream) : launch_fwd<float>(d, (hipStream_t)stream);
    if (rc == 0 && trace_conv()) trace_line("fwd", d);     // one line per LAUNCH (a refused ConvLn request, -4, launched nothing): tools/conv_instep.py
    return rc;
}

extern "C" int gwd_conv_wgrad(const gwd_conv_desc *d, float *dw, void *stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dw) return -1;
    if (trace_conv()) trace_line("wgrad", d);
    if (gwd_thin_conv_wgrad(d, dw, (hipStream_t)stream) || gwd_tile_conv_wgrad(d, dw, (hipStream_t)stream)) {
        GWD_CHECK_LAUNCH();
        return 0;
    }
    return d->dtype == GWD_BF16 ? launch_wgrad<__bf16>(d, dw, (hipStream_t)stream)
                                : launch_wgrad<float>(d, dw, (hipStream_t)stream);
}

extern "C" int gwd_conv_wgrad_batch(const gwd_conv_desc *descs, float *const *dws, int32_t n, void *stream) {
    if (!descs || !dws || n <= 0) return -1;
    for (int i = 0; i < n; ++i) {                         // validate everything before the first launch
        const int rc = check_desc(descs + i);
        if (rc) return rc;
        if (!dws[i]) return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    WgradCollector coll(s);
    for (int i = 0; i < n; ++i) {
        const gwd_conv_desc *d = descs + i;
        if (trace_conv()) trace_line("wgrad", d);
        if (gwd_thin_conv_wgrad(d, dws[i], s) || gwd_tile_conv_wgrad(d, dws[i], s)) continue;
        const int rc = d->dtype == GWD_BF16 ? launch_wgrad<__bf16>(d, dws[i], s, &coll) : launch_wgrad<float>(d, dws[i], s);
        if (rc) return rc;
    }
    coll.flush();
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_weight_prep(const float *w, const float *row_scale, void *w_fwd, void *w_dgrad, int32_t N,
                               int32_t taps, int32_t C, int32_t dtype, void *stream) {
    if (!w || N <= 0 || taps <= 0 || C <= 0) return -1;
    const size_t total = (size_t)N * taps * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (dtype == GWD_BF16)
        weight_prep_kernel<__bf16><<<blocks, 256, 0, (hipStream_t)stream>>>(w, row_scale, (__bf16 *)w_fwd, (__bf16 *)w_dgrad, N, taps, C);
    else if (dtype == GWD_F32)
        weight_prep_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>(w, row_scale, (float *)w_fwd, (float *)w_dgrad, N, taps, C);
    else
        return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_unpad_add_batch(const gwd_unpad_job *jobs, int32_t n_jobs, void *stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > GWD_UNPAD_BATCH) return -1;
    UnpadBatch b;
    int total = 0;
    for (int i = 0; i < n_jobs; ++i) {
        gwd_unpad_job j = jobs[i];
        if (!j.src || !j.dst || j.N <= 0 || j.taps <= 0 || j.G <= 0 || j.Cg <= 0 || j.Cgp < j.Cg) return -1;
        if ((int64_t)j.N * j.taps * j.G * j.Cg >= (1LL << 31)) return -7;
        j.block0 = total;
        total += (j.N * j.taps * j.G * j.Cg + 255) / 256;
        b.j[i] = j;
    }
    b.n = n_jobs;
    unpad_add_batch_kernel<<<total, 256, 0, (hipStream_t)stream>>>(b);
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_weight_prep_batch(const gwd_prep_job *jobs, int32_t n_jobs, int32_t total_blocks, const int32_t *block_job, void *stream) {
    if (!jobs || n_jobs <= 0 || total_blocks <= 0) return -1;
    weight_prep_batch_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>(jobs, n_jobs, block_job);
    GWD_CHECK_LAUNCH();
    return 0;
}
