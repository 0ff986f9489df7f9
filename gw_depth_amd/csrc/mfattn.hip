// 7x7 window attention on the matrix cores (bf16 storage, fp32 accumulate), forward and backward:
//
//   S = scale * Q K^T + rel_pos_bias[head] (+ shift mask: -100 between tokens of different regions);  P = softmax(S);  O = P V
//   (WindowAttention / WindowClassAttention core, /root/reference/src/models/multiscale_transformerr.py:311-328, 538-556, 937-955)
//
// One 64-lane wave owns one (window, head) problem at a time.  N = 49 tokens are padded to 64 = 2 tiles of 32, and every
// product runs on v_mfma_f32_32x32x16_bf16 in the orientation that keeps the softmax lane-local and needs no lane movement:
//
//   S^T[key][query]  = K . Q^T          A = K rows, B = Q rows: the accumulator has the QUERY on the lane, 16 keys of each tile
//                                       in its registers (the other 16 in lane ^ 32) -> max / sum over keys are in-lane
//   O^T[d][query]    = V^T . P^T        B = the S^T accumulator itself, converted to bf16 (guide: "an accumulator tile as the
//                                       next MFMA's operand"; its k order is permuted, the V operand is gathered in the same
//                                       order), A = V gathered by column with ds_read_b64_tr_b16 -> lane = query holds its
//                                       output row, 4 consecutive channels per register quad: 8-byte stores
//   backward:  dP^T = V . dO^T,  dQ^T = K^T . dS^T  in the same way; dK^T = Q^T . dS and dV^T = dO^T . P sum over the QUERY,
//   which sits on the lane: P and dS cross LDS once as bf16 [query][key] images and come back through transposed reads.
//
// Operands travel global -> a wave-private, zero-padded LDS image [64 tokens][16 or 32 channels] once (8/16-byte accesses) and
// are read from there both as row fragments (ds_read_b128) and as column gathers (ds_read_b64_tr_b16).  The rel-pos bias of
// the wave's head lives in registers in accumulator layout for the whole launch, the bias gradient accumulates in registers
// across the windows a wave visits and leaves with one set of fp32 atomics.  The shift mask is one extra MFMA k-step over
// one-hot region codes (+100 where the regions AGREE: softmax is shift-invariant, so this equals -100 where they differ).
#include "common.h"

namespace mfattn {

constexpr int NT = 49;          // tokens per window
constexpr int PK = 72;          // row stride (elements) of the [query][key] bf16 images of P and dS: 144 B, 16-byte multiple

struct Operand {
    const void *p;
    long ws, ts, hs;            // window / token / head strides in elements
};
struct OperandW {
    void *p;
    long ws, ts, hs;
};
// where bias(h, i, j) lives: dense [heads][49][49] (rel == nullptr), or the [n_rel][heads] parameter table behind rel[i*49+j]
struct BiasRef {
    const int *rel;
    int heads, n_rel;
    int grad_hm;                 // table-shaped gradient stored head-major [heads][n_rel] (a wave's flush is one contiguous run)
    __device__ __forceinline__ long at(int h, int e) const { return rel ? (long)rel[e] * heads + h : (long)h * (49 * 49) + e; }
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int HD> struct Geo {
    static constexpr int COLS = HD <= 16 ? 16 : 32;     // channels of the zero-padded LDS image
    static constexpr int RS = COLS + 8;                 // row stride in elements: 48 / 80 bytes (16-byte multiples)
    static constexpr int KS = COLS / 16;                // MFMA k-steps over the head dim
    static constexpr int IMG = 64 * RS;                 // elements per image
};

__device__ __forceinline__ f32x16 mma(const bf16x8 &a, const bf16x8 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// row of accumulator register i in lane half h (C/D layout of the 32x32 MFMA)
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// ---- global -> registers -> LDS image: lane = token, HD channels, zero-filled to COLS.  Split in two so that the next
// window's rows are in flight while the current window is being computed.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;      // plain vector type: HIP's uint4 class keeps loop-carried values in scratch
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int HD> struct Row {
    u32x4 a, b, c, d;            // 8 channels each
};

template <int HD>
__device__ __forceinline__ void fetch(Row<HD> &r, const Operand &op, long w, int head, int lane) {
    const int tok = lane < NT ? lane : NT - 1;           // idle lanes shadow the last token (no divergent loads; put() skips them)
    const __bf16 *src = (const __bf16 *)op.p + w * op.ws + (long)tok * op.ts + (long)head * op.hs;
    const u32x4 z = {0u, 0u, 0u, 0u};
    r.b = z;
    r.c = z;
    r.d = z;
    if constexpr (HD == 4) {
        const u32x2 v = *(const u32x2 *)src;
        r.a = u32x4{v.x, v.y, 0u, 0u};
    } else {
        r.a = *(const u32x4 *)src;
        if constexpr (HD >= 16) r.b = *(const u32x4 *)(src + 8);
        if constexpr (HD >= 32) {
            r.c = *(const u32x4 *)(src + 16);
            r.d = *(const u32x4 *)(src + 24);
        }
    }
}

template <int HD>
__device__ __forceinline__ void put(__bf16 *img, const Row<HD> &r, int lane) {
    using G = Geo<HD>;
    if (lane < NT) {
        __bf16 *dst = img + lane * G::RS;
        *(u32x4 *)dst = r.a;
        *(u32x4 *)(dst + 8) = r.b;                        // zeros for head_dim 4 / 8
        if constexpr (HD >= 32) {
            *(u32x4 *)(dst + 16) = r.c;
            *(u32x4 *)(dst + 24) = r.d;
        }
    }
}

// 8 consecutive channels of token 32*tile + (lane & 31): the A operand (rows = tokens) or the B operand (columns = tokens)
template <int HD>
__device__ __forceinline__ bf16x8 rowfrag(const __bf16 *img, int tile, int s, int lane) {
    using G = Geo<HD>;
    return *(const bf16x8 *)(img + (32 * tile + (lane & 31)) * G::RS + 16 * s + 8 * (lane >> 5));
}

__device__ __forceinline__ bf16x8 tr_pair(const __bf16 *lo, const __bf16 *hi) {
    union { s16x4 h[2]; bf16x8 v; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)lo);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)hi);
    return u.v;
}

// Column gather for a product whose OTHER operand is an accumulator tile (k order permuted): lane (r = lane & 31, h) gets
// img[row0 + 16 s + 8 (j >> 2) + 4 h + (j & 3)][col0 + r], j = 0..7.  ds_read_b64_tr_b16: lane 4q+p of a 16-lane group g
// supplies the address of block row q, columns 4p..4p+3, and receives column (lane & 15) of the four rows.
// wide: the image has 32 valid columns from col0 (lane r reads column r); otherwise 16 (lanes r >= 16 duplicate r - 16, their
// output rows / columns are never used).
__device__ __forceinline__ bf16x8 gather_perm(const __bf16 *img, int rs, int row0, int col0, int s, bool wide, int lane) {
    const int g = lane >> 4, t = lane & 15, q = t >> 2, p = t & 3;
    const int row = row0 + 16 * s + 4 * (g >> 1) + q;
    const int col = col0 + (wide ? 16 * (g & 1) : 0) + 4 * p;
    const __bf16 *a = img + row * rs + col;
    return tr_pair(a, a + 8 * rs);
}

// Column gather in natural k order: lane (r, h) gets img[row0 + 16 s + 8 h + j][col0 + r], j = 0..7
__device__ __forceinline__ bf16x8 gather_nat(const __bf16 *img, int rs, int row0, int col0, int s, bool wide, int lane) {
    const int g = lane >> 4, t = lane & 15, q = t >> 2, p = t & 3;
    const int row = row0 + 16 * s + 8 * (g >> 1) + q;
    const int col = col0 + (wide ? 16 * (g & 1) : 0) + 4 * p;
    const __bf16 *a = img + row * rs + col;
    return tr_pair(a, a + 4 * rs);
}

// registers 8s .. 8s+7 of an accumulator tile as the bf16 operand of k-step s
__device__ __forceinline__ bf16x8 accfrag(const f32x16 &x, int s) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (__bf16)x[8 * s + j];
    return f;
}

// one-hot region code of token 32*tile + r, 10.0 at k = region id: (10 * 10 = 100 where two tokens share the region)
__device__ __forceinline__ bf16x8 region_frag(const int *__restrict__ region_w, int tile, int lane) {
    const int tok = 32 * tile + (lane & 31);
    const int rid = tok < NT ? region_w[tok] : -1;
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (rid == 8 * (lane >> 5) + j) ? (__bf16)10.0f : (__bf16)0.0f;
    return f;
}

__device__ __forceinline__ void lds_settle() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // compiler-level ordering of the LDS writes before the reads below
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): this wave's LDS writes are visible to all of its lanes
    __builtin_amdgcn_wave_barrier();
}

// the rel-pos bias of one head in S^T accumulator layout: element i of tile (kt, qt) in lane (c, h) is
// bias[query 32 qt + c][key 32 kt + acc_row(i, h)]; padded keys get -1e30 (softmax weight 0), padded queries 0.
// All 64 loads are unconditional (clamped addresses) so that they issue back to back - a load under a lane-dependent
// condition waits for its predecessor: 64 serial L2 round trips were 80 us of fixed cost per wave.
// With a relative-position TABLE ([n_rel][heads], the parameter itself) the head's n_rel entries are staged in wave-private LDS
// first (three strided loads per lane) and the 64 per-lane values come from there: gathered straight from the table they were 64
// scattered 4-byte loads per lane, each from its own 64-byte segment - 13-40 us of every forward AND backward launch (bisected with
// the bias load compiled out: profiles/r03_ablations.txt 14).  tbl: >= 256 floats of the wave's own LDS, free at this point.
__device__ __forceinline__ void load_bias(f32x16 (&b)[2][2], const float *__restrict__ bias, const BiasRef &br, int head, int lane, float *tbl) {
    const int c = lane & 31, h = lane >> 5;
    float raw[2][2][16];
    if (br.rel) {
        for (int e = lane; e < br.n_rel; e += 64) tbl[e] = bias[(long)e * br.heads + head];
        int ridx[2][2][16];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = 32 * kt + acc_row(i, h), qry = 32 * qt + c;
                    ridx[kt][qt][i] = br.rel[(qry < NT ? qry : NT - 1) * NT + (key < NT ? key : NT - 1)];
                }
        lds_settle();
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int i = 0; i < 16; ++i) raw[kt][qt][i] = tbl[ridx[kt][qt][i]];
        lds_settle();
    } else {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = 32 * kt + acc_row(i, h), qry = 32 * qt + c;
                    raw[kt][qt][i] = bias[br.at(head, (qry < NT ? qry : NT - 1) * NT + (key < NT ? key : NT - 1))];
                }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = 32 * kt + acc_row(i, h), qry = 32 * qt + c;
                b[kt][qt][i] = key >= NT ? -1e30f : (qry < NT ? raw[kt][qt][i] : 0.f);
            }
}

// O^T / dQ^T / dK^T / dV^T accumulator (rows = channels, columns = tokens): lane c stores its token's channels, 4 per register quad
template <int HD>
__device__ __forceinline__ void store_t(const OperandW &op, long w, int head, int tile, const f32x16 &acc, float mul, int lane) {
    const int c = lane & 31, h = lane >> 5, tok = 32 * tile + c;
    if (tok >= NT) return;
    __bf16 *dst = (__bf16 *)op.p + w * op.ws + (long)tok * op.ts + (long)head * op.hs;
    if constexpr (HD == 4) {
        if (h == 0) {
            union { uint2 u; __bf16 e[4]; } v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v.e[j] = (__bf16)(acc[j] * mul);
            *(uint2 *)dst = v.u;
        }
    } else {
#pragma unroll
        for (int g = 0; g < HD / 8; ++g) {
            union { uint2 u; __bf16 e[4]; } v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v.e[j] = (__bf16)(acc[4 * g + j] * mul);
            *(uint2 *)(dst + 8 * g + 4 * h) = v.u;
        }
    }
}


// ------------------------------------------------------------------------------------------------ forward
template <int HD, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void win_fwd_kernel(Operand q, Operand k, Operand v, OperandW o,
                                                             const float *__restrict__ bias, BiasRef br, const int *__restrict__ region,
                                                             long n_windows, int windows_per_image, float scale) {
    using G = Geo<HD>;
    extern __shared__ __attribute__((aligned(16))) __bf16 smem_f[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int head = blockIdx.y * WAVES + wave;
    __bf16 *qi = smem_f + wave * 3 * G::IMG, *ki = qi + G::IMG, *vi = ki + G::IMG;
    f32x16 bs[2][2];
    load_bias(bs, bias, br, head, lane, (float *)qi);     // the images are its scratch: before the zero fill
    for (int e = lane * 8; e < 3 * G::IMG; e += 512) *(uint4 *)(qi + e) = make_uint4(0, 0, 0, 0);     // padding rows stay zero
    lds_settle();
    constexpr bool WIDE = G::COLS == 32;
    Row<HD> rq, rk, rv;
    if ((long)blockIdx.x < n_windows) {
        fetch<HD>(rq, q, blockIdx.x, head, lane);
        fetch<HD>(rk, k, blockIdx.x, head, lane);
        fetch<HD>(rv, v, blockIdx.x, head, lane);
    }
    for (long w = blockIdx.x; w < n_windows; w += gridDim.x) {
        put<HD>(qi, rq, lane);
        put<HD>(ki, rk, lane);
        put<HD>(vi, rv, lane);
        if (w + gridDim.x < n_windows) {               // the next window's rows travel while this one is computed
            fetch<HD>(rq, q, w + gridDim.x, head, lane);
            fetch<HD>(rk, k, w + gridDim.x, head, lane);
            fetch<HD>(rv, v, w + gridDim.x, head, lane);
        }
        bf16x8 rf[2];
        if (region) {
            const int *rw = region + (w % windows_per_image) * NT;
            rf[0] = region_frag(rw, 0, lane);
            rf[1] = region_frag(rw, 1, lane);
        }
        lds_settle();
        bf16x8 kf[2][G::KS], vf[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int s = 0; s < G::KS; ++s) kf[kt][s] = rowfrag<HD>(ki, kt, s, lane);
#pragma unroll
            for (int s = 0; s < 2; ++s) vf[kt][s] = gather_perm(vi, G::RS, 32 * kt, 0, s, WIDE, lane);
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            bf16x8 qf[G::KS];
#pragma unroll
            for (int s = 0; s < G::KS; ++s) qf[s] = rowfrag<HD>(qi, qt, s, lane);
            f32x16 sc[2];
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x16 a = zero16();
#pragma unroll
                for (int s = 0; s < G::KS; ++s) a = mma(kf[kt][s], qf[s], a);
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = a[i] * scale + bs[kt][qt][i];
                if (region) a = mma(rf[kt], rf[qt], a);
#pragma unroll
                for (int i = 0; i < 16; ++i) m = fmaxf(m, a[i]);
                sc[kt] = a;
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = __expf(sc[kt][i] - m);
                    sc[kt][i] = p;
                    l += p;
                }
            l += __shfl_xor(l, 32, 64);
            f32x16 ot = zero16();
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s) ot = mma(vf[kt][s], accfrag(sc[kt], s), ot);
            store_t<HD>(o, w, head, qt, ot, 1.0f / l, lane);
        }
        lds_settle();        // every lane is done reading the images before the next window overwrites them
    }
}

// ------------------------------------------------------------------------------------------------ backward
template <int HD, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void win_bwd_kernel(Operand q, Operand k, Operand v, Operand go, OperandW gq, OperandW gk,
                                                             OperandW gv, const float *__restrict__ bias, float *__restrict__ dbias,
                                                             BiasRef br, const int *__restrict__ region, long n_windows,
                                                             int windows_per_image, float scale) {
    using G = Geo<HD>;
    extern __shared__ __attribute__((aligned(16))) __bf16 smem_b[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int head = blockIdx.y * WAVES + wave;
    constexpr int PER_WAVE = 4 * G::IMG + 2 * 64 * PK;
    __bf16 *qi = smem_b + wave * PER_WAVE, *ki = qi + G::IMG, *vi = ki + G::IMG, *oi = vi + G::IMG;
    __bf16 *pi = oi + G::IMG, *si = pi + 64 * PK;       // P and dS as [query][key] images
    f32x16 bs[2][2], db[2][2];
    load_bias(bs, bias, br, head, lane, (float *)qi);
    for (int e = lane * 8; e < PER_WAVE; e += 512) *(uint4 *)(qi + e) = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) db[a][b] = zero16();
    lds_settle();
    constexpr bool WIDE = G::COLS == 32;
    const int c = lane & 31, h = lane >> 5;
    constexpr bool PREFETCH = HD <= 16;                  // head_dim 32: 64 more registers, and one window per wave anyway
    Row<HD> rq, rk, rv, ro;
    if (PREFETCH && (long)blockIdx.x < n_windows) {
        fetch<HD>(rq, q, blockIdx.x, head, lane);
        fetch<HD>(rk, k, blockIdx.x, head, lane);
        fetch<HD>(rv, v, blockIdx.x, head, lane);
        fetch<HD>(ro, go, blockIdx.x, head, lane);
    }
    for (long w = blockIdx.x; w < n_windows; w += gridDim.x) {
        if (!PREFETCH) {
            fetch<HD>(rq, q, w, head, lane);
            fetch<HD>(rk, k, w, head, lane);
            fetch<HD>(rv, v, w, head, lane);
            fetch<HD>(ro, go, w, head, lane);
        }
        put<HD>(qi, rq, lane);
        put<HD>(ki, rk, lane);
        put<HD>(vi, rv, lane);
        put<HD>(oi, ro, lane);
        if (PREFETCH && w + gridDim.x < n_windows) {
            fetch<HD>(rq, q, w + gridDim.x, head, lane);
            fetch<HD>(rk, k, w + gridDim.x, head, lane);
            fetch<HD>(rv, v, w + gridDim.x, head, lane);
            fetch<HD>(ro, go, w + gridDim.x, head, lane);
        }
        bf16x8 rf[2];
        if (region) {
            const int *rw = region + (w % windows_per_image) * NT;
            rf[0] = region_frag(rw, 0, lane);
            rf[1] = region_frag(rw, 1, lane);
        }
        lds_settle();
        // ---- pass 1, query on the lane: P, dP, dS, dQ; P and dS written to their [query][key] images
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            bf16x8 qf[G::KS], of[G::KS];
#pragma unroll
            for (int s = 0; s < G::KS; ++s) {
                qf[s] = rowfrag<HD>(qi, qt, s, lane);
                of[s] = rowfrag<HD>(oi, qt, s, lane);
            }
            f32x16 sc[2], dp[2];
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x16 a = zero16(), d = zero16();
#pragma unroll
                for (int s = 0; s < G::KS; ++s) {
                    a = mma(rowfrag<HD>(ki, kt, s, lane), qf[s], a);
                    d = mma(rowfrag<HD>(vi, kt, s, lane), of[s], d);      // dP^T[key][query] = V[key] . dO[query]
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = a[i] * scale + bs[kt][qt][i];
                if (region) a = mma(rf[kt], rf[qt], a);
#pragma unroll
                for (int i = 0; i < 16; ++i) m = fmaxf(m, a[i]);
                sc[kt] = a;
                dp[kt] = d;
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = __expf(sc[kt][i] - m);
                    sc[kt][i] = p;
                    l += p;
                }
            l += __shfl_xor(l, 32, 64);
            const float inv = 1.0f / l;
            float delta = 0.f;                                            // sum_k P dP  (= dO . O)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    sc[kt][i] *= inv;
                    delta += sc[kt][i] * dp[kt][i];
                }
            delta += __shfl_xor(delta, 32, 64);
            f32x16 dq = zero16();
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float ds = sc[kt][i] * (dp[kt][i] - delta);
                    dp[kt][i] = ds;
                    db[kt][qt][i] += ds;
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) dq = mma(gather_perm(ki, G::RS, 32 * kt, 0, s, WIDE, lane), accfrag(dp[kt], s), dq);
                // registers 4g..4g+3 are keys 32 kt + 8 g + 4 h + 0..3 of query 32 qt + c: 8-byte writes into the [query][key] images
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    union { uint2 u; __bf16 e[4]; } pv, sv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        pv.e[j] = (__bf16)sc[kt][4 * g + j];
                        sv.e[j] = (__bf16)dp[kt][4 * g + j];
                    }
                    const int off = (32 * qt + c) * PK + 32 * kt + 8 * g + 4 * h;
                    *(uint2 *)(pi + off) = pv.u;
                    *(uint2 *)(si + off) = sv.u;
                }
            }
            store_t<HD>(gq, w, head, qt, dq, scale, lane);
        }
        lds_settle();
        // ---- pass 2, key on the lane: dK^T = Q^T . dS, dV^T = dO^T . P, summed over the 64 (padded) queries in 4 k-steps
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 dk = zero16(), dv = zero16();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                dk = mma(gather_nat(qi, G::RS, 0, 0, s, WIDE, lane), gather_nat(si, PK, 0, 32 * kt, s, true, lane), dk);
                dv = mma(gather_nat(oi, G::RS, 0, 0, s, WIDE, lane), gather_nat(pi, PK, 0, 32 * kt, s, true, lane), dv);
            }
            store_t<HD>(gk, w, head, kt, dk, scale, lane);
            store_t<HD>(gv, w, head, kt, dv, 1.0f, lane);
        }
        lds_settle();
    }
    if (dbias && !br.rel) {
        float *dh = dbias + (long)head * NT * NT;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = 32 * kt + acc_row(i, h), qry = 32 * qt + c;
                    if (key < NT && qry < NT) unsafeAtomicAdd(dh + qry * NT + key, db[kt][qt][i]);
                }
    } else if (dbias) {
        // table-shaped gradient: up to 49 (query, key) pairs share one table row - fold the wave's 2 401 values into its
        // n_rel rows in LDS first (the P image is free now), then one atomic per row and wave
        float *acc = (float *)pi;
        for (int e = lane; e < 256; e += 64) acc[e] = 0.f;
        lds_settle();
        // the 64 table-row indices of this lane first, unconditional (clamped addresses) and back to back: under the lane-dependent
        // bound every one of them waits for its predecessor
        int ridx[2][2][16];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = 32 * kt + acc_row(i, h), qry = 32 * qt + c;
                    ridx[kt][qt][i] = br.rel[(qry < NT ? qry : NT - 1) * NT + (key < NT ? key : NT - 1)];
                }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = 32 * kt + acc_row(i, h), qry = 32 * qt + c;
                    if (key < NT && qry < NT) atomicAdd(acc + ridx[kt][qt][i], db[kt][qt][i]);
                }
        lds_settle();
        // Table-shaped ([n_rel][heads]) the wave's n_rel atomics land 4 bytes each in n_rel different 64-byte segments, from every wave
        // of the launch: the slow shape of MI355X_MICROARCH.md 'Global float atomics' (x17) - 41-59 us of EVERY backward launch.
        // Head-major ([heads][n_rel]: the caller's scratch, added to the table gradient afterwards) it is one contiguous run per wave.
        if (br.grad_hm) {
            for (int e = lane; e < br.n_rel; e += 64) unsafeAtomicAdd(dbias + (long)head * br.n_rel + e, acc[e]);
        } else {
            for (int e = lane; e < br.n_rel; e += 64) unsafeAtomicAdd(dbias + (long)e * br.heads + head, acc[e]);
        }
    }
}

struct Args {
    Operand q, k, v, go;
    OperandW o, gq, gk, gv;
};

template <int HD>
bool aligned_for(const void *p, long ws, long ts, long hs) {
    constexpr int A = HD == 4 ? 4 : 8;          // 8-byte rows for head_dim 4, 16-byte pieces otherwise; 8-byte stores of 4 channels
    return (uintptr_t)p % (A * 2) == 0 && ws % A == 0 && ts % A == 0 && hs % A == 0;
}

template <int HD>
int launch(bool backward, const Args &a, const float *bias, float *dbias, const BiasRef br, const int *region, long n_windows, int wpi,
           int heads, float scale, hipStream_t s) {
    using G = Geo<HD>;
    bool ok = aligned_for<HD>(a.q.p, a.q.ws, a.q.ts, a.q.hs) && aligned_for<HD>(a.k.p, a.k.ws, a.k.ts, a.k.hs) &&
              aligned_for<HD>(a.v.p, a.v.ws, a.v.ts, a.v.hs);
    if (!backward) ok = ok && aligned_for<HD>(a.o.p, a.o.ws, a.o.ts, a.o.hs);
    else ok = ok && aligned_for<HD>(a.go.p, a.go.ws, a.go.ts, a.go.hs) && aligned_for<HD>(a.gq.p, a.gq.ws, a.gq.ts, a.gq.hs) &&
              aligned_for<HD>(a.gk.p, a.gk.ws, a.gk.ts, a.gk.hs) && aligned_for<HD>(a.gv.p, a.gv.ws, a.gv.ts, a.gv.hs);
    if (!ok) return 1;                          // not taken: the caller keeps the lane-per-row kernel
    // a wave keeps one head's bias (and bias gradient) in registers and walks windows blockIdx.x, + gridDim.x, ...:
    // enough waves to fill the chip, few enough that the 2 401 bias-gradient atomics per wave stay negligible
    long gx = (n_windows + 7) / 8;
    if (gx * heads < 2048) gx = (2048 + heads - 1) / heads;       // ... but never fewer waves than fill the chip twice
    if (gx > n_windows) gx = n_windows;
    if (gx > 256) gx = 256;
    if (gx >= 8) gx &= ~7L;                    // gridDim.x a multiple of 8: the head groups (blockIdx.y) of a window share its XCD / L2
    if (!backward) {
        constexpr int WAVES = 4;
        if (heads % WAVES) return 1;
        const size_t lds = (size_t)WAVES * 3 * G::IMG * 2;
        win_fwd_kernel<HD, WAVES><<<dim3((unsigned)gx, heads / WAVES), 64 * WAVES, lds, s>>>(a.q, a.k, a.v, a.o, bias, br, region, n_windows, wpi, scale);
    } else {
        constexpr int WAVES = HD == 32 ? 1 : 2;
        if (heads % WAVES) return 1;
        const size_t lds = (size_t)WAVES * (4 * G::IMG + 2 * 64 * PK) * 2;
        win_bwd_kernel<HD, WAVES><<<dim3((unsigned)gx, heads / WAVES), 64 * WAVES, lds, s>>>(a.q, a.k, a.v, a.go, a.gq, a.gk, a.gv, bias, dbias,
                                                                                            br, region, n_windows, wpi, scale);
    }
    return 0;
}

}  // namespace mfattn

// 0 = launched, 1 = shape / alignment not covered (caller falls back), < 0 = error.  bf16 only.
int gwd_mfattn_window(bool backward, const gwd_strided *q, const gwd_strided *k, const gwd_strided *v, const gwd_strided *o_or_go,
                      const gwd_strided *gq, const gwd_strided *gk, const gwd_strided *gv, const float *bias, float *dbias,
                      const int32_t *rel_index, int32_t n_rel, const int32_t *region, int64_t n_windows, int32_t wpi, int32_t heads,
                      int32_t head_dim, float scale, int32_t dbias_head_major, hipStream_t s) {
    using namespace mfattn;
    const BiasRef br{rel_index, heads, n_rel, (rel_index && dbias_head_major) ? 1 : 0};
    Args a{};
    a.q = {q->p, q->ws, q->ts, q->hs};
    a.k = {k->p, k->ws, k->ts, k->hs};
    a.v = {v->p, v->ws, v->ts, v->hs};
    if (!backward) {
        a.o = {o_or_go->p, o_or_go->ws, o_or_go->ts, o_or_go->hs};
    } else {
        a.go = {o_or_go->p, o_or_go->ws, o_or_go->ts, o_or_go->hs};
        a.gq = {gq->p, gq->ws, gq->ts, gq->hs};
        a.gk = {gk->p, gk->ws, gk->ts, gk->hs};
        a.gv = {gv->p, gv->ws, gv->ts, gv->hs};
    }
    if (region) {                                  // one-hot codes over 16 k slots: region ids must be 0..15 (they are 0..8)
        // checked on the host side of the ABI by construction (model.shift_regions); nothing to verify on device memory here
    }
    switch (head_dim) {
        case 4: return launch<4>(backward, a, bias, dbias, br, region, n_windows, wpi, heads, scale, s);
        case 8: return launch<8>(backward, a, bias, dbias, br, region, n_windows, wpi, heads, scale, s);
        case 16: return launch<16>(backward, a, bias, dbias, br, region, n_windows, wpi, heads, scale, s);
        case 32: return launch<32>(backward, a, bias, dbias, br, region, n_windows, wpi, heads, scale, s);
        default: return 1;
    }
}

// =====================================================================================================================
// Class-token ("transposed") attention of WindowClassAttention on the matrix cores (multiscale_transformerr.py:560-578):
//   S[r][c] = scale * sum_n q[n][r] k[n][c]  (r < 4 token channels, c < E feature channels, n = the 49 tokens),
//   A = softmax over c,  O[n][r] = sum_c A[r][c] v[n][c].
// S^T[c][r] = K^T . Q (4 k-steps over the tokens; both operands column gathers from [token][channel] LDS images) leaves the
// softmax axis c in the REGISTERS of lane r; O^T[r][n] = A . V^T takes that accumulator as its A operand (k = c, permuted) and V
// rows as B: lane = token holds its 4 outputs -> one 8-byte store.  Backward: dA^T = V^T . dO the same way, dQ like O with K,
// dK^T[c][n] = dS^T . Q^T and dV^T = A^T . dO^T sum over r (4 values, on the lane): dS and A cross LDS as tiny [c][16] bf16 images
// and come back as row fragments; lane = token stores its E gradients in 8-byte pieces.  ~8 MFMAs forward, ~14 backward per
// (window, head) instead of 48-96 six-step DPP wave reductions each way.
namespace tok {
using namespace mfattn;

constexpr int QS = 24, KSTR = 40, AS = 24;          // row strides (elements): q / dO images [64][16+8], k / v images [64][32+8], A / dS [32][16+8]

struct TOp {
    void *p;
    long ws, ts, hs;
};

// k / v rows of E channels (8-byte pieces: E = 12 is 24 bytes) -> image row, zero-filled to 32 columns
template <int E>
__device__ __forceinline__ void put_row_e(__bf16 *img, const __bf16 *src, int lane) {
    if (lane < NT) {
        __bf16 *dst = img + lane * KSTR;
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            uint2 v = make_uint2(0, 0);
            if (c < E) v = *(const uint2 *)(src + c);
            *(uint2 *)(dst + c) = v;
        }
    }
}
// q / dO rows: 4 channels (R = 8: the 4 channels of a second query set behind them - the two class tokens share k and v, and the softmax is per
// query channel, so the pair is ONE problem with 8 query channels: k / v staged once, dk / dv come out summed), zero-filled to 16
template <int R>
__device__ __forceinline__ void put_row_4(__bf16 *img, const __bf16 *src, const __bf16 *src2, int lane) {
    if (lane < NT) {
        __bf16 *dst = img + lane * QS;
        *(uint2 *)dst = *(const uint2 *)src;
        *(uint2 *)(dst + 4) = R == 8 ? *(const uint2 *)src2 : make_uint2(0, 0);
        *(uint4 *)(dst + 8) = make_uint4(0, 0, 0, 0);
    }
}

// X^T[c][r] = sum_n a[n][c] * b[n][r] over the 64 (padded) tokens: A = column gather of the [token][32] image, B = of the [token][16] image
__device__ __forceinline__ f32x16 tokens_product(const __bf16 *img_c, const __bf16 *img_r, int lane) {
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mma(gather_nat(img_c, KSTR, 0, 0, s, true, lane), gather_nat(img_r, QS, 0, 0, s, false, lane), acc);
    return acc;
}

// Z[r][n] = sum_c x[c][r] * img[token n][c] for the 32 tokens of tile nt: A = the accumulator (k = c, permuted), B = image rows
template <int E>
__device__ __forceinline__ f32x16 channels_product(const f32x16 &x, const __bf16 *img, int nt, int lane) {
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < (E > 16 ? 2 : 1); ++s) {
        const __bf16 *row = img + (32 * nt + r) * KSTR + 16 * s + 4 * h;
        union { uint2 u[2]; bf16x8 v; } b;
        b.u[0] = *(const uint2 *)row;
        b.u[1] = *(const uint2 *)(row + 8);
        acc = mma(accfrag(x, s), b.v, acc);
    }
    return acc;
}

// softmax over the rows c < E of an S^T accumulator (lane = column r): in-register + one exchange with lane ^ 32
template <int E>
__device__ __forceinline__ void softmax_rows(f32x16 &x, float scale, int h) {
    float m = -1e30f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        x[i] = acc_row(i, h) < E ? x[i] * scale : -1e30f;
        m = fmaxf(m, x[i]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        x[i] = acc_row(i, h) < E ? __expf(x[i] - m) : 0.f;
        l += x[i];
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] *= inv;
}

__device__ __forceinline__ const __bf16 *tok_row(const TOp &t, long w, int tok, int head) {
    return (const __bf16 *)t.p + w * t.ws + (long)tok * t.ts + (long)head * t.hs;
}

template <int R>
__device__ __forceinline__ void store4(const TOp &o, const TOp &o2, long w, int head, int nt, const f32x16 &z, float mul, int lane) {
    const int tokn = 32 * nt + (lane & 31), h = lane >> 5;
    if ((h == 0 || R == 8) && tokn < NT) {                // rows r = 0..3 are registers 0..3 of the lower half, rows 4..7 of the upper half
        union { uint2 u; __bf16 e[4]; } v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v.e[j] = (__bf16)(z[j] * mul);
        const TOp &t = (R == 8 && h) ? o2 : o;
        *(uint2 *)((__bf16 *)t.p + w * t.ws + (long)tokn * t.ts + (long)head * t.hs) = v.u;
    }
}

template <int E, int WAVES, int R>
__global__ __launch_bounds__(64 * WAVES) void tok_fwd_kernel(TOp q, TOp k, TOp v, TOp o, TOp q2, TOp o2, long n_problems, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem_t[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    constexpr int PER = 64 * QS + 2 * 64 * KSTR;
    __bf16 *qi = smem_t + wave * PER, *ki = qi + 64 * QS, *vi = ki + 64 * KSTR;
    for (int e = lane * 8; e < PER; e += 512) *(uint4 *)(qi + e) = make_uint4(0, 0, 0, 0);
    lds_settle();
    const int tokc = lane < NT ? lane : NT - 1;
    const long vblock = (heads % WAVES) ? (long)blockIdx.x : xcd_grouped_block(blockIdx.x, gridDim.x, heads / WAVES);   // a window's head groups on one XCD
    for (long pb = vblock * WAVES + wave; pb < n_problems; pb += (long)gridDim.x * WAVES) {
        const long w = pb / heads;
        const int head = (int)(pb - w * heads);
        put_row_4<R>(qi, tok_row(q, w, tokc, head), R == 8 ? tok_row(q2, w, tokc, head) : nullptr, lane);
        put_row_e<E>(ki, (const __bf16 *)k.p + w * k.ws + (long)tokc * k.ts + (long)head * k.hs, lane);
        put_row_e<E>(vi, (const __bf16 *)v.p + w * v.ws + (long)tokc * v.ts + (long)head * v.hs, lane);
        lds_settle();
        f32x16 a = tokens_product(ki, qi, lane);           // S^T[c][r]
        softmax_rows<E>(a, scale, h);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) store4<R>(o, o2, w, head, nt, channels_product<E>(a, vi, nt, lane), 1.0f, lane);
        lds_settle();
    }
}

// accumulator X[c][r] (lane = r < R real) -> bf16 image [c][16] (columns R..15 stay zero)
template <int R>
__device__ __forceinline__ void put_cr(__bf16 *img, const f32x16 &x, int lane) {
    const int r = lane & 31, h = lane >> 5;
    if (r < R) {
#pragma unroll
        for (int i = 0; i < 16; ++i) img[acc_row(i, h) * AS + r] = (__bf16)x[i];
    }
}

// D^T[c][n] = sum_r img_cr[c][r] * img_tok[token n][r] for token tile nt (one k-step: r = 0..15, 4 real)
__device__ __forceinline__ f32x16 r_product(const __bf16 *img_cr, const __bf16 *img_tok, int nt, int lane) {
    const int r = lane & 31, h = lane >> 5;
    const bf16x8 a = *(const bf16x8 *)(img_cr + r * AS + 8 * h);
    const bf16x8 b = *(const bf16x8 *)(img_tok + (32 * nt + r) * QS + 8 * h);
    return mma(a, b, zero16());
}

// D^T accumulator (rows = channels c, lane = token) -> E channels of the token's row, 8-byte pieces
template <int E>
__device__ __forceinline__ void store_e(const TOp &o, long w, int head, int nt, const f32x16 &z, float mul, int lane) {
    const int tokn = 32 * nt + (lane & 31), h = lane >> 5;
    if (tokn >= NT) return;
    __bf16 *dst = (__bf16 *)o.p + w * o.ws + (long)tokn * o.ts + (long)head * o.hs;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c0 = 8 * g + 4 * h;
        if (c0 < E) {
            union { uint2 u; __bf16 e[4]; } v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v.e[j] = (__bf16)(z[4 * g + j] * mul);
            *(uint2 *)(dst + c0) = v.u;
        }
    }
}

template <int E, int WAVES, int R>
__global__ __launch_bounds__(64 * WAVES) void tok_bwd_kernel(TOp q, TOp k, TOp v, TOp go, TOp gq, TOp gk, TOp gv, TOp q2, TOp go2, TOp gq2, long n_problems,
                                                            int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem_u[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    constexpr int PER = 2 * 64 * QS + 2 * 64 * KSTR + 2 * 32 * AS;
    __bf16 *qi = smem_u + wave * PER, *oi = qi + 64 * QS, *ki = oi + 64 * QS, *vi = ki + 64 * KSTR, *ai = vi + 64 * KSTR, *si = ai + 32 * AS;
    for (int e = lane * 8; e < PER; e += 512) *(uint4 *)(qi + e) = make_uint4(0, 0, 0, 0);
    lds_settle();
    const int tokc = lane < NT ? lane : NT - 1;
    const long vblock = (heads % WAVES) ? (long)blockIdx.x : xcd_grouped_block(blockIdx.x, gridDim.x, heads / WAVES);   // a window's head groups on one XCD
    for (long pb = vblock * WAVES + wave; pb < n_problems; pb += (long)gridDim.x * WAVES) {
        const long w = pb / heads;
        const int head = (int)(pb - w * heads);
        put_row_4<R>(qi, tok_row(q, w, tokc, head), R == 8 ? tok_row(q2, w, tokc, head) : nullptr, lane);
        put_row_4<R>(oi, tok_row(go, w, tokc, head), R == 8 ? tok_row(go2, w, tokc, head) : nullptr, lane);
        put_row_e<E>(ki, (const __bf16 *)k.p + w * k.ws + (long)tokc * k.ts + (long)head * k.hs, lane);
        put_row_e<E>(vi, (const __bf16 *)v.p + w * v.ws + (long)tokc * v.ts + (long)head * v.hs, lane);
        lds_settle();
        f32x16 a = tokens_product(ki, qi, lane);           // S^T[c][r]
        softmax_rows<E>(a, scale, h);
        f32x16 da = tokens_product(vi, oi, lane);          // dA^T[c][r] = sum_n v[n][c] dO[n][r]
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) dot += a[i] * da[i];
        dot += __shfl_xor(dot, 32, 64);
#pragma unroll
        for (int i = 0; i < 16; ++i) da[i] = a[i] * (da[i] - dot);      // dS^T (unscaled; the scale rides on the stores)
        put_cr<R>(ai, a, lane);
        put_cr<R>(si, da, lane);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) store4<R>(gq, gq2, w, head, nt, channels_product<E>(da, ki, nt, lane), scale, lane);     // dq[n][r] = scale sum_c dS[r][c] k[n][c]
        lds_settle();
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            store_e<E>(gk, w, head, nt, r_product(si, qi, nt, lane), scale, lane);       // dk[n][c] = scale sum_r dS[r][c] q[n][r]
            store_e<E>(gv, w, head, nt, r_product(ai, oi, nt, lane), 1.0f, lane);        // dv[n][c] = sum_r A[r][c] dO[n][r]
        }
        lds_settle();
    }
}

// operands: forward q k v o [q2 o2], backward q k v go gq gk gv [q2 go2 gq2]
template <int E, int R>
int run(bool bwd, const gwd_strided *const *s, long n_problems, int heads, float scale, hipStream_t st) {
    for (int i = 0; i < (bwd ? 7 : 4) + (R == 8 ? (bwd ? 3 : 2) : 0); ++i)                 // 8-byte rows everywhere
        if ((uintptr_t)s[i]->p % 8 || s[i]->ws % 4 || s[i]->ts % 4 || s[i]->hs % 4) return 1;
    auto mk = [](const gwd_strided *x) { return TOp{x->p, x->ws, x->ts, x->hs}; };
    if (!bwd) {
        constexpr int WAVES = 4;
        long bx = (n_problems + WAVES - 1) / WAVES;
        if (bx > 1024) bx = 1024;
        const size_t lds = (size_t)WAVES * (64 * QS + 2 * 64 * KSTR) * 2;
        tok_fwd_kernel<E, WAVES, R><<<(unsigned)bx, 64 * WAVES, lds, st>>>(mk(s[0]), mk(s[1]), mk(s[2]), mk(s[3]), mk(s[R == 8 ? 4 : 0]), mk(s[R == 8 ? 5 : 3]),
                                                                           n_problems, heads, scale);
    } else {
        constexpr int WAVES = 2;
        long bx = (n_problems + WAVES - 1) / WAVES;
        if (bx > 2048) bx = 2048;
        const size_t lds = (size_t)WAVES * (2 * 64 * QS + 2 * 64 * KSTR + 2 * 32 * AS) * 2;
        tok_bwd_kernel<E, WAVES, R><<<(unsigned)bx, 64 * WAVES, lds, st>>>(mk(s[0]), mk(s[1]), mk(s[2]), mk(s[3]), mk(s[4]), mk(s[5]), mk(s[6]), mk(s[R == 8 ? 7 : 0]),
                                                                            mk(s[R == 8 ? 8 : 3]), mk(s[R == 8 ? 9 : 4]), n_problems, heads, scale);
    }
    return 0;
}

}  // namespace tok

// 0 = launched, 1 = not covered (the caller keeps the lane-per-token kernels); bf16 only
int gwd_mfattn_token(bool backward, bool pair, const gwd_strided *const *ops, long n_problems, int heads, int e, float scale, hipStream_t s) {
    // measured per launch (53 k problems at E = 12, 13.8 k at 16, 3.8 k at 24): forward 107 / 39 / 23 us on the lane-per-token kernels ->
    // 90 / 20 / ~12 here; backward 224 / 77 / 43 -> 269 / 52 / 20: at E = 12 the 48 wave reductions of the VALU form are cheaper than this
    // kernel's 8-24-byte-per-lane row traffic, so that one case stays there
    // (with a window's head groups on one XCD - xcd_grouped_block - the E = 12 backward is the faster one here as well: same-box
    //  A/B -0.1 ... -0.25 ms per step)
    switch (e) {
        case 12: return pair ? tok::run<12, 8>(backward, ops, n_problems, heads, scale, s) : tok::run<12, 4>(backward, ops, n_problems, heads, scale, s);
        case 16: return pair ? tok::run<16, 8>(backward, ops, n_problems, heads, scale, s) : tok::run<16, 4>(backward, ops, n_problems, heads, scale, s);
        case 24: return pair ? tok::run<24, 8>(backward, ops, n_problems, heads, scale, s) : tok::run<24, 4>(backward, ops, n_problems, heads, scale, s);
        default: return 1;
    }
}

// =====================================================================================================================
// DETR multi-head attention core (head_dim 32), forward and backward, flash style: no L x S matrix ever reaches memory.
//   O = dropout(softmax(scale Q K^T + key-padding mask)) V, heads merged   (/root/reference/src/models/multi_head_attention.py:329-375)
// One wave per (batch, head, 32-query tile) walks the key tiles with an online softmax (query on the lane, so the running
// maximum / sum and the rescale of O^T are lane-local); the forward saves LSE = m + log l per query.  The backward is two
// roles in one launch: role A (query tile on the lane) recomputes P from LSE and produces dQ, role B (key tile on the lane)
// produces dK and dV - each output row is written by exactly one wave, no atomics, no transposes through memory.
// Key-padding mask and out-of-range keys enter as one more MFMA k-step (A: -1e30 in slot 0 of a masked key, B: 1 in slot 0).
// Dropout multipliers (0 or 1/(1-p), bf16, (B,H,L,S)) are read in place.
namespace mha {
using namespace mfattn;

constexpr int D = 32, RSV = 40;       // head dim; row stride of a 32 x 32 tile image (80 bytes)
constexpr int TILE = 32 * RSV;

struct Tok {
    const void *p;
    long ts;                          // token stride in elements (dense batches: batch stride = tokens * ts)
};
struct TokW {
    void *p;
    long ts;
};

__device__ __forceinline__ bf16x8 zero_frag() {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (__bf16)0.0f;
    return f;
}

// 8 consecutive channels (16 s + 8 h ..) of token tok0 + (lane & 31) of one (batch, head) slab, straight from global memory
__device__ __forceinline__ bf16x8 rowfrag_g(const __bf16 *slab, long ts, int ntok, int tok0, int s, int lane) {
    const int tok = tok0 + (lane & 31);
    if (tok >= ntok) return zero_frag();
    return *(const bf16x8 *)(slab + (long)tok * ts + 16 * s + 8 * (lane >> 5));
}

// 32 tokens x 32 channels -> wave-private LDS image (rows of out-of-range tokens zero)
__device__ __forceinline__ void stage_tile(__bf16 *img, const __bf16 *slab, long ts, int ntok, int tok0, int lane) {
    const int r = lane & 31, half = lane >> 5, tok = tok0 + r;
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
    if (tok < ntok) {
        const __bf16 *src = slab + (long)tok * ts + 16 * half;
        a = *(const uint4 *)src;
        b = *(const uint4 *)(src + 8);
    }
    *(uint4 *)(img + r * RSV + 16 * half) = a;
    *(uint4 *)(img + r * RSV + 16 * half + 8) = b;
}

// the two halves of stage_tile: global -> registers (issued one tile ahead of its use), registers -> LDS image
__device__ __forceinline__ void load_tile_regs(uint4 &a, uint4 &b, const __bf16 *slab, long ts, int ntok, int tok0, int lane) {
    const int r = lane & 31, half = lane >> 5, tok = tok0 + r;
    a = make_uint4(0, 0, 0, 0);
    b = a;
    if (tok < ntok) {
        const __bf16 *src = slab + (long)tok * ts + 16 * half;
        a = *(const uint4 *)src;
        b = *(const uint4 *)(src + 8);
    }
}
__device__ __forceinline__ void store_tile_regs(__bf16 *img, const uint4 &a, const uint4 &b, int lane) {
    const int r = lane & 31, half = lane >> 5;
    *(uint4 *)(img + r * RSV + 16 * half) = a;
    *(uint4 *)(img + r * RSV + 16 * half + 8) = b;
}

__device__ __forceinline__ bf16x8 rowfrag_l(const __bf16 *img, int s, int lane) {
    return *(const bf16x8 *)(img + (lane & 31) * RSV + 16 * s + 8 * (lane >> 5));
}

// A operand of the mask k-step for keys on the ROWS: slot 0 of key row r carries -1e30 when the key is padding / out of range
__device__ __forceinline__ bf16x8 keymask_frag(const unsigned char *__restrict__ kpm_b, int S, int key0, int lane) {
    bf16x8 f = zero_frag();
    const int key = key0 + (lane & 31);
    if ((lane >> 5) == 0) {
        const bool masked = key >= S || (kpm_b && kpm_b[key]);
        f[0] = masked ? (__bf16)-1e30f : (__bf16)0.0f;
    }
    return f;
}
// the same in two halves: the mask byte (loaded a tile ahead), then the fragment
__device__ __forceinline__ unsigned keymask_load(const unsigned char *__restrict__ kpm_b, int S, int key0, int lane) {
    const int key = key0 + (lane & 31);
    unsigned raw = 0;                                      // the raw byte: nothing here waits for it
    if ((lane >> 5) == 0 && key < S && kpm_b) raw = kpm_b[key];
    return raw;
}
__device__ __forceinline__ bf16x8 keymask_from(unsigned raw, int S, int key0, int lane) {
    bf16x8 f = zero_frag();
    if ((lane >> 5) == 0) f[0] = (key0 + (lane & 31) >= S || raw) ? (__bf16)-1e30f : (__bf16)0.0f;
    return f;
}
__device__ __forceinline__ bf16x8 one_frag(int lane) {
    bf16x8 f = zero_frag();
    if ((lane >> 5) == 0) f[0] = (__bf16)1.0f;
    return f;
}

// dropout multipliers of query row q for the 16 keys this lane holds of a key tile (registers 4g..4g+3 <-> keys key0 + 8g + 4h + 0..3)
template <bool VEC4>
__device__ __forceinline__ void load_mult_keys(float (&mv)[16], const __bf16 *__restrict__ mrow, bool qok, int S, int key0, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int kk = key0 + 8 * g + 4 * h;
        if constexpr (VEC4) {
            union { uint2 u; __bf16 e[4]; } v;
            v.u = make_uint2(0, 0);
            if (qok && kk < S) v.u = *(const uint2 *)(mrow + kk);
#pragma unroll
            for (int j = 0; j < 4; ++j) mv[4 * g + j] = (float)v.e[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) mv[4 * g + j] = (qok && kk + j < S) ? (float)mrow[kk + j] : 0.f;
        }
    }
}

// the raw multipliers of load_mult_keys, loaded a tile ahead and converted at the use
union MultRaw {
    uint2 u[4];
    __bf16 e[16];
};
template <bool VEC4>
__device__ __forceinline__ void load_mult_raw(MultRaw &r, const __bf16 *__restrict__ mrow, bool qok, int S, int key0, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int kk = key0 + 8 * g + 4 * h;
        if constexpr (VEC4) {
            r.u[g] = make_uint2(0, 0);
            if (qok && kk < S) r.u[g] = *(const uint2 *)(mrow + kk);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) r.e[4 * g + j] = (qok && kk + j < S) ? mrow[kk + j] : (__bf16)0.0f;
        }
    }
}

// O^T-style accumulator (rows = channels, columns = tokens) -> (B, tokens, E) rows: lane = token, 4 channels per register quad
__device__ __forceinline__ void store_rows(__bf16 *slab, long ts, int ntok, int tok0, const f32x16 &acc, float mul, int lane) {
    const int tok = tok0 + (lane & 31), h = lane >> 5;
    if (tok >= ntok) return;
    __bf16 *dst = slab + (long)tok * ts;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        union { uint2 u; __bf16 e[4]; } v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v.e[j] = (__bf16)(acc[4 * g + j] * mul);
        *(uint2 *)(dst + 8 * g + 4 * h) = v.u;
    }
}

template <bool VEC4>
__global__ __launch_bounds__(64) void mha_fwd_kernel(Tok q, Tok k, Tok v, const unsigned char *__restrict__ kpm,
                                                     const __bf16 *__restrict__ mult, TokW o, float *__restrict__ lse,
                                                     int H, int L, int S, float scale) {
    __shared__ __attribute__((aligned(16))) __bf16 vimg[TILE];
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    const int qt = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const __bf16 *qs = (const __bf16 *)q.p + (long)b * L * q.ts + head * D;
    const __bf16 *ks = (const __bf16 *)k.p + (long)b * S * k.ts + head * D;
    const __bf16 *vs = (const __bf16 *)v.p + (long)b * S * v.ts + head * D;
    const unsigned char *kpm_b = kpm ? kpm + (long)b * S : nullptr;
    const int qrow = 32 * qt + c;
    const bool qok = qrow < L;
    const __bf16 *mrow = mult ? mult + (((long)b * H + head) * L + (qok ? qrow : 0)) * S : nullptr;
    bf16x8 qf[2];
    qf[0] = rowfrag_g(qs, q.ts, L, 32 * qt, 0, lane);
    qf[1] = rowfrag_g(qs, q.ts, L, 32 * qt, 1, lane);
    const bf16x8 ones = one_frag(lane);
    float m = -1e30f, l = 0.f;
    f32x16 ot = zero16();
    const int nkt = (S + 31) / 32;
    // Every global load of key tile kt + 1 (V rows, K fragments, mask byte, dropout multipliers) is issued before the arithmetic of tile
    // kt: a wave is alone on its SIMD here (B x H x L / 32 single-wave workgroups), so an un-prefetched tile exposed one L2 / HBM round
    // trip per 32 keys - most of the launch (24 us for 300 keys).  Loads of a tile past the end are predicated off by their bounds.
    uint4 va, vb;
    bf16x8 kf0, kf1;
    unsigned msk;
    MultRaw mraw;
    load_tile_regs(va, vb, vs, v.ts, S, 0, lane);
    kf0 = rowfrag_g(ks, k.ts, S, 0, 0, lane);
    kf1 = rowfrag_g(ks, k.ts, S, 0, 1, lane);
    msk = keymask_load(kpm_b, S, 0, lane);
    if (mult) load_mult_raw<VEC4>(mraw, mrow, qok, S, 0, h);
    for (int kt = 0; kt < nkt; ++kt) {
        store_tile_regs(vimg, va, vb, lane);
        const bf16x8 ck0 = kf0, ck1 = kf1;
        const unsigned cmsk = msk;
        const MultRaw cm = mraw;
        load_tile_regs(va, vb, vs, v.ts, S, 32 * (kt + 1), lane);
        kf0 = rowfrag_g(ks, k.ts, S, 32 * (kt + 1), 0, lane);
        kf1 = rowfrag_g(ks, k.ts, S, 32 * (kt + 1), 1, lane);
        msk = keymask_load(kpm_b, S, 32 * (kt + 1), lane);
        if (mult) load_mult_raw<VEC4>(mraw, mrow, qok, S, 32 * (kt + 1), h);
        f32x16 a = zero16();
        a = mma(ck0, qf[0], a);
        a = mma(ck1, qf[1], a);
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] *= scale;
        a = mma(keymask_from(cmsk, S, 32 * kt, lane), ones, a);
        float tm = a[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) tm = fmaxf(tm, a[i]);
        tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
        const float mn = fmaxf(m, tm), alpha = __expf(m - mn);
        float ts = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            a[i] = __expf(a[i] - mn);
            ts += a[i];
        }
        ts += __shfl_xor(ts, 32, 64);
        l = l * alpha + ts;
        m = mn;
        if (mult) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] *= (float)cm.e[i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[i] *= alpha;
        lds_settle();
        ot = mma(gather_perm(vimg, RSV, 0, 0, 0, true, lane), accfrag(a, 0), ot);
        ot = mma(gather_perm(vimg, RSV, 0, 0, 1, true, lane), accfrag(a, 1), ot);
        lds_settle();
    }
    store_rows((__bf16 *)o.p + (long)b * L * o.ts + head * D, o.ts, L, 32 * qt, ot, 1.0f / l, lane);
    if (qok && h == 0) lse[((long)b * H + head) * L + qrow] = m + __logf(l);
}

// delta[b][h][q] = dO[q] . O[q] over the head's 32 channels (= sum_k P dP): one thread per (b, h, q)
__global__ void mha_delta_kernel(Tok go, Tok o, float *__restrict__ delta, int B, int H, int L) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * H * L) return;
    const int qi = idx % L, head = (idx / L) % H, b = idx / ((long)L * H);
    const __bf16 *g = (const __bf16 *)go.p + ((long)b * L + qi) * go.ts + head * D;
    const __bf16 *y = (const __bf16 *)o.p + ((long)b * L + qi) * o.ts + head * D;
    float acc = 0.f;
#pragma unroll
    for (int c8 = 0; c8 < D; c8 += 8) {
        const bf16x8 gv = *(const bf16x8 *)(g + c8), yv = *(const bf16x8 *)(y + c8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += (float)gv[j] * (float)yv[j];
    }
    delta[idx] = acc;
}

template <bool VEC4>
__global__ __launch_bounds__(64) void mha_bwd_kernel(Tok q, Tok k, Tok v, Tok go, const unsigned char *__restrict__ kpm,
                                                     const __bf16 *__restrict__ mult, const float *__restrict__ lse,
                                                     const float *__restrict__ delta, TokW gq, TokW gk, TokW gv, int H, int L,
                                                     int S, float scale, int q_tiles) {
    __shared__ __attribute__((aligned(16))) __bf16 img0[TILE], img1[TILE];
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const __bf16 *qs = (const __bf16 *)q.p + (long)b * L * q.ts + head * D;
    const __bf16 *ks = (const __bf16 *)k.p + (long)b * S * k.ts + head * D;
    const __bf16 *vs = (const __bf16 *)v.p + (long)b * S * v.ts + head * D;
    const __bf16 *gs = (const __bf16 *)go.p + (long)b * L * go.ts + head * D;
    const unsigned char *kpm_b = kpm ? kpm + (long)b * S : nullptr;
    const float *lse_h = lse + ((long)b * H + head) * L, *dl_h = delta + ((long)b * H + head) * L;
    const __bf16 *mult_h = mult ? mult + ((long)b * H + head) * L * S : nullptr;
    if ((int)blockIdx.x < q_tiles) {
        // ---------------- role A: query tile on the lane -> dQ
        const int qt = blockIdx.x, qrow = 32 * qt + c;
        const bool qok = qrow < L;
        const float my_lse = qok ? lse_h[qrow] : 1e30f, my_dl = qok ? dl_h[qrow] : 0.f;
        const __bf16 *mrow = mult_h ? mult_h + (long)(qok ? qrow : 0) * S : nullptr;
        bf16x8 qf[2], of[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qf[s] = rowfrag_g(qs, q.ts, L, 32 * qt, s, lane);
            of[s] = rowfrag_g(gs, go.ts, L, 32 * qt, s, lane);
        }
        const bf16x8 ones = one_frag(lane);
        f32x16 dq = zero16();
        const int nkt = (S + 31) / 32;
        // loads of key tile kt + 1 in flight during the arithmetic of tile kt (see mha_fwd_kernel)
        uint4 ka, kb;
        bf16x8 vf0, vf1;
        unsigned msk;
        MultRaw mraw;
        load_tile_regs(ka, kb, ks, k.ts, S, 0, lane);
        vf0 = rowfrag_g(vs, v.ts, S, 0, 0, lane);
        vf1 = rowfrag_g(vs, v.ts, S, 0, 1, lane);
        msk = keymask_load(kpm_b, S, 0, lane);
        if (mult) load_mult_raw<VEC4>(mraw, mrow, qok, S, 0, h);
        for (int kt = 0; kt < nkt; ++kt) {
            store_tile_regs(img0, ka, kb, lane);
            const bf16x8 cv0 = vf0, cv1 = vf1;
            const unsigned cmsk = msk;
            const MultRaw cm = mraw;
            load_tile_regs(ka, kb, ks, k.ts, S, 32 * (kt + 1), lane);
            vf0 = rowfrag_g(vs, v.ts, S, 32 * (kt + 1), 0, lane);
            vf1 = rowfrag_g(vs, v.ts, S, 32 * (kt + 1), 1, lane);
            msk = keymask_load(kpm_b, S, 32 * (kt + 1), lane);
            if (mult) load_mult_raw<VEC4>(mraw, mrow, qok, S, 32 * (kt + 1), h);
            f32x16 dp = zero16();
            dp = mma(cv0, of[0], dp);          // dP^T[key][query] = V[key] . dO[query]
            dp = mma(cv1, of[1], dp);
            lds_settle();
            f32x16 a = zero16();
            a = mma(rowfrag_l(img0, 0, lane), qf[0], a);
            a = mma(rowfrag_l(img0, 1, lane), qf[1], a);
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] *= scale;
            a = mma(keymask_from(cmsk, S, 32 * kt, lane), ones, a);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __expf(a[i] - my_lse);
                const float dpi = mult ? dp[i] * (float)cm.e[i] : dp[i];
                a[i] = p * (dpi - my_dl);                                            // dS
            }
            dq = mma(gather_perm(img0, RSV, 0, 0, 0, true, lane), accfrag(a, 0), dq);   // dQ^T[d][q] += K[key][d] dS^T[key][q]
            dq = mma(gather_perm(img0, RSV, 0, 0, 1, true, lane), accfrag(a, 1), dq);
            lds_settle();
        }
        store_rows((__bf16 *)gq.p + (long)b * L * gq.ts + head * D, gq.ts, L, 32 * qt, dq, scale, lane);
    } else {
        // ---------------- role B: key tile on the lane -> dK, dV
        const int kt = blockIdx.x - q_tiles, key = 32 * kt + c;
        const bool kok = key < S;
        const float mk = (!kok || (kpm_b && kpm_b[kok ? key : 0])) ? -1e30f : 0.f;
        bf16x8 kf[2], vf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            kf[s] = rowfrag_g(ks, k.ts, S, 32 * kt, s, lane);
            vf[s] = rowfrag_g(vs, v.ts, S, 32 * kt, s, lane);
        }
        f32x16 dk = zero16(), dv = zero16();
        const int nqt = (L + 31) / 32;
        // loads of query tile qt + 1 (Q and dO rows, the 16 row statistics and multipliers of this lane) in flight during tile qt
        uint4 qa, qb, ga, gb;
        float ls[16], dl[16];
        __bf16 mu[16];
        auto load_q = [&](int qt) {
            load_tile_regs(qa, qb, qs, q.ts, L, 32 * qt, lane);
            load_tile_regs(ga, gb, gs, go.ts, L, 32 * qt, lane);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int qrow = 32 * qt + acc_row(i, h);
                const bool qok = qrow < L;
                ls[i] = qok ? lse_h[qrow] : 1e30f;
                dl[i] = qok ? dl_h[qrow] : 0.f;
                mu[i] = mult_h ? ((qok && kok) ? mult_h[(long)qrow * S + key] : (__bf16)0.0f) : (__bf16)1.0f;
            }
        };
        load_q(0);
        for (int qt = 0; qt < nqt; ++qt) {
            store_tile_regs(img0, qa, qb, lane);
            store_tile_regs(img1, ga, gb, lane);
            float cls[16], cdl[16], cmu[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                cls[i] = ls[i];
                cdl[i] = dl[i];
                cmu[i] = (float)mu[i];
            }
            load_q(qt + 1);
            lds_settle();
            f32x16 a = zero16(), dp = zero16();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                a = mma(rowfrag_l(img0, s, lane), kf[s], a);                        // S[query][key]: rows = queries, key on the lane
                dp = mma(rowfrag_l(img1, s, lane), vf[s], dp);                      // dP[query][key] = dO[query] . V[key]
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __expf(a[i] * scale + mk - cls[i]);
                a[i] = p * cmu[i];                                                  // dropped probabilities -> dV
                dp[i] = p * (dp[i] * cmu[i] - cdl[i]);                              // dS -> dK
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                dv = mma(gather_perm(img1, RSV, 0, 0, s, true, lane), accfrag(a, s), dv);    // dV^T[d][key] += dO[q][d] P[q][key]
                dk = mma(gather_perm(img0, RSV, 0, 0, s, true, lane), accfrag(dp, s), dk);   // dK^T[d][key] += Q[q][d] dS[q][key]
            }
            lds_settle();
        }
        store_rows((__bf16 *)gk.p + (long)b * S * gk.ts + head * D, gk.ts, S, 32 * kt, dk, scale, lane);
        store_rows((__bf16 *)gv.p + (long)b * S * gv.ts + head * D, gv.ts, S, 32 * kt, dv, 1.0f, lane);
    }
}

bool tok_ok(const void *p, long ts) { return p && (uintptr_t)p % 16 == 0 && ts % 8 == 0; }

}  // namespace mha

// q (B,L,*) / k, v (B,S,*) bf16 with token strides *_ts (elements, multiples of 8; 16-byte aligned bases; head h = channels
// 32h..32h+31), kpm (B,S) uint8 or null, mult (B,H,L,S) bf16 or null -> out (B,L,*) bf16 token stride o_ts, lse (B,H,L) fp32.
extern "C" int gwd_mha_flash_forward(const void *q, const void *k, const void *v, int64_t q_ts, int64_t k_ts, int64_t v_ts,
                                     const uint8_t *key_padding_mask, const void *mult, void *out, int64_t o_ts, float *lse,
                                     int32_t B, int32_t H, int32_t L, int32_t S, float scale, int32_t dtype, void *stream) {
    using namespace mha;
    if (dtype != GWD_BF16) return -2;
    if (!tok_ok(q, q_ts) || !tok_ok(k, k_ts) || !tok_ok(v, v_ts) || !tok_ok(out, o_ts) || !lse || B <= 0 || H <= 0 || L <= 0 || S <= 0) return -1;
    if (B > 65535 || H > 65535) return -4;
    const dim3 grid((L + 31) / 32, H, B);
    const Tok tq{q, q_ts}, tk{k, k_ts}, tv{v, v_ts};
    const TokW to{out, o_ts};
    if (S % 4 == 0 && (!mult || (uintptr_t)mult % 8 == 0))
        mha_fwd_kernel<true><<<grid, 64, 0, (hipStream_t)stream>>>(tq, tk, tv, key_padding_mask, (const __bf16 *)mult, to, lse, H, L, S, scale);
    else
        mha_fwd_kernel<false><<<grid, 64, 0, (hipStream_t)stream>>>(tq, tk, tv, key_padding_mask, (const __bf16 *)mult, to, lse, H, L, S, scale);
    GWD_CHECK_LAUNCH();
    return 0;
}

// Backward of the above: go (B,L,*) -> gq (B,L,*), gk, gv (B,S,*); out / lse are the forward's; delta (B,H,L) fp32 is scratch.
extern "C" int gwd_mha_flash_backward(const void *q, const void *k, const void *v, const void *go, const void *out, int64_t q_ts,
                                      int64_t k_ts, int64_t v_ts, int64_t go_ts, int64_t o_ts, const uint8_t *key_padding_mask,
                                      const void *mult, const float *lse, float *delta, void *gq, void *gk, void *gv, int64_t gq_ts,
                                      int64_t gk_ts, int64_t gv_ts, int32_t B, int32_t H, int32_t L, int32_t S, float scale,
                                      int32_t dtype, void *stream) {
    using namespace mha;
    if (dtype != GWD_BF16) return -2;
    if (!tok_ok(q, q_ts) || !tok_ok(k, k_ts) || !tok_ok(v, v_ts) || !tok_ok(go, go_ts) || !tok_ok(out, o_ts) || !tok_ok(gq, gq_ts) ||
        !tok_ok(gk, gk_ts) || !tok_ok(gv, gv_ts) || !lse || !delta || B <= 0 || H <= 0 || L <= 0 || S <= 0)
        return -1;
    if (B > 65535 || H > 65535) return -4;
    hipStream_t s = (hipStream_t)stream;
    const Tok tq{q, q_ts}, tk{k, k_ts}, tv{v, v_ts}, tg{go, go_ts}, to{out, o_ts};
    const long rows = (long)B * H * L;
    mha_delta_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, s>>>(tg, to, delta, B, H, L);
    const int q_tiles = (L + 31) / 32, k_tiles = (S + 31) / 32;
    const dim3 grid(q_tiles + k_tiles, H, B);
    const TokW wq{gq, gq_ts}, wk{gk, gk_ts}, wv{gv, gv_ts};
    if (S % 4 == 0 && (!mult || (uintptr_t)mult % 8 == 0))
        mha_bwd_kernel<true><<<grid, 64, 0, s>>>(tq, tk, tv, tg, key_padding_mask, (const __bf16 *)mult, lse, delta, wq, wk, wv, H, L, S, scale, q_tiles);
    else
        mha_bwd_kernel<false><<<grid, 64, 0, s>>>(tq, tk, tv, tg, key_padding_mask, (const __bf16 *)mult, lse, delta, wq, wk, wv, H, L, S, scale, q_tiles);
    GWD_CHECK_LAUNCH();
    return 0;
}
