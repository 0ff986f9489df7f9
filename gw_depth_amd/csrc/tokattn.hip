// Class-token ("transposed") attention of WindowClassAttention (src/models/multiscale_transformerr.py:560-578).
// Per (window, head):   S[r][c] = scale * sum_n q[n][r] k[n][c]   (r < 4 token channels, c < E feature channels)
//                       A = softmax_c(S);   O[n][r] = sum_c A[r][c] v[n][c]
// i.e. the 49 tokens of the window are the CONTRACTION index of the score and the free index of the output.
// One 64-lane wave per problem, lane = token n: S and dA are 4*E wave reductions (DPP/shuffle butterflies, every
// lane ends up with the full 4 x E matrix), everything else is lane-local.  E = (dim + 128) / 16 in {12, 16, 24}.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int NT = 49, R = 4;

struct Op {
    void *p;
    long ws, ts, hs;
};

// 4 consecutive elements per access (8 bytes of bf16, 16 of fp32) when the host has checked pointers and strides;
// element-wise otherwise.  The rows are 24-96 bytes at a token stride of hundreds of bytes, so each lane touches its
// own cache line: one wide access per line instead of one per element.
template <typename T> struct Quad;
template <> struct Quad<float> { typedef float4 type; };
template <> struct Quad<__bf16> { typedef uint2 type; };

template <typename T, int N, bool VEC>
__device__ __forceinline__ void load_n(const T *p, bool live, float (&dst)[N]) {
    if constexpr (VEC) {
#pragma unroll
        for (int c = 0; c < N; c += 4) {
            typename Quad<T>::type raw = {};
            if (live) raw = *(const typename Quad<T>::type *)(p + c);
            const T *e = (const T *)&raw;
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[c + j] = to_f32(e[j]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < N; ++c) dst[c] = live ? to_f32(p[c]) : 0.f;
    }
}

template <typename T, int N, bool VEC>
__device__ __forceinline__ void store_n(T *p, const float (&src)[N]) {
    if constexpr (VEC) {
#pragma unroll
        for (int c = 0; c < N; c += 4) {
            typename Quad<T>::type raw;
            T *e = (T *)&raw;
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = from_f32<T>(src[c + j]);
            *(typename Quad<T>::type *)(p + c) = raw;
        }
    } else {
#pragma unroll
        for (int c = 0; c < N; ++c) p[c] = from_f32<T>(src[c]);
    }
}

template <typename T, int E, bool VEC>
__device__ __forceinline__ void load_rows(const Op &q, const Op &k, const Op &v, long w, int h, int n, bool live, float (&qr)[R],
                                          float (&kr)[E], float (&vr)[E]) {
    const T *qp = (const T *)q.p + w * q.ws + n * q.ts + h * q.hs;
    const T *kp = (const T *)k.p + w * k.ws + n * k.ts + h * k.hs;
    const T *vp = (const T *)v.p + w * v.ws + n * v.ts + h * v.hs;
    load_n<T, R, VEC>(qp, live, qr);
    load_n<T, E, VEC>(kp, live, kr);
    load_n<T, E, VEC>(vp, live, vr);
}

template <int E>
__device__ __forceinline__ void scores_softmax(const float (&qr)[R], const float (&kr)[E], float scale, float (&a)[R][E]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < E; ++c) {
            a[r][c] = wave_sum_uniform(qr[r] * kr[c]) * scale;
            mx = fmaxf(mx, a[r][c]);
        }
        float l = 0.f;
#pragma unroll
        for (int c = 0; c < E; ++c) {
            a[r][c] = __expf(a[r][c] - mx);
            l += a[r][c];
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int c = 0; c < E; ++c) a[r][c] *= inv;
    }
}

template <typename T, int E, bool VEC>
__global__ __launch_bounds__(256) void tokattn_fwd_kernel(Op q, Op k, Op v, Op o, long n_problems, int heads, float scale) {
    const int lane = threadIdx.x & 63;
    const bool live = lane < NT;
    const int n = live ? lane : 0;
    const long vblock = (heads % 4) ? (long)blockIdx.x : xcd_grouped_block(blockIdx.x, gridDim.x, heads / 4);      // a window's head groups on one XCD
    for (long pb = vblock * 4 + (threadIdx.x >> 6); pb < n_problems; pb += (long)gridDim.x * 4) {
        const long w = pb / heads;
        const int h = (int)(pb - w * heads);
        float qr[R], kr[E], vr[E], a[R][E];
        load_rows<T, E, VEC>(q, k, v, w, h, n, live, qr, kr, vr);
        scores_softmax<E>(qr, kr, scale, a);
        if (live) {
            T *op = (T *)o.p + w * o.ws + n * o.ts + h * o.hs;
            float ov[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < E; ++c) acc += a[r][c] * vr[c];
                ov[r] = acc;
            }
            store_n<T, R, VEC>(op, ov);
        }
    }
}

template <typename T, int E, bool VEC>
__global__ __launch_bounds__(256) void tokattn_bwd_kernel(Op q, Op k, Op v, Op go, Op gq, Op gk, Op gv, long n_problems, int heads,
                                                          float scale) {
    const int lane = threadIdx.x & 63;
    const bool live = lane < NT;
    const int n = live ? lane : 0;
    const long vblock = (heads % 4) ? (long)blockIdx.x : xcd_grouped_block(blockIdx.x, gridDim.x, heads / 4);      // a window's head groups on one XCD
    for (long pb = vblock * 4 + (threadIdx.x >> 6); pb < n_problems; pb += (long)gridDim.x * 4) {
        const long w = pb / heads;
        const int h = (int)(pb - w * heads);
        float qr[R], kr[E], vr[E], a[R][E], dor[R];
        load_rows<T, E, VEC>(q, k, v, w, h, n, live, qr, kr, vr);
        load_n<T, R, VEC>((const T *)go.p + w * go.ws + n * go.ts + h * go.hs, live, dor);
        scores_softmax<E>(qr, kr, scale, a);
        float dv[E], dk[E], dq[R];
#pragma unroll
        for (int c = 0; c < E; ++c) dv[c] = dk[c] = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float da[E];
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < E; ++c) {
                da[c] = wave_sum_uniform(dor[r] * vr[c]);          // dA[r][c] = sum_n dO[n][r] v[n][c]
                dot += a[r][c] * da[c];
                dv[c] += a[r][c] * dor[r];
            }
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < E; ++c) {
                const float ds = a[r][c] * (da[c] - dot) * scale;
                acc += ds * kr[c];
                dk[c] += ds * qr[r];
            }
            dq[r] = acc;
        }
        if (live) {
            T *qp = (T *)gq.p + w * gq.ws + n * gq.ts + h * gq.hs;
            T *kp = (T *)gk.p + w * gk.ws + n * gk.ts + h * gk.hs;
            T *vp = (T *)gv.p + w * gv.ws + n * gv.ts + h * gv.hs;
            store_n<T, R, VEC>(qp, dq);
            store_n<T, E, VEC>(kp, dk);
            store_n<T, E, VEC>(vp, dv);
        }
    }
}

inline Op mk(const gwd_strided *s) { return Op{s->p, s->ws, s->ts, s->hs}; }

template <typename T, int E>
int run(bool bwd, const gwd_strided *const *s, long n_problems, int heads, float scale, hipStream_t st) {
    long bx = (n_problems + 3) / 4;
    if (bx > 4096) bx = 4096;
    // wide accesses need every operand's base and strides to be multiples of 4 elements
    bool vec = true;
    for (int i = 0; i < (bwd ? 7 : 4); ++i)
        vec = vec && ((uintptr_t)s[i]->p % (4 * sizeof(T)) == 0) && s[i]->ws % 4 == 0 && s[i]->ts % 4 == 0 && s[i]->hs % 4 == 0;
    if (!bwd) {
        if (vec) tokattn_fwd_kernel<T, E, true><<<(unsigned)bx, 256, 0, st>>>(mk(s[0]), mk(s[1]), mk(s[2]), mk(s[3]), n_problems, heads, scale);
        else tokattn_fwd_kernel<T, E, false><<<(unsigned)bx, 256, 0, st>>>(mk(s[0]), mk(s[1]), mk(s[2]), mk(s[3]), n_problems, heads, scale);
    } else {
        if (vec) tokattn_bwd_kernel<T, E, true><<<(unsigned)bx, 256, 0, st>>>(mk(s[0]), mk(s[1]), mk(s[2]), mk(s[3]), mk(s[4]), mk(s[5]), mk(s[6]), n_problems, heads, scale);
        else tokattn_bwd_kernel<T, E, false><<<(unsigned)bx, 256, 0, st>>>(mk(s[0]), mk(s[1]), mk(s[2]), mk(s[3]), mk(s[4]), mk(s[5]), mk(s[6]), n_problems, heads, scale);
    }
    GWD_CHECK_LAUNCH();
    return 0;
}

template <typename T>
int by_e(int e, bool bwd, const gwd_strided *const *s, long np, int heads, float scale, hipStream_t st) {
    switch (e) {
        case 12: return run<T, 12>(bwd, s, np, heads, scale, st);
        case 16: return run<T, 16>(bwd, s, np, heads, scale, st);
        case 24: return run<T, 24>(bwd, s, np, heads, scale, st);
        default: return -4;
    }
}

}  // namespace

// mfattn.hip: the same problem on the matrix cores (bf16); 0 = launched, 1 = not covered
int gwd_mfattn_token(bool backward, bool pair, const gwd_strided *const *ops, long n_problems, int heads, int e, float scale, hipStream_t s);

static bool mfma_token_enabled() { return true; }       // bf16: csrc/mfattn.hip; the lane-per-token kernels below are the fp32 (parity) path

extern "C" int gwd_tokattn_forward(const gwd_strided *q, const gwd_strided *k, const gwd_strided *v, const gwd_strided *o,
                                   int64_t n_windows, int32_t heads, int32_t e, float scale, int32_t dtype, void *stream) {
    if (!q || !k || !v || !o || !q->p || !k->p || !v->p || !o->p || n_windows <= 0 || heads <= 0) return -1;
    const gwd_strided *s[4] = {q, k, v, o};
    if (dtype == GWD_BF16 && mfma_token_enabled() && gwd_mfattn_token(false, false, s, n_windows * heads, heads, e, scale, (hipStream_t)stream) == 0) {
        GWD_CHECK_LAUNCH();
        return 0;
    }
    if (dtype == GWD_BF16) return by_e<__bf16>(e, false, s, n_windows * heads, heads, scale, (hipStream_t)stream);
    if (dtype == GWD_F32) return by_e<float>(e, false, s, n_windows * heads, heads, scale, (hipStream_t)stream);
    return -2;
}

extern "C" int gwd_tokattn_backward(const gwd_strided *q, const gwd_strided *k, const gwd_strided *v, const gwd_strided *go,
                                    const gwd_strided *gq, const gwd_strided *gk, const gwd_strided *gv, int64_t n_windows,
                                    int32_t heads, int32_t e, float scale, int32_t dtype, void *stream) {
    if (!q || !k || !v || !go || !gq || !gk || !gv || n_windows <= 0 || heads <= 0) return -1;
    if (!q->p || !k->p || !v->p || !go->p || !gq->p || !gk->p || !gv->p) return -1;
    const gwd_strided *s[7] = {q, k, v, go, gq, gk, gv};
    if (dtype == GWD_BF16 && mfma_token_enabled() && gwd_mfattn_token(true, false, s, n_windows * heads, heads, e, scale, (hipStream_t)stream) == 0) {
        GWD_CHECK_LAUNCH();
        return 0;
    }
    if (dtype == GWD_BF16) return by_e<__bf16>(e, true, s, n_windows * heads, heads, scale, (hipStream_t)stream);
    if (dtype == GWD_F32) return by_e<float>(e, true, s, n_windows * heads, heads, scale, (hipStream_t)stream);
    return -2;
}

// Both class tokens of a WindowClassAttention block in one launch (multiscale_transformerr.py:561-578: the depth and the segmentation token query
// the SAME global_k / global_v): q / q2 (W,49,heads,4) -> o / o2; the backward returns gq / gq2 and the SUM of the two calls' gk / gv.  bf16 only
// (-2 otherwise: the caller issues two gwd_tokattn_* calls), e in {12, 16, 24} (-4).
extern "C" int gwd_tokattn_pair_forward(const gwd_strided *q, const gwd_strided *q2, const gwd_strided *k, const gwd_strided *v, const gwd_strided *o,
                                        const gwd_strided *o2, int64_t n_windows, int32_t heads, int32_t e, float scale, int32_t dtype, void *stream) {
    if (!q || !q2 || !k || !v || !o || !o2 || !q->p || !q2->p || !k->p || !v->p || !o->p || !o2->p || n_windows <= 0 || heads <= 0) return -1;
    if (dtype != GWD_BF16) return -2;
    const gwd_strided *s[6] = {q, k, v, o, q2, o2};
    if (gwd_mfattn_token(false, true, s, n_windows * heads, heads, e, scale, (hipStream_t)stream) != 0) return -4;
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_tokattn_pair_backward(const gwd_strided *q, const gwd_strided *q2, const gwd_strided *k, const gwd_strided *v, const gwd_strided *go,
                                         const gwd_strided *go2, const gwd_strided *gq, const gwd_strided *gq2, const gwd_strided *gk, const gwd_strided *gv,
                                         int64_t n_windows, int32_t heads, int32_t e, float scale, int32_t dtype, void *stream) {
    if (!q || !q2 || !k || !v || !go || !go2 || !gq || !gq2 || !gk || !gv || n_windows <= 0 || heads <= 0) return -1;
    if (!q->p || !q2->p || !k->p || !v->p || !go->p || !go2->p || !gq->p || !gq2->p || !gk->p || !gv->p) return -1;
    if (dtype != GWD_BF16) return -2;
    const gwd_strided *s[10] = {q, k, v, go, gq, gk, gv, q2, go2, gq2};
    if (gwd_mfattn_token(true, true, s, n_windows * heads, heads, e, scale, (hipStream_t)stream) != 0) return -4;
    GWD_CHECK_LAUNCH();
    return 0;
}
