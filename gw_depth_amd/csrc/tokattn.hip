// Class-token ("transposed") attention of WindowClassAttention (src/models/multiscale_transformerr.py:560-578).
// Per (window, head):   S[r][c] = scale * sum_n q[n][r] k[n][c]   (r < 4 token channels, c < E feature channels)
//                       A = softmax_c(S);   O[n][r] = sum_c A[r][c] v[n][c]
// i.e. the 49 tokens of the window are the CONTRACTION index of the score and the free index of the output.
// One 64-lane wave per problem, lane = token n: S and dA are 4*E wave reductions (DPP/shuffle butterflies, every
// lane ends up with the full 4 x E matrix), everything else is lane-local.  E = (dim + 128) / 16 in {12, 16, 24}.
#include "common.h"

namespace {

constexpr int NT = 49, R = 4;

struct Op {
    void *p;
    long ws, ts, hs;
};

template <typename T, int E>
__device__ __forceinline__ void load_rows(const Op &q, const Op &k, const Op &v, long w, int h, int n, bool live, float (&qr)[R],
                                          float (&kr)[E], float (&vr)[E]) {
    const T *qp = (const T *)q.p + w * q.ws + n * q.ts + h * q.hs;
    const T *kp = (const T *)k.p + w * k.ws + n * k.ts + h * k.hs;
    const T *vp = (const T *)v.p + w * v.ws + n * v.ts + h * v.hs;
#pragma unroll
    for (int r = 0; r < R; ++r) qr[r] = live ? to_f32(qp[r]) : 0.f;
#pragma unroll
    for (int c = 0; c < E; ++c) {
        kr[c] = live ? to_f32(kp[c]) : 0.f;
        vr[c] = live ? to_f32(vp[c]) : 0.f;
    }
}

template <int E>
__device__ __forceinline__ void scores_softmax(const float (&qr)[R], const float (&kr)[E], float scale, float (&a)[R][E]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < E; ++c) {
            a[r][c] = wave_sum(qr[r] * kr[c]) * scale;
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

template <typename T, int E>
__global__ __launch_bounds__(256) void tokattn_fwd_kernel(Op q, Op k, Op v, Op o, long n_problems, int heads, float scale) {
    const int lane = threadIdx.x & 63;
    const bool live = lane < NT;
    const int n = live ? lane : 0;
    for (long pb = (long)blockIdx.x * 4 + (threadIdx.x >> 6); pb < n_problems; pb += (long)gridDim.x * 4) {
        const long w = pb / heads;
        const int h = (int)(pb - w * heads);
        float qr[R], kr[E], vr[E], a[R][E];
        load_rows<T, E>(q, k, v, w, h, n, live, qr, kr, vr);
        scores_softmax<E>(qr, kr, scale, a);
        if (live) {
            T *op = (T *)o.p + w * o.ws + n * o.ts + h * o.hs;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < E; ++c) acc += a[r][c] * vr[c];
                op[r] = from_f32<T>(acc);
            }
        }
    }
}

template <typename T, int E>
__global__ __launch_bounds__(256) void tokattn_bwd_kernel(Op q, Op k, Op v, Op go, Op gq, Op gk, Op gv, long n_problems, int heads,
                                                          float scale) {
    const int lane = threadIdx.x & 63;
    const bool live = lane < NT;
    const int n = live ? lane : 0;
    for (long pb = (long)blockIdx.x * 4 + (threadIdx.x >> 6); pb < n_problems; pb += (long)gridDim.x * 4) {
        const long w = pb / heads;
        const int h = (int)(pb - w * heads);
        float qr[R], kr[E], vr[E], a[R][E], dor[R];
        load_rows<T, E>(q, k, v, w, h, n, live, qr, kr, vr);
        {
            const T *gp = (const T *)go.p + w * go.ws + n * go.ts + h * go.hs;
#pragma unroll
            for (int r = 0; r < R; ++r) dor[r] = live ? to_f32(gp[r]) : 0.f;
        }
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
                da[c] = wave_sum(dor[r] * vr[c]);          // dA[r][c] = sum_n dO[n][r] v[n][c]
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
#pragma unroll
            for (int r = 0; r < R; ++r) qp[r] = from_f32<T>(dq[r]);
#pragma unroll
            for (int c = 0; c < E; ++c) {
                kp[c] = from_f32<T>(dk[c]);
                vp[c] = from_f32<T>(dv[c]);
            }
        }
    }
}

inline Op mk(const gwd_strided *s) { return Op{s->p, s->ws, s->ts, s->hs}; }

template <typename T, int E>
int run(bool bwd, const gwd_strided *const *s, long n_problems, int heads, float scale, hipStream_t st) {
    long bx = (n_problems + 3) / 4;
    if (bx > 4096) bx = 4096;
    if (!bwd)
        tokattn_fwd_kernel<T, E><<<(unsigned)bx, 256, 0, st>>>(mk(s[0]), mk(s[1]), mk(s[2]), mk(s[3]), n_problems, heads, scale);
    else
        tokattn_bwd_kernel<T, E><<<(unsigned)bx, 256, 0, st>>>(mk(s[0]), mk(s[1]), mk(s[2]), mk(s[3]), mk(s[4]), mk(s[5]), mk(s[6]),
                                                             n_problems, heads, scale);
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

extern "C" int gwd_tokattn_forward(const gwd_strided *q, const gwd_strided *k, const gwd_strided *v, const gwd_strided *o,
                                   int64_t n_windows, int32_t heads, int32_t e, float scale, int32_t dtype, void *stream) {
    if (!q || !k || !v || !o || !q->p || !k->p || !v->p || !o->p || n_windows <= 0 || heads <= 0) return -1;
    const gwd_strided *s[4] = {q, k, v, o};
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
    if (dtype == GWD_BF16) return by_e<__bf16>(e, true, s, n_windows * heads, heads, scale, (hipStream_t)stream);
    if (dtype == GWD_F32) return by_e<float>(e, true, s, n_windows * heads, heads, scale, (hipStream_t)stream);
    return -2;
}
