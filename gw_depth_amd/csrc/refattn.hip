// Line-point-guided part of the 1/32-stage WindowAttention (/root/reference/src/models/multiscale_transformerr.py:295-310):
//
//   ra[b, t, r, h]    = scale * sum_d q[b, t, h, d] * ref_k[b, r, h, d]          (:296-298; t = window * 49 + token)
//   ... three rounds of conv3x3 / instance-norm / GELU over the (t, r) map with the heads as channels (own kernels) ...
//   att               = softmax over r of ra2[b, t, :, h]                          (:304-305)
//   q_new[b, t, h, d] = sum_r att[b, t, r, h] * ref_v[b, r, h, d]                 (:306-309)
//
// The reference runs these as two batched einsums plus permutes; here each is one kernel writing the layout its consumer
// wants: `ra` pixel-major (B, T, R, H) = what the diffusion conv reads, q_new as the (windows, 49, heads, hd) operand of the
// window attention.  R reference tokens (40), H heads (16), hd 32: ~70 MFLOP per call - latency-class VALU work, no MFMA.
// Gradients that sum over the T = windows * 49 tokens (d ref_k, d ref_v) are gathered by one thread per output element
// walking t, so nothing needs atomics and the result is bit-reproducible.
#include "common.h"

namespace {

struct QOp {                 // q as the (windows, 49, heads, hd) operand inside the packed qkv projection
    const void *p;
    long ws, ts, hs;         // window / token / head strides in elements
};
struct QOpW {
    void *p;
    long ws, ts, hs;
};

// ra[b][t][r][h] = scale * q[b,t,h,:] . refk[b,r,h,:]; one thread per output, h fastest
template <typename T>
__global__ void ref_scores_fwd_kernel(QOp q, const T *__restrict__ refk, T *__restrict__ ra, int B, int nwin, int R, int H, int hd,
                                      float scale) {
    const long total = (long)B * nwin * 49 * R * H;
    const int C = H * hd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int h = (int)(i % H);
        const int r = (int)((i / H) % R);
        const long bt = i / ((long)H * R);
        const int t = (int)(bt % (nwin * 49)), b = (int)(bt / (nwin * 49));
        const T *qp = (const T *)q.p + ((long)b * nwin + t / 49) * q.ws + (long)(t % 49) * q.ts + (long)h * q.hs;
        const T *kp = refk + ((long)b * R + r) * C + h * hd;
        float acc = 0.f;
        for (int d = 0; d < hd; ++d) acc += to_f32(qp[d]) * to_f32(kp[d]);
        ra[i] = from_f32<T>(acc * scale);
    }
}

// role 0: dq[b,t,h,d] = scale * sum_r g[b,t,r,h] * refk[b,r,h,d]      (one thread per (b,t,h,d))
// role 1: drefk[b,r,h,d] = scale * sum_t g[b,t,r,h] * q[b,t,h,d]      (one thread per (b,r,h,d), walks t)
template <typename T>
__global__ void ref_scores_bwd_kernel(QOp q, const T *__restrict__ refk, const T *__restrict__ g, QOpW dq, float *__restrict__ drefk,
                                      int B, int nwin, int R, int H, int hd, float scale, int dq_blocks) {
    const int C = H * hd, Tn = nwin * 49;
    if ((int)blockIdx.x < dq_blocks) {
        const long total = (long)B * Tn * C;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)dq_blocks * blockDim.x) {
            const int d = (int)(i % hd), h = (int)((i / hd) % H);
            const long bt = i / C;
            const int t = (int)(bt % Tn), b = (int)(bt / Tn);
            const T *gp = g + (bt * R) * H + h;
            const T *kp = refk + (long)b * R * C + h * hd + d;
            float acc = 0.f;
            for (int r = 0; r < R; ++r) acc += to_f32(gp[(long)r * H]) * to_f32(kp[(long)r * C]);
            T *dst = (T *)dq.p + ((long)b * nwin + t / 49) * dq.ws + (long)(t % 49) * dq.ts + (long)h * dq.hs + d;
            *dst = from_f32<T>(acc * scale);
        }
    } else {
        const long total = (long)B * R * C;
        const long nb = gridDim.x - dq_blocks;
        for (long i = (long)(blockIdx.x - dq_blocks) * blockDim.x + threadIdx.x; i < total; i += nb * blockDim.x) {
            const int d = (int)(i % hd), h = (int)((i / hd) % H);
            const int r = (int)((i / C) % R), b = (int)(i / ((long)C * R));
            const T *gp = g + ((long)b * Tn * R + r) * H + h;
            float acc = 0.f;
            for (int t = 0; t < Tn; ++t) {
                const T *qp = (const T *)q.p + ((long)b * nwin + t / 49) * q.ws + (long)(t % 49) * q.ts + (long)h * q.hs + d;
                acc += to_f32(gp[(long)t * R * H]) * to_f32(*qp);
            }
            drefk[i] = acc * scale;
        }
    }
}

// One workgroup per (b, t): softmax over the R reference tokens for every head, then q_new[b,t,h,:] = att[:,h] . refv[b,:,h,:].
// att (B,T,R,H) is written for the backward pass when asked for.
template <typename T>
__global__ void ref_mix_fwd_kernel(const T *__restrict__ ra, const T *__restrict__ refv, T *__restrict__ qnew, T *__restrict__ att_out,
                                   int R, int H, int hd, int Tn) {
    extern __shared__ float sm[];          // [R][H]
    const long bt = blockIdx.x;
    const int C = H * hd, RH = R * H;
    const T *src = ra + bt * RH;
    for (int e = threadIdx.x; e < RH; e += blockDim.x) sm[e] = to_f32(src[e]);
    __syncthreads();
    for (int h = threadIdx.x; h < H; h += blockDim.x) {       // H <= 64: one lane per head, R terms each
        float m = -INFINITY;
        for (int r = 0; r < R; ++r) m = fmaxf(m, sm[r * H + h]);
        float l = 0.f;
        for (int r = 0; r < R; ++r) {
            const float p = __expf(sm[r * H + h] - m);
            sm[r * H + h] = p;
            l += p;
        }
        const float inv = 1.0f / l;
        for (int r = 0; r < R; ++r) sm[r * H + h] *= inv;
    }
    __syncthreads();
    if (att_out)
        for (int e = threadIdx.x; e < RH; e += blockDim.x) att_out[bt * RH + e] = from_f32<T>(sm[e]);
    const T *vb = refv + (bt / Tn) * R * C;                   // blockIdx.x = b * T + t
    for (int e = threadIdx.x; e < C; e += blockDim.x) {
        const int h = e / hd;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += sm[r * H + h] * to_f32(vb[(long)r * C + e]);
        qnew[bt * C + e] = from_f32<T>(acc);
    }
}

// backward, per (b, t): datt[r,h] = g[b,t,h,:] . refv[b,r,h,:];  dra = att * (datt - sum_r att * datt)
template <typename T>
__global__ void ref_mix_bwd_kernel(const T *__restrict__ att, const T *__restrict__ refv, const T *__restrict__ g, T *__restrict__ dra,
                                   int R, int H, int hd, int Tn) {
    extern __shared__ float sm[];          // g row [C] | datt [R][H] | dot [H]
    const long bt = blockIdx.x;
    const int C = H * hd, RH = R * H;
    float *gs = sm, *da = sm + C, *dot = da + RH;
    for (int e = threadIdx.x; e < C; e += blockDim.x) gs[e] = to_f32(g[bt * C + e]);
    __syncthreads();
    const T *vb = refv + (bt / Tn) * R * C;
    for (int e = threadIdx.x; e < RH; e += blockDim.x) {
        const int r = e / H, h = e % H;
        const T *vp = vb + (long)r * C + h * hd;
        float acc = 0.f;
        for (int d = 0; d < hd; ++d) acc += gs[h * hd + d] * to_f32(vp[d]);
        da[e] = acc;
    }
    __syncthreads();
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += to_f32(att[bt * RH + r * H + h]) * da[r * H + h];
        dot[h] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < RH; e += blockDim.x)
        dra[bt * RH + e] = from_f32<T>(to_f32(att[bt * RH + e]) * (da[e] - dot[e % H]));
}

// drefv[b,r,h,d] = sum_t att[b,t,r,h] * g[b,t,h,d]: one thread per output walks t
template <typename T>
__global__ void ref_mix_bwd_v_kernel(const T *__restrict__ att, const T *__restrict__ g, float *__restrict__ drefv, int B, int Tn, int R,
                                     int H, int hd) {
    const int C = H * hd;
    const long total = (long)B * R * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int e = (int)(i % C), h = e / hd;
        const int r = (int)((i / C) % R), b = (int)(i / ((long)C * R));
        const T *ap = att + ((long)b * Tn * R + r) * H + h;
        const T *gp = g + (long)b * Tn * C + e;
        float acc = 0.f;
        for (int t = 0; t < Tn; ++t) acc += to_f32(ap[(long)t * R * H]) * to_f32(gp[(long)t * C]);
        drefv[i] = acc;
    }
}

bool shape_ok(int B, int nwin, int R, int H, int hd) { return B > 0 && nwin > 0 && R > 0 && R <= 128 && H > 0 && H <= 64 && hd > 0 && hd <= 64; }
int blocks_for(long n, int per, int cap) {
    long b = (n + per - 1) / per;
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace

// q: (B*nwin, 49, H, hd) strided operand; ref_k (B, R, H*hd); ra OUT (B, nwin*49, R, H), all `dtype`.
extern "C" int gwd_ref_scores_forward(const gwd_strided *q, const void *ref_k, void *ra, int32_t B, int32_t nwin, int32_t R, int32_t H,
                                      int32_t hd, float scale, int32_t dtype, void *stream) {
    if (!q || !q->p || !ref_k || !ra || !shape_ok(B, nwin, R, H, hd)) return -1;
    const QOp qo{q->p, q->ws, q->ts, q->hs};
    const long total = (long)B * nwin * 49 * R * H;
    const int grid = blocks_for(total, 256, 16384);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GWD_BF16) ref_scores_fwd_kernel<__bf16><<<grid, 256, 0, s>>>(qo, (const __bf16 *)ref_k, (__bf16 *)ra, B, nwin, R, H, hd, scale);
    else if (dtype == GWD_F32) ref_scores_fwd_kernel<float><<<grid, 256, 0, s>>>(qo, (const float *)ref_k, (float *)ra, B, nwin, R, H, hd, scale);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

// g (B, nwin*49, R, H) -> dq (strided operand like q, every element written) and d_ref_k fp32 (B, R, H*hd), overwritten.
extern "C" int gwd_ref_scores_backward(const gwd_strided *q, const void *ref_k, const void *g, const gwd_strided *dq, float *d_ref_k,
                                       int32_t B, int32_t nwin, int32_t R, int32_t H, int32_t hd, float scale, int32_t dtype,
                                       void *stream) {
    if (!q || !q->p || !dq || !dq->p || !ref_k || !g || !d_ref_k || !shape_ok(B, nwin, R, H, hd)) return -1;
    const QOp qo{q->p, q->ws, q->ts, q->hs};
    const QOpW dqo{dq->p, dq->ws, dq->ts, dq->hs};
    const int dq_blocks = blocks_for((long)B * nwin * 49 * H * hd, 256, 8192);
    const int dk_blocks = blocks_for((long)B * R * H * hd, 64, 8192);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GWD_BF16)
        ref_scores_bwd_kernel<__bf16><<<dq_blocks + dk_blocks, 256, 0, s>>>(qo, (const __bf16 *)ref_k, (const __bf16 *)g, dqo, d_ref_k, B, nwin, R, H, hd, scale, dq_blocks);
    else if (dtype == GWD_F32)
        ref_scores_bwd_kernel<float><<<dq_blocks + dk_blocks, 256, 0, s>>>(qo, (const float *)ref_k, (const float *)g, dqo, d_ref_k, B, nwin, R, H, hd, scale, dq_blocks);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

// ra (B, T, R, H), ref_v (B, R, H*hd) -> q_new (B, T, H*hd) [= (B*nwin, 49, H, hd)], att (B, T, R, H) or NULL.
extern "C" int gwd_ref_mix_forward(const void *ra, const void *ref_v, void *q_new, void *att, int32_t B, int32_t T, int32_t R, int32_t H,
                                   int32_t hd, int32_t dtype, void *stream) {
    if (!ra || !ref_v || !q_new || T <= 0 || !shape_ok(B, 1, R, H, hd)) return -1;
    const unsigned grid = (unsigned)((long)B * T);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)R * H * sizeof(float);
    if (dtype == GWD_BF16) ref_mix_fwd_kernel<__bf16><<<grid, 256, lds, s>>>((const __bf16 *)ra, (const __bf16 *)ref_v, (__bf16 *)q_new, (__bf16 *)att, R, H, hd, T);
    else if (dtype == GWD_F32) ref_mix_fwd_kernel<float><<<grid, 256, lds, s>>>((const float *)ra, (const float *)ref_v, (float *)q_new, (float *)att, R, H, hd, T);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

// att (B, T, R, H) from the forward, g (B, T, H*hd) -> d_ra (B, T, R, H) and d_ref_v fp32 (B, R, H*hd), both overwritten.
extern "C" int gwd_ref_mix_backward(const void *att, const void *ref_v, const void *g, void *d_ra, float *d_ref_v, int32_t B, int32_t T,
                                    int32_t R, int32_t H, int32_t hd, int32_t dtype, void *stream) {
    if (!att || !ref_v || !g || !d_ra || !d_ref_v || T <= 0 || !shape_ok(B, 1, R, H, hd)) return -1;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = ((size_t)H * hd + (size_t)R * H + H) * sizeof(float);
    const int vb = blocks_for((long)B * R * H * hd, 64, 8192);
    if (dtype == GWD_BF16) {
        ref_mix_bwd_kernel<__bf16><<<(unsigned)((long)B * T), 256, lds, s>>>((const __bf16 *)att, (const __bf16 *)ref_v, (const __bf16 *)g, (__bf16 *)d_ra, R, H, hd, T);
        ref_mix_bwd_v_kernel<__bf16><<<vb, 64, 0, s>>>((const __bf16 *)att, (const __bf16 *)g, d_ref_v, B, T, R, H, hd);
    } else if (dtype == GWD_F32) {
        ref_mix_bwd_kernel<float><<<(unsigned)((long)B * T), 256, lds, s>>>((const float *)att, (const float *)ref_v, (const float *)g, (float *)d_ra, R, H, hd, T);
        ref_mix_bwd_v_kernel<float><<<vb, 64, 0, s>>>((const float *)att, (const float *)g, d_ref_v, B, T, R, H, hd);
    } else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}
