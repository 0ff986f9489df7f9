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

template <typename T, int HD>
__device__ __forceinline__ void load_row(float (&v)[HD], const T *p) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int c = 0; c < HD; c += 8) {
            const bf16x8 x = *(const bf16x8 *)(p + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c + j] = (float)x[j];
        }
    } else {
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
            const float4 x = *(const float4 *)(p + c);
            v[c] = x.x, v[c + 1] = x.y, v[c + 2] = x.z, v[c + 3] = x.w;
        }
    }
}

// ra[b][t][r][h] = scale * q[b,t,h,:] . refk[b,r,h,:]: one thread per (b, t, h) keeps its query row in registers and walks the
// R reference rows (the same for every token of the image: L1-resident)
template <typename T, int HD>
__global__ void ref_scores_fwd_kernel(QOp q, const T *__restrict__ refk, T *__restrict__ ra, int B, int nwin, int R, int H, float scale) {
    const long total = (long)B * nwin * 49 * H;
    const int C = H * HD, Tn = nwin * 49;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int h = (int)(i % H);
        const long bt = i / H;
        const int t = (int)(bt % Tn), b = (int)(bt / Tn);
        float qv[HD];
        load_row<T, HD>(qv, (const T *)q.p + ((long)b * nwin + t / 49) * q.ws + (long)(t % 49) * q.ts + (long)h * q.hs);
        const T *kp = refk + (long)b * R * C + h * HD;
        T *dst = ra + bt * R * H + h;
        for (int r = 0; r < R; ++r) {
            float kv[HD];
            load_row<T, HD>(kv, kp + (long)r * C);
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc += qv[d] * kv[d];
            dst[(long)r * H] = from_f32<T>(acc * scale);
        }
    }
}

// dq[b,t,h,:] = scale * sum_r g[b,t,r,h] * refk[b,r,h,:]: one thread per (b, t, h), HD accumulators
template <typename T, int HD>
__global__ void ref_scores_dq_kernel(const T *__restrict__ refk, const T *__restrict__ g, QOpW dq, int B, int nwin, int R, int H, float scale) {
    const long total = (long)B * nwin * 49 * H;
    const int C = H * HD, Tn = nwin * 49;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int h = (int)(i % H);
        const long bt = i / H;
        const int t = (int)(bt % Tn), b = (int)(bt / Tn);
        const T *gp = g + bt * R * H + h;
        const T *kp = refk + (long)b * R * C + h * HD;
        float acc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = 0.f;
        for (int r = 0; r < R; ++r) {
            float kv[HD];
            load_row<T, HD>(kv, kp + (long)r * C);
            const float gv = to_f32(gp[(long)r * H]);
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] += gv * kv[d];
        }
        T *dst = (T *)dq.p + ((long)b * nwin + t / 49) * dq.ws + (long)(t % 49) * dq.ts + (long)h * dq.hs;
#pragma unroll
        for (int d = 0; d < HD; ++d) dst[d] = from_f32<T>(acc[d] * scale);
    }
}

// out[b][r][h][:] = scale * sum_t a[b,t,r,h] * x[b,t,h,:]   (d ref_k: a = g, x = q;  d ref_v: a = att, x = g of q_new)
// One WAVE per (b, r, h): the lanes split the tokens, HD accumulators each, then a cross-lane sum - no atomics.
template <typename T, int HD>
__global__ void token_reduce_kernel(const T *__restrict__ a, QOp x, float *__restrict__ out, int B, int nwin, int R, int H, float scale) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, total = (long)B * R * H;
    if (wave >= total) return;                       // whole waves leave together
    const int h = (int)(wave % H), r = (int)((wave / H) % R), b = (int)(wave / ((long)H * R));
    const int Tn = nwin * 49;
    const T *ap = a + ((long)b * Tn * R + r) * H + h;
    float acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
    for (int t = lane; t < Tn; t += 64) {
        float xv[HD];
        load_row<T, HD>(xv, (const T *)x.p + ((long)b * nwin + t / 49) * x.ws + (long)(t % 49) * x.ts + (long)h * x.hs);
        const float av = to_f32(ap[(long)t * R * H]);
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] += av * xv[d];
    }
    float *dst = out + ((long)b * R + r) * H * HD + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        const float s = wave_sum(acc[d]);
        if (lane == (d & 63)) dst[d] = s * scale;
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
    {   // softmax over r for every head with ALL threads: thread (h, part) owns the references part, part + parts, ...
        float *red = sm + RH;                                    // [parts][H]
        const int parts = blockDim.x / H, hh = threadIdx.x % H, pp = threadIdx.x / H;
        float m = -INFINITY;
        if (pp < parts)
            for (int r = pp; r < R; r += parts) m = fmaxf(m, sm[r * H + hh]);
        if (pp < parts) red[pp * H + hh] = m;
        __syncthreads();
        for (int q = 0; q < parts; ++q) m = fmaxf(m, red[q * H + hh]);
        __syncthreads();
        float l = 0.f;
        if (pp < parts)
            for (int r = pp; r < R; r += parts) {
                const float p = __expf(sm[r * H + hh] - m);
                sm[r * H + hh] = p;
                l += p;
            }
        if (pp < parts) red[pp * H + hh] = l;
        __syncthreads();
        l = 0.f;
        for (int q = 0; q < parts; ++q) l += red[q * H + hh];
        const float inv = 1.0f / l;
        if (pp < parts)
            for (int r = pp; r < R; r += parts) sm[r * H + hh] *= inv;
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
    const int GP = hd + 1;                                        // padded head pitch: lanes of one wave read gs at stride hd (16 heads, same bank)
    float *gs = sm, *da = sm + H * GP, *dot = da + RH;
    for (int e = threadIdx.x; e < C; e += blockDim.x) gs[(e / hd) * GP + e % hd] = to_f32(g[bt * C + e]);
    __syncthreads();
    const T *vb = refv + (bt / Tn) * R * C;
    constexpr int VE = 16 / (int)sizeof(T);
    const bool vec = (hd % VE) == 0 && ((uintptr_t)vb % 16) == 0;          // 16-byte pieces of a head's row (was: hd scalar loads)
    for (int e = threadIdx.x; e < RH; e += blockDim.x) {
        const int r = e / H, h = e % H;
        const T *vp = vb + (long)r * C + h * hd;
        float acc = 0.f;
        if (vec) {
            for (int d = 0; d < hd; d += VE) {
                const uint4 raw = *(const uint4 *)(vp + d);
                const T *pv = (const T *)&raw;
#pragma unroll
                for (int k = 0; k < VE; ++k) acc += gs[h * GP + d + k] * to_f32(pv[k]);
            }
        } else {
            for (int d = 0; d < hd; ++d) acc += gs[h * GP + d] * to_f32(vp[d]);
        }
        da[e] = acc;
    }
    __syncthreads();
    // dot[h] = sum_r att * datt: every thread takes a few r of one head, the partial sums meet in LDS (was: H threads walking all R
    // references with dependent global loads while the other 240 waited)
    float *part = dot + H;                                       // [blockDim / H][H]
    const int parts = blockDim.x / H, hh = threadIdx.x % H, pp = threadIdx.x / H;
    float av[4];                                                 // this thread's att values (R <= 128, parts >= 4 x ... see launch)
    float s = 0.f;
    int n = 0;
    if (pp < parts)
        for (int r = pp; r < R; r += parts, ++n) {
            const float a = to_f32(att[bt * RH + r * H + hh]);
            if (n < 4) av[n] = a;
            s += a * da[r * H + hh];
        }
    if (pp < parts) part[pp * H + hh] = s;
    __syncthreads();
    if (threadIdx.x < H) {
        float t = 0.f;
        for (int q = 0; q < parts; ++q) t += part[q * H + threadIdx.x];
        dot[threadIdx.x] = t;
    }
    __syncthreads();
    n = 0;
    if (pp < parts)
        for (int r = pp; r < R; r += parts, ++n) {
            const float a = n < 4 ? av[n] : to_f32(att[bt * RH + r * H + hh]);
            dra[bt * RH + r * H + hh] = from_f32<T>(a * (da[r * H + hh] - dot[hh]));
        }
}

bool shape_ok(int B, int nwin, int R, int H, int hd) { return B > 0 && nwin > 0 && R > 0 && R <= 128 && H > 0 && H <= 64 && hd > 0 && hd <= 64; }
int blocks_for(long n, int per, int cap) {
    long b = (n + per - 1) / per;
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace

#define HD_SWITCH(hd, CALL)            \
    switch (hd) {                      \
        case 8: { CALL(8) } break;     \
        case 16: { CALL(16) } break;   \
        case 32: { CALL(32) } break;   \
        case 64: { CALL(64) } break;   \
        default: return -4;            \
    }

bool q_aligned(const gwd_strided *q, int32_t dtype) {
    const int a = dtype == GWD_BF16 ? 8 : 4;          // 16-byte row pieces
    return (uintptr_t)q->p % 16 == 0 && q->ws % a == 0 && q->ts % a == 0 && q->hs % a == 0;
}

// q: (B*nwin, 49, H, hd) strided operand; ref_k (B, R, H*hd); ra OUT (B, nwin*49, R, H), all `dtype`.  hd in {8, 16, 32, 64}.
extern "C" int gwd_ref_scores_forward(const gwd_strided *q, const void *ref_k, void *ra, int32_t B, int32_t nwin, int32_t R, int32_t H,
                                      int32_t hd, float scale, int32_t dtype, void *stream) {
    if (!q || !q->p || !ref_k || !ra || !shape_ok(B, nwin, R, H, hd)) return -1;
    if (!q_aligned(q, dtype) || (uintptr_t)ref_k % 16) return -5;
    const QOp qo{q->p, q->ws, q->ts, q->hs};
    const int grid = blocks_for((long)B * nwin * 49 * H, 64, 16384);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GWD_BF16) {
#define CALL(HD_) ref_scores_fwd_kernel<__bf16, HD_><<<grid, 64, 0, s>>>(qo, (const __bf16 *)ref_k, (__bf16 *)ra, B, nwin, R, H, scale);
        HD_SWITCH(hd, CALL)
#undef CALL
    } else if (dtype == GWD_F32) {
#define CALL(HD_) ref_scores_fwd_kernel<float, HD_><<<grid, 64, 0, s>>>(qo, (const float *)ref_k, (float *)ra, B, nwin, R, H, scale);
        HD_SWITCH(hd, CALL)
#undef CALL
    } else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

// g (B, nwin*49, R, H) -> dq (strided operand like q, every element written) and d_ref_k fp32 (B, R, H*hd), overwritten.
extern "C" int gwd_ref_scores_backward(const gwd_strided *q, const void *ref_k, const void *g, const gwd_strided *dq, float *d_ref_k,
                                       int32_t B, int32_t nwin, int32_t R, int32_t H, int32_t hd, float scale, int32_t dtype,
                                       void *stream) {
    if (!q || !q->p || !dq || !dq->p || !ref_k || !g || !d_ref_k || !shape_ok(B, nwin, R, H, hd)) return -1;
    if (!q_aligned(q, dtype) || (uintptr_t)ref_k % 16) return -5;
    const QOp qo{q->p, q->ws, q->ts, q->hs};
    const QOpW dqo{dq->p, dq->ws, dq->ts, dq->hs};
    const int grid = blocks_for((long)B * nwin * 49 * H, 64, 16384);
    const unsigned rgrid = (unsigned)(((long)B * R * H + 3) / 4);          // 4 waves per workgroup, one (b, r, h) each
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GWD_BF16) {
#define CALL(HD_)                                                                                                             \
    ref_scores_dq_kernel<__bf16, HD_><<<grid, 64, 0, s>>>((const __bf16 *)ref_k, (const __bf16 *)g, dqo, B, nwin, R, H, scale);    \
    token_reduce_kernel<__bf16, HD_><<<rgrid, 256, 0, s>>>((const __bf16 *)g, qo, d_ref_k, B, nwin, R, H, scale);
        HD_SWITCH(hd, CALL)
#undef CALL
    } else if (dtype == GWD_F32) {
#define CALL(HD_)                                                                                                             \
    ref_scores_dq_kernel<float, HD_><<<grid, 64, 0, s>>>((const float *)ref_k, (const float *)g, dqo, B, nwin, R, H, scale);       \
    token_reduce_kernel<float, HD_><<<rgrid, 256, 0, s>>>((const float *)g, qo, d_ref_k, B, nwin, R, H, scale);
        HD_SWITCH(hd, CALL)
#undef CALL
    } else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

// ra (B, T, R, H), ref_v (B, R, H*hd) -> q_new (B, T, H*hd) [= (B*nwin, 49, H, hd)], att (B, T, R, H) or NULL.
extern "C" int gwd_ref_mix_forward(const void *ra, const void *ref_v, void *q_new, void *att, int32_t B, int32_t T, int32_t R, int32_t H,
                                   int32_t hd, int32_t dtype, void *stream) {
    if (!ra || !ref_v || !q_new || T <= 0 || !shape_ok(B, 1, R, H, hd)) return -1;
    const unsigned grid = (unsigned)((long)B * T);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = ((size_t)R * H + (size_t)(256 / H) * H) * sizeof(float);
    if (dtype == GWD_BF16) ref_mix_fwd_kernel<__bf16><<<grid, 256, lds, s>>>((const __bf16 *)ra, (const __bf16 *)ref_v, (__bf16 *)q_new, (__bf16 *)att, R, H, hd, T);
    else if (dtype == GWD_F32) ref_mix_fwd_kernel<float><<<grid, 256, lds, s>>>((const float *)ra, (const float *)ref_v, (float *)q_new, (float *)att, R, H, hd, T);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

// att (B, T, R, H) from the forward, g (B, T, H*hd) -> d_ra (B, T, R, H) and d_ref_v fp32 (B, R, H*hd), both overwritten.
// T must be a multiple of 49 (tokens of whole windows).
extern "C" int gwd_ref_mix_backward(const void *att, const void *ref_v, const void *g, void *d_ra, float *d_ref_v, int32_t B, int32_t T,
                                    int32_t R, int32_t H, int32_t hd, int32_t dtype, void *stream) {
    if (!att || !ref_v || !g || !d_ra || !d_ref_v || T <= 0 || T % 49 || !shape_ok(B, 1, R, H, hd)) return -1;
    if ((uintptr_t)g % 16) return -5;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = ((size_t)H * (hd + 1) + (size_t)R * H + H + (size_t)(256 / H) * H) * sizeof(float);
    const unsigned rgrid = (unsigned)(((long)B * R * H + 3) / 4);
    const long C = (long)H * hd;
    const QOp go{g, 49 * C, C, hd};                    // g of q_new viewed as the (B*nwin, 49, H, hd) operand
    if (dtype == GWD_BF16) {
        ref_mix_bwd_kernel<__bf16><<<(unsigned)((long)B * T), 256, lds, s>>>((const __bf16 *)att, (const __bf16 *)ref_v, (const __bf16 *)g, (__bf16 *)d_ra, R, H, hd, T);
#define CALL(HD_) token_reduce_kernel<__bf16, HD_><<<rgrid, 256, 0, s>>>((const __bf16 *)att, go, d_ref_v, B, T / 49, R, H, 1.0f);
        HD_SWITCH(hd, CALL)
#undef CALL
    } else if (dtype == GWD_F32) {
        ref_mix_bwd_kernel<float><<<(unsigned)((long)B * T), 256, lds, s>>>((const float *)att, (const float *)ref_v, (const float *)g, (float *)d_ra, R, H, hd, T);
#define CALL(HD_) token_reduce_kernel<float, HD_><<<rgrid, 256, 0, s>>>((const float *)att, go, d_ref_v, B, T / 49, R, H, 1.0f);
        HD_SWITCH(hd, CALL)
#undef CALL
    } else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}
