// Fused 7x7 window attention (N = 49 tokens, head_dim 4/8/16/32), forward and backward.
//
//   S = scale * Q K^T + rel_pos_bias[head] (+ shift mask: -100 where the two tokens come from different
//   regions of the cyclic shift);  P = softmax(S);  O = P V
//
// One 64-lane wave owns one (window, head) problem at a time: lane i is query row i in the forward and in
// backward pass A (dQ, dBias), lane j is key row j in backward pass B (dK, dV), so every reduction is
// lane-local and the 49x49 score matrix only ever lives in registers.  K/V (and Q/dO in the backward) tiles
// of the head sit in LDS as fp32 and are read as broadcasts.  A workgroup (4 waves) works on ONE head
// (blockIdx.y) so the dense bias of that head is staged in LDS once and dBias is accumulated in registers
// across all windows the wave visits, then reduced through LDS and flushed with 2401 atomics per workgroup.
// head_dim <= 32 and FLOPs are tiny (9.6 kFLOP * hd per problem): this is a latency/HBM kernel, VALU math.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int NT = 49;   // tokens per window
constexpr int NB = 2404; // floats reserved for one 49x49 bias copy in LDS (padded so the tiles behind it stay 16-byte aligned)

struct Operand {
    const void *p;
    long ws, ts, hs;     // window / token / head strides in elements
};
// where bias(h, i, j) lives: dense [heads][49][49], or the [n_rel][heads] parameter table behind rel[i*49+j]
struct BiasRef {
    const int *rel;
    int heads;
    int n_rel, grad_hm;          // grad_hm: the table-shaped GRADIENT is head-major [heads][n_rel] (gwd_winattn_backward's dbias_head_major)
    __device__ __forceinline__ long at(int h, int e) const { return rel ? (long)rel[e] * heads + h : (long)h * (NT * NT) + e; }
    __device__ __forceinline__ long grad_at(int h, int e) const { return (rel && grad_hm) ? (long)h * n_rel + rel[e] : at(h, e); }
};
struct OperandW {
    void *p;
    long ws, ts, hs;
};

template <typename T> struct Quad;
template <> struct Quad<float> { typedef float4 type; };
template <> struct Quad<__bf16> { typedef uint2 type; };

// VEC: every operand's base and strides are multiples of 4 elements (checked on the host) -> 4 elements per access
template <typename T, int HD, bool VEC>
__device__ __forceinline__ void load_tile(float *dst, const Operand &op, long w, int h, int lane) {
    const T *base = (const T *)op.p + w * op.ws + h * op.hs;
    if constexpr (VEC) {
        for (int e = lane * 4; e < NT * HD; e += 256) {
            const int t = e / HD, d = e - t * HD;
            const typename Quad<T>::type raw = *(const typename Quad<T>::type *)(base + t * op.ts + d);
            const T *v = (const T *)&raw;
            *(float4 *)(dst + e) = make_float4(to_f32(v[0]), to_f32(v[1]), to_f32(v[2]), to_f32(v[3]));
        }
    } else {
        for (int e = lane; e < NT * HD; e += 64) {
            const int t = e / HD, d = e - t * HD;
            dst[e] = to_f32(base[t * op.ts + d]);
        }
    }
}

template <typename T, int HD, bool VEC>
__device__ __forceinline__ void store_row(T *p, const float (&v)[HD], float mul) {
    if constexpr (VEC) {
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            typename Quad<T>::type raw;
            T *e = (T *)&raw;
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = from_f32<T>(v[d + j] * mul);
            *(typename Quad<T>::type *)(p + d) = raw;
        }
    } else {
#pragma unroll
        for (int d = 0; d < HD; ++d) p[d] = from_f32<T>(v[d] * mul);
    }
}

template <typename T, int HD, bool VEC>
__global__ __launch_bounds__(256) void winattn_fwd_kernel(Operand q, Operand k, Operand v, OperandW o,
                                                          const float *__restrict__ bias, BiasRef br, const int *__restrict__ region,
                                                          long n_windows, int windows_per_image, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *bias_s = smem;                                  // [49*49]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = blockIdx.y;
    float *ks = smem + NB + wave * (2 * NT * HD + 64);
    float *vs = ks + NT * HD;
    int *reg_s = (int *)(vs + NT * HD);
    for (int e = threadIdx.x; e < NT * NT; e += blockDim.x) bias_s[e] = bias[br.at(h, e)];
    __syncthreads();
    const int i = lane < NT ? lane : NT - 1;               // idle lanes shadow the last row (keeps loops uniform)
    for (long w = (long)blockIdx.x * 4 + wave; w < n_windows; w += (long)gridDim.x * 4) {
        load_tile<T, HD, VEC>(ks, k, w, h, lane);
        load_tile<T, HD, VEC>(vs, v, w, h, lane);
        if (region) reg_s[lane] = lane < NT ? region[(w % windows_per_image) * NT + lane] : 0;
        float qr[HD];
        {
            const T *qp = (const T *)q.p + w * q.ws + i * q.ts + h * q.hs;
            if constexpr (VEC) {
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    const typename Quad<T>::type raw = *(const typename Quad<T>::type *)(qp + d);
                    const T *v4 = (const T *)&raw;
#pragma unroll
                    for (int j = 0; j < 4; ++j) qr[d + j] = to_f32(v4[j]) * scale;
                }
            } else {
#pragma unroll
                for (int d = 0; d < HD; ++d) qr[d] = to_f32(qp[d]) * scale;
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes are visible to itself
        const int my_reg = region ? reg_s[i] : 0;
        float s[NT];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float a = bias_s[i * NT + j];
#pragma unroll
            for (int d = 0; d < HD; ++d) a += qr[d] * ks[j * HD + d];
            if (region && reg_s[j] != my_reg) a += -100.0f;
            s[j] = a;
            mx = fmaxf(mx, a);
        }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            s[j] = __expf(s[j] - mx);
            l += s[j];
        }
        const float inv = 1.0f / l;
        float acc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] += s[j] * vs[j * HD + d];
        if (lane < NT) {
            T *op = (T *)o.p + w * o.ws + i * o.ts + h * o.hs;
            store_row<T, HD, VEC>(op, acc, inv);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename T, int HD, bool VEC>
__global__ __launch_bounds__(256) void winattn_bwd_kernel(Operand q, Operand k, Operand v, Operand go, OperandW gq,
                                                          OperandW gk, OperandW gv, const float *__restrict__ bias,
                                                          float *__restrict__ dbias, BiasRef br, const int *__restrict__ region,
                                                          long n_windows, int windows_per_image, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *bias_s = smem;                                  // [49*49]
    float *dbias_s = smem + NB;                            // [49*49]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = blockIdx.y;
    float *qs = smem + 2 * NB + wave * (4 * NT * HD + 4 * 64);
    float *ks = qs + NT * HD, *vs = ks + NT * HD, *os = vs + NT * HD;
    float *m_s = os + NT * HD, *l_s = m_s + 64, *dl_s = l_s + 64;
    int *reg_s = (int *)(dl_s + 64);
    for (int e = threadIdx.x; e < NT * NT; e += blockDim.x) {
        bias_s[e] = bias[br.at(h, e)];
        dbias_s[e] = 0.f;
    }
    __syncthreads();
    const int i = lane < NT ? lane : NT - 1;
    const bool live = lane < NT;
    float db[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) db[j] = 0.f;

    for (long w = (long)blockIdx.x * 4 + wave; w < n_windows; w += (long)gridDim.x * 4) {
        load_tile<T, HD, VEC>(qs, q, w, h, lane);
        load_tile<T, HD, VEC>(ks, k, w, h, lane);
        load_tile<T, HD, VEC>(vs, v, w, h, lane);
        load_tile<T, HD, VEC>(os, go, w, h, lane);
        if (region) reg_s[lane] = lane < NT ? region[(w % windows_per_image) * NT + lane] : 0;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const int my_reg = region ? reg_s[i] : 0;
        // ---------------- pass A: lane = query row i -> dQ_i, dBias row i, softmax statistics
        float qr[HD], dor[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            qr[d] = qs[i * HD + d] * scale;
            dor[d] = os[i * HD + d];
        }
        float s[NT];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float a = bias_s[i * NT + j];
#pragma unroll
            for (int d = 0; d < HD; ++d) a += qr[d] * ks[j * HD + d];
            if (region && reg_s[j] != my_reg) a += -100.0f;
            s[j] = a;
            mx = fmaxf(mx, a);
        }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            s[j] = __expf(s[j] - mx);
            l += s[j];
        }
        const float inv = 1.0f / l;
        // delta_i = sum_j P_ij dP_ij = dO_i . O_i  (keeps the 49 dP values out of registers)
        float oacc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) oacc[d] = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int d = 0; d < HD; ++d) oacc[d] += s[j] * vs[j * HD + d];
        float delta = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) delta += dor[d] * oacc[d];
        delta *= inv;
        float dq[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) dq[d] = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) a += dor[d] * vs[j * HD + d];
            const float ds = s[j] * inv * (a - delta);
            if (live) db[j] += ds;
#pragma unroll
            for (int d = 0; d < HD; ++d) dq[d] += ds * ks[j * HD + d];
        }
        if (live) {
            T *gp = (T *)gq.p + w * gq.ws + i * gq.ts + h * gq.hs;
            store_row<T, HD, VEC>(gp, dq, scale);
        }
        m_s[lane] = mx;
        l_s[lane] = inv;
        dl_s[lane] = delta;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // ---------------- pass B: lane = key row j -> dK_j, dV_j
        const int jj = i;
        float kr[HD], vr[HD], dk[HD], dv[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            kr[d] = ks[jj * HD + d];
            vr[d] = vs[jj * HD + d];
            dk[d] = 0.f;
            dv[d] = 0.f;
        }
#pragma unroll 7
        for (int r = 0; r < NT; ++r) {
            float a = bias_s[r * NT + jj], b = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                a += qs[r * HD + d] * scale * kr[d];
                b += os[r * HD + d] * vr[d];
            }
            if (region && reg_s[r] != my_reg) a += -100.0f;
            const float p = __expf(a - m_s[r]) * l_s[r];
            const float ds = p * (b - dl_s[r]);
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                dk[d] += ds * qs[r * HD + d];
                dv[d] += p * os[r * HD + d];
            }
        }
        if (live) {
            T *kp = (T *)gk.p + w * gk.ws + jj * gk.ts + h * gk.hs;
            T *vp = (T *)gv.p + w * gv.ws + jj * gv.ts + h * gv.hs;
            store_row<T, HD, VEC>(kp, dk, scale);
            store_row<T, HD, VEC>(vp, dv, 1.0f);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (dbias) {
        if (live) {
#pragma unroll
            for (int j = 0; j < NT; ++j) atomicAdd(&dbias_s[i * NT + j], db[j]);
        }
        __syncthreads();
        for (int e = threadIdx.x; e < NT * NT; e += blockDim.x) unsafeAtomicAdd(dbias + br.grad_at(h, e), dbias_s[e]);
    }
}

struct Args {
    Operand q, k, v, go;
    OperandW o, gq, gk, gv;
};

template <typename T, int HD>
int launch(bool backward, const Args &a, const float *bias, float *dbias, const BiasRef br, const int *region, long n_windows,
           int windows_per_image, int heads, float scale, hipStream_t s) {
    // a workgroup stages one head's bias (and, backward, flushes 2401 dBias atomics): give every wave ~8 windows so that
    // cost is amortised, as long as that still leaves >= 4 workgroups per CU
    long bx = (n_windows + 3) / 4;
    const long want = (n_windows + 31) / 32;
    if (want * heads >= 1024) bx = want;
    if (bx > 512) bx = 512;
    dim3 grid((unsigned)bx, heads);
    auto al = [](const void *p, long ws, long ts, long hs) {
        return (uintptr_t)p % (4 * sizeof(T)) == 0 && ws % 4 == 0 && ts % 4 == 0 && hs % 4 == 0;
    };
    bool vec = al(a.q.p, a.q.ws, a.q.ts, a.q.hs) && al(a.k.p, a.k.ws, a.k.ts, a.k.hs) && al(a.v.p, a.v.ws, a.v.ts, a.v.hs);
    if (!backward) {
        vec = vec && al(a.o.p, a.o.ws, a.o.ts, a.o.hs);
        const size_t lds = (NB + 4 * (2 * NT * HD + 64)) * sizeof(float);
        if (vec) winattn_fwd_kernel<T, HD, true><<<grid, 256, lds, s>>>(a.q, a.k, a.v, a.o, bias, br, region, n_windows, windows_per_image, scale);
        else winattn_fwd_kernel<T, HD, false><<<grid, 256, lds, s>>>(a.q, a.k, a.v, a.o, bias, br, region, n_windows, windows_per_image, scale);
    } else {
        vec = vec && al(a.go.p, a.go.ws, a.go.ts, a.go.hs) && al(a.gq.p, a.gq.ws, a.gq.ts, a.gq.hs) &&
              al(a.gk.p, a.gk.ws, a.gk.ts, a.gk.hs) && al(a.gv.p, a.gv.ws, a.gv.ts, a.gv.hs);
        const size_t lds = (2 * NB + 4 * (4 * NT * HD + 4 * 64)) * sizeof(float);
        if (vec) winattn_bwd_kernel<T, HD, true><<<grid, 256, lds, s>>>(a.q, a.k, a.v, a.go, a.gq, a.gk, a.gv, bias, dbias, br, region, n_windows, windows_per_image, scale);
        else winattn_bwd_kernel<T, HD, false><<<grid, 256, lds, s>>>(a.q, a.k, a.v, a.go, a.gq, a.gk, a.gv, bias, dbias, br, region, n_windows, windows_per_image, scale);
    }
    GWD_CHECK_LAUNCH();
    return 0;
}

template <typename T>
int dispatch_hd(int hd, bool backward, const Args &a, const float *bias, float *dbias, const BiasRef br, const int *region, long nw,
                int wpi, int heads, float scale, hipStream_t s) {
    switch (hd) {
        case 4: return launch<T, 4>(backward, a, bias, dbias, br, region, nw, wpi, heads, scale, s);
        case 8: return launch<T, 8>(backward, a, bias, dbias, br, region, nw, wpi, heads, scale, s);
        case 16: return launch<T, 16>(backward, a, bias, dbias, br, region, nw, wpi, heads, scale, s);
        case 32: return launch<T, 32>(backward, a, bias, dbias, br, region, nw, wpi, heads, scale, s);
        default: return -4;
    }
}

}  // namespace

// mfattn.hip: the same problem on the matrix cores (bf16); 0 = launched, 1 = not covered (fall through to the kernels above)
int gwd_mfattn_window(bool backward, const gwd_strided *q, const gwd_strided *k, const gwd_strided *v, const gwd_strided *o_or_go,
                      const gwd_strided *gq, const gwd_strided *gk, const gwd_strided *gv, const float *bias, float *dbias,
                      const int32_t *rel_index, int32_t n_rel, const int32_t *region, int64_t n_windows, int32_t wpi, int32_t heads,
                      int32_t head_dim, float scale, int32_t dbias_head_major, hipStream_t s);

static bool mfma_window_enabled() { return true; }      // bf16 with aligned operands: csrc/mfattn.hip; the lane-per-row kernels below are the fp32 (parity) path

extern "C" int gwd_winattn_forward(const gwd_strided *q, const gwd_strided *k, const gwd_strided *v, const gwd_strided *o,
                                   const float *bias, const int32_t *rel_index, int32_t n_rel, const int32_t *region,
                                   int64_t n_windows, int32_t windows_per_image, int32_t heads, int32_t head_dim, float scale,
                                   int32_t dtype, void *stream) {
    if (!q || !k || !v || !o || !q->p || !k->p || !v->p || !o->p || !bias || n_windows <= 0 || heads <= 0) return -1;
    if (rel_index && (n_rel <= 0 || n_rel > 256)) return -1;
    const BiasRef br{rel_index, heads, n_rel, 0};
    if (region && windows_per_image <= 0) return -1;
    Args a{};
    a.q = {q->p, q->ws, q->ts, q->hs};
    a.k = {k->p, k->ws, k->ts, k->hs};
    a.v = {v->p, v->ws, v->ts, v->hs};
    a.o = {o->p, o->ws, o->ts, o->hs};
    if (dtype == GWD_BF16 && mfma_window_enabled()) {
        const int rc = gwd_mfattn_window(false, q, k, v, o, nullptr, nullptr, nullptr, bias, nullptr, rel_index, n_rel, region, n_windows,
                                         windows_per_image, heads, head_dim, scale, 0, (hipStream_t)stream);
        if (rc <= 0) {
            if (rc == 0) GWD_CHECK_LAUNCH();
            return rc;
        }
    }
    if (dtype == GWD_BF16) return dispatch_hd<__bf16>(head_dim, false, a, bias, nullptr, br, region, n_windows, windows_per_image, heads, scale, (hipStream_t)stream);
    if (dtype == GWD_F32) return dispatch_hd<float>(head_dim, false, a, bias, nullptr, br, region, n_windows, windows_per_image, heads, scale, (hipStream_t)stream);
    return -2;
}

extern "C" int gwd_winattn_backward(const gwd_strided *q, const gwd_strided *k, const gwd_strided *v, const gwd_strided *go,
                                    const gwd_strided *gq, const gwd_strided *gk, const gwd_strided *gv, const float *bias,
                                    float *dbias, const int32_t *rel_index, int32_t n_rel, const int32_t *region, int64_t n_windows,
                                    int32_t windows_per_image, int32_t heads, int32_t head_dim, float scale, int32_t dbias_head_major,
                                    int32_t dtype, void *stream) {
    if (!q || !k || !v || !go || !gq || !gk || !gv || !bias || n_windows <= 0 || heads <= 0) return -1;
    if (rel_index && (n_rel <= 0 || n_rel > 256)) return -1;
    if (dbias_head_major && !rel_index) return -1;             // only the table-shaped gradient has a second layout
    const BiasRef br{rel_index, heads, n_rel, dbias_head_major ? 1 : 0};
    if (!q->p || !k->p || !v->p || !go->p || !gq->p || !gk->p || !gv->p) return -1;
    if (region && windows_per_image <= 0) return -1;
    Args a{};
    a.q = {q->p, q->ws, q->ts, q->hs};
    a.k = {k->p, k->ws, k->ts, k->hs};
    a.v = {v->p, v->ws, v->ts, v->hs};
    a.go = {go->p, go->ws, go->ts, go->hs};
    a.gq = {gq->p, gq->ws, gq->ts, gq->hs};
    a.gk = {gk->p, gk->ws, gk->ts, gk->hs};
    a.gv = {gv->p, gv->ws, gv->ts, gv->hs};
    if (dtype == GWD_BF16 && mfma_window_enabled()) {
        const int rc = gwd_mfattn_window(true, q, k, v, go, gq, gk, gv, bias, dbias, rel_index, n_rel, region, n_windows, windows_per_image,
                                         heads, head_dim, scale, dbias_head_major, (hipStream_t)stream);
        if (rc <= 0) {
            if (rc == 0) GWD_CHECK_LAUNCH();
            return rc;
        }
    }
    if (dtype == GWD_BF16) return dispatch_hd<__bf16>(head_dim, true, a, bias, dbias, br, region, n_windows, windows_per_image, heads, scale, (hipStream_t)stream);
    if (dtype == GWD_F32) return dispatch_hd<float>(head_dim, true, a, bias, dbias, br, region, n_windows, windows_per_image, heads, scale, (hipStream_t)stream);
    return -2;
}
