// Fused multi-head attention FORWARD core for head_dim 32 and up to 320 keys - the DETR encoder / decoder attention
// (src/models/multi_head_attention.py:329-372: q*scaling, q k^T, key-padding mask -inf, softmax, dropout, . v, head merge).
// Replaces per attention: two batched GEMMs, the softmax and dropout kernels and the four head split / merge copies
// (q, k, v, output are read / written in their (batch, token, heads*32) projection layout, row strides given).
//
// These products are tiny (300 x 300 x 32 per head): latency, not FLOPs, is what the seven launches cost.  One workgroup =
// 32 queries of one (image, head); K and V of the head sit in LDS (80-byte rows: conflict-free 16-byte reads for 16
// consecutive lanes); a wave takes one query at a time with lane = key (5 keys per lane), so the softmax reductions are
// two DPP ladders and the probabilities never leave registers before they are written once for the backward pass.
#include "common.h"

namespace {

constexpr int HD = 32, MAXKPL = 5, MAXS = 64 * MAXKPL, QB = 32;

template <typename T>
__device__ __forceinline__ void load_row(const T *__restrict__ p, float (&r)[HD]) {
    constexpr int VEC = 16 / (int)sizeof(T);
#pragma unroll
    for (int c = 0; c < HD / VEC; ++c) {
        const uint4 raw = *(const uint4 *)(p + c * VEC);
        const T *e = (const T *)&raw;
#pragma unroll
        for (int i = 0; i < VEC; ++i) r[c * VEC + i] = to_f32(e[i]);
    }
}

// KPL = keys per lane, a compile-time bound >= ceil(S / 64): 2 for the decoder's 100 queries, 5 for the 300 encoder tokens
template <typename T, int KPL>
__global__ __launch_bounds__(256) void mha_fwd_kernel(const T *__restrict__ q, const T *__restrict__ k, const T *__restrict__ v,
                                                      int64_t q_rs, int64_t k_rs, int64_t v_rs,
                                                      const unsigned char *__restrict__ kpm, const T *__restrict__ mult,
                                                      T *__restrict__ P, T *__restrict__ out, int H, int L, int S, float scale) {
    constexpr int ROW = HD + 16 / (int)sizeof(T);            // LDS row in elements: 80 B (bf16) / 144 B (fp32)
    constexpr int VEC = 16 / (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    T *Ks = (T *)lds_raw;
    T *Vs = Ks + (size_t)(64 * KPL) * ROW;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- K and V of this head -> LDS (rows beyond S are never read)
    for (int c = tid; c < S * (HD / VEC); c += 256) {
        const int row = c / (HD / VEC), ch = c - row * (HD / VEC);
        *(uint4 *)(Ks + (size_t)row * ROW + ch * VEC) = *(const uint4 *)(k + ((int64_t)b * S + row) * k_rs + h * HD + ch * VEC);
        *(uint4 *)(Vs + (size_t)row * ROW + ch * VEC) = *(const uint4 *)(v + ((int64_t)b * S + row) * v_rs + h * HD + ch * VEC);
    }
    __syncthreads();
    bool ok[KPL], live[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int key = lane + 64 * j;
        ok[j] = key < S;
        live[j] = ok[j] && !(kpm && kpm[(int64_t)b * S + key]);
    }
    const int l0 = blockIdx.x * QB + wave * (QB / 4);
    for (int i = 0; i < QB / 4; ++i) {
        const int l = l0 + i;
        if (l >= L) break;                                    // wave-uniform
        float qr[HD];
        load_row(q + ((int64_t)b * L + l) * q_rs + h * HD, qr);
#pragma unroll
        for (int d = 0; d < HD; ++d) qr[d] *= scale;          // multi_head_attention.py:329 (q = q * scaling)
        float sc[KPL];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            sc[j] = -INFINITY;
            if (live[j]) {
                float kr[HD];
                load_row(Ks + (size_t)(lane + 64 * j) * ROW, kr);
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) s += qr[d] * kr[d];
                sc[j] = s;
            }
            mx = fmaxf(mx, sc[j]);
        }
        mx = wave_max_uniform(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            sc[j] = (sc[j] == -INFINITY) ? 0.f : expf(sc[j] - mx);
            sum += sc[j];
        }
        const float inv = 1.0f / wave_sum_uniform(sum);
        const int64_t prow = ((int64_t)bh * L + l) * S;
        float acc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = 0.f;
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            if (!ok[j]) continue;
            const int key = lane + 64 * j;
            const T pt = from_f32<T>(sc[j] * inv);
            P[prow + key] = pt;
            float pd = to_f32(pt);                             // the value the backward pass will read
            if (mult) pd *= to_f32(mult[prow + key]);          // dropout multiplier: 0 or 1/(1-p)
            if (pd != 0.f) {
                float vr[HD];
                load_row(Vs + (size_t)key * ROW, vr);
#pragma unroll
                for (int d = 0; d < HD; ++d) acc[d] += pd * vr[d];
            }
        }
        float mine = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            const float s = wave_sum_uniform(acc[d]);
            if (lane == d) mine = s;
        }
        if (lane < HD) out[((int64_t)b * L + l) * ((int64_t)H * HD) + h * HD + lane] = from_f32<T>(mine);
    }
}

}  // namespace

extern "C" int gwd_mha_forward(const void *q, const void *k, const void *v, int64_t q_rs, int64_t k_rs, int64_t v_rs,
                               const uint8_t *key_padding_mask, const void *mult, void *P, void *out, int32_t B, int32_t H,
                               int32_t L, int32_t S, float scale, int32_t dtype, void *stream) {
    if (!q || !k || !v || !P || !out || B <= 0 || H <= 0 || L <= 0 || S <= 0) return -1;
    if (S > MAXS) return -4;                                   // caller keeps the batched-GEMM path
    if ((int64_t)B * H > 65535) return -7;
    const int esz = dtype == GWD_BF16 ? 2 : 4;
    if (dtype != GWD_BF16 && dtype != GWD_F32) return -2;
    if ((q_rs * esz) % 16 || (k_rs * esz) % 16 || (v_rs * esz) % 16 ||
        ((uintptr_t)q % 16) || ((uintptr_t)k % 16) || ((uintptr_t)v % 16)) return -5;      // 16-byte row loads
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((L + QB - 1) / QB, B * H);
    const int kpl = S <= 128 ? 2 : MAXKPL;
    const size_t lds = (size_t)2 * 64 * kpl * (HD + 16 / esz) * esz;
#define MHA_LAUNCH(T_, K_)                                                                                                     \
    do {                                                                                                                       \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)mha_fwd_kernel<T_, K_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        mha_fwd_kernel<T_, K_><<<grid, 256, lds, s>>>((const T_ *)q, (const T_ *)k, (const T_ *)v, q_rs, k_rs, v_rs, key_padding_mask, \
                                                      (const T_ *)mult, (T_ *)P, (T_ *)out, H, L, S, scale);                    \
    } while (0)
    if (dtype == GWD_BF16) {
        if (kpl == 2) MHA_LAUNCH(__bf16, 2); else MHA_LAUNCH(__bf16, MAXKPL);
    } else {
        if (kpl == 2) MHA_LAUNCH(float, 2); else MHA_LAUNCH(float, MAXKPL);
    }
#undef MHA_LAUNCH
    GWD_CHECK_LAUNCH();
    return 0;
}
