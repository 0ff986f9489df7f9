// HBM-bound row kernels: LayerNorm (+GELU) fwd/bwd, softmax fwd/bwd, activation backward, column sums.
// One 64-lane wave owns one row at a time (lane-strided, fully coalesced accesses, shuffle
// reductions, no LDS); grids are capped at ~8 blocks per CU and stride over rows.
#include "common.h"

namespace {

constexpr int MAX_PER_LANE = 8;   // C <= 512
constexpr int SM_PER_LANE = 16;   // softmax row length <= 1024

inline int row_grid(int64_t rows, int waves_per_block, int cap = 2048) {
    int64_t blocks = (rows + waves_per_block - 1) / waves_per_block;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, T *__restrict__ y,
                                                            float *__restrict__ mean, float *__restrict__ rstd,
                                                            int64_t rows, int C, int gelu, const T *__restrict__ res = nullptr) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    const int per = (C + 63) / 64;
    for (int64_t r = wave; r < rows; r += nw) {
        const T *xr = x + r * C;
        float v[MAX_PER_LANE];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            v[i] = (i < per && c < C) ? to_f32(xr[c]) : 0.f;
            s += v[i];
        }
        const float mu = wave_sum_uniform(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            const float dlt = (i < per && c < C) ? v[i] - mu : 0.f;
            q += dlt * dlt;
        }
        const float rs = rsqrtf(wave_sum_uniform(q) / (float)C + 1e-5f);
        if (lane == 0) {
            mean[r] = mu;
            rstd[r] = rs;
        }
        T *yr = y + r * C;
#pragma unroll
        for (int i = 0; i < MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (i < per && c < C) {
                float o = (v[i] - mu) * rs;
                if (gamma) o = o * gamma[c] + beta[c];
                if (gelu) o = gelu_t<T>(o);
                if (res) o += to_f32(res[r * C + c]);
                yr[c] = from_f32<T>(o);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T *__restrict__ gy, const T *__restrict__ x,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const float *__restrict__ mean, const float *__restrict__ rstd,
                                                            T *__restrict__ gx, float *__restrict__ dgamma,
                                                            float *__restrict__ dbeta, int64_t rows, int C, int gelu) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    const int per = (C + 63) / 64;
    float ag[MAX_PER_LANE], ab[MAX_PER_LANE], gm[MAX_PER_LANE], bt[MAX_PER_LANE];
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        ag[i] = ab[i] = 0.f;
        gm[i] = (gamma && i < per && c < C) ? gamma[c] : 1.f;
        bt[i] = (beta && i < per && c < C) ? beta[c] : 0.f;
    }
    for (int64_t r = wave; r < rows; r += nw) {
        const T *xr = x + r * C;
        const T *gr = gy + r * C;
        const float mu = mean[r], rs = rstd[r];
        float xh[MAX_PER_LANE], gw[MAX_PER_LANE];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            const bool ok = i < per && c < C;
            xh[i] = ok ? (to_f32(xr[c]) - mu) * rs : 0.f;
            float g = ok ? to_f32(gr[c]) : 0.f;
            if (gelu) g *= gelu_grad_t<T>(xh[i] * gm[i] + bt[i]);
            ag[i] += g * xh[i];
            ab[i] += g;
            gw[i] = g * gm[i];
            s1 += gw[i];
            s2 += gw[i] * xh[i];
        }
        s1 = wave_sum_uniform(s1) / (float)C;
        s2 = wave_sum_uniform(s2) / (float)C;
        T *gxr = gx + r * C;
#pragma unroll
        for (int i = 0; i < MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (i < per && c < C) gxr[c] = from_f32<T>(rs * (gw[i] - s1 - xh[i] * s2));
        }
    }
    if (dgamma) {
#pragma unroll
        for (int i = 0; i < MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (i < per && c < C) {
                unsafeAtomicAdd(dgamma + c, ag[i]);
                unsafeAtomicAdd(dbeta + c, ab[i]);
            }
        }
    }
}

// y = softmax(scale * x + key mask): mask (uint8, [rows / rows_per_mask][L], nonzero = excluded key, -inf) may be NULL
// NPL = elements per lane, a compile-time bound >= ceil(L / 64): with the one-size-fits-all bound of 16 an 80-wide row paid for
// 16 predicated loads and 16 expf per lane where 2 are live
template <typename T, int NPL>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const T *__restrict__ x, T *__restrict__ y, int64_t rows, int L,
                                                          float scale = 1.0f, const unsigned char *__restrict__ mask = nullptr,
                                                          int64_t rows_per_mask = 1) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    const int per = (L + 63) / 64;
    for (int64_t r = wave; r < rows; r += nw) {
        const T *xr = x + r * L;
        const unsigned char *mr = mask ? mask + (r / rows_per_mask) * L : nullptr;
        float v[NPL];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int c = lane + 64 * i;
            v[i] = (i < per && c < L) ? to_f32(xr[c]) * scale : -INFINITY;
            if (mr && i < per && c < L && mr[c]) v[i] = -INFINITY;
            mx = fmaxf(mx, v[i]);
        }
        mx = wave_max_uniform(mx);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            v[i] = (v[i] == -INFINITY) ? 0.f : expf(v[i] - mx);
            s += v[i];
        }
        const float inv = 1.0f / wave_sum_uniform(s);
        T *yr = y + r * L;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int c = lane + 64 * i;
            if (i < per && c < L) yr[c] = from_f32<T>(v[i] * inv);
        }
    }
}

template <typename T, int NPL>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T *__restrict__ gy, const T *__restrict__ y,
                                                          T *__restrict__ gx, int64_t rows, int L, float scale = 1.0f) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    const int per = (L + 63) / 64;
    for (int64_t r = wave; r < rows; r += nw) {
        float yv[NPL], gv[NPL];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int c = lane + 64 * i;
            const bool ok = i < per && c < L;
            yv[i] = ok ? to_f32(y[r * L + c]) : 0.f;
            gv[i] = ok ? to_f32(gy[r * L + c]) : 0.f;
            dot += yv[i] * gv[i];
        }
        dot = wave_sum_uniform(dot);
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int c = lane + 64 * i;
            if (i < per && c < L) gx[r * L + c] = from_f32<T>(scale * yv[i] * (gv[i] - dot));
        }
    }
}

inline bool softmax_generic() { return false; }      // (the one-size kernel, NPL = 16, for every row length: -0.56 ms per step against it)
template <typename T>
void launch_softmax_fwd(int grid, hipStream_t s, const T *x, T *y, int64_t rows, int L, float scale, const unsigned char *mask, int64_t rpm) {
    const int per = softmax_generic() ? SM_PER_LANE : (L + 63) / 64;
#define SMF(N) softmax_fwd_kernel<T, N><<<grid, 256, 0, s>>>(x, y, rows, L, scale, mask, rpm)
    if (per <= 1) SMF(1); else if (per == 2) SMF(2); else if (per == 3) SMF(3); else if (per <= 5) SMF(5); else if (per <= 8) SMF(8); else SMF(SM_PER_LANE);
#undef SMF
}
template <typename T>
void launch_softmax_bwd(int grid, hipStream_t s, const T *gy, const T *y, T *gx, int64_t rows, int L, float scale) {
    const int per = softmax_generic() ? SM_PER_LANE : (L + 63) / 64;
#define SMB(N) softmax_bwd_kernel<T, N><<<grid, 256, 0, s>>>(gy, y, gx, rows, L, scale)
    if (per <= 1) SMB(1); else if (per == 2) SMB(2); else if (per == 3) SMB(3); else if (per <= 5) SMB(5); else if (per <= 8) SMB(8); else SMB(SM_PER_LANE);
#undef SMB
}

// Rows longer than 64 * SM_PER_LANE (the DETR encoder at 960x1280 has 1200 keys): three streaming passes per row
// (max, sum, write) instead of holding the row in registers.
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_long_kernel(const T *__restrict__ x, T *__restrict__ y, int64_t rows, int L, float scale,
                                                               const unsigned char *__restrict__ mask, int64_t rows_per_mask) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < rows; r += nw) {
        const T *xr = x + r * L;
        const unsigned char *mr = mask ? mask + (r / rows_per_mask) * L : nullptr;
        float mx = -INFINITY;
        for (int c = lane; c < L; c += 64) {
            const float v = (mr && mr[c]) ? -INFINITY : to_f32(xr[c]) * scale;
            mx = fmaxf(mx, v);
        }
        mx = wave_max_uniform(mx);
        float sum = 0.f;
        for (int c = lane; c < L; c += 64) {
            const float v = (mr && mr[c]) ? -INFINITY : to_f32(xr[c]) * scale;
            sum += (v == -INFINITY) ? 0.f : expf(v - mx);
        }
        const float inv = 1.0f / wave_sum_uniform(sum);
        T *yr = y + r * L;
        for (int c = lane; c < L; c += 64) {
            const float v = (mr && mr[c]) ? -INFINITY : to_f32(xr[c]) * scale;
            yr[c] = from_f32<T>(((v == -INFINITY) ? 0.f : expf(v - mx)) * inv);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_long_kernel(const T *__restrict__ gy, const T *__restrict__ y, T *__restrict__ gx,
                                                               int64_t rows, int L, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < rows; r += nw) {
        float dot = 0.f;
        for (int c = lane; c < L; c += 64) dot += to_f32(y[r * L + c]) * to_f32(gy[r * L + c]);
        dot = wave_sum_uniform(dot);
        for (int c = lane; c < L; c += 64) gx[r * L + c] = from_f32<T>(scale * to_f32(y[r * L + c]) * (to_f32(gy[r * L + c]) - dot));
    }
}

template <typename T>
__global__ void act_bwd_kernel(const T *__restrict__ gy, const T *__restrict__ ref, T *__restrict__ gx,
                               const float *__restrict__ scale, int64_t total, int C, int act, float act_scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float g = to_f32(gy[i]);
        const float rv = ref ? to_f32(ref[i]) : 0.f;
        switch (act) {
            case GWD_ACT_RELU: g = rv > 0.f ? g : 0.f; break;
            case GWD_ACT_GELU: g *= gelu_grad_t<T>(rv); break;
            case GWD_ACT_ELU: g *= (rv > 0.f ? 1.0f : rv / act_scale + 1.0f); break;
            case GWD_ACT_SIGMOID: {
                const float sg = rv / act_scale;
                g *= sg * (1.0f - sg);
                break;
            }
            default: break;
        }
        g *= act_scale;
        if (scale) g *= scale[i % C];
        gx[i] = from_f32<T>(g);
    }
}

// out[c] += sum over rows; block = 256 threads = (256/CW rows) x CW columns, coalesced along c.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T *__restrict__ g, float *__restrict__ out, int64_t rows, int C) {
    const int cw = C < 256 ? (C <= 32 ? 32 : (C <= 64 ? 64 : (C <= 128 ? 128 : 256))) : 256;
    const int rpb = 256 / cw;
    const int tc = threadIdx.x % cw, tr = threadIdx.x / cw;
    for (int c0 = 0; c0 < C; c0 += cw) {
        const int c = c0 + tc;
        float s = 0.f;
        if (c < C)
            for (int64_t r = (int64_t)blockIdx.x * rpb + tr; r < rows; r += (int64_t)gridDim.x * rpb) s += to_f32(g[r * C + c]);
        if (c < C) unsafeAtomicAdd(out + c, s);
    }
}


// ---------------------------------------------------------------------------------------------------------
// Vectorised LayerNorm: LPR lanes share one row, each lane owns NCH chunks of VEC = 16 bytes of channels, a wave
// normalises 64/LPR rows at once (C = 64 bf16 -> 8 rows per wave, one 1 KiB coalesced load per instruction).
// ---------------------------------------------------------------------------------------------------------
template <typename T> struct VecOf { static constexpr int N = 16 / sizeof(T); };
// VB = bytes per lane access: 16, or 8 for channel counts that are only a multiple of half a vector (60, 120, 300)
template <int VB> struct Raw;
template <> struct Raw<16> { uint4 v; };
template <> struct Raw<8> { uint2 v; };

template <typename T, int LPR, int NCH, int VB, bool PAD = false, bool GELU = false>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const T *__restrict__ x, const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, T *__restrict__ y,
                                                                float *__restrict__ mean, float *__restrict__ rstd,
                                                                int64_t rows, int C, int gelu, const T *__restrict__ res, int ld) {
    // PAD: rows are ld >= C elements apart and the channels C..ld-1 are padding - their input is ignored (whatever it holds), their
    // output is written as zeros (the zero-padded activations of a layer that runs on a rounded-up channel count, see
    // gwd_unpad_add_batch); C need not be a multiple of the vector then, ld is.  !PAD: ld == C.
    constexpr int VEC = VB / (int)sizeof(T), RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane % LPR, rsel = lane / LPR;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    float g[NCH][VEC], b[NCH][VEC];
    bool okc[NCH];              // the lane's vector lies inside the row
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = (sub + c * LPR) * VEC;
        okc[c] = ch < (PAD ? ld : C);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const bool ok = okc[c] && (!PAD || ch + e < C);
            g[c][e] = (gamma && ok) ? gamma[ch + e] : 1.f;
            b[c][e] = (beta && ok) ? beta[ch + e] : 0.f;
        }
    }
    for (int64_t r0 = wave * RPW; r0 < rows; r0 += nw * RPW) {
        const int64_t r = r0 + rsel;
        const bool okr = r < rows;
        float v[NCH][VEC];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            Raw<VB> raw = {};
            if (okr && okc[c]) raw = *(const Raw<VB> *)(x + r * ld + (sub + c * LPR) * VEC);
            const T *pv = (const T *)&raw;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                v[c][e] = (!PAD || (sub + c * LPR) * VEC + e < C) ? to_f32(pv[e]) : 0.f;
                s += v[c][e];
            }
        }
        s = segment_sum<LPR>(s);
        const float mu = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float dlt = (okc[c] && (!PAD || (sub + c * LPR) * VEC + e < C)) ? v[c][e] - mu : 0.f;
                q += dlt * dlt;
            }
        q = segment_sum<LPR>(q);
        const float rs = rsqrtf(q / (float)C + 1e-5f);
        if (okr && sub == 0) {
            mean[r] = mu;
            rstd[r] = rs;
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (!(okr && okc[c])) continue;
            alignas(16) T outv[VEC];
            Raw<VB> rraw = {};
            if (res) rraw = *(const Raw<VB> *)(res + r * ld + (sub + c * LPR) * VEC);
            const T *pr = (const T *)&rraw;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float o = (v[c][e] - mu) * rs * g[c][e] + b[c][e];
                if (GELU) o = gelu_t<T>(o);         // compile-time: the erf code is not in the plain kernels
                if (res) o += to_f32(pr[e]);
                if (PAD && (sub + c * LPR) * VEC + e >= C) o = 0.f;
                outv[e] = from_f32<T>(o);
            }
            *(Raw<VB> *)(y + r * ld + (sub + c * LPR) * VEC) = *(const Raw<VB> *)outv;
        }
    }
}

// NWV waves per workgroup: 4, or 16 for narrow rows - the affine-gradient reduction ends in one atomic per channel per
// WORKGROUP, and with hundreds of small workgroups those serialise on the same few cache lines (10-18 us of a 25 us call)
template <typename T, int LPR, int NCH, int VB, int NWV, bool PAD = false, bool GELU = false>
__global__ __launch_bounds__(NWV * 64) void layernorm_bwd_vec_kernel(const T *__restrict__ gy, const T *__restrict__ x,
                                                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                T *__restrict__ gx, float *__restrict__ dgamma,
                                                                float *__restrict__ dbeta, int64_t rows, int C, int gelu, int ld,
                                                                const T *__restrict__ gskip) {
    // gskip (may be NULL): a second gradient of x (x also feeds a skip connection: ops._LayerNormFn fan-out), added to gx here
    constexpr int VEC = VB / (int)sizeof(T), RPW = 64 / LPR;
    __shared__ float red[2][NWV][LPR * NCH * VEC];    // [gamma|beta][wave][channel slot]
    const int lane = threadIdx.x & 63, sub = lane % LPR, rsel = lane / LPR, wv = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * NWV + wv, nw = (int64_t)gridDim.x * NWV;
    float g[NCH][VEC], b[NCH][VEC], ag[NCH][VEC], ab[NCH][VEC];
    bool okc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = (sub + c * LPR) * VEC;
        okc[c] = ch < (PAD ? ld : C);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const bool ok = okc[c] && (!PAD || ch + e < C);
            g[c][e] = (gamma && ok) ? gamma[ch + e] : 1.f;
            b[c][e] = (beta && ok) ? beta[ch + e] : 0.f;
            ag[c][e] = ab[c][e] = 0.f;
        }
    }
    for (int64_t r0 = wave * RPW; r0 < rows; r0 += nw * RPW) {
        const int64_t r = r0 + rsel;
        const bool okr = r < rows;
        const float mu = okr ? mean[r] : 0.f, rs = okr ? rstd[r] : 0.f;
        float xh[NCH][VEC], gw[NCH][VEC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            Raw<VB> rx = {}, rg = {};
            const bool okv = okr && okc[c];
            if (okv) {
                rx = *(const Raw<VB> *)(x + r * ld + (sub + c * LPR) * VEC);
                rg = *(const Raw<VB> *)(gy + r * ld + (sub + c * LPR) * VEC);
            }
            const T *px = (const T *)&rx;
            const T *pg = (const T *)&rg;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const bool ok = okv && (!PAD || (sub + c * LPR) * VEC + e < C);
                xh[c][e] = ok ? (to_f32(px[e]) - mu) * rs : 0.f;
                float gg = ok ? to_f32(pg[e]) : 0.f;
                if (GELU) gg *= gelu_grad_t<T>(xh[c][e] * g[c][e] + b[c][e]);     // compile-time (bit 0 of the flags): no erf / exp code in the plain kernels
                ag[c][e] += gg * xh[c][e];
                ab[c][e] += gg;
                gw[c][e] = gg * g[c][e];
                s1 += gw[c][e];
                s2 += gw[c][e] * xh[c][e];
            }
        }
        s1 = segment_sum<LPR>(s1);
        s2 = segment_sum<LPR>(s2);
        s1 /= (float)C;
        s2 /= (float)C;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (!(okr && okc[c])) continue;
            alignas(16) T outv[VEC];
            Raw<VB> rk = {};
            if (gskip) rk = *(const Raw<VB> *)(gskip + r * ld + (sub + c * LPR) * VEC);
            const T *pk = (const T *)&rk;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float o = rs * (gw[c][e] - s1 - xh[c][e] * s2);
                if (gskip) o += to_f32(pk[e]);
                if (gelu & 2) {                          // x is the output of an ELU whose backward runs here (GWD_LN_ELU_INPUT): elu'
                    const float xo = xh[c][e] / rs + mu; // is continuous, so the rounding of the reconstructed x does not matter
                    o *= xo > 0.f ? 1.0f : xo + 1.0f;
                }
                if (PAD && (sub + c * LPR) * VEC + e >= C) o = 0.f;
                outv[e] = from_f32<T>(o);
            }
            *(Raw<VB> *)(gx + r * ld + (sub + c * LPR) * VEC) = *(const Raw<VB> *)outv;
        }
    }
    if (dgamma) {
        // rows of the wave (lanes with equal `sub`) -> one value per channel, then across the 4 waves through LDS
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
#pragma unroll
                for (int o = LPR; o < 64; o <<= 1) {
                    ag[c][e] += __shfl_xor(ag[c][e], o, 64);
                    ab[c][e] += __shfl_xor(ab[c][e], o, 64);
                }
                if (rsel == 0) {
                    red[0][wv][(c * LPR + sub) * VEC + e] = ag[c][e];
                    red[1][wv][(c * LPR + sub) * VEC + e] = ab[c][e];
                }
            }
        __syncthreads();
        for (int t = threadIdx.x; t < LPR * NCH * VEC; t += NWV * 64) {
            const int slot = t / VEC, e = t % VEC;
            const int c = slot / LPR, sb = slot % LPR;
            const int ch = (sb + c * LPR) * VEC + e;
            if (ch < C) {
                float sg = 0.f, sb = 0.f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) {
                    sg += red[0][w][t];
                    sb += red[1][w][t];
                }
                unsafeAtomicAdd(dgamma + ch, sg);
                unsafeAtomicAdd(dbeta + ch, sb);
            }
        }
    }
}

// out[c] += column sums, 16-byte vector loads: thread = (row slot, vector column); rows stride over the grid.
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T *__restrict__ gmat, float *__restrict__ out, int64_t rows, int C) {
    constexpr int VEC = VecOf<T>::N;
    __shared__ float red[256 * VEC];
    const int vpr = C / VEC;                 // vectors per row (<= 256)
    const int rpb = 256 / vpr;               // row slots per block
    const int slot = threadIdx.x / vpr, v = threadIdx.x % vpr;
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    if (slot < rpb) {
        for (int64_t r = (int64_t)blockIdx.x * rpb + slot; r < rows; r += (int64_t)gridDim.x * rpb) {
            const uint4 raw = *(const uint4 *)(gmat + r * C + v * VEC);
            const T *p = (const T *)&raw;
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[e] += to_f32(p[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[threadIdx.x * VEC + e] = s[e];
    __syncthreads();
    for (int t = threadIdx.x; t < vpr * VEC; t += 256) {
        const int vv = t / VEC, e = t % VEC;
        float a = 0.f;
        for (int sl = 0; sl < rpb; ++sl) a += red[(sl * vpr + vv) * VEC + e];
        unsafeAtomicAdd(out + t, a);
    }
}

// Up to GWD_COLSUM_BATCH column sums in ONE launch: the job records travel by value in the kernel arguments (no table
// upload, so the launch is capturable as it is); workgroup -> job through the block0 prefix.  The step has ~180 bias
// gradients of 6-25 us each; most of that is per-launch latency.
struct ColsumBatch {
    gwd_colsum_job j[GWD_COLSUM_BATCH];
    int n;
};
template <typename T>
__global__ __launch_bounds__(256) void colsum_batch_kernel(const ColsumBatch b) {
    constexpr int VEC = VecOf<T>::N;
    __shared__ float red[256 * VEC];
    int ji = 0;
#pragma unroll 1
    for (int k = 1; k < b.n; ++k)
        if ((int)blockIdx.x >= b.j[k].block0) ji = k;
    const gwd_colsum_job job = b.j[ji];
    const int blk = blockIdx.x - job.block0, nblk = job.blocks;
    const T *gmat = (const T *)job.g;
    const int C = job.C;
    const int64_t rows = job.rows;
    const int vpr = C / VEC, rpb = 256 / vpr;
    const int slot = threadIdx.x / vpr, v = threadIdx.x % vpr;
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    if (slot < rpb) {
        for (int64_t r = (int64_t)blk * rpb + slot; r < rows; r += (int64_t)nblk * rpb) {
            const uint4 raw = *(const uint4 *)(gmat + r * C + v * VEC);
            const T *p = (const T *)&raw;
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[e] += to_f32(p[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[threadIdx.x * VEC + e] = s[e];
    __syncthreads();
    for (int t = threadIdx.x; t < vpr * VEC; t += 256) {
        const int vv = t / VEC, e = t % VEC;
        float a = 0.f;
        for (int sl = 0; sl < rpb; ++sl) a += red[(sl * vpr + vv) * VEC + e];
        unsafeAtomicAdd(job.out + t, a);
    }
}

// 16-byte vector form of act_bwd_kernel (C a multiple of the vector width, so a vector never straddles rows)
template <typename T>
__global__ void act_bwd_vec_kernel(const T *__restrict__ gy, const T *__restrict__ ref, T *__restrict__ gx,
                                   const float *__restrict__ scale, int64_t nvec, int C, int act, float act_scale) {
    constexpr int VEC = VecOf<T>::N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 rg = *(const uint4 *)(gy + i * VEC);
        uint4 rr = make_uint4(0u, 0u, 0u, 0u);
        if (ref) rr = *(const uint4 *)(ref + i * VEC);
        const T *pg = (const T *)&rg;
        const T *pr = (const T *)&rr;
        const int c0 = (int)((i * VEC) % C);
        alignas(16) T outv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float g = to_f32(pg[e]);
            const float rv = to_f32(pr[e]);
            switch (act) {
                case GWD_ACT_RELU: g = rv > 0.f ? g : 0.f; break;
                case GWD_ACT_GELU: g *= gelu_grad_t<T>(rv); break;
                case GWD_ACT_ELU: g *= (rv > 0.f ? 1.0f : rv / act_scale + 1.0f); break;
                case GWD_ACT_SIGMOID: {
                    const float sg = rv / act_scale;
                    g *= sg * (1.0f - sg);
                    break;
                }
                default: break;
            }
            g *= act_scale;
            if (scale) g *= scale[c0 + e];
            outv[e] = from_f32<T>(g);
        }
        *(uint4 *)(gx + i * VEC) = *(const uint4 *)outv;
    }
}

// Activation backward and the bias gradient in one pass: gx = act'(ref) * gy (16-byte vectors), and the column sums
// of gx (= dBias of the layer) accumulated in registers, reduced through LDS, one atomic per channel per workgroup.
// ACTK: the activation as a compile-time constant (none / ReLU / GELU), -1 = the run-time switch
template <typename T, int ACTK = -1>
__global__ __launch_bounds__(256) void act_bwd_colsum_kernel(const T *__restrict__ gy, const T *__restrict__ ref, T *__restrict__ gx,
                                                             float *__restrict__ dbias, int64_t rows, int C, int act_rt, float act_scale,
                                                             const T *__restrict__ mult) {
    const int act = ACTK < 0 ? act_rt : ACTK;
    constexpr int VEC = VecOf<T>::N;
    __shared__ float red[256 * VEC];
    const int vpr = C / VEC, rpb = 256 / vpr;
    const int slot = threadIdx.x / vpr, v = threadIdx.x % vpr;
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    if (slot < rpb) {
        for (int64_t r = (int64_t)blockIdx.x * rpb + slot; r < rows; r += (int64_t)gridDim.x * rpb) {
            const int64_t o = r * C + v * VEC;
            const uint4 rg = *(const uint4 *)(gy + o);
            uint4 rr = make_uint4(0u, 0u, 0u, 0u);
            if (ref) rr = *(const uint4 *)(ref + o);
            uint4 rm = make_uint4(0u, 0u, 0u, 0u);
            if (mult) rm = *(const uint4 *)(mult + o);
            const T *pg = (const T *)&rg;
            const T *pr = (const T *)&rr;
            const T *pm = (const T *)&rm;
            alignas(16) T outv[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float g = to_f32(pg[e]);
                if (mult) g = to_f32(from_f32<T>(g * to_f32(pm[e])));      // rounded as the separate multiply it replaces
                const float rv = to_f32(pr[e]);
                switch (act) {
                    case GWD_ACT_RELU: g = rv > 0.f ? g : 0.f; break;
                    case GWD_ACT_GELU: g *= gelu_grad_t<T>(rv); break;
                    case GWD_ACT_ELU: g *= (rv > 0.f ? 1.0f : rv / act_scale + 1.0f); break;
                    case GWD_ACT_SIGMOID: {
                        const float sg = rv / act_scale;
                        g *= sg * (1.0f - sg);
                        break;
                    }
                    default: break;
                }
                g *= act_scale;
                outv[e] = from_f32<T>(g);
                s[e] += to_f32(outv[e]);           // the sum of what the consumers of gx will read
            }
            *(uint4 *)(gx + o) = *(const uint4 *)outv;
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[threadIdx.x * VEC + e] = s[e];
    __syncthreads();
    for (int t = threadIdx.x; t < vpr * VEC; t += 256) {
        const int vv = t / VEC, e = t % VEC;
        float a = 0.f;
        for (int sl = 0; sl < rpb; ++sl) a += red[(sl * vpr + vv) * VEC + e];
        unsafeAtomicAdd(dbias + t, a);
    }
}

template <typename T, int VB, bool PAD = false>
int launch_ln_fwd_vec(const T *x, const float *ga, const float *be, T *y, float *mean, float *rstd, int64_t rows, int C, int gelu,
                      const T *res, int ld, hipStream_t s) {
    constexpr int VEC = VB / (int)sizeof(T);
    if (ld % VEC) return -5;
    const int need = ld / VEC;
#define LN_FWD(LPR, NCH)                                                                                            \
    {                                                                                                               \
        const int64_t wv = (rows + (64 / LPR) - 1) / (64 / LPR);                                                    \
        /* ~20 KiB of rows per workgroup: measured optimum between 10 MB (512 workgroups) and 49 MB (2048) tensors */  \
        int64_t cap = rows * C * (int64_t)sizeof(T) / 20480;                                                        \
        cap = cap < 256 ? 256 : (cap > 2048 ? 2048 : cap);                                                          \
        if (gelu) layernorm_fwd_vec_kernel<T, LPR, NCH, VB, PAD, true><<<row_grid(wv, 4, (int)cap), 256, 0, s>>>(x, ga, be, y, mean, rstd, rows, C, gelu, res, ld); \
        else layernorm_fwd_vec_kernel<T, LPR, NCH, VB, PAD, false><<<row_grid(wv, 4, (int)cap), 256, 0, s>>>(x, ga, be, y, mean, rstd, rows, C, gelu, res, ld); \
        return 0;                                                                                                   \
    }
    if (need <= 8) LN_FWD(8, 1)
    if (need <= 16) LN_FWD(16, 1)
    if (need <= 32) LN_FWD(32, 1)
    if (need <= 64) LN_FWD(64, 1)
    if (need <= 128) LN_FWD(64, 2)
#undef LN_FWD
    return -5;
}

template <typename T, int VB, bool PAD = false>
int launch_ln_bwd_vec(const T *gy, const T *x, const float *ga, const float *be, const float *mean, const float *rstd, T *gx,
                      float *dg, float *db, int64_t rows, int C, int gelu, int ld, const T *gskip, hipStream_t s) {
    constexpr int VEC = VB / (int)sizeof(T);
    if (ld % VEC) return -5;
    const int need = ld / VEC;
#define LN_BWD(LPR, NCH)                                                                                            \
    {                                                                                                               \
        const int64_t wv = (rows + (64 / LPR) - 1) / (64 / LPR);                                                    \
        if (LPR * NCH * VEC <= 512 && dg) {                     /* narrow rows: 16-wave workgroups, one per CU */   \
            int grid = row_grid(wv, 16);                                                                            \
            if (grid > 256) grid = 256;                                                                             \
            if (gelu & 1) layernorm_bwd_vec_kernel<T, LPR, NCH, VB, 16, PAD, true><<<grid, 1024, 0, s>>>(gy, x, ga, be, mean, rstd, gx, dg, db, rows, C, gelu, ld, gskip); \
            else layernorm_bwd_vec_kernel<T, LPR, NCH, VB, 16, PAD, false><<<grid, 1024, 0, s>>>(gy, x, ga, be, mean, rstd, gx, dg, db, rows, C, gelu, ld, gskip); \
            return 0;                                                                                               \
        }                                                                                                           \
        int grid = row_grid(wv, 4);                                                                                 \
        if (grid > 768) grid = 768;                                                                                 \
        if (gelu & 1) layernorm_bwd_vec_kernel<T, LPR, NCH, VB, 4, PAD, true><<<grid, 256, 0, s>>>(gy, x, ga, be, mean, rstd, gx, dg, db, rows, C, gelu, ld, gskip); \
        else layernorm_bwd_vec_kernel<T, LPR, NCH, VB, 4, PAD, false><<<grid, 256, 0, s>>>(gy, x, ga, be, mean, rstd, gx, dg, db, rows, C, gelu, ld, gskip); \
        return 0;                                                                                                   \
    }
    if (need <= 8) LN_BWD(8, 1)
    if (need <= 16) LN_BWD(16, 1)
    if (need <= 32) LN_BWD(32, 1)
    if (need <= 64) LN_BWD(64, 1)
    if (need <= 128) LN_BWD(64, 2)
#undef LN_BWD
    return -5;
}

}  // namespace

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32) \
    if ((dtype) == GWD_BF16) { CALL_BF16; }    \
    else if ((dtype) == GWD_F32) { CALL_F32; } \
    else return -2;

extern "C" int gwd_layernorm_forward(const void *x, const float *gamma, const float *beta, const void *residual, void *y, float *mean,
                                     float *rstd, int64_t rows, int32_t C, int32_t ld, int32_t gelu, int32_t dtype, void *stream) {
    gelu &= 1;
    if (!x || !y || !mean || !rstd || rows < 0 || C <= 0 || C > 64 * MAX_PER_LANE) return -1;
    if ((gamma == nullptr) != (beta == nullptr)) return -1;
    if (ld == 0) ld = C;
    if (ld < C || ld > 64 * MAX_PER_LANE) return -1;
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (ld != C) {                           // zero-padded rows: 16-byte vectors over the pitch, per-element channel masks
        int rc = -5;
        if (dtype == GWD_BF16 && ld % 8 == 0) rc = launch_ln_fwd_vec<__bf16, 16, true>((const __bf16 *)x, gamma, beta, (__bf16 *)y, mean, rstd, rows, C, gelu, (const __bf16 *)residual, ld, s);
        else if (dtype == GWD_F32 && ld % 4 == 0) rc = launch_ln_fwd_vec<float, 16, true>((const float *)x, gamma, beta, (float *)y, mean, rstd, rows, C, gelu, (const float *)residual, ld, s);
        if (rc == 0) { GWD_CHECK_LAUNCH(); return 0; }
        return -4;
    }
    {
        int rc = -5;
        if (dtype == GWD_BF16 && C % 8 == 0) rc = launch_ln_fwd_vec<__bf16, 16>((const __bf16 *)x, gamma, beta, (__bf16 *)y, mean, rstd, rows, C, gelu, (const __bf16 *)residual, ld, s);
        else if (dtype == GWD_BF16 && C % 4 == 0) rc = launch_ln_fwd_vec<__bf16, 8>((const __bf16 *)x, gamma, beta, (__bf16 *)y, mean, rstd, rows, C, gelu, (const __bf16 *)residual, ld, s);
        else if (dtype == GWD_F32 && C % 4 == 0) rc = launch_ln_fwd_vec<float, 16>((const float *)x, gamma, beta, (float *)y, mean, rstd, rows, C, gelu, (const float *)residual, ld, s);
        else if (dtype == GWD_F32 && C % 2 == 0) rc = launch_ln_fwd_vec<float, 8>((const float *)x, gamma, beta, (float *)y, mean, rstd, rows, C, gelu, (const float *)residual, ld, s);
        if (rc == 0) { GWD_CHECK_LAUNCH(); return 0; }
    }
    const int grid = row_grid(rows, 4);
    DISPATCH_T(dtype,
               (layernorm_fwd_kernel<__bf16><<<grid, 256, 0, s>>>((const __bf16 *)x, gamma, beta, (__bf16 *)y, mean, rstd, rows, C, gelu, (const __bf16 *)residual)),
               (layernorm_fwd_kernel<float><<<grid, 256, 0, s>>>((const float *)x, gamma, beta, (float *)y, mean, rstd, rows, C, gelu, (const float *)residual)));
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_layernorm_backward(const void *gy, const void *x, const float *gamma, const float *beta,
                                      const float *mean, const float *rstd, void *gx, float *dgamma, float *dbeta,
                                      int64_t rows, int32_t C, int32_t ld, int32_t gelu, const void *gskip, int32_t dtype, void *stream) {
    if (!gy || !x || !gx || !mean || !rstd || rows < 0 || C <= 0 || C > 64 * MAX_PER_LANE) return -1;
    if ((dgamma == nullptr) != (dbeta == nullptr)) return -1;
    if (ld == 0) ld = C;
    if (ld < C || ld > 64 * MAX_PER_LANE) return -1;
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (ld != C) {
        int rc = -5;
        if (dtype == GWD_BF16 && ld % 8 == 0) rc = launch_ln_bwd_vec<__bf16, 16, true>((const __bf16 *)gy, (const __bf16 *)x, gamma, beta, mean, rstd, (__bf16 *)gx, dgamma, dbeta, rows, C, gelu, ld, (const __bf16 *)gskip, s);
        else if (dtype == GWD_F32 && ld % 4 == 0) rc = launch_ln_bwd_vec<float, 16, true>((const float *)gy, (const float *)x, gamma, beta, mean, rstd, (float *)gx, dgamma, dbeta, rows, C, gelu, ld, (const float *)gskip, s);
        if (rc == 0) { GWD_CHECK_LAUNCH(); return 0; }
        return -4;
    }
    {
        int rc = -5;
        if (dtype == GWD_BF16 && C % 8 == 0) rc = launch_ln_bwd_vec<__bf16, 16>((const __bf16 *)gy, (const __bf16 *)x, gamma, beta, mean, rstd, (__bf16 *)gx, dgamma, dbeta, rows, C, gelu, ld, (const __bf16 *)gskip, s);
        else if (dtype == GWD_BF16 && C % 4 == 0) rc = launch_ln_bwd_vec<__bf16, 8>((const __bf16 *)gy, (const __bf16 *)x, gamma, beta, mean, rstd, (__bf16 *)gx, dgamma, dbeta, rows, C, gelu, ld, (const __bf16 *)gskip, s);
        else if (dtype == GWD_F32 && C % 4 == 0) rc = launch_ln_bwd_vec<float, 16>((const float *)gy, (const float *)x, gamma, beta, mean, rstd, (float *)gx, dgamma, dbeta, rows, C, gelu, ld, (const float *)gskip, s);
        else if (dtype == GWD_F32 && C % 2 == 0) rc = launch_ln_bwd_vec<float, 8>((const float *)gy, (const float *)x, gamma, beta, mean, rstd, (float *)gx, dgamma, dbeta, rows, C, gelu, ld, (const float *)gskip, s);
        if (rc == 0) { GWD_CHECK_LAUNCH(); return 0; }
    }
    if (gskip || (gelu & 2)) return -4;      // the generic kernel has neither a skip input nor the ELU gate: the caller does both
    int grid = row_grid(rows, 4);
    if (grid > 1024) grid = 1024;
    DISPATCH_T(dtype,
               (layernorm_bwd_kernel<__bf16><<<grid, 256, 0, s>>>((const __bf16 *)gy, (const __bf16 *)x, gamma, beta, mean, rstd, (__bf16 *)gx, dgamma, dbeta, rows, C, gelu)),
               (layernorm_bwd_kernel<float><<<grid, 256, 0, s>>>((const float *)gy, (const float *)x, gamma, beta, mean, rstd, (float *)gx, dgamma, dbeta, rows, C, gelu)));
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_softmax_forward(const void *x, void *y, int64_t rows, int32_t L, int32_t dtype, void *stream) {
    if (!x || !y || rows < 0 || L <= 0) return -1;
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int grid = row_grid(rows, 4);
    if (L > 64 * SM_PER_LANE) {
        DISPATCH_T(dtype, (softmax_fwd_long_kernel<__bf16><<<grid, 256, 0, s>>>((const __bf16 *)x, (__bf16 *)y, rows, L, 1.0f, nullptr, 1)),
                   (softmax_fwd_long_kernel<float><<<grid, 256, 0, s>>>((const float *)x, (float *)y, rows, L, 1.0f, nullptr, 1)));
        GWD_CHECK_LAUNCH();
        return 0;
    }
    DISPATCH_T(dtype, (launch_softmax_fwd<__bf16>(grid, s, (const __bf16 *)x, (__bf16 *)y, rows, L, 1.0f, nullptr, 1)),
               (launch_softmax_fwd<float>(grid, s, (const float *)x, (float *)y, rows, L, 1.0f, nullptr, 1)));
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_softmax_backward(const void *gy, const void *y, void *gx, int64_t rows, int32_t L, int32_t dtype,
                                    void *stream) {
    if (!gy || !y || !gx || rows < 0 || L <= 0) return -1;
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int grid = row_grid(rows, 4);
    if (L > 64 * SM_PER_LANE) {
        DISPATCH_T(dtype, (softmax_bwd_long_kernel<__bf16><<<grid, 256, 0, s>>>((const __bf16 *)gy, (const __bf16 *)y, (__bf16 *)gx, rows, L, 1.0f)),
                   (softmax_bwd_long_kernel<float><<<grid, 256, 0, s>>>((const float *)gy, (const float *)y, (float *)gx, rows, L, 1.0f)));
        GWD_CHECK_LAUNCH();
        return 0;
    }
    DISPATCH_T(dtype,
               (launch_softmax_bwd<__bf16>(grid, s, (const __bf16 *)gy, (const __bf16 *)y, (__bf16 *)gx, rows, L, 1.0f)),
               (launch_softmax_bwd<float>(grid, s, (const float *)gy, (const float *)y, (float *)gx, rows, L, 1.0f)));
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_act_backward(const void *gy, const void *ref, void *gx, const float *scale, int64_t rows, int32_t C,
                                int32_t act, float act_scale, int32_t dtype, void *stream) {
    if (!gy || !gx || rows < 0 || C <= 0) return -1;
    if (act != GWD_ACT_NONE && !ref) return -1;
    const int64_t total = rows * C;
    if (total == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    {
        const int vec = dtype == GWD_BF16 ? 8 : 4;
        if ((dtype == GWD_BF16 || dtype == GWD_F32) && C % vec == 0) {
            const int64_t nvec = total / vec;
            int64_t nb = (nvec + 255) / 256;
            const int vgrid = (int)(nb > 8192 ? 8192 : nb);
            if (dtype == GWD_BF16) act_bwd_vec_kernel<__bf16><<<vgrid, 256, 0, s>>>((const __bf16 *)gy, (const __bf16 *)ref, (__bf16 *)gx, scale, nvec, C, act, act_scale);
            else act_bwd_vec_kernel<float><<<vgrid, 256, 0, s>>>((const float *)gy, (const float *)ref, (float *)gx, scale, nvec, C, act, act_scale);
            GWD_CHECK_LAUNCH();
            return 0;
        }
    }
    int64_t b = (total + 255) / 256;
    const int grid = (int)(b > 4096 ? 4096 : b);
    DISPATCH_T(dtype,
               (act_bwd_kernel<__bf16><<<grid, 256, 0, s>>>((const __bf16 *)gy, (const __bf16 *)ref, (__bf16 *)gx, scale, total, C, act, act_scale)),
               (act_bwd_kernel<float><<<grid, 256, 0, s>>>((const float *)gy, (const float *)ref, (float *)gx, scale, total, C, act, act_scale)));
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_colsum(const void *g, float *out, int64_t rows, int32_t C, int32_t dtype, void *stream) {
    if (!g || !out || rows < 0 || C <= 0) return -1;
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    {
        const int vec = dtype == GWD_BF16 ? 8 : 4;
        if (C % vec == 0 && C / vec <= 256 && (dtype == GWD_BF16 || dtype == GWD_F32)) {
            const int rpb = 256 / (C / vec);
            int64_t nb = (rows + (int64_t)rpb * 16 - 1) / ((int64_t)rpb * 16);     // >= 16 rows per row slot
            const int grid = (int)(nb > 512 ? 512 : (nb < 1 ? 1 : nb));
            if (dtype == GWD_BF16) colsum_vec_kernel<__bf16><<<grid, 256, 0, s>>>((const __bf16 *)g, out, rows, C);
            else colsum_vec_kernel<float><<<grid, 256, 0, s>>>((const float *)g, out, rows, C);
            GWD_CHECK_LAUNCH();
            return 0;
        }
    }
    int64_t b = (rows + 63) / 64;
    const int grid = (int)(b > 512 ? 512 : (b < 1 ? 1 : b));
    DISPATCH_T(dtype, (colsum_kernel<__bf16><<<grid, 256, 0, s>>>((const __bf16 *)g, out, rows, C)),
               (colsum_kernel<float><<<grid, 256, 0, s>>>((const float *)g, out, rows, C)));
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_colsum_batch(const gwd_colsum_job *jobs, int32_t n_jobs, int32_t dtype, void *stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > GWD_COLSUM_BATCH) return -1;
    if (dtype != GWD_BF16 && dtype != GWD_F32) return -2;
    const int vec = dtype == GWD_BF16 ? 8 : 4;
    ColsumBatch b;
    int total = 0;
    for (int i = 0; i < n_jobs; ++i) {
        gwd_colsum_job j = jobs[i];
        if (!j.g || !j.out || j.rows <= 0 || j.C <= 0) return -1;
        if (j.C % vec != 0 || j.C / vec > 256) return -4;          // not a vector shape: use gwd_colsum for this one
        const int rpb = 256 / (j.C / vec);
        int64_t nb = (j.rows + (int64_t)rpb * 16 - 1) / ((int64_t)rpb * 16);
        j.blocks = (int)(nb > 512 ? 512 : (nb < 1 ? 1 : nb));    // the grid gwd_colsum gives the same shape
        j.block0 = total;
        total += j.blocks;
        b.j[i] = j;
    }
    b.n = n_jobs;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GWD_BF16) colsum_batch_kernel<__bf16><<<total, 256, 0, s>>>(b);
    else colsum_batch_kernel<float><<<total, 256, 0, s>>>(b);
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_act_backward_colsum(const void *gy, const void *ref, void *gx, float *dbias, int64_t rows, int32_t C,
                                       int32_t act, float act_scale, const void *mult, int32_t dtype, void *stream) {
    if (!gy || !gx || !dbias || rows < 0 || C <= 0) return -1;
    if (act != GWD_ACT_NONE && !ref) return -1;
    if (rows == 0) return 0;
    const int vec = dtype == GWD_BF16 ? 8 : (dtype == GWD_F32 ? 4 : 0);
    if (!vec) return -2;
    if (C % vec != 0 || C / vec > 256) return -4;              // caller falls back to gwd_act_backward + gwd_colsum
    hipStream_t s = (hipStream_t)stream;
    const int rpb = 256 / (C / vec);
    int64_t nb = (rows + (int64_t)rpb * 8 - 1) / ((int64_t)rpb * 8);
    const int grid = (int)(nb > 1024 ? 1024 : (nb < 1 ? 1 : nb));
    if (dtype == GWD_BF16) {
#define ABC(K_) act_bwd_colsum_kernel<__bf16, K_><<<grid, 256, 0, s>>>((const __bf16 *)gy, (const __bf16 *)ref, (__bf16 *)gx, dbias, rows, C, act, act_scale, (const __bf16 *)mult)
        if (act == GWD_ACT_NONE) ABC(GWD_ACT_NONE);
        else if (act == GWD_ACT_RELU) ABC(GWD_ACT_RELU);
        else if (act == GWD_ACT_GELU) ABC(GWD_ACT_GELU);
        else ABC(-1);
#undef ABC
    } else
        act_bwd_colsum_kernel<float><<<grid, 256, 0, s>>>((const float *)gy, (const float *)ref, (float *)gx, dbias, rows, C, act, act_scale, (const float *)mult);
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_softmax_masked_forward(const void *x, const uint8_t *key_mask, void *y, int64_t rows, int32_t L,
                                          int64_t rows_per_mask, float scale, int32_t dtype, void *stream) {
    if (!x || !y || rows < 0 || L <= 0 || rows_per_mask <= 0) return -1;
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int grid = row_grid(rows, 4);
    if (L > 64 * SM_PER_LANE) {
        DISPATCH_T(dtype, (softmax_fwd_long_kernel<__bf16><<<grid, 256, 0, s>>>((const __bf16 *)x, (__bf16 *)y, rows, L, scale, key_mask, rows_per_mask)),
                   (softmax_fwd_long_kernel<float><<<grid, 256, 0, s>>>((const float *)x, (float *)y, rows, L, scale, key_mask, rows_per_mask)));
        GWD_CHECK_LAUNCH();
        return 0;
    }
    DISPATCH_T(dtype, (launch_softmax_fwd<__bf16>(grid, s, (const __bf16 *)x, (__bf16 *)y, rows, L, scale, key_mask, rows_per_mask)),
               (launch_softmax_fwd<float>(grid, s, (const float *)x, (float *)y, rows, L, scale, key_mask, rows_per_mask)));
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_softmax_scaled_backward(const void *gy, const void *y, void *gx, int64_t rows, int32_t L, float scale,
                                           int32_t dtype, void *stream) {
    if (!gy || !y || !gx || rows < 0 || L <= 0) return -1;
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int grid = row_grid(rows, 4);
    if (L > 64 * SM_PER_LANE) {
        DISPATCH_T(dtype, (softmax_bwd_long_kernel<__bf16><<<grid, 256, 0, s>>>((const __bf16 *)gy, (const __bf16 *)y, (__bf16 *)gx, rows, L, scale)),
                   (softmax_bwd_long_kernel<float><<<grid, 256, 0, s>>>((const float *)gy, (const float *)y, (float *)gx, rows, L, scale)));
        GWD_CHECK_LAUNCH();
        return 0;
    }
    DISPATCH_T(dtype,
               (launch_softmax_bwd<__bf16>(grid, s, (const __bf16 *)gy, (const __bf16 *)y, (__bf16 *)gx, rows, L, scale)),
               (launch_softmax_bwd<float>(grid, s, (const float *)gy, (const float *)y, (float *)gx, rows, L, scale)));
    GWD_CHECK_LAUNCH();
    return 0;
}
