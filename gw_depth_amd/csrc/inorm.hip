// Instance-normalised GELU residual update of the reference-point attention logits (the "diffusion" loop of
// multiscale_transformerr.py:299-302):   y = a + gelu((u - mean_bc(u)) * rsqrt(var_bc(u) + eps))
// with statistics over all L positions of one image for each of the C (= heads) channels; tensors are (B, L, C),
// channel innermost.  Two launches each way, no atomics and no memset (HIP-graph safe):
//   stats : grid (S, B) - slice s of image b -> per-channel (mean, M2) of its positions   (two passes over the slice)
//   apply : every workgroup merges the S slice statistics (Chan et al.) and applies the update
//   backward: slice sums of g_n and g_n * n, then du = rstd * (g_n - mean(g_n) - n * mean(g_n * n)).
#include "common.h"

namespace {

constexpr int NT = 256;

template <typename T> struct Vec16;
template <> struct Vec16<float> { static constexpr int N = 4; };
template <> struct Vec16<__bf16> { static constexpr int N = 8; };

template <typename T> __device__ __forceinline__ void load_vec(const T *p, float (&v)[Vec16<T>::N]) {
    uint4 raw = *reinterpret_cast<const uint4 *>(p);
    const T *e = reinterpret_cast<const T *>(&raw);
#pragma unroll
    for (int i = 0; i < Vec16<T>::N; ++i) v[i] = to_f32(e[i]);
}

template <typename T> __device__ __forceinline__ void store_vec(T *p, const float (&v)[Vec16<T>::N]) {
    uint4 raw;
    T *e = reinterpret_cast<T *>(&raw);
#pragma unroll
    for (int i = 0; i < Vec16<T>::N; ++i) e[i] = from_f32<T>(v[i]);
    *reinterpret_cast<uint4 *>(p) = raw;
}

// sum over the threads that own the same channel group (tid % CG); result valid in threads tid < CG
template <int VN> __device__ __forceinline__ void group_reduce(float (&v)[VN], float *lds, int CG) {
    const int tid = threadIdx.x;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VN; ++i) lds[tid * VN + i] = v[i];
    __syncthreads();
    for (int stride = NT / 2; stride >= CG; stride >>= 1) {
        if (tid < stride) {
#pragma unroll
            for (int i = 0; i < VN; ++i) lds[tid * VN + i] += lds[(tid + stride) * VN + i];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < VN; ++i) v[i] = lds[(tid % CG) * VN + i];      // broadcast to every thread of the group
}

__device__ __forceinline__ void slice_bounds(int64_t L, int S, int s, int64_t &lo, int64_t &hi) {
    const int64_t per = (L + S - 1) / S;
    lo = (int64_t)s * per;
    hi = lo + per < L ? lo + per : L;
    if (lo > L) lo = L;
}

template <typename T>
__global__ __launch_bounds__(NT) void inorm_stats_kernel(const T *__restrict__ u, float *__restrict__ part, int64_t L, int C, int S) {
    constexpr int VN = Vec16<T>::N;
    __shared__ float lds[NT * VN];
    const int CG = C / VN, g = threadIdx.x % CG, b = blockIdx.y, s = blockIdx.x;
    int64_t lo, hi;
    slice_bounds(L, S, s, lo, hi);
    const T *base = u + (int64_t)b * L * C + g * VN;
    float sum[VN], v[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) sum[i] = 0.f;
    for (int64_t l = lo + threadIdx.x / CG; l < hi; l += NT / CG) {
        load_vec(base + l * C, v);
#pragma unroll
        for (int i = 0; i < VN; ++i) sum[i] += v[i];
    }
    group_reduce<VN>(sum, lds, CG);
    const float n = (float)(hi - lo);
    float mean[VN], m2[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) { mean[i] = n > 0.f ? sum[i] / n : 0.f; m2[i] = 0.f; }
    for (int64_t l = lo + threadIdx.x / CG; l < hi; l += NT / CG) {
        load_vec(base + l * C, v);
#pragma unroll
        for (int i = 0; i < VN; ++i) { const float d = v[i] - mean[i]; m2[i] += d * d; }
    }
    group_reduce<VN>(m2, lds, CG);
    if (threadIdx.x < CG) {
        float *o = part + (((int64_t)b * S + s) * C + g * VN) * 2;
#pragma unroll
        for (int i = 0; i < VN; ++i) { o[2 * i] = mean[i]; o[2 * i + 1] = m2[i]; }
    }
}

// merge the S slice statistics of image b for the VN channels starting at c0 -> mean, rstd (in every thread of the channel group).
// The threads of a group share the slices (one global load each, then an LDS tree) - a per-thread walk over the S = 32 slices was 32
// DEPENDENT L2 round trips in front of one iteration of real work: 29 us per launch for a 4.5 MB tensor.
//   mean = sum_s n_s mean_s / L,   M2 = sum_s (M2_s + n_s (mean_s - mean)^2)
template <int VN>
__device__ __forceinline__ void merge_stats(const float *__restrict__ part, float *lds, int b, int c0, int64_t L, int C, int S, float eps, int CG,
                                            float (&mean)[VN], float (&rstd)[VN]) {
    const int sl0 = threadIdx.x / CG, nsl = NT / CG;
    float acc[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) acc[i] = 0.f;
    for (int s = sl0; s < S; s += nsl) {
        int64_t lo, hi;
        slice_bounds(L, S, s, lo, hi);
        const float nb = (float)(hi - lo);
        const float *p = part + (((int64_t)b * S + s) * C + c0) * 2;
#pragma unroll
        for (int i = 0; i < VN; ++i) acc[i] += nb * p[2 * i];
    }
    group_reduce<VN>(acc, lds, CG);
#pragma unroll
    for (int i = 0; i < VN; ++i) { mean[i] = acc[i] / (float)L; acc[i] = 0.f; }
    for (int s = sl0; s < S; s += nsl) {
        int64_t lo, hi;
        slice_bounds(L, S, s, lo, hi);
        const float nb = (float)(hi - lo);
        const float *p = part + (((int64_t)b * S + s) * C + c0) * 2;
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            const float d = p[2 * i] - mean[i];
            acc[i] += p[2 * i + 1] + nb * d * d;
        }
    }
    group_reduce<VN>(acc, lds, CG);
#pragma unroll
    for (int i = 0; i < VN; ++i) rstd[i] = rsqrtf(acc[i] / (float)L + eps);
}

template <typename T>
__global__ __launch_bounds__(NT) void inorm_gelu_fwd_kernel(const T *__restrict__ a, const T *__restrict__ u, const float *__restrict__ part,
                                                             T *__restrict__ y, float *__restrict__ stat, int64_t L, int C, int S, float eps) {
    constexpr int VN = Vec16<T>::N;
    __shared__ float lds[NT * VN];
    const int CG = C / VN, g = threadIdx.x % CG, b = blockIdx.y;
    float mean[VN], rstd[VN];
    merge_stats<VN>(part, lds, b, g * VN, L, C, S, eps, CG, mean, rstd);
    if (blockIdx.x == 0 && threadIdx.x < CG) {
#pragma unroll
        for (int i = 0; i < VN; ++i) { stat[((int64_t)b * C + g * VN + i) * 2] = mean[i]; stat[((int64_t)b * C + g * VN + i) * 2 + 1] = rstd[i]; }
    }
    const int64_t off = (int64_t)b * L * C + g * VN;
    for (int64_t l = (int64_t)blockIdx.x * (NT / CG) + threadIdx.x / CG; l < L; l += (int64_t)gridDim.x * (NT / CG)) {
        float av[VN], uv[VN];
        load_vec(a + off + l * C, av);
        load_vec(u + off + l * C, uv);
#pragma unroll
        for (int i = 0; i < VN; ++i) av[i] += gelu_t<T>((uv[i] - mean[i]) * rstd[i]);
        store_vec(y + off + l * C, av);
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void inorm_gelu_bwd_sums_kernel(const T *__restrict__ gy, const T *__restrict__ u, const float *__restrict__ stat,
                                                                  float *__restrict__ part, int64_t L, int C, int S) {
    constexpr int VN = Vec16<T>::N;
    __shared__ float lds[NT * VN];
    const int CG = C / VN, g = threadIdx.x % CG, b = blockIdx.y, s = blockIdx.x;
    int64_t lo, hi;
    slice_bounds(L, S, s, lo, hi);
    float mean[VN], rstd[VN], s1[VN], s2[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) {
        mean[i] = stat[((int64_t)b * C + g * VN + i) * 2];
        rstd[i] = stat[((int64_t)b * C + g * VN + i) * 2 + 1];
        s1[i] = s2[i] = 0.f;
    }
    const int64_t off = (int64_t)b * L * C + g * VN;
    for (int64_t l = lo + threadIdx.x / CG; l < hi; l += NT / CG) {
        float gv[VN], uv[VN];
        load_vec(gy + off + l * C, gv);
        load_vec(u + off + l * C, uv);
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            const float n = (uv[i] - mean[i]) * rstd[i], gn = gv[i] * gelu_grad_t<T>(n);
            s1[i] += gn;
            s2[i] += gn * n;
        }
    }
    group_reduce<VN>(s1, lds, CG);
    group_reduce<VN>(s2, lds, CG);
    if (threadIdx.x < CG) {
        float *o = part + (((int64_t)b * S + s) * C + g * VN) * 2;
#pragma unroll
        for (int i = 0; i < VN; ++i) { o[2 * i] = s1[i]; o[2 * i + 1] = s2[i]; }
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void inorm_gelu_bwd_kernel(const T *__restrict__ gy, const T *__restrict__ u, const float *__restrict__ stat,
                                                             const float *__restrict__ part, T *__restrict__ du, int64_t L, int C, int S) {
    constexpr int VN = Vec16<T>::N;
    __shared__ float lds[NT * VN];
    const int CG = C / VN, g = threadIdx.x % CG, b = blockIdx.y;
    float mean[VN], rstd[VN], m1[VN], m2[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) {
        mean[i] = stat[((int64_t)b * C + g * VN + i) * 2];
        rstd[i] = stat[((int64_t)b * C + g * VN + i) * 2 + 1];
        m1[i] = m2[i] = 0.f;
    }
    for (int s = threadIdx.x / CG; s < S; s += NT / CG) {      // the group's threads share the slices: one load each, then an LDS tree
        const float *p = part + (((int64_t)b * S + s) * C + g * VN) * 2;
#pragma unroll
        for (int i = 0; i < VN; ++i) { m1[i] += p[2 * i]; m2[i] += p[2 * i + 1]; }
    }
    group_reduce<VN>(m1, lds, CG);
    group_reduce<VN>(m2, lds, CG);
    const float inv = 1.0f / (float)L;
#pragma unroll
    for (int i = 0; i < VN; ++i) { m1[i] *= inv; m2[i] *= inv; }
    const int64_t off = (int64_t)b * L * C + g * VN;
    for (int64_t l = (int64_t)blockIdx.x * (NT / CG) + threadIdx.x / CG; l < L; l += (int64_t)gridDim.x * (NT / CG)) {
        float gv[VN], uv[VN];
        load_vec(gy + off + l * C, gv);
        load_vec(u + off + l * C, uv);
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            const float n = (uv[i] - mean[i]) * rstd[i], gn = gv[i] * gelu_grad_t<T>(n);
            gv[i] = rstd[i] * (gn - m1[i] - n * m2[i]);
        }
        store_vec(du + off + l * C, gv);
    }
}

bool shape_ok(int64_t B, int64_t L, int C, int S, int esz) {
    if (B <= 0 || L <= 0 || C <= 0 || S <= 0 || S > 4096) return false;
    const int vn = 16 / esz;
    if (C % vn) return false;
    const int cg = C / vn;
    return cg <= NT && NT % cg == 0 && (cg & (cg - 1)) == 0;
}

}  // namespace

extern "C" int gwd_inorm_gelu_forward(const void *a, const void *u, void *y, float *part, float *stat, int64_t B, int64_t L, int32_t C,
                                      int32_t S, float eps, int32_t dtype, void *stream) {
    if (!a || !u || !y || !part || !stat) return -1;
    const int esz = dtype == GWD_BF16 ? 2 : (dtype == GWD_F32 ? 4 : 0);
    if (!esz) return -2;
    if (!shape_ok(B, L, C, S, esz)) return -4;
    hipStream_t st = (hipStream_t)stream;
    const dim3 gs(S, (unsigned)B);
    const int cg = C / (16 / esz);
    int64_t nb = (L + NT / cg - 1) / (NT / cg);
    const dim3 ga((unsigned)(nb > 256 ? 256 : nb), (unsigned)B);
    if (dtype == GWD_BF16) {
        inorm_stats_kernel<__bf16><<<gs, NT, 0, st>>>((const __bf16 *)u, part, L, C, S);
        inorm_gelu_fwd_kernel<__bf16><<<ga, NT, 0, st>>>((const __bf16 *)a, (const __bf16 *)u, part, (__bf16 *)y, stat, L, C, S, eps);
    } else {
        inorm_stats_kernel<float><<<gs, NT, 0, st>>>((const float *)u, part, L, C, S);
        inorm_gelu_fwd_kernel<float><<<ga, NT, 0, st>>>((const float *)a, (const float *)u, part, (float *)y, stat, L, C, S, eps);
    }
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_inorm_gelu_backward(const void *gy, const void *u, const float *stat, float *part, void *du, int64_t B, int64_t L,
                                       int32_t C, int32_t S, int32_t dtype, void *stream) {
    if (!gy || !u || !stat || !part || !du) return -1;
    const int esz = dtype == GWD_BF16 ? 2 : (dtype == GWD_F32 ? 4 : 0);
    if (!esz) return -2;
    if (!shape_ok(B, L, C, S, esz)) return -4;
    hipStream_t st = (hipStream_t)stream;
    const dim3 gs(S, (unsigned)B);
    const int cg = C / (16 / esz);
    int64_t nb = (L + NT / cg - 1) / (NT / cg);
    const dim3 ga((unsigned)(nb > 256 ? 256 : nb), (unsigned)B);
    if (dtype == GWD_BF16) {
        inorm_gelu_bwd_sums_kernel<__bf16><<<gs, NT, 0, st>>>((const __bf16 *)gy, (const __bf16 *)u, stat, part, L, C, S);
        inorm_gelu_bwd_kernel<__bf16><<<ga, NT, 0, st>>>((const __bf16 *)gy, (const __bf16 *)u, stat, part, (__bf16 *)du, L, C, S);
    } else {
        inorm_gelu_bwd_sums_kernel<float><<<gs, NT, 0, st>>>((const float *)gy, (const float *)u, stat, part, L, C, S);
        inorm_gelu_bwd_kernel<float><<<ga, NT, 0, st>>>((const float *)gy, (const float *)u, stat, part, (float *)du, L, C, S);
    }
    GWD_CHECK_LAUNCH();
    return 0;
}
