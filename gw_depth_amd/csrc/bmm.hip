// Strided batched GEMM on the matrix cores:  C[b] (M x N) = alpha * A[b] (M x K) * B[b] (N x K)^T  over a two-level batch, each operand
// stored either "row-major in k" ([row][k], k contiguous) or "k-major" ([k][row], row contiguous) - the four combinations are the
// products and both gradients of a @ b^T and a @ b without a transposed copy of anything.
//
// Users (all small, or outside the timed mode): the point heads' affinity map xg @ refer^T and its gradients
// (src/models/points/points_sample.py:271-279; was torch.bmm -> rocBLAS), and the attention products of the fp32 parity mode
// (src/models/multi_head_attention.py:347-371; was torch.matmul -> rocBLAS) - so neither mode runs a vendor GEMM.
// bf16: v_mfma_f32_32x32x16_bf16; fp32: the exact v_mfma_f32_32x32x2_f32 (an fmaf chain: parity mode).
//
// 64 x 64 tiles, 4 waves (2 x 2) of one 32 x 32 accumulator each, register-staged double-buffered LDS (no alignment requirement: a
// 16-byte vector is loaded whole when it is aligned and in range, element by element otherwise).  A k-major operand is loaded along
// its rows and scattered into the [row][k] LDS image.  A long reduction on few tiles (d refer = d rg^T @ xg: 19 200 pixels into an
// 80 x 64 matrix) is split over workgroups (grid.z) with fp32 atomics into a caller-zeroed fp32 result.
#include "common.h"

namespace {

template <typename T> struct BCfg;
template <> struct BCfg<float> { static constexpr int VEC = 4, BK = 16; };
template <> struct BCfg<__bf16> { static constexpr int VEC = 8, BK = 32; };

struct Operand {
    const void *p;
    int64_t sb0, sb1, ld;          // element strides of the two batch dims and of the outer matrix dim (the inner one is 1)
};

__device__ __forceinline__ f32x16 bmma(const bf16x8 &a, const bf16x8 &b, const f32x16 &c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 bmma(float a, float b, const f32x16 &c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// VEC consecutive elements starting at p[0], of which the first `n` (0..VEC) exist
template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T *p, int n, T (&out)[VEC]) {
    if (n >= VEC && (((uintptr_t)p) & 15) == 0) {
        *(uint4 *)out = *(const uint4 *)p;
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) out[e] = e < n ? p[e] : from_f32<T>(0.f);
    }
}

// KM: operand stored [k][row]; else [row][k].  Tile rows r0 .. r0+63 (of R), k range k0 .. k0+BK-1 (of k_end) -> regs (one vector per thread)
template <typename T, bool KM>
__device__ __forceinline__ void load_tile(const T *base, int64_t ld, int r0, int R, int k0, int k_end, int tid, T (&v)[BCfg<T>::VEC]) {
    constexpr int VEC = BCfg<T>::VEC, BK = BCfg<T>::BK;
    if constexpr (!KM) {
        constexpr int KV = BK / VEC;
        const int row = r0 + tid / KV, k = k0 + (tid % KV) * VEC;
        const int n = (row < R) ? max(0, min(VEC, k_end - k)) : 0;
        load_vec<T, VEC>(base + (int64_t)row * ld + k, n, v);
    } else {
        constexpr int RV = 64 / VEC;
        const int k = k0 + tid / RV, row = r0 + (tid % RV) * VEC;
        const int n = (k < k_end) ? max(0, min(VEC, R - row)) : 0;
        load_vec<T, VEC>(base + (int64_t)k * ld + row, n, v);
    }
}
template <typename T, bool KM>
__device__ __forceinline__ void store_tile(T *lds, int ldk, int tid, const T (&v)[BCfg<T>::VEC]) {
    constexpr int VEC = BCfg<T>::VEC, BK = BCfg<T>::BK;
    if constexpr (!KM) {
        constexpr int KV = BK / VEC;
        T *q = lds + (tid / KV) * ldk + (tid % KV) * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) q[e] = v[e];
    } else {
        constexpr int RV = 64 / VEC;
        const int k = tid / RV, row = (tid % RV) * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) lds[(row + e) * ldk + k] = v[e];
    }
}

template <typename T, bool AKM, bool BKM>
__global__ __launch_bounds__(256) void bmm_kernel(Operand A, Operand B, void *c, int64_t cb0, int64_t cb1, int64_t cld, int M, int N, int K, int nb1,
                                                  float alpha, int k_per_split, int atomic_out) {
    constexpr int VEC = BCfg<T>::VEC, BK = BCfg<T>::BK, LDK = BK + VEC;
    __shared__ __attribute__((aligned(16))) T As[2][64 * LDK];
    __shared__ __attribute__((aligned(16))) T Bs[2][64 * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (N + 63) / 64;
    const int m0 = ((int)blockIdx.x / tiles_n) * 64, n0 = ((int)blockIdx.x % tiles_n) * 64;
    const int b0 = (int)blockIdx.y / nb1, b1 = (int)blockIdx.y % nb1;
    const T *a = (const T *)A.p + b0 * A.sb0 + b1 * A.sb1;
    const T *b = (const T *)B.p + b0 * B.sb0 + b1 * B.sb1;
    const int k_begin = (int)blockIdx.z * k_per_split, k_end = min(K, k_begin + k_per_split);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int steps = (k_end - k_begin + BK - 1) / BK;
    T ra[VEC], rb[VEC];
    if (steps > 0) {
        load_tile<T, AKM>(a, A.ld, m0, M, k_begin, k_end, tid, ra);
        load_tile<T, BKM>(b, B.ld, n0, N, k_begin, k_end, tid, rb);
        store_tile<T, AKM>(As[0], LDK, tid, ra);
        store_tile<T, BKM>(Bs[0], LDK, tid, rb);
    }
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    for (int s = 0; s < steps; ++s) {
        const int cur = s & 1;
        if (s + 1 < steps) {
            load_tile<T, AKM>(a, A.ld, m0, M, k_begin + (s + 1) * BK, k_end, tid, ra);
            load_tile<T, BKM>(b, B.ld, n0, N, k_begin + (s + 1) * BK, k_end, tid, rb);
        }
        const T *Ab = As[cur] + (wm * 32 + fr) * LDK, *Bb = Bs[cur] + (wn * 32 + fr) * LDK;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks)
                acc = bmma(*(const bf16x8 *)(Ab + ks * 16 + fh * 8), *(const bf16x8 *)(Bb + ks * 16 + fh * 8), acc);
        } else {
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) acc = bmma(Ab[ks * 2 + fh], Bb[ks * 2 + fh], acc);
        }
        if (s + 1 < steps) {
            store_tile<T, AKM>(As[cur ^ 1], LDK, tid, ra);
            store_tile<T, BKM>(Bs[cur ^ 1], LDK, tid, rb);
        }
        __syncthreads();
    }
    // D layout of the 32 x 32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int n = n0 + wn * 32 + fr;
    if (n >= N) return;
    const int64_t cbase = b0 * cb0 + b1 * cb1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (m >= M) continue;
        const float v = acc[r] * alpha;
        if (atomic_out) unsafeAtomicAdd((float *)c + cbase + (int64_t)m * cld + n, v);
        else ((T *)c)[cbase + (int64_t)m * cld + n] = from_f32<T>(v);
    }
}

}  // namespace

extern "C" int gwd_bmm(const gwd_bmm_desc *d, void *stream) {
    if (!d || !d->a || !d->b || !d->c) return -1;
    if (d->dtype != GWD_F32 && d->dtype != GWD_BF16) return -2;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->nb0 <= 0 || d->nb1 <= 0 || d->splits <= 0) return -3;
    if ((int64_t)d->nb0 * d->nb1 > 65535 || d->splits > 65535) return -7;
    if (d->splits > 1 && !d->c_is_f32_accumulate) return -4;     // a split reduction adds into a zeroed fp32 result
    const Operand A{d->a, d->a_sb0, d->a_sb1, d->a_ld}, B{d->b, d->b_sb0, d->b_sb1, d->b_ld};
    const int bk = d->dtype == GWD_BF16 ? 32 : 16;
    int kps = (d->K + d->splits - 1) / d->splits;
    kps = (kps + bk - 1) / bk * bk;
    const dim3 grid(((d->M + 63) / 64) * ((d->N + 63) / 64), d->nb0 * d->nb1, (d->K + kps - 1) / kps);
    hipStream_t s = (hipStream_t)stream;
    const int atomic = d->c_is_f32_accumulate ? 1 : 0;
#define BMM_GO(T_, AK_, BK_) bmm_kernel<T_, AK_, BK_><<<grid, 256, 0, s>>>(A, B, d->c, d->c_sb0, d->c_sb1, d->c_ld, d->M, d->N, d->K, d->nb1, d->alpha, kps, atomic)
#define BMM_T(T_)                                          \
    if (d->a_kmajor && d->b_kmajor) BMM_GO(T_, true, true);   \
    else if (d->a_kmajor) BMM_GO(T_, true, false);         \
    else if (d->b_kmajor) BMM_GO(T_, false, true);         \
    else BMM_GO(T_, false, false)
    if (d->dtype == GWD_BF16) { BMM_T(__bf16); } else { BMM_T(float); }
#undef BMM_T
#undef BMM_GO
    GWD_CHECK_LAUNCH();
    return 0;
}
