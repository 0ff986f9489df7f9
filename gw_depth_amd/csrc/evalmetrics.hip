// Dense evaluation metrics on the device: the clamp / validity mask of src/engine_glassrgbd.py:249-253, the nine
// depth measures of src/util/metrics.py:198-218 and the 2x2 confusion counts of :37-99 as ONE streaming pass over
// (prediction, GT, logits, labels) - 18 bytes per pixel in fp32 - instead of a device->host copy and per-image numpy.
//
// Two launches, no atomics, no memset: eval_partial_kernel writes one record per (image, workgroup) into the caller's
// workspace; eval_finalize_kernel folds them in a fixed order (bit-reproducible), turns the sums into the measures of
// each image and adds them to the caller's running totals in image order, like the reference's loop does.
#include "common.h"

namespace {

constexpr int NSUM = GWD_EVAL_NSUM;   // n, d1, d2, d3, sum sq, sum abs_rel, sum sq_rel, sum err, sum err^2, sum |log10|

__device__ __forceinline__ float log_rn(float v) { return (float)log((double)v); }        // = correctly rounded logf
__device__ __forceinline__ float log10_rn(float v) { return (float)log10((double)v); }

template <typename TD, typename TS>
__global__ __launch_bounds__(256) void eval_partial_kernel(const TD *__restrict__ pred, const float *__restrict__ gt,
                                                           const TS *__restrict__ seg, int64_t seg_sb, int64_t seg_sp,
                                                           int64_t seg_sc, const int64_t *__restrict__ seg_gt,
                                                           double *__restrict__ psum, long long *__restrict__ pconf,
                                                           int64_t HW, float dmin, float dmax) {
    const int b = blockIdx.y, nb = gridDim.x;
    double s[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) s[k] = 0.0;
    int cf[4] = {0, 0, 0, 0};                       // per-thread pixel counts stay far below 2^31
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)nb * 256) {
        if (pred) {
            const float g = gt[b * HW + i];
            float p = to_f32(pred[b * HW + i]);
            // engine_glassrgbd.py:249-252, in that order; a NaN fails both comparisons and is caught last
            if (p < dmin) p = dmin;
            if (p > dmax) p = dmax;
            if (p != p) p = dmin;
            if (g > dmin && g < dmax) {             // :253
                const float thr = fmaxf(__fdiv_rn(g, p), __fdiv_rn(p, g));
                const float diff = g - p, sq = diff * diff;
                const float err = log_rn(p) - log_rn(g);
                s[0] += 1.0;
                s[1] += thr < 1.25f ? 1.0 : 0.0;
                s[2] += thr < 1.5625f ? 1.0 : 0.0;
                s[3] += thr < 1.953125f ? 1.0 : 0.0;
                s[4] += (double)sq;
                s[5] += (double)__fdiv_rn(fabsf(diff), g);
                s[6] += (double)__fdiv_rn(sq, g);
                s[7] += (double)err;
                s[8] += (double)(err * err);
                s[9] += (double)fabsf(log10_rn(p) - log10_rn(g));
            }
        }
        if (seg) {
            const int64_t t = seg_gt[b * HW + i];
            if (t == 0 || t == 1) {                 // 255 = ignore (metrics.py:69); other labels fall outside the 2x2 table
                const float l0 = to_f32(seg[b * seg_sb + i * seg_sp]), l1 = to_f32(seg[b * seg_sb + i * seg_sp + seg_sc]);
                const int c = (l1 > l0 || (l1 != l1 && l0 == l0)) ? 1 : 0;       // argmax: first maximum, NaN counts as maximal
                cf[t * 2 + c] += 1;
            }
        }
    }
    __shared__ double sh[4][NSUM];
    __shared__ int shc[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NSUM; ++k) {
        const double v = wave_sum_d(s[k]);
        if (lane == 0) sh[wave][k] = v;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int v = cf[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) shc[wave][k] = v;
    }
    __syncthreads();
    const int64_t rec = (int64_t)b * nb + blockIdx.x;
    if (threadIdx.x < NSUM) psum[rec * NSUM + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    if (threadIdx.x >= 64 && threadIdx.x < 68) {
        const int k = threadIdx.x - 64;
        pconf[rec * 4 + k] = (long long)shc[0][k] + shc[1][k] + shc[2][k] + shc[3][k];
    }
}

__global__ __launch_bounds__(256) void eval_finalize_kernel(const double *__restrict__ psum, const long long *__restrict__ pconf,
                                                            int nb, int B, double *__restrict__ measures,
                                                            double *__restrict__ running, long long *__restrict__ confusion,
                                                            int with_depth, int with_seg) {
    for (int b = threadIdx.x; b < B; b += 256) {
        if (!with_depth) continue;
        double s[NSUM];
        for (int k = 0; k < NSUM; ++k) s[k] = 0.0;
        for (int j = 0; j < nb; ++j)
            for (int k = 0; k < NSUM; ++k) s[k] += psum[((int64_t)b * nb + j) * NSUM + k];
        const double n = s[0];                       // n == 0: every measure is NaN, as numpy's mean of an empty array
        const double me = s[7] / n, me2 = s[8] / n;
        double *m = measures + (int64_t)b * 9;       // order of engine_glassrgbd.py:204
        m[0] = sqrt(me2 - me * me) * 100.0;          // silog
        m[1] = s[5] / n;                             // abs_rel
        m[2] = s[9] / n;                             // log10
        m[3] = sqrt(s[4] / n);                       // rms
        m[4] = s[6] / n;                             // sq_rel
        m[5] = sqrt(me2);                            // log_rms  ((log g - log p)^2 == err^2)
        m[6] = s[1] / n;
        m[7] = s[2] / n;
        m[8] = s[3] / n;
    }
    __syncthreads();
    if (with_seg && threadIdx.x < 4) {
        long long c = 0;
        for (int64_t j = 0; j < (int64_t)B * nb; ++j) c += pconf[j * 4 + threadIdx.x];
        confusion[threadIdx.x] += c;
    }
    if (with_depth && threadIdx.x == 64) {           // engine_glassrgbd.py:262-263: running sums in image order
        for (int b = 0; b < B; ++b) {
            for (int k = 0; k < 9; ++k) running[k] += measures[(int64_t)b * 9 + k];
            running[9] += 1.0;
        }
    }
}

int eval_blocks(int64_t HW) {
    int64_t nb = (HW + 256 * 8 - 1) / (256 * 8);     // >= 8 pixels per thread
    return (int)(nb < 1 ? 1 : (nb > 256 ? 256 : nb));
}

}  // namespace

// bytes of gwd_query_workspace(GWD_WS_EVAL, {B, HW}) (optim.hip)
int64_t gwd_eval_workspace_bytes(int64_t B, int64_t HW) {
    if (B <= 0 || HW <= 0) return -1;
    return B * eval_blocks(HW) * (NSUM * (int64_t)sizeof(double) + 4 * (int64_t)sizeof(long long));
}

extern "C" int gwd_eval_accumulate(const void *pred_depth, const float *gt_depth, const void *seg_logits, int64_t seg_sb,
                                   int64_t seg_sp, int64_t seg_sc, const int64_t *seg_gt, void *workspace, double *measures,
                                   double *running, int64_t *confusion, int32_t B, int64_t HW, float min_depth,
                                   float max_depth, int32_t depth_dtype, int32_t seg_dtype, void *stream) {
    if (B <= 0 || B > 65535 || HW <= 0 || !workspace) return -1;
    if ((pred_depth == nullptr) != (gt_depth == nullptr) || (seg_logits == nullptr) != (seg_gt == nullptr)) return -1;
    if (!pred_depth && !seg_logits) return -1;
    if (pred_depth && (!measures || !running)) return -1;
    if (seg_logits && !confusion) return -1;
    if ((depth_dtype != GWD_F32 && depth_dtype != GWD_BF16) || (seg_dtype != GWD_F32 && seg_dtype != GWD_BF16)) return -2;
    hipStream_t s = (hipStream_t)stream;
    const int nb = eval_blocks(HW);
    double *psum = (double *)workspace;
    long long *pconf = (long long *)(psum + (int64_t)B * nb * NSUM);
    const dim3 grid(nb, B);
#define EVAL_LAUNCH(TD, TS)                                                                                                  \
    eval_partial_kernel<TD, TS><<<grid, 256, 0, s>>>((const TD *)pred_depth, gt_depth, (const TS *)seg_logits, seg_sb, seg_sp, \
                                                     seg_sc, seg_gt, psum, pconf, HW, min_depth, max_depth)
    if (depth_dtype == GWD_F32 && seg_dtype == GWD_F32) EVAL_LAUNCH(float, float);
    else if (depth_dtype == GWD_F32) EVAL_LAUNCH(float, __bf16);
    else if (seg_dtype == GWD_F32) EVAL_LAUNCH(__bf16, float);
    else EVAL_LAUNCH(__bf16, __bf16);
#undef EVAL_LAUNCH
    GWD_CHECK_LAUNCH();
    eval_finalize_kernel<<<1, 256, 0, s>>>(psum, pconf, nb, B, measures, running, (long long *)confusion, pred_depth != nullptr,
                                           seg_logits != nullptr);
    GWD_CHECK_LAUNCH();
    return 0;
}
