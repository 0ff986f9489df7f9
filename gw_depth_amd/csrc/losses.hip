// Loss reductions: SiLog (masked, per-scale nearest-resized GT gathered on the fly) and 2-class CE.
// Sums are accumulated in double (block shuffle-reduce, then one f64 atomic per block), which keeps
// the fp32 loss value run-to-run stable without a second pass.
#include "common.h"

namespace {

__device__ __forceinline__ double block_sum_d(double v, double *sh) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    __syncthreads();
    return r;  // valid on thread 0
}

// nearest source index exactly as aten::upsample_nearest2d: min(floor(dst * in/out), in-1)
__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
    return min((int)floorf((float)dst * ((float)in / (float)out)), in - 1);
}

// One (image, row) of the prediction at a time: the GT row and the row's base offsets are computed once (no per-element 64-bit
// div / mod - they were most of the kernel's instructions), threads walk the columns.
template <typename T>
__device__ __forceinline__ bool silog_term(const T *prow, const float *grow, int x, int w, int W, int log_err, float &d, float &p) {
    const float g = grow[nearest_src(x, W, w)];
    if (!(g >= 0.2f && g < 10.0f)) return false;
    p = to_f32(prow[x]);
    d = log_err ? (logf(p) - logf(g)) : ((p + logf(p)) - (g + logf(g)));
    return true;
}

template <typename T>
__global__ __launch_bounds__(256) void silog_sums_kernel(const T *__restrict__ pred, const float *__restrict__ gt,
                                                         double *__restrict__ sums, int rows, int h, int w, int H,
                                                         int W, int log_err) {
    __shared__ double sh[4];
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int b = row / h, y = row - b * h;
        const T *prow = pred + (int64_t)row * w;
        const float *grow = gt + ((int64_t)b * H + nearest_src(y, H, h)) * W;
        for (int x = threadIdx.x; x < w; x += blockDim.x) {
            float d, p;
            if (silog_term(prow, grow, x, w, W, log_err, d, p)) {
                s1 += d;
                s2 += (double)d * d;
                cnt += 1.0;
            }
        }
    }
    s1 = block_sum_d(s1, sh);
    s2 = block_sum_d(s2, sh);
    cnt = block_sum_d(cnt, sh);
    if (threadIdx.x == 0) {
        atomicAdd(sums + 0, s1);
        atomicAdd(sums + 1, s2);
        atomicAdd(sums + 2, cnt);
    }
}

template <typename T>
__global__ void silog_bwd_kernel(const T *__restrict__ pred, const float *__restrict__ gt, const double *__restrict__ sums,
                                 const float *__restrict__ gloss, float loss_weight, float lambda, T *__restrict__ gpred,
                                 int rows, int h, int w, int H, int W, int log_err) {
    const double n = sums[2];
    const double mean = sums[0] / n, var = sums[1] / n - (double)lambda * mean * mean;
    // L = 10 sqrt(var);  dL/dd_i = 10/(2 sqrt(var)) * (2 d_i / n - 2 lambda mean / n)
    const float c = (float)(10.0 / sqrt(var) / n) * loss_weight * gloss[0];
    const float lm = (float)((double)lambda * mean);
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int b = row / h, y = row - b * h;
        const T *prow = pred + (int64_t)row * w;
        const float *grow = gt + ((int64_t)b * H + nearest_src(y, H, h)) * W;
        T *orow = gpred + (int64_t)row * w;
        for (int x = threadIdx.x; x < w; x += blockDim.x) {
            float d, p, g = 0.f;
            if (silog_term(prow, grow, x, w, W, log_err, d, p)) g = c * (d - lm) * (log_err ? 1.0f / p : 1.0f + 1.0f / p);
            orow[x] = from_f32<T>(g);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void seg_ce_sum_kernel(const T *__restrict__ logits, const int64_t *__restrict__ target,
                                                         double *__restrict__ sum, int64_t P) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = to_f32(logits[2 * i]), b = to_f32(logits[2 * i + 1]);
        const float mx = fmaxf(a, b);
        const float lse = mx + logf(expf(a - mx) + expf(b - mx));
        s += (double)(lse - (target[i] ? b : a));
    }
    s = block_sum_d(s, sh);
    if (threadIdx.x == 0) atomicAdd(sum, s);
}

template <typename T>
__global__ void seg_ce_bwd_kernel(const T *__restrict__ logits, const int64_t *__restrict__ target,
                                  const float *__restrict__ gloss, float scale, T *__restrict__ gl, int64_t P) {
    const float c = gloss[0] * scale / (float)P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = to_f32(logits[2 * i]), b = to_f32(logits[2 * i + 1]);
        const float mx = fmaxf(a, b);
        const float ea = expf(a - mx), eb = expf(b - mx), inv = 1.0f / (ea + eb);
        const int t = target[i] != 0;
        gl[2 * i] = from_f32<T>(c * (ea * inv - (t ? 0.f : 1.f)));
        gl[2 * i + 1] = from_f32<T>(c * (eb * inv - (t ? 1.f : 0.f)));
    }
}

inline int flat_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int gwd_silog_sums(const void *pred, const float *gt, double *sums, int32_t B, int32_t h, int32_t w, int32_t H,
                              int32_t W, int32_t log_depth_error, int32_t dtype, void *stream) {
    if (!pred || !gt || !sums || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return -1;
    if ((int64_t)B * h >= (1LL << 31)) return -7;
    const int rows = B * h, grid = rows < 512 ? rows : 512, block = w > 128 ? 256 : (w > 64 ? 128 : 64);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GWD_BF16)
        silog_sums_kernel<__bf16><<<grid, block, 0, s>>>((const __bf16 *)pred, gt, sums, rows, h, w, H, W, log_depth_error);
    else if (dtype == GWD_F32)
        silog_sums_kernel<float><<<grid, block, 0, s>>>((const float *)pred, gt, sums, rows, h, w, H, W, log_depth_error);
    else
        return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

// loss = scale * sqrt(E[d^2] - lambda E[d]^2) from the three sums, in f64 as the torch expression it replaces (eight scalar launches)
__global__ void silog_finalize_kernel(const double *__restrict__ sums, double lambda, double scale, float *__restrict__ loss) {
    const double n = sums[2], mean = sums[0] / n;
    *loss = (float)(sqrt(sums[1] / n - lambda * mean * mean) * scale);
}

extern "C" int gwd_silog_finalize(const double *sums, float lambda, float scale, float *loss, void *stream) {
    if (!sums || !loss) return -1;
    silog_finalize_kernel<<<1, 1, 0, (hipStream_t)stream>>>(sums, (double)lambda, (double)scale, loss);
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_silog_backward(const void *pred, const float *gt, const double *sums, const float *gloss,
                                  float loss_weight, float lambda, void *gpred, int32_t B, int32_t h, int32_t w,
                                  int32_t H, int32_t W, int32_t log_depth_error, int32_t dtype, void *stream) {
    if (!pred || !gt || !sums || !gloss || !gpred || B <= 0 || h <= 0 || w <= 0) return -1;
    if ((int64_t)B * h >= (1LL << 31)) return -7;
    const int rows = B * h, grid = rows < 2048 ? rows : 2048, block = w > 128 ? 256 : (w > 64 ? 128 : 64);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GWD_BF16)
        silog_bwd_kernel<__bf16><<<grid, block, 0, s>>>((const __bf16 *)pred, gt, sums, gloss, loss_weight, lambda, (__bf16 *)gpred, rows, h, w, H, W, log_depth_error);
    else if (dtype == GWD_F32)
        silog_bwd_kernel<float><<<grid, block, 0, s>>>((const float *)pred, gt, sums, gloss, loss_weight, lambda, (float *)gpred, rows, h, w, H, W, log_depth_error);
    else
        return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_seg_ce_sum(const void *logits, const int64_t *target, double *sum, int64_t P, int32_t dtype, void *stream) {
    if (!logits || !target || !sum || P <= 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GWD_BF16)
        seg_ce_sum_kernel<__bf16><<<flat_grid(P), 256, 0, s>>>((const __bf16 *)logits, target, sum, P);
    else if (dtype == GWD_F32)
        seg_ce_sum_kernel<float><<<flat_grid(P), 256, 0, s>>>((const float *)logits, target, sum, P);
    else
        return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_seg_ce_backward(const void *logits, const int64_t *target, const float *gloss, float scale, void *glogits,
                                   int64_t P, int32_t dtype, void *stream) {
    if (!logits || !target || !gloss || !glogits || P <= 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GWD_BF16)
        seg_ce_bwd_kernel<__bf16><<<flat_grid(P), 256, 0, s>>>((const __bf16 *)logits, target, gloss, scale, (__bf16 *)glogits, P);
    else if (dtype == GWD_F32)
        seg_ce_bwd_kernel<float><<<flat_grid(P), 256, 0, s>>>((const float *)logits, target, gloss, scale, (float *)glogits, P);
    else
        return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

// ----------------------------------------------------------------------------------------------------------------
// Anchor-weighted depth of PointBasedPred (src/models/points/points_sample.py:277-279): pred[b][p] = sum_r att[b][p][r] *
// anchor[b][r] over the S point channels - and its gradients.  The batched-GEMM library runs these matrix-VECTOR products
// (N = 1) at 60-175 us each on a 49 MB operand; they are one streaming pass.
namespace {

template <typename T>
__global__ __launch_bounds__(256) void anchor_depth_fwd_kernel(const T *__restrict__ att, const float *__restrict__ anchor,
                                                               float *__restrict__ pred, int64_t P, int R) {
    const int b = blockIdx.y;
    extern __shared__ float an[];
    for (int r = threadIdx.x; r < R; r += 256) an[r] = anchor[(int64_t)b * R + r];
    __syncthreads();
    // 16 lanes share one pixel row (coalesced 32-byte steps along the row), four pixels per wave, DPP segment sum
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4, wave = threadIdx.x >> 6;
    const int64_t per_round = (int64_t)gridDim.x * 16;
    const int64_t rounds = (P + per_round - 1) / per_round;
    for (int64_t it = 0; it < rounds; ++it) {
        const int64_t p = it * per_round + ((int64_t)blockIdx.x * 4 + wave) * 4 + grp;
        const bool ok = p < P;
        const T *row = att + ((int64_t)b * P + (ok ? p : 0)) * R;
        float s = 0.f;
        for (int r = sub; r < R; r += 16) s += to_f32(row[r]) * an[r];
        s = segment_sum<16>(s);
        if (ok && sub == 0) pred[(int64_t)b * P + p] = s;
    }
}

// datt[b][p][r] = g[b][p] * anchor[b][r];  danchor[b][r] += sum_p att[b][p][r] * g[b][p]  (block partials, one atomic each)
template <typename T>
__global__ __launch_bounds__(256) void anchor_depth_bwd_kernel(const T *__restrict__ att, const float *__restrict__ anchor,
                                                               const float *__restrict__ g, T *__restrict__ datt,
                                                               float *__restrict__ danchor, int64_t P, int R) {
    const int b = blockIdx.y;
    extern __shared__ float sh[];
    float *an = sh, *acc = sh + R;                       // [R] anchors, [R] partial sums of this workgroup
    for (int r = threadIdx.x; r < R; r += 256) {
        an[r] = anchor[(int64_t)b * R + r];
        acc[r] = 0.f;
    }
    __syncthreads();
    // thread = (pixel slot, channel r): consecutive threads walk consecutive channels of one pixel row -> coalesced
    const int rr = threadIdx.x % R, slots = 256 / R, slot = threadIdx.x / R;
    float mine = 0.f;
    if (slot < slots) {
        for (int64_t p = (int64_t)blockIdx.x * slots + slot; p < P; p += (int64_t)gridDim.x * slots) {
            const int64_t o = ((int64_t)b * P + p) * R + rr;
            const float gp = g[(int64_t)b * P + p];
            mine += to_f32(att[o]) * gp;
            if (datt) datt[o] = from_f32<T>(gp * an[rr]);
        }
        atomicAdd(&acc[rr], mine);
    }
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += 256) unsafeAtomicAdd(danchor + (int64_t)b * R + r, acc[r]);
}

}  // namespace

extern "C" int gwd_anchor_depth_forward(const void *att, const float *anchor, float *pred, int32_t B, int64_t P, int32_t R,
                                        int32_t dtype, void *stream) {
    if (!att || !anchor || !pred || B <= 0 || P <= 0 || R <= 0 || R > 256) return -1;
    int64_t nb = (P + 255) / 256;
    if (nb > 1024) nb = 1024;
    const dim3 grid((unsigned)nb, (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GWD_BF16) anchor_depth_fwd_kernel<__bf16><<<grid, 256, R * sizeof(float), s>>>((const __bf16 *)att, anchor, pred, P, R);
    else if (dtype == GWD_F32) anchor_depth_fwd_kernel<float><<<grid, 256, R * sizeof(float), s>>>((const float *)att, anchor, pred, P, R);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_anchor_depth_backward(const void *att, const float *anchor, const float *gpred, void *datt, float *danchor,
                                         int32_t B, int64_t P, int32_t R, int32_t dtype, void *stream) {
    if (!att || !anchor || !gpred || !danchor || B <= 0 || P <= 0 || R <= 0 || R > 256) return -1;
    const int slots = 256 / R;
    int64_t nb = (P + (int64_t)slots * 16 - 1) / ((int64_t)slots * 16);
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    const dim3 grid((unsigned)nb, (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GWD_BF16) anchor_depth_bwd_kernel<__bf16><<<grid, 256, 2 * R * sizeof(float), s>>>((const __bf16 *)att, anchor, gpred, (__bf16 *)datt, danchor, P, R);
    else if (dtype == GWD_F32) anchor_depth_bwd_kernel<float><<<grid, 256, 2 * R * sizeof(float), s>>>((const float *)att, anchor, gpred, (float *)datt, danchor, P, R);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}
