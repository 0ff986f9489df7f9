// Flat-buffer optimizer step: global gradient norm, clip and AdamW in two streaming passes over the
// 66 M trainable elements (4 reads + 3 writes of fp32 per element + an optional bf16 shadow write).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sqnorm_kernel(const float *__restrict__ g, double *__restrict__ sq, int64_t n) {
    __shared__ double sh[4];
    double s = 0.0;
    const int64_t n4 = n / 4;
    const f32x4 *g4 = (const f32x4 *)g;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = g4[i];
        s += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[n4 * 4 + threadIdx.x];
        s += (double)v * v;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sq, sh[0] + sh[1] + sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void adamw_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                    float *__restrict__ v, __bf16 *__restrict__ p16,
                                                    const double *__restrict__ sq, int64_t n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2, float max_norm,
                                                    float grad_scale) {
    float coef = grad_scale;
    if (sq && max_norm > 0.f) {
        // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
        const float total = (float)sqrt(sq[0]) * grad_scale;
        coef *= fminf(max_norm / (total + 1e-6f), 1.0f);
    }
    const float step = lr / bc1, rs2 = rsqrtf(bc2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);     // lerp, as torch's exp_avg.lerp_
        const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
        pi -= step * mi / (sqrtf(vi) * rs2 + eps);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
        if (p16) p16[i] = (__bf16)pi;
    }
}

}  // namespace

extern "C" int gwd_sqnorm(const float *g, double *sq, int64_t n, void *stream) {
    if (!g || !sq || n < 0) return -1;
    if (((uintptr_t)g & 15) != 0) return -3;
    if (n == 0) return 0;
    int64_t b = (n / 4 + 255) / 256;
    const int grid = (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
    sqnorm_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(g, sq, n);
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_adamw_step(float *p, const float *g, float *m, float *v, void *p_bf16, const double *sq, int64_t n,
                              float lr, float beta1, float beta2, float eps, float weight_decay, float bias_corr1,
                              float bias_corr2, float max_norm, float grad_scale, void *stream) {
    if (!p || !g || !m || !v || n < 0) return -1;
    if (n == 0) return 0;
    int64_t b = (n + 255) / 256;
    const int grid = (int)(b > 4096 ? 4096 : b);
    adamw_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p, g, m, v, (__bf16 *)p_bf16, sq, n, lr, beta1, beta2, eps,
                                                        weight_decay, bias_corr1, bias_corr2, max_norm, grad_scale);
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_version(void) { return GWD_VERSION; }

int64_t gwd_eval_workspace_bytes(int64_t B, int64_t HW);    // evalmetrics.hip
int64_t gwd_plane_workspace_bytes(int64_t P, int64_t HW);   // planeloss.hip

extern "C" int64_t gwd_query_workspace(int32_t op, const int64_t *dims, int32_t ndims) {
    if (!dims) return -1;
    switch (op) {
        case GWD_WS_INORM_GELU: return ndims == 3 ? dims[0] * dims[1] * dims[2] * 2 * (int64_t)sizeof(float) : -1;
        case GWD_WS_RESAMPLE_BWD: return ndims == 4 ? dims[0] * dims[1] * dims[2] * dims[3] * (int64_t)sizeof(float) : -1;
        case GWD_WS_EVAL: return ndims == 2 ? gwd_eval_workspace_bytes(dims[0], dims[1]) : -1;
        case GWD_WS_PLANE: return ndims == 2 ? gwd_plane_workspace_bytes(dims[0], dims[1]) : -1;
        default: return -1;
    }
}
extern "C" const char *gwd_arch(void) { return "gfx950"; }
