// Batch assembly on the device: the tail of the reference's input pipeline - ToTensor + Normalize
// (src/datasets/transforms_depth.py:618-660 -> torchvision.transforms.functional.to_tensor / normalize), the dataset's
// depth / segmentation conversion (src/datasets/glassrgbd_norhint.py:277-281) and the zero-padding collate with its
// padding mask (src/util/misc.py:273-313) - as ONE launch over the raw decoded images: the host uploads uint8 RGB,
// 16-bit depth and uint8 labels (5 bytes per pixel instead of 24) and never touches a float.
//
// Per pixel, in the reference's fp32 order: x/255 (IEEE division), - mean, / std (IEEE division); depth_mm / 1000;
// label > 0.  The image jobs travel by value in the kernel arguments.
#include "common.h"

namespace {

struct CollateArgs {
    gwd_image_job j[GWD_COLLATE_BATCH];
    float mean[3], std[3];
    int n, H, W;
};

template <typename T>
__global__ __launch_bounds__(256) void collate_kernel(const CollateArgs a, T *__restrict__ images, unsigned char *__restrict__ mask,
                                                      float *__restrict__ depth, int64_t *__restrict__ seg) {
    const int b = blockIdx.y;
    const gwd_image_job job = a.j[b];
    const int HW = a.H * a.W;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        const int y = p / a.W, x = p - y * a.W;
        const bool in = y < job.h && x < job.w;
        const size_t o = (size_t)b * HW + p;
        float c[3] = {0.f, 0.f, 0.f};
        float d = 0.f;
        long long s = 0;
        if (in) {
            const size_t q = (size_t)y * job.w + x;
            if (job.rgb) {
                const unsigned char *px = (const unsigned char *)job.rgb + q * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) c[k] = __fdiv_rn(__fdiv_rn((float)px[k], 255.0f) - a.mean[k], a.std[k]);
            }
            if (job.depth_mm) d = __fdiv_rn((float)((const int32_t *)job.depth_mm)[q], 1000.0f);
            if (job.labels) s = ((const unsigned char *)job.labels)[q] > 0 ? 1 : 0;
        }
        if (images) {
#pragma unroll
            for (int k = 0; k < 3; ++k) images[o * 3 + k] = from_f32<T>(c[k]);
        }
        if (mask) mask[o] = in ? 0 : 1;
        if (depth) depth[o] = d;
        if (seg) seg[o] = s;
    }
}

}  // namespace

extern "C" int gwd_collate(const gwd_image_job *jobs, int32_t n, int32_t H, int32_t W, const float *mean, const float *std,
                           void *images, uint8_t *mask, float *depth, int64_t *seg, int32_t dtype, void *stream) {
    if (!jobs || n <= 0 || n > GWD_COLLATE_BATCH || H <= 0 || W <= 0 || !mean || !std) return -1;
    if ((int64_t)H * W >= (1LL << 31)) return -7;
    CollateArgs a;
    for (int i = 0; i < n; ++i) {
        if (jobs[i].h <= 0 || jobs[i].w <= 0 || jobs[i].h > H || jobs[i].w > W) return -3;
        if ((images && !jobs[i].rgb) || (depth && !jobs[i].depth_mm) || (seg && !jobs[i].labels)) return -1;
        a.j[i] = jobs[i];
    }
    for (int k = 0; k < 3; ++k) {
        if (!(std[k] != 0.f)) return -4;
        a.mean[k] = mean[k];
        a.std[k] = std[k];
    }
    a.n = n;
    a.H = H;
    a.W = W;
    int64_t nb = ((int64_t)H * W + 256 * 4 - 1) / (256 * 4);
    if (nb > 1024) nb = 1024;
    const dim3 grid((unsigned)nb, (unsigned)n);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GWD_F32) collate_kernel<float><<<grid, 256, 0, s>>>(a, (float *)images, mask, depth, seg);
    else if (dtype == GWD_BF16) collate_kernel<__bf16><<<grid, 256, 0, s>>>(a, (__bf16 *)images, mask, depth, seg);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}
