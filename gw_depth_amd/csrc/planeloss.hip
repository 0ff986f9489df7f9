// PlaneLoss on the device (src/models/glassrgbd.py:385-450): Sobel normals of the predicted depth
// (src/models/losses/sobel.py:5-27), point-in-triangle masks of the chosen line triplets (the reference rasterises them on
// the HOST with matplotlib.path.Path.contains_points, one D2H copy + one H2D copy per plane) and the per-plane variances of
// the two normal components - forward and backward, no host round trip, no atomics, no memset.
//
// Inside test = matplotlib's point_in_path_impl (crossing test of Graphics Gems IV over the implicitly closed path): for
// every edge a->b, if (a.y >= ty) != (b.y >= ty) and ((b.y-ty)*(a.x-b.x) >= (b.x-tx)*(a.y-b.y)) == (b.y >= ty): flip.
// Vertices and pixel coordinates are integers, so the test is exact in int32 and the masks are bit-identical.
#include "common.h"

namespace {

constexpr int NS = 5;                      // n, sum nx, sum nx^2, sum ny, sum ny^2

struct Tri { int x[3], y[3]; };

__device__ __forceinline__ Tri load_tri(const int64_t *t) {
    Tri r;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.x[k] = (int)t[2 * k];
        r.y[k] = (int)t[2 * k + 1];
    }
    return r;
}

__device__ __forceinline__ bool inside_tri(const Tri &t, int tx, int ty) {
    bool in = false;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int a = e, b = e == 2 ? 0 : e + 1;
        const bool f0 = t.y[a] >= ty, f1 = t.y[b] >= ty;
        const bool hit = ((t.y[b] - ty) * (t.x[a] - t.x[b]) >= (t.x[b] - tx) * (t.y[a] - t.y[b])) == f1;
        in ^= (f0 != f1) & hit;
    }
    return in;
}

// normals (-dx, -dy) of the 3x3 Sobel cross-correlation with zero padding
template <typename T>
__device__ __forceinline__ void sobel_normal(const T *__restrict__ d, int H, int W, int y, int x, float &nx, float &ny) {
    auto at = [&](int yy, int xx) -> float { return ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? to_f32(d[(size_t)yy * W + xx]) : 0.f; };
    const float a = at(y - 1, x - 1), b = at(y - 1, x), c = at(y - 1, x + 1);
    const float e = at(y, x - 1), f = at(y, x + 1);
    const float g = at(y + 1, x - 1), h = at(y + 1, x), i = at(y + 1, x + 1);
    nx = -((a - c) + 2.f * (e - f) + (g - i));
    ny = -((a + 2.f * b + c) - (g + 2.f * h + i));
}

template <typename T>
__global__ __launch_bounds__(256) void plane_partial_kernel(const T *__restrict__ depth, const unsigned char *__restrict__ valid,
                                                            const int64_t *__restrict__ tri, const int *__restrict__ n_planes,
                                                            int H, int W, double *__restrict__ part) {
    const int j = blockIdx.y, nb = gridDim.x;
    double s[NS] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (j < *n_planes) {
        const Tri t = load_tri(tri + 6 * j);
        const int HW = H * W;
        for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += nb * 256) {
            if (!valid[p]) continue;
            const int y = p / W, x = p - y * W;
            if (!inside_tri(t, x, y)) continue;
            float nx, ny;
            sobel_normal(depth, H, W, y, x, nx, ny);
            s[0] += 1.0;
            s[1] += (double)nx;
            s[2] += (double)nx * nx;
            s[3] += (double)ny;
            s[4] += (double)ny * ny;
        }
    }
    __shared__ double sh[4][NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const double v = wave_sum_d(s[k]);
        if (lane == 0) sh[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NS) part[((size_t)j * nb + blockIdx.x) * NS + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// stats[j] = {n, mean_x, mean_y, active}; stats[4*P] = number of active planes; loss = sum(var_x + var_y) / max(1, active)
__global__ __launch_bounds__(64) void plane_finalize_kernel(const double *__restrict__ part, int nb, int P, const int *__restrict__ n_planes,
                                                            int min_area, double *__restrict__ stats, float *__restrict__ loss) {
    __shared__ double var[64];
    __shared__ int act[64];
    const int j = threadIdx.x;
    var[j] = 0.0;
    act[j] = 0;
    if (j < P) {
        double s[NS] = {0.0, 0.0, 0.0, 0.0, 0.0};
        if (j < *n_planes)
            for (int b = 0; b < nb; ++b)
                for (int k = 0; k < NS; ++k) s[k] += part[((size_t)j * nb + b) * NS + k];
        const bool on = j < *n_planes && s[0] >= (double)min_area;              // :437-440
        const double n = on ? s[0] : 1.0, mx = s[1] / n, my = s[3] / n;
        stats[4 * j + 0] = s[0];
        stats[4 * j + 1] = mx;
        stats[4 * j + 2] = my;
        stats[4 * j + 3] = on ? 1.0 : 0.0;
        var[j] = on ? (s[2] / n - mx * mx) + (s[4] / n - my * my) : 0.0;        // torch.var(unbiased=False) of both components
        act[j] = on;
    }
    __syncthreads();
    if (j == 0) {
        double tot = 0.0;
        int cnt = 0;
        for (int k = 0; k < P; ++k) {                                            // plane order, as the reference's loop
            tot += var[k];
            cnt += act[k];
        }
        stats[4 * P] = (double)cnt;
        loss[0] = (float)(tot / (double)(cnt > 0 ? cnt : 1));
    }
}

// d loss / d depth: the per-pixel gradients of (nx, ny) are recomputed on the fly for the 3x3 neighbourhood of each depth
// pixel (every pixel is inside at most a few of the <= 28 planes) and pushed through the transposed Sobel stencil.
template <typename T>
__global__ __launch_bounds__(256) void plane_bwd_kernel(const T *__restrict__ depth, const unsigned char *__restrict__ valid,
                                                        const int64_t *__restrict__ tri, const int *__restrict__ n_planes, int P,
                                                        int H, int W, const double *__restrict__ stats, const float *__restrict__ gloss,
                                                        T *__restrict__ gdepth) {
    __shared__ Tri tris[64];
    __shared__ float inv_n[64], mean_x[64], mean_y[64];
    __shared__ int n_act;
    const int np = min(*n_planes, P);
    if (threadIdx.x < 64) {
        const int j = threadIdx.x;
        const bool on = j < np && stats[4 * j + 3] != 0.0;
        if (j < np) tris[j] = load_tri(tri + 6 * j);
        inv_n[j] = on ? (float)(1.0 / stats[4 * j]) : 0.f;
        mean_x[j] = on ? (float)stats[4 * j + 1] : 0.f;
        mean_y[j] = on ? (float)stats[4 * j + 2] : 0.f;
    }
    if (threadIdx.x == 0) n_act = (int)stats[4 * P];
    __syncthreads();
    const float scale = 2.0f * gloss[0] / (float)(n_act > 0 ? n_act : 1);
    const int HW = H * W;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < HW; q += gridDim.x * 256) {
        const int qy = q / W, qx = q - qy * W;
        float acc = 0.f;
        if (n_act > 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int jx = 0; jx < 3; ++jx) {
                    if (i == 1 && jx == 1) continue;                               // both stencils are zero at the centre
                    const int py = qy - (i - 1), px = qx - (jx - 1);               // output pixel that read q through tap (i, jx)
                    if ((unsigned)py >= (unsigned)H || (unsigned)px >= (unsigned)W || !valid[py * W + px]) continue;
                    float gx = 0.f, gy = 0.f, nx = 0.f, ny = 0.f;
                    bool have = false;
                    for (int j = 0; j < np; ++j) {
                        if (inv_n[j] == 0.f || !inside_tri(tris[j], px, py)) continue;
                        if (!have) {
                            sobel_normal(depth, H, W, py, px, nx, ny);
                            have = true;
                        }
                        gx += (nx - mean_x[j]) * inv_n[j];
                        gy += (ny - mean_y[j]) * inv_n[j];
                    }
                    if (!have) continue;
                    const float kx = (jx == 0 ? 1.f : (jx == 2 ? -1.f : 0.f)) * (i == 1 ? 2.f : 1.f);   // [[1,0,-1],[2,0,-2],[1,0,-1]]
                    const float ky = (i == 0 ? 1.f : (i == 2 ? -1.f : 0.f)) * (jx == 1 ? 2.f : 1.f);   // [[1,2,1],[0,0,0],[-1,-2,-1]]
                    acc -= kx * gx + ky * gy;                                     // normals are the NEGATED Sobel responses
                }
        }
        gdepth[q] = from_f32<T>(acc * scale);
    }
}

int plane_blocks(int64_t HW) {
    int64_t nb = (HW + 256 * 4 - 1) / (256 * 4);
    return (int)(nb < 1 ? 1 : (nb > 128 ? 128 : nb));
}

}  // namespace

// bytes of gwd_query_workspace(GWD_WS_PLANE, {P, H*W}) (optim.hip)
int64_t gwd_plane_workspace_bytes(int64_t P, int64_t HW) {
    if (P <= 0 || P > 64 || HW <= 0) return -1;
    return P * plane_blocks(HW) * NS * (int64_t)sizeof(double);
}

extern "C" int gwd_plane_loss_forward(const void *depth, const uint8_t *valid, const int64_t *tri, const int32_t *n_planes, int32_t P,
                                      int32_t H, int32_t W, int32_t min_area, void *workspace, double *stats, float *loss,
                                      int32_t dtype, void *stream) {
    if (!depth || !valid || !tri || !n_planes || !workspace || !stats || !loss || P <= 0 || P > 64 || H <= 0 || W <= 0) return -1;
    if ((int64_t)H * W >= (1LL << 31) || H > 32768 || W > 32768) return -7;      // the inside test multiplies coordinate differences in int32
    hipStream_t s = (hipStream_t)stream;
    const int nb = plane_blocks((int64_t)H * W);
    const dim3 grid(nb, P);
    if (dtype == GWD_F32) plane_partial_kernel<float><<<grid, 256, 0, s>>>((const float *)depth, valid, tri, n_planes, H, W, (double *)workspace);
    else if (dtype == GWD_BF16) plane_partial_kernel<__bf16><<<grid, 256, 0, s>>>((const __bf16 *)depth, valid, tri, n_planes, H, W, (double *)workspace);
    else return -2;
    GWD_CHECK_LAUNCH();
    plane_finalize_kernel<<<1, 64, 0, s>>>((const double *)workspace, nb, P, n_planes, min_area, stats, loss);
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_plane_loss_backward(const void *depth, const uint8_t *valid, const int64_t *tri, const int32_t *n_planes, int32_t P,
                                       int32_t H, int32_t W, const double *stats, const float *gloss, void *gdepth, int32_t dtype,
                                       void *stream) {
    if (!depth || !valid || !tri || !n_planes || !stats || !gloss || !gdepth || P <= 0 || P > 64 || H <= 0 || W <= 0) return -1;
    if ((int64_t)H * W >= (1LL << 31) || H > 32768 || W > 32768) return -7;
    hipStream_t s = (hipStream_t)stream;
    int64_t nb = ((int64_t)H * W + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (dtype == GWD_F32) plane_bwd_kernel<float><<<(int)nb, 256, 0, s>>>((const float *)depth, valid, tri, n_planes, P, H, W, stats, gloss, (float *)gdepth);
    else if (dtype == GWD_BF16) plane_bwd_kernel<__bf16><<<(int)nb, 256, 0, s>>>((const __bf16 *)depth, valid, tri, n_planes, P, H, W, stats, gloss, (__bf16 *)gdepth);
    else return -2;
    GWD_CHECK_LAUNCH();
    return 0;
}
