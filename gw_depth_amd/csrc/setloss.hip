// The DETR-style set criterion of the line branch on the device, three small launches around the device LSAP (csrc/lsap.hip):
//   gwd_match_cost           cost[l][b][q][t] = w_line * |pred_lines - tgt_lines|_1 - w_class * softmax(logits)[label[t]]
//                            (HungarianMatcher_Line.forward, /root/reference/src/models/matcher.py:52-70)
//   gwd_set_losses_forward   per decoder layer: weighted cross entropy over all queries (matched queries carry their target's label,
//                            the rest "no object", weight eos_coef) and the L1 distance of the matched line pairs / num_items
//                            (SetCriterion.loss_lines_labels / loss_lines_POST, /root/reference/src/models/glassrgbd.py:160-170,231-244)
//   gwd_set_losses_backward  their gradients w.r.t. logits and lines.
// Targets arrive PADDED to a fixed capacity (criteria.PackedTargets): column t belongs to image bidx[t], valid[t] = 0 marks padding,
// the LSAP hands padding columns the dummy query Q.  As torch ops this was ~40 launches forward and ~25 backward on a few KB of data.
#include "common.h"

namespace {

constexpr int KMAX = 8, DMAX = 8;

__device__ __forceinline__ void softmax_k(const float *lg, int K, float *p) {
    float mx = lg[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, lg[k]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) {
        p[k] = expf(lg[k] - mx);
        s += p[k];
    }
    for (int k = 0; k < K; ++k) p[k] = p[k] / s;
}

__global__ void match_cost_kernel(const float *__restrict__ logits, const float *__restrict__ lines, const float *__restrict__ tl,
                                  const int64_t *__restrict__ labels, float *__restrict__ cost, int64_t LBQ, int cap, int K, int D,
                                  float w_line, float w_class) {
    const int64_t total = LBQ * cap;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % cap);
        const int64_t r = i / cap;
        float p[KMAX];
        softmax_k(logits + r * K, K, p);
        float l1 = 0.f;
        for (int d = 0; d < D; ++d) l1 += fabsf(lines[r * D + d] - tl[(size_t)t * D + d]);
        int lab = (int)labels[t];
        lab = lab < 0 ? 0 : (lab >= K ? K - 1 : lab);
        cost[i] = w_line * l1 + w_class * (-p[lab]);
    }
}

// one workgroup per decoder layer
__global__ __launch_bounds__(256) void set_losses_fwd_kernel(const float *__restrict__ logits, const float *__restrict__ lines,
                                                             const float *__restrict__ tl, const int64_t *__restrict__ labels,
                                                             const int32_t *__restrict__ bidx, const int32_t *__restrict__ valid,
                                                             const int32_t *__restrict__ qot, const float *__restrict__ cls_w,
                                                             const float *__restrict__ num_items, float world, int32_t *__restrict__ tc,
                                                             float *__restrict__ ce, float *__restrict__ l1, float *__restrict__ wsum,
                                                             int B, int Q, int cap, int K, int D) {
    __shared__ double red[3][256];
    const int l = blockIdx.x, tid = threadIdx.x, BQ = B * Q;
    int32_t *tcl = tc + (size_t)l * BQ;
    for (int i = tid; i < BQ; i += 256) tcl[i] = K - 1;                         // "no object"
    __syncthreads();
    double s_l1 = 0.0;
    for (int t = tid; t < cap; t += 256) {
        const int q = qot[(size_t)l * cap + t], b = bidx[t];
        if (q < Q && valid[t]) tcl[b * Q + q] = (int32_t)labels[t];             // padding columns sit on the dummy query Q
        if (valid[t]) {
            const int qc = q < Q ? q : Q - 1;
            const float *pl = lines + ((size_t)l * BQ + (size_t)b * Q + qc) * D;
            float a = 0.f;
            for (int d = 0; d < D; ++d) a += fabsf(pl[d] - tl[(size_t)t * D + d]);
            s_l1 += (double)a;
        }
    }
    __syncthreads();
    double s_n = 0.0, s_w = 0.0;
    for (int i = tid; i < BQ; i += 256) {
        const float *lg = logits + ((size_t)l * BQ + i) * K;
        float mx = lg[0];
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, lg[k]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += expf(lg[k] - mx);
        const int c = tcl[i];
        const float nll = -((lg[c] - mx) - logf(s)), w = cls_w[c];
        s_n += (double)(nll * w);
        s_w += (double)w;
    }
    red[0][tid] = s_n;
    red[1][tid] = s_w;
    red[2][tid] = s_l1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            red[0][tid] += red[0][tid + o];
            red[1][tid] += red[1][tid + o];
            red[2][tid] += red[2][tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float n = fmaxf(num_items[0] / world, 1.0f);
        ce[l] = (float)(red[0][0] / red[1][0]);
        wsum[l] = (float)red[1][0];
        l1[l] = (float)(red[2][0] / (double)n);
    }
}

// dlogits fully written; dlines must be zero on entry (matched rows are added)
__global__ __launch_bounds__(256) void set_losses_bwd_kernel(const float *__restrict__ logits, const float *__restrict__ lines,
                                                             const float *__restrict__ tl, const int32_t *__restrict__ bidx,
                                                             const int32_t *__restrict__ valid, const int32_t *__restrict__ qot,
                                                             const float *__restrict__ cls_w, const float *__restrict__ num_items,
                                                             float world, const int32_t *__restrict__ tc, const float *__restrict__ wsum,
                                                             const float *__restrict__ g_ce, const float *__restrict__ g_l1,
                                                             float *__restrict__ dlogits, float *__restrict__ dlines, int B, int Q, int cap,
                                                             int K, int D) {
    const int l = blockIdx.x, tid = threadIdx.x, BQ = B * Q;
    const float gce = g_ce ? g_ce[l] : 0.f, gl1 = g_l1 ? g_l1[l] : 0.f;
    const float ws = wsum[l];
    for (int i = tid; i < BQ; i += 256) {
        const float *lg = logits + ((size_t)l * BQ + i) * K;
        float p[KMAX];
        softmax_k(lg, K, p);
        const int c = tc[(size_t)l * BQ + i];
        const float f = gce * cls_w[c] / ws;
        for (int k = 0; k < K; ++k) dlogits[((size_t)l * BQ + i) * K + k] = f * (p[k] - (k == c ? 1.f : 0.f));
    }
    const float n = fmaxf(num_items[0] / world, 1.0f);
    for (int t = tid; t < cap; t += 256) {
        if (!valid[t]) continue;
        const int q = qot[(size_t)l * cap + t], b = bidx[t];
        const int qc = q < Q ? q : Q - 1;
        const size_t row = ((size_t)l * BQ + (size_t)b * Q + qc) * D;
        for (int d = 0; d < D; ++d) {
            const float df = lines[row + d] - tl[(size_t)t * D + d];
            const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            atomicAdd(dlines + row + d, gl1 * sg / n);
        }
    }
}

}  // namespace

extern "C" int gwd_match_cost(const float *logits, const float *lines, const float *tgt_lines, const int64_t *tgt_labels, float *cost,
                              int32_t L, int32_t B, int32_t Q, int32_t cap, int32_t K, int32_t D, float w_line, float w_class,
                              void *stream) {
    if (!logits || !lines || !tgt_lines || !tgt_labels || !cost || L <= 0 || B <= 0 || Q <= 0 || cap <= 0) return -1;
    if (K <= 0 || K > KMAX || D <= 0 || D > DMAX) return -4;
    const int64_t lbq = (int64_t)L * B * Q, total = lbq * cap;
    int64_t nb = (total + 255) / 256;
    match_cost_kernel<<<(int)(nb > 2048 ? 2048 : nb), 256, 0, (hipStream_t)stream>>>(logits, lines, tgt_lines, tgt_labels, cost, lbq, cap, K, D,
                                                                                    w_line, w_class);
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_set_losses_forward(const float *logits, const float *lines, const float *tgt_lines, const int64_t *tgt_labels,
                                      const int32_t *bidx, const int32_t *valid, const int32_t *qot, const float *class_weight,
                                      const float *num_items, float world, int32_t *target_class, float *ce, float *l1, float *wsum,
                                      int32_t L, int32_t B, int32_t Q, int32_t cap, int32_t K, int32_t D, void *stream) {
    if (!logits || !lines || !tgt_lines || !tgt_labels || !bidx || !valid || !qot || !class_weight || !num_items || !target_class || !ce ||
        !l1 || !wsum || L <= 0 || B <= 0 || Q <= 0 || cap <= 0)
        return -1;
    if (K <= 0 || K > KMAX || D <= 0 || D > DMAX) return -4;
    set_losses_fwd_kernel<<<L, 256, 0, (hipStream_t)stream>>>(logits, lines, tgt_lines, tgt_labels, bidx, valid, qot, class_weight, num_items,
                                                             world, target_class, ce, l1, wsum, B, Q, cap, K, D);
    GWD_CHECK_LAUNCH();
    return 0;
}

extern "C" int gwd_set_losses_backward(const float *logits, const float *lines, const float *tgt_lines, const int32_t *bidx,
                                       const int32_t *valid, const int32_t *qot, const float *class_weight, const float *num_items,
                                       float world, const int32_t *target_class, const float *wsum, const float *g_ce, const float *g_l1,
                                       float *dlogits, float *dlines, int32_t L, int32_t B, int32_t Q, int32_t cap, int32_t K, int32_t D,
                                       void *stream) {
    if (!logits || !lines || !tgt_lines || !bidx || !valid || !qot || !class_weight || !num_items || !target_class || !wsum || !dlogits ||
        !dlines || L <= 0 || B <= 0 || Q <= 0 || cap <= 0)
        return -1;
    if (K <= 0 || K > KMAX || D <= 0 || D > DMAX) return -4;
    set_losses_bwd_kernel<<<L, 256, 0, (hipStream_t)stream>>>(logits, lines, tgt_lines, bidx, valid, qot, class_weight, num_items, world,
                                                             target_class, wsum, g_ce, g_l1, dlogits, dlines, B, Q, cap, K, D);
    GWD_CHECK_LAUNCH();
    return 0;
}
