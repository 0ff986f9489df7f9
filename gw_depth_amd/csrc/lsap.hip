// Linear sum assignment on the device for the line matcher (src/models/matcher.py:71-74 calls
// scipy.optimize.linear_sum_assignment on the host, 6x per step, each a device->host sync).
// One workgroup per (decoder layer, image) problem: T targets x Q queries with T <= Q, solved by the same
// shortest-augmenting-path algorithm scipy uses (Crouse 2016, rectangular case solved on the transposed problem:
// every target row gets a distinct query column), in double precision like scipy.  The candidate scan over the Q
// columns is spread over the 64 lanes; the optimum is unique unless costs tie exactly.
#include "common.h"

namespace {

constexpr int MAXQ = 1024, MAXT = 64;

__global__ __launch_bounds__(64) void lsap_kernel(const float *__restrict__ cost, const int *__restrict__ col_off,
                                                  int32_t *__restrict__ query_of_target, int B, int Q, int sumT) {
    __shared__ double v[MAXQ], sp[MAXQ], u[MAXT];
    __shared__ int path[MAXQ], row4col[MAXQ], col4row[MAXT];
    __shared__ unsigned char SC[MAXQ], SR[MAXT];
    const int layer = blockIdx.x / B, b = blockIdx.x % B, lane = threadIdx.x;
    const int c0 = col_off[b];
    int T = col_off[b + 1] - c0;
    if (T > MAXT) T = MAXT;              // the counts are device data: never index past the LDS tables (the host checks them too)
    if (T < 0) T = 0;
    const float *C = cost + ((size_t)layer * B + b) * Q * sumT + c0;      // element (q, t) at C[q * sumT + t]
    for (int j = lane; j < Q; j += 64) {
        v[j] = 0.0;
        row4col[j] = -1;
    }
    for (int i = lane; i < T; i += 64) {
        u[i] = 0.0;
        col4row[i] = -1;
    }
    __syncthreads();
    for (int cur = 0; cur < T; ++cur) {
        for (int j = lane; j < Q; j += 64) {
            sp[j] = INFINITY;
            SC[j] = 0;
            path[j] = -1;
        }
        for (int i = lane; i < T; i += 64) SR[i] = 0;
        __syncthreads();
        double minVal = 0.0;
        int i = cur, sink = -1;
        while (sink < 0) {
            if (lane == 0) SR[i] = 1;
            double best = INFINITY;
            int bj = 0x7fffffff;
            for (int j = lane; j < Q; j += 64) {
                if (SC[j]) continue;
                const double r = minVal + (double)C[(size_t)j * sumT + i] - u[i] - v[j];
                if (r < sp[j]) {
                    sp[j] = r;
                    path[j] = i;
                }
                const double s = sp[j];
                // lowest cost; among equals prefer an unassigned column (a new sink), then the lowest index
                if (s < best || (s == best && ((row4col[j] < 0) > (row4col[bj < Q ? bj : 0] < 0) || ((row4col[j] < 0) == (row4col[bj < Q ? bj : 0] < 0) && j < bj)))) {
                    best = s;
                    bj = j;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o, 64);
                const int oj = __shfl_xor(bj, o, 64);
                const bool of = oj < Q && row4col[oj] < 0, mf = bj < Q && row4col[bj] < 0;
                if (ob < best || (ob == best && (of > mf || (of == mf && oj < bj)))) {
                    best = ob;
                    bj = oj;
                }
            }
            minVal = best;
            const int j = bj;
            __syncthreads();
            if (lane == 0) SC[j] = 1;
            if (row4col[j] < 0) sink = j; else i = row4col[j];
            __syncthreads();
        }
        // dual update
        if (lane == 0) u[cur] += minVal;
        for (int r = lane; r < T; r += 64)
            if (SR[r] && r != cur) u[r] += minVal - sp[col4row[r]];
        for (int j = lane; j < Q; j += 64)
            if (SC[j]) v[j] -= minVal - sp[j];
        __syncthreads();
        if (lane == 0) {                         // augment along the path
            int j = sink;
            while (true) {
                const int r = path[j];
                row4col[j] = r;
                const int prev = col4row[r];
                col4row[r] = j;
                j = prev;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
    for (int t = lane; t < T; t += 64) query_of_target[(size_t)layer * sumT + c0 + t] = col4row[t];
    if (b == B - 1)                      // padding columns of a fixed-capacity target table: the dummy query slot Q
        for (int t = col_off[B] + lane; t < sumT; t += 64) query_of_target[(size_t)layer * sumT + t] = Q;
}

}  // namespace

extern "C" int gwd_lsap(const float *cost, const int32_t *col_offsets, int32_t *query_of_target, int32_t layers, int32_t B,
                        int32_t Q, int32_t sum_targets, int32_t max_targets, void *stream) {
    if (!cost || !col_offsets || !query_of_target || layers <= 0 || B <= 0 || Q <= 0 || sum_targets <= 0) return -1;
    if (Q > MAXQ || max_targets > MAXT || max_targets > Q) return -4;
    lsap_kernel<<<layers * B, 64, 0, (hipStream_t)stream>>>(cost, col_offsets, query_of_target, B, Q, sum_targets);
    GWD_CHECK_LAUNCH();
    return 0;
}
