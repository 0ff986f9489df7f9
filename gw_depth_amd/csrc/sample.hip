// CertainSample on the device (src/models/points/points_sample.py:291-364) — no host round trip.
// One workgroup per image: variance map = (bilinear_align_corners(pred_small) - pred_large)^2 in LDS, per-interval
// pixel counts n_i, k_i = min(floor(fl32(n_i / HW) * S), n_i) in IEEE fp32 exactly as the CPU reference evaluates
// it, the top-max(k_i) pixels of the WHOLE variance map by repeated block-wide arg-max (ties -> lowest index, i.e.
// a stable descending order), then the reference's group / repeat / trim rules and the (x/W, y/H)*2-1 coordinates.
#include "common.h"

namespace {

constexpr int MAX_PIX = 36864, MAX_S = 256, MAX_INT = 8, THREADS = 256;
constexpr int TABLE_BYTES = (6 * MAX_S + 2 * MAX_INT + 2 * (THREADS / 64) + 4) * 4;   // variance map in dynamic LDS: up to 144 KiB (192 x 192)

__global__ __launch_bounds__(THREADS) void certain_sample_kernel(const float *__restrict__ small, const float *__restrict__ large,
                                                                  float *__restrict__ coords, int hs, int ws, int H, int W,
                                                                  const float *__restrict__ edges, int n_int, int S) {
    // one dynamic LDS block, carved by hand: [small tables (fixed size)] [variance map: H * W floats]
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    int *order = (int *)dyn_lds;            // [MAX_S] top pixels, descending variance
    int *outidx = order + MAX_S;            // [4 * MAX_S]
    int *cnt = outidx + 4 * MAX_S;          // [MAX_INT]
    float *red_v = (float *)(cnt + MAX_INT);  // [THREADS / 64]
    int *red_i = (int *)(red_v + THREADS / 64);
    int *kk = red_i + THREADS / 64;         // [MAX_INT]
    int *sel = kk + MAX_INT;                // [MAX_S] selected pixels, unordered
    int *ctrs = sel + MAX_S;                // [3] rotating counters of block_count, [3] = list length
    float *var = (float *)(ctrs + 4);       // [H * W]
    const int b = blockIdx.x, tid = threadIdx.x, HW = H * W;
    const float *sm = small + (size_t)b * hs * ws;
    const float *lg = large + (size_t)b * HW;
    if (tid < MAX_INT) cnt[tid] = 0;
    if (tid < 4) ctrs[tid] = 0;
    __syncthreads();
    const float sh = H > 1 ? (float)(hs - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(ws - 1) / (float)(W - 1) : 0.f;
    int local[MAX_INT];
#pragma unroll
    for (int i = 0; i < MAX_INT; ++i) local[i] = 0;
    for (int p = tid; p < HW; p += THREADS) {
        const int y = p / W, x = p - y * W;
        const float fy = sh * y, fx = sw * x;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < hs - 1), x1 = x0 + (x0 < ws - 1);
        const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
        const float up = hy * (hx * sm[y0 * ws + x0] + lx * sm[y0 * ws + x1]) + ly * (hx * sm[y1 * ws + x0] + lx * sm[y1 * ws + x1]);
        const float l = lg[p];
        const float dlt = up - l;
        var[p] = dlt * dlt;
#pragma unroll
        for (int i = 0; i < MAX_INT; ++i)
            if (i < n_int && l >= edges[i] && l < edges[i + 1]) local[i]++;
    }
#pragma unroll
    for (int i = 0; i < MAX_INT; ++i)
        if (i < n_int && local[i]) atomicAdd(&cnt[i], local[i]);
    __syncthreads();
    if (tid == 0) {
        for (int i = 0; i < n_int; ++i) {
            const float n = (float)cnt[i];
            const float k = floorf(__fmul_rn(__fdiv_rn(n, (float)HW), (float)S));
            kk[i] = (int)fminf(k, n);
        }
    }
    __syncthreads();
    int kmax = 0, total = 0;
    for (int i = 0; i < n_int; ++i) {
        kmax = max(kmax, kk[i]);
        total += kk[i];
    }
    if (total == 0) kmax = S;                     // "sample globally when no interval points found" (:331-339)
    // ---- the kmax largest variances in descending order, ties to the lowest pixel index - by SELECTION, not by kmax
    // rounds of arg-max (each round was a dependent sweep of a thread's ~19 LDS values + two barriers: 0.4 ms for S = 80):
    // (1) bitwise binary search for the bit pattern T of the kmax-th largest value (variances are >= 0, so their bit
    // patterns order like the values; 31 counting rounds, one barrier each), (2) among the pixels equal to T, the index
    // threshold I that admits exactly the missing number of them (16 rounds), (3) the kmax selected pixels are appended
    // to a list and rank-sorted by (value descending, index ascending).  Same set, same order as the arg-max loop.
    auto block_count = [&](auto pred, int round) -> int {                  // number of pixels satisfying pred; 1 barrier
        int c = 0;
        for (int p = tid; p < HW; p += THREADS) c += pred(__float_as_uint(var[p]), p) ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        int *ctr = ctrs + (round % 3);
        if ((tid & 63) == 0) atomicAdd(ctr, c);
        if (tid == 0) ctrs[(round + 1) % 3] = 0;                           // last read two rounds ago
        __syncthreads();
        return *ctr;
    };
    int round = 0;
    unsigned T = 0;
    for (int bit = 30; bit >= 0; --bit) {
        const unsigned cand = T | (1u << bit);
        if (block_count([&](unsigned v, int) { return v >= cand; }, round++) >= kmax) T = cand;
    }
    const int n_gt = block_count([&](unsigned v, int) { return v > T; }, round++);
    const int need_eq = kmax - n_gt;                                        // >= 1
    int I = 0;
    for (int bit = 15; bit >= 0; --bit) {
        const int cand = I | (1 << bit);
        if (block_count([&](unsigned v, int p) { return v == T && p < cand; }, round++) < need_eq) I = cand;
    }
    for (int p = tid; p < HW; p += THREADS) {
        const unsigned v = __float_as_uint(var[p]);
        if (v > T || (v == T && p <= I)) sel[atomicAdd(&ctrs[3], 1)] = p;  // exactly kmax pixels, any order
    }
    __syncthreads();
    for (int a = tid; a < kmax; a += THREADS) {
        const int pa = sel[a];
        const unsigned va = __float_as_uint(var[pa]);
        int rank = 0;
        for (int c = 0; c < kmax; ++c) {
            const int pc = sel[c];
            const unsigned vc = __float_as_uint(var[pc]);
            rank += (vc > va || (vc == va && pc < pa)) ? 1 : 0;
        }
        order[rank] = pa;
    }
    __syncthreads();
    // ---- group assembly.  The ascending-index copies of each interval's best k_i pixels are rank sorts done by all
    // threads (thread 0 alone spent 130 us in these O(k^2) loops); the list surgery that follows is O(S) and serial.
    int n_out = 0;
    int gstart[MAX_INT], gcount[MAX_INT], ng = 0;            // uniform: every thread derives the same bookkeeping
    auto emit_sorted_prefix = [&](int k, int at) {           // ascending pixel index of the k best (:320,335)
        for (int a = tid; a < k; a += THREADS) {
            const int oa = order[a];
            int rank = 0;
            for (int c = 0; c < k; ++c) rank += order[c] < oa;
            outidx[at + rank] = oa;
        }
    };
    if (total > 0) {
        for (int i = 0; i < n_int; ++i)
            if (kk[i] > 0) {
                gstart[ng] = n_out;
                gcount[ng] = kk[i];
                emit_sorted_prefix(kk[i], n_out);
                n_out += kk[i];
                ++ng;
            }
    } else {
        emit_sorted_prefix(S, 0);
        n_out = S;
    }
    __syncthreads();
    if (tid == 0) {
        int remain = total > 0 ? S - total : 0;
        const int already = total;
        if (remain > 0 && remain >= already) {              // :343-346 repeat the whole list
            const int times = remain / already + 1;
            for (int t = 1; t < times; ++t)
                for (int a = 0; a < already; ++a) outidx[t * already + a] = outidx[a];
            n_out = already * times;
            remain = S - n_out;
        }
        if (remain > 0) {                                    // :348-350 complement with the tail
            for (int a = 0; a < remain; ++a) outidx[n_out + a] = outidx[n_out - remain + a];
            n_out += remain;
        }
        if (remain < 0) {                                    // :351-355 trim the largest group (first on ties)
            int mid = 0;
            for (int gi = 1; gi < ng; ++gi)
                if (gcount[gi] > gcount[mid]) mid = gi;
            const int cut = -remain;                         // drop the last `cut` entries of group mid
            const int from = gstart[mid] + gcount[mid];
            for (int a = from; a < n_out; ++a) outidx[a - cut] = outidx[a];
            n_out -= cut;
        }
    }
    __syncthreads();
    for (int a = tid; a < S; a += THREADS) {
        const int p = outidx[a];
        const int row = p / W, col = p - row * W;
        coords[((size_t)b * S + a) * 2 + 0] = __fdiv_rn((float)col, (float)W) * 2.f - 1.f;
        coords[((size_t)b * S + a) * 2 + 1] = __fdiv_rn((float)row, (float)H) * 2.f - 1.f;
    }
}

}  // namespace

extern "C" int gwd_certain_sample(const float *pred_small, const float *pred_large, float *coords, int32_t B, int32_t hs,
                                  int32_t ws, int32_t H, int32_t W, const float *edges, int32_t n_intervals,
                                  int32_t sample_num, void *stream) {
    if (!pred_small || !pred_large || !coords || !edges || B <= 0 || hs <= 0 || ws <= 0 || H <= 0 || W <= 0) return -1;
    if (H * W > MAX_PIX || sample_num <= 0 || sample_num > MAX_S || n_intervals <= 0 || n_intervals > MAX_INT) return -4;
    if (sample_num > H * W) return -4;
    static bool attr_set = false;
    if (!attr_set) {                                      // dynamic LDS beyond 64 KiB has to be requested once
        (void)hipFuncSetAttribute((const void *)certain_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  TABLE_BYTES + MAX_PIX * (int)sizeof(float));
        attr_set = true;
    }
    certain_sample_kernel<<<B, THREADS, TABLE_BYTES + (size_t)H * W * sizeof(float), (hipStream_t)stream>>>(
        pred_small, pred_large, coords, hs, ws, H, W, edges, n_intervals, sample_num);
    GWD_CHECK_LAUNCH();
    return 0;
}
