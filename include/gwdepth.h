/* gwdepth.h — C ABI of libgwdepth_hip.so: hand-written gfx950 (MI355X) kernels for the GW-Depth
 * train step.
 *
 * The reference (ViktorLiang/GW-Depth) has no FFI/plugin boundary: it is 100 % PyTorch and every
 * device operation is an ATen call.  Each entry point below therefore replaces an
 * ATen operator sequence at the cited reference call sites; the host side (the gw_depth_amd Python package) binds
 * them with ctypes and wraps them in torch.autograd.Function objects.
 *
 * Conventions:
 *   - plain C, no torch types; raw device pointers, explicit sizes, dtype enum, caller's hipStream_t
 *     passed as void*;
 *   - return 0 on success, negative = invalid argument / unsupported shape, positive = hipError_t;
 *   - never allocate or free caller memory, never synchronise the device, no mutable global state;
 *   - activation tensors are NHWC ("pixel-major": [B][H][W][C], a Linear input is [rows][C]);
 *   - convolution / linear weights are [Cout][KH][KW][Cin] (a Linear weight is [out][in] as in torch).
 */
#ifndef GWDEPTH_H
#define GWDEPTH_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GWD_VERSION 10

enum { GWD_F32 = 0, GWD_BF16 = 1 };
enum { GWD_ACT_NONE = 0, GWD_ACT_RELU = 1, GWD_ACT_GELU = 2, GWD_ACT_ELU = 3, GWD_ACT_SIGMOID = 4 };
/* how output pixel (oh,ow) and tap (kh,kw) map to an input pixel */
enum {
    GWD_GATHER_CONV = 0,      /* ih = oh*stride - pad + kh                                     */
    GWD_GATHER_TRANSPOSED = 1,/* ih = (oh + pad - kh)/stride when divisible (data gradient)    */
    GWD_GATHER_UPSAMPLED = 2  /* stride 1 over a nearest-upsampled (Hv,Wv) view of the input   */
};

typedef struct {
    const void *x;        /* [B][Hi][Wi][Cin]                                                   */
    const void *w;        /* [Cout][KH][KW][Cin]                                                */
    void *y;              /* [B][Ho][Wo][Cout]                                                  */
    void *z;              /* optional copy of the value BEFORE the activation (for GELU bwd)    */
    const float *scale;   /* optional per-Cout multiplier (FrozenBN scale)                      */
    const float *shift;   /* optional per-Cout addend (bias / FrozenBN shift)                   */
    const void *residual; /* optional [B][Ho][Wo][Cout] added before the activation             */
    const void *zero_page;/* optional: >= 64 zero bytes of device memory.  Enables the LDS-DMA     */
                          /* pipeline (out-of-image taps are fetched from it instead of branching) */
    const void *mult;     /* optional [B][Ho][Wo][Cout] element-wise multiplier applied AFTER the    */
                          /* activation (dropout keep-mask / (1-p)); with it, `residual` is added    */
                          /* after the multiply instead of before the activation:                    */
                          /*   y = act_scale * act(conv + shift) * mult + residual                   */
                          /* (x + dropout(sublayer(x)) of src/models/transformer.py:149-162,212-233)  */
    const void *gate;     /* optional [B][Ho][Wo][Cout], with gate_act: the LAST step of the epilogue multiplies the   */
                          /* result by act'(.) of an activation whose OUTPUT is `gate` (ReLU: gate > 0, ELU with       */
                          /* alpha 1: gate > 0 ? 1 : gate + 1).  For a data-gradient launch this is the backward of    */
                          /* the activation that produced the layer's input (gate = that input, which the layer keeps  */
                          /* for its weight gradient anyway): the producer then skips its own activation-backward pass */
    int32_t B, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int32_t gather;       /* GWD_GATHER_*                                                       */
    int32_t Hv, Wv;       /* virtual input size for GWD_GATHER_UPSAMPLED                        */
    int32_t act;          /* GWD_ACT_*                                                          */
    float act_scale;      /* y = act_scale * act(v)  (max_depth * sigmoid)                      */
    int32_t dtype;        /* GWD_F32 / GWD_BF16 for x, w, y, z, residual, mult, gate            */
    int32_t gate_act;     /* GWD_ACT_RELU or GWD_ACT_ELU when gate != NULL, else ignored         */
    /* ConvLn mode (src/models/points/points_sample.py:12-25: conv -> LayerNorm over channels [-> GELU]), when ln_mean != NULL:   */
    /*   y = act(LN(conv) * scale + shift) + residual,  LN over the first ln_C of the Cout channels of a row (eps 1e-5; channels   */
    /*   ln_C .. Cout-1 are zero padding and come out as zeros; scale / shift = gamma / beta with ln_C entries, both required;     */
    /*   act NONE or GELU; the residual is added AFTER the activation, as gwd_layernorm_forward adds it).  z (optional) receives   */
    /*   the convolution itself, ln_mean / ln_rstd [B*Ho*Wo] the row statistics - exactly what gwd_layernorm_backward takes.      */
    /*   gwd_conv_forward returns -4 when it has no fused kernel for the shape (bf16, LDS-DMA route, a whole row inside one tile:  */
    /*   Cout <= 64 or Cout == 160); the caller then runs the convolution and gwd_layernorm_forward separately.                  */
    float *ln_mean;
    float *ln_rstd;
    int32_t ln_C;
    int32_t reserved;
} gwd_conv_desc;

int gwd_version(void);
const char *gwd_arch(void);

/* Scratch sizes.  The library never allocates: the two entry points that need scratch take a caller-provided buffer,
 * and this tells the caller how many BYTES it must hold (fully overwritten by the call, no initialisation needed).
 *   GWD_WS_INORM_GELU      dims = {B, S, C}        -> `part` of gwd_inorm_gelu_forward / _backward
 *   GWD_WS_RESAMPLE_BWD    dims = {B, Ho, Ws, C}   -> `tmp`  of gwd_resample_backward_sep
 *   GWD_WS_EVAL            dims = {B, H*W}         -> `workspace` of gwd_eval_accumulate
 *   GWD_WS_PLANE           dims = {P, H*W}         -> `workspace` of gwd_plane_loss_forward
 * Returns -1 for an unknown op or a wrong dimension count.                                                     */
enum { GWD_WS_INORM_GELU = 0, GWD_WS_RESAMPLE_BWD = 1, GWD_WS_EVAL = 2, GWD_WS_PLANE = 3 };
int64_t gwd_query_workspace(int32_t op, const int64_t *dims, int32_t ndims);

/* Implicit-GEMM convolution on MFMA with fused epilogue y = act(scale*conv(x,w) + shift + residual).
 * Also the Linear layer (KH=KW=1, Hi=Wi=1, B=rows) and, with GWD_GATHER_TRANSPOSED and
 * the transposed weights of gwd_weight_prep, the data gradient.
 * Replaces: torchvision Bottleneck conv+FrozenBN+ReLU (src/models/backbone.py:45-55,90-92), ConvLn /
 * ConvA / upconv convolutions (src/models/points/points_sample.py:15-22,
 * src/models/multiscale_transformerr.py:110-116, src/models/dense_upsample.py:82-90,126-146),
 * every nn.Linear on the path (src/models/transformer.py:133-136,
 * src/models/multiscale_transformerr.py:62-64,249-251,425-429,445-449).                          */
int gwd_conv_forward(const gwd_conv_desc *d, void *stream);

/* Weight gradient dw[Cout][KH][KW][Cin] (fp32, ACCUMULATED into dw with atomics; caller zeroes it):
 * d->x = layer input, d->y = gradient w.r.t. the layer output (already multiplied by act'), gather
 * as in the forward.  d->scale (may be NULL): per-Cout multiplier of the result, dw[n] += scale[n] * (...) - the
 * layer computed with w * scale (folded FrozenBatchNorm, src/models/backbone.py:45-55), so this IS the gradient of
 * the unscaled parameter.  Replaces aten::convolution_backward (weight) / addmm weight grads.      */
int gwd_conv_wgrad(const gwd_conv_desc *d, float *dw, void *stream);
/* n independent gwd_conv_wgrad calls handed over together (descs[i] -> dws[i], each accumulated as gwd_conv_wgrad does;
 * two jobs may name the same dw).  Results are those of n single calls; the library is free to run jobs of one
 * kernel shape as ONE launch - the weight gradients of the ~230 Linear / 1x1 layers of a train step are 10-60 us
 * launches of 16-400 workgroups that cannot overlap inside one stream.  Nothing in descs / dws must outlive the call. */
int gwd_conv_wgrad_batch(const gwd_conv_desc *descs, float *const *dws, int32_t n, void *stream);

/* w (fp32 [N][taps][C]), optionally multiplied by row_scale[N] (a frozen BatchNorm folded into the
 * convolution) -> w_fwd (dtype, same layout; may be NULL) and w_dgrad (dtype, [C][taps][N]; may be
 * NULL).                                                                                          */
int gwd_weight_prep(const float *w, const float *row_scale, void *w_fwd, void *w_dgrad, int32_t N,
                    int32_t taps, int32_t C, int32_t dtype, void *stream);

/* gx = gy * act'(.) ; `ref` is the activation OUTPUT for RELU/ELU/SIGMOID (divided by act_scale
 * internally) and the PRE-activation for GELU.  Optional per-channel multiplier `scale` ([C]).   */
int gwd_act_backward(const void *gy, const void *ref, void *gx, const float *scale, int64_t rows,
                     int32_t C, int32_t act, float act_scale, int32_t dtype, void *stream);

/* out[c] += sum_rows g[row][c]  (bias / shift gradients; fp32 atomics, caller zeroes).           */
int gwd_colsum(const void *g, float *out, int64_t rows, int32_t C, int32_t dtype, void *stream);
/* Up to GWD_COLSUM_BATCH gwd_colsum calls (same dtype) as ONE launch: out[c] += column sums of g for every job.  The
 * records are read on the host and travel in the kernel arguments (nothing is uploaded, nothing must outlive the call).
 * block0 / blocks are filled in by the library.  Returns -4 if a job's C is not a multiple of 16 bytes or wider than
 * 256 vectors (run that one through gwd_colsum).                                                               */
#define GWD_COLSUM_BATCH 16
typedef struct {
    const void *g;        /* [rows][C] (dtype)                       */
    float *out;           /* [C] fp32, accumulated                   */
    int64_t rows;
    int32_t C;
    int32_t block0, blocks;   /* library-internal                    */
} gwd_colsum_job;
int gwd_colsum_batch(const gwd_colsum_job *jobs, int32_t n_jobs, int32_t dtype, void *stream);
/* gwd_act_backward and gwd_colsum of its result in one pass (activation backward of a biased layer: dBias is the column
 * sum of gx).  dbias is ACCUMULATED into.  Returns -4 when C is not a multiple of 16 bytes / wider than 256 vectors:
 * the caller then uses the two separate entry points.                                                            */
/* mult (may be NULL): [rows][C] element-wise multiplier of gy applied first (the dropout multiplier of gwd_conv_desc.mult: the
 * layer computed act(v) * mult, so gx = act'(ref) * (gy * mult)).                                                   */
int gwd_act_backward_colsum(const void *gy, const void *ref, void *gx, float *dbias, int64_t rows, int32_t C, int32_t act,
                            float act_scale, const void *mult, int32_t dtype, void *stream);


/* LayerNorm over the last dim (C <= 512), eps 1e-5, optional fused exact GELU on the output.
 * Replaces nn.LayerNorm (+ nn.GELU) call sites: src/models/points/points_sample.py:19-25,
 * src/models/multiscale_transformerr.py:612-632,659-665,755-777, src/models/transformer.py:138-162,
 * src/models/dense_upsample.py:125,138,166,177.                                                  */
/* residual (may be NULL): [rows][C], added AFTER the normalisation / GELU: y = gelu?(LN(x)) + residual - the skip
 * connection of BasicBlock (src/models/points/points_sample.py:41-42); its gradient is gy itself.              */
/* ld: row pitch in elements of x, residual, y (forward) / gy, x, gx (backward); 0 = C.  With ld > C the channels C..ld-1 are
 * zero padding: never read, WRITTEN as zeros in y / gx (statistics, gamma, beta and their gradients cover the C real channels);
 * needs ld to be a multiple of 16 bytes, else -4 (C itself may be anything).                                      */
int gwd_layernorm_forward(const void *x, const float *gamma, const float *beta, const void *residual, void *y, float *mean,
                          float *rstd, int64_t rows, int32_t C, int32_t ld, int32_t gelu, int32_t dtype, void *stream);
/* gskip (may be NULL): [rows][ld], a second gradient of x added to gx (x also feeds a skip connection); returns -4 when the shape
 * has no vector kernel (the caller then adds it).                                                                   */
/* gelu, backward: bit 0 = GELU behind the norm (as forward); bit 1 (GWD_LN_ELU_INPUT) = x is the output of an ELU (alpha 1) whose
 * own backward pass is left to this call: gx = (gx + gskip) * elu'(x) (upconv1 -> norm of src/models/dense_upsample.py:163-166);
 * -4 when the shape has no vector kernel.                                                                             */
#define GWD_LN_ELU_INPUT 2
int gwd_layernorm_backward(const void *gy, const void *x, const float *gamma, const float *beta,
                           const float *mean, const float *rstd, void *gx, float *dgamma, float *dbeta,
                           int64_t rows, int32_t C, int32_t ld, int32_t gelu, const void *gskip, int32_t dtype, void *stream);

/* Softmax over the last dim (row length L <= 1024), forward and backward.
 * Replaces F.softmax in src/models/multi_head_attention.py:366, multiscale_transformerr.py:307,
 * 322-324,550-552,570,576, points_sample.py:277.                                                 */
int gwd_softmax_forward(const void *x, void *y, int64_t rows, int32_t L, int32_t dtype, void *stream);
int gwd_softmax_backward(const void *gy, const void *y, void *gx, int64_t rows, int32_t L, int32_t dtype,
                         void *stream);
/* Attention softmax with the score scaling and the key-padding mask folded in (src/models/multi_head_attention.py:
 * 329-352: q * scaling, masked_fill(-inf), softmax): y = softmax_row(scale * x + mask), key_mask uint8
 * [rows / rows_per_mask][L] (nonzero = key excluded) or NULL.  Backward: gx = scale * y * (gy - <y, gy>).       */
int gwd_softmax_masked_forward(const void *x, const uint8_t *key_mask, void *y, int64_t rows, int32_t L, int64_t rows_per_mask,
                               float scale, int32_t dtype, void *stream);
int gwd_softmax_scaled_backward(const void *gy, const void *y, void *gx, int64_t rows, int32_t L, float scale, int32_t dtype,
                                void *stream);


/* SiLog masked reduction (src/models/glassrgbd.py:366-374 with the per-scale nearest-resized GT and
 * validity mask of src/engine_glassrgbd.py:65,76-78 gathered on the fly).
 * pred [B][h][w] (dtype), gt [B][H][W] fp32 at full resolution.  sums[3] (double; caller zeroes)
 * receive sum d, sum d^2, count.  log_depth_error selects d = log p - log g, else (p+log p)-(g+log g). */
int gwd_silog_sums(const void *pred, const float *gt, double *sums, int32_t B, int32_t h, int32_t w,
                   int32_t H, int32_t W, int32_t log_depth_error, int32_t dtype, void *stream);
/* loss (one fp32) = scale * sqrt(sums[1]/n - lambda (sums[0]/n)^2), n = sums[2]: the scalar tail of SiLog as one launch.  */
int gwd_silog_finalize(const double *sums, float lambda, float scale, float *loss, void *stream);
/* gpred = gloss * d loss / d pred, loss = 10*sqrt(E[d^2] - lambda*E[d]^2); reads the sums.        */
int gwd_silog_backward(const void *pred, const float *gt, const double *sums, const float *gloss,
                       float loss_weight, float lambda, void *gpred, int32_t B, int32_t h, int32_t w,
                       int32_t H, int32_t W, int32_t log_depth_error, int32_t dtype, void *stream);

/* Mean 2-class cross entropy over pixels (SegLoss, src/models/glassrgbd.py:376-383):
 * logits [P][2] (dtype), target [P] int64.  sum[1] double (caller zeroes).                        */
int gwd_seg_ce_sum(const void *logits, const int64_t *target, double *sum, int64_t P, int32_t dtype,
                   void *stream);
int gwd_seg_ce_backward(const void *logits, const int64_t *target, const float *gloss, float scale,
                        void *glogits, int64_t P, int32_t dtype, void *stream);

/* Anchor-weighted depth of PointBasedPred (src/models/points/points_sample.py:277-279): pred[b][p] = sum_r att[b][p][r] *
 * anchor[b][r]  (att [B][P][R] dtype = the softmax over the R point channels, anchor [B][R] fp32, pred [B][P] fp32), R <= 256.
 * backward: datt[b][p][r] = gpred[b][p] * anchor[b][r] (dtype, may be NULL), danchor [B][R] fp32 ACCUMULATED (caller zeroes). */
int gwd_anchor_depth_forward(const void *att, const float *anchor, float *pred, int32_t B, int64_t P, int32_t R,
                             int32_t dtype, void *stream);
int gwd_anchor_depth_backward(const void *att, const float *anchor, const float *gpred, void *datt, float *danchor,
                              int32_t B, int64_t P, int32_t R, int32_t dtype, void *stream);

/* A (window, token, head, channel) operand: element = p[w*ws + t*ts + h*hs + d], channel stride 1.      */
typedef struct {
    void *p;
    int64_t ws, ts, hs;
} gwd_strided;

/* Fused 7x7 window attention, 49 tokens per window, head_dim in {4, 8, 16, 32}:
 *   O = softmax(scale * Q K^T + bias[head] (+ -100 where region[token_i] != region[token_j])) V
 * bias is the dense [heads][49][49] fp32 relative-position bias when rel_index is NULL; with rel_index (int32 [49*49], the
 * reference's relative_position_index buffer) bias is the PARAMETER ITSELF, the [n_rel][heads] table (n_rel = 169), and
 * bias(h, i, j) = table[rel_index[i*49+j]][h] is gathered in the kernel - no dense copy, and the backward accumulates
 * straight into a table-shaped gradient (multiscale_transformerr.py:313-316).  region (optional, int32
 * [windows_per_image][49]) encodes the SW-MSA shift mask.  Replaces the attention core of
 * WindowAttention / WindowClassAttention (src/models/multiscale_transformerr.py:311-328, 538-556, mask
 * :937-955).  The backward recomputes P, writes dQ/dK/dV and ACCUMULATES dbias (fp32 atomics; caller zeroes). */
int gwd_winattn_forward(const gwd_strided *q, const gwd_strided *k, const gwd_strided *v, const gwd_strided *o,
                        const float *bias, const int32_t *rel_index, int32_t n_rel, const int32_t *region, int64_t n_windows,
                        int32_t windows_per_image, int32_t heads, int32_t head_dim, float scale, int32_t dtype, void *stream);
int gwd_winattn_backward(const gwd_strided *q, const gwd_strided *k, const gwd_strided *v, const gwd_strided *go,
                         const gwd_strided *gq, const gwd_strided *gk, const gwd_strided *gv, const float *bias,
                         float *dbias, const int32_t *rel_index, int32_t n_rel, const int32_t *region, int64_t n_windows,
                         int32_t windows_per_image, int32_t heads, int32_t head_dim, float scale, int32_t dbias_head_major,
                         int32_t dtype, void *stream);
/* dbias_head_major (with rel_index only): dbias is a [heads][n_rel] scratch instead of the [n_rel][heads] table gradient - a wave's
 * flush is then one contiguous run of atomics instead of n_rel 4-byte adds in n_rel different 64-byte segments (41-59 us of every
 * backward launch); the caller adds the transposed scratch to the table's gradient.                                             */

/* Line-point-guided query rewrite of the 1/32-stage WindowAttention (src/models/multiscale_transformerr.py:295-310), the two
 * batched einsums of the reference as kernels that write the layout their consumer reads:
 *   ref_scores:  ra[b][t][r][h] = scale * sum_d q[b,t,h,d] * ref_k[b,r,h,d]   (t = window*49 + token; q is the (B*nwin, 49, H, hd)
 *                operand inside the packed qkv projection; ref_k (B, R, H*hd); ra (B, nwin*49, R, H) = the pixel-major map the
 *                3x3 "diffusion" conv runs over).  backward: g -> dq (same operand addressing, every element written) and
 *                d_ref_k fp32 (B, R, H*hd), overwritten.
 *   ref_mix:     att = softmax over r of ra[b,t,:,h];  q_new[b][t][h*hd+d] = sum_r att[b,t,r,h] * ref_v[b,r,h,d]; att (B,T,R,H)
 *                is saved for the backward pass (NULL: not written).  backward: att, g -> d_ra (B,T,R,H), d_ref_v fp32.
 * R <= 128, H <= 64, hd <= 64.  No atomics: sums over the tokens are gathered per output element (bit-reproducible).     */
int gwd_ref_scores_forward(const gwd_strided *q, const void *ref_k, void *ra, int32_t B, int32_t nwin, int32_t R, int32_t H,
                           int32_t hd, float scale, int32_t dtype, void *stream);
int gwd_ref_scores_backward(const gwd_strided *q, const void *ref_k, const void *g, const gwd_strided *dq, float *d_ref_k, int32_t B,
                            int32_t nwin, int32_t R, int32_t H, int32_t hd, float scale, int32_t dtype, void *stream);
int gwd_ref_mix_forward(const void *ra, const void *ref_v, void *q_new, void *att, int32_t B, int32_t T, int32_t R, int32_t H,
                        int32_t hd, int32_t dtype, void *stream);
int gwd_ref_mix_backward(const void *att, const void *ref_v, const void *g, void *d_ra, float *d_ref_v, int32_t B, int32_t T,
                         int32_t R, int32_t H, int32_t hd, int32_t dtype, void *stream);

/* Class-token attention of WindowClassAttention (src/models/multiscale_transformerr.py:560-578), per
 * (window, head): A = softmax_c(scale * sum_n q[n][r] k[n][c]), o[n][r] = sum_c A[r][c] v[n][c]; 49 tokens n,
 * r < 4, c < e with e in {12, 16, 24}.  q/o/gq are (W,49,heads,4) operands, k/v/gk/gv (W,49,heads,e).     */
int gwd_tokattn_forward(const gwd_strided *q, const gwd_strided *k, const gwd_strided *v, const gwd_strided *o,
                        int64_t n_windows, int32_t heads, int32_t e, float scale, int32_t dtype, void *stream);
int gwd_tokattn_backward(const gwd_strided *q, const gwd_strided *k, const gwd_strided *v, const gwd_strided *go,
                         const gwd_strided *gq, const gwd_strided *gk, const gwd_strided *gv, int64_t n_windows,
                         int32_t heads, int32_t e, float scale, int32_t dtype, void *stream);

/* A 3x3 convolution over a 2x nearest-upsampled map (src/models/dense_upsample.py:150-181, F.interpolate + Conv2d) seen from its
 * low-resolution input is a 4x4 / stride 2 / pad 1 convolution: the data gradient and the weight gradient run in that form on
 * gwd_conv_forward / gwd_conv_wgrad_batch, 16 taps per low-res pixel instead of 36.  These two are its tap bookkeeping:
 * collapse: w (Cout,3,3,Cin) fp32 -> wk (Cin,4,4,Cout) in `dtype`, wk[ci][t][s][co] = sum_{kh in G(t), kw in G(s)} w[co][kh][kw][ci],
 *           G(0) = {2}, G(1) = {1,2}, G(2) = {0,1}, G(3) = {0};
 * fold:     D (Cin,4,4,Cout) fp32 (the weight gradient of the 4x4 form) -> dw (Cout,3,3,Cin) fp32 += sum_{t in T(kh), s in T(kw)} D[ci][t][s][co],
 *           T(0) = {2,3}, T(1) = {1,2}, T(2) = {0,1}.                                                                              */
int gwd_upsample_taps_collapse(const float *w, void *wk, int32_t Cout, int32_t Cin, int32_t dtype, void *stream);
int gwd_upsample_taps_fold(const float *D, float *dw, int32_t Cout, int32_t Cin, void *stream);

/* Both class tokens of a WindowClassAttention block in ONE launch (multiscale_transformerr.py:561-578: the depth token and the
 * segmentation token query the same global_k / global_v).  The softmax is per query channel, so the pair is one problem with 8
 * query channels: k / v are staged once and the backward's gk / gv are the SUM over both tokens (what autograd adds up after two
 * gwd_tokattn_backward calls).  bf16 only (-2 otherwise: issue the two single calls), e in {12, 16, 24} (-4).                  */
int gwd_tokattn_pair_forward(const gwd_strided *q, const gwd_strided *q2, const gwd_strided *k, const gwd_strided *v,
                             const gwd_strided *o, const gwd_strided *o2, int64_t n_windows, int32_t heads, int32_t e,
                             float scale, int32_t dtype, void *stream);
int gwd_tokattn_pair_backward(const gwd_strided *q, const gwd_strided *q2, const gwd_strided *k, const gwd_strided *v,
                              const gwd_strided *go, const gwd_strided *go2, const gwd_strided *gq, const gwd_strided *gq2,
                              const gwd_strided *gk, const gwd_strided *gv, int64_t n_windows, int32_t heads, int32_t e,
                              float scale, int32_t dtype, void *stream);

/* CertainSample entirely on the device (src/models/points/points_sample.py:291-364): pred_small [B][hs][ws],
 * pred_large [B][H][W] fp32 sigmoid maps, edges[n_intervals+1] fp32 interval bounds -> coords [B][S][2] fp32
 * ((x/W)*2-1, (y/H)*2-1).  Integer-valued selection; bit-exact against the CPU reference on identical operands. */
int gwd_certain_sample(const float *pred_small, const float *pred_large, float *coords, int32_t B, int32_t hs,
                       int32_t ws, int32_t H, int32_t W, const float *edges, int32_t n_intervals,
                       int32_t sample_num, void *stream);

/* Linear sum assignment of the line matcher on the device (replaces scipy.optimize.linear_sum_assignment at
 * src/models/matcher.py:74 and its 6 host syncs per step).  cost [layers][B][Q][sum_targets] fp32 is the block
 * cost matrix of matcher.py:52-70; image b owns columns col_offsets[b] .. col_offsets[b+1]-1 (int32 [B+1]).
 * query_of_target [layers][sum_targets] int32 receives the query matched to every target (targets <= Q, <= 64).
 * The offsets are DEVICE data: sum_targets may be a fixed capacity larger than col_offsets[B]; the padding columns
 * col_offsets[B] .. sum_targets-1 are never read and receive the dummy query index Q, so one captured launch serves
 * batches with any per-image target counts (max_targets is the host's bound on them, <= 64).                    */
int gwd_lsap(const float *cost, const int32_t *col_offsets, int32_t *query_of_target, int32_t layers, int32_t B,
             int32_t Q, int32_t sum_targets, int32_t max_targets, void *stream);

/* grid_sample (align_corners=False, zero padding) of a pixel-major map (B,H,W,C) at S points per image: coords fp32
 * (B,S,2) = normalised (x, y); mode 0 bilinear, 1 nearest; out fp32 (B,S,C) (src/models/points_sample.py:262-268,
 * multiscale_transformerr.py:688-691).  backward adds the scatter of gout into gmap (map dtype, pre-zeroed by the
 * caller); no gradient w.r.t. coords.                                                                          */
int gwd_point_sample_forward(const void *map, const float *coords, float *out, int32_t B, int32_t H, int32_t W, int32_t C,
                             int32_t S, int32_t mode, int32_t dtype, void *stream);
int gwd_point_sample_backward(const float *gout, const float *coords, void *gmap, int32_t B, int32_t H, int32_t W, int32_t C,
                              int32_t S, int32_t mode, int32_t dtype, void *stream);
/* The same gradient, every element of gmap WRITTEN (no pre-zeroing): a gather over the S <= 256 points per pixel.  Returns -4 when
 * S > 256 (use the pair above).                                                     */
int gwd_point_sample_backward_gather(const float *gout, const float *coords, void *gmap, int32_t B, int32_t H, int32_t W, int32_t C,
                                     int32_t S, int32_t mode, int32_t dtype, void *stream);

/* Nearest sampling of the map as torch.roll(F.pad(map, to (Hf, Wf)), (-shift, -shift), (1, 2)) presents it - the shifted-window frame
 * the 1/32 stage samples its reference points in (multiscale_transformerr.py:662-691) - without building that tensor: coords address
 * the (Hf, Wf) frame, frame pixel (y, x) is map pixel ((y + shift) mod Hf, (x + shift) mod Wf), zero in the padding.  The backward
 * writes every element of gmap (gather form); -4 when S > 256 (pad + roll, then the plain entry points).                          */
int gwd_point_sample_framed_forward(const void *map, const float *coords, float *out, int32_t B, int32_t H, int32_t W, int32_t C,
                                    int32_t S, int32_t Hf, int32_t Wf, int32_t shift, int32_t dtype, void *stream);
int gwd_point_sample_framed_backward(const float *gout, const float *coords, void *gmap, int32_t B, int32_t H, int32_t W, int32_t C,
                                     int32_t S, int32_t Hf, int32_t Wf, int32_t shift, int32_t dtype, void *stream);

/* gwd_weight_prep for many weights in one launch (bf16 outputs).  `jobs` is a DEVICE array; job i owns the blocks
 * [block0_i, block0_{i+1}) of the launch, one 32(n) x 32(c) tile of one tap each: blocks_i = taps*ceil(N/32)*ceil(C/32),
 * block0_0 = 0, total_blocks = sum of blocks_i.
 * w_fwd / w_dgrad may be NULL per job.                                                                          */
typedef struct gwd_prep_job {
    const float *w;          /* fp32 master weight (N, taps, C)                         */
    const float *row_scale;  /* optional per-output-row scale (folded FrozenBN), or NULL */
    void *w_fwd;             /* bf16 (N, taps, C) or NULL                                */
    void *w_dgrad;           /* bf16 (C, taps, N) or NULL                                */
    int32_t N, taps, C;
    int32_t block0;
    int32_t Np, Cg, Cgp;     /* Np > 0: ZERO-PADDED copies - w_fwd is (Np, taps, G*Cgp), w_dgrad (G*Cgp, taps, Np), where the C */
    int32_t reserved;        /* input channels are G = C / Cg groups of Cg, each padded to Cgp; the caller zeroes the copies     */
                             /* once, the launch writes the N x taps x C real entries only.  Np = 0: dense copies as above.      */
} gwd_prep_job;
/* block_job (may be NULL): DEVICE array, job index of every block of the launch (total_blocks int32) - without it each block finds
 * its job by a binary search over the device table (~9 dependent loads in front of one 32 x 32 tile).                       */
int gwd_weight_prep_batch(const gwd_prep_job *jobs, int32_t n_jobs, int32_t total_blocks, const int32_t *block_job, void *stream);

/* The set criterion of the line branch around the device LSAP (gwd_lsap), on PADDED targets (cap columns: column t belongs to image
 * bidx[t], valid[t] = 0 marks padding; the LSAP gives padding columns the dummy query Q).  All tensors fp32 unless noted; K = classes
 * incl. "no object" (<= 8), D = line coordinates (<= 8), else -4.
 *   gwd_match_cost: cost (L,B,Q,cap) = w_line * L1(lines (L,B,Q,D), tgt_lines (cap,D)) - w_class * softmax(logits (L,B,Q,K))[tgt_labels
 *     (cap, int64)]  (HungarianMatcher_Line.forward, src/models/matcher.py:52-70).
 *   gwd_set_losses_forward: per layer l: target_class (L,B,Q) int32 (the matched target's label at query qot[l][t], K-1 elsewhere),
 *     ce[l] = sum(nll * w) / sum(w) with w = class_weight[target class], wsum[l] = sum(w), l1[l] = sum over valid t of
 *     |lines[l, bidx[t], min(qot, Q-1)] - tgt_lines[t]|_1 / max(num_items[0] / world, 1)   (src/models/glassrgbd.py:160-170,231-244).
 *   gwd_set_losses_backward: dlogits (fully written) and dlines (ADDED to: caller zeroes) from g_ce[L] / g_l1[L] (either may be NULL). */
int gwd_match_cost(const float *logits, const float *lines, const float *tgt_lines, const int64_t *tgt_labels, float *cost, int32_t L,
                   int32_t B, int32_t Q, int32_t cap, int32_t K, int32_t D, float w_line, float w_class, void *stream);
int gwd_set_losses_forward(const float *logits, const float *lines, const float *tgt_lines, const int64_t *tgt_labels, const int32_t *bidx,
                           const int32_t *valid, const int32_t *qot, const float *class_weight, const float *num_items, float world,
                           int32_t *target_class, float *ce, float *l1, float *wsum, int32_t L, int32_t B, int32_t Q, int32_t cap,
                           int32_t K, int32_t D, void *stream);
int gwd_set_losses_backward(const float *logits, const float *lines, const float *tgt_lines, const int32_t *bidx, const int32_t *valid,
                            const int32_t *qot, const float *class_weight, const float *num_items, float world,
                            const int32_t *target_class, const float *wsum, const float *g_ce, const float *g_l1, float *dlogits,
                            float *dlines, int32_t L, int32_t B, int32_t Q, int32_t cap, int32_t K, int32_t D, void *stream);

/* Padding mask of one feature level + sine position embeddings from it (src/models/backbone.py:81-88: F.interpolate(nearest) of the
 * batch mask; src/models/position_encoding.py:28-48: PositionEmbeddingSine).  Two stages, either may be skipped:
 *   mask_full != NULL: mask_level (B,h,w) u8 = nearest-resized mask_full (B,H,W) u8 (non-zero = padding), counts (B,h,w,2) int16 =
 *                      cumulative counts of un-masked pixels along y and along x;
 *   out != NULL:       out (B,h,w,2F) fp32 = [sin|cos interleaved of y / dim_t[c]] | [the same of x], from `counts`; normalize: counts
 *                      scaled to 2 pi by the last row / column (eps 1e-6); dim_t (F) fp32 = temperature^(2 (c/2) / F) from the caller.
 * One level's counts serve any number of embeddings (the reference builds several widths from one mask).                          */
int gwd_pos_sine(const uint8_t *mask_full, uint8_t *mask_level, int16_t *counts, const float *dim_t, float *out, int32_t B, int32_t H,
                 int32_t W, int32_t h, int32_t w, int32_t F, int32_t normalize, void *stream);

/* The ResNet stem in one forward-only kernel: conv 7x7 / stride 2 / pad 3 (3 -> 64 channels) + folded FrozenBatchNorm + ReLU +
 * max-pool 3x3 / stride 2 / pad 1 - conv1 / bn1 / relu / maxpool of torchvision's resnet50 as src/models/backbone.py:65-92 runs
 * them (frozen there: backbone.py:62-64).  x bf16 [B][H][W][3] -> y bf16 [B][Hp][Wp][64], Hc = (H-1)/2+1, Hp = (Hc-1)/2+1 (W alike).
 * `packed` = the weights as gwd_stem_pack leaves them (w fp32 [64][7][7][3], times scale[64] when given: the BN scale), i.e.
 * GWD_STEM_PACKED_ELEMS bf16 values in the kernel's operand order; shift fp32 [64] (BN shift) or NULL.  dtype: GWD_BF16 only (-2). */
#define GWD_STEM_PACKED_ELEMS (14 * 2 * 64 * 8)
int gwd_stem_pack(const float *w, const float *scale, void *packed, void *stream);
int gwd_stem_forward(const void *x, const void *packed, const float *shift, void *y, int32_t B, int32_t H, int32_t W, int32_t dtype,
                     void *stream);

/* The other half of the zero-padded copies: a layer that ran on padded channel counts (its activations keep the padded width, so every
 * conv / Linear on them takes the LDS-DMA route; the 30 / 60 / 300-channel pyramid of src/models/points/points_sample.py:45-125
 * is the user) gets its weight gradient in the padded shape; this folds it back: dst (N, taps, G*Cg) += src (.., taps, G*Cgp)
 * rows 0..N-1, group g channels 0..Cg-1.  Up to GWD_UNPAD_BATCH jobs per call = ONE launch; jobs are read on the host (by value
 * in the kernel arguments); block0 is filled in by the library.                                                              */
#define GWD_UNPAD_BATCH 24
typedef struct gwd_unpad_job {
    const float *src;        /* fp32 (>= N, taps, G*Cgp)                                */
    float *dst;              /* fp32 (N, taps, G*Cg), accumulated                       */
    int32_t N, taps, G, Cg, Cgp;
    int32_t block0;
} gwd_unpad_job;
int gwd_unpad_add_batch(const gwd_unpad_job *jobs, int32_t n_jobs, void *stream);

/* y = a + gelu((u - mean_bc(u)) * rsqrt(var_bc(u) + eps)), statistics over the L positions of image b for channel c;
 * a, u, y are (B, L, C), channel innermost (src/models/multiscale_transformerr.py:299-302: conv -> instance
 * normalisation -> GELU -> residual of the reference-point attention logits).  part: fp32 scratch [B][S][C][2]
 * (S slices per image, caller-chosen, fully overwritten); stat: fp32 [B][C][2] = (mean, rstd) saved for backward.
 * backward: du from gy (the gradient w.r.t. a is gy itself).  C * sizeof(dtype) / 16 must be a power of two.     */
int gwd_inorm_gelu_forward(const void *a, const void *u, void *y, float *part, float *stat, int64_t B, int64_t L, int32_t C,
                           int32_t S, float eps, int32_t dtype, void *stream);
int gwd_inorm_gelu_backward(const void *gy, const void *u, const float *stat, float *part, void *du, int64_t B, int64_t L,
                            int32_t C, int32_t S, int32_t dtype, void *stream);

/* Window partition (gather != 0) / reverse (gather == 0) of a (B,H,W,C) token map into (B*nWin,49,C) 7x7 windows
 * with zero padding to multiples of 7 and cyclic shift `shift` (src/models/multiscale_transformerr.py:667-676,
 * 705-707 / 730-747).  C * sizeof(dtype) must be a multiple of 16.  residual (reverse only, may be NULL): a (B,H,W,C)
 * map added to the result - the block's residual stream, `x = shortcut + x` of :749.                           */
int gwd_window_map(const void *src, void *dst, const void *residual, int32_t B, int32_t H, int32_t W, int32_t C, int32_t shift,
                   int32_t gather, int32_t dtype, void *stream);
/* The same for n <= GWD_WINMAP_JOBS maps of one geometry (B, H, W, shift) and channel counts C[i] in ONE launch: the three maps a
 * Swin block with class tokens hands over together (features, depth tokens, seg tokens: multiscale_transformerr.py:700-747).
 * src / dst / residual: HOST arrays of n device pointers (residual, or single entries of it, may be NULL).        */
#define GWD_WINMAP_JOBS 4
int gwd_window_map_multi(const void *const *src, void *const *dst, const void *const *residual, const int32_t *C, int32_t n,
                         int32_t B, int32_t H, int32_t W, int32_t shift, int32_t gather, int32_t dtype, void *stream);

/* Pixel-major resampling ([B][H][W][C]).  mode 0 = bilinear align_corners=True (PSP branches of
 * src/models/points/points_sample.py:114-121, CertainSample :293), mode 1 = legacy nearest
 * floor(dst*in/out) (src/models/multiscale_transformerr.py:1193,1230,1240,1267).  The backward kernels are
 * gathers over each source pixel's footprint (no atomics, bitwise reproducible).                    */
enum { GWD_RESAMPLE_BILINEAR_AC = 0, GWD_RESAMPLE_NEAREST = 1 };
/* ldy: pixel pitch of y in elements (0 = C).  ldy > C writes the result into a channel slice of a wider map (the PSP concat of
 * points_sample.py:114-122 without a concat pass); vector-sized C and ldy only, else -4.                          */
int gwd_resample_forward(const void *x, void *y, int32_t B, int32_t Hs, int32_t Ws, int32_t Ho, int32_t Wo,
                         int32_t C, int32_t mode, int32_t ldy, int32_t dtype, void *stream);
/* gate (may be NULL) [B][Hs][Ws][C] + gate_act (GWD_ACT_RELU / GWD_ACT_ELU): gx is multiplied by act'(.) of the activation whose
 * OUTPUT is `gate` (the source map of an up-sampling convolution, see gwd_conv_desc.gate); nearest mode with C a multiple of 4, else -4. */
int gwd_resample_backward(const void *gy, void *gx, int32_t B, int32_t Hs, int32_t Ws, int32_t Ho, int32_t Wo,
                          int32_t C, int32_t mode, const void *gate, int32_t gate_act, int32_t dtype, void *stream);
/* gwd_resample_backward as two separable passes (x, then y) through a caller-provided fp32 scratch tmp[B][Ho][Ws][C]:
 * the same sums in the same order, but 2r+2 taps per thread instead of (2r+2)^2 (13x faster at the 16x pyramid
 * branches).  Returns -4 when C is not a multiple of 16 bytes (use gwd_resample_backward).                       */
/* ldg: pixel pitch of gy in elements (0 = C): the gradient of a channel slice read in place.                        */
int gwd_resample_backward_sep(const void *gy, float *tmp, void *gx, int32_t B, int32_t Hs, int32_t Ws, int32_t Ho, int32_t Wo,
                              int32_t C, int32_t mode, int32_t ldg, int32_t dtype, void *stream);

/* k x k / stride k average pooling (nn.AvgPool2d(k, stride=k), points_sample.py:61-75), floor mode.  */
/* dst (B,H,W,C) = [residual +] src (B,Ho,Wo,C) placed on the pixels (stride*i, stride*j), zero elsewhere: with a plain GEMM over
 * the output pixels this is the data gradient of a 1x1 convolution with stride > 1 (the ResNet downsample convs,
 * src/models/backbone.py:90-92), a quarter of the transposed-gather work.  C a multiple of 16 bytes, else -4.        */
int gwd_stride_place(const void *src, const void *residual, void *dst, int32_t B, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t C,
                     int32_t stride, int32_t dtype, void *stream);

/* The four average pools of the PSP module (F.avg_pool2d with k = 16, 8, 4, 2; src/models/points/points_sample.py:107-113) from
 * ONE pass over x (B,H,W,C): p_k (B,H/k,W/k,C).  Backward in one pass as well: gx = g_pass + sum_k g_k[y/k][x/k] / k^2, where g_pass
 * (may be NULL; pixel pitch ldg, 0 = C) is the gradient that reaches the map directly - on this path the first channel slice of the
 * concat's gradient, read in place.  Any g_k may be NULL.  H, W >= 16; C (and ldg) multiples of 16 bytes, else -4.            */
int gwd_psp_pool_forward(const void *x, void *p16, void *p8, void *p4, void *p2, int32_t B, int32_t H, int32_t W, int32_t C,
                         int32_t dtype, void *stream);
int gwd_psp_pool_backward(const void *g_pass, const void *g16, const void *g8, const void *g4, const void *g2, void *gx, int32_t B,
                          int32_t H, int32_t W, int32_t C, int32_t ldg, int32_t dtype, void *stream);
int gwd_avgpool_forward(const void *x, void *y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                        int32_t dtype, void *stream);
int gwd_avgpool_backward(const void *gy, void *gx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                         int32_t dtype, void *stream);

/* sq[0] += sum g^2 over a flat fp32 buffer (double; caller zeroes).  clip_grad_norm_ numerator.    */
int gwd_sqnorm(const float *g, double *sq, int64_t n, void *stream);
/* Fused clip_grad_norm_(max_norm) + torch.optim.AdamW step on a flat fp32 range
 * (src/engine_glassrgbd.py:157-159, src/main_glassrgbd.py:59-66).  coef = min(1, max_norm/(sqrt(sq)+1e-6))
 * is computed on device from sq[0]; grad_scale pre-multiplies g (1/world_size for DDP mean).
 * Optionally refreshes the bf16 shadow copy of the parameters.                                    */
int gwd_adamw_step(float *p, const float *g, float *m, float *v, void *p_bf16, const double *sq,
                   int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                   float bias_corr1, float bias_corr2, float max_norm, float grad_scale, void *stream);

/* Dense evaluation metrics of a batch of B images with H*W = HW pixels each, accumulated on the device.
 * Replaces the device->host copies and per-image numpy of evaluate(): the prediction clamp and GT validity mask
 * (src/engine_glassrgbd.py:249-253), compute_depth_errors (src/util/metrics.py:198-218), the argmax + ignore-255
 * confusion counts of compute_mean_ioU / get_confusion_matrix (src/util/metrics.py:37-74, engine :236-241) and the
 * running sums `depth_eval_measures[:9] += measures; [9] += 1` (engine :262-263).
 *   pred_depth [B][HW] (depth_dtype), gt_depth [B][HW] fp32 - both NULL: no depth part;
 *   seg_logits: class c of pixel i of image b at seg_logits[b*seg_sb + i*seg_sp + c*seg_sc] (seg_dtype; element strides, so
 *               both the reference's (B,2,H,W) and the pixel-major (B,H,W,2) layout are read in place),
 *   seg_gt [B][HW] int64 (255 = ignore)                     - both NULL: no segmentation part;
 *   workspace: gwd_query_workspace(GWD_WS_EVAL, {B, HW}) bytes, fully overwritten;
 *   measures [B][9] f64 OUT: silog, abs_rel, log10, rms, sq_rel, log_rms, d1, d2, d3 of each image (engine :204 order;
 *               all NaN for an image without a valid pixel, as numpy's mean of an empty array);
 *   running [10] f64 IN/OUT: += the nine measures of every image, in image order; [9] += B;
 *   confusion [4] int64 IN/OUT: [gt*2 + pred] += pixel counts.
 * Per-pixel terms are formed in fp32 exactly as numpy forms them (IEEE division, so the three threshold counts are
 * bit-exact); sums are f64 and folded in a fixed order: results are bit-reproducible.  Two launches, no atomics.     */
#define GWD_EVAL_NSUM 10
int gwd_eval_accumulate(const void *pred_depth, const float *gt_depth, const void *seg_logits, int64_t seg_sb,
                        int64_t seg_sp, int64_t seg_sc, const int64_t *seg_gt, void *workspace, double *measures,
                        double *running, int64_t *confusion, int32_t B, int64_t HW, float min_depth,
                        float max_depth, int32_t depth_dtype, int32_t seg_dtype, void *stream);

/* PlaneLoss (src/models/glassrgbd.py:385-450, --with_plane_norm_loss) of ONE image, on the device: Sobel normals of the
 * predicted depth (src/models/losses/sobel.py:5-27), the masks of up to P <= 64 line triangles restricted to the valid
 * pixels (the reference: matplotlib.path.Path.contains_points on the host; same crossing test here, exact in integers),
 * per-plane biased variances of both normal components; loss = sum over planes with >= min_area pixels / max(1, count).
 *   depth [H][W] (dtype), valid [H][W] uint8, tri [P][6] int64 = (x0,y0,x1,y1,x2,y2) already rounded and clamped (:413-418),
 *   n_planes: DEVICE int32 - only the first *n_planes triangles count (the reference's data-dependent top_num, :400, without
 *   a host round trip); workspace: gwd_query_workspace(GWD_WS_PLANE, {P, H*W}) bytes;
 *   stats [4*P + 1] f64 OUT: per plane n, mean_x, mean_y, active; then the number of active planes (read by the backward);
 *   loss [1] fp32 OUT.  backward: gdepth [H][W] (dtype) = gloss[0] * d loss / d depth (every element written).          */
int gwd_plane_loss_forward(const void *depth, const uint8_t *valid, const int64_t *tri, const int32_t *n_planes, int32_t P,
                           int32_t H, int32_t W, int32_t min_area, void *workspace, double *stats, float *loss,
                           int32_t dtype, void *stream);
int gwd_plane_loss_backward(const void *depth, const uint8_t *valid, const int64_t *tri, const int32_t *n_planes, int32_t P,
                            int32_t H, int32_t W, const double *stats, const float *gloss, void *gdepth, int32_t dtype,
                            void *stream);

/* Multi-head attention core on the matrix cores, flash style (csrc/mfattn.hip): head_dim 32, bf16, ANY L and S; no L x S
 * matrix reaches memory (replaces q*scaling, bmm, masked_fill, softmax, dropout, bmm and the head split / merge copies of
 * src/models/multi_head_attention.py:329-375, and their autograd backward).
 *   q: row (b, l) at q + (b*L + l)*q_ts, head h in channels [32h, 32h+32); q_ts, k_ts, ... are TOKEN strides in elements
 *   (multiples of 8, 16-byte aligned bases), so slices of a packed in-projection are read - and their gradients written -
 *   in place; k, v likewise with S tokens per image.  key_padding_mask [B][S] uint8 (nonzero = excluded) or NULL;
 *   mult [B][H][L][S] bf16 dropout multipliers (0 or 1/(1-p)) or NULL.
 *   forward:  out (b, l) rows with token stride o_ts, heads merged; lse [B][H][L] fp32 = log-sum-exp of the scaled, masked
 *             scores of every query (saved for the backward pass).
 *   backward: go -> gq, gk, gv (same addressing); out / lse from the forward; delta [B][H][L] fp32 is scratch.  Every output
 *             row is written by exactly one wave (no atomics): bit-reproducible.
 * Returns -2 for a dtype other than GWD_BF16 (the fp32 parity mode keeps the unfused path).                               */
int gwd_mha_flash_forward(const void *q, const void *k, const void *v, int64_t q_ts, int64_t k_ts, int64_t v_ts,
                          const uint8_t *key_padding_mask, const void *mult, void *out, int64_t o_ts, float *lse, int32_t B,
                          int32_t H, int32_t L, int32_t S, float scale, int32_t dtype, void *stream);
int gwd_mha_flash_backward(const void *q, const void *k, const void *v, const void *go, const void *out, int64_t q_ts,
                           int64_t k_ts, int64_t v_ts, int64_t go_ts, int64_t o_ts, const uint8_t *key_padding_mask,
                           const void *mult, const float *lse, float *delta, void *gq, void *gk, void *gv, int64_t gq_ts,
                           int64_t gk_ts, int64_t gv_ts, int32_t B, int32_t H, int32_t L, int32_t S, float scale, int32_t dtype,
                           void *stream);

/* Batch assembly from raw decoded images (the tail of the input pipeline): ToTensor + Normalize
 * (src/datasets/transforms_depth.py:618-660; torchvision's to_tensor / normalize: x/255, - mean, / std in fp32), the
 * dataset's depth_mm / 1000 and label > 0 (src/datasets/glassrgbd_norhint.py:277-281) and collate_fn_aux's zero padding
 * + padding mask (src/util/misc.py:273-313), one launch for up to GWD_COLLATE_BATCH images.
 *   jobs[i]: device pointers to image i as decoded - rgb uint8 [h][w][3], depth_mm int32 [h][w], labels uint8 [h][w] (a
 *            pointer may be NULL when the matching output is NULL); read on the host, passed in the kernel arguments;
 *   images [n][H][W][3] (dtype, pixel-major), mask [n][H][W] uint8 (1 = padding), depth [n][H][W] fp32 metres,
 *   seg [n][H][W] int64 {0,1}; every element is written (padding = 0); any output may be NULL.                      */
/* The geometric transforms of the input pipeline on decoded images (src/datasets/transforms_depth.py:59-372: hflip / vflip, crop,
 * resize), bit-exact with the Pillow calls the reference makes through torchvision; tables are built by the caller on the host
 * (gw_depth_amd/data.py shows how; oracle/pil_resize_ref.py is the numpy restatement the tests pin against Pillow).
 * gwd_resample_u8_pass: ONE separable pass of Image.resize(BILINEAR) over uint8 pixels with C interleaved channels:
 *   axis 1 (horizontal): dst [other][n_out][C],  dst[j][o][c] = clip8((2^21 + sum_t src[row(j)][col(first_o + t)][c] * kk[o][t]) >> 22)
 *   axis 0 (vertical):   dst [n_out][other][C],  dst[o][j][c] = ... src[row(first_o + t)][col(j)][c] ...
 *   bounds (n_out,2) int32 = (first source index, tap count), kk (n_out,ksize) int32 fixed-point coefficients (22 fractional bits);
 *   index maps along the resampled axis i -> base0 + step0 * i and along the other axis j -> base1 + step1 * j (step = +-1): a flip
 *   and / or a crop window of the source folded into the read; src_row_stride in elements.
 * gwd_gather2d: dst[y][x] = src[ytab[y]][xtab[x]] for elements of 1-4 bytes (NEAREST resize / flip / crop of depth and label maps;
 *   flips and crops of RGB), src_row_stride in BYTES.                                                                             */
int gwd_resample_u8_pass(const uint8_t *src, uint8_t *dst, const int32_t *bounds, const int32_t *kk, int32_t ksize, int32_t axis,
                         int32_t n_out, int32_t other, int32_t C, int64_t src_row_stride, int32_t base0, int32_t step0, int32_t base1,
                         int32_t step1, void *stream);
int gwd_gather2d(const void *src, void *dst, const int32_t *ytab, const int32_t *xtab, int32_t oh, int32_t ow,
                 int64_t src_row_stride_bytes, int32_t elem_bytes, void *stream);

/* One adjustment of the reference's ColorJitter (src/datasets/transforms_depth.py:551-600) on a uint8 RGB image of npix pixels,
 * bit-exact with torchvision's PIL path: mode 0 brightness, 1 contrast, 2 saturation (Pillow ImageEnhance: Image.blend with a black /
 * mean-luma / luma degenerate image, factor >= 0), 3 hue (RGB -> HSV, H + shift with uint8 wrap-around, HSV -> RGB; here `factor` is
 * the SHIFT 0..255 = uint8(hue_factor * 255) as the caller's host arithmetic casts it: int(hue_factor * 255) & 255).  scratch: 8 bytes of device memory (the contrast mean's luma sum; may be NULL for the other modes).  Data-pipeline
 * entry point: not meant for HIP-graph capture (mode 1 clears its scratch with a memset node).                                     */
int gwd_color_adjust(const uint8_t *rgb, uint8_t *out, uint64_t *scratch, int64_t npix, int32_t mode, float factor, void *stream);

/* Strided batched GEMM  C[b0][b1] (M x N) = alpha * A[b0][b1] (M x K) * B[b0][b1] (N x K)^T  (csrc/bmm.hip).
 * Every operand has inner stride 1; a_ld / b_ld / c_ld are the element strides of the outer matrix dimension, *_sb0 / *_sb1 those of
 * the two batch dimensions (0 broadcasts an operand over a batch dimension).  a_kmajor / b_kmajor: the operand is stored [k][row]
 * (k is the outer dimension) instead of [row][k] - with these the gradients of a product need no transposed copies.
 * c_is_f32_accumulate: C is fp32 whatever `dtype` says and the result is ADDED to it (fp32 atomics; the caller zeroes it); required
 * for splits > 1, which cuts the reduction over K into `splits` workgroups per tile (long reductions onto a small result).
 * Replaces torch.bmm of the point heads (src/models/points/points_sample.py:271-279) and, in the fp32 parity mode, the two attention
 * products of src/models/multi_head_attention.py:347-371 (exact fp32: v_mfma_f32_32x32x2_f32).                                   */
typedef struct {
    const void *a, *b;
    void *c;
    int64_t a_sb0, a_sb1, a_ld, b_sb0, b_sb1, b_ld, c_sb0, c_sb1, c_ld;
    int32_t M, N, K, nb0, nb1;
    int32_t a_kmajor, b_kmajor, c_is_f32_accumulate, splits;
    float alpha;
    int32_t dtype;        /* GWD_F32 / GWD_BF16 of a, b (and of c unless c_is_f32_accumulate) */
} gwd_bmm_desc;
int gwd_bmm(const gwd_bmm_desc *d, void *stream);

#define GWD_COLLATE_BATCH 16
typedef struct {
    const void *rgb, *depth_mm, *labels;
    int32_t h, w;
} gwd_image_job;
int gwd_collate(const gwd_image_job *jobs, int32_t n, int32_t H, int32_t W, const float *mean, const float *std,
                void *images, uint8_t *mask, float *depth, int64_t *seg, int32_t dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif
