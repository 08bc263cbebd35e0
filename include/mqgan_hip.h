/* mqgan_hip.h - C ABI of libmqgan_hip.so: the MI355X (gfx950) kernels behind the
 * MQGAN PreEncoder VQ-GAN training / encode / decode hot path.
 *
 * The reference (ZDisket/MQGAN) is pure Python on PyTorch and has no FFI of its
 * own; each entry point below therefore cites the
 * reference Python code whose arithmetic it replaces, and INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host
 *   - fp32 activations are channels-last: 1-D maps (B, T, C), 2-D maps (N, H, W, C)
 *   - frame masks are uint8, (B, T), 1 = padded (the reference's True = padded)
 *   - `stream` is a hipStream_t (pass 0 for the null stream); calls are
 *     asynchronous, allocate nothing, and never synchronise, so they can be
 *     captured into a hipGraph
 *   - return value: 0 on success, non-zero on failure; mq_last_error() then
 *     returns a description (thread-local).  Shapes are validated on the host
 *     before any launch; a rejected call launches nothing.
 */
#ifndef MQGAN_HIP_H
#define MQGAN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *mqStream; /* hipStream_t */

/* Building blocks behind the planned convolution calls (the Winograd / polyphase / row / column transforms) and the
 * fused-Winograd prototype: libmqgan_hip.so - the PRODUCT library - uses them internally and does NOT export them; the test-variant
 * build (make -C mqgan_amd/csrc variants -> libmqgan_hip_variants.so, -DMQ_VARIANTS=1) exports them for tests/, tools/ and
 * micro-benchmarks.  A consumer of the product library reaches them through mq_conv_plan / mq_conv_*_planned. */
#if defined(MQ_VARIANTS) && MQ_VARIANTS
#define MQ_BLOCKS_API
#else
#define MQ_BLOCKS_API __attribute__((visibility("hidden")))
#endif

int mq_abi_version(void);
const char *mq_last_error(void);
/* Name of the kernel instantiation the last mq_conv_* / mq_gemm-type call on this thread launched, spelled as rocprofv3
 * prints it (e.g. "conv_nt_kernel<128, false, 3>"); "" before the first such call.  For labelling timings: the dispatch
 * decides the kernel, callers ask. */
const char *mq_last_kernel(void);

/* ------------------------------------------------------------------ quantiser
 * FSQ (reference quantizer.py:109-114 bound, :128-140 quantize, :39-42 round_ste,
 * :177-181 codes_to_indices, :183-205 indices_to_codes).  d <= 8 levels.  The
 * per-dimension constants are computed on the host as the reference does
 * (quantizer.py:68-72,111-113,132) and passed by value. */
typedef struct {
    int d;
    int levels[8];
    int half_width[8];
    int basis[8];
    float half_l[8];
    float offset[8];
    float shift[8];
    /* thresh[j][k], k < levels[j]-1: smallest float z for which dimension j quantises to level >= k+1
     * under the reference arithmetic (found on the host by bisection, mqgan_amd/quantizer.py). */
    float thresh[8][8];
} mqFsqDesc;

/* z (n, d) -> codes (n, d) [nullable], idx (n) int32 [nullable] */
int mq_fsq_forward(const mqFsqDesc *desc_host, const float *z, int64_t n, float *codes, int32_t *idx, mqStream stream);
/* straight-through gradient: dz = dcodes * half_l * (1 - tanh^2(z + shift)) / half_width */
int mq_fsq_backward(const mqFsqDesc *desc_host, const float *z, const float *dcodes, int64_t n, float *dz,
                    mqStream stream);
/* idx (n) int64 -> codes (n, d) */
int mq_fsq_decode(const mqFsqDesc *desc_host, const int64_t *idx, int64_t n, float *codes, mqStream stream);

/* Generic L2 nearest neighbour (no reference code, arithmetic
 * defined in oracle/vq_ref.c): cnorm[k] = fma-chain_d c[k][d]^2; idx[n] = argmin_k
 * fma(-2, fma-chain_d x[n][d]*c[k][d], cnorm[k]), ties to the lowest k, NaN distances count as -inf.
 * Requires d % 4 == 0.  best_dist nullable. */
int mq_l2_codebook_norms(const float *codebook, int k, int d, float *cnorm, mqStream stream);
size_t mq_l2_argmin_workspace(int64_t n);
int mq_l2_argmin(const float *x, const float *codebook, const float *cnorm, int64_t n, int k, int d, int32_t *idx,
                 float *best_dist, void *workspace, mqStream stream);
/* The same search, same results bit for bit (indices AND best distances), past the f32-MFMA roof (csrc/vq_screen.hip;
 * SURVEY.md 8d "bf16 candidate scoring + exact fp32 re-rank"): every (row, code) pair is scored ONCE on the bf16 matrix pipe
 * (three launches: x -> bf16 + per-row error bound; scores with the code norm as the accumulator's initial value, every code
 * within the proven bound of the row's RUNNING best appended to a private list; lists filtered against the final best, the
 * survivors re-ranked with the exact k-ordered f32 distance of the definition above, idx / best_dist written); rows without a
 * finite bound (NaN / inf / |x| |c| beyond 1e18) or with an overrun list are searched exhaustively in exact arithmetic.  Needs
 * d % 8 == 0 (mq_l2_screen_supported).
 *   prepared : per-codebook blob of mq_l2_screen_codebook_bytes(k, d) bytes, 256-byte aligned, filled once per codebook by
 *              mq_l2_screen_prepare (code norms, accumulator seeds -norm / 2, bf16 copy, max norm)
 *   workspace: mq_l2_screen_workspace(n, k, d) bytes, 256-byte aligned (bf16 copy of x; 4 private lists of 16 entries per row
 *              and code split; per-row bound and running best)
 * mq_l2_screen_stats (synchronises `stream`): entries the last call on that workspace appended to its lists, list entries
 * available, and the number of rows it searched exhaustively (still exact). */
int mq_l2_screen_supported(int64_t n, int k, int d);
size_t mq_l2_screen_codebook_bytes(int k, int d);
size_t mq_l2_screen_workspace(int64_t n, int k, int d);
int mq_l2_screen_prepare(const float *codebook, int k, int d, void *prepared, mqStream stream);
int mq_l2_argmin_screened(const float *x, const float *codebook, const void *prepared, int64_t n, int k, int d, int32_t *idx,
                          float *best_dist, void *workspace, size_t workspace_bytes, mqStream stream);
int mq_l2_screen_stats(const void *workspace, int64_t n, int k, int d, uint64_t *appended, uint64_t *slots, uint64_t *rows_over,
                       mqStream stream);
/* codebook EMA update (row Q3; no reference code - arithmetic AND summation order are defined in oracle/vq_ref.c
 * section 3, and the kernels are bit-exact against it, deterministic, no float atomics):
 *   cluster_size = decay*cluster_size + (1-decay)*count_k;  embed_sum = decay*embed_sum + (1-decay)*sum_{n: idx[n]==k} x[n];
 *   codebook = embed_sum / smoothed(cluster_size)  (Laplace smoothing with eps; dead codes keep embed_sum / smoothed).
 * Rows whose idx lies outside [0, k) are ignored.  idx out of a NaN row: mq_l2_argmin counts a NaN distance as -inf,
 * so it always returns a valid index. */
size_t mq_codebook_ema_workspace(int64_t n, int k, int d);
int mq_codebook_ema_update(const float *x, const int32_t *idx, int64_t n, int k, int d, float decay, float eps,
                           float *cluster_size, float *embed_sum, float *codebook, void *workspace, size_t workspace_bytes,
                           mqStream stream);

/* ------------------------------------------------------------------ convolutions
 * Implicit-GEMM convolution on f32 MFMA (v_mfma_f32_32x32x2_f32), NHWC.  Replaces
 * every nn.Conv1d / nn.Conv2d / nn.Linear on the path: ResidualBlock1D convs
 * (attentions.py:490-495,522), UNetRefiner convs (preencoder.py:51-53,86-130,165),
 * discriminator convs (discriminators.py:139-167), Linear projections
 * (preencoder.py:321,334,336,354,356).  1-D convs are H = R = 1; Linear is R = S = 1.
 *   x  (N, H, W, C)   w (groups, K, R, S, C)   bias (groups, K)   y (N, P, Q, K)
 * `groups` splits the BATCH (N % groups == 0): images of group g use filter bank g
 * (this is how the eight independent band discriminators run as one launch,
 * discriminators.py:280-311).  pad_h/pad_w are the top/left paddings; P, Q are given
 * explicitly so that causal (left-only) padding is expressible (attentions.py:471-474).
 * `precision` selects the matrix engine.  MQ_PRECISION_F32 (0): exact f32 products, the 1e-4 parity path.
 * MQ_PRECISION_BF16 (1): the counterpart of the reference's torch.autocast(bfloat16) region (train.py:330-333,
 * 386,437): operands are rounded to bf16 on their way into LDS (v_mfma_f32_32x32x16_bf16), accumulation,
 * bias and every tensor in HBM stay f32.  Layers that do not fit the vector path (C or K not a multiple of 4)
 * run in f32 either way. */
#define MQ_PRECISION_F32 0
#define MQ_PRECISION_BF16 1
/* MQ_PRECISION_F32X3 (2): f32 results from the bf16 matrix pipe.  Every operand is split exactly into three bf16
 * terms (8+8+8 significant bits) inside the kernel and 6 of the 9 cross products (all terms down to 2^-24
 * relative) are accumulated in f32: the error of an f32 FMA chain (measured: same rms error against f64 as
 * MQ_PRECISION_F32), not bit-identical to it.  Opt-in; the default engine is MQ_PRECISION_F32. */
#define MQ_PRECISION_F32X3 2
typedef struct {
    int N, H, W, C;
    int K, R, S;
    int P, Q;
    int stride_h, stride_w;
    int pad_h, pad_w;
    int dil_h, dil_w;
    int groups;
    int precision;
} mqConvDesc;

int mq_conv_fprop(const mqConvDesc *desc_host, const float *x, const float *w_krsc, const float *bias, float *y,
                  mqStream stream);
/* fprop with the discriminators' activation fused into the store (discriminators.py:219,232):
 *   y = padded(row) ? 0 : LeakyReLU_slope(conv(x) + bias);  padded(row m) = mask[(m / d1) * d2 + m % d3] != 0  (mask may be NULL)
 * Any engine (desc->precision); C % 4 == 0.  The pre-activation is not stored: LeakyReLU's derivative is read off the
 * sign of y (mq_pointwise_bwd with h = y). */
int mq_conv_fprop_leaky(const mqConvDesc *desc_host, const float *x, const float *w_krsc, const float *bias, const uint8_t *mask,
                        int64_t d1, int64_t d2, int64_t d3, float slope, float *y, mqStream stream);
/* bf16-storage variants for MQ_PRECISION_BF16: the operands are bf16 COPIES (mq_cast_bf16) of the f32 tensors - half
 * the bytes and half the loads of the variant that rounds inside the kernel, bit-identical results.  Requires the
 * reduction channel count (C for fprop, K for dgrad) to be a multiple of 16; outputs stay f32. */
int mq_cast_bf16(const float *in, uint16_t *out_bf16, int64_t n, mqStream stream);
int mq_conv_fprop_h(const mqConvDesc *desc_host, const uint16_t *x_bf16, const uint16_t *w_krsc_bf16, const float *bias,
                    float *y, mqStream stream);
int mq_conv_dgrad_h(const mqConvDesc *desc_host, const uint16_t *dy_bf16, const uint16_t *w_crsk_bf16, float *dx,
                    mqStream stream);
/* ---- bf16-STORAGE convolutions ("H2" kernels, csrc/igemm_h2.hip): MQ_PRECISION_BF16 with activations, output
 * gradients and effective weights held as bf16 tensors in HBM - what the reference's autocast region stores
 * (train.py:220,523: conv outputs are bf16 tensors there).  Block tile 256 x {64..256} x 64, both operands staged by
 * LDS-DMA, f32 accumulation.  Shapes: C % 32 == 0, K % 32 == 0, strides <= 2 (mq_h2_supported; the weight gradient
 * additionally needs Q >= 16: mq_h2_wgrad_supported); every other layer stays on the kernels above.
 *   mq_h2_weight_image: f32 effective weights (KRSC for fprop, CRSK for dgrad, as mq_weight_prep writes them) ->
 *                       bf16 image [groups * rows][Kp], Kp = R*S*Cin rounded up to 64, reduction order
 *                       (chunk of 64 or 32 channels, tap, channel); mq_h2_weight_elems gives its size.
 *   mq_conv_fprop_h2  : y = epilogue(conv(x) + bias);  act 0: none;  act 2: LeakyReLU(slope) then zero on padded
 *                       rows (mask index of row m: (m / d1) * d2 + m % d3), as mq_conv_fprop_leaky.  y is bf16
 *                       (out_f32 = 0) or f32 (out_f32 = 1).
 *   mq_conv_dgrad_h2  : dx = conv_transpose(dy), dx bf16 or f32; strided layers run by stride parity.
 *                       Layers with 3, 5 or 7 stride-1 taps along the columns on whole 64-channel chunks (C % 64 == 0 forward,
 *                       K % 64 == 0 backward; stride 1 both ways, column dilation 1, >= 128 output columns) take
 *                       conv_h2s_kernel: one activation image per (chunk, tap row) for all column taps, results bit-identical to
 *                       the per-tap kernel (MQGAN_H2_NO_SHARE=1 in the environment selects that one for A/B runs).
 *   mq_conv_wgrad_h2  : dw_krsc (f32, the layout mq_weight_prep_bwd reads) and, if dbias != NULL, dbias (f32) from
 *                       bf16 x and dy; split-K over pixels with a deterministic slab reduction.  Layers with three stride-1
 *                       column taps, C % 128 == 0 and K % 192 == 0 take conv_wgrad_h2s_kernel (one x image per K-step for the
 *                       three taps; equal to the per-tap kernel within rounding; MQGAN_H2_NO_SHARE_WGRAD=1 for A/B runs). */
int mq_h2_supported(const mqConvDesc *desc_host);
int mq_h2_wgrad_supported(const mqConvDesc *desc_host);
size_t mq_h2_weight_elems(const mqConvDesc *desc_host, int dgrad);
int mq_h2_weight_image(const mqConvDesc *desc_host, int dgrad, const float *w, uint16_t *w_h2, mqStream stream);
int mq_conv_fprop_h2(const mqConvDesc *desc_host, const uint16_t *x, const uint16_t *w_h2, const float *bias,
                     const uint8_t *mask, int64_t d1, int64_t d2, int64_t d3, int act, float slope, int out_f32, void *y,
                     mqStream stream);
int mq_conv_dgrad_h2(const mqConvDesc *desc_host, const uint16_t *dy, const uint16_t *wt_h2, int out_f32, void *dx,
                     mqStream stream);
/* mq_conv_dgrad_h2 for a conv whose bf16 input x came out of a LeakyReLU(slope) + masked_fill layer (discriminators.py:219,
 * 232): dx is returned with respect to that layer's PRE-activation, dx * (x > 0 ? 1 : slope), zero on its padded frames
 * (mask index of input pixel m = (n*H + h)*W + w: (m / d1) * d2 + m % d3) - the producer's activation backward without a
 * pass of its own.  bf16 results are bit-identical to mq_conv_dgrad_h2 followed by mq_pointwise_bwd_h. */
int mq_conv_dgrad_h2_gated(const mqConvDesc *desc_host, const uint16_t *dy, const uint16_t *wt_h2, const uint16_t *x,
                           const uint8_t *mask, int64_t d1, int64_t d2, int64_t d3, float slope, int out_f32, void *dx,
                           mqStream stream);
size_t mq_conv_wgrad_h2_workspace(const mqConvDesc *desc_host);
int mq_conv_wgrad_h2(const mqConvDesc *desc_host, const uint16_t *x, const uint16_t *dy, float *dw_krsc, float *dbias,
                     void *workspace, size_t workspace_bytes, mqStream stream);
/* dx (N,H,W,C) = conv_transpose(dy (N,P,Q,K)); weights given as (groups, C, R, S, K) */
int mq_conv_dgrad(const mqConvDesc *desc_host, const float *dy, const float *w_crsk, float *dx, mqStream stream);
/* the f32-tensor counterpart of mq_conv_dgrad_h2_gated (all f32-tensor engines; same summation order as mq_conv_dgrad) */
int mq_conv_dgrad_gated(const mqConvDesc *desc_host, const float *dy, const float *w_crsk, const float *x, const uint8_t *mask,
                        int64_t d1, int64_t d2, int64_t d3, float slope, float *dx, mqStream stream);
/* dw (groups, K, R, S, C); split-K over pixels with a deterministic slab reduction.
 * workspace: mq_conv_wgrad_workspace() bytes (may be 0 -> pass NULL). */
size_t mq_conv_wgrad_workspace(const mqConvDesc *desc_host);
int mq_conv_wgrad(const mqConvDesc *desc_host, const float *x, const float *dy, float *dw_krsc, void *workspace,
                  size_t workspace_bytes, mqStream stream);
/* Same, plus dbias (groups, K) = column sums of dy, taken from the dy tiles the weight-gradient GEMM streams anyway
 * (no second pass over dy).  workspace: mq_conv_wgrad_bias_workspace() bytes, never 0. */
size_t mq_conv_wgrad_bias_workspace(const mqConvDesc *desc_host);
int mq_conv_wgrad_bias(const mqConvDesc *desc_host, const float *x, const float *dy, float *dw_krsc, float *dbias,
                       void *workspace, size_t workspace_bytes, mqStream stream);
/* Thin convolutions (single input or output channel: the first conv of each discriminator and of the refiner,
 * discriminators.py:139-150, preencoder.py:148; the logits convs, discriminators.py:162-167, and the refiner's
 * last conv, preencoder.py:165).  They run as a tap gather / fold around a 1x1 GEMM so that the wide tensor is
 * read once.  T = R*S rounded up to a multiple of 4; `desc` is the descriptor of the thin conv itself.
 *   gather : img (N,H,W)   -> cols (N,P,Q,T)   cols[.,t] = img at tap t of the output pixel (0 outside / t >= R*S)
 *   scatter: cols (N,P,Q,T)-> img (N,H,W)      transpose of gather (gradient wrt the image)
 *   reduce : u (N,H,W,T)   -> y (N,P,Q)        y = bias[group] + sum_t u[input pixel of tap t][t]
 *   expand : dy (N,P,Q)    -> du (N,H,W,T)     transpose of reduce */
int mq_taps_gather(const mqConvDesc *desc_host, int T, const float *img, float *cols, mqStream stream);
int mq_taps_scatter(const mqConvDesc *desc_host, int T, const float *cols, float *img, mqStream stream);
int mq_taps_reduce(const mqConvDesc *desc_host, int T, const float *u, const float *bias, float *y, mqStream stream);
int mq_taps_expand(const mqConvDesc *desc_host, int T, const float *dy, float *du, mqStream stream);
/* out[c] = sum_r x[r][c]  (bias gradients; deterministic two-stage).  workspace: 1024*cols floats. */
int mq_colsum(const float *x, int64_t rows, int cols, float *out, float *workspace, mqStream stream);

/* ------------------------------------------------------------------ weight preparation
 * Effective conv weights, produced in the two layouts the implicit GEMM reads.
 * mode 0: plain weight  (nn.Linear / nn.Conv*)           w = v
 * mode 1: weight norm   (torch weight_norm, both APIs)    w = g * v / ||v||_(per out channel)
 * v is in the reference's layout (K, C, R*S) ; g is (K).  inv_norm (K) is saved for backward. */
int mq_weight_prep(int mode, const float *v, const float *g, int K, int C, int RS, float *w_krsc, float *w_crsk,
                   float *inv_norm, mqStream stream);
/* dw_krsc -> dv (K, C, RS) and, mode 1, dg (K) */
int mq_weight_prep_bwd(int mode, const float *dw_krsc, const float *v, const float *g, const float *inv_norm, int K,
                       int C, int RS, float *dv, float *dg, mqStream stream);
/* Legacy spectral norm (torch.nn.utils.spectral_norm, discriminators.py:139,162):
 * optional one power iteration (training mode) updating u (K), v (C*RS) in place, then
 * sigma = u^T W v and w = W / sigma in both layouts.  sigma_out (1).  scratch: K + C*RS floats. */
int mq_spectral_prep(const float *w_orig, float *u, float *v, int K, int C, int RS, int power_iteration, float eps,
                     float *w_krsc, float *w_crsk, float *sigma_out, float *scratch, mqStream stream);
/* dW_orig = dW/sigma - (sum(dW * W_orig) / sigma^2) * u v^T ; scratch: 1 + 256 floats */
int mq_spectral_prep_bwd(const float *dw_krsc, const float *w_orig, const float *u, const float *v,
                         const float *sigma, int K, int C, int RS, float *dw_orig, float *scratch, mqStream stream);
/* The same layer of up to MQ_SN_MAX_BANDS band discriminators (discriminators.py:252-300: eight identical networks) in one
 * launch per stage, without power iteration (eval-mode discriminators).  w_orig / u / v / dw_orig are HOST arrays of
 * `bands` device pointers; band g fills rows [g*K, (g+1)*K) of the stacked KRSC filter (and [g*C, (g+1)*C) of the CRSK
 * one), sigma_out[g].  scratch: bands * K floats (forward), bands * 256 floats (backward).  Per band the arithmetic and
 * its order are those of mq_spectral_prep / mq_spectral_prep_bwd. */
#define MQ_SN_MAX_BANDS 16
int mq_spectral_prep_bands(const float *const *w_orig, const float *const *u, const float *const *v, int bands, int K, int C,
                           int RS, float *w_krsc, float *w_crsk, float *sigma_out, float *scratch, mqStream stream);
int mq_spectral_prep_bands_bwd(const float *dw_krsc, const float *const *w_orig, const float *const *u, const float *const *v,
                               const float *sigma, int bands, int K, int C, int RS, float *const *dw_orig, float *scratch,
                               mqStream stream);

/* ------------------------------------------------------------------ fused ConvBlock2D
 * preencoder.py:277-301 as instantiated at :322,:351: 1-channel C0 x T image ->
 * weight-normed depthwise 5x5 -> mask -> 1x1 conv 1->C0 -> mask -> APTx(1,1,.5) ->
 * Dropout(p) -> 1x1 conv C0->1, i.e. per pixel  y = b2 + sum_k w2[k]*drop*aptx(w1[k]*v + b1[k]).
 * x, y, v, dv: (B, T, C0).  The (B, C0, C0, T) intermediate of the reference is never
 * materialised.  Padded frames come out as b2 (reference quirk).
 * wdw: 25 effective depthwise taps [i over channel axis][j over time]; scalars by pointer. */
typedef struct {
    const float *wdw;   /* 25 */
    const float *bdw;   /* 1  */
    const float *w1;    /* C0 effective point-wise weights */
    const float *b1;    /* C0 */
    const float *w2;    /* C0 conv_out weights */
    const float *b2;    /* 1 */
} mqCb2dWeights;

/* Dropout (p_drop > 0): the keep decision of every (pixel, hidden unit) is drawn once, in the forward call (a counter-seeded
 * xorshift + Weyl stream per pixel; same Bernoulli(1 - p) law as the pointwise kernels' counter-based words, 16 bits per decision), left as
 * one bit each in keep_bits - mq_convblock2d_keep_bytes(B, T, C0) bytes, [C0 / 32][B * T * C0] words - and read back by the backward
 * call: nothing is drawn twice.  keep_bits may be NULL when p_drop == 0. */
size_t mq_convblock2d_keep_bytes(int B, int T, int C0);
int mq_convblock2d_fwd(const mqCb2dWeights *w_host, const float *x, const uint8_t *mask, int B, int T, int C0,
                       float p_drop, uint64_t seed, float *y, float *v_save, uint32_t *keep_bits, mqStream stream);
/* grads: dx (B,T,C0); dwdw (25), dbdw (1), dw1 (C0), db1 (C0), dw2 (C0), db2 (1).
 * workspace bytes: mq_convblock2d_bwd_workspace(B, T, C0). */
size_t mq_convblock2d_bwd_workspace(int B, int T, int C0);
int mq_convblock2d_bwd(const mqCb2dWeights *w_host, const float *x, const float *v_save, const float *dy,
                       const uint8_t *mask, int B, int T, int C0, float p_drop, const uint32_t *keep_bits, float *dx, float *dwdw,
                       float *dbdw, float *dw1, float *db1, float *dw2, float *db2, void *workspace, mqStream stream);

/* ------------------------------------------------------------------ pointwise
 * One fused elementwise pass:  t = h (+ r_pre);  mask_pre: t = 0 on padded frames;
 * a = act(t);  a = dropout(a);  a += r_post;  mask_post: a = 0 on padded frames.
 * act: 0 none, 1 APTx (alpha=1; beta, gamma read from device pointers, attentions.py:34-35),
 * 2 LeakyReLU(0.2) (discriminators.py:170).
 * rows x C tensors; the frame-mask index of row m is (m / d1) * d2 + (m % d3). */
typedef struct {
    int64_t rows;
    int C;
    int act;
    int mask_pre, mask_post;
    int64_t d1, d2, d3;
    float p_drop;
    uint64_t seed;
} mqPointwiseDesc;

int mq_pointwise_fwd(const mqPointwiseDesc *desc_host, const float *h, const float *r_pre, const float *r_post,
                     const uint8_t *mask, const float *beta, const float *gamma, float *out, mqStream stream);
/* dt (gradient wrt h and r_pre), dr_post is dout after mask_post (written to dpost if non-null);
 * dbeta/dgamma (1 each, nullable) need workspace of 2*1024 floats. */
int mq_pointwise_bwd(const mqPointwiseDesc *desc_host, const float *dout, const float *h, const float *r_pre,
                     const uint8_t *mask, const float *beta, const float *gamma, float *dt, float *dpost,
                     float *dbeta, float *dgamma, float *workspace, mqStream stream);

/* bf16-storage variants (activations held as bf16 tensors, see the H2 convolutions): same semantics, arithmetic in
 * f32, results rounded to bf16 on the store; C % 8 == 0 and 16-byte aligned tensors. */
int mq_pointwise_fwd_h(const mqPointwiseDesc *desc_host, const uint16_t *h, const uint16_t *r_pre, const uint16_t *r_post,
                       const uint8_t *mask, const float *beta, const float *gamma, uint16_t *out, mqStream stream);
int mq_pointwise_bwd_h(const mqPointwiseDesc *desc_host, const uint16_t *dout, const uint16_t *h, const uint16_t *r_pre,
                       const uint8_t *mask, const float *beta, const float *gamma, uint16_t *dt, uint16_t *dpost,
                       float *dbeta, float *dgamma, float *workspace, mqStream stream);
int mq_cast_f32(const uint16_t *in_bf16, float *out, int64_t n, mqStream stream);
int mq_avgpool_t2_fwd_h(const uint16_t *x, int N, int T, int F, int C, const uint8_t *mask, uint16_t *y, mqStream stream);
int mq_avgpool_t2_bwd_h(const uint16_t *dy, int N, int T, int F, int C, const uint8_t *mask, uint16_t *dx, mqStream stream);
int mq_upcat_fwd_h(const uint16_t *x, const uint16_t *skip, int N, int T, int F, int C1, int C2, const uint8_t *mask,
                   uint16_t *out, mqStream stream);
int mq_upcat_bwd_h(const uint16_t *dout, int N, int T, int F, int C1, int C2, const uint8_t *mask, uint16_t *dx,
                   uint16_t *dskip, mqStream stream);

/* frame masks */
int mq_sequence_mask(const int32_t *lengths, int B, int T, uint8_t *mask, mqStream stream);           /* preencoder.py:15-24 */
/* max-pool (padding stays padded) along T with window = stride, optional ceil mode
 * (preencoder.py:63-65, discriminators.py:223-229); frames past Tin count as `pad_value`. */
int mq_mask_pool(const uint8_t *in, int B, int Tin, int Tout, int stride, uint8_t *out, mqStream stream);
int mq_mask_upsample2(const uint8_t *in, int B, int Tin, uint8_t *out, mqStream stream);              /* preencoder.py:68-70 */

/* UNet data movement (preencoder.py:105-130): time-only avg-pool /2, nearest x2 fused with
 * the channel concat, and their backward passes.  x (N, T, F, C).  `mask` (optional, uint8, nonzero = padded)
 * has the OUTPUT's (N, T_out) frames: the consuming ConvBlock's entry masked_fill (preencoder.py:93) applied
 * in the same pass (forward: padded output frames are zeros; backward: their gradient is dropped).  It needs
 * the vectorised path (C, C1, C2 multiples of 8). */
int mq_avgpool_t2_fwd(const float *x, int N, int T, int F, int C, const uint8_t *mask, float *y, mqStream stream);
int mq_avgpool_t2_bwd(const float *dy, int N, int T, int F, int C, const uint8_t *mask, float *dx, mqStream stream);
/* out (N, 2T, F, C1+C2) = cat(upsample2(x (N,T,F,C1)), skip (N,2T,F,C2)) */
int mq_upcat_fwd(const float *x, const float *skip, int N, int T, int F, int C1, int C2, const uint8_t *mask, float *out,
                 mqStream stream);
int mq_upcat_bwd(const float *dout, int N, int T, int F, int C1, int C2, const uint8_t *mask, float *dx, float *dskip,
                 mqStream stream);
/* band split for the multi-bin discriminator (discriminators.py:298, 211):
 * x (B, T, F) -> out (n_bins*B, F/n_bins, T): out[j*B+b][f][t] = x[b][t][j*bs+f]; bwd is the inverse. */
int mq_band_split_fwd(const float *x, int B, int T, int F, int n_bins, float *out, mqStream stream);
int mq_band_split_bwd(const float *dout, int B, int T, int F, int n_bins, float *dx, mqStream stream);
/* y[r][0:C1] = a[r], y[r][C1:C1+C2] = b[r] */
int mq_concat2(const float *a, const float *b, int64_t rows, int C1, int C2, float *y, mqStream stream);
int mq_axpby(const float *a, const float *b, float alpha, float beta_, int64_t n, float *out, mqStream stream);

/* ------------------------------------------------------------------ CBAM1D / SE
 * attentions.py:217-273, 310-365, 393-419 incl. the no-op-mask quirk. */
/* per (b, c): max over ALL t (with argmax), mean over VALID t.  x (B, T, C). */
int mq_cam_pool(const float *x, const uint8_t *mask, int B, int T, int C, float *maxv, int32_t *argmax_t, float *avgv,
                mqStream stream);
/* gate (B, C) = sigmoid(a + b) */
int mq_sigmoid_add(const float *a, const float *b, int64_t n, float *out, mqStream stream);
/* da = dout * out * (1 - out) */
int mq_sigmoid_add_bwd(const float *dout, const float *out, int64_t n, float *da, mqStream stream);
int mq_relu_fwd(const float *x, int64_t n, float *y, mqStream stream);
int mq_relu_bwd(const float *dy, const float *y, int64_t n, float *dx, mqStream stream);
/* SAM: feat (B, T, 2) = [max_c, mean_c] of gate*x, with arg-max channel */
int mq_sam_pool(const float *x, const float *gate, int B, int T, int C, float *feat, int32_t *argmax_c,
                mqStream stream);
/* s = sigmoid(conv_k7(feat)) ; out = s*gate*x + x.  w7: (2, ks) as in Conv1d(2,1,ks).weight[0] */
int mq_sam_apply(const float *x, const float *gate, const float *feat, const float *w7, int ks, int B, int T, int C,
                 float *s_save, float *out, mqStream stream);
/* backward of the whole CBAM given dout: produces dx (B,T,C), dgate (B,C), dw7 (2*ks).
 * workspace: mq_cbam_bwd_workspace bytes */
size_t mq_cbam_bwd_workspace(int B, int T, int C, int ks);
int mq_cbam_bwd(const float *dout, const float *x, const float *gate, const float *feat, const int32_t *argmax_c,
                const float *s_save, const float *w7, int ks, int B, int T, int C, float *dx, float *dgate, float *dw7,
                void *workspace, mqStream stream);
/* (dgate then flows through mq_sigmoid_add_bwd and the two MLP linears back to dmax / davg)
 * adds the pooling gradients into dx: dx[b, argmax_t[b,c], c] += dmax[b,c]; dx[b,t,c] += davg[b,c]/count_b (valid t) */
int mq_cam_pool_bwd(const float *dmax, const float *davg, const int32_t *argmax_t, const uint8_t *mask, int B, int T,
                    int C, float *dx, mqStream stream);

/* masked SE (discriminators.py:40-67): squeeze (B, C) = sum over valid (h, w) / max(count, 1).
 * x (B, H, W, C); the padded mask is per (b, w) (time on W). */
int mq_se_squeeze(const float *x, const uint8_t *mask, int B, int H, int W, int C, float *squeeze, mqStream stream);
int mq_se_scale_fwd(const float *x, const float *excite, int B, int HW, int C, float *y, mqStream stream);
/* dx = dy*excite ; dexcite (B,C) = sum_p dy*x.  mq_se_squeeze_bwd then ADDS dsq[b,c]/count on valid pixels. */
int mq_se_scale_bwd(const float *dy, const float *x, const float *excite, int B, int HW, int C, float *dx,
                    float *dexcite, mqStream stream);
int mq_se_squeeze_bwd(const float *dsq, const uint8_t *mask, int B, int H, int W, int C, float *dx_accum,
                      mqStream stream);

/* ------------------------------------------------------------------ losses
 * Every loss on the path is a ratio of sums over the batch (losses.py:26-32, 40-50, 178-181; train.py:38-45).  The
 * kernels are split into *_sums (per-rank numerators and counts in a small device buffer), *_final (the loss from the
 * sums) and the backward pass (which reads the counts from the same buffer).  Between the two the host may all-reduce
 * the buffer over the data-parallel ranks: the normalisers are then GLOBAL, every rank computes the global loss value,
 * and the SUM of the ranks' gradients is the gradient of the single-device step on the global batch, ragged or not
 * (SURVEY.md 8e).  With one process the buffers are used as they are.
 *
 * MaskedMelLoss("mse") (losses.py:146-182): loss = mean_g( sum_g / (count_g + 1e-12) ) for group sizes g1 and g2 at
 * once.  x, y (B, T, C).  sums: C per-column sums of (x-y)^2 over valid frames + 1 (number of valid frames).
 * workspace: 64*C floats. */
int mq_mel_loss_sums(const float *x, const float *y, const int32_t *lengths, int B, int T, int C, float *sums,
                     float *workspace, mqStream stream);
int mq_mel_loss_final(const float *sums, int C, int g1, int g2, float *out2, mqStream stream);
/* dx = (w1 * dloss(g1) + w2 * dloss(g2)) * gout[0] */
int mq_mel_loss_bwd(const float *x, const float *y, const int32_t *lengths, int B, int T, int C, int g1, int g2,
                    float w1, float w2, const float *sums, const float *gout, float *dx, mqStream stream);
/* LSGANLoss (losses.py:5-116).  Logits are single-channel maps (B, H, W).  Masks mark PADDED elements (1), the loss
 * mask of the reference is their complement: mask_mode 0 = frame mask (B, W), constant along H (what the
 * discriminators produce and train.py:385,395 passes); mask_mode 1 = one byte per element (B, H, W) - any mask the
 * reference accepts; NULL = everything valid.  real and fake have their own masks (losses.py:103-104).
 *   mq_lsgan_d_sums : sums6 = [S m_r (r-1)^2, S m_f f^2, S m_r r, S m_f f, S m_r, S m_f]
 *   mq_lsgan_ema    : the EMA chain of n consecutive calls (losses.py:36-58) from n x 6 sums: state (4 floats:
 *                     ema_real, ema_fake, initialised, unused - ONE state shared by all nine discriminators,
 *                     train.py:216) is advanced n times; ema_out[i] = the two EMAs call i's LeCam term uses
 *   mq_lsgan_d_lecam: lsums2 = [S m_r relu(r - ema_f)^2, S m_f relu(ema_r - f)^2]
 *   mq_lsgan_d_final: out[0] = 0.5*(mse_real + mse_fake) + lecam_real + lecam_fake */
int mq_lsgan_d_sums(const float *real, const float *fake, const uint8_t *real_padded, const uint8_t *fake_padded,
                    int mask_mode, int B, int H, int W, float *sums6, mqStream stream);
int mq_lsgan_ema(const float *sums6, int n, float decay, float *state, float *ema_out, mqStream stream);
int mq_lsgan_d_lecam(const float *real, const float *fake, const uint8_t *real_padded, const uint8_t *fake_padded,
                     int mask_mode, int B, int H, int W, const float *ema2, float *lsums2, mqStream stream);
int mq_lsgan_d_final(const float *sums6, const float *lsums2, float *out, mqStream stream);
int mq_lsgan_d_bwd(const float *real, const float *fake, const uint8_t *real_padded, const uint8_t *fake_padded,
                   int mask_mode, int B, int H, int W, const float *sums6, const float *ema2, const float *gout,
                   float *dreal, float *dfake, mqStream stream);
/* generator_loss: sums2 = [S m (f-1)^2, S m]; loss = mq_ratio_final(sums2, 0) = (count > 0 ? sum / count : 0) */
int mq_lsgan_g_sums(const float *fake, const uint8_t *padded, int mask_mode, int B, int H, int W, float *sums2,
                    mqStream stream);
int mq_ratio_final(const float *sums2, float eps, float *out, mqStream stream);   /* eps > 0: sum / (count + eps) */
int mq_lsgan_g_bwd(const float *fake, const uint8_t *padded, int mask_mode, int B, int H, int W, const float *sums2,
                   const float *gout, float *dfake, mqStream stream);
/* masked_mae (train.py:38-45): pred/target (B, H, W, C), padded mask (B, W).  sums2: [sum |pred - target|, valid
 * count]; loss = mq_ratio_final(sums2, 1e-8).  workspace: 512 floats. */
int mq_masked_mae_sums(const float *pred, const float *target, const uint8_t *padded, int B, int H, int W, int C,
                       float *sums2, float *workspace, mqStream stream);
/* leaky_slope: 1 for the plain gradient; s != 1 when pred = LeakyReLU_s(h) of a layer that takes its gradient with respect
 * to h (mq_conv_dgrad_gated below): dpred *= pred > 0 ? 1 : s. */
int mq_masked_mae_bwd(const float *pred, const float *target, const uint8_t *padded, int B, int H, int W, int C,
                      const float *sums2, const float *gout, float leaky_slope, float *dpred, mqStream stream);

/* ------------------------------------------------------------------ Winograd F(mh x mw, 3x3), tiles 3 x 3, 4 x 4, 4 x 5 (and 6 x 6)
 * The refiner's 3x3 / stride 1 / pad 1 convolutions (preencoder.py:86-102) as (mh + 2)(mw + 2) batch-grouped GEMMs around
 * HBM-bound transforms (csrc/winogradn.hip): 25 products per 9 outputs (2.78 per output), 36 per 16 (2.25) or 42 per 20 (2.10)
 * instead of 9 per output; f32 products on the same MFMA engine, f32 accumulation.  Transform matrices per axis:
 * csrc/wino_tables.h (point sets chosen for f32 accuracy).  th = ceil(H / mh), tw = ceil(W / mw) tiles per image, np =
 * (mh + 2)(mw + 2) positions; the GEMM descriptor is (N = np * N * th, H = 1, W = tw, R = S = 1, groups = np).
 *   forward : V = input_transform(x) (np, tiles, C);  U = weight_transform(w, flip 0) (np, K, C);  M = mq_conv_fprop(that desc,
 *             V, U) (np, tiles, K);  y = output_transform(M, bias [, act 2: LeakyReLU(slope) + mask as mq_conv_fprop_leaky])
 *   dgrad   : the same with dy as input and weight_transform(w, flip 1) = U of the filter rotated by 180 degrees, (np, C, K)
 *   wgrad   : Z = A dY A^T (mq_winon_dy_fused, which in the same pass over dy also writes Vd = B^T dy B for the input gradient
 *             and per-block column sums [mq_winon_dy_fused_rows(...)][K] for the bias gradient; Vd / sums may be NULL; K <= 1024);
 *             P = mq_conv_wgrad(that desc, V, Z) (np, K, C);  dw = wgrad_inverse(P) (KRSC)
 * The *_s2d / *_d2s functions are the same tiles over the POLYPHASE form of a 5x5 / stride-2 / pad-2 layer (the patch
 * discriminator's wide layers, discriminators.py:86-173): x[2p + rho][2q + sigma][c] as 4C channels on the (H/2, W/2) grid and
 * g'[(rho, sigma, c)][a][b] = w[c][2a + rho][2b + sigma] (zero where an index would be 5) - 11.1 / 9 / 8.4 instead of 25 products
 * per output.  input_transform_s2d reads x with that indexing (X' is never materialised); output_transform_d2s writes the input
 * gradient at full resolution and applies the producer layer's gate (as mq_conv_dgrad_gated).  H and W even.
 * mq_conv_plan / mq_conv_*_planned below run these sequences; the entry points stay public for callers with their own scheduling.
 * Tile 6 x 6 (64 products per 36 outputs, 1.78) is built but never chosen by the plan unless asked for (wino_tile 6): measured on
 * config 2 it takes the step from 191.7 to 187.7 ms at twice the rounding error of 4 x 5 (6.1e-6 against 3.2e-6 of the layer's output,
 * relative L2 against float64).  For that tile mq_winon_dy_fused is two launches (Z, then Vd as an input transform).
 * (ABI 8 folded the F(3x3) kernels of mq_wino3_* into these and removed F(2x2, 3x3), mq_wino_*: no configuration selected it.) */
MQ_BLOCKS_API int mq_winon_input_transform(const float *x, int N, int H, int W, int C, float *V, int mh, int mw, mqStream stream);
MQ_BLOCKS_API int mq_winon_output_transform(const float *M, const float *bias, int N, int H, int W, int K, int act, float slope, const uint8_t *padded,
                              int64_t d1, int64_t d2, int64_t d3, float *y, int mh, int mw, mqStream stream);
MQ_BLOCKS_API int mq_winon_weight_transform(const float *w_krsc, int K, int C, int flip_transpose, float *U, int mh, int mw, mqStream stream);
MQ_BLOCKS_API int mq_winon_wgrad_inverse(const float *P, int K, int C, float *dw_krsc, int mh, int mw, mqStream stream);
MQ_BLOCKS_API int mq_winon_dy_fused_rows(int N, int H, int W, int K, int mh, int mw);
MQ_BLOCKS_API int mq_winon_dy_fused(const float *dy, int N, int H, int W, int K, float *Z, float *Vd, float *bias_partial, int mh, int mw, mqStream stream);
MQ_BLOCKS_API int mq_winon_input_transform_s2d(const float *x, int N, int H, int W, int C, float *V, int mh, int mw, mqStream stream);
MQ_BLOCKS_API int mq_winon_weight_transform_s2d(const float *w_krsc, int K, int C, int flip_transpose, float *U, int mh, int mw, mqStream stream);
MQ_BLOCKS_API int mq_winon_output_transform_d2s(const float *M, int N, int H, int W, int C, const float *gate, float slope, const uint8_t *padded_in,
                                  int64_t d1, int64_t d2, int64_t d3, float *dx, int mh, int mw, mqStream stream);
MQ_BLOCKS_API int mq_winon_wgrad_inverse_s2d(const float *P, int K, int C, float *dw_krsc, int mh, int mw, mqStream stream);

/* ------------------------------------------------------------------ one-dimensional Winograd F(m, 3) along the rows
 * The multi-bin discriminator's (3 x k) layers (discriminators.py:251-290: 3 taps and stride 1 along the bins, k taps and
 * stride 1 or 2 along time): m + 2 row products per m output rows instead of 3 m (csrc/wino1d.hip), m = 2, 4 or 5
 * (mq_wino1d_tile(H): the tile with the fewest products for H rows - 5 for the 10-bin bands: 14 instead of 30).
 * th = ceil(H / m) row tiles, np = m + 2 positions; every tensor f32, channels-last, channel counts multiples of 4;
 * `groups` = batch groups of the layer (N % groups == 0).
 *   forward : V = input_transform(x) (np, N, th, W, C);  U = weight_transform(w_krsc (groups * K, 3 * S, C));
 *             M = mq_conv_fprop(desc{N = np * N * th, H = 1, W, C, K, R = 1, S, stride (1, sw), pad (0, pw), groups = np * groups},
 *             V, U_krsc) (np, N, th, Q, K);  y = output_transform(M, bias): act 2 = LeakyReLU(slope) then zero where
 *             padded[(m / d1) * d2 + m % d3] (m = output pixel), as mq_conv_fprop_leaky
 *   backward: Z = dy_transform(dy) (np, N, th, Q, K) and, if bias_partial != NULL, deterministic per-block column sums of dy:
 *             bias_partial [mq_wino1d_dy_rows(...)][groups * K], db = its column sums (mq_colsum);
 *             dV = mq_conv_dgrad(that desc, Z, U_crsk);  dx = dx_transform(dV): with gate != NULL the result is multiplied by
 *             LeakyReLU'(gate) and zeroed where padded_in[...] (the producer layer's activation backward, mq_conv_dgrad_gated's
 *             contract);  dU = mq_conv_wgrad(that desc, V, Z);  dw_krsc = wgrad_inverse(dU). */
int mq_wino1d_tile(int H);
MQ_BLOCKS_API int mq_wino1d_input_transform(const float *x, int N, int H, int W, int C, int m, float *V, mqStream stream);
MQ_BLOCKS_API int mq_wino1d_weight_transform(const float *w_krsc, int groups, int K, int C, int S, int m, float *U_krsc, float *U_crsk, mqStream stream);
MQ_BLOCKS_API int mq_wino1d_output_transform(const float *M, const float *bias, int N, int H, int Q, int K, int groups, int m, int act, float slope,
                               const uint8_t *padded, int64_t d1, int64_t d2, int64_t d3, float *y, mqStream stream);
MQ_BLOCKS_API int mq_wino1d_dy_rows(int N, int H, int Q, int K, int groups, int m);
MQ_BLOCKS_API int mq_wino1d_dy_transform(const float *dy, int N, int H, int Q, int K, int groups, int m, float *Z, float *bias_partial, mqStream stream);
MQ_BLOCKS_API int mq_wino1d_dx_transform(const float *dV, int N, int H, int W, int C, int m, const float *gate, float slope, const uint8_t *padded_in,
                           int64_t d1, int64_t d2, int64_t d3, float *dx, mqStream stream);
MQ_BLOCKS_API int mq_wino1d_wgrad_inverse(const float *dU, int groups, int K, int C, int S, int m, float *dw_krsc, mqStream stream);

/* ------------------------------------------------------------------ one-dimensional Winograd F(m, s) along the columns (time)
 * Convolutions with `taps` = 3, 5 or 7 column taps, stride 1 and no dilation along the columns, R row taps with any stride,
 * `pad_left` zero columns in front (the right padding follows from the output width Q), batch groups (csrc/winow.hip): the
 * encoder / decoder blocks' 1-D convolutions (attentions.py:95-130), the patch discriminator's 3x3 / stride-(2, 1) layers
 * (discriminators.py:86-173) and the (1 x 5) row convolutions inside the multi-bin discriminator's F(5, 3) form.
 * m = mq_winow_tile(taps) outputs per tile (4, 4, 2), np = m + taps - 1 positions, tw = ceil(Q / m) tiles per row.
 *   forward : V = input_transform(x) (np, N, H, tw, C);  (U_krsc, U_crsk) = weight_transform(w_krsc (groups * K, R * taps, C));
 *             M = mq_conv_fprop(desc{N = np * N, H, W = tw, C, K, R, S = 1, stride (sh, 1), pad (ph, 0), groups = np * groups},
 *             V, U_krsc) (np, N, P, tw, K);  y = output_transform(M, bias) (N, P, Q, K), epilogue as mq_wino1d_output_transform
 *   backward: Z = dy_transform(dy) (np, N, P, tw, K) (+ per-block column sums of dy);  dV = mq_conv_dgrad(that desc, Z, U_crsk);
 *             dx = dx_transform(dV) (gate / padded_in as mq_wino1d_dx_transform);  dU = mq_conv_wgrad(that desc, V, Z);
 *             dw_krsc = wgrad_inverse(dU). */
int mq_winow_tile(int taps);
MQ_BLOCKS_API int mq_winow_input_transform(const float *x, int N, int H, int W, int C, int taps, int Q, int pad_left, float *V, mqStream stream);
MQ_BLOCKS_API int mq_winow_weight_transform(const float *w_krsc, int groups, int K, int C, int R, int taps, float *U_krsc, float *U_crsk, mqStream stream);
MQ_BLOCKS_API int mq_winow_output_transform(const float *M, const float *bias, int N, int P, int Q, int K, int taps, int groups, int act, float slope,
                              const uint8_t *padded, int64_t d1, int64_t d2, int64_t d3, float *y, mqStream stream);
MQ_BLOCKS_API int mq_winow_dy_rows(int N, int P, int Q, int K, int taps, int groups);
MQ_BLOCKS_API int mq_winow_dy_transform(const float *dy, int N, int P, int Q, int K, int taps, int groups, float *Z, float *bias_partial, mqStream stream);
MQ_BLOCKS_API int mq_winow_dx_transform(const float *dV, int N, int H, int W, int C, int taps, int Q, int pad_left, const float *gate, float slope,
                          const uint8_t *padded_in, int64_t d1, int64_t d2, int64_t d3, float *dx, mqStream stream);
MQ_BLOCKS_API int mq_winow_wgrad_inverse(const float *dU, int groups, int K, int C, int R, int taps, float *dw_krsc, mqStream stream);

/* ------------------------------------------------------------------ optimiser
 * clip_grad_norm_ + Adam on flat arenas (train.py:401-407,485-492; torch.optim.Adam
 * defaults eps=1e-8, no weight decay, no amsgrad).
 * mq_sumsq: out[0] = sum g^2 (deterministic two-stage; workspace 1024 floats). */
int mq_sumsq(const float *g, int64_t n, float *out, float *workspace, mqStream stream);
/* clip coefficient read on device: coef = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)); max_norm <= 0 disables.
 * step_host is the 1-based Adam step.  grad_scale multiplies the gradients first (1 / world size after a
 * SUM all-reduce; 1 otherwise); sumsq is the squared norm of the UNSCALED arena. */
int mq_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr, double beta1,
                 double beta2, double eps, int step_host, const float *sumsq, float max_norm, float grad_scale,
                 mqStream stream);
/* Fill of a flat gradient arena from per-parameter gradient tensors in one launch.  table (device, int64 [n_tensors][4]):
 * {source address or 0 (zeros), destination element offset (multiple of 4), elements, first block}, rows ordered by
 * first block; a block covers 2048 elements; total_blocks = sum over rows of ceil(elements / 2048). */
int mq_multi_gather(const int64_t *table_dev, int n_tensors, int64_t total_blocks, float *dst, mqStream stream);
/* The same fill from a HOST table, int64 [n_tensors][3 used of 4]: {source address or 0, destination element offset
 * (multiple of 4), elements > 0, ignored}.  The rows travel by value in the kernel arguments (120 per launch), so the
 * table is consumed before the call returns: the caller may rewrite it immediately, however far the host runs ahead of
 * the stream (what the data-parallel gradient exchange uses; train.py:140-197 makes shapes, hence addresses, vary per step). */
int mq_multi_gather_args(const int64_t *table_host, int n_tensors, float *dst, mqStream stream);

/* PROTOTYPE (tools/microbench/winofused.hip, tools/winofused_micro.py; never chosen by mq_conv_plan; compiled into the test-variant
 * library only - r5 shipped it in the product library): the 3x3 / stride-1 / pad-1 convolution
 * of the refiner's 64-channel level (preencoder.py:86-102) as F(4x5, 3x3) in ONE kernel - input transform in the MFMA's operand
 * registers, all 42 positions of a tile in one lane's accumulators, output transform lane-local; no V / M tensors in HBM.  U: the
 * transformed filters of mq_winon_weight_transform(w_krsc, K, C, 0, U, 4, 5); Ut: scratch of 42 * C * K floats; K % 16 == 0,
 * K <= 64, C % 4 == 0. */
MQ_BLOCKS_API int mq_winofused_fprop(const float *x, const float *U, const float *bias, int N, int H, int W, int C, int K, float *Ut, float *y,
                       mqStream stream);

/* ------------------------------------------------------------------ algorithm choice and planned execution (csrc/plan.cpp)
 * WHICH form a convolution layer takes is decided behind this ABI: direct implicit GEMM, or one of the reduced-multiplication
 * forms above and below - by the rule the Python package applies itself (mqgan_amd.ops calls mq_conv_plan; there is one copy of
 * the eligibility logic, in csrc/plan.cpp).  A caller asks for a plan, allocates what the plan sizes and runs the planned calls:
 *
 *   mqConvPlan p;  mq_conv_plan(&desc, leaky, NULL, &p);              // NULL: configuration from the MQGAN_* environment
 *   ws  = device buffer of max(p.ws_fprop, p.ws_backward) bytes, 256-byte aligned;  keep = p.keep_bytes bytes or NULL
 *   mq_conv_fprop_planned(&p, x, w_krsc, bias, act, slope, mask, d1, d2, d3, y, keep, ws, bytes, stream);
 *   mq_conv_backward_planned(&p, x, keep, w_krsc, w_crsk, dy, gate, ..., dx, dw, db, ws, bytes, stream);
 *
 * `leaky` = 1: the layer is conv -> LeakyReLU -> masked_fill (the discriminators, discriminators.py:219, 232; act 2 in the planned
 * forward); 0: a plain convolution (+ bias).  Forms and the layers of the reference that take them:
 *   MQ_FORM_DIRECT  everything else (mq_conv_fprop / _leaky, mq_conv_dgrad / _gated, mq_conv_wgrad / _bias)
 *   MQ_FORM_WINO2D  plain 3x3 / stride 1 / pad 1, one group, C K / (C + K) >= 32: the refiner (preencoder.py:86-130) as
 *                   F(mh x mw, 3x3), tile = fewest products for the image (4 x 5 for T x 90)
 *   MQ_FORM_S2D     leaky 5x5 / stride 2 / pad 2, >= 128 channels: patch discriminator (discriminators.py:196-242), polyphase
 *   MQ_FORM_WINO1D  leaky (3 x k), stride (1, 1 or 2), batch groups, >= 128 channels: multi-bin discriminator
 *                   (discriminators.py:292-311) as F(m, 3) along the bins; `nested`: its (1 x k) row convolutions run as
 *                   MQ_FORM_WINOW themselves when their column stride is 1
 *   MQ_FORM_WINOW   3 / 5 / 7 stride-1 taps along time, >= 128 channels: encoder / decoder 1-D convolutions
 *                   (attentions.py:95-130), the patch discriminator's 3x3 / stride-(2, 1) layers, as F(4,3) / F(4,5) / F(2,7)
 * All reduced forms: f32 tensors, MQ_PRECISION_F32 / _F32X3, C % 16 == K % 16 == 0.  keep buffer: what forward hands to
 * backward (the transformed input for the weight gradient, the CRSK transformed filters for the input gradient); without it
 * backward transforms x / w again.  mqPlanCfg: the MQGAN_WINO_* switches as values (minimum channel counts: 0 = form off). */
#define MQ_FORM_DIRECT 0
#define MQ_FORM_WINO2D 1
#define MQ_FORM_S2D 2
#define MQ_FORM_WINO1D 3
#define MQ_FORM_WINOW 4
typedef struct {
    int wino_minc;            /* MQGAN_WINO_MINC (128); with wino_minc_explicit = 0 eligibility is by benefit, C K / (C + K) */
    int wino_minc_explicit;
    int wino_tile;            /* MQGAN_WINO_TILE: 0 = auto, 3, 4, 45, 6 (F(6x6): opt-in only, twice the error of F(4x5)) */
    int s2d_minc;             /* MQGAN_WINO_S2D_MINC (128) */
    int wino1d_minc;          /* MQGAN_WINO1D_MINC (128) */
    int wino1d_tile;          /* MQGAN_WINO1D_TILE: 0 = mq_wino1d_tile(H), 2, 4, 5 */
    int winow_minc;           /* MQGAN_WINOW_MINC (128) */
    int keep_mib;             /* MQGAN_WINO_KEEP_MIB (6144): larger transformed inputs are recomputed in backward */
} mqPlanCfg;
typedef struct {
    int form;                 /* MQ_FORM_* */
    int mh, mw;               /* output tile: WINO2D / S2D rows x columns; WINO1D mh = rows per tile; WINOW mw = columns per tile */
    int nested;               /* WINO1D: row convolutions as WINOW */
    int leaky;
    int npos;                 /* transform positions */
    int64_t tiles;
    mqConvDesc desc;          /* the layer */
    mqConvDesc gemm;          /* the batch-grouped GEMM / row / column convolution the form multiplies with */
    mqConvDesc inner;         /* nested: the column form's convolution */
    size_t keep_bytes, ws_fprop, ws_backward;
    double flops;             /* 2 x multiplications of one forward pass in this form */
    int keep_filters_only;    /* set by the CALLER on its copy of the plan (mq_conv_plan leaves 0): the keep buffer handed to the planned
                               * calls holds the transformed CRSK filters only and is keep_uc_bytes large - for passes whose weights get no
                               * gradient (the generator step through the discriminators): the transformed input, 1.5x the layer's input
                               * and more, is then not carried from forward to backward; ignored where keep_uc_bytes == 0 */
    size_t keep_uc_bytes;     /* row / column forms: bytes of the transformed CRSK filters inside the keep buffer; 0 for the other forms */
} mqConvPlan;
void mq_plan_cfg_default(mqPlanCfg *cfg);
const char *mq_form_name(int form);
int mq_conv_plan(const mqConvDesc *desc_host, int leaky, const mqPlanCfg *cfg_or_null, mqConvPlan *plan);
/* y = epilogue(conv(x, w) + bias).  act 0 (plans with leaky = 0) / act 2 (leaky = 1): LeakyReLU(slope), then zero where
 * mask[(m / d1) * d2 + m % d3] (m = output pixel; mask may be NULL).  keep: NULL or plan->keep_bytes bytes. */
int mq_conv_fprop_planned(const mqConvPlan *plan, const float *x, const float *w_krsc, const float *bias, int act, float slope,
                          const uint8_t *mask, int64_t d1, int64_t d2, int64_t d3, float *y, void *keep, void *workspace,
                          size_t workspace_bytes, mqStream stream);
/* Gradients from dy = the gradient with respect to the convolution's output BEFORE its activation (for a LeakyReLU layer apply
 * LeakyReLU'(y) and the mask first, mq_pointwise_bwd, unless the consumer of y already did: mq_conv_dgrad_gated's contract).
 * dx / dw / db NULL = not wanted.  x: needed by the direct form and when keep is NULL.  w_crsk: the direct input gradient only.
 * workspace: plan->ws_backward bytes whatever subset of gradients is asked for (NULL allowed when that is 0, as in the forward).
 * gate (NULL = none): x itself when x came out of a LeakyReLU(gate_slope) + mask layer that expects the gradient with respect to
 * ITS pre-activation: dx *= x > 0 ? 1 : gate_slope, zero where gate_mask[(m / g1) * g2 + m % g3] (m = input pixel). */
int mq_conv_backward_planned(const mqConvPlan *plan, const float *x, const void *keep, const float *w_krsc, const float *w_crsk,
                             const float *dy, const float *gate, float gate_slope, const uint8_t *gate_mask, int64_t g1, int64_t g2,
                             int64_t g3, float *dx, float *dw, float *db, void *workspace, size_t workspace_bytes, mqStream stream);
/* ------------------------------------------------------------------ data-parallel exchange (SURVEY.md 8b `allreduce_bucket`, 8e)
 * The reference trains on ONE device (train.py:229-233) and has no collective; under data parallelism the gradients that
 * train.py:400-404 (discriminators) and :489-493 (generator) clip and apply are the SUMS over the ranks of the per-rank gradients
 * (the 1 / world factor goes into mq_clip_adam's grad_scale; the loss normalisers are global sums, mqgan_amd/comm.py).  RCCL is
 * resolved at run time (dlopen: a copy the process already maps - PyTorch-ROCm's own - wins, else the system librccl.so, or the
 * file MQGAN_RCCL_LIB names); the library has no link-time dependency on it and nothing here is touched by a single-GPU run.
 *   mq_comm_unique_id   rank 0 fills MQ_COMM_ID_BYTES host bytes (ncclUniqueId) and hands them to every rank over any host channel
 *   mq_comm_init        collective over the `world` ranks; the calling thread's current HIP device is the rank's GPU (one per rank)
 *   mq_allreduce_bucket in-place SUM of n floats over the ranks, asynchronous on `stream` (a side stream, so that a bucket's exchange
 *                       runs behind the rest of backward: the caller orders it after the bucket's last gradient with an event and
 *                       makes the compute stream wait for it before mq_clip_adam); one call at a time per communicator
 *   mq_comm_destroy     after the last collective has completed */
#define MQ_COMM_ID_BYTES 128
typedef struct mqComm mqComm;
int mq_comm_unique_id(void *id_host);
int mq_comm_init(mqComm **comm, int world, int rank, const void *id_host);
int mq_allreduce_bucket(mqComm *comm, float *buf, size_t n, mqStream stream);
int mq_comm_destroy(mqComm *comm);

/* DEBUG / MEASUREMENT ONLY - the one piece of process-global state in this ABI (everything else is per call and per stream):
 * two function pointers, not thread-safe, meant for ONE measuring thread (bench.py, tools/layer_profile.py) and NULL otherwise;
 * a production caller never sets them, and with both NULL the library keeps no state between calls.
 * Timing hook: called before (phase 0) and after (phase 1, with mq_last_kernel()) every GEMM launch of the planned calls, with
 * the launch's algorithmic flop count - bench.py records HIP events on the launch stream from it.  NULL switches it off. */
typedef void (*mqLaunchHook)(int phase, double flops, const char *kernel);
void mq_debug_set_launch_hook(mqLaunchHook hook);
/* The HBM-bound half of the same calls ("HBM roof for ... the fused passes"): called before (phase 0) and after
 * (phase 1) every transform launch of the reduced forms with family = "winograd_transforms", and once per planned call of a reduced
 * form with phase 2, bytes = the layer's ALGORITHMIC activation bytes for that pass (each activation tensor it reads or writes,
 * once: x + y forward; dy + dx + x backward) and family = mq_form_name(form) - what bench.py's wasted-traffic ratio divides by. */
typedef void (*mqPassHook)(int phase, double bytes, const char *family);
void mq_debug_set_pass_hook(mqPassHook hook);

#ifdef __cplusplus
}
#endif
#endif /* MQGAN_HIP_H */
