/* afd.h -- C ABI of libafd_hip.so: the MI355X (gfx950) engine under the alias-free DDPM hot path.
 *
 * The reference (MDFahimAnjum/AliasFree-Diffusion-Models-PyTorch) has no FFI: its hot path is a
 * plain Python API over torch ATen ops.  Each entry point below names the reference lines whose
 * device work it replaces (paths relative to the reference root).  The Python host package binds
 * these with ctypes; nothing here takes or returns a torch type.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 (NCHW, contiguous unless a stride is given) except
 *     where a parameter is documented as HOST (filter taps, shape tables);
 *   - `stream` is a hipStream_t passed as void*; entry points only enqueue work: they never
 *     allocate, never synchronise and are legal inside hipGraph stream capture;
 *   - workspaces are caller-provided; `*_workspace_bytes` tells how much;
 *   - return value: AFD_OK or an AFD_E* code; afd_last_error() gives the message for this thread.
 *   - batch strides are in ELEMENTS; 0 means "contiguous" (C*H*W of that tensor).
 */
#ifndef AFD_H_
#define AFD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFD_OK 0
#define AFD_EINVAL 1     /* null pointer, non-positive size, unsupported shape */
#define AFD_ELAUNCH 2    /* hip launch / runtime error */
#define AFD_MAX_TAPS 15  /* largest supported filter side N */

typedef void* afd_stream_t;

const char* afd_version(void);
const char* afd_last_error(void);
/* number of HIP devices visible (0 on a CPU-only host); never initialises a context */
int afd_device_count(void);

/* ---- F1: filter design (HOST code, no GPU) -------------------------------------- filtrs.py:20-37
 * N x N radial jinc * outer-product Kaiser(beta) window, sum-normalised, fp64 -> fp32.
 * taps_out: HOST float[N*N].  has_beta = 0 reproduces beta=None. */
int afd_lowpass_kernel(double omega_c, int N, int has_beta, double beta, float* taps_out);

/* ---- F2: custom_upsample(x, f, factor=2) ---------------------------------------- filtrs.py:79-94
 * x (B,C,H,W) -> y (B,C,2H,2W).  taps: HOST float[N*N].  *_bwd is the exact adjoint (dx from dy). */
int afd_filt_up2_fwd(const float* x, float* y, int B, int C, int H, int W, long x_bstride, long y_bstride,
                     const float* taps, int N, afd_stream_t stream);
int afd_filt_up2_bwd(const float* dy, float* dx, int B, int C, int H, int W, long dy_bstride, long dx_bstride,
                     const float* taps, int N, afd_stream_t stream);

/* ---- F3: custom_downsample(x, f, factor=2) -------------------------------------- filtrs.py:71-77
 * x (B,C,H,W) -> y (B,C,ceil(H/2),ceil(W/2)) (even rows / columns of the filtered image). */
int afd_filt_down2_fwd(const float* x, float* y, int B, int C, int H, int W, long x_bstride, long y_bstride,
                       const float* taps, int N, afd_stream_t stream);
int afd_filt_down2_bwd(const float* dy, float* dx, int B, int C, int H, int W, long dy_bstride, long dx_bstride,
                       const float* taps, int N, afd_stream_t stream);

/* ---- F4: filtered nonlinearity  y = down2(GELU_erf(up2(v)))  ------------ ddpm_utils.py:123-125,129-131
 * Optional fused prologue (any pointer may be NULL):
 *     v[b,c,:,:] = x[b,c,:,:] * (rstd[b]*gamma[c]) + (beta[c] - mean[b]*rstd[b]*gamma[c]) + res[b,c,:,:]
 * i.e. GroupNorm(1,C)-apply (ddpm_utils.py:122,127) and the residual add (:128) folded into the load.
 * stats: float[B*2] = {mean, rstd} per sample, or NULL (then gamma/beta are ignored).
 * workspace: only read when the shape is off the fused fast path (N != 3 or a non-square /
 * non-power-of-two plane); size from afd_filt_act_workspace_bytes (0 on the fast path).
 * bwd writes dv = dL/dv (same shape as x); the caller chains dv into GroupNorm-backward / the residual.
 * gn_partials (or NULL; needs stats; fused fast path only): the kernel also emits GroupNorm-backward's
 * per-plane sums of dv, layout (B,2,C) -- pass them to afd_groupnorm1_bwd with have_partials = 1. */
size_t afd_filt_act_workspace_bytes(int B, int C, int H, int W, int N, int backward);
int afd_filt_act_fwd(const float* x, float* y, int B, int C, int H, int W,
                     const float* stats, const float* gamma, const float* beta, const float* res,
                     const float* taps_up, const float* taps_down, int N,
                     void* workspace, afd_stream_t stream);
/* the forward INCLUDING the GroupNorm(1,C) statistics, for samples small enough that one workgroup holds one (C * H <= 1024
 * threads, H = W in {4, 8, 16}, N = 3: afd_filt_act_fwd_gn_supported != 0): the statistics launch disappears; stats_out
 * (B*2) receives {mean, rstd} for the backward entry points. */
size_t afd_filt_act_fwd_gn_supported(int C, int H, int W, int N);
int afd_filt_act_fwd_gn(const float* x, float* y, int B, int C, int H, int W, float eps, float* stats_out,
                        const float* gamma, const float* beta, const float* res,
                        const float* taps_up, const float* taps_down, int N, afd_stream_t stream);
/* ... and the backward that also finishes GroupNorm's backward for such samples: dx = dL/dx of the GroupNorm INPUT leaves
 * instead of dv (dres, or NULL: also write dv = the residual branch's gradient); gn_partials (B,2,C) as in afd_filt_act_bwd --
 * the caller still folds them into dgamma / dbeta (afd_colsum2).  No afd_groupnorm1_bwd call for these sites. */
int afd_filt_act_bwd_gn(const float* x, const float* dy, float* dx, float* dres, int B, int C, int H, int W, const float* stats,
                        const float* gamma, const float* beta, const float* res, const float* taps_up, const float* taps_down, int N,
                        float* gn_partials, afd_stream_t stream);
int afd_filt_act_bwd(const float* x, const float* dy, float* dv, int B, int C, int H, int W,
                     const float* stats, const float* gamma, const float* beta, const float* res,
                     const float* taps_up, const float* taps_down, int N,
                     void* workspace, float* gn_partials, afd_stream_t stream);

/* ---- F6: nn.GroupNorm(1, C), eps, affine ---------------------------- ddpm_utils.py:85,88,113,116
 * fwd: stats_out[b] = {mean, rstd}; if y != NULL:
 *     y = act( gn(x)*gamma + beta + res ) + emb[b,c]
 *   res  (B,C,H,W) or NULL : residual of DoubleConv (ddpm_utils.py:93);
 *   act  0 = identity, 1 = exact GELU (nn.GELU, :86 / F.gelu, :93);
 *   emb  (B,C) or NULL     : time-embedding add of Down/Up (:218-219, :244-245).
 * bwd: given dy (= dL/dy) recomputes the chain; writes dx, dres (may be NULL), and per-sample
 *   partials laid out (B, 2, C): [b][0][c] = sum dz*xhat (-> dgamma), [b][1][c] = sum dz (-> dbeta), which
 *   the kernel's tail reduces over b into dgamma / dbeta (both NULL: the caller reduces the partials itself, e.g. with
 *   afd_colsum; accumulate = 1 adds into them, as autograd's .grad accumulation would).  The partial buffer holds
 *   B*C*2 floats.  demb = sum_hw dy is (B,C). */
int afd_groupnorm1_fwd(const float* x, float* y, float* stats_out, int B, int C, int HW, float eps,
                       const float* gamma, const float* beta, const float* res, int act, const float* emb,
                       afd_stream_t stream);
/* test hook: 0 = the sample-resident backward (one launch, x / dy read once) wherever the sample fits the registers
 * (C*HW <= 32768, HW/4 a power of two: default), 1 = always the plane pass + apply pass of round 1 */
int afd_debug_norm_path(int mode);
int afd_groupnorm1_bwd(const float* x, const float* dy, const float* stats, int B, int C, int HW,
                       const float* gamma, const float* beta, const float* res, int act,
                       float* dx, float* dres, float* dgamma_dbeta_partial /* B*C*2 floats, (B,2,C) */, float* demb /* (B,C) or NULL */,
                       int have_partials /* 1: the partials were already produced by afd_filt_act_bwd */,
                       float* dgamma /* (C) or NULL */, float* dbeta /* (C) or NULL */, int accumulate, afd_stream_t stream);
/* out[j] (+)= sum_i in[i*cols + j], i < rows (deterministic tree; accumulate != 0 adds into out) */
int afd_colsum(const float* in, float* out, int rows, int cols, int accumulate, afd_stream_t stream);
/* same with an explicit row stride (elements): sums a column block of a wider matrix */
int afd_colsum_strided(const float* in, long row_stride, float* out, int rows, int cols, int accumulate, afd_stream_t stream);
/* (rows,2,C) partials -> out_a[C] (+)= sum over rows of block 0, out_b[C] (+)= block 1, in one launch */
int afd_colsum2(const float* in, float* out_a, float* out_b, int rows, int C, int accumulate, afd_stream_t stream);

/* ---- F5/F10: convolution as implicit GEMM (3x3 pad 1, or 1x1) --------- ddpm_utils.py:84,87,112,115;
 *      nn.Linear / MHA projections on NCHW tokens (ddpm_utils.py:59-66,71,73); outc (ddpm_models.py:84)
 * x (B,Cin,H,W), w (Cout,Cin,k,k) k in {1,3}, y (B,Cout,H,W).
 * epilogue: y = act(conv + bias[co]) + res      (bias/res may be NULL; act 0 none, 1 exact GELU)
 * dgrad:   dx = conv_transpose(dy, w)           wgrad: dw = sum_b,hw dy (x) x ; dbias = sum dy
 * wgrad needs a workspace (split-K partial slabs + the (B,Cout) dbias partials); size from
 * afd_conv_wgrad_workspace_bytes. */
/* test hook: 0 = choose the tile by workgroup count (default), 1 = always the 64x128 tile,
 * 2 = always the 32x64 in-workgroup split-K tile (when the shape allows it);
 * 8 / 9 / 10 = 1x1 streaming kernel chosen by the measured rule (default) / never / whenever the shape is covered;
 * 32 / 33 / 34 = wgrad workgroups of 4 waves / 8 waves / chosen per layer (default);
 * 96 / 97 / 98 = Winograd form of the 3x3 wgrad chosen by rule (default) / never / whenever the shape is covered;
 * 84 / 85 / 86 = bf16x3 form of the 3x3 wgrad (csrc/bf3_wgrad.hip) by rule (default) / never / whenever the shape is covered
 *                (85 also switches the 1x1 form off); 88 / 89 = bf16x3 form of the 1x1 wgrad by rule (default) / never;
 * 80 / 81 / 82 = the DIRECT matrix-core form of the 3x3 forward / dgrad (see afd_conv3x3_weight_kinds below) by the measured
 *                rule (default) / never / wherever the shape is covered;
 * 74 / 75 = the f16x2 split-K kernel for the 4x4 maps and the thin 8x8 launches (csrc/h2.hip conv_h2_sk; also a direct form:
 *           afd_conv3x3_weight_kinds reports it) by rule (default) / never (round 1's fp32 Winograd split-K kernel instead);
 *           73 = wherever the shape is covered, ahead of the tile kernel (tests);
 * 48 / 49 = workgroups per launch of the matrix-core 3x3 weight gradient: one per CU (256: fastest alone, default) / 160 (fastest
 *           beside the dependent chain on another stream: -1 % on the train step; training.TrainStep asks for it);
 * 60 / 61 = the f16x2 tile kernel's operand feed: by rule (default: both operands from LDS, the weights by LDS-DMA, where a
 *           workgroup holds one block of 32 output channels on the 32 x 32 maps; the weights straight from L2 into registers
 *           elsewhere) / always the register-fed kernel;  62 / 63 / 59 = the LDS-fed kernel wherever the shape is covered,
 *           trying its (128 pixels x 64 channels) / (256 x 32) / (128 x 32) workgroup first (tests);
 * 76 / 77 = arithmetic of that direct form: two fp16 pieces under an online power-of-two scale (csrc/h2.hip, default) /
 *           three bf16 pieces (round 2, csrc/bf3.hip);  78 / 79 = the same choice for the matrix-core 3x3 weight gradient
 *           (csrc/h2_wgrad.hip / csrc/bf3_wgrad.hip);
 * 92 / 93 = streaming vector kernels for the 1x1 output layer (<= 4 output channels) and its dgrad (csrc/ends.hip) by rule
 *           (default) / never */
int afd_debug_conv_path(int mode);
int afd_conv_fwd(const float* x, const float* w, const float* bias, const float* res, float* y,
                 int B, int Cin, int Cout, int H, int W, int ksize, int act, afd_stream_t stream);
int afd_conv_dgrad(const float* dy, const float* w, float* dx,
                   int B, int Cin, int Cout, int H, int W, int ksize, afd_stream_t stream);
size_t afd_conv_wgrad_workspace_bytes(int B, int Cin, int Cout, int H, int W, int ksize);
/* which kernel afd_conv_wgrad (dbias == NULL) runs for the shape: 0 = direct implicit GEMM on the fp32 MFMA,
 * 1 = Winograd F(3x3,2x2) on the fp32 MFMA, 2 = pixel-reduction GEMM on the bf16 MFMA with exact three-piece splits
 * (fp32 accuracy), 3 = the first layer's vector-FMA form (3 input channels: memory-bound), 4 = the pixel-reduction GEMM on
 * the fp16 MFMA with two-piece splits under online power-of-two scales (fp32-class accuracy, half the products of 2; the
 * default since round 3).  For reporting (bench.py prices each launch against the peak of the instruction it issues). */
int afd_conv_wgrad_form(int B, int Cin, int Cout, int H, int W, int ksize);
int afd_conv_wgrad(const float* x, const float* dy, float* dw, float* dbias /* or NULL */,
                   int B, int Cin, int Cout, int H, int W, int ksize, int accumulate,
                   void* workspace, afd_stream_t stream);

/* ---- the deterministic tail of every parameter gradient, batched (csrc/fold.hip) ----------------------------------
 *      backward of nn.Conv2d / nn.Linear / nn.GroupNorm / nn.LayerNorm parameters: ddpm_utils.py:59-66,84-88,112-118
 * A fold: dst[(j % inner) * rstride + j / inner] (+)= sum_{s < splits} part[s * stride + j], j < n, fixed order, no
 * atomics (identity map: inner = n, rstride = 1; [tap][plane] slabs -> OIHW: inner = n / 9, rstride = 9).
 * afd_fold_batched folds any number of them in ceil(n / 56) launches (descs: HOST array; the descriptors travel as
 * kernel arguments, so the call is legal under stream capture and needs no device table).  The result of a fold does
 * not depend on what it is batched with.
 * afd_conv_wgrad_partials = afd_conv_wgrad without its final fold: launches the slab producer and writes the fold
 * descriptor(s) (weights, then bias) to folds_out[0..*n_folds) (HOST, room for 2) for a later afd_fold_batched ON THE
 * SAME STREAM; the workspace must stay alive until that fold has run.  Shapes whose kernel has no slab stage are
 * completed at once and report *n_folds = 0. */
typedef struct afd_fold_desc {
  const float* part;     /* partial slabs */
  float* dst;
  long n;                /* elements per slab */
  long stride;           /* elements between consecutive slabs (>= n) */
  long inner, rstride;   /* element map, see above */
  int splits;            /* number of slabs */
  int accumulate;        /* != 0: add into dst */
} afd_fold_desc;         /* 56 bytes */
int afd_fold_batched(const afd_fold_desc* descs, int n, afd_stream_t stream);
int afd_conv_wgrad_partials(const float* x, const float* dy, float* dw, float* dbias /* or NULL */,
                            int B, int Cin, int Cout, int H, int W, int ksize, int accumulate,
                            void* workspace, afd_fold_desc* folds_out, int* n_folds, afd_stream_t stream);

/* Winograd F(2x2,3x3) form of the 3x3 forward / dgrad (same results up to fp32 re-association: every product
 * and sum is fp32; 16 multiplies per 2x2 output tile and channel pair instead of 36).  Covers the square
 * 64x64 ... 4x4 maps with Cin % 8 == 0 and Cout % 32 == 0 (dgrad: roles swapped).
 * afd_conv3x3_wino_workspace_bytes returns 0 when the layer is not covered (or too small to gain): the caller
 * then uses afd_conv_fwd / afd_conv_dgrad.  The workspace holds the transformed weights (16*Cin*Cout floats; the
 * forward and the dgrad forms differ); weights_ready != 0 says it still holds them from an earlier call with the
 * same w and the same pass, so the transform launch is skipped (sampling: w is constant over the 999 steps).
 * afd_debug_conv_path: 64 / 65 = Winograd chosen by the measured rule (default) / never; 66..69 = whenever covered,
 * with workgroups of 64x64 / 32x64 / 64x32 / 32x32 (output channels x tiles); 70 = the small-map (4x4, 8x8)
 * in-workgroup split-K kernel wherever it is covered. */
size_t afd_conv3x3_wino_workspace_bytes(int B, int Cin, int Cout, int H, int W, int dgrad);
/* the transformed weights of both passes (either pointer may be NULL) in ONE launch: the dgrad image is the forward
 * image with permuted transform indices.  Buffers of 16*Cin*Cout floats each; afterwards call the entry points below
 * with weights_ready = 1. */
int afd_conv3x3_wino_weights(const float* w, float* u_fwd, float* u_dgrad, int Cin, int Cout, int kinds, afd_stream_t stream);
/* Some layers run a DIRECT form on the bf16 matrix cores at fp32 accuracy instead (csrc/bf3.hip: every operand split
 * exactly into three bf16 pieces, six cross terms per product; chosen by the library per pass from the shape): their
 * workspace holds the split weights (54 bytes per (cin, cout) pair -- the same buffer size serves both forms).
 * kinds: bit 0 / bit 1 set = the forward / the dgrad image of this layer is the DIRECT form's (records of 8 matrix-core
 * inputs per (piece, tap, channel group, output row)) instead of the Winograd one; bit 2 (only with bit 0 or 1) = that direct
 * image holds three bf16 pieces (round 2's arithmetic, afd_debug_conv_path 77) instead of two fp16 pieces + one scale per
 * output row.  The value depends on B (the rule wants >= 2 workgroups per CU) and on afd_debug_conv_path: pass what this
 * function returned WHEN THE IMAGE WAS BUILT to the two weight entry points (afd_wino_desc.kinds for the batched one) and to
 * afd_conv3x3_wino_fwd / _dgrad.  afd_debug_conv_path 80 / 81 / 82 = the direct form by the measured rule (default) / never /
 * wherever the shape is covered. */
int afd_conv3x3_weight_kinds(int B, int Cin, int Cout, int H, int W);
/* ... and of EVERY layer of a model in one launch (a train step transforms ~30 weight tensors; one launch instead of
 * 30 takes them off the critical path).  descs (DEVICE array): one entry per layer; wg_desc (DEVICE, n_wg ints): the
 * layer each 256-thread workgroup works on -- layer i owns workgroups [first_wg, first_wg + ceil(Cin*Cout/256)). */
typedef struct afd_wino_desc {
  const float* w;        /* (Cout,Cin,3,3) */
  float* u_fwd;          /* 16*Cin*Cout floats or NULL */
  float* u_dgrad;        /* 16*Cin*Cout floats or NULL */
  int Cin, Cout, first_wg, kinds;   /* kinds: afd_conv3x3_weight_kinds of the layer */
} afd_wino_desc;
int afd_conv3x3_wino_weights_batched(const afd_wino_desc* descs, const int* wg_desc, int n_wg, afd_stream_t stream);
/* weights_ready != 0: the workspace already holds the image, built for `kinds` (what afd_conv3x3_weight_kinds returned when
 * it was built); the call fails with AFD_EINVAL if that is not the form it is about to read (the choice between the forms
 * depends on the batch size -- e.g. a last partial batch -- and on afd_debug_conv_path).  weights_ready == 0: the image is
 * built first, in the form this call reads; `kinds` is ignored. */
int afd_conv3x3_wino_fwd(const float* x, const float* w, const float* bias, const float* res, float* y,
                         int B, int Cin, int Cout, int H, int W, int act, void* workspace, int weights_ready, int kinds,
                         afd_stream_t stream);
int afd_conv3x3_wino_dgrad(const float* dy, const float* w, float* dx, const float* add_to_dx /* or NULL: dx = dgrad + add,
                           the gradient that reaches x through the block's residual branch */,
                           int B, int Cin, int Cout, int H, int W, void* workspace, int weights_ready, int kinds,
                           afd_stream_t stream);

/* ---- F10: LayerNorm over channels of an NCHW tensor (= nn.LayerNorm([C]) on (B,L,C) tokens) ------
 * ddpm_utils.py:60,62,70.  stats_out (B,HW,2) = {mean, rstd}. */
int afd_layernorm_c_fwd(const float* x, float* y, float* stats_out, int B, int C, int HW, float eps,
                        const float* gamma, const float* beta, afd_stream_t stream);
int afd_layernorm_c_bwd(const float* x, const float* dy, const float* stats, int B, int C, int HW,
                        const float* gamma, float* dx, const float* add_to_dx /* or NULL: dx = LN'(dy) + add (the gradient
                        that reaches x through the block's residual branch: one pass instead of an extra add) */,
                        float* dgamma_dbeta_partial /* (B,2,C) */,
                        float* dgamma /* (C) or NULL */, float* dbeta /* (C) or NULL */, int accumulate,
                        afd_stream_t stream);
/* the two halves separately: afd_layernorm_c_bwd with dgamma_dbeta_partial = dgamma = dbeta = NULL writes dx only (the
 * critical path of backward); this one computes only the parameter gradients (nothing downstream waits for them, so
 * the host may issue it on another stream) */
int afd_layernorm_c_bwd_params(const float* x, const float* dy, const float* stats, int B, int C, int HW,
                               float* dgamma_dbeta_partial /* (B,2,C) */, float* dgamma, float* dbeta, int accumulate,
                               afd_stream_t stream);
/* the plane pass of afd_layernorm_c_bwd_params alone: part (B,2,C) = per-sample sums over the pixels of dy*xhat and dy;
 * the caller folds them (afd_fold_batched: two descriptors, n = C, stride = 2C, splits = B) */
int afd_layernorm_c_bwd_partials(const float* x, const float* dy, const float* stats, int B, int C, int HW, float* part,
                                 afd_stream_t stream);

/* ---- F10: multi-head self-attention core (softmax(QK^T/sqrt(d))V), flash-style ------------------
 * ddpm_utils.py:71 (nn.MultiheadAttention, batch_first, 4 heads).  qkv (B,3C,L): channel
 * n = {0:q,1:k,2:v}*C + head*d + j, token index contiguous (the NCHW image of in_proj's output).
 * o (B,C,L); lse (B,heads,L) saved for backward.  Never materialises the L x L scores. */
/* tuning hook: force R rows per lane (1, 2, 4) of the all-vector kernels for head dim 8; 0 = default;
 * 10 / 11 = the MFMA two-pass backward for head dim 8 only below L = 1024 / at every L (default) */
int afd_debug_attn_rows(int rows);
int afd_attn_fwd(const float* qkv, float* o, float* lse, int B, int heads, int d, int L, afd_stream_t stream);
int afd_attn_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
                 float* delta_workspace /* (B,heads,L) floats */, int B, int heads, int d, int L, afd_stream_t stream);

/* ---- F10: the token-wise chains of SelfAttention fused around the core ------------------- ddpm_utils.py:68-74
 * Tokens are the pixels of the NCHW tensor (P = H*W per image), every nn.Linear is a 1x1 convolution with its (out,in)
 * weight, so the block is   x -> [head] -> qkv -> afd_attn_fwd -> att -> [tail] -> out   with
 *   head: h = LayerNorm(x; gamma, beta) (:70);  qkv = w_in h + b_in  (the MHA in-projection, :71);
 *   tail: a = w_o att + b_o + x (:71-72);  f = LayerNorm(a) ;  u = w_1 f + b_1;  g = GELU(u);  out = w_2 g + b_2 + a  (:73).
 * Covered: C in {32, 64, 128} (afd_tok_supported); other widths use the unfused entry points above.
 * head_fwd: h_out / stats_out (B,P,2 = {mean, rstd}) may be NULL (inference).
 * tail_fwd: a_out, stats_out, f_out, u_out, g_out are the tensors backward needs -- all five or none (NULL = inference).
 * tail_bwd: du = (w_2^T d_out) * GELU'(u);  df = w_1^T du;  d_a = LayerNorm'(df; a) + d_out;  d_att = w_o^T d_a.
 *           du / df / d_a are also what the weight-gradient kernels and afd_layernorm_c_bwd_params consume
 *           (dW_2 from (g, d_out), dW_1 from (f, du), dW_o from (att, d_a), LayerNorm parameters from (a, df)).
 * head_bwd: dh = w_in^T dqkv;  dx = LayerNorm'(dh; x) + d_res   (d_res = d_a of the tail: the residual branch);
 *           dh_out (or NULL) for the LayerNorm parameter gradients. */
int afd_tok_supported(int C);
/* test hook: cap the workgroups per launch of the four kernels below (forces several passes per workgroup); 0 = default */
int afd_debug_tok_grid(int max_workgroups);
/* test hook: 0 = kernel form by the rule, 1 = never the wide (one wave per 32-channel block) forms, 2 = wide wherever they exist */
int afd_debug_tok_path(int mode);
int afd_tok_head_fwd(const float* x, const float* gamma, const float* beta, const float* w_in, const float* b_in,
                     float* h_out, float* stats_out, float* qkv, int B, int C, int P, float eps, afd_stream_t stream);
int afd_tok_tail_fwd(const float* att, const float* x, const float* w_o, const float* b_o, const float* gamma, const float* beta,
                     const float* w_1, const float* b_1, const float* w_2, const float* b_2,
                     float* a_out, float* stats_out, float* f_out, float* u_out, float* g_out, float* out,
                     int B, int C, int P, float eps, afd_stream_t stream);
int afd_tok_tail_bwd(const float* d_out, const float* u, const float* a, const float* stats, const float* gamma,
                     const float* w_2, const float* w_1, const float* w_o,
                     float* du_out, float* df_out, float* da_out, float* datt_out, int B, int C, int P, afd_stream_t stream);
int afd_tok_head_bwd(const float* dqkv, const float* x, const float* stats, const float* gamma, const float* w_in,
                     const float* d_res, float* dh_out, float* dx_out, int B, int C, int P, afd_stream_t stream);

/* ---- elementwise / pooling used by variants 0 and 2 --------------------------------------------
 * gelu: nn.GELU (ddpm_utils.py:64); maxpool: nn.MaxPool2d(2) (:203,258);
 * bilinear: nn.Upsample(scale_factor=2, 'bilinear', align_corners=True) (:226,280). */
int afd_gelu_fwd(const float* x, float* y, long n, afd_stream_t stream);
int afd_gelu_bwd(const float* x, const float* dy, float* dx, long n, afd_stream_t stream);
int afd_maxpool2_fwd(const float* x, float* y, int B, int C, int H, int W, afd_stream_t stream);
int afd_maxpool2_bwd(const float* x, const float* dy, float* dx, int B, int C, int H, int W, afd_stream_t stream);
int afd_bilinear_up2_fwd(const float* x, float* y, int B, int C, int H, int W, long y_bstride, afd_stream_t stream);
int afd_bilinear_up2_bwd(const float* dy, float* dx, int B, int C, int H, int W, long dy_bstride, afd_stream_t stream);
/* strided batch copy: dst[b, 0:n] = src[b, 0:n] (torch.cat of the skip, ddpm_utils.py:242) */
int afd_copy_batched(const float* src, float* dst, int B, long n, long src_bstride, long dst_bstride, afd_stream_t stream);
/* y = a + b (gradient accumulation of skip connections) */
int afd_add(const float* a, const float* b, float* y, long n, afd_stream_t stream);

/* ---- F9: time embedding ------------------------------------- ddpm_models.py:261-269,272-273
 * temb[b, k] = sin(t[b]*inv_freq[k]), temb[b, half+k] = cos(...); t int64, inv_freq (half,) fp32
 * (computed once on the host exactly as the reference does).
 * silu_linear: out[b, n] = sum_k silu(temb[b,k]) * w[n,k] + bias[n]      ddpm_utils.py:208-214 */
int afd_pos_encoding(const int64_t* t, const float* inv_freq, float* temb, int B, int half, afd_stream_t stream);
int afd_silu_linear_fwd(const float* temb, const float* w, const float* bias, float* out,
                        int B, int K, int N, afd_stream_t stream);
int afd_silu_linear_bwd(const float* temb, const float* w, const float* dout, float* dw, float* dbias,
                        float* dtemb /* or NULL; accumulated into */, int B, int K, int N, int accumulate, afd_stream_t stream);
/* the forward of up to 8 such layers that share temb in ONE launch (the six stages' emb_layer of a UNet forward: their
 * input exists as soon as the forward starts).  descs: HOST array of n entries. */
typedef struct afd_silu_desc {
  const float* w;        /* (N, K) */
  const float* bias;     /* (N) or NULL */
  float* out;            /* (B, N) */
  int N;
} afd_silu_desc;
int afd_silu_linear_fwd_batched(const float* temb, const afd_silu_desc* descs, int n, int B, int K, afd_stream_t stream);
/* out_i[b][:] = table_i[idx[b]][:] (idx clamped to [0, rows)) for up to 8 tables (descs[i].w = table_i (rows, N_i),
 * descs[i].out = (B, N_i), bias unused) in one launch.  Sampling: emb_layer(pos_encoding(t)) depends on the integer t only,
 * so the host package tabulates it for every timestep once per trajectory (with the two entry points above) and a denoise
 * step gathers its rows -- bit-identical to computing them (ddpm_models.py:261-269, ddpm_utils.py:208-214). */
int afd_gather_rows_batched(const int64_t* idx, const afd_silu_desc* descs, int n, int B, int rows, afd_stream_t stream);

/* label conditioning: out[b, :] = temb[b, :] + table[y[b], :]   (nn.Embedding lookup + add, ddpm_models.py:254,276-277)
 * y: (B,) int64 class indices in [0, num_classes); out may alias temb.
 * bwd: dtable[k, :] (+)= sum over {b : y[b] == k} of dout[b, :], rows summed in batch order (deterministic, no atomics);
 * rows of classes absent from y are zeroed (accumulate = 0) or left alone (accumulate = 1). */
int afd_embed_add_fwd(const float* temb, const float* table, const int64_t* y, float* out, int B, int D, int num_classes,
                      afd_stream_t stream);
int afd_embed_add_bwd(const float* dout, const int64_t* y, float* dtable, int B, int D, int num_classes, int accumulate,
                      afd_stream_t stream);

/* ---- F14/F16: DDPM noise / denoise / quantise ------------------ ddpm_models.py:317-321, 367-374, 381-385
 * Bit-exact restatements of the reference's fp32 expression order (no FMA contraction).
 * t: (B,) int64 indices into the (T,) schedule tables. */
int afd_noise_images(const float* x, const float* eps, const int64_t* t, const float* alpha_hat,
                     float* x_t, int B, long per_sample, afd_stream_t stream);
int afd_denoise_step(const float* x, const float* eps_pred, const float* noise /* NULL => zeros (i == 1) */,
                     const float* alpha, const float* alpha_hat, const float* beta, int i,
                     float* x_out, long n, afd_stream_t stream);
/* same update with the step index read from device memory (t_dev[0]): the form a captured hipGraph replays */
int afd_denoise_step_dev(const float* x, const float* eps_pred, const float* noise,
                         const float* alpha, const float* alpha_hat, const float* beta, const int64_t* t_dev,
                         float* x_out, long n, afd_stream_t stream);
int afd_quantize_u8(const float* x, uint8_t* out, long n, afd_stream_t stream);

/* ---- F17 (Config E): scipy.ndimage.rotate(order=3, mode='grid-wrap', prefilter=True) per (H,W) plane ----
 * ddpm_models.py:421-429.  in_coord = M @ out_coord + offset (row, col); matrix4 / offset2 are HOST doubles built
 * exactly as scipy.ndimage.rotate builds them; fp64 prefilter + interpolation, one rounding to fp32.
 * workspace: afd_rotate_workspace_bytes(planes, H, W). */
size_t afd_rotate_workspace_bytes(long planes, int H, int W);
int afd_affine_spline3_wrap(const float* x, float* y, long planes, int H, int W, const double* matrix4, const double* offset2,
                            void* workspace, afd_stream_t stream);

/* ---- F15: loss + optimiser ------------------------------------------- ddpm_utils.py:489-490,503-507
 * mse: loss_out[0] = mean((pred-target)^2) (deterministic two-stage reduction; workspace >= 4096 floats);
 * mse_bwd: dpred = 2*(pred-target)/n * dloss[0].
 * adamw: torch.optim.AdamW(lr, betas, eps, weight_decay) over flat fp32 buffers; `state` is
 * device float[4] {step, bias_corr1, bias_corr2, _} advanced by afd_adamw_tick (graph-replay safe). */
int afd_mse_fwd(const float* pred, const float* target, float* loss_out, float* workspace, long n, afd_stream_t stream);
int afd_mse_bwd(const float* pred, const float* target, const float* dloss, float* dpred, long n, afd_stream_t stream);
int afd_adamw_tick(float* state, float beta1, float beta2, afd_stream_t stream);
int afd_adamw_step(float* p, const float* g, float* m, float* v, long n, const float* state,
                   float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                   afd_stream_t stream);

/* ---- two-lane replay of a captured step (csrc/replay.hip) --------------------------------- training.TrainStep(graph="lanes")
 * A step captured by the host framework as a hipGraph (forward, backward with the weight gradients forked to a side stream,
 * AdamW) is re-issued from a C++ loop on TWO REAL STREAMS: afd_replay_build walks the graph once (kernel / memset / flat
 * memcpy nodes; no-op and event nodes are folded into the dependencies), puts the weight-gradient kernels (by name) on the side
 * lane and everything else on the main lane, and turns the dependencies that cross lanes into event record / wait pairs;
 * afd_replay_run launches the nodes in capture order, each on its lane's stream.  The graph -- and the memory its nodes
 * point into -- must outlive the handle.  counts (may be NULL) receives {work nodes, main lane, side lane, cross-lane waits}.
 * Replaces nothing in the reference (its loop is eager PyTorch, ddpm_utils.py:494-509): it removes the interpreter from the
 * steady-state step without the cross-branch cost of a hipGraph launch. */
int afd_replay_build(void* hip_graph, void** out_handle, int* counts);
int afd_replay_run(void* handle, afd_stream_t main_stream, afd_stream_t side_stream);
int afd_replay_free(void* handle);

#ifdef __cplusplus
}
#endif
#endif /* AFD_H_ */
