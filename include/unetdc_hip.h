/* libunetdc_hip.so -- C ABI of the MI355X (gfx950) U-Net / U-Net-DC forward+backward kernels.
 *
 * This is the drop-in boundary for the hot path of malani86/unet-DC-segmentation.  The reference
 * has no FFI of its own: its boundary is the nn.Module contract of UNetDC / UNet plus autograd
 * (SURVEY.md section 8b), and every operator below replaces the ATen call the reference makes at
 * the cited line of /root/reference.  Conventions:
 *
 *   - plain C: raw DEVICE pointers, explicit sizes/strides, a hipStream_t passed as void*;
 *     no torch types, no C++ exceptions.  Every function returns 0 on success, a negative
 *     UNETDC_E* code otherwise; unetdc_last_error() returns a thread-local message.
 *   - the library never allocates or frees device memory: activations, packed weights and
 *     workspaces are owned by the caller (PyTorch's caching allocator in the Python host layer).
 *   - activations are NHWC ("pixel-major") in the compute type `dtype` (UNETDC_F32 / UNETDC_BF16):
 *     element (n,y,x,c) at ((n*H + y)*W + x)*ld + c.  `ld` (in elements, multiple of 16 bytes)
 *     lets a tensor be a channel slice of a wider buffer, which is how torch.cat
 *     (models/model_2.py:68,71,74,77) costs zero bytes: the up-convolution writes channels
 *     [0,C) and the encoder skip writes channels [C,2C) of one [pixels][2C] buffer.
 *   - the network input (NCHW fp32, train_DC_focal.py:250) and output (NCHW fp32 probabilities,
 *     model_2.py:80) keep the reference's layout; parameters and their gradients are fp32 in
 *     PyTorch layout (Conv2d [Cout][Cin][3][3], ConvTranspose2d [Cin][Cout][2][2]).
 *   - kernels are enqueued on the given stream and are re-entrant; reductions are two-stage with
 *     a fixed order (no float atomics), so results are bitwise reproducible.
 */
#ifndef UNETDC_HIP_H
#define UNETDC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNETDC_ABI_VERSION 2

#define UNETDC_F32 0
#define UNETDC_BF16 1

#define UNETDC_OK 0
#define UNETDC_EINVAL (-1)      /* bad argument (shape, alignment, null pointer) */
#define UNETDC_ELAUNCH (-2)     /* HIP reported an error for a launch */
#define UNETDC_EWORKSPACE (-3)  /* caller-provided workspace too small */

typedef void* unetdc_stream_t; /* hipStream_t */

int unetdc_version(void);
const char* unetdc_last_error(void);
/* symbol (as rocprofv3 prints it) of the matrix-core kernel the last conv/convT/wgrad call dispatched;
 * the library picks the kernel per layer shape, profiling tools use this to attribute time. */
const char* unetdc_last_kernel(void);

/* ---- weight packing (derived caches of the fp32 parameters; redo after optimizer.step()) ------
 * conv3x3:  w [Cout][Cin][3][3] -> w_fwd [9][Cout][Cin], w_dgrad [9][Cin][Cout] (taps flipped)
 * convT2x2: w [Cin][Cout][2][2] -> w_fwd [4*Cout][Cin] (row (a*2+b)*Cout+co), w_dgrad [4][Cin][Cout]
 * w_dgrad may be NULL (inference). */
int unetdc_pack_conv3x3(const float* w, void* w_fwd, void* w_dgrad, int cout, int cin, int dtype, unetdc_stream_t s);
int unetdc_pack_convT2x2(const float* w, void* w_fwd, void* w_dgrad, int cin, int cout, int dtype, unetdc_stream_t s);
/* Every layer in ONE launch (LDS-tiled transpose).  `table_dev` is a DEVICE array of n descriptors
 * sorted by `begin` = running sum of 32x32 channel tiles ((a/32)*(b/32) per tensor; a, b multiples
 * of 32); kind 0 = conv3x3 with (a,b) = (cout,cin), kind 1 = convT2x2 with (a,b) = (cin,cout);
 * `total_tiles` = the grand total.  Same images as the two calls above. */
typedef struct unetdc_pack_desc {
  const float* w;
  void* w_fwd;
  void* w_dgrad; /* nullable */
  int64_t begin;
  int32_t a, b;
  int32_t kind;
  int32_t pad;
} unetdc_pack_desc;
int unetdc_pack_many(const unetdc_pack_desc* table_dev, int n, int64_t total_tiles, int dtype, unetdc_stream_t s);

/* ---- optimizer step ---------------------------------------------------------------------------
 * Replaces `optimizer.step()` of torch.optim.Adam (/root/reference/train_DC_focal.py:224,255; train.py:125) AND the
 * re-pack above, in one launch: for every parameter   g <- grad_scale * g;   m <- m + (1-beta1)(g - m);
 * v <- beta2 v + (1-beta2) g g;   p <- p - lr/(1-beta1^step) * m / (sqrt(v)/sqrt(1-beta2^step) + eps)
 * (torch's _fused_adam_ formulation; no weight decay / amsgrad, which the reference does not use), and the packed
 * images of the conv / convT weights are rewritten from the NEW p.
 * `table_dev`: DEVICE array of n descriptors sorted by `begin` (first workgroup of the tensor); a packed tensor takes
 * (a/32)*(b/32) workgroups, a plain one ceil(numel/4096); `total_blocks` = the grand total.  The gradient of tensor i is
 * flat_grad[g_off .. g_off + numel): the flat fp32 buffer the backward kernels write (parameters() order).
 * wf / wd may be NULL for a packed-kind tensor (then only p, m, v are updated).  `step` counts from 1. */
typedef struct unetdc_adam_desc {
  float* p; float* m; float* v;   /* fp32 master parameter, exp_avg, exp_avg_sq (same layout as p) */
  void* wf; void* wd;             /* packed images in `dtype`, or NULL */
  int64_t g_off, begin, numel;
  int32_t a, b;                   /* kind 0: (cout, cin); kind 1: (cin, cout); kind 2: unused */
  int32_t kind;                   /* 0 conv3x3 (packed), 1 convT2x2 (packed), 2 plain */
  int32_t pad;
} unetdc_adam_desc;
int unetdc_adam_step(const unetdc_adam_desc* table_dev, int n, int64_t total_blocks, const float* flat_grad, double lr,
                     double beta1, double beta2, double eps, int64_t step, double grad_scale, int dtype, unetdc_stream_t s);

/* ---- dilated 3x3 convolution, padding = dilation: nn.Conv2d at models/model_2.py:41-44,48-51 ---
 * y = conv(x) + bias                                  (scale == NULL; training: raw pre-BN output)
 * y = relu(conv(x)*scale + shift)                     (scale != NULL; eval: BN folded, bias inside shift)
 * stats_part (nullable, training): per-block partial [rows][2][Cout] sums / sums of squares of y
 * as stored, rows = unetdc_conv3x3_stats_rows(); the buffer must hold rows+64 rows.
 * Cin must be a multiple of 64 (bf16) / 32 (fp32) and Cout of 64: every layer but enc1.0. */
int unetdc_conv3x3_stats_rows(int64_t npixels, int cout);
/* stats_rows (out, nullable; written when stats_part is given): rows of the statistics buffer that carry data
 * (<= unetdc_conv3x3_stats_rows(): the persistent kernels write one row per workgroup and zeros into the rest, so
 * summing all rows stays valid; passing this count to unetdc_bn_finalize saves it reading the zero rows). */
int unetdc_conv3x3_fwd(const void* x, int ldx, const void* w_fwd, const float* bias, const float* scale,
                       const float* shift, void* y, int ldy, float* stats_part, int* stats_rows, int n, int h, int w,
                       int cin, int cout, int dilation, int dtype, unetdc_stream_t s);
/* "bnin" forms (round 3): the convolution / weight gradient are fed from the RAW conv output of the stage in front of them and
 * apply that stage's BatchNorm + ReLU -- relu(in_scale * x + in_shift), models/model_2.py:45-46, rounded through the storage
 * type like a stored activation -- once per staged tile in LDS: the stand-alone unetdc_bn_relu_apply pass of that stage
 * disappears; outputs are bit-identical to the two-pass form.  bf16 only, shapes the persistent lattice kernel takes.
 * unetdc_conv3x3_bnin_supported() returns 0 (no), 1 (64-channel output blocks: forward + unetdc_conv3x3_wgrad_bnin, the activation
 * tensor is never stored) or 2 (round 5, 128-channel output blocks: the forward ALSO stores the normalised activation into
 * act_out [n*h*w][ldact] on request -- the bytes the stand-alone pass would have written, without its read of x and without its
 * launch -- so that any weight-gradient kernel can follow; act_out must be NULL where the answer is 1).
 * fwd: statistics mode only (training); in_scale / in_shift [cin] are the producing stage's batch scale / shift. */
int unetdc_conv3x3_bnin_supported(int n, int h, int w, int cin, int cout, int dilation, int dtype);
int unetdc_conv3x3_fwd_bnin(const void* x_raw, int ldx, const float* in_scale, const float* in_shift, const void* w_fwd,
                            const float* bias, void* y, int ldy, float* stats_part, int* stats_rows, void* act_out, int ldact,
                            int n, int h, int w, int cin, int cout, int dilation, int dtype, unetdc_stream_t s);
int unetdc_conv3x3_wgrad_bnin(const void* x_raw, int ldx, const float* in_scale, const float* in_shift, const void* dy,
                              int lddy, float* dw, void* workspace, int64_t workspace_bytes, int n, int h, int w, int cin,
                              int cout, int dilation, int dtype, unetdc_stream_t s);
/* dx = conv_transpose(dy): autograd of the above w.r.t. its input (dx has cin channels). */
int unetdc_conv3x3_dgrad(const void* dy, int lddy, const void* w_dgrad, void* dx, int lddx, int n, int h, int w,
                         int cin, int cout, int dilation, int dtype, unetdc_stream_t s);
/* dw [Cout][Cin][3][3] fp32 = sum over pixels; workspace >= unetdc_conv3x3_wgrad_workspace() bytes. */
int64_t unetdc_conv3x3_wgrad_workspace(int n, int h, int w, int cin, int cout, int dtype);
int unetdc_conv3x3_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, void* workspace,
                         int64_t workspace_bytes, int n, int h, int w, int cin, int cout, int dilation, int dtype,
                         unetdc_stream_t s);

/* ---- first encoder convolution (small Cin, reads the NCHW fp32 image): model_2.py:10 (enc1.0) ---
 * w is the fp32 PyTorch-layout parameter itself.  Same scale/shift/stats semantics as above. */
int unetdc_conv3x3_first_stats_rows(int64_t npixels, int cin, int cout);
int unetdc_conv3x3_first_fwd(const float* x_nchw, const float* w, const float* bias, const float* scale,
                             const float* shift, void* y, int ldy, float* stats_part, int n, int h, int wd, int cin,
                             int cout, int dilation, int dtype, unetdc_stream_t s);
int64_t unetdc_conv3x3_first_wgrad_workspace(int n, int h, int w, int cin, int cout);
int unetdc_conv3x3_first_wgrad(const float* x_nchw, const void* dy, int lddy, float* dw, void* workspace,
                               int64_t workspace_bytes, int n, int h, int w, int cin, int cout, int dilation,
                               int dtype, unetdc_stream_t s);
/* The first layer's weight gradient with the BatchNorm + ReLU backward of its stage applied ON LOAD (models/model_2.py:41-46
 * under autograd): dz = gradient of the stage's activated output, y = its saved conv output, coeffs [3][cout] from
 * unetdc_bn_relu_bwd_coeffs (which also finalises dgamma / dbeta / dbias from the partial sums a fused dgrad epilogue left:
 * unetdc_conv3x3_dgrad_bnstats).  dy = k1 * [a > 0] * dz - k2 - k3 * xhat is formed per loaded chunk, rounded through the
 * storage type -- bit-identical to unetdc_bn_relu_bwd + unetdc_conv3x3_first_wgrad, without the pass that writes dy and the
 * one that reads it.  Only when nothing else needs dy (no dL/dx) and for the shapes _supported answers 1 (one input channel,
 * dilation 1, w % 8 == 0). */
int unetdc_conv3x3_first_wgrad_bn_supported(int n, int h, int w, int cin, int cout, int dilation, int dtype);
int unetdc_bn_relu_bwd_coeffs(const float* pre_parts, int pre_nparts, const float* gamma, const float* rstd, float* dgamma,
                              float* dbeta, float* dbias, float* coeffs, int n, int h, int w, int c, unetdc_stream_t s);
int unetdc_conv3x3_first_wgrad_bn(const float* x_nchw, const void* dz, int lddz, const void* y, int ldy, const float* scale,
                                  const float* shift, const float* mean, const float* rstd, const float* coeffs, float* dw,
                                  void* workspace, int64_t workspace_bytes, int n, int h, int w, int cin, int cout,
                                  int dilation, int dtype, unetdc_stream_t s);
/* Gradient with respect to the INPUT image, dL/dx of the module's forward (autograd of the first nn.Conv2d, models/model_2.py:
 * 10,41-44,58; the reference's loops never request it, a saliency-style caller does): dx NCHW fp32 [n][cin][h][w] from the
 * gradient dy [n*h*w][lddy] of the first convolution's output and its fp32 weights in PyTorch layout [cout][cin][3][3]. */
int unetdc_conv3x3_first_dgrad(const void* dy, int lddy, const float* w, float* dx_nchw, int n, int h, int wd, int cin,
                               int cout, int dilation, int dtype, unetdc_stream_t s);

/* ---- ConvTranspose2d(k=2, s=2): models/model_2.py:20,23,26,29 and :67,70,73,76 -----------------
 * fwd: x [n,h,w,cin] -> up [n,2h,2w,cout] (+bias) written with pixel stride ldup (concat slice). */
int unetdc_convT2x2_fwd(const void* x, int ldx, const void* w_fwd, const float* bias, void* up, int ldup, int n,
                        int h, int w, int cin, int cout, int dtype, unetdc_stream_t s);
int unetdc_convT2x2_dgrad(const void* dup, int lddup, const void* w_dgrad, void* dx, int lddx, int n, int h, int w,
                          int cin, int cout, int dtype, unetdc_stream_t s);
int64_t unetdc_convT2x2_wgrad_workspace(int n, int h, int w, int cin, int cout, int dtype);
/* dw [Cin][Cout][2][2] fp32 (the bias gradient is unetdc_channel_sum of dup). */
int unetdc_convT2x2_wgrad(const void* x, int ldx, const void* dup, int lddup, float* dw, void* workspace,
                          int64_t workspace_bytes, int n, int h, int w, int cin, int cout, int dtype,
                          unetdc_stream_t s);

/* ---- BatchNorm2d + ReLU (+ max_pool2d): models/model_2.py:45-46,52-53 and :59-61,64 -------------
 * bn_finalize: batch statistics from the conv's partials -> scale = gamma*rstd,
 *   shift = beta - mean*scale, saved mean/rstd; running stats updated with momentum (unbiased var)
 *   when running_mean != NULL. `count` = n*h*w. */
int unetdc_bn_finalize(const float* stats_part, int rows, int64_t count, const float* gamma, const float* beta,
                       float eps, float momentum, float* running_mean, float* running_var, float* scale,
                       float* shift, float* mean, float* rstd, int c, unetdc_stream_t s);
/* eval: scale = gamma/sqrt(running_var+eps), shift = beta + (conv_bias - running_mean)*scale */
int unetdc_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                          const float* running_var, const float* conv_bias, float eps, float* scale, float* shift,
                          int c, unetdc_stream_t s);
/* a = relu(scale*y + shift) (a = y when scale == NULL); pooled (nullable) = max_pool2d(a, 2).
 * `a` may be NULL when only the pooled tensor is wanted and scale == NULL. */
int unetdc_bn_relu_apply(const void* y, int ldy, const float* scale, const float* shift, void* a, int lda,
                         void* pooled, int ldp, int n, int h, int w, int c, int dtype, unetdc_stream_t s);
/* backward of conv-output y -> BN(train) -> ReLU (-> skip and/or 2x2 max-pool consumers):
 *   incoming gradient = dskip (nullable, full resolution) + scatter(dpool) (nullable, half
 *   resolution, routed to the window arg-max recomputed from y); outputs dy (gradient of the conv
 *   output), dgamma, dbeta and the conv-bias gradient dbias (nullable).
 *   pre_parts (nullable, non-pooled form only): [pre_nparts][3][c] partial sums already produced by
 *   the *_dgrad_bnstats call that wrote dskip -- the reduction pass over (dskip, y) is then skipped.
 *   The buffer must have 64 spare rows after pre_nparts. */
int64_t unetdc_bn_relu_bwd_workspace(int n, int h, int w, int c, int pooled, int dtype);
int unetdc_bn_relu_bwd(const void* dskip, int ldskip, const void* dpool, int ldpool, const void* y, int ldy,
                       const float* scale, const float* shift, const float* mean, const float* rstd,
                       const float* gamma, void* dy, int lddy, float* dgamma, float* dbeta, float* dbias,
                       void* workspace, int64_t workspace_bytes, const float* pre_parts, int pre_nparts, int n, int h,
                       int w, int c, int dtype, unetdc_stream_t s);
/* unetdc_bn_relu_bwd for the stage in front of a ONE-channel head (dec1's second stage, models/model_2.py:76-80): the
 * gradient of the head's input is dz * w[c] with dz = dprobs * p * (1 - p) per pixel, so it is recomputed from the fp32
 * [N, 1, H, W] tensors instead of being written by unetdc_head_bwd_bnstats (call that with da = NULL) and read back -- two
 * activation-sized transfers less per step; rounded through the storage type, i.e. bit-identical to the stored form.
 * pre_parts / pre_nparts: the sums unetdc_head_bwd_bnstats produced (required). */
int unetdc_bn_relu_bwd_head(const float* dprobs, const float* probs, const float* head_w, const void* y, int ldy,
                            const float* scale, const float* shift, const float* mean, const float* rstd,
                            const float* gamma, void* dy, int lddy, float* dgamma, float* dbeta, float* dbias,
                            void* workspace, int64_t workspace_bytes, const float* pre_parts, int pre_nparts, int n, int h,
                            int w, int c, int dtype, unetdc_stream_t s);
/* BatchNorm with FROZEN statistics under autograd -- model.eval() with gradients enabled, i.e. fine-tuning with fixed
 * running statistics, which the reference module supports through plain autograd (nn.BatchNorm2d in eval mode,
 * models/model_2.py:45,52): unetdc_bn_frozen_affine fills scale / shift / mean / rstd from the running buffers (the conv bias
 * is added by the conv epilogue as in training), the forward then uses the training-path kernels; unetdc_bn_relu_bwd_frozen is
 * unetdc_bn_relu_bwd for that case: dy = gamma * rstd * dyhat (no batch terms), dgamma = sum dyhat * xhat, dbeta = sum dyhat,
 * dbias = sum dy. */
int unetdc_bn_frozen_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                            float eps, float* scale, float* shift, float* mean, float* rstd, int c, unetdc_stream_t s);
int unetdc_bn_relu_bwd_frozen(const void* dskip, int ldskip, const void* dpool, int ldpool, const void* y, int ldy,
                              const float* scale, const float* shift, const float* mean, const float* rstd,
                              const float* gamma, void* dy, int lddy, float* dgamma, float* dbeta, float* dbias,
                              void* workspace, int64_t workspace_bytes, const float* pre_parts, int pre_nparts, int n, int h,
                              int w, int c, int dtype, unetdc_stream_t s);
/* dgrad fused with the BatchNorm-backward REDUCTION of the stage that consumes dx: besides dx the
 * kernel epilogue accumulates, per channel of dx, S1 = sum dx*[n>0] and S2 = sum dx*[n>0]*xhat with
 * n = scale*y_prev + shift, xhat = (y_prev - mean)*rstd (y_prev = that stage's saved conv output, same
 * pixel grid as dx).  parts: [*nparts][3][cin] fp32 (third row zero), parts_floats >= (rows+64)*3*cin
 * with rows = unetdc_conv3x3_stats_rows(n*h*w, cin); *nparts receives the rows written.  The y_prev
 * read overlaps with matrix-core work here instead of costing a separate HBM pass. */
int unetdc_conv3x3_dgrad_bnstats(const void* dy, int lddy, const void* w_dgrad, void* dx, int lddx,
                                 const void* y_prev, int ldy_prev, const float* scale, const float* shift,
                                 const float* mean, const float* rstd, float* parts, int64_t parts_floats,
                                 int* nparts, int n, int h, int w, int cin, int cout, int dilation, int dtype,
                                 unetdc_stream_t s);
int unetdc_convT2x2_dgrad_bnstats(const void* dup, int lddup, const void* w_dgrad, void* dx, int lddx,
                                  const void* y_prev, int ldy_prev, const float* scale, const float* shift,
                                  const float* mean, const float* rstd, float* parts, int64_t parts_floats,
                                  int* nparts, int n, int h, int w, int cin, int cout, int dtype, unetdc_stream_t s);

/* ---- head: Conv2d(C, OC, 1) + sigmoid, models/model_2.py:32,79-80 ------------------------------
 * w [OC][C] fp32, probs/dprobs NCHW fp32 [n][OC][h][w]. */
int unetdc_head_fwd(const void* a, int lda, const float* w, const float* b, float* probs, int n, int h, int wd,
                    int c, int oc, int dtype, unetdc_stream_t s);
/* The same head fed from the RAW conv output y of the stage in front of it (dec1's second stage, models/model_2.py:76-79):
 * that stage's BatchNorm + ReLU, a = relu(scale * y + shift) (models/model_2.py:52-53), is applied while loading, rounded
 * through the storage type like a stored activation would be -- the activation tensor itself is never written or read
 * (train-mode forward; the eval-mode convolution folds BatchNorm into its own epilogue and feeds unetdc_head_fwd). */
int unetdc_head_fwd_bn(const void* y, int ldy, const float* scale, const float* shift, const float* w, const float* b,
                       float* probs, int n, int h, int wd, int c, int oc, int dtype, unetdc_stream_t s);
int64_t unetdc_head_bwd_workspace(int n, int h, int w, int c, int oc, int dtype);
int unetdc_head_bwd(const float* dprobs, const float* probs, const void* a, int lda, const float* w, void* da,
                    int ldda, float* dw, float* db, void* workspace, int64_t workspace_bytes, int n, int h, int wd,
                    int c, int oc, int dtype, unetdc_stream_t s);
/* head_bwd that also produces the BatchNorm-backward partial sums of the stage whose activated output `a` is (the
 * block feeding out_conv, models/model_2.py:76-79), in the layout unetdc_bn_relu_bwd takes as pre_parts: the
 * gradient da is then read once by that stage's backward instead of twice.  y_prev/scale/shift/mean/rstd: that
 * stage's saved conv output and batch statistics; parts needs (rows + 64) * 3 * c floats, *nparts receives rows.
 * a == NULL: the activation is recomputed from y_prev / scale / shift (the counterpart of unetdc_head_fwd_bn). */
int unetdc_head_bwd_bnstats(const float* dprobs, const float* probs, const void* a, int lda, const float* w, void* da,
                            int ldda, float* dw, float* db, void* workspace, int64_t workspace_bytes, const void* y_prev,
                            int ldy_prev, const float* scale, const float* shift, const float* mean, const float* rstd,
                            float* parts, int64_t parts_floats, int* nparts, int n, int h, int wd, int c, int oc, int dtype,
                            unetdc_stream_t s);

/* ---- fused Focal + Dice loss on probabilities: utils/metrics_DC.py:65-73 (FocalLoss :43-63, dice_loss :11-17) ---
 * probs/target: fp32 [nimg][hw] (nimg = N*C maps, hw pixels each).  fwd writes the scalar loss and the
 * per-map coefficients {2/(U+s), (2I+s)/(U+s)^2} that bwd needs; bwd writes
 * dprobs = grad_out[0] * d loss / d probs.  alpha/gamma: focal parameters; ratio: focal weight;
 * smooth: the Dice smoothing constant (1e-7 in the reference). */
int64_t unetdc_focal_dice_loss_workspace(int nimg, int64_t hw);
int unetdc_focal_dice_loss_fwd(const float* probs, const float* target, float* loss_out, float* coef, void* workspace,
                               int64_t workspace_bytes, int nimg, int64_t hw, float alpha, float gamma, float ratio,
                               float smooth, unetdc_stream_t s);
int unetdc_focal_dice_loss_bwd(const float* probs, const float* target, const float* coef, const float* grad_out,
                               float* dprobs, int nimg, int64_t hw, float alpha, float gamma, float ratio,
                               unetdc_stream_t s);

/* conv3x3_dgrad that also returns colsum[i] = sum over pixels of the STORED dx[:, c0 + i], i < c (fp32).
 * The decoder's first conv writes the gradient of torch.cat([up, enc]) (models/model_2.py:68,71,74,77); the
 * column sums of its first half are the ConvTranspose2d bias gradient (:20,23,26,29), so that gradient costs
 * no extra pass over the tensor. */
int64_t unetdc_conv3x3_dgrad_colsum_workspace(int n, int h, int w, int cin);
int unetdc_conv3x3_dgrad_colsum(const void* dy, int lddy, const void* w_dgrad, void* dx, int lddx, float* colsum, int c0,
                                int c, void* workspace, int64_t workspace_bytes, int n, int h, int w, int cin, int cout,
                                int dilation, int dtype, unetdc_stream_t s);

/* ---- per-channel column sum of an NHWC tensor (ConvTranspose2d bias gradient) ------------------ */
int64_t unetdc_channel_sum_workspace(int64_t npixels, int c);
int unetdc_channel_sum(const void* x, int ldx, float* out, void* workspace, int64_t workspace_bytes,
                       int64_t npixels, int c, int dtype, unetdc_stream_t s);

/* ---- droplet quantification: /root/reference/quantify_droplets_batch.py:56-57,81-95 ----------
 * unetdc_mask_from_probs: mask[y][x] = probs[sy][sx] > thresh (strict, fp32) with the nearest-neighbour index rule of
 *   cv2.resize(..., INTER_NEAREST): s = min(floor(d * src/dst), src-1); probs [ph][pw] fp32 (one image, one channel),
 *   mask [oh][ow] uint8 {0,1}.
 * unetdc_ccl_stats: 4-connected components of a {0,1} mask, objects smaller than min_area dropped, the rest numbered in
 *   raster order of their first pixel (= skimage.measure.label twice + regionprops order).  Outputs, per kept object in
 *   that order: out_area (pixels), out_sumy / out_sumx (sums of row / column indices: centroid = sum / area),
 *   out_root (linear index of the first pixel; may be NULL); *out_count = number of kept objects (may exceed max_out:
 *   only the first max_out are written).  All outputs are DEVICE pointers.  Exact integer arithmetic, order-independent. */
int unetdc_mask_from_probs(const float* probs, int ph, int pw, float thresh, uint8_t* mask, int oh, int ow,
                           unetdc_stream_t s);
/* The mask the reference's call actually produces: `cv2.resize(mask512, (ow, oh), cv2.INTER_NEAREST)`
 * (/root/reference/quantify_droplets_batch.py:57) passes the flag in the positional slot of `dst`, so OpenCV's default
 * 8-bit INTER_LINEAR runs on the {0,1} mask.  Tables as for unetdc_resize_linear_u8_to_chw_f32. */
int unetdc_mask_from_probs_linear(const float* probs, int ph, int pw, float thresh, uint8_t* mask, int oh, int ow,
                                  const int32_t* xofs, const int16_t* xcoef, const int32_t* yofs, const int16_t* ycoef,
                                  unetdc_stream_t s);
int64_t unetdc_ccl_workspace(int h, int w);
int unetdc_ccl_stats(const uint8_t* mask, int h, int w, int min_area, void* workspace, int64_t workspace_bytes,
                     int32_t* out_count, int32_t* out_area, int64_t* out_sumy, int64_t* out_sumx, int32_t* out_root,
                     int max_out, unetdc_stream_t s);

/* ---- input preprocessing: /root/reference/utils/data_loader.py:11-24, quantify_droplets_batch.py:40-46
 * unetdc_rolling_ball_u8: per channel of an interleaved HWC uint8 image: background = opening with the ksize x ksize
 *   ellipse of cv2.getStructuringElement (dilate(erode(.)), pixels outside the image ignored), dst = normalize_minmax(
 *   saturate(src - background)) to 0..255 (scale/shift in double, applied in float, round half to even).  ksize <= 128,
 *   channels <= 4.  dst may not alias src.
 * unetdc_resize_linear_u8_to_chw_f32: OpenCV's 8-bit INTER_LINEAR resize to (dh, dw), / 255, HWC -> CHW float32 (the
 *   network input layout).  xofs[dw] / yofs[dh]: source index of the first tap (x already clamped to [0, w-1]),
 *   xcoef[dw][2] / ycoef[dh][2]: the 11-bit coefficients (utils/data_loader.py:linear_tables builds them). */
int64_t unetdc_rolling_ball_workspace(int h, int w, int channels);
int unetdc_rolling_ball_u8(const uint8_t* src_hwc, uint8_t* dst_hwc, int h, int w, int channels, int ksize, void* workspace,
                           int64_t workspace_bytes, unetdc_stream_t s);
int unetdc_resize_linear_u8_to_chw_f32(const uint8_t* src_hwc, int h, int w, int channels, float* dst_chw, int dh, int dw,
                                       const int32_t* xofs, const int16_t* xcoef, const int32_t* yofs, const int16_t* ycoef,
                                       unetdc_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* UNETDC_HIP_H */
