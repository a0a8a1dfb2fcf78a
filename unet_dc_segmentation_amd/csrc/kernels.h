// Parameter blocks and host launchers shared between the kernel files and the C-ABI layer.
#pragma once
#include "common.h"

namespace unetdc {

// MODE_BNBWD: plain store of a gradient tensor dA plus, fused, the per-channel partial sums of the
// BatchNorm-backward reduction of the stage that consumes dA:  S1 = sum dA*[n>0], S2 = sum dA*[n>0]*xhat
// with n = scale*y + shift, xhat = (y - mean)*rstd read from that stage's saved conv output y.
enum { MODE_STORE = 0, MODE_STATS = 1, MODE_AFFINE_RELU = 2, MODE_SHUFFLE = 3, MODE_BNBWD = 4 };
#define UNETDC_EUNSUPPORTED (-4)

struct IgemmParams {
  const void* x;
  const void* w;
  void* out;
  const float* bias;
  const float* scale;
  const float* shift;
  float* stats;
  const void* bn_y;          // MODE_BNBWD: saved conv output of the consuming stage, [M][bn_ldy]
  const float* bn_mean;
  const float* bn_rstd;
  int bn_ldy;
  int M, Ho, Wo, Hi, Wi, Cin, Cout, ldx, ldo, ntaps, stride, mode, shuf_c;
  int mblocks, nblocks;
  int wo_shift, howo_shift;  // log2(Wo), log2(Ho*Wo) when both are powers of two, else -1 (set by launch_igemm)
  int quad_bpr, quad_bpi;    // > 0: the M index walks 16 x 16 pixel blocks (blocks per image row / per image), see igemm_dma16.hip
  // input normalisation ("bnin"): x is the RAW conv output of the producing stage; its BatchNorm + ReLU, relu(in_scale * x +
  // in_shift) rounded through the storage type, is applied to every staged patch in LDS (igemm_lattice.hip, INORM)
  const float* in_scale;
  const float* in_shift;
  // optional (wide INORM form): the normalised activation is ALSO written out, [M][ld_act], by the items of n-block 0 -- the
  // weight gradient of the stage then reads a stored activation (plain kernel) without a stand-alone normalisation pass
  void* act_out;
  int ld_act;
  int offy[9];
  int offx[9];
};
int launch_igemm(IgemmParams& p, int dtype, hipStream_t stream);
int igemm_mblocks(long M, int Cout);
bool igemm_lattice_supported(const IgemmParams& p, int dtype);      // igemm_lattice.hip: persistent lattice-halo conv (bf16)
int launch_igemm_lattice(IgemmParams& p, hipStream_t stream);
bool igemm_lattice_bnin_supported(const IgemmParams& p, int dtype);  // x = raw conv output, BatchNorm + ReLU applied per staged patch
bool igemm_lattice_bnin_writes_activation(const IgemmParams& p, int dtype);   // ... and that form can store the normalised activation
bool igemm_dma16_supported(const IgemmParams& p, int dtype);
int launch_igemm_dma16(IgemmParams& p, int cfg, hipStream_t stream);

struct WgradParams {
  const void* a;
  const void* b;
  float* part;
  int N, H, W, Hb, Wb, CI, CJ, lda, ldb, ntaps, stride;
  int P;            // N*H*W
  int chunk;        // pixels per K-slice (multiple of the block's pixel step)
  int ksplit, itiles, jtiles;
  int adv_y, adv_x; // pixel step decomposed: step = adv_y*W + adv_x
  int offy[9];
  int offx[9];
  const float* in_scale;   // "bnin": b (the conv input) is the RAW output of the producing stage, normalised on load
  const float* in_shift;
};
int launch_wgrad(WgradParams& p, float* out, void* workspace, long workspace_bytes, int dtype, hipStream_t stream);
bool wgrad_bnin_supported(int N, int H, int W, int CI, int CJ, int lda, int ldb, int d, int dtype);
long wgrad_workspace_bytes(long P, int CI, int CJ, int ntaps, int dtype);
long wgrad_fused_workspace_bytes(int N, int H, int W, int CI, int CJ, int dtype);
long wgrad_rect_workspace_bytes(int N, int H, int W, int CI, int CJ, int d);
// convt_wgrad.hip: tap-fused ConvTranspose2d weight gradient (bf16): X staged once for the four taps
bool convt_wgrad_fused_supported(int N, int H, int W, int CI, int CJ, int ldx, int lddy, int dtype);
long convt_wgrad_fused_workspace_bytes(int N, int H, int W, int CI, int CJ);
int launch_convt_wgrad_fused(const void* x, int ldx, const void* dy, int lddy, float* part, int N, int H, int W, int CI, int CJ,
                             int* units_out, hipStream_t stream);

struct FirstParams {
  const float* x;        // [N][Cin][H][W] fp32
  const float* w;        // [Cout][Cin][3][3] fp32 (PyTorch layout)
  const float* bias;     // [Cout] or null
  const float* scale;    // eval mode: y = relu(acc*scale + shift)
  const float* shift;
  void* y;               // [N*H*W][ldy] T
  float* stats;          // [gridDim.x][2][Cout] or null
  int N, H, W, Cin, Cout, ldy, dil;
};
struct FirstWgradParams {
  const float* x;     // [N][Cin][H][W]
  const void* dy;     // [P][lddy] T
  float* part;        // [gridDim.x][Cin][9][Cout]
  int N, H, W, Cin, Cout, lddy, dil;
  // "bn" form (row-run kernel, Cin = 1): `dy` is the gradient of the stage's ACTIVATED output and the BatchNorm + ReLU backward
  // of the stage is applied on load, dy = k1 * [a > 0] * dz - k2 - k3 * xhat rounded through the storage type (exactly what the
  // stand-alone pass would have stored) from the saved conv output y and the coefficients of unetdc_bn_relu_bwd_coeffs
  const void* bn_y; int bn_ldy;
  const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_rstd; const float* bn_k;   // bn_k: [3][Cout]
};
int launch_first_fwd(FirstParams& p, int dtype, hipStream_t stream);
bool first_wgrad_bn_supported(int N, int H, int W, int Cin, int Cout, int dil, int dtype);
int launch_bn_bwd_coeffs(const float* pre_parts, int pre_nparts, long count, const float* gamma, const float* rstd, float* dgamma,
                         float* dbeta, float* dbias, float* coeffs, int C, hipStream_t stream);
int first_conv_mblocks(long P, int Cin, int Cout);
bool first_mfma_supported(long P, int Cin, int Cout);
int first_mfma_mblocks(long P);
int launch_first_mfma_fwd(FirstParams& p, int dtype, hipStream_t stream);
long first_mfma_wgrad_workspace_bytes(long P, int Cin, int Cout);
int launch_first_mfma_wgrad(FirstWgradParams& p, int* nblk_out, int dtype, hipStream_t stream);
long first_wgrad_workspace_bytes(long P, int Cin, int Cout);
int launch_first_wgrad(FirstWgradParams& p, float* dw, void* workspace, long workspace_bytes, int dtype,
                       hipStream_t stream);
int launch_first_dgrad(const void* dy, int lddy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout, int dil,
                       int dtype, hipStream_t stream);

struct ApplyParams {
  const void* y; void* a; void* pooled;
  const float* scale; const float* shift;
  int N, H, W, C, ldy, lda, ldp;
};
struct BnBwdParams {
  const void* dskip; const void* dpool; const void* y; void* dy;
  const float* scale; const float* shift; const float* mean; const float* rstd;
  const float* k1; const float* k2; const float* k3;
  float* parts;            // [gridDim.x][3][C]
  int N, H, W, C, lds, ldp, ldy, lddy;
  // the stage feeds the 1 x 1 head with ONE output channel: its activation gradient is dz * w[c], dz = dprobs * p * (1 - p)
  // per pixel -- recomputed here from the fp32 [N, 1, H, W] tensors instead of being stored by the head backward and read
  // back (non-pooled apply pass only; rounded through the storage type like the stored tensor was)
  const float* head_dprobs; const float* head_probs; const float* head_w;
};
struct HeadParams {
  const void* a; const float* w; const float* b; float* probs;
  const float* dprobs; void* da; float* parts;      // backward
  int N, H, W, C, OC, lda, ldda;
  // optional: BatchNorm-backward partial sums of the stage that produced `a` (its saved conv output y + batch statistics)
  const void* bn_y; const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_rstd;
  float* bn_parts;  // [gridDim.x][3][C]
  int bn_ldy;
};
int launch_bn_finalize(const float* parts, int nparts, long count, const float* gamma, const float* beta, float eps,
                       float momentum, float* rm, float* rv, float* scale, float* shift, float* mean, float* rstd,
                       int C, hipStream_t stream);
int launch_bn_eval_affine(const float* gamma, const float* beta, const float* rm, const float* rv,
                          const float* conv_bias, float eps, float* scale, float* shift, int C, hipStream_t stream);
int launch_apply(ApplyParams& p, int dtype, hipStream_t stream);
long bn_bwd_workspace_bytes(int N, int H, int W, int C, int pooled, int dtype);
int launch_bn_bwd(BnBwdParams& p, const float* gamma, float* dgamma, float* dbeta, float* dbias, void* workspace,
                  long workspace_bytes, const float* pre_parts, int pre_nparts, int dtype, hipStream_t stream,
                  bool frozen = false);
int launch_bn_frozen_affine(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* scale,
                            float* shift, float* mean, float* rstd, int C, hipStream_t stream);
int launch_stats_colsum_rows(const float* parts, int nparts, int row_floats, int c0, int c, float* out, hipStream_t stream);
int launch_bn_bwd_reduce_only(BnBwdParams& p, float* parts, long parts_floats, int* nparts, int dtype,
                              hipStream_t stream);
long head_bwd_workspace_bytes(int N, int H, int W, int C, int OC, int dtype);
int launch_head_fwd(HeadParams& p, int dtype, hipStream_t stream);
int launch_head_bwd(HeadParams& p, float* dw, float* db, void* workspace, long workspace_bytes, int dtype,
                    hipStream_t stream, int* bn_nparts = nullptr, long bn_parts_floats = 0);
int launch_pack_conv3x3(const float* w, void* wf, void* wd, int Co, int Ci, int dtype, hipStream_t stream);
int launch_pack_convT2x2(const float* w, void* wf, void* wd, int Ci, int Co, int dtype, hipStream_t stream);
int launch_pack_many(const void* table_dev, int n, long total, int dtype, hipStream_t stream);
int launch_adam_step(const void* table_dev, int n, long total_blocks, const float* flat_grad, double lr, double beta1,
                     double beta2, double eps, long step, double grad_scale, int dtype, hipStream_t stream);   // optim.hip
int launch_stats_colsum(const float* parts, int nparts, int ctotal, int c0, int c, float* out, hipStream_t stream);
long channel_sum_workspace_bytes(long P, int C);
int launch_channel_sum(const void* x, int ldx, float* out, void* workspace, long workspace_bytes, long P, int C,
                       int dtype, hipStream_t stream);

// ccl.hip: droplet quantification
int launch_mask_from_probs(const float* probs, int ph, int pw, float thresh, unsigned char* mask, int oh, int ow,
                           hipStream_t stream);
int launch_mask_from_probs_linear(const float* probs, int ph, int pw, float thresh, unsigned char* mask, int oh, int ow,
                                  const int* xofs, const short* xcoef, const int* yofs, const short* ycoef, hipStream_t stream);
long ccl_workspace_bytes(int h, int w);
int launch_ccl_stats(const unsigned char* mask, int h, int w, int min_area, void* workspace, long workspace_bytes,
                     int* out_count, int* out_area, long long* out_sumy, long long* out_sumx, int* out_root, int max_out,
                     hipStream_t stream);

// preprocess.hip: rolling-ball correction + bilinear resize to the network input
long rolling_ball_workspace_bytes(int h, int w, int cn);
int launch_rolling_ball(const unsigned char* src, unsigned char* dst, int h, int w, int cn, int k, void* workspace,
                        long workspace_bytes, hipStream_t stream);
int launch_resize_linear_chw(const unsigned char* src, int h, int w, int cn, float* dst, int dh, int dw, const int* xofs,
                             const short* xa, const int* yofs, const short* ya, hipStream_t stream);

long loss_workspace_bytes(int nimg, long hw);
int launch_loss_fwd(const float* p, const float* t, float* loss_out, float* coef, void* workspace, long workspace_bytes,
                    int nimg, long hw, float alpha, float gamma, float ratio, float smooth, hipStream_t stream);
int launch_loss_bwd(const float* p, const float* t, const float* coef, const float* gout, float* dp, int nimg, long hw,
                    float alpha, float gamma, float ratio, hipStream_t stream);

}  // namespace unetdc
