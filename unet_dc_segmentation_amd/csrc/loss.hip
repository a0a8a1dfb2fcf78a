// Fused Focal + Dice loss on probabilities (the step between forward and
// backward).  Semantics = /root/reference/utils/metrics_DC.py:65-73 (+ FocalLoss.forward :43-63,
// dice_loss :11-17):
//   bce_j   = -(t_j * max(log p_j, -100) + (1 - t_j) * max(log(1 - p_j), -100))   (torch's BCE clamp)
//   focal   = mean_j alpha * (1 - exp(-bce_j))^gamma * bce_j
//   dice_i  = (2 sum_j p t + s) / (sum_j p + sum_j t + s)    per (image, channel) i,  s = 1e-7
//   loss    = ratio * focal + (1 - ratio) * (1 - mean_i dice_i)
// Forward: one two-stage reduction (4 sums per image) + a one-block finalise that also emits the
// per-image coefficients of d dice / d p; backward: one elementwise kernel.  Replaces ~25 ATen
// elementwise/reduction launches and their autograd hops per training step.
#include "kernels.h"

namespace unetdc {

__device__ __forceinline__ void focal_terms(float p, float t, float alpha, float gamma, float& focal, float& dfocal) {
  const float lp = logf(p), lq = log1pf(-p);
  const float clp = fmaxf(lp, -100.f), clq = fmaxf(lq, -100.f);
  const float bce = -(t * clp + (1.f - t) * clq);
  const float dbce = -((lp > -100.f ? t / p : 0.f) - (lq > -100.f ? (1.f - t) / (1.f - p) : 0.f));
  const float pt = expf(-bce);
  const float om = 1.f - pt;
  const float omg1 = (gamma == 2.f) ? om : powf(om, gamma - 1.f);       // (1-pt)^(gamma-1)
  const float omg = omg1 * om;                                          // (1-pt)^gamma
  focal = alpha * omg * bce;
  dfocal = alpha * (gamma * omg1 * pt * bce + omg) * dbce;
}

// partial sums: parts[(img * nblk + blk) * 4 + {focal, p*t, p, t}]
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                          float* __restrict__ parts, long hw, float alpha,
                                                          float gamma) {
  __shared__ float red[4][4];
  const int img = blockIdx.y, nblk = gridDim.x;
  const float* pp = p + (long)img * hw;
  const float* tt = t + (long)img * hw;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < hw; j += (long)nblk * 256) {
    const float pv = pp[j], tv = tt[j];
    float f, df;
    focal_terms(pv, tv, alpha, gamma, f, df);
    s[0] += f; s[1] = fmaf(pv, tv, s[1]); s[2] += pv; s[3] += tv;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] = wave_sum(s[k]);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 4; ++k) red[wave][k] = s[k];
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    parts[((long)img * nblk + blockIdx.x) * 4 + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  }
}

// one block: out[0] = loss; coef[img] = {2/(U+s), (2I+s)/(U+s)^2}
// 32 lanes share an image: lane l adds block partials l, l+32, ... (fp64), then a fixed shuffle tree.
__device__ __forceinline__ double loss_lane32_sum(double v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
  return v;
}
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ parts, int nimg, int nblk,
                                                            double numel, float ratio, float smooth, float* out,
                                                            float* coef) {
  __shared__ double sh_focal[8], sh_dice[8];
  const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
  double focal = 0.0, dice = 0.0;
  for (int img = grp; img < nimg; img += 8) {
    double f = 0.0, I = 0.0, P = 0.0, T = 0.0;
    for (int b = l; b < nblk; b += 32) {
      const float* q = parts + ((long)img * nblk + b) * 4;
      f += q[0]; I += q[1]; P += q[2]; T += q[3];
    }
    f = loss_lane32_sum(f); I = loss_lane32_sum(I); P = loss_lane32_sum(P); T = loss_lane32_sum(T);
    const double U = P + T + (double)smooth;
    focal += f;
    dice += (2.0 * I + (double)smooth) / U;
    if (l == 0) {
      coef[img * 2 + 0] = (float)(2.0 / U);
      coef[img * 2 + 1] = (float)((2.0 * I + (double)smooth) / (U * U));
    }
  }
  if (l == 0) { sh_focal[grp] = focal; sh_dice[grp] = dice; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double f = 0.0, d = 0.0;
    for (int i = 0; i < 8; ++i) { f += sh_focal[i]; d += sh_dice[i]; }
    out[0] = (float)((double)ratio * (f / numel) + (1.0 - (double)ratio) * (1.0 - d / nimg));
  }
}

// dp_j = gout * ( ratio * dfocal_j / numel - (1-ratio)/nimg * (t_j*c1 - c2) )
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                       const float* __restrict__ coef, const float* __restrict__ gout,
                                                       float* __restrict__ dp, long hw, int nimg, double numel,
                                                       float alpha, float gamma, float ratio) {
  const int img = blockIdx.y;
  const float c1 = coef[img * 2 + 0], c2 = coef[img * 2 + 1];
  const float g = gout[0];
  const float kf = (float)((double)ratio / numel), kd = (1.f - ratio) / (float)nimg;
  const long base = (long)img * hw;
  for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < hw; j += (long)gridDim.x * 256) {
    const float pv = p[base + j], tv = t[base + j];
    float f, df;
    focal_terms(pv, tv, alpha, gamma, f, df);
    dp[base + j] = g * (kf * df - kd * (tv * c1 - c2));
  }
}

static int loss_blocks(long hw) {
  long nb = (hw + 256L * 8 - 1) / (256L * 8);
  if (nb > 128) nb = 128;
  if (nb < 1) nb = 1;
  return (int)nb;
}

long loss_workspace_bytes(int nimg, long hw) { return (long)nimg * loss_blocks(hw) * 4 * 4; }

int launch_loss_fwd(const float* p, const float* t, float* loss_out, float* coef, void* workspace, long workspace_bytes,
                    int nimg, long hw, float alpha, float gamma, float ratio, float smooth, hipStream_t stream) {
  UNETDC_REQUIRE(p && t && loss_out && coef && workspace, "focal_dice_loss: null pointer");
  UNETDC_REQUIRE(nimg > 0 && nimg <= 65535 && hw > 0, "focal_dice_loss: bad shape");
  const int nb = loss_blocks(hw);
  if ((long)nimg * nb * 16 > workspace_bytes) {
    set_error("focal_dice_loss: workspace too small");
    return UNETDC_EWORKSPACE;
  }
  float* parts = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(nb, nimg), dim3(256), 0, stream, p, t, parts, hw, alpha, gamma);
  int rc = check_launch("loss_reduce_kernel");
  if (rc != UNETDC_OK) return rc;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, stream, parts, nimg, nb, (double)nimg * (double)hw,
                     ratio, smooth, loss_out, coef);
  return check_launch("loss_finalize_kernel");
}

int launch_loss_bwd(const float* p, const float* t, const float* coef, const float* gout, float* dp, int nimg, long hw,
                    float alpha, float gamma, float ratio, hipStream_t stream) {
  UNETDC_REQUIRE(p && t && coef && gout && dp, "focal_dice_loss_bwd: null pointer");
  UNETDC_REQUIRE(nimg > 0 && nimg <= 65535 && hw > 0, "focal_dice_loss_bwd: bad shape");
  const int nb = loss_blocks(hw) * 4;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(nb, nimg), dim3(256), 0, stream, p, t, coef, gout, dp, hw, nimg,
                     (double)nimg * (double)hw, alpha, gamma, ratio);
  return check_launch("loss_bwd_kernel");
}

}  // namespace unetdc
