// Fused optimizer step: Adam on the fp32 master parameters AND the re-pack of the K-contiguous compute-type weight
// images, one launch for the whole network  (replaces torch.optim.Adam's multi-tensor kernel +
// pack_many_kernel: three passes over the 124 MB of parameters become one).
//
//   reference: optimizer = optim.Adam(model.parameters(), lr=0.001) ... optimizer.step()      train_DC_focal.py:224,255
//   m <- m + (1 - b1) (g - m);  v <- b2 v + (1 - b2) g g;  p <- p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
//   (the formulation of torch's _fused_adam_; g is first scaled by `gscale`, the 1/G of a SUM all-reduce)
//
// Work decomposition = the one of pack_many_kernel (elementwise.hip): a device-resident descriptor table; a workgroup
// owns one 32 x 32 (out x in) channel tile of a packed tensor with all its taps -- it streams g, p, m, v of that tile in
// the parameter's own contiguous rows, writes p, m, v back in place and the two packed images from an LDS copy of the
// NEW p -- or 4096 consecutive elements of an unpacked tensor (biases, BatchNorm affine, first layer, head).
// HBM-bound: 28 bytes per parameter + 4 (bf16) / 8 (fp32) for the images; nothing else reads the parameters per step.
#include "kernels.h"

namespace unetdc {

struct AdamDesc {
  float* p;            // fp32 master parameter (PyTorch layout), updated in place
  float* m;            // exp_avg
  float* v;            // exp_avg_sq
  void* wf;            // forward image or null
  void* wd;            // dgrad image or null
  long g_off;          // element offset of this tensor's gradient in the flat gradient buffer
  long begin;          // first workgroup of this tensor in the launch
  long numel;
  int a, b;            // conv3x3: (Cout, Cin); convT: (Cin, Cout); plain: unused
  int kind;            // 0 conv3x3 packed, 1 convT2x2 packed, 2 plain
  int pad;
};
static_assert(sizeof(AdamDesc) == 80, "AdamDesc layout (mirrored by unet_dc_segmentation_amd/optim.py)");

struct AdamHyper {
  float b2, omb1, omb2, eps, step_size, inv_bc2_sqrt, gscale;   // omb = 1 - beta, formed in double on the host like torch does
};

__device__ __forceinline__ float adam_update(float& p, float& m, float& v, float g, const AdamHyper& h) {
  g *= h.gscale;
  m = fmaf(h.omb1, g - m, m);
  v = fmaf(v, h.b2, h.omb2 * g * g);
  const float denom = sqrtf(v) * h.inv_bc2_sqrt + h.eps;
  p -= h.step_size * (m / denom);
  return p;
}

template <typename T>
__global__ __launch_bounds__(256) void adam_pack_kernel(const AdamDesc* __restrict__ table, int n,
                                                        const float* __restrict__ flat_grad, const AdamHyper h) {
  __shared__ float tile[32 * 289];
  const int tid = threadIdx.x;
  const long tix = blockIdx.x;
  int lo = 0, hi = n - 1;                      // last descriptor with begin <= tix
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].begin <= tix) lo = mid; else hi = mid - 1;
  }
  const AdamDesc d = table[lo];
  const long local = tix - d.begin;
  const float* g = flat_grad + d.g_off;
  if (d.kind == 2) {
    const long i0 = local * 4096;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
      const long i = i0 + k * 256 + tid;
      if (i < d.numel) {
        float p = d.p[i], m = d.m[i], v = d.v[i];
        adam_update(p, m, v, g[i], h);
        d.p[i] = p; d.m[i] = m; d.v[i] = v;
      }
    }
    return;
  }
  T* wf = reinterpret_cast<T*>(d.wf);
  T* wd = reinterpret_cast<T*>(d.wd);
  if (d.kind == 0) {
    const int Co = d.a, Ci = d.b, tci = Ci / 32;
    const int co0 = (int)(local / tci) * 32, ci0 = (int)(local % tci) * 32;
    for (int i = tid; i < 32 * 288; i += 256) {            // w[co][ci0 .. ci0+31][9]: 288 contiguous floats
      const int co = i / 288, rem = i - co * 288;
      const long gi = ((long)(co0 + co) * Ci + ci0) * 9 + rem;
      float p = d.p[gi], m = d.m[gi], v = d.v[gi];
      tile[co * 289 + rem] = adam_update(p, m, v, g[gi], h);
      d.p[gi] = p; d.m[gi] = m; d.v[gi] = v;
    }
    if (!wf) return;
    __syncthreads();
    for (int i = tid; i < 9 * 1024; i += 256) {
      const int t = i >> 10, x = (i >> 5) & 31, y = i & 31;
      wf[((long)t * Co + co0 + x) * Ci + ci0 + y] = from_f32<T>(tile[x * 289 + y * 9 + t]);            // x = co, y = ci
      if (wd) wd[((long)(8 - t) * Ci + ci0 + x) * Co + co0 + y] = from_f32<T>(tile[y * 289 + x * 9 + t]);  // x = ci, y = co
    }
  } else {
    const int Ci = d.a, Co = d.b, tco = Co / 32;
    const int ci0 = (int)(local / tco) * 32, co0 = (int)(local % tco) * 32;
    for (int i = tid; i < 32 * 128; i += 256) {            // w[ci][co0 .. co0+31][4]: 128 contiguous floats
      const int ci = i >> 7, rem = i & 127;
      const long gi = ((long)(ci0 + ci) * Co + co0) * 4 + rem;
      float p = d.p[gi], m = d.m[gi], v = d.v[gi];
      tile[ci * 129 + rem] = adam_update(p, m, v, g[gi], h);
      d.p[gi] = p; d.m[gi] = m; d.v[gi] = v;
    }
    if (!wf) return;
    __syncthreads();
    for (int i = tid; i < 4 * 1024; i += 256) {
      const int ab = i >> 10, x = (i >> 5) & 31, y = i & 31;
      wf[((long)ab * Co + co0 + x) * Ci + ci0 + y] = from_f32<T>(tile[y * 129 + x * 4 + ab]);              // x = co, y = ci
      if (wd) wd[((long)ab * Ci + ci0 + x) * Co + co0 + y] = from_f32<T>(tile[x * 129 + y * 4 + ab]);        // x = ci, y = co
    }
  }
}

int launch_adam_step(const void* table_dev, int n, long total_blocks, const float* flat_grad, double lr, double beta1,
                     double beta2, double eps, long step, double grad_scale, int dtype, hipStream_t stream) {
  UNETDC_REQUIRE(table_dev && flat_grad && n > 0 && total_blocks > 0 && total_blocks < (1L << 31), "adam_step: empty table");
  UNETDC_REQUIRE(dtype == UNETDC_F32 || dtype == UNETDC_BF16, "adam_step: bad dtype %d", dtype);
  UNETDC_REQUIRE(step >= 1 && lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0.,
                 "adam_step: bad hyper-parameters");
  // bias corrections in double like torch (1 - beta ** step), then rounded once
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  AdamHyper h;
  h.b2 = (float)beta2; h.omb1 = (float)(1.0 - beta1); h.omb2 = (float)(1.0 - beta2); h.eps = (float)eps;
  h.gscale = (float)grad_scale;
  h.step_size = (float)(lr / bc1);
  h.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  const AdamDesc* t = reinterpret_cast<const AdamDesc*>(table_dev);
  if (dtype == UNETDC_BF16)
    hipLaunchKernelGGL(adam_pack_kernel<bf16_t>, dim3((unsigned)total_blocks), dim3(256), 0, stream, t, n, flat_grad, h);
  else
    hipLaunchKernelGGL(adam_pack_kernel<float>, dim3((unsigned)total_blocks), dim3(256), 0, stream, t, n, flat_grad, h);
  return check_launch("adam_pack_kernel");
}

}  // namespace unetdc
