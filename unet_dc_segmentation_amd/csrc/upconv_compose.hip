// Composed decoder up-path (round 5): conv3x3(W3[:, :C]) o convT2x2(WT) as ONE operator on the low-res tensor.
//
//   up  = ConvTranspose2d(2C -> C, k = 2, s = 2)(h) + bT          /root/reference/models/model_2.py:20-29, 67-76
//   y   = Conv2d(2C -> C, 3 x 3, pad 1)(cat[up, skip]) + b3       :21-30 (decN.0), no BatchNorm in between
//
// For an output pixel (2r + py, 2s + px) the nine taps over `up` touch the 2 x 2 low-res neighbourhood rows {r - 1 + py, r + py}
// x cols {s - 1 + px, s + px} of h, each through ONE ConvT tap (a, b) = ((py + ky + 1) & 1, (px + kx + 1) & 1):
//   y_up[phase p][pixel] = sum_{t = (ty, tx)} sum_ci  W'[p][t][co][ci] * h[lo + t - 1 + p][ci]
//   W'[p][t][co][ci]     = sum_{ky in T(py, ty)} sum_{kx in T(px, tx)} sum_c  W3[co][c][ky][kx] * WT[ci][c][a][b],   ty = ((py + ky + 1) >> 1) - py
// (zero padding of `up` = zero padding of h; the ConvT bias needs border classes: btap[t9][co] = sum_c W3[co][c][t9] bT[c]).
// This file: the weight-side GEMMs -- W' from the packed images, and back: dW3[:, :C] and dWT from dW' -- plus the small
// tables and passes around them.  The pixel-side kernels are the CMP forms of igemm_lattice_kernel and upconv_wgrad.hip.
#include <stdio.h>

#include "../../include/unetdc_hip.h"
#include "kernels.h"

namespace unetdc {

#if defined(__HIP_DEVICE_COMPILE__)
// ty (or tx) and the ConvT tap a (or b) of conv tap ky (kx) for output phase py (px)
__device__ __forceinline__ int upc_t(int ph, int k) { return ((ph + k + 1) >> 1) - ph; }
__device__ __forceinline__ int upc_a(int ph, int k) { return (ph + k + 1) & 1; }
#endif

// One 64 x 64 tile of  C = sum_terms A_term [M x K] * B_term [N x K]^T  (both operands K-contiguous bf16, fp32 accumulate) per
// workgroup; wave (wm, wn) owns 32 x 32 = 2 x 2 MFMA tiles; fragments come straight from global memory (the operands are weight
// images of a few MB that live in L2; 36 products of [C x C] x [C x 2C] per level: 7 GFLOP at C = 256).
//   KIND 0  compose:  blockIdx.z = phase * 4 + t;  M = co (C), N = ci (2C), K = c (C)
//           A = w3_fwd[ky * 3 + kx][co][c] (ld 2C), B = wt_dgrad[a * 2 + b][ci][c] (ld C); out: wc_fwd[z][co][ci], wc_dgrad[phase * 4 + 3 - t][ci][co] (bf16)
//   KIND 1  dW3[:, :C]:  blockIdx.z = ky * 3 + kx;  M = co (C), N = c (C), K = ci (2C)
//           A = dwb[phase * 4 + t][co][ci] (ld 2C), B = wt_fwd[(a * 2 + b) * C + c][ci] (ld 2C); out: dw3[(co * 2C + c) * 9 + z] (fp32)
//   KIND 2  dWT:  blockIdx.z = a * 2 + b;  M = ci (2C), N = c (C), K = co (C)
//           A = dwbt[phase * 4 + t][ci][co] (ld C), B = w3_dgrad[8 - (ky * 3 + kx)][c][co] (ld C); out: dwt[(ci * C + c) * 4 + z] (fp32)
struct UpcGemmParams {
  const bf16_t* a;
  const bf16_t* b;
  void* out0;
  void* out1;
  int C;
};

template <int KIND>
__global__ __launch_bounds__(256) void upc_gemm_kernel(const UpcGemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int C = p.C, C2 = 2 * p.C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 64 + wm * 32, n0 = blockIdx.x * 64 + wn * 32;
  const int r16 = lane & 15, kq = lane >> 4;
  const int z = blockIdx.z;
  const int K = KIND == 1 ? C2 : C;
  const int lda = KIND == 2 ? C : C2, ldb = KIND == 1 ? C2 : C;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  auto product = [&](const bf16_t* A, const bf16_t* B) {
    for (int k0 = 0; k0 < K; k0 += 32) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fa[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(A + (long)(m0 + 16 * i + r16) * lda + k0 + 8 * kq));
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fb[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(B + (long)(n0 + 16 * j + r16) * ldb + k0 + 8 * kq));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };

  if (KIND == 0) {
    const int ph = z >> 2, t = z & 3, py = ph >> 1, px = ph & 1, ty = t >> 1, tx = t & 1;
    for (int ky = 0; ky < 3; ++ky) {
      if (upc_t(py, ky) != ty) continue;
      for (int kx = 0; kx < 3; ++kx) {
        if (upc_t(px, kx) != tx) continue;
        const int ab = upc_a(py, ky) * 2 + upc_a(px, kx);
        product(p.a + (long)(ky * 3 + kx) * C * C2, p.b + (long)ab * C2 * C);
      }
    }
  } else if (KIND == 1) {
    const int ky = z / 3, kx = z - 3 * ky;
    for (int ph = 0; ph < 4; ++ph) {
      const int py = ph >> 1, px = ph & 1;
      const int t = upc_t(py, ky) * 2 + upc_t(px, kx), ab = upc_a(py, ky) * 2 + upc_a(px, kx);
      product(p.a + (long)(ph * 4 + t) * C * C2, p.b + (long)ab * C * C2);
    }
  } else {
    const int a = z >> 1, b = z & 1;
    for (int py = 0; py < 2; ++py)
      for (int ky = 0; ky < 3; ++ky) {
        if (upc_a(py, ky) != a) continue;
        for (int px = 0; px < 2; ++px)
          for (int kx = 0; kx < 3; ++kx) {
            if (upc_a(px, kx) != b) continue;
            const int ph = py * 2 + px, t = upc_t(py, ky) * 2 + upc_t(px, kx);
            product(p.a + (long)(ph * 4 + t) * C2 * C, p.b + (long)(8 - (ky * 3 + kx)) * C2 * C);
          }
      }
  }

  // accumulator element v of a 16 x 16 tile: row 4 * (lane >> 4) + v, column lane & 15
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int m = m0 + 16 * i + 4 * kq + v, n = n0 + 16 * j + r16;
        const float val = acc[i][j][v];
        if (KIND == 0) {
          const int ph = z >> 2, t = z & 3;
          reinterpret_cast<bf16_t*>(p.out0)[((long)z * C + m) * C2 + n] = from_f32<bf16_t>(val);
          reinterpret_cast<bf16_t*>(p.out1)[((long)(ph * 4 + 3 - t) * C2 + n) * C + m] = from_f32<bf16_t>(val);
        } else if (KIND == 1) {
          reinterpret_cast<float*>(p.out0)[((long)m * C2 + n) * 9 + z] = val;
        } else {
          reinterpret_cast<float*>(p.out0)[((long)m * C + n) * 4 + z] = val;
        }
      }
#endif
}

// skip-half slices of the packed dec.0 images (contiguous copies: the convolution kernels take a dense [9][Cout][Cin] image)
//   wskip_fwd[t][co][c] = w3_fwd[t][co][C + c];   wskip_dgrad[t'][c][co] = w3_dgrad[t'][C + c][co]
__global__ __launch_bounds__(256) void upc_skip_slices_kernel(const bf16_t* __restrict__ w3f, const bf16_t* __restrict__ w3d,
                                                              bf16_t* __restrict__ sf, bf16_t* __restrict__ sd, int C) {
  const long n = 9L * C * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long r = i / C;                                  // t * C + row
    const int row = (int)(r % C), t = (int)(r / C);
    sf[i] = w3f[((long)t * C + row) * 2 * C + C + c];      // row = co, c = skip channel
    sd[i] = w3d[((long)t * 2 * C + C + row) * C + c];      // row = skip channel, c = co
  }
}

// btab[0][co] = b3[co] + sum_t btap[t][co] (the bias every interior pixel sees), btab[1 + t][co] = btap[t][co] = sum_c W3[co][c][t] bT[c]
// one workgroup per output channel: thread c reads the nine contiguous taps of W3[co][c] (and c + 256, ...), fixed-order tree in LDS
__global__ __launch_bounds__(256) void upc_bias_table_kernel(const float* __restrict__ w3, const float* __restrict__ b3,
                                                             const float* __restrict__ bt, float* __restrict__ btab, int C) {
  __shared__ float red[9 * 256];
  const int co = blockIdx.x, tid = threadIdx.x;
  float s[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float b = bt[c];
    const float* w = w3 + ((long)co * 2 * C + c) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) s[t] = fmaf(w[t], b, s[t]);
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) red[t * 256 + tid] = s[t];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
#pragma unroll
      for (int t = 0; t < 9; ++t) red[t * 256 + tid] += red[t * 256 + tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) {
    float tot = b3 ? b3[co] : 0.f;
    for (int t = 0; t < 9; ++t) { btab[(1 + t) * C + co] = red[t * 256]; tot += red[t * 256]; }
    btab[co] = tot;
  }
}

// border pixels of the 2x finer map: the taps that leave the image carry no ConvT bias -- y[pix][co] -= sum_{t outside} btap[t][co]
// (one thread per (border pixel, 8 channels); a corner pixel is handled once, by the row pass)
__global__ __launch_bounds__(256) void upc_border_bias_kernel(bf16_t* __restrict__ y, int ldy, const float* __restrict__ btab, int N,
                                                              int H, int W, int C) {
  const int cpp = C / 8;
  const int per_img = 2 * W + 2 * (H - 2);                 // top row, bottom row, left / right columns without the corners
  const long total = (long)N * per_img * cpp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % cpp);
    const long r = i / cpp;
    const int k = (int)(r % per_img), n = (int)(r / per_img);
    int Y, X;
    if (k < W) { Y = 0; X = k; }
    else if (k < 2 * W) { Y = H - 1; X = k - W; }
    else { const int j = k - 2 * W; Y = 1 + (j >> 1); X = (j & 1) ? W - 1 : 0; }
    bf16_t* dst = y + ((long)(n * H + Y) * W + X) * ldy + ch * 8;
    float v[8];
    Chunk<bf16_t>::unpack(ld16(dst), v);
    for (int t = 0; t < 9; ++t) {
      const int yy = Y + t / 3 - 1, xx = X + t % 3 - 1;
      if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) continue;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] -= btab[(1 + t) * C + ch * 8 + e];
    }
    st16(dst, Chunk<bf16_t>::pack(v));
  }
}

// ---- ConvT bias gradient without `dup`:  dbT[c] = sum_q dup[q, c] = sum_t sum_co W3[co][c][t] * S_t[co],
//      S_t[co] = sum of dY over the pixels p whose tap t stays inside the image = total - border rows / columns (+ corners)
// border sums of dY [N, H, W, C] (bf16): bs[q][n][c], q = 0 top row, 1 bottom row, 2 left column, 3 right column
__global__ __launch_bounds__(256) void upc_border_sums_kernel(const bf16_t* __restrict__ dy, int lddy, float* __restrict__ bs, int N,
                                                              int H, int W, int C) {
  __shared__ float red[256 * 8];
  const int q = blockIdx.x & 3, n = blockIdx.x >> 2, cpp = C / 8;
  const int tid = threadIdx.x, cl = tid % cpp, pl = tid / cpp, plane = 256 / cpp;      // cpp <= 64: C <= 512
  const int len = q < 2 ? W : H;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (pl < plane) {
    for (int i = pl; i < len; i += plane) {
      const int Y = q == 0 ? 0 : (q == 1 ? H - 1 : i), X = q == 2 ? 0 : (q == 3 ? W - 1 : i);
      float v[8];
      Chunk<bf16_t>::unpack(ld16(dy + ((long)(n * H + Y) * W + X) * lddy + cl * 8), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[tid * 8 + e] = s[e];
  __syncthreads();
  for (int i = tid; i < C; i += 256) {
    const int c2 = i / 8, e = i % 8;
    float t = 0.f;
    for (int k = 0; k < plane; ++k) t += red[(k * cpp + c2) * 8 + e];
    bs[((long)q * N + n) * C + i] = t;
  }
}

// S[t][co] = sum of dY over the pixels whose tap t stays inside the image (batch sums in image order: reproducible)
__global__ __launch_bounds__(256) void upc_tap_sums_kernel(const float* __restrict__ total, const float* __restrict__ bs,
                                                           const bf16_t* __restrict__ dy, int lddy, float* __restrict__ S, int N, int H,
                                                           int W, int C) {
  const int co = blockIdx.x * 256 + threadIdx.x;
  if (co >= C) return;
  float b[4] = {0.f, 0.f, 0.f, 0.f}, k[4] = {0.f, 0.f, 0.f, 0.f};       // border rows / columns, corners (00, 0L, L0, LL)
  for (int n = 0; n < N; ++n) {
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] += bs[((long)q * N + n) * C + co];
    k[0] += to_f32(dy[((long)(n * H) * W) * lddy + co]);
    k[1] += to_f32(dy[((long)(n * H) * W + W - 1) * lddy + co]);
    k[2] += to_f32(dy[((long)(n * H + H - 1) * W) * lddy + co]);
    k[3] += to_f32(dy[((long)(n * H + H - 1) * W + W - 1) * lddy + co]);
  }
  const float T = total[co];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int ky = t / 3, kx = t % 3;
    float v = T;
    if (ky == 0) v -= b[0];
    if (ky == 2) v -= b[1];
    if (kx == 0) v -= b[2];
    if (kx == 2) v -= b[3];
    if (ky == 0 && kx == 0) v += k[0];
    if (ky == 0 && kx == 2) v += k[1];
    if (ky == 2 && kx == 0) v += k[2];
    if (ky == 2 && kx == 2) v += k[3];
    S[t * C + co] = v;
  }
}

// one workgroup per ConvT output channel c:  dbT[c] = sum_t sum_co W3[co][c][t] * S[t][co]   (fixed-order tree)
__global__ __launch_bounds__(256) void upc_dbt_kernel(const float* __restrict__ w3, const float* __restrict__ S, float* __restrict__ dbt, int C) {
  __shared__ float red[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  float acc = 0.f;
  for (int co = tid; co < C; co += 256) {
    const float* w = w3 + ((long)co * 2 * C + c) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc = fmaf(w[t], S[t * C + co], acc);
  }
  red[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) dbt[c] = red[0];
}

// the ConvT bias reaches decN.0's weight gradient too: up[q, c] = bT[c] + ..., so dW3[co][c][t] += bT[c] * S[t][co]  (c < C)
__global__ __launch_bounds__(256) void upc_dw3_bias_kernel(float* __restrict__ dw3, const float* __restrict__ bt, const float* __restrict__ S, int C) {
  const long n = (long)C * C * 9;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int t = (int)(i % 9);
    const long r = i / 9;
    const int c = (int)(r % C), co = (int)(r / C);
    dw3[((long)co * 2 * C + c) * 9 + t] += bt[c] * S[t * C + co];
  }
}

// dw3[co][C + c][t] = tmp[co][c][t]: the skip half's weight gradient (a dense [C][C][9] result of the ordinary kernel) into its place
__global__ __launch_bounds__(256) void upc_skip_scatter_kernel(const float* __restrict__ tmp, float* __restrict__ dw3, int C) {
  const long n = (long)C * C * 9;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int r = (int)(i % (C * 9));                      // c * 9 + t
    const int co = (int)(i / (C * 9));
    dw3[((long)co * 2 * C + C) * 9 + r] = tmp[i];
  }
}

// bs_scratch: 4 N C floats (border sums) + 9 C floats (tap sums).  Adds the bias term to the up half of dw3 (which the decomposition
// GEMM has written) and writes dbt.
int launch_upc_dbt(const float* w3_master, const float* bt, const float* total, const void* dy, int lddy, float* bs_scratch, float* dw3,
                   float* dbt, int N, int H, int W, int C, hipStream_t stream) {
  UNETDC_REQUIRE(C % 8 == 0 && C <= 512 && 256 % (C / 8) == 0, "upconv dbT: C = %d unsupported", C);
  float* S = bs_scratch + 4L * N * C;
  hipLaunchKernelGGL(upc_border_sums_kernel, dim3(4 * N), dim3(256), 0, stream, reinterpret_cast<const bf16_t*>(dy), lddy, bs_scratch,
                     N, H, W, C);
  int rc = check_launch("upc_border_sums_kernel");
  if (rc != UNETDC_OK) return rc;
  hipLaunchKernelGGL(upc_tap_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, total, bs_scratch, reinterpret_cast<const bf16_t*>(dy),
                     lddy, S, N, H, W, C);
  rc = check_launch("upc_tap_sums_kernel");
  if (rc != UNETDC_OK) return rc;
  hipLaunchKernelGGL(upc_dbt_kernel, dim3(C), dim3(256), 0, stream, w3_master, S, dbt, C);
  rc = check_launch("upc_dbt_kernel");
  if (rc != UNETDC_OK) return rc;
  const long n = (long)C * C * 9;
  long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(upc_dw3_bias_kernel, dim3((unsigned)nb), dim3(256), 0, stream, dw3, bt, S, C);
  return check_launch("upc_dw3_bias_kernel");
}

int launch_upc_skip_scatter(const float* tmp, float* dw3, int C, hipStream_t stream) {
  const long n = (long)C * C * 9;
  long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(upc_skip_scatter_kernel, dim3((unsigned)nb), dim3(256), 0, stream, tmp, dw3, C);
  return check_launch("upc_skip_scatter_kernel");
}

// ------------------------------------------------------------------------------------------------
int launch_upc_compose(const void* w3_fwd, const void* w3_dgrad, const void* wt_dgrad, const float* w3_master, const float* b3,
                       const float* bt, void* wc_fwd, void* wc_dgrad, void* wskip_fwd, void* wskip_dgrad, float* btab, int C,
                       hipStream_t stream) {
  UNETDC_REQUIRE(C >= 64 && C % 64 == 0, "upconv compose: C = %d must be a multiple of 64", C);
  UNETDC_REQUIRE(w3_fwd && w3_dgrad && wt_dgrad && w3_master && bt && wc_fwd && wc_dgrad && wskip_fwd && wskip_dgrad && btab,
                 "upconv compose: null pointer");
  UpcGemmParams g{};
  g.a = reinterpret_cast<const bf16_t*>(w3_fwd); g.b = reinterpret_cast<const bf16_t*>(wt_dgrad);
  g.out0 = wc_fwd; g.out1 = wc_dgrad; g.C = C;
  hipLaunchKernelGGL((upc_gemm_kernel<0>), dim3(2 * C / 64, C / 64, 16), dim3(256), 0, stream, g);
  int rc = check_launch("upc_gemm_kernel<0>");
  if (rc != UNETDC_OK) return rc;
  const long n = 9L * C * C;
  hipLaunchKernelGGL(upc_skip_slices_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, stream,
                     reinterpret_cast<const bf16_t*>(w3_fwd), reinterpret_cast<const bf16_t*>(w3_dgrad),
                     reinterpret_cast<bf16_t*>(wskip_fwd), reinterpret_cast<bf16_t*>(wskip_dgrad), C);
  rc = check_launch("upc_skip_slices_kernel");
  if (rc != UNETDC_OK) return rc;
  hipLaunchKernelGGL(upc_bias_table_kernel, dim3(C), dim3(256), 0, stream, w3_master, b3, bt, btab, C);
  return check_launch("upc_bias_table_kernel");
}

int launch_upc_border_bias(void* y, int ldy, const float* btab, int N, int H, int W, int C, hipStream_t stream) {
  const long total = (long)N * (2 * W + 2 * (H - 2)) * (C / 8);
  long nb = (total + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(upc_border_bias_kernel, dim3((unsigned)nb), dim3(256), 0, stream, reinterpret_cast<bf16_t*>(y), ldy, btab, N, H, W, C);
  return check_launch("upc_border_bias_kernel");
}

// dW3[:, :C] and dWT from the 16 blocks of dW' (bf16 copies in both layouts, written by the weight-gradient reduction)
int launch_upc_decompose(const void* dwb, const void* dwbt, const void* wt_fwd, const void* w3_dgrad, float* dw3, float* dwt, int C,
                         hipStream_t stream) {
  UNETDC_REQUIRE(C >= 64 && C % 64 == 0 && dwb && dwbt && wt_fwd && w3_dgrad && dw3 && dwt, "upconv decompose: bad arguments");
  UpcGemmParams g{};
  g.C = C;
  g.a = reinterpret_cast<const bf16_t*>(dwb); g.b = reinterpret_cast<const bf16_t*>(wt_fwd); g.out0 = dw3;
  hipLaunchKernelGGL((upc_gemm_kernel<1>), dim3(C / 64, C / 64, 9), dim3(256), 0, stream, g);
  int rc = check_launch("upc_gemm_kernel<1>");
  if (rc != UNETDC_OK) return rc;
  g.a = reinterpret_cast<const bf16_t*>(dwbt); g.b = reinterpret_cast<const bf16_t*>(w3_dgrad); g.out0 = dwt;
  hipLaunchKernelGGL((upc_gemm_kernel<2>), dim3(C / 64, 2 * C / 64, 4), dim3(256), 0, stream, g);
  return check_launch("upc_gemm_kernel<2>");
}

}  // namespace unetdc
