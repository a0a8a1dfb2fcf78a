// First encoder convolution on the matrix cores (enc1.0: Cin = 1 or 3 -> 64, models/model_2.py:10,41-44).
//
// The VALU kernels of first_conv.hip are instruction-issue bound (229 us forward, 150 us weight gradient at
// 8 x 512 x 512: ~270 instructions per pixel-lane for 72 FMAs).  K = 9*Cin is short, but the fp32 MFMA
// v_mfma_f32_32x32x2_f32 takes K two at a time, so 9 (27) taps are 5 (14) instructions per 32 x 32 tile with
// the operands fed STRAIGHT FROM GLOBAL MEMORY -- no LDS staging, nothing to amortise:
//   forward : A[pixel][k] = x shifted by tap k (one fp32 load per lane and k-pair, zero outside the image),
//             B[k][cout]  = the PyTorch weights, loop-invariant, in registers;
//             epilogue    = igemm_epilogue.h (bias / BatchNorm statistics / folded BN+ReLU, channel pairs per lane);
//   wgrad   : dW[k][cout] = sum_p x_k[p] * dy[p][cout]: A[k][pixel] (lane r = tap r, gathers its shifted pixel),
//             B[pixel][cout] = one dword (bf16) / qword (fp32) of dy per lane = both 32-channel tiles at once.
// fp32 operands and fp32 accumulation in both dtypes (the first layer keeps fp32 input and weights).
#include <stdio.h>

#include "igemm_epilogue.h"
#include "kernels.h"

namespace unetdc {

constexpr int FM_TM = 4;                        // 32-pixel tiles per wave (forward)
constexpr int FM_BM = 4 * FM_TM * 32;           // pixels per workgroup = one row of partial statistics

template <typename T, int CIN>
__global__ __launch_bounds__(256) void first_mfma_fwd_kernel(const IgemmParams p, int dil) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int K = 9 * CIN, KP = (K + 1) / 2;
  constexpr int ES = (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) unsigned char red_s[4 * 4 * 32 * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const float* __restrict__ xg = reinterpret_cast<const float*>(p.x);
  const float* __restrict__ wg = reinterpret_cast<const float*>(p.w);
  const int HW = p.Ho * p.Wo;

  // B operand (loop invariant): k = 2q + h, output channel 2r + j  (tile j holds the even / odd channel of a pair)
  float bw[2][KP];
  int kdy[KP], kdx[KP], kci[KP];
  bool kok[KP];
#pragma unroll
  for (int q = 0; q < KP; ++q) {
    const int k = 2 * q + h;
    kok[q] = k < K;
    const int kk = kok[q] ? k : 0;
    const int ci = kk / 9, tap = kk - ci * 9;
    kci[q] = ci;
    kdy[q] = (tap / 3 - 1) * dil;
    kdx[q] = (tap % 3 - 1) * dil;
#pragma unroll
    for (int j = 0; j < 2; ++j) bw[j][q] = kok[q] ? wg[(2 * r + j) * K + kk] : 0.f;
  }

  f32x16 acc[FM_TM][2];
#pragma unroll
  for (int i = 0; i < FM_TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int m0 = blockIdx.x * FM_BM + wave * FM_TM * 32;
#pragma unroll
  for (int mi = 0; mi < FM_TM; ++mi) {
    const int m = m0 + mi * 32 + r;
    float a[KP];
    if (m < p.M) {
      int n, y, x;
      if (p.wo_shift >= 0) {
        n = m >> p.howo_shift;
        const int rem = m & (HW - 1);
        y = rem >> p.wo_shift;
        x = rem & (p.Wo - 1);
      } else {
        n = m / HW;
        const int rem = m - n * HW;
        y = rem / p.Wo;
        x = rem - y * p.Wo;
      }
#pragma unroll
      for (int q = 0; q < KP; ++q) {
        const int iy = y + kdy[q], ix = x + kdx[q];
        const bool ok = kok[q] && (unsigned)iy < (unsigned)p.Ho && (unsigned)ix < (unsigned)p.Wo;
        a[q] = ok ? xg[((long)(n * CIN + kci[q]) * p.Ho + iy) * p.Wo + ix] : 0.f;
      }
    } else {
#pragma unroll
      for (int q = 0; q < KP; ++q) a[q] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < KP; ++q)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], bw[j][q], acc[mi][j], 0, 0, 0);
  }

  const int col = 2 * r;
  const unsigned ldob = (unsigned)(p.ldo * ES);
  bool tile_ok[FM_TM];
  unsigned voff[FM_TM], yoff[FM_TM];
#pragma unroll
  for (int mi = 0; mi < FM_TM; ++mi) {
    const int mb = m0 + mi * 32;
    tile_ok[mi] = mb < p.M;
    voff[mi] = (unsigned)(mb + 4 * h) * ldob + (unsigned)(col * ES);
    yoff[mi] = 0;
  }
  float st[4] = {0.f, 0.f, 0.f, 0.f};
  switch (p.mode) {
    case MODE_STATS: epilogue_tiles<T, MODE_STATS, FM_TM>(p, acc, tile_ok, voff, ldob, yoff, 0u, col, st); break;
    case MODE_AFFINE_RELU: epilogue_tiles<T, MODE_AFFINE_RELU, FM_TM>(p, acc, tile_ok, voff, ldob, yoff, 0u, col, st); break;
    default: epilogue_tiles<T, MODE_STORE, FM_TM>(p, acc, tile_ok, voff, ldob, yoff, 0u, col, st); break;
  }
  if (p.mode == MODE_STATS) write_stat_rows<4, 1>(p, red_s, st, blockIdx.x, 0, tid, wave, r, h);
#endif
}

bool first_mfma_supported(long P, int Cin, int Cout) {
  static int off = -1;                                   // UNETDC_FIRST=valu: the VALU kernels of first_conv.hip (A/B)
  if (off < 0) { const char* e = getenv("UNETDC_FIRST"); off = (e && e[0] == 'v') ? 1 : 0; }
  return !off && Cout == 64 && (Cin == 1 || Cin == 3) && P % 32 == 0 && P < (1L << 23);
}
int first_mfma_mblocks(long P) { return ceil_div(P, FM_BM); }

int launch_first_mfma_fwd(FirstParams& f, int dtype, hipStream_t stream) {
  IgemmParams p{};
  p.x = f.x; p.w = f.w; p.out = f.y; p.bias = f.bias; p.scale = f.scale; p.shift = f.shift; p.stats = f.stats;
  p.M = f.N * f.H * f.W; p.Ho = f.H; p.Wo = f.W; p.Hi = f.H; p.Wi = f.W; p.Cin = f.Cin; p.Cout = f.Cout; p.ldo = f.ldy;
  p.mode = f.scale ? MODE_AFFINE_RELU : (f.stats ? MODE_STATS : MODE_STORE);
  const long howo = (long)f.H * f.W;
  const bool p2 = (f.W & (f.W - 1)) == 0 && (howo & (howo - 1)) == 0;
  p.wo_shift = p2 ? __builtin_ctz((unsigned)f.W) : -1;
  p.howo_shift = p2 ? __builtin_ctzl((unsigned long)howo) : -1;
  UNETDC_REQUIRE((long)p.M * f.ldy * (dtype == UNETDC_BF16 ? 2 : 4) < (1L << 32), "first_conv: output too large");
  const dim3 grid((unsigned)first_mfma_mblocks(p.M)), block(256);
  if (dtype == UNETDC_BF16) {
    if (f.Cin == 1) hipLaunchKernelGGL((first_mfma_fwd_kernel<bf16_t, 1>), grid, block, 0, stream, p, f.dil);
    else hipLaunchKernelGGL((first_mfma_fwd_kernel<bf16_t, 3>), grid, block, 0, stream, p, f.dil);
  } else {
    if (f.Cin == 1) hipLaunchKernelGGL((first_mfma_fwd_kernel<float, 1>), grid, block, 0, stream, p, f.dil);
    else hipLaunchKernelGGL((first_mfma_fwd_kernel<float, 3>), grid, block, 0, stream, p, f.dil);
  }
  return check_launch("first_mfma_fwd_kernel");
}

// ---- weight gradient ---------------------------------------------------------------------------------------
// part[blk][k][cout] (k = ci*9 + tap), reduced over blocks by first_wgrad_reduce_kernel (first_conv.hip)
template <typename T> struct DyPair;
template <> struct DyPair<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* q, float& a, float& b) {
    const unsigned int u = *reinterpret_cast<const unsigned int*>(q);
    a = bits_f32(u << 16);
    b = bits_f32(u & 0xffff0000u);
  }
};
template <> struct DyPair<float> {
  static __device__ __forceinline__ void load(const float* q, float& a, float& b) {
    const float2 v = *reinterpret_cast<const float2*>(q);
    a = v.x;
    b = v.y;
  }
};

constexpr int FW_UNROLL = 8;                    // pixel pairs in flight per wave

template <typename T, int CIN>
__global__ __launch_bounds__(256) void first_mfma_wgrad_kernel(const FirstWgradParams p, int pairs_per_wave, int wo_shift,
                                                               int howo_shift) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int K = 9 * CIN;
  __shared__ float red[4][K][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int HW = p.H * p.W;
  const long P = (long)p.N * HW;
  const T* __restrict__ dyg = reinterpret_cast<const T*>(p.dy);
  // A operand: lane r feeds row k = r of the [k][pixel] matrix
  const bool kok = r < K;
  const int kk = kok ? r : 0;
  const int ci = kk / 9, tap = kk - ci * 9;
  const int dyo = (tap / 3 - 1) * p.dil, dxo = (tap % 3 - 1) * p.dil;

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  // pairs_per_wave is a multiple of FW_UNROLL (host); every load is UNCONDITIONAL (clamped address, value zeroed
  // afterwards) so that the 16 loads of an iteration are issued back to back instead of one wait per branch
  const long pair0 = ((long)blockIdx.x * 4 + wave) * pairs_per_wave;
  for (int it = 0; it < pairs_per_wave; it += FW_UNROLL) {
    float a[FW_UNROLL], b0[FW_UNROLL], b1[FW_UNROLL];
    bool av[FW_UNROLL], bv[FW_UNROLL];
#pragma unroll
    for (int u = 0; u < FW_UNROLL; ++u) {
      const long m = (pair0 + it + u) * 2 + h;
      bv[u] = m < P;
      const long mc = bv[u] ? m : P - 1;
      DyPair<T>::load(dyg + mc * p.lddy + 2 * r, b0[u], b1[u]);
      int n, y, x;
      if (wo_shift >= 0) {
        n = (int)(mc >> howo_shift);
        const int rem = (int)(mc & (HW - 1));
        y = rem >> wo_shift;
        x = rem & (p.W - 1);
      } else {
        n = (int)(mc / HW);
        const int rem = (int)(mc - (long)n * HW);
        y = rem / p.W;
        x = rem - y * p.W;
      }
      const int iy = y + dyo, ix = x + dxo;
      av[u] = bv[u] && kok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);
      a[u] = p.x[((long)(n * CIN + ci) * p.H + iyc) * p.W + ixc];
    }
#pragma unroll
    for (int u = 0; u < FW_UNROLL; ++u) {
      const float au = av[u] ? a[u] : 0.f;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(au, bv[u] ? b0[u] : 0.f, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(au, bv[u] ? b1[u] : 0.f, acc[1], 0, 0, 0);
    }
  }
  // rows k = (reg&3) + 8*(reg>>2) + 4h, column cout = 2r + j; four waves added in wave order
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int k = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (k < K) red[wave][k][2 * r + j] = acc[j][reg];
    }
  __syncthreads();
  for (int i = tid; i < K * 64; i += 256) {
    const int k = i >> 6, co = i & 63;
    p.part[((long)blockIdx.x * K + k) * 64 + co] = (red[0][k][co] + red[1][k][co]) + (red[2][k][co] + red[3][k][co]);
  }
#endif
}

static int first_mfma_wgrad_blocks(long P) {
  long nb = (P / 2 + 4 * 64 - 1) / (4 * 64);          // >= 64 pixel pairs per wave
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return (int)nb;
}
long first_mfma_wgrad_workspace_bytes(long P, int Cin, int Cout) { return (long)first_mfma_wgrad_blocks(P) * Cin * 9 * Cout * 4; }

int launch_first_mfma_wgrad(FirstWgradParams& p, int* nblk_out, int dtype, hipStream_t stream) {
  const long P = (long)p.N * p.H * p.W;
  const int nb = first_mfma_wgrad_blocks(P);
  const long pairs = (P + 1) / 2;
  int ppw = (int)((pairs + (long)nb * 4 - 1) / ((long)nb * 4));
  ppw = (ppw + FW_UNROLL - 1) / FW_UNROLL * FW_UNROLL;
  const long howo = (long)p.H * p.W;
  const bool p2 = (p.W & (p.W - 1)) == 0 && (howo & (howo - 1)) == 0;
  const int ws = p2 ? __builtin_ctz((unsigned)p.W) : -1, hs = p2 ? __builtin_ctzl((unsigned long)howo) : -1;
  const dim3 grid((unsigned)nb), block(256);
  if (dtype == UNETDC_BF16) {
    if (p.Cin == 1) hipLaunchKernelGGL((first_mfma_wgrad_kernel<bf16_t, 1>), grid, block, 0, stream, p, ppw, ws, hs);
    else hipLaunchKernelGGL((first_mfma_wgrad_kernel<bf16_t, 3>), grid, block, 0, stream, p, ppw, ws, hs);
  } else {
    if (p.Cin == 1) hipLaunchKernelGGL((first_mfma_wgrad_kernel<float, 1>), grid, block, 0, stream, p, ppw, ws, hs);
    else hipLaunchKernelGGL((first_mfma_wgrad_kernel<float, 3>), grid, block, 0, stream, p, ppw, ws, hs);
  }
  *nblk_out = nb;
  return check_launch("first_mfma_wgrad_kernel");
}

}  // namespace unetdc
