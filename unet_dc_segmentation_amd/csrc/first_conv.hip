// First encoder convolution (enc1.0: Cin = 1 or 3 -> 64, models/model_2.py:10,41-44).
//
// K = 9*Cin is far too short for the matrix cores and the layer is HBM-bound on its OUTPUT
// (arithmetic intensity 4.4 FLOP/B), so this is a direct VALU kernel:
//   * reads the caller's NCHW fp32 image as it is (no layout pass; a C=1 image is already NHWC),
//   * 8 consecutive lanes own one pixel and write its 64 channels as 8 x 16-byte chunks
//     = one contiguous 128/256-byte NHWC row,
//   * weights live in LDS as [tap*Cin + ci][Cout] fp32, read as two ds_read_b128 per tap,
//   * BatchNorm batch statistics (sum, sum of squares of the stored values) are reduced per block.
// The matching weight gradient streams dY once per input channel and keeps the 9 x 8 products per
// lane in registers; block partials are summed by a second deterministic kernel.
#include <stdint.h>
#include <stdlib.h>

#include "kernels.h"

namespace unetdc {

template <typename T>
__global__ __launch_bounds__(256) void first_conv_fwd_kernel(const FirstParams p) {
  extern __shared__ __attribute__((aligned(16))) float wl[];     // [9*Cin][Cout] + reduction scratch
  const int tid = threadIdx.x;
  const int G = p.Cout / 8;                 // lanes per pixel
  const int PPB = 256 / G;                  // pixels per block iteration
  const int K = 9 * p.Cin;
  for (int i = tid; i < K * p.Cout; i += 256) {
    const int co = i % p.Cout, k = i / p.Cout;          // k = tap*Cin + ci
    const int tap = k / p.Cin, ci = k - tap * p.Cin;
    wl[i] = p.w[(co * p.Cin + ci) * 9 + tap];
  }
  __syncthreads();
  const int g = tid % G, pl = tid / G;
  const int HW = p.H * p.W;
  const long P = (long)p.N * HW;
  float k0[8], k1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int co = g * 8 + e;
    k0[e] = p.scale ? p.scale[co] : 1.f;
    k1[e] = p.scale ? p.shift[co] : (p.bias ? p.bias[co] : 0.f);
  }
  float su[8], sq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) su[e] = sq[e] = 0.f;
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);

  for (long pb = (long)blockIdx.x * PPB; pb < P; pb += (long)gridDim.x * PPB) {
    const long pix = pb + pl;
    if (pl < PPB && pix < P) {
      const int n = (int)(pix / HW), rem = (int)(pix - (long)n * HW);
      const int y = rem / p.W, x = rem - y * p.W;
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
      for (int tap = 0; tap < 9; ++tap) {
        const int iy = y + (tap / 3 - 1) * p.dil, ix = x + (tap % 3 - 1) * p.dil;
        if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.W) continue;
        for (int ci = 0; ci < p.Cin; ++ci) {
          const float xv = p.x[((long)(n * p.Cin + ci) * p.H + iy) * p.W + ix];
          const float* wr = wl + (tap * p.Cin + ci) * p.Cout + g * 8;
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr), w1 = *reinterpret_cast<const f32x4*>(wr + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[e] = fmaf(xv, w0[e], acc[e]);
            acc[4 + e] = fmaf(xv, w1[e], acc[4 + e]);
          }
        }
      }
      float out[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = p.scale ? fmaxf(fmaf(acc[e], k0[e], k1[e]), 0.f) : acc[e] + k1[e];
        out[e] = round_through<T>(v);              // statistics of the values as stored
        su[e] += out[e];
        sq[e] = fmaf(out[e], out[e], sq[e]);
      }
      T* dst = yg + pix * p.ldy + g * 8;
#pragma unroll
      for (int q = 0; q < 8 / Chunk<T>::N; ++q) st16(dst + q * Chunk<T>::N, Chunk<T>::pack(out + q * Chunk<T>::N));
    }
  }
  if (p.stats) {
    // block reduction over the PPB pixel lanes that share a channel group (fixed order)
    __syncthreads();
    float* red = wl + K * p.Cout;                     // [256][16]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[tid * 16 + e] = su[e];
      red[tid * 16 + 8 + e] = sq[e];
    }
    __syncthreads();
    if (tid < p.Cout * 2) {
      const int which = tid / p.Cout, co = tid - which * p.Cout;
      const int gg = co / 8, e = co % 8;
      float s = 0.f;
      for (int q = 0; q < PPB; ++q) s += red[(q * G + gg) * 16 + which * 8 + e];
      p.stats[((long)blockIdx.x * 2 + which) * p.Cout + co] = s;
    }
  }
}

// ---- weight gradient of the first layer -------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void first_conv_wgrad_kernel(const FirstWgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float red[];    // [4 waves][G][72]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = p.Cout / 8, PPB = 256 / G;
  const int g = tid % G, pl = tid / G;
  const int ci = blockIdx.y;
  const int HW = p.H * p.W;
  const long P = (long)p.N * HW;
  const T* __restrict__ dyg = reinterpret_cast<const T*>(p.dy);
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;

  for (long pb = (long)blockIdx.x * PPB; pb < P; pb += (long)gridDim.x * PPB) {
    const long pix = pb + pl;
    if (pl < PPB && pix < P) {
      const int n = (int)(pix / HW), rem = (int)(pix - (long)n * HW);
      const int y = rem / p.W, x = rem - y * p.W;
      float d[8];
#pragma unroll
      for (int q = 0; q < 8 / Chunk<T>::N; ++q)
        Chunk<T>::unpack(ld16(dyg + pix * p.lddy + g * 8 + q * Chunk<T>::N), d + q * Chunk<T>::N);
      const float* xp = p.x + (long)(n * p.Cin + ci) * HW;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = y + (t / 3 - 1) * p.dil, ix = x + (t % 3 - 1) * p.dil;
        float xv = 0.f;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) xv = xp[iy * p.W + ix];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = fmaf(xv, d[e], acc[t][e]);
      }
    }
  }
  // reduce over the pixel lanes of a wave that share g (lanes g, g+G, g+2G, ...), fixed xor tree
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = acc[t][e];
      for (int o = G; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
      acc[t][e] = v;
    }
  if (lane < G) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(wave * G + lane) * 72 + t * 8 + e] = acc[t][e];
  }
  __syncthreads();
  for (int i = tid; i < 9 * p.Cout; i += 256) {
    const int t = i / p.Cout, co = i - t * p.Cout, gg = co / 8, e = co % 8;
    float s = 0.f;
    for (int w2 = 0; w2 < 4; ++w2) {
      s += red[(w2 * G + gg) * 72 + t * 8 + e];
      asm volatile("" : "+v"(s));     // opaque: the loop is unrolled by two and the two sums would pair into v_pk_add_f32 (no packed fp32 in this library, build.py)
    }
    p.part[(((long)blockIdx.x * p.Cin + ci) * 9 + t) * p.Cout + co] = s;
  }
}

// Row-run form (dilation 1, W % 8 == 0: the first layer of both networks).  SQ/TA view of the kernel above: 9 bounds-checked
// 4-byte input loads and ~50 address instructions per 16 bytes of dY, 512 workgroups -> 1.5 TB/s on a pass whose only real
// traffic is dY.  Here a lane owns 8 channels and walks RUNS of 8 consecutive pixels of one image row: the 3 x 10 input window
// of a run is loaded once (two float4 + two edge values per row: 12 loads per 8 pixels instead of 72) and slides through
// registers; the eight dY chunks of a run are issued together.
// BN: the BatchNorm + ReLU backward of the stage on load (FirstWgradParams::bn_*): the stand-alone pass that would read y and
// the activation gradient and write dy -- three activation-sized transfers at full resolution -- is not run at all; this
// kernel reads y next to the gradient instead (one more), two pixels at a time to stay inside the register file.
template <typename T, bool BN>
__global__ __launch_bounds__(256) void first_wgrad_rows_kernel(const FirstWgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float red[];    // [4 waves][G][72]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = p.Cout / 8, PPB = 256 / G;
  const int g = tid % G, pl = tid / G;
  const int ci = blockIdx.y;
  const int HW = p.H * p.W, rpr = p.W / 8;
  const long runs = (long)p.N * p.H * rpr;
  const T* __restrict__ dyg = reinterpret_cast<const T*>(p.dy);
  const T* __restrict__ yg = reinterpret_cast<const T*>(p.bn_y);
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
  float sc[8], sh[8], mu[8], rs[8], k1[8], k2[8], k3[8];
  if (BN) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = g * 8 + e;
      sc[e] = p.bn_scale[c]; sh[e] = p.bn_shift[c]; mu[e] = p.bn_mean[c]; rs[e] = p.bn_rstd[c];
      k1[e] = p.bn_k[c]; k2[e] = p.bn_k[p.Cout + c]; k3[e] = p.bn_k[2 * p.Cout + c];
    }
  }

  for (long r = (long)blockIdx.x * PPB + pl; r < runs; r += (long)gridDim.x * PPB) {
    const int n = (int)(r / ((long)p.H * rpr));
    const int rem = (int)(r - (long)n * p.H * rpr);
    const int y = rem / rpr, x0 = (rem - y * rpr) * 8;
    const float* xp = p.x + (long)(n * p.Cin + ci) * HW;
    float xw[3][10];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = y + ky - 1;
      const bool ok = (unsigned)iy < (unsigned)p.H;
      const float* row = xp + (long)(ok ? iy : y) * p.W + x0;
      const float4 a = *reinterpret_cast<const float4*>(row), b = *reinterpret_cast<const float4*>(row + 4);
      const float l = x0 > 0 ? row[-1] : 0.f, rr = x0 + 8 < p.W ? row[8] : 0.f;
      const float m = ok ? 1.f : 0.f;                        // rows outside the image contribute zeros
      xw[ky][0] = l * m; xw[ky][1] = a.x * m; xw[ky][2] = a.y * m; xw[ky][3] = a.z * m; xw[ky][4] = a.w * m;
      xw[ky][5] = b.x * m; xw[ky][6] = b.y * m; xw[ky][7] = b.z * m; xw[ky][8] = b.w * m; xw[ky][9] = rr * m;
    }
    const long pix0 = ((long)n * p.H + y) * p.W + x0;
    constexpr int PB = BN ? 2 : 4;                         // pixels' chunks in flight at a time (register budget)
#pragma unroll
    for (int hf = 0; hf < 8 / PB; ++hf) {
      float d[PB][8];
#pragma unroll
      for (int i = 0; i < PB; ++i)
#pragma unroll
        for (int q = 0; q < 8 / Chunk<T>::N; ++q)
          Chunk<T>::unpack(ld16(dyg + (pix0 + PB * hf + i) * p.lddy + g * 8 + q * Chunk<T>::N), d[i] + q * Chunk<T>::N);
      if (BN) {
        float yv[PB][8];
#pragma unroll
        for (int i = 0; i < PB; ++i)
#pragma unroll
          for (int q = 0; q < 8 / Chunk<T>::N; ++q)
            Chunk<T>::unpack(ld16(yg + (pix0 + PB * hf + i) * p.bn_ldy + g * 8 + q * Chunk<T>::N), yv[i] + q * Chunk<T>::N);
#pragma unroll
        for (int i = 0; i < PB; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) {                      // bn_bwd_kernel's apply arithmetic, operation for operation
            const float nrm = fmaf(yv[i][e], sc[e], sh[e]);
            const float gh = nrm > 0.f ? d[i][e] : 0.f;
            const float xh = (yv[i][e] - mu[e]) * rs[e];
            d[i][e] = round_through<T>(fmaf(k1[e], gh, -k2[e]) - k3[e] * xh);
          }
      }
#pragma unroll
      for (int i = 0; i < PB; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float xv = xw[t / 3][PB * hf + i + t % 3];
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[t][e] = fmaf(xv, d[i][e], acc[t][e]);
        }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = acc[t][e];
      for (int o = G; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
      acc[t][e] = v;
    }
  if (lane < G) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(wave * G + lane) * 72 + t * 8 + e] = acc[t][e];
  }
  __syncthreads();
  for (int i = tid; i < 9 * p.Cout; i += 256) {
    const int t = i / p.Cout, co = i - t * p.Cout, gg = co / 8, e = co % 8;
    float s = 0.f;
    for (int w2 = 0; w2 < 4; ++w2) {
      s += red[(w2 * G + gg) * 72 + t * 8 + e];
      asm volatile("" : "+v"(s));     // opaque: the loop is unrolled by two and the two sums would pair into v_pk_add_f32 (no packed fp32 in this library, build.py)
    }
    p.part[(((long)blockIdx.x * p.Cin + ci) * 9 + t) * p.Cout + co] = s;
  }
}

// dw[co][ci][t] = sum_blk part[blk][ci][t][co]
// 32 lanes per output: lane l adds blocks l, l+32, ... in order, then a fixed shuffle tree (deterministic).  One thread
// per output walking all 512 block partials took 120 us -- pure load latency on three workgroups.
__global__ __launch_bounds__(256) void first_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                 int nblk, int Cin, int Cout) {
  const int n = Cin * 9 * Cout;
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  float s = 0.f;
  if (i < n)
    for (int b = l; b < nblk; b += 32) s += part[(long)b * n + i];
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
  if (i < n && l == 0) {
    const int co = i % Cout, k = i / Cout;      // k = ci*9 + t
    dw[(long)co * Cin * 9 + k] = s;
  }
}

static int first_blocks(long P, int Cout) {
  const int ppb = 256 / (Cout / 8);
  long nb = (P + ppb - 1) / ppb;
  if (nb > 2048) nb = 2048;
  return (int)nb;
}

int first_conv_mblocks(long P, int Cin, int Cout) {
  return first_mfma_supported(P, Cin, Cout) ? first_mfma_mblocks(P) : first_blocks(P, Cout);
}

int launch_first_fwd(FirstParams& p, int dtype, hipStream_t stream) {
  UNETDC_REQUIRE(dtype == UNETDC_F32 || dtype == UNETDC_BF16, "first_conv: bad dtype %d", dtype);
  UNETDC_REQUIRE(p.x && p.w && p.y, "first_conv: null pointer");
  UNETDC_REQUIRE(p.Cin >= 1 && p.Cin <= 8, "first_conv: Cin=%d unsupported (1..8)", p.Cin);
  UNETDC_REQUIRE(p.Cout % 8 == 0 && p.Cout >= 8 && p.Cout / 8 <= 16 && 64 % (p.Cout / 8) == 0,
                 "first_conv: Cout=%d unsupported (8,16,32,64,128)", p.Cout);
  UNETDC_REQUIRE(p.ldy % 8 == 0 && (uintptr_t)p.y % 16 == 0, "first_conv: output not 16-byte aligned");
  const long P = (long)p.N * p.H * p.W;
  if (first_mfma_supported(P, p.Cin, p.Cout)) return launch_first_mfma_fwd(p, dtype, stream);     // first_conv_mfma.hip
  const int nb = first_blocks(P, p.Cout);
  const size_t lds = (size_t)(9 * p.Cin * p.Cout + 256 * 16) * 4;
  if (dtype == UNETDC_BF16)
    hipLaunchKernelGGL(first_conv_fwd_kernel<bf16_t>, dim3(nb), dim3(256), lds, stream, p);
  else
    hipLaunchKernelGGL(first_conv_fwd_kernel<float>, dim3(nb), dim3(256), lds, stream, p);
  return check_launch("first_conv_fwd_kernel");
}

// The MFMA weight-gradient kernel takes all 9*Cin taps in ONE pass over dY; the VALU kernel needs one pass per input
// channel but is as fast for Cin = 1 (measured 155 vs 169 us at 8 x 512 x 512), so the MFMA kernel serves Cin = 3.
static bool first_wgrad_mfma(long P, int Cin, int Cout) { return Cin > 1 && first_mfma_supported(P, Cin, Cout); }

constexpr long FIRST_ROWS_MAXB = 512;     // measured at 8 x 512 x 512: 2048 -> 122 us, 1024 -> 106, 512 -> 79, 256 -> 98
static long first_rows_blocks(long P, int Cout) {          // row-run kernel: one run of 8 pixels per lane group and trip
  const long ppb = 256 / (Cout / 8);
  long nb = (P / 8 + ppb - 1) / ppb;
  return nb > FIRST_ROWS_MAXB ? FIRST_ROWS_MAXB : (nb < 1 ? 1 : nb);
}

// One input channel only: the row-run kernel makes one pass over dY PER input channel, and with the BatchNorm backward on load
// every pass would also re-read y.
bool first_wgrad_bn_supported(int N, int H, int W, int Cin, int Cout, int dil, int dtype) {
  static int off = -1;                                   // UNETDC_FIRST_ROWS=0 also turns this form off (it is the row-run kernel)
  if (off < 0) { const char* e = getenv("UNETDC_FIRST_ROWS"); off = (e && e[0] == '0') ? 1 : 0; }
  (void)N; (void)H;
  return !off && (dtype == UNETDC_F32 || dtype == UNETDC_BF16) && Cin == 1 && dil == 1 && W % 8 == 0 && Cout % 8 == 0 &&
         Cout / 8 <= 16 && 64 % (Cout / 8) == 0;
}

long first_wgrad_workspace_bytes(long P, int Cin, int Cout) {
  if (first_wgrad_mfma(P, Cin, Cout)) return first_mfma_wgrad_workspace_bytes(P, Cin, Cout);
  long nb = first_blocks(P, Cout);
  if (nb > 512) nb = 512;
  const long nr = first_rows_blocks(P, Cout);            // whichever VALU kernel the launch picks (dilation / W decide there)
  return (nb > nr ? nb : nr) * Cin * 9 * Cout * 4;
}

// ------------------------------------------------------------------------------------------------
// Gradient with respect to the INPUT image of the first convolution (dL/dx of UNetDC.forward: the reference module supports
// it through plain autograd, models/model_2.py:56-80; its training loops never ask for it):
//   dx[n][ci][y][x] = sum_{co, tap} w[co][ci][tap] * dy[n][y - oy(tap)][x - ox(tap)][co]
// HBM-bound like the forward: Cout/8 consecutive lanes own one pixel, each reads its 16-byte channel chunk of the (up to) 9
// shifted dY pixels (neighbouring pixels re-read the same lines out of L1/L2), multiplies with the weights held in LDS as
// [tap][ci][Cout] fp32 and the lanes of a pixel are summed with cross-lane shuffles; output NCHW fp32 like the input.
// ------------------------------------------------------------------------------------------------
template <typename T, int CIN>
__global__ __launch_bounds__(256) void first_dgrad_kernel(const FirstWgradParams p, float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) float wl[];     // [9][CIN][Cout]
  constexpr int EPC = Chunk<T>::N;
  const int tid = threadIdx.x;
  const int G = p.Cout / EPC;               // lanes per pixel (power of two <= 64)
  const int PPB = 256 / G;
  for (int i = tid; i < 9 * CIN * p.Cout; i += 256) {
    const int co = i % p.Cout, k = i / p.Cout, ci = k % CIN, tap = k / CIN;
    wl[i] = reinterpret_cast<const float*>(p.part)[(co * CIN + ci) * 9 + tap];     // p.part carries the fp32 weight pointer here
  }
  __syncthreads();
  const int g = tid % G, pl = tid / G;
  const int HW = p.H * p.W;
  const long P = (long)p.N * HW;
  const T* __restrict__ dyg = reinterpret_cast<const T*>(p.dy);
  for (long pb = (long)blockIdx.x * PPB; pb < P; pb += (long)gridDim.x * PPB) {
    const long pix = pb + pl;
    const bool ok = pix < P;
    const long px = ok ? pix : 0;
    const int n = (int)(px / HW), rem = (int)(px - (long)n * HW);
    const int y = rem / p.W, x = rem - y * p.W;
    float acc[CIN];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
      const int iy = y - (tap / 3 - 1) * p.dil, ix = x - (tap % 3 - 1) * p.dil;      // the output pixel this tap came from
      if (!ok || (unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.W) continue;
      float v[EPC];
      Chunk<T>::unpack(ld16(dyg + ((long)(n * p.H + iy) * p.W + ix) * p.lddy + g * EPC), v);
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        const float* wr = wl + (tap * CIN + ci) * p.Cout + g * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[ci] = fmaf(v[e], wr[e], acc[ci]);
      }
    }
    for (int o = 1; o < G; o <<= 1)
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) acc[ci] += __shfl_xor(acc[ci], o, 64);
    if (ok && g == 0) {
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) dx[((long)(n * CIN + ci) * p.H + y) * p.W + x] = acc[ci];
    }
  }
}

int launch_first_dgrad(const void* dy, int lddy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout, int dil,
                       int dtype, hipStream_t stream) {
  UNETDC_REQUIRE(dtype == UNETDC_F32 || dtype == UNETDC_BF16, "first_dgrad: bad dtype %d", dtype);
  UNETDC_REQUIRE(dy && w && dx && N > 0 && H > 0 && W > 0 && dil >= 1, "first_dgrad: bad arguments");
  UNETDC_REQUIRE(Cin == 1 || Cin == 3, "first_dgrad: Cin=%d unsupported (1 or 3)", Cin);
  const int epc = dtype == UNETDC_BF16 ? 8 : 4;
  const int G = Cout / epc;
  UNETDC_REQUIRE(Cout % epc == 0 && G >= 1 && G <= 64 && (G & (G - 1)) == 0 && lddy % epc == 0 && lddy >= Cout,
                 "first_dgrad: Cout=%d / lddy=%d unsupported", Cout, lddy);
  FirstWgradParams p{};
  p.dy = dy; p.part = const_cast<float*>(w); p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.lddy = lddy; p.dil = dil;
  const long P = (long)N * H * W;
  long nb = (P + (256 / G) * 4 - 1) / ((256 / G) * 4);
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  const size_t lds = (size_t)9 * Cin * Cout * 4;
  if (dtype == UNETDC_BF16) {
    if (Cin == 1) hipLaunchKernelGGL((first_dgrad_kernel<bf16_t, 1>), dim3((unsigned)nb), dim3(256), lds, stream, p, dx);
    else hipLaunchKernelGGL((first_dgrad_kernel<bf16_t, 3>), dim3((unsigned)nb), dim3(256), lds, stream, p, dx);
  } else {
    if (Cin == 1) hipLaunchKernelGGL((first_dgrad_kernel<float, 1>), dim3((unsigned)nb), dim3(256), lds, stream, p, dx);
    else hipLaunchKernelGGL((first_dgrad_kernel<float, 3>), dim3((unsigned)nb), dim3(256), lds, stream, p, dx);
  }
  return check_launch("first_dgrad_kernel");
}

int launch_first_wgrad(FirstWgradParams& p, float* dw, void* workspace, long workspace_bytes, int dtype,
                       hipStream_t stream) {
  UNETDC_REQUIRE(dtype == UNETDC_F32 || dtype == UNETDC_BF16, "first_wgrad: bad dtype %d", dtype);
  UNETDC_REQUIRE(p.x && p.dy && dw && workspace, "first_wgrad: null pointer");
  UNETDC_REQUIRE(p.Cin >= 1 && p.Cin <= 8, "first_wgrad: Cin=%d unsupported", p.Cin);
  UNETDC_REQUIRE(p.Cout % 8 == 0 && p.Cout / 8 <= 16 && 64 % (p.Cout / 8) == 0, "first_wgrad: Cout=%d unsupported", p.Cout);
  const long P = (long)p.N * p.H * p.W;
  if (p.bn_y) {                                          // BatchNorm backward on load: the row-run kernel only
    UNETDC_REQUIRE(p.bn_scale && p.bn_shift && p.bn_mean && p.bn_rstd && p.bn_k, "first_wgrad (bn): null pointer");
    UNETDC_REQUIRE(first_wgrad_bn_supported(p.N, p.H, p.W, p.Cin, p.Cout, p.dil, dtype) && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0 &&
                       p.bn_ldy % (dtype == UNETDC_BF16 ? 8 : 4) == 0,
                   "first_wgrad (bn): shape not supported (ask unetdc_conv3x3_first_wgrad_bn_supported)");
    const long nr = first_rows_blocks(P, p.Cout);
    const long need_r = nr * p.Cin * 9 * p.Cout * 4;
    if (need_r > workspace_bytes) {
      set_error("first_wgrad (bn): workspace too small (%ld < %ld bytes)", workspace_bytes, need_r);
      return UNETDC_EWORKSPACE;
    }
    p.part = reinterpret_cast<float*>(workspace);
    const size_t lds_r = (size_t)4 * (p.Cout / 8) * 72 * 4;
    if (dtype == UNETDC_BF16)
      hipLaunchKernelGGL((first_wgrad_rows_kernel<bf16_t, true>), dim3((unsigned)nr, p.Cin), dim3(256), lds_r, stream, p);
    else
      hipLaunchKernelGGL((first_wgrad_rows_kernel<float, true>), dim3((unsigned)nr, p.Cin), dim3(256), lds_r, stream, p);
    int rc = check_launch("first_wgrad_rows_kernel(bn)");
    if (rc != UNETDC_OK) return rc;
    const int n = p.Cin * 9 * p.Cout;
    hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3((n + 7) / 8), dim3(256), 0, stream, p.part, dw, (int)nr, p.Cin, p.Cout);
    return check_launch("first_wgrad_reduce_kernel");
  }
  const bool mfma = first_wgrad_mfma(P, p.Cin, p.Cout);
  long nb = first_blocks(P, p.Cout);
  if (nb > 512) nb = 512;
  const long need = mfma ? first_mfma_wgrad_workspace_bytes(P, p.Cin, p.Cout) : nb * p.Cin * 9 * p.Cout * 4;
  if (need > workspace_bytes) {
    set_error("first_wgrad: workspace too small (%ld < %ld bytes)", workspace_bytes, need);
    return UNETDC_EWORKSPACE;
  }
  p.part = reinterpret_cast<float*>(workspace);
  if (mfma) {                                            // first_conv_mfma.hip
    int nblk = 0;
    int rc = launch_first_mfma_wgrad(p, &nblk, dtype, stream);
    if (rc != UNETDC_OK) return rc;
    const int n = p.Cin * 9 * p.Cout;
    hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3((n + 7) / 8), dim3(256), 0, stream, p.part, dw, nblk, p.Cin, p.Cout);
    return check_launch("first_wgrad_reduce_kernel");
  }
  const size_t lds = (size_t)4 * (p.Cout / 8) * 72 * 4;
  static int rows_off = -1;                              // UNETDC_FIRST_ROWS=0: the per-pixel kernel (A/B)
  if (rows_off < 0) { const char* e = getenv("UNETDC_FIRST_ROWS"); rows_off = (e && e[0] == '0') ? 1 : 0; }
  if (!rows_off && p.dil == 1 && p.W % 8 == 0 && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0) {
    const long nr = first_rows_blocks(P, p.Cout);
    if (nr * p.Cin * 9 * p.Cout * 4 <= workspace_bytes) {
      if (dtype == UNETDC_BF16)
        hipLaunchKernelGGL((first_wgrad_rows_kernel<bf16_t, false>), dim3((unsigned)nr, p.Cin), dim3(256), lds, stream, p);
      else
        hipLaunchKernelGGL((first_wgrad_rows_kernel<float, false>), dim3((unsigned)nr, p.Cin), dim3(256), lds, stream, p);
      int rc = check_launch("first_wgrad_rows_kernel");
      if (rc != UNETDC_OK) return rc;
      const int n = p.Cin * 9 * p.Cout;
      hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3((n + 7) / 8), dim3(256), 0, stream, p.part, dw, (int)nr,
                         p.Cin, p.Cout);
      return check_launch("first_wgrad_reduce_kernel");
    }
  }
  if (dtype == UNETDC_BF16)
    hipLaunchKernelGGL(first_conv_wgrad_kernel<bf16_t>, dim3((unsigned)nb, p.Cin), dim3(256), lds, stream, p);
  else
    hipLaunchKernelGGL(first_conv_wgrad_kernel<float>, dim3((unsigned)nb, p.Cin), dim3(256), lds, stream, p);
  int rc = check_launch("first_conv_wgrad_kernel");
  if (rc != UNETDC_OK) return rc;
  const int n = p.Cin * 9 * p.Cout;
  hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3((n + 7) / 8), dim3(256), 0, stream, p.part, dw, (int)nb,
                     p.Cin, p.Cout);
  return check_launch("first_wgrad_reduce_kernel");
}

}  // namespace unetdc
