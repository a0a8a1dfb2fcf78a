// Weight-gradient GEMM on the matrix cores: the reduction runs over PIXELS.
//
//   part[ks][t][i][j] = sum_{p in slice ks} A[p][i] * B[shift_t(p)][j]
//
//   conv3x3 wgrad  (autograd of nn.Conv2d, models/model_2.py:41-51):
//       A = dY (i = cout), B = X shifted by (ky-1)*d,(kx-1)*d (j = cin)      -> dW[co][ci][ky][kx]
//   ConvTranspose2d(2,2,s=2) wgrad (model_2.py:20-29):
//       A = X  (i = cin),  B = dUp at (2y+a, 2x+b)            (j = cout)     -> dW[ci][co][a][b]
//
// Both operands are pixel-major (NHWC), i.e. K-major for this GEMM ("TN").  The MFMA wants each
// lane to hold consecutive K for one row, so the LDS image stays [pixel][channel] exactly as it
// arrives from HBM (coalesced 16-byte chunks) and
//   * bf16: fragments are read with ds_read_b64_tr_b16 (hardware transpose, 4 pixels x 16
//           channels per 16-lane group); the 64-byte units of a row are XOR-swizzled with the
//           pixel index so the 4 pixel rows of one read hit 4 distinct bank windows;
//   * fp32: v_mfma_f32_32x32x2_f32 takes one scalar per lane, read with ds_read_b32 from
//           [pixel][channel] (32 consecutive channels per half-wave: conflict-free).
// Tiles: 64x64 outputs per wave (2x2 MFMA 32x32).  Narrow layers (64 channels) use a 64x64 block
// tile with the 4 waves splitting the pixel chunk (summed through LDS at the end); wide layers
// use a 128x128 block tile (2x2 waves).  Across blocks the pixel range is split `ksplit` ways
// into fp32 partial slabs that a second, deterministic kernel sums and permutes into PyTorch's
// parameter layout -- no atomics, bitwise reproducible.
#include <stdlib.h>

#include <stdio.h>

#include "kernels.h"
#include "wgrad_frag.h"

namespace unetdc {

// TW = 1: 64x64 block tile, the 4 waves split each pixel chunk; TW = 2: 128x128, waves 2x2.
template <typename T, int TW>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
  constexpr int RB = Frag<T, TW>::RB;                 // LDS bytes per pixel row (per operand)
  constexpr int OPB = 16384;                          // LDS bytes per operand per stage
  constexpr int BKP = OPB / RB;                       // pixels per block step
  constexpr int CPR = RB / 16;                        // 16-byte chunks per pixel row
  constexpr int RPP = 256 / CPR;                      // pixel rows staged per pass of the block
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int STAGE = 2 * OPB;
  constexpr int WK = (TW == 1) ? BKP / 4 : BKP;       // pixels per wave per step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per_slice = p.ntaps * p.itiles * p.jtiles;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int ks = L / per_slice;
  int rest = L - ks * per_slice;
  const int t = rest / (p.itiles * p.jtiles);
  rest -= t * p.itiles * p.jtiles;
  const int it = rest / p.jtiles, jt = rest - it * p.jtiles;
  const int i0 = it * TW * 64, j0 = jt * TW * 64;
  const int oy = p.offy[t], ox = p.offx[t];

  const T* __restrict__ ag = reinterpret_cast<const T*>(p.a);
  const T* __restrict__ bg = reinterpret_cast<const T*>(p.b);

  const long pbeg = (long)ks * p.chunk;
  const long pend = (pbeg + p.chunk < (long)p.P) ? pbeg + p.chunk : (long)p.P;
  const int nsteps = (int)((pend - pbeg + BKP - 1) / BKP);

  // staging assignment: chunk c of pixel rows rr + RPP*i (i = 0..3)
  const int c = tid % CPR, rr = tid / CPR;
  int wr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wr[i] = Frag<T, TW>::wr_off(rr + RPP * i, c);
  // coordinates of the pixel this thread stages in row rr + RPP*i (advanced by BKP per step)
  int cn[4], cy[4], cx[4];
  const int HW = p.H * p.W;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long pp = pbeg + rr + RPP * i;
    const int n = (int)(pp / HW), rem = (int)(pp - (long)n * HW);
    cn[i] = n;
    cy[i] = rem / p.W;
    cx[i] = rem - cy[i] * p.W;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  u32x4 ra[4], rb[4];
  long pcur = pbeg;      // first pixel of the step being loaded
  auto gload = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long pp = pcur + rr + RPP * i;
      const bool inr = pp < pend;
      u32x4 va = {0u, 0u, 0u, 0u}, vb = {0u, 0u, 0u, 0u};
      if (inr) {
        va = ld16(ag + (pp * p.lda + i0 + c * EPC));
        const int iy = cy[i] * p.stride + oy, ix = cx[i] * p.stride + ox;
        if ((unsigned)iy < (unsigned)p.Hb && (unsigned)ix < (unsigned)p.Wb)
          vb = ld16(bg + (((long)(cn[i] * p.Hb + iy) * p.Wb + ix) * p.ldb + j0 + c * EPC));
      }
      ra[i] = va;
      rb[i] = vb;
      // advance this row's pixel by BKP
      cx[i] += p.adv_x;
      if (cx[i] >= p.W) { cx[i] -= p.W; ++cy[i]; }
      cy[i] += p.adv_y;
      while (cy[i] >= p.H) { cy[i] -= p.H; ++cn[i]; }
    }
    pcur += BKP;
  };
  auto lds_store = [&](int stage) {
    unsigned char* base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      st16(base + wr[i], ra[i]);
      st16(base + OPB + wr[i], rb[i]);
    }
  };

  const int wi = (TW == 1) ? 0 : (wave >> 1), wj = (TW == 1) ? 0 : (wave & 1);
  const int kbase = (TW == 1) ? wave * WK : 0;

  if (nsteps > 0) {
    gload();
    lds_store(0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const bool more = (s + 1 < nsteps);
    if (more) gload();
    const unsigned char* sa = smem + (s & 1) * STAGE;
    const unsigned char* sb = sa + OPB;
#pragma unroll
    for (int k16 = 0; k16 < WK / 16; ++k16)
      Frag<T, TW>::mma16(acc, sa, sb, lane, wi * 64, wj * 64, kbase + 16 * k16);
    if (more) lds_store((s + 1) & 1);
    __syncthreads();
  }

  // ---- write the partial slab --------------------------------------------------------------------
  // acc[mi][nj][reg]: i = mi*32 + (reg&3) + 8*(reg>>2) + 4*h, j = nj*32 + (lane&31)
  const int r = lane & 31, h = lane >> 5;
  float* slab = p.part + ((long)ks * p.ntaps + t) * p.CI * p.CJ;
  if (TW == 1) {
    float* red = reinterpret_cast<float*>(smem);       // [wave][64][64] fp32 = 64 KB
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int i = mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h, j = nj * 32 + r;
          red[(wave * 64 + i) * 64 + j] = acc[mi][nj][reg];
        }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int idx = e * 256 + tid, i = idx >> 6, j = idx & 63;
      const float v = red[idx] + red[4096 + idx] + red[8192 + idx] + red[12288 + idx];
      slab[(long)(i0 + i) * p.CJ + j0 + j] = v;
    }
  } else {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int i = wi * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h, j = wj * 64 + nj * 32 + r;
          slab[(long)(i0 + i) * p.CJ + j0 + j] = acc[mi][nj][reg];
        }
  }
}

// out[(i*CJ + j)*ntaps + t] = sum_ks part[ks][t][i][j]   (PyTorch [Cout][Cin][3][3] / [Cin][Cout][2][2])
// A block owns 64 float4 columns (256 consecutive (t, ij) entries; CI*CJ is a multiple of 4096, so a float4
// never straddles a tap) and splits the slabs over its 4 waves: wave g sums slabs g, g+4, g+8, ... with two
// independent accumulators, then the four partial sums are added in wave order through LDS.  16-byte
// coalesced reads, 4x the loads in flight of a thread-per-output loop (which ran the 75 MB of slabs of a
// tap-fused layer at 1.6 TB/s); the summation order is fixed, so the result is bitwise reproducible.
// (A variant with all taps of four (co, ci) pairs per lane -- contiguous 144-byte output runs -- has 9x fewer blocks:
// 16 blocks for a 64 x 64 layer walking 512 slabs each; it measured 82 us per launch against 18 us.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                           int ksplit, int ntaps, int CI, int CJ) {
  __shared__ float4 red[4][64];
  const long n = (long)CI * CJ;
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long q = (long)blockIdx.x * 64 + lane;             // float4 index into one slab [ntaps][n]
  const long total4 = n * ntaps / 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a, d = a;
  if (q < total4) {
    const float4* src = reinterpret_cast<const float4*>(part) + q;
    const long stride4 = total4;
    int ks = g;
    for (; ks + 12 < ksplit; ks += 16) {                   // four slabs in flight per lane (fixed order: reproducible)
      const float4 u = src[(long)ks * stride4], v = src[(long)(ks + 4) * stride4];
      const float4 w = src[(long)(ks + 8) * stride4], z = src[(long)(ks + 12) * stride4];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
      c.x += w.x; c.y += w.y; c.z += w.z; c.w += w.w;
      d.x += z.x; d.y += z.y; d.z += z.z; d.w += z.w;
    }
    for (; ks < ksplit; ks += 4) {
      const float4 u = src[(long)ks * stride4];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    }
  }
  red[g][lane] = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
  __syncthreads();
  if (g == 0 && q < total4) {
    float4 s = red[0][lane];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      s.x += red[k][lane].x; s.y += red[k][lane].y; s.z += red[k][lane].z; s.w += red[k][lane].w;
    }
    const long e = q * 4;
    const int t = (int)(e / n);
    const long ij = e - (long)t * n;
    out[ij * ntaps + t] = s.x;
    out[(ij + 1) * ntaps + t] = s.y;
    out[(ij + 2) * ntaps + t] = s.z;
    out[(ij + 3) * ntaps + t] = s.w;
  }
}

// ------------------------------------------------------------------------------------------------
// second-generation (LDS-DMA) kernel, wgrad_dma.hip
int wgrad_dma_tile(int CI, int CJ);
int wgrad_dma_pixel_step(int dtype, int tw);
bool wgrad_dma_supported(const WgradParams& p, int dtype);
int launch_wgrad_dma_kernel(WgradParams& p, int tw, int dtype, hipStream_t stream);
// tap-fused kernel for narrow 3x3 layers, wgrad_fused.hip
bool wgrad_fused_supported(int N, int H, int W, int CI, int CJ, int lda, int ldb, int d, int ntaps, int stride,
                           int dtype);
long wgrad_fused_workspace_bytes(int N, int H, int W, int CI, int CJ, int dtype);
int launch_wgrad_fused(const void* dy, int lddy, const void* x, int ldx, float* part, int N, int H, int W, int CI,
                       int CJ, int d, int dtype, int* units_out, hipStream_t stream, const float* in_scale = nullptr,
                       const float* in_shift = nullptr);

// valid-rectangle kernel for strongly dilated layers, wgrad_rect.hip
bool wgrad_rect_supported(int N, int H, int W, int CI, int CJ, int lda, int ldb, int d, int ntaps, int stride, int dtype);
int launch_wgrad_rect(const void* dy, int lddy, const void* x, int ldx, float* out, void* workspace, long workspace_bytes,
                      int N, int H, int W, int CI, int CJ, int d, hipStream_t stream);

// UNETDC_WGRAD=legacy: first-generation register-staged kernel; =dma: per-tap LDS-DMA kernels only
// (no tap-fused kernel).  Default: best kernel per layer.
static int wgrad_choice() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("UNETDC_WGRAD");
    v = (e && e[0] == 'l') ? 1 : ((e && e[0] == 'd') ? 2 : 0);
  }
  return v;
}
static bool wgrad_legacy() { return wgrad_choice() == 1; }

static int pixel_step(int dtype, bool wide) {
  const int rb = (wide ? 2 : 1) * 64 * (dtype == UNETDC_BF16 ? 2 : 4);
  return 16384 / rb;
}

static bool wgrad_wide(int CI, int CJ) { return CI % 128 == 0 && CJ % 128 == 0; }

// Choose the K split.  The grid is tiles*ksplit workgroups on 256 CUs with `bpc` resident blocks per
// CU.  Cost model (microseconds): MFMA time at ~1 PFLOP/s divided by the fill efficiency of the
// last wave of workgroups, plus writing and re-reading ksplit fp32 slabs at ~4 TB/s.
static void plan(long P, int tiles, int step, int bpc, double slab_bytes, double flops, int& ksplit, int& chunk) {
  const long maxsplit = (P + (long)step * 4 - 1) / ((long)step * 4);
  const long slots = 256L * bpc;
  const double t_mfma = flops / 1.0e15 * 1e6;
  int best = 1;
  double best_cost = 1e30;
  for (int ks = 1; ks <= 160 && ks <= maxsplit; ++ks) {
    const long blocks = (long)tiles * ks;
    const long rounds = (blocks + slots - 1) / slots;
    const double eff = (double)blocks / (double)(rounds * slots);
    const double cost = t_mfma / eff + 2.0 * ks * slab_bytes / 4.0e12 * 1e6 + 0.02 * ks;
    if (cost < best_cost) { best_cost = cost; best = ks; }
  }
  long ch = (P + best - 1) / best;
  ch = ((ch + step - 1) / step) * step;
  chunk = (int)ch;
  ksplit = (int)((P + ch - 1) / ch);
}

static void plan_legacy(long P, int CI, int CJ, int ntaps, int dtype, int& ksplit, int& chunk) {
  const bool wide = wgrad_wide(CI, CJ);
  const int tiles = ntaps * (CI / (wide ? 128 : 64)) * (CJ / (wide ? 128 : 64));
  plan(P, tiles, pixel_step(dtype, wide), 2, (double)ntaps * CI * CJ * 4, 2.0 * P * CI * CJ * ntaps, ksplit, chunk);
}

static void plan_dma(long P, int CI, int CJ, int ntaps, int dtype, int& ksplit, int& chunk) {
  const int tw = wgrad_dma_tile(CI, CJ);
  const int tiles = ntaps * (CI / (tw * 64)) * (CJ / (tw * 64));
  plan(P, tiles, wgrad_dma_pixel_step(dtype, tw), tw == 4 ? 1 : 2, (double)ntaps * CI * CJ * 4,
       2.0 * P * CI * CJ * ntaps, ksplit, chunk);
}

long wgrad_workspace_bytes(long P, int CI, int CJ, int ntaps, int dtype) {
  int k1, c1, k2, c2;
  plan_legacy(P, CI, CJ, ntaps, dtype, k1, c1);
  plan_dma(P, CI, CJ, ntaps, dtype, k2, c2);
  return (long)(k1 > k2 ? k1 : k2) * ntaps * CI * CJ * 4;
}

template <typename T, int TW>
static int launch_w(WgradParams& p, hipStream_t stream) {
  if (const int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&wgrad_kernel<T, TW>), 65536, "wgrad_kernel")) return rc_;
  const long nwg = (long)p.ksplit * p.ntaps * p.itiles * p.jtiles;
  hipLaunchKernelGGL((wgrad_kernel<T, TW>), dim3((unsigned)nwg), dim3(256), 65536, stream, p);
  char nm[96];
  snprintf(nm, sizeof(nm), "wgrad_kernel<%s, %d>", sizeof(T) == 2 ? "__bf16" : "float", TW);
  note_kernel(nm);
  return check_launch("wgrad_kernel");
}

// Generic driver: fills partial slabs in `workspace` and reduces them into `out`.
int launch_wgrad(WgradParams& p, float* out, void* workspace, long workspace_bytes, int dtype, hipStream_t stream) {
  const int esz = dtype == UNETDC_BF16 ? 2 : 4;
  UNETDC_REQUIRE(dtype == UNETDC_F32 || dtype == UNETDC_BF16, "wgrad: bad dtype %d", dtype);
  UNETDC_REQUIRE(p.a && p.b && out && workspace, "wgrad: null pointer");
  UNETDC_REQUIRE(p.CI % 64 == 0 && p.CJ % 64 == 0, "wgrad: channel counts (%d,%d) must be multiples of 64", p.CI, p.CJ);
  UNETDC_REQUIRE(p.lda % (16 / esz) == 0 && p.ldb % (16 / esz) == 0, "wgrad: ld not 16-byte aligned");
  UNETDC_REQUIRE(((uintptr_t)p.a % 16 == 0) && ((uintptr_t)p.b % 16 == 0), "wgrad: pointers must be 16-byte aligned");
  const long P = (long)p.N * p.H * p.W;
  UNETDC_REQUIRE(P > 0 && P < (1L << 31) - 4096, "wgrad: pixel count out of range");
  p.P = (int)P;
  if (p.in_scale) {                               // input normalisation on load: the tap-split ring kernel only (caller asked wgrad_bnin_supported)
    UNETDC_REQUIRE(p.in_shift && p.Hb == p.H && p.Wb == p.W && p.ntaps == 9 && p.stride == 1 &&
                       wgrad_bnin_supported(p.N, p.H, p.W, p.CI, p.CJ, p.lda, p.ldb, p.offy[8], dtype),
                   "wgrad (bnin): shape not supported by the input-normalising kernel");
  }
  if (!p.in_scale && wgrad_choice() == 0 && p.ntaps == 4 && p.stride == 2 && p.Hb == 2 * p.H && p.Wb == 2 * p.W &&
      p.offy[3] == 1 && p.offx[3] == 1 && convt_wgrad_fused_supported(p.N, p.H, p.W, p.CI, p.CJ, p.lda, p.ldb, dtype)) {
    // ConvTranspose2d(2, 2): one GEMM [Cin] x [4 Cout] with the input staged once for the four taps (convt_wgrad.hip)
    const long need_c = convt_wgrad_fused_workspace_bytes(p.N, p.H, p.W, p.CI, p.CJ);
    if (need_c > workspace_bytes) {
      set_error("wgrad: workspace too small (%ld < %ld bytes)", workspace_bytes, need_c);
      return UNETDC_EWORKSPACE;
    }
    int units = 0;
    int rc = launch_convt_wgrad_fused(p.a, p.lda, p.b, p.ldb, reinterpret_cast<float*>(workspace), p.N, p.H, p.W, p.CI, p.CJ,
                                      &units, stream);
    if (rc != UNETDC_OK) return rc;
    const long n = (long)p.CI * p.CJ * 4;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, stream,
                       reinterpret_cast<float*>(workspace), out, units, 4, p.CI, p.CJ);
    return check_launch("wgrad_reduce_kernel");
  }
  if (!p.in_scale && wgrad_choice() == 0 && p.Hb == p.H && p.Wb == p.W &&
      wgrad_rect_supported(p.N, p.H, p.W, p.CI, p.CJ, p.lda, p.ldb, p.offy[8], p.ntaps, p.stride, dtype))
    return launch_wgrad_rect(p.a, p.lda, p.b, p.ldb, out, workspace, workspace_bytes, p.N, p.H, p.W, p.CI, p.CJ, p.offy[8],
                             stream);
  if ((p.in_scale || wgrad_choice() == 0) && p.Hb == p.H && p.Wb == p.W &&
      wgrad_fused_supported(p.N, p.H, p.W, p.CI, p.CJ, p.lda, p.ldb, p.offy[8], p.ntaps, p.stride, dtype)) {
    const long need_f = wgrad_fused_workspace_bytes(p.N, p.H, p.W, p.CI, p.CJ, dtype);
    if (need_f > workspace_bytes) {
      set_error("wgrad: workspace too small (%ld < %ld bytes)", workspace_bytes, need_f);
      return UNETDC_EWORKSPACE;
    }
    int units = 0;
    int rc = launch_wgrad_fused(p.a, p.lda, p.b, p.ldb, reinterpret_cast<float*>(workspace), p.N, p.H, p.W, p.CI,
                                p.CJ, p.offy[8], dtype, &units, stream, p.in_scale, p.in_shift);
    if (rc != UNETDC_OK) return rc;
    const long n = (long)p.CI * p.CJ * p.ntaps;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, stream,
                       reinterpret_cast<float*>(workspace), out, units, p.ntaps, p.CI, p.CJ);
    return check_launch("wgrad_reduce_kernel");
  }
  const bool dma = !wgrad_legacy() && wgrad_dma_supported(p, dtype);
  const bool wide = wgrad_wide(p.CI, p.CJ);
  const int tw = dma ? wgrad_dma_tile(p.CI, p.CJ) : (wide ? 2 : 1);
  if (dma) plan_dma(P, p.CI, p.CJ, p.ntaps, dtype, p.ksplit, p.chunk);
  else plan_legacy(P, p.CI, p.CJ, p.ntaps, dtype, p.ksplit, p.chunk);
  const long need = (long)p.ksplit * p.ntaps * p.CI * p.CJ * 4;
  if (need > workspace_bytes) {
    set_error("wgrad: workspace too small (%ld < %ld bytes)", workspace_bytes, need);
    return UNETDC_EWORKSPACE;
  }
  p.part = reinterpret_cast<float*>(workspace);
  p.itiles = p.CI / (tw * 64);
  p.jtiles = p.CJ / (tw * 64);
  const int step = dma ? wgrad_dma_pixel_step(dtype, tw) : pixel_step(dtype, wide);
  p.adv_y = step / p.W;
  p.adv_x = step % p.W;
  int rc;
  if (dma)
    rc = launch_wgrad_dma_kernel(p, tw, dtype, stream);
  else if (dtype == UNETDC_BF16)
    rc = wide ? launch_w<bf16_t, 2>(p, stream) : launch_w<bf16_t, 1>(p, stream);
  else
    rc = wide ? launch_w<float, 2>(p, stream) : launch_w<float, 1>(p, stream);
  if (rc != UNETDC_OK) return rc;
  const long n = (long)p.CI * p.CJ * p.ntaps;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, stream, p.part, out,
                     p.ksplit, p.ntaps, p.CI, p.CJ);
  return check_launch("wgrad_reduce_kernel");
}

}  // namespace unetdc
