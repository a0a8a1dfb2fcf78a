// Weight gradient of strongly dilated 3x3 convolutions (the bottleneck: d = 16 on a 32 x 32 map), bf16.
//
//   dW[t][co][ci] = sum_{n,y,x} dY[n,y,x,co] * X[n, y + (ky-1)d, x + (kx-1)d, ci]        (autograd of model_2.py:16)
//
// With d = 16 on 32 x 32 only 16 of the 36 (tap, pixel-quadrant) pairs are in bounds: the centre tap sees every pixel,
// an edge tap half of them, a corner tap a quarter.  The per-tap kernel (wgrad_dma.hip) multiplies the zero padding
// anyway (9 x 8192 pixels of K per tile pair) and splits K evenly, which also left it with 144 k workgroups on 256 CUs.
// Here every tap sums over ITS valid output rectangle only -- K = 8192 / 4096 / 2048 pixels -- and K is cut into units
// of equal length, so the centre tap gets four units, an edge tap two, a corner tap one: 16 units x (Cout/256 x Cin/256)
// tiles = 256 workgroups of identical length for the 1024 x 1024 layer, 2.25x fewer MFMAs and 16 instead of 27+ slabs.
//
// Tile 256 x 256 (co x ci), 8 waves (2 x 4, 128 x 64 each), 64 pixels per step, both operands by LDS-DMA from inline asm
// (lds_dma.h) into two stages, fragments by ds_read_b64_tr_b16 (wgrad_frag.h), one raw barrier per step.
// Slabs part[unit][co][ci] are reduced per tap in unit order (deterministic) by wgrad_rect_reduce_kernel.
#include <stdio.h>
#include <stdlib.h>

#include "kernels.h"
#include "lds_dma.h"
#include "wgrad_frag.h"

namespace unetdc {

constexpr int RECT_MAXU = 40;
constexpr unsigned ROOB = 0x80000000u;

struct WgradRectParams {
  const void* dy;                    // [P][lddy]  channels -> i (Cout)
  const void* x;                     // [P][ldx]   channels -> j (Cin)
  float* part;                       // [units][CI][CJ]
  int N, H, W, CI, CJ, lddy, ldx, d;
  int itiles, jtiles, nunits;
  // per unit (all 32-bit, indexed by the unit alone: the kernel reads them with scalar loads at base + 16*u; a table
  // indexed by the TAP, itself loaded per unit, went through `v_readfirstlane -> s_load_dword ..., soffset` and came back
  // with the wrong entries on gfx950): tap | first pixel index within the tap's rectangle list | pixel count | -,
  // rectangle origin y | x | width | height*width, and the magic dividers ceil(2^32 / v) of width and height*width
  int4 ua[RECT_MAXU];                // tap, kbeg, kcnt, 0
  int4 ub[RECT_MAXU];                // ry0, rx0, rw, rh*rw
  uint2 um[RECT_MAXU];               // magic(rw), magic(rh*rw)
};

__global__ __launch_bounds__(512, 2) void wgrad_rect_kernel(const WgradRectParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef bf16_t T;
  constexpr int TW = 4, NW = 8, NI = 4;
  constexpr int OPB = 32768;                           // bytes per operand per stage: 64 pixels x 256 channels
  constexpr int RB = Frag<T, TW>::RB;                  // 512 bytes per pixel row
  constexpr int BKP = OPB / RB;                        // 64 pixels per step
  constexpr int CPR = RB / 16, RPI = 64 / CPR;         // 32 lanes per pixel row, 2 rows per DMA instruction
  constexpr int STAGE = 2 * OPB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = lds_addr_of(smem);
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = p.itiles * p.jtiles;
  const int u = L / tiles, trem = L - u * tiles;
  const int it = trem / p.jtiles, jt = trem - it * p.jtiles;
  const int i0 = it * 256, j0 = jt * 256;
  const int4 ua = p.ua[u], ub = p.ub[u];
  const uint2 um = p.um[u];
  const int t = ua.x, kbeg = ua.y, kcnt = ua.z;
  const int ry0 = ub.x, rx0 = ub.y, rw = ub.z, rhw = ub.w;
  const unsigned mg_rw = um.x, mg_rhw = um.y;
  const int oy = (t / 3 - 1) * p.d, ox = (t % 3 - 1) * p.d;

  const unsigned abytes = (unsigned)((long)p.N * p.H * p.W * p.lddy * 2);
  const unsigned bbytes = (unsigned)((long)p.N * p.H * p.W * p.ldx * 2);
  const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, bbytes, 0x00020000);

  const int sub = lane / CPR, pc = lane % CPR;
  int rowj[4];
  unsigned coff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rowj[j] = (wave + NW * j) * RPI + sub;             // pixel row of the 64-pixel step this lane feeds
    coff[j] = (unsigned)(Frag<T, TW>::src_chunk(rowj[j], pc) * 16);
  }
  const int nsteps = (kcnt + BKP - 1) / BKP;

  f32x16 acc[NI][2];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto issue = [&](int stage, int s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = s * BKP + rowj[j];                // index within this unit
      const unsigned k = (unsigned)(kbeg + kk);        // index within the tap's rectangle list: (image, row, column)
      const unsigned n = rhw == 1 ? k : __umulhi(k, mg_rhw);
      const unsigned rem = k - n * (unsigned)rhw;
      const unsigned ry = rw == 1 ? rem : __umulhi(rem, mg_rw);
      const int y = ry0 + (int)ry, x = rx0 + (int)(rem - ry * (unsigned)rw);
      const bool ok = kk < kcnt;
      const unsigned pa = (unsigned)((((int)n * p.H + y) * p.W + x));
      const unsigned pb = (unsigned)((((int)n * p.H + y + oy) * p.W + x + ox));
      lds_dma16(ar, lds_base + stage * STAGE + (wave + NW * j) * 1024, ok ? pa * (unsigned)(p.lddy * 2) + coff[j] : ROOB,
                (unsigned)(i0 * 2));
      lds_dma16(br, lds_base + stage * STAGE + OPB + (wave + NW * j) * 1024, ok ? pb * (unsigned)(p.ldx * 2) + coff[j] : ROOB,
                (unsigned)(j0 * 2));
    }
  };

  const int wi = wave >> 2, wj = wave & 3;
  const int ca = wi * NI * 32, cb = wj * 64;
  if (nsteps > 0) issue(0, 0);
  for (int s = 0; s < nsteps; ++s) {
    wait_vmcnt<0>();
    raw_barrier();
    if (s + 1 < nsteps) issue((s + 1) & 1, s + 1);
    const unsigned char* sa = smem + (s & 1) * STAGE;
    const unsigned char* sb = sa + OPB;
#pragma unroll
    for (int k16 = 0; k16 < BKP / 16; ++k16) Frag<T, TW>::template mma16n<NI>(acc, sa, sb, lane, ca, cb, 16 * k16);
  }

  const int r = lane & 31, h = lane >> 5;
  float* slab = p.part + (long)u * p.CI * p.CJ;
#pragma unroll
  for (int mi = 0; mi < NI; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = ca + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h, j = cb + nj * 32 + r;
        slab[(long)(i0 + i) * p.CJ + j0 + j] = acc[mi][nj][reg];
      }
#endif
}

// out[(i*CJ + j)*9 + t] = sum over the units of tap t, in unit order
struct RectReduceParams {
  int ufirst[10];                     // units of tap t: ufirst[t] .. ufirst[t+1]-1
};
__global__ __launch_bounds__(256) void wgrad_rect_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                const RectReduceParams rp, long n) {
  // a lane owns four consecutive (co, ci) pairs and all nine taps: float4 reads per slab, 36 consecutive floats written
  const long q = (long)blockIdx.x * 256 + threadIdx.x;     // float4 index into one slab
  if (q * 4 >= n) return;
  float o[36];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int u = rp.ufirst[t]; u < rp.ufirst[t + 1]; ++u) {
      const float4 v = reinterpret_cast<const float4*>(part + (long)u * n)[q];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    o[t] = a.x; o[9 + t] = a.y; o[18 + t] = a.z; o[27 + t] = a.w;
  }
  float4* dst = reinterpret_cast<float4*>(out + q * 36);
#pragma unroll
  for (int k = 0; k < 9; ++k) dst[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
}

// ------------------------------------------------------------------------------------------------------------------------
static int rect_enabled() {
  static int v = -1;                                    // UNETDC_WGRAD_RECT=0: per-tap kernel over the padded K (A/B)
  if (v < 0) { const char* e = getenv("UNETDC_WGRAD_RECT"); v = (e && e[0] == '0') ? 0 : 1; }
  return v;
}

// plan: K unit = the smallest tap rectangle list (rounded up to 64 pixels); returns the number of units or 0
// tiles = number of 256 x 256 channel tiles: with few tiles the K unit is halved so that units x tiles still gives every
// CU a workgroup (512 -> 1024 channels: 8 tiles x 16 units = 128 workgroups ran on half of the chip)
static int rect_plan(int N, int H, int W, int d, int tiles, WgradRectParams* out, int* tap_of_unit = nullptr) {
  int cnt[9], ry0[9], rx0[9], rh[9], rw[9], minc = 1 << 30;
  long total = 0;
  for (int t = 0; t < 9; ++t) {
    const int oy = (t / 3 - 1) * d, ox = (t % 3 - 1) * d;
    const int y0 = oy < 0 ? -oy : 0, y1 = oy > 0 ? H - oy : H, x0 = ox < 0 ? -ox : 0, x1 = ox > 0 ? W - ox : W;
    ry0[t] = y0; rx0[t] = x0; rh[t] = y1 - y0; rw[t] = x1 - x0;
    if (rh[t] <= 0 || rw[t] <= 0) return 0;
    cnt[t] = N * rh[t] * rw[t];
    total += cnt[t];
    if (cnt[t] < minc) minc = cnt[t];
  }
  if (total * 10 > 7L * 9 * N * H * W) return 0;        // >= 70 % of the padded pixels are live: nothing to gain
  auto magic = [](unsigned v) { return v <= 1 ? 0u : (unsigned)(((1ull << 32) + v - 1) / v); };
  const int split = (16L * tiles < 256 && minc >= 256) ? 2 : 1;
  const int unit = ((minc / split + 63) / 64) * 64;
  int nu = 0;
  for (int t = 0; t < 9; ++t)
    for (int b = 0; b < cnt[t]; b += unit) {
      if (nu >= RECT_MAXU) return 0;
      if (out) {
        out->ua[nu] = make_int4(t, b, cnt[t] - b < unit ? cnt[t] - b : unit, 0);
        out->ub[nu] = make_int4(ry0[t], rx0[t], rw[t], rh[t] * rw[t]);
        out->um[nu] = make_uint2(magic((unsigned)rw[t]), magic((unsigned)(rh[t] * rw[t])));
      }
      if (tap_of_unit) tap_of_unit[nu] = t;
      ++nu;
    }
  if (out) out->nunits = nu;
  return nu;
}

bool wgrad_rect_supported(int N, int H, int W, int CI, int CJ, int lda, int ldb, int d, int ntaps, int stride, int dtype) {
  if (!rect_enabled() || dtype != UNETDC_BF16 || ntaps != 9 || stride != 1) return false;
  if (CI % 256 != 0 || CJ % 256 != 0 || d < 1) return false;
  const long P = (long)N * H * W;
  if (P * lda * 2 >= (1L << 31) || P * ldb * 2 >= (1L << 31)) return false;
  return rect_plan(N, H, W, d, (CI / 256) * (CJ / 256), nullptr) > 0;
}

long wgrad_rect_workspace_bytes(int N, int H, int W, int CI, int CJ, int d) {
  if (CI % 256 != 0 || CJ % 256 != 0) return 0;
  const int nu = rect_plan(N, H, W, d, (CI / 256) * (CJ / 256), nullptr);
  return (long)nu * CI * CJ * 4;
}

int launch_wgrad_rect(const void* dy, int lddy, const void* x, int ldx, float* out, void* workspace, long workspace_bytes,
                      int N, int H, int W, int CI, int CJ, int d, hipStream_t stream) {
  WgradRectParams p{};
  p.dy = dy; p.x = x; p.part = reinterpret_cast<float*>(workspace);
  p.N = N; p.H = H; p.W = W; p.CI = CI; p.CJ = CJ; p.lddy = lddy; p.ldx = ldx; p.d = d;
  p.itiles = CI / 256; p.jtiles = CJ / 256;
  int utap[RECT_MAXU];
  const int nu = rect_plan(N, H, W, d, p.itiles * p.jtiles, &p, utap);
  UNETDC_REQUIRE(nu > 0, "wgrad_rect: unsupported geometry");
  const long need = (long)nu * CI * CJ * 4;
  if (need > workspace_bytes) {
    set_error("wgrad_rect: workspace too small (%ld < %ld bytes)", workspace_bytes, need);
    return UNETDC_EWORKSPACE;
  }
  if (const int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&wgrad_rect_kernel), 131072, "wgrad_rect_kernel")) return rc_;
  const long nwg = (long)nu * p.itiles * p.jtiles;
  hipLaunchKernelGGL(wgrad_rect_kernel, dim3((unsigned)nwg), dim3(512), 131072, stream, p);
  note_kernel("wgrad_rect_kernel");
  int rc = check_launch("wgrad_rect_kernel");
  if (rc != UNETDC_OK) return rc;
  RectReduceParams rp{};
  int tcur = 0;
  rp.ufirst[0] = 0;
  for (int u = 0; u < nu; ++u)
    while (tcur < utap[u]) rp.ufirst[++tcur] = u;
  while (tcur < 9) rp.ufirst[++tcur] = nu;
  const long n = (long)CI * CJ;
  hipLaunchKernelGGL(wgrad_rect_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, p.part, out, rp, n);
  return check_launch("wgrad_rect_reduce_kernel");
}

}  // namespace unetdc
