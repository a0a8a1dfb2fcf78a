// Second-generation weight-gradient GEMM: same math, slabs and deterministic reduce as wgrad.hip,
// but both pixel-major operands travel HBM/L2 -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`),
// with the bank swizzle applied on the source side and out-of-image / out-of-range pixels zero-filled
// by the buffer descriptor's range check (tools/probes/dma_oob.hip).  Adds a 256x256 block tile
// (8 waves, 128x64 per wave) for the wide layers.
//
//   part[ks][t][i][j] = sum_{p in slice ks} A[p][i] * B[shift_t(p)][j]      (see wgrad.hip)
#include <stdio.h>

#include "kernels.h"
#include "lds_dma.h"
#include "wgrad_frag.h"

namespace unetdc {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned WOOB = 0x80000000u;

// TW = 1: 64x64 block tile, 4 waves split the pixel chunk;  TW = 2: 128x128, waves 2x2 (64x64 each);
// TW = 4: 256x256, 8 waves as 2(i) x 4(j), 128x64 each.
template <typename T, int TW>
__global__ __launch_bounds__(512, 2) void wgrad_dma_kernel(const WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = (TW == 4) ? 8 : 4;
  constexpr int NI = (TW == 4) ? 4 : 2;               // 32-row MFMA tiles along i per wave
  constexpr int OPB = (TW == 4) ? 32768 : 16384;      // LDS bytes per operand per stage
  constexpr int RB = Frag<T, TW>::RB;                 // bytes per pixel row
  constexpr int BKP = OPB / RB;                       // pixels per block step
  constexpr int CPR = RB / 16;                        // lanes per pixel row
  constexpr int RPI = 64 / CPR;                       // pixel rows per DMA wave-instruction
  constexpr int ES = (int)sizeof(T);
  constexpr int STAGE = 2 * OPB;
  constexpr int WK = (TW == 1) ? BKP / 4 : BKP;       // pixels per wave per step
  static_assert(OPB / 1024 / NW == 4, "4 DMA instructions per wave per operand per step");
  static_assert(WK % 16 == 0, "wave K slice must be a multiple of the MFMA K");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = lds_addr_of(smem);
  const int per_slice = p.ntaps * p.itiles * p.jtiles;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int ks = L / per_slice;
  int rest = L - ks * per_slice;
  const int t = rest / (p.itiles * p.jtiles);
  rest -= t * p.itiles * p.jtiles;
  const int it = rest / p.jtiles, jt = rest - it * p.jtiles;
  const int i0 = it * TW * 64, j0 = jt * TW * 64;
  const int oy = p.offy[t], ox = p.offx[t];

  const unsigned abytes = (unsigned)((long)p.P * p.lda * ES);
  const unsigned bbytes = (unsigned)((long)p.N * p.Hb * p.Wb * p.ldb * ES);
  const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.b), 0, bbytes, 0x00020000);

  const long pbeg = (long)ks * p.chunk;
  const long pend = (pbeg + p.chunk < (long)p.P) ? pbeg + p.chunk : (long)p.P;
  const int nsteps = (int)((pend - pbeg + BKP - 1) / BKP);

  // the 4 pixel rows (one per DMA instruction) this lane feeds, per operand
  const int sub = lane / CPR, pc = lane % CPR;
  int rowj[4], cn[4], cy[4], cx[4];
  unsigned coff[4];                                   // byte offset of the (swizzled) source chunk in a row
  const int HW = p.H * p.W;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rowj[j] = (wave + NW * j) * RPI + sub;
    coff[j] = (unsigned)(Frag<T, TW>::src_chunk(rowj[j], pc) * 16);
    const long pp = pbeg + rowj[j];
    const int n = (int)(pp / HW), rem = (int)(pp - (long)n * HW);
    cn[j] = n;
    cy[j] = rem / p.W;
    cx[j] = rem - cy[j] * p.W;
  }

  f32x16 acc[NI][2];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  long pcur = pbeg;
  auto issue = [&](int stage) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long pp = pcur + rowj[j];
      const bool inr = pp < pend;
      const unsigned va = inr ? (unsigned)((pp * p.lda + i0) * ES) + coff[j] : WOOB;
      const int iy = cy[j] * p.stride + oy, ix = cx[j] * p.stride + ox;
      const bool okb = inr && (unsigned)iy < (unsigned)p.Hb && (unsigned)ix < (unsigned)p.Wb;
      const unsigned vb = okb ? (unsigned)((((long)(cn[j] * p.Hb + iy) * p.Wb + ix) * p.ldb + j0) * ES) + coff[j] : WOOB;
      // DMA from inline asm (lds_dma.h): with the builtin, hipcc put an `s_waitcnt vmcnt(0)` in front of the first
      // transposed fragment read of EVERY step, i.e. the tile just requested for step s+1 had to land before step s
      // could compute -- no overlap at all (2.5 us per 64-pixel step measured in round 1 = DMA time + MFMA time)
      lds_dma16(ar, lds_base + stage * STAGE + (wave + NW * j) * 1024, va, 0u);
      lds_dma16(br, lds_base + stage * STAGE + OPB + (wave + NW * j) * 1024, vb, 0u);
      cx[j] += p.adv_x;
      if (cx[j] >= p.W) { cx[j] -= p.W; ++cy[j]; }
      cy[j] += p.adv_y;
      while (cy[j] >= p.H) { cy[j] -= p.H; ++cn[j]; }
    }
    pcur += BKP;
  };

  const int wi = (TW == 1) ? 0 : ((TW == 2) ? (wave >> 1) : (wave >> 2));
  const int wj = (TW == 1) ? 0 : ((TW == 2) ? (wave & 1) : (wave & 3));
  const int ca = wi * NI * 32, cb = wj * 64;
  const int kbase = (TW == 1) ? wave * WK : 0;

  if (nsteps > 0) issue(0);
  for (int s = 0; s < nsteps; ++s) {
    wait_vmcnt<0>();                                   // my pieces of step s have landed ...
    raw_barrier();                                     // ... everyone's; and everyone has issued the MFMAs of step s-1
    if (s + 1 < nsteps) issue((s + 1) & 1);
    const unsigned char* sa = smem + (s & 1) * STAGE;
    const unsigned char* sb = sa + OPB;
#pragma unroll
    for (int k16 = 0; k16 < WK / 16; ++k16)
      Frag<T, TW>::template mma16n<NI>(acc, sa, sb, lane, ca, cb, kbase + 16 * k16);
  }

  // ---- write the partial slab --------------------------------------------------------------------
  const int r = lane & 31, h = lane >> 5;
  float* slab = p.part + ((long)ks * p.ntaps + t) * p.CI * p.CJ;
  if (TW == 1) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);       // [wave][64][64] fp32 = 64 KB
#pragma unroll
    for (int mi = 0; mi < NI; ++mi)
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
    for (int mi = 0; mi < NI; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int i = ca + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h, j = cb + nj * 32 + r;
          slab[(long)(i0 + i) * p.CJ + j0 + j] = acc[mi][nj][reg];
        }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// ------------------------------------------------------------------------------------------------
int wgrad_dma_tile(int CI, int CJ) {          // tile width in units of 64 channels
  if (CI % 256 == 0 && CJ % 256 == 0) return 4;
  if (CI % 128 == 0 && CJ % 128 == 0) return 2;
  return 1;
}

int wgrad_dma_pixel_step(int dtype, int tw) {
  const int rb = tw * 64 * (dtype == UNETDC_BF16 ? 2 : 4);
  return (tw == 4 ? 32768 : 16384) / rb;
}

bool wgrad_dma_supported(const WgradParams& p, int dtype) {
  const long es = dtype == UNETDC_BF16 ? 2 : 4;
  const long P = (long)p.N * p.H * p.W;
  return P * p.lda * es < (1L << 31) && (long)p.N * p.Hb * p.Wb * p.ldb * es < (1L << 31);
}

template <typename T, int TW>
static int launch_wd(WgradParams& p, hipStream_t stream) {
  constexpr int LDS = (TW == 4) ? 131072 : 65536;
  constexpr int NT = (TW == 4) ? 512 : 256;
  if (const int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&wgrad_dma_kernel<T, TW>), LDS, "wgrad_dma_kernel")) return rc_;
  const long nwg = (long)p.ksplit * p.ntaps * p.itiles * p.jtiles;
  hipLaunchKernelGGL((wgrad_dma_kernel<T, TW>), dim3((unsigned)nwg), dim3(NT), LDS, stream, p);
  char nm[96];
  snprintf(nm, sizeof(nm), "wgrad_dma_kernel<%s, %d>", sizeof(T) == 2 ? "__bf16" : "float", TW);
  note_kernel(nm);
  return check_launch("wgrad_dma_kernel");
}

int launch_wgrad_dma_kernel(WgradParams& p, int tw, int dtype, hipStream_t stream) {
  if (dtype == UNETDC_BF16) {
    if (tw == 4) return launch_wd<bf16_t, 4>(p, stream);
    if (tw == 2) return launch_wd<bf16_t, 2>(p, stream);
    return launch_wd<bf16_t, 1>(p, stream);
  }
  if (tw == 4) return launch_wd<float, 4>(p, stream);
  if (tw == 2) return launch_wd<float, 2>(p, stream);
  return launch_wd<float, 1>(p, stream);
}

}  // namespace unetdc
