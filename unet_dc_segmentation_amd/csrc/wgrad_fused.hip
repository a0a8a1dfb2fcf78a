// Tap-fused weight gradient of the dilated 3x3 convolutions (64x64 channel tiles, d <= 8).
//
//   dW[t][co][ci] = sum_p dY[p][co] * X[p + off_t][ci]                (autograd of nn.Conv2d,
//                                                                     models/model_2.py:41-51)
// The per-tap kernel (wgrad_dma.hip) re-stages dY and X for each of the 9 taps and, on a 64x64 tile,
// gets only 8 MFMAs per wave between barriers: it ran at ~370 TFLOP/s on enc1/dec1.  Here one
// workgroup walks down a vertical strip of the image (SEG pixels wide).  Per image row it stages
//   * the dY row segment            [SEG pixels][64 co]
//   * three X row segments y-d,y,y+d [SEG + 2d (padded to SEG+16) pixels][64 ci]  (zero outside)
// by LDS-DMA and accumulates ALL NINE taps from them: tap (ky,kx) reads X segment ky at pixel rows
// shifted by kx*d.  9x the MFMA work per staged byte and per barrier.
// Waves: 2x2 quadrants of the 64x64 (co x ci) tile, 9 accumulators (one per tap) each.
// Fragments as in wgrad_frag.h: bf16 via ds_read_b64_tr_b16 on the [pixel][channel] image (64-byte
// units XOR-swizzled by pixel row), fp32 via scalar reads.  Strips x y-ranges give the K split;
// partial slabs are reduced by wgrad_reduce_kernel (deterministic).
#include <stdio.h>
#include <stdlib.h>

#include "kernels.h"
#include "lds_dma.h"
#include "wgrad_frag.h"

namespace unetdc {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned FOOB = 0x80000000u;

struct WgradFusedParams {
  const void* dy;   // [P][lddy], channels -> i
  const void* x;    // [P][ldx],  channels -> j
  float* part;      // [units][9][CI][CJ]
  int N, H, W, CI, CJ, lddy, ldx, d;
  int ysplit, rows_per_unit, itiles, jtiles;
  int imgs_per_unit;   // tap-split ring kernel with ysplit == 1: a workgroup walks this many images of its segment in turn
  int half_lds;        // paired form (HV = 2): LDS bytes of one half's ring
  // input normalisation ("bnin", tap-split ring kernel): x is the RAW conv output of the producing stage; every X row is
  // normalised in LDS, relu(in_scale * x + in_shift) rounded through bf16, once, when it has landed
  const float* in_scale;
  const float* in_shift;
};

#ifdef UNETDC_WGRAD_STAMPS
// DIAGNOSTIC BUILD ONLY (tools/probes/wgrad_stamps.sh; never part of libunetdc_hip.so): per wave, the shader-clock cycles of a
// step spent (0) in the counted vmcnt wait, (1) at the workgroup barrier, (2) from the barrier to the first MFMA (first fragment
// reads + DMA issue), (3) in the item loop, summed over the steps; [4] = steps.  Written once at the end, read by nothing.
__device__ unsigned long long g_wgrad_stamps[512][8][5];
extern "C" int unetdc_dbg_wgrad_stamps(unsigned long long* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wgrad_stamps), sizeof(g_wgrad_stamps));
}
#define STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif

template <typename T> struct FusedCfg;
template <> struct FusedCfg<bf16_t> { static constexpr int SEG = 64; };
template <> struct FusedCfg<float> { static constexpr int SEG = 32; };

template <typename T>
__global__ __launch_bounds__(256, 2) void wgrad_fused_kernel(const WgradFusedParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int SEG = FusedCfg<T>::SEG;
  constexpr int XR = SEG + 16;                         // X segment rows (x0-d .. x0+SEG-1+d, d <= 8)
  constexpr int ES = (int)sizeof(T);
  constexpr int RB = 64 * ES;                          // bytes per pixel row (64 channels)
  constexpr int CPR = RB / 16, RPI = 64 / CPR;         // lanes per row, rows per DMA instruction
  constexpr int DYI = SEG / RPI, XI = XR / RPI;        // DMA instructions: dY segment, one X segment
  constexpr int NSLOT = (DYI + 3 * XI + 3) / 4;        // per wave
  constexpr int DYB = SEG * RB, XB = XR * RB;
  constexpr int STAGE = DYB + 3 * XB;
  static_assert(SEG % RPI == 0 && XR % RPI == 0, "segment/instruction mismatch");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = lds_addr_of(smem);
  const int qi = wave >> 1, qj = wave & 1;
  // block -> (unit, it, jt); unit -> (image, x segment, y range)
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = p.itiles * p.jtiles;
  const int unit = L / tiles, trem = L - unit * tiles;
  const int it = trem / p.jtiles, jt = trem - it * p.jtiles;
  const int i0 = it * 64, j0 = jt * 64;
  const int segs = p.W / SEG;
  const int ys = unit % p.ysplit, strip = unit / p.ysplit;
  const int n = strip / segs, x0 = (strip - n * segs) * SEG;
  const int ybeg = ys * p.rows_per_unit;
  const int yend = min(ybeg + p.rows_per_unit, p.H);

  const unsigned dybytes = (unsigned)((long)p.N * p.H * p.W * p.lddy * ES);
  const unsigned xbytes = (unsigned)((long)p.N * p.H * p.W * p.ldx * ES);
  const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, dybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, xbytes, 0x00020000);

  // ---- DMA slots of this wave: slot q -> global instruction index gi = wave + 4*q --------------
  //   gi < DYI            : dY segment, instruction gi
  //   gi = DYI + ky*XI + k: X segment ky, instruction k
  const int sub = lane / CPR, pc = lane % CPR;
  unsigned colb[NSLOT];          // byte offset inside the image row (pixel * ld + channel chunk) or FOOB
#pragma unroll
  for (int q = 0; q < NSLOT; ++q) {
    const int gi = wave + 4 * q;
    if (gi < DYI) {
      const int row = gi * RPI + sub;                                  // pixel x0 + row
      const int c = Frag<T, 1>::src_chunk(row, pc);
      colb[q] = (unsigned)(((x0 + row) * p.lddy + i0) * ES + c * 16);
    } else if (gi < DYI + 3 * XI) {
      const int k = (gi - DYI) % XI;
      const int row = k * RPI + sub;                                   // pixel x0 - d + row
      const int gx = x0 - p.d + row;
      const int c = Frag<T, 1>::src_chunk(row, pc);
      colb[q] = ((unsigned)gx < (unsigned)p.W) ? (unsigned)((gx * p.ldx + j0) * ES + c * 16) : FOOB;
    } else {
      colb[q] = FOOB;
    }
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  auto issue = [&](int stage, int y) {
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) {
      const int gi = wave + 4 * q;                                     // wave-uniform
      if (gi < DYI) {
        const unsigned rowbase = (unsigned)((long)(n * p.H + y) * p.W * p.lddy * ES);
        lds_dma16(dyr, lds_base + stage * STAGE + gi * 1024, colb[q], rowbase);
      } else if (gi < DYI + 3 * XI) {
        const int ky = (gi - DYI) / XI, k = (gi - DYI) - ky * XI;
        const int yy = y + (ky - 1) * p.d;
        const bool yok = (unsigned)yy < (unsigned)p.H;
        const unsigned rowbase = (unsigned)((long)(n * p.H + (yok ? yy : 0)) * p.W * p.ldx * ES);
        const unsigned v = (yok && colb[q] != FOOB) ? colb[q] : FOOB;
        lds_dma16(xr, lds_base + stage * STAGE + DYB + ky * XB + k * 1024, v, rowbase);
      }
    }
  };

  const int nsteps = yend - ybeg;
  if (nsteps > 0) issue(0, ybeg);
  for (int s = 0; s < nsteps; ++s) {
    wait_vmcnt<0>();                                   // asm DMA + raw barrier: see lds_dma.h
    raw_barrier();
    if (s + 1 < nsteps) issue((s + 1) & 1, ybeg + s + 1);
    const unsigned char* sdy = smem + (s & 1) * STAGE;
    const unsigned char* sx = sdy + DYB;
#pragma unroll
    for (int k16 = 0; k16 < SEG / 16; ++k16) {
      if constexpr (sizeof(T) == 2) {
        const bf16x8 fa = Frag<bf16_t, 1>::frag(sdy, lane, qi * 32, 16 * k16);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int ky = t / 3, kx = t - ky * 3;
          const bf16x8 fb = Frag<bf16_t, 1>::frag(sx + ky * XB, lane, qj * 32, 16 * k16 + kx * p.d);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[t], 0, 0, 0);
        }
      } else {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) {
          const int krow = 16 * k16 + 2 * kp + h;
          const float fa = *reinterpret_cast<const float*>(sdy + krow * RB + (qi * 32 + r) * 4);
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
            const float fb = *reinterpret_cast<const float*>(sx + ky * XB + (krow + kx * p.d) * RB + (qj * 32 + r) * 4);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[t], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- partial slab: part[unit][t][i][j] ---------------------------------------------------------
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float* slab = p.part + ((long)unit * 9 + t) * p.CI * p.CJ;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int i = i0 + qi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h, j = j0 + qj * 32 + r;
      slab[(long)i * p.CJ + j] = acc[t][reg];
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// ------------------------------------------------------------------------------------------------
// Ring variant for d <= 2 (every decoder level, enc1, enc2: most of the weight-gradient time).
//
// SQ counters of the kernel above on enc1 (64->64, 512x512): MFMA pipe busy 40 %; a step lasts ~4.8K cycles of
// which the two resident waves of a SIMD keep the MFMA pipe busy 2.3K -- the rest is the LDS-DMA of the NEXT
// step (issued one step ahead, 38 KB) not having landed.  But consecutive image rows share their X rows:
// rows y-d, y, y+d of step s are rows y+1-2d.. of later steps, so the three X segments live in a RING of
// R = 2d + PF + 1 row slots and a step only fetches ONE new X row plus the dY row (18 KB instead of 38 KB),
// PF steps ahead (PF = 2 when the ring fits 80 KB so that two workgroups still share a CU).
//   group k (the loads of step k) = dY row ybeg+k -> dY slot k % (PF+1);  X row ybeg+k+d -> X slot (k+2d) % R
//   step s: wait for group s (counted vmcnt: the younger group may stay in flight) ; barrier (everyone's
//   group s has landed AND everyone is done with step s-1, whose oldest X row / dY slot are now free) ;
//   issue group s+PF into exactly those slots ; 36 MFMAs per wave.
template <typename T, int PF>
__global__ __launch_bounds__(256, 2) void wgrad_ring_kernel(const WgradFusedParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int SEG = FusedCfg<T>::SEG;
  constexpr int XR = SEG + 16;
  constexpr int ES = (int)sizeof(T);
  constexpr int RB = 64 * ES;
  constexpr int CPR = RB / 16, RPI = 64 / CPR;
  constexpr int DYI = SEG / RPI, XI = XR / RPI;
  constexpr int GI = DYI + XI;                         // DMA instructions of one group (whole workgroup)
  constexpr int NQ = (GI + 3) / 4;                     // slots per wave
  constexpr int DYB = SEG * RB, XB = XR * RB;
  constexpr int NDY = PF + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const xring = smem + NDY * DYB;
  const unsigned lds_base = lds_addr_of(smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = wave >> 1, qj = wave & 1;
  const int d = p.d, R = 2 * d + PF + 1;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = p.itiles * p.jtiles;
  const int unit = L / tiles, trem = L - unit * tiles;
  const int it = trem / p.jtiles, jt = trem - it * p.jtiles;
  const int i0 = it * 64, j0 = jt * 64;
  const int segs = p.W / SEG;
  const int ys = unit % p.ysplit, strip = unit / p.ysplit;
  const int n = strip / segs, x0 = (strip - n * segs) * SEG;
  const int ybeg = ys * p.rows_per_unit;
  const int yend = min(ybeg + p.rows_per_unit, p.H);
  const int nsteps = yend - ybeg;

  const unsigned dybytes = (unsigned)((long)p.N * p.H * p.W * p.lddy * ES);
  const unsigned xbytes = (unsigned)((long)p.N * p.H * p.W * p.ldx * ES);
  const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, dybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, xbytes, 0x00020000);

  // DMA slot q of this wave -> instruction gi = wave + 4q of a group: gi < DYI: dY row; else X row instruction gi - DYI
  const int sub = lane / CPR, pc = lane % CPR;
  unsigned colb[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int gi = wave + 4 * q;
    if (gi < DYI) {
      const int row = gi * RPI + sub;
      colb[q] = (unsigned)(((x0 + row) * p.lddy + i0) * ES + Frag<T, 1>::src_chunk(row, pc) * 16);
    } else if (gi < GI) {
      const int row = (gi - DYI) * RPI + sub;
      const int gx = x0 - d + row;
      colb[q] = ((unsigned)gx < (unsigned)p.W) ? (unsigned)((gx * p.ldx + j0) * ES + Frag<T, 1>::src_chunk(row, pc) * 16) : FOOB;
    } else {
      colb[q] = FOOB;
    }
  }
  const bool five = (wave + 4 * (NQ - 1)) < GI;          // does this wave's last slot exist? (wave-uniform)

  // X row `yy` of the image -> ring slot `slot` (zeros when the row lies outside the image)
  auto issue_x = [&](int slot, int yy) {
    const bool yok = (unsigned)yy < (unsigned)p.H;
    const unsigned rowbase = (unsigned)((long)(n * p.H + (yok ? yy : 0)) * p.W * p.ldx * ES);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int gi = wave + 4 * q;
      if (gi >= DYI && gi < GI) {
        const unsigned v = (yok && colb[q] != FOOB) ? colb[q] : FOOB;
        lds_dma16(xr, lds_base + NDY * DYB + slot * XB + (gi - DYI) * 1024, v, rowbase);
      }
    }
  };
  auto issue_dy = [&](int slot, int y) {
    const unsigned rowbase = (unsigned)((long)(n * p.H + y) * p.W * p.lddy * ES);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int gi = wave + 4 * q;
      if (gi < DYI) lds_dma16(dyr, lds_base + slot * DYB + gi * 1024, colb[q], rowbase);
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // prologue: X rows ybeg-d .. ybeg+d-1 (ring rows 0 .. 2d-1), then groups 0 .. PF-1
  for (int rho = 0; rho < 2 * d; ++rho) issue_x(rho, ybeg - d + rho);
  int gslot_x = 2 * d, gslot_dy = 0;                     // slots of the next group to issue
#pragma unroll
  for (int k = 0; k < PF; ++k) {
    if (k < nsteps) {
      issue_dy(gslot_dy, ybeg + k);
      issue_x(gslot_x, ybeg + k + d);
      gslot_dy = (gslot_dy + 1 == NDY) ? 0 : gslot_dy + 1;
      gslot_x = (gslot_x + 1 == R) ? 0 : gslot_x + 1;
    }
  }
  int sl0 = 0, sl1 = d, sl2 = 2 * d, sdy = 0;            // ring slots of rows y-d, y, y+d and the dY slot of step s
  for (int s = 0; s < nsteps; ++s) {
    if (PF == 2 && s + 1 < nsteps) {                     // the group of step s+1 may stay in flight
      if (five) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ - 1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // Raw barrier behind the counted wait (lds_dma.h: with the builtin DMA + __syncthreads() of round 1 the binary
    // drained the VM counter twice per step -- `vmcnt(4|5) ; vmcnt(0) ; s_barrier` here and another compiler-inserted
    // `vmcnt(0)` in front of the first fragment read -- so the group issued for step s+2 never stayed in flight).
    raw_barrier();
    if (s + PF < nsteps) {
      issue_dy(gslot_dy, ybeg + s + PF);
      issue_x(gslot_x, ybeg + s + PF + d);
      gslot_dy = (gslot_dy + 1 == NDY) ? 0 : gslot_dy + 1;
      gslot_x = (gslot_x + 1 == R) ? 0 : gslot_x + 1;
    }
    const unsigned char* sdyp = smem + sdy * DYB;
    const unsigned char* sx[3] = {xring + sl0 * XB, xring + sl1 * XB, xring + sl2 * XB};
    if constexpr (sizeof(T) == 2) {
      // Fragment reads two groups ahead of the MFMAs that consume them (group = one k16 x one X row = 3 taps).
      // Left to itself the compiler reads each fragment one MFMA before its use, and every MFMA then waits out the
      // LDS latency: 36 x ~130 cycles per step instead of 36 x 32.
      constexpr int NG = (SEG / 16) * 3;
      bf16x8 fa[2], fb[3][3];
      auto load_group = [&](int g) {
        const int k16 = g / 3, ky = g - k16 * 3;
        if (ky == 0) fa[k16 & 1] = Frag<bf16_t, 1>::frag(sdyp, lane, qi * 32, 16 * k16);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) fb[g % 3][kx] = Frag<bf16_t, 1>::frag(sx[ky], lane, qj * 32, 16 * k16 + kx * d);
      };
      load_group(0);
      load_group(1);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 2 < NG) load_group(g + 2);
        __builtin_amdgcn_sched_barrier(0);
        const int k16 = g / 3, ky = g - k16 * 3;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          acc[3 * ky + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k16 & 1], fb[g % 3][kx], acc[3 * ky + kx], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int k16 = 0; k16 < SEG / 16; ++k16) {
      if constexpr (sizeof(T) == 2) {
      } else {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) {
          const int krow = 16 * k16 + 2 * kp + h;
          const float fa = *reinterpret_cast<const float*>(sdyp + krow * RB + (qi * 32 + r) * 4);
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
            const float fb = *reinterpret_cast<const float*>(sx[ky] + (krow + kx * d) * RB + (qj * 32 + r) * 4);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[t], 0, 0, 0);
          }
        }
      }
    }
    sl0 = (sl0 + 1 == R) ? 0 : sl0 + 1;
    sl1 = (sl1 + 1 == R) ? 0 : sl1 + 1;
    sl2 = (sl2 + 1 == R) ? 0 : sl2 + 1;
    sdy = (sdy + 1 == NDY) ? 0 : sdy + 1;
  }

  // ---- partial slab: part[unit][t][i][j] ---------------------------------------------------------
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float* slab = p.part + ((long)unit * 9 + t) * p.CI * p.CJ + (long)(i0 + qi * 32 + 4 * h) * p.CJ + j0 + qj * 32 + r;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) slab[(long)((reg & 3) + 8 * (reg >> 2)) * p.CJ] = acc[t][reg];
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// ------------------------------------------------------------------------------------------------
// Tap-split ring kernel (bf16): same staging as wgrad_ring_kernel, different split of the work over the four waves.
//
// SQ counters of wgrad_ring_kernel (profiles/r02_sq_baseline.json, 128->128 @ 256x256): 2.22 LDS instructions per MFMA,
// MFMA pipe busy 50 %.  With one 32x32 (co x ci) quadrant and all nine taps per wave, every MFMA needs a B fragment of
// its own (two ds_read_b64_tr_b16 = 1 KB): 10 fragments per 9 MFMAs.  Four SIMDs x 1 KB per 32-cycle MFMA is more than the
// LDS delivers, so the matrix pipe waits on fragment reads half the time.
// Here a wave owns BOTH co halves (64 co x 32 ci) and HALF of the taps: a B fragment feeds two MFMAs.
//   wave = (qj: ci half, tg: tap group);  tg 0: taps 0..3 and tap 4 on the first two k16 steps of a row,
//                                         tg 1: taps 5..8 and tap 4 on the last two  (18 B fragments, 36 MFMAs each)
//   per k16 step: 2 A fragments + 4.5 B fragments for 9 MFMAs  ->  1.44 LDS instructions per MFMA.
// The two partial sums of tap 4 are added through LDS at the end, tg 0's + tg 1's (fixed order).
// (That first form ran on v_mfma_f32_32x32x16_bf16; since round 3 only the 16x16x32 form below is built.)
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// The same wave roles on v_mfma_f32_16x16x32_bf16 (M16 = true: the form that is built and launched).
// These kernels are power bound like the convolutions (igemm_dma16.hip): MI355X_MICROARCH.md measures 1.12-1.15x the FLOP/s
// for the 16x16x32 shape at equal cycles per FLOP with operands re-read from LDS.  LDS traffic per FLOP is unchanged (a
// fragment is still two transposed 8-byte reads per lane for 512 multiply-adds per lane), the accumulators are the same
// 160 registers: a wave owns 64 co x 32 ci = 4 x 2 tiles of 16 x 16 per tap slot.  A row of 64 pixels is two k32 steps;
// tg 0 takes taps 0..4 on the first and 0..3 on the second, tg 1 taps 5..8 and 4..8 (tap 4 split over the two, joined through
// LDS at the end as before).  Image swizzle and fragment addresses: wgrad_frag.h, Frag16.
// ------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int s16_k32(int tg, int i) { return tg == 0 ? (i < 5 ? 0 : 1) : (i < 4 ? 0 : 1); }
__host__ __device__ constexpr int s16_tap(int tg, int i) { return tg == 0 ? (i < 5 ? i : i - 5) : (i < 4 ? 5 + i : i); }

struct Split16Offs {       // per-lane byte offsets of the transposed reads at k32 step 0 (step 1: + 32 rows = 4096 bytes)
  int dy[4][2];            // [co tile][block jj]
  int x[3][2][2];          // [kx][ci tile][block jj], pixel rows shifted by kx * d
};

template <bool M16> struct SplitAcc;
template <> struct SplitAcc<true> { f32x4 a[5][4][2]; };

__device__ __forceinline__ void split_zero(SplitAcc<true>& A) {
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) A.a[t][c][j][e] = 0.f;
}

// `issue`: the row's DMA statements, placed behind the step's first fragment reads (those are in flight while the wave gets
// its DMA instructions accepted; round 4, as in igemm_lattice.hip)
template <int TG, typename Issue>
__device__ __forceinline__ void ring_split_step16(f32x4 (&acc)[5][4][2], const unsigned char* sdy, const unsigned char* sx0,
                                                  const unsigned char* sx1, const unsigned char* sx2, const Split16Offs& o,
                                                  Issue&& issue) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NI = 9;
  // A (dY) fragments: ONE set of four, refilled in place for the second k32 step -- each one right behind the two MFMAs of
  // the first step's last item that read it (an in-order wave: an issued MFMA has its operands) -- 16 registers instead of
  // 32: with a double set the kernel needs more than the 256 registers two workgroups per CU leave a wave.
  // B (X) fragments: two sets of two, item i + 1 loaded in front of the MFMAs of item i.
  bf16x8 fa[4], fb[2][2];
  auto load_a1 = [&](int k32, int c) { fa[c] = Frag16::frag_at(sdy + k32 * 4096, o.dy[c][0], o.dy[c][1]); };
  auto load_b = [&](int i) {
    const int k32 = s16_k32(TG, i), t = s16_tap(TG, i), ky = t / 3, kx = t - 3 * ky;
    const unsigned char* sx = (ky == 0 ? sx0 : (ky == 1 ? sx1 : sx2)) + k32 * 4096;
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[i & 1][j] = Frag16::frag_at(sx, o.x[kx][j][0], o.x[kx][j][1]);
  };
#pragma unroll
  for (int c = 0; c < 4; ++c) load_a1(0, c);
  load_b(0);
  issue();
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int k32 = s16_k32(TG, i);
    const bool turn = i + 1 < NI && s16_k32(TG, i + 1) != k32;      // last item of the first k32 step
    if (i + 1 < NI) load_b(i + 1);
    __builtin_amdgcn_sched_barrier(0);
    const int slot = s16_tap(TG, i) - (TG == 0 ? 0 : 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[slot][c][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[c], fb[i & 1][j], acc[slot][c][j], 0, 0, 0);
      if (turn) {
        __builtin_amdgcn_sched_barrier(0);
        load_a1(k32 + 1, c);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#endif
}

// one image row of the strip: both forms behind one name
template <int TG, typename Issue>
__device__ __forceinline__ void split_step(SplitAcc<true>& A, const unsigned char* sdy, const unsigned char* sx0,
                                           const unsigned char* sx1, const unsigned char* sx2, int, int, int,
                                           const Split16Offs& o, Issue&& issue) {
  ring_split_step16<TG>(A.a, sdy, sx0, sx1, sx2, o, issue);
}

// tap 4: tg 1's half (slot 0) joins tg 0's (slot 4) through LDS (fixed order), then the partial slab part[unit][t][i][j]:
// tg 0 stores taps 0..4, tg 1 taps 5..8.  The caller guarantees that every DMA has landed and every fragment has been read.
// PAIRED FORM (HV = 2, 512 threads): the workgroup is two halves of four waves with the roles above, each half walking its
// own half of the unit's image rows through its own LDS ring; here the second half hands its accumulators to the first
// through LDS (ex: [wave of the half][register quad][lane], 144 KB of the then idle rings) and only the first half stores.
// Why: the fp32 slabs are the accumulator state of the whole chip (2 x 256 threads x 160 registers per CU = 84 MB per launch),
// written once and re-read by the reduce kernel; two K ranges that meet in LDS halve both (round 4).
template <int HV>
__device__ __forceinline__ void split_finish(SplitAcc<true>& A, unsigned char* smem, unsigned char* smem_all, int half,
                                             const WgradFusedParams& p, int unit, int i0, int j0, int qj, int tg, int lane) {
  float* xch = reinterpret_cast<float*>(smem);             // [qj][co tile][ci tile][v][lane]: 16 KB
  __syncthreads();
  if (tg == 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 4; ++v) xch[(((qj * 4 + c) * 2 + j) * 4 + v) * 64 + lane] = A.a[0][c][j][v];
  }
  __syncthreads();
  if (tg == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 4; ++v) A.a[4][c][j][v] += xch[(((qj * 4 + c) * 2 + j) * 4 + v) * 64 + lane];
  }
  if (HV == 2) {
    __syncthreads();                                       // both halves are done with their tap-4 exchange areas
    f32x4* ex = reinterpret_cast<f32x4*>(smem_all) + (tg == 0 ? qj * 40 : 80 + qj * 32) * 64 + lane;
    if (half == 1) {
#pragma unroll
      for (int sl = 0; sl < 5; ++sl) {
        if (tg == 1 && sl == 0) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int j = 0; j < 2; ++j) ex[(((sl - (tg == 1 ? 1 : 0)) * 4 + c) * 2 + j) * 64] = A.a[sl][c][j];
      }
    }
    __syncthreads();
    if (half == 1) return;
#pragma unroll
    for (int sl = 0; sl < 5; ++sl) {
      if (tg == 1 && sl == 0) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 o = ex[(((sl - (tg == 1 ? 1 : 0)) * 4 + c) * 2 + j) * 64];
#pragma unroll
          for (int v = 0; v < 4; ++v) A.a[sl][c][j][v] += o[v];          // element by element: no packed-fp32 VALU (build guard)
        }
    }
  }
  // accumulator element v of a 16 x 16 tile: row (co) 4 * (lane >> 4) + v, column (ci) lane & 15
  const int col = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int sl = 0; sl < 5; ++sl) {
    if (tg == 1 && sl == 0) continue;
    const int t = sl + (tg == 0 ? 0 : 4);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float* slab = p.part + ((long)unit * 9 + t) * p.CI * p.CJ + (long)(i0 + 16 * c + 4 * rq) * p.CJ + j0 + qj * 32 + 16 * j + col;
#pragma unroll
        for (int v = 0; v < 4; ++v) slab[(long)v * p.CJ] = A.a[sl][c][j][v];
      }
  }
}

template <bool M16>
__device__ __forceinline__ Split16Offs split_offsets(int lane, int qj, int d) {
  Split16Offs o;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) o.dy[c][jj] = M16 ? Frag16::rd_off(lane, 16 * c, 0, jj) : 0;
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) o.x[kx][j][jj] = M16 ? Frag16::rd_off(lane, qj * 32 + 16 * j, kx * d, jj) : 0;
  return o;
}
template <bool M16> __device__ __forceinline__ int split_src_chunk(int row, int pc) {
  return M16 ? Frag16::src_chunk(row, pc) : Frag<bf16_t, 1>::src_chunk(row, pc);
}

template <int PF, int TG, bool M16, bool INORM, int HV>
__device__ __forceinline__ void ring_split_body(const WgradFusedParams& p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using T = bf16_t;
  constexpr int SEG = FusedCfg<T>::SEG;
  constexpr int XR = SEG + 16;
  constexpr int ES = (int)sizeof(T);
  constexpr int RB = 64 * ES;
  constexpr int CPR = RB / 16, RPI = 64 / CPR;
  constexpr int DYI = SEG / RPI, XI = XR / RPI;
  constexpr int GI = DYI + XI;
  constexpr int NQ = (GI + 3) / 4;
  constexpr int DYB = SEG * RB, XB = XR * RB;
  constexpr int NDY = PF + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  // paired form: tid / wave count inside the half, every LDS address is relative to the half's ring
  const int half = HV == 2 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
  unsigned char* const smem = smem_all + half * p.half_lds;
  unsigned char* const xring = smem + NDY * DYB;
  const unsigned lds_base = lds_addr_of(smem);

  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qj = wave & 1;
  constexpr int tg = TG;
  const int d = p.d, R = 2 * d + PF + 1;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = p.itiles * p.jtiles;
  const int unit = L / tiles, trem = L - unit * tiles;
  const int it = trem / p.jtiles, jt = trem - it * p.jtiles;
  const int i0 = it * 64, j0 = jt * 64;
  const int segs = p.W / SEG;
  const int ys = unit % p.ysplit, strip = unit / p.ysplit;
  // imgs_per_unit > 1 (only with ysplit == 1): the workgroup accumulates over several images of one segment, so a layer
  // with many channel tiles writes fewer fp32 slabs (1024 -> 512 channels at 64 x 64: 8 slabs of 18.9 MB were written and
  // re-read by the reduce kernel, and 1024 workgroups made two rounds on the chip)
  const int ipu = p.imgs_per_unit > 1 ? p.imgs_per_unit : 1;
  const int n0 = (strip / segs) * ipu, x0 = (strip - (strip / segs) * segs) * SEG;
  int n = n0;
  // paired form: the unit's rows are cut in two equal parts (the host only pairs when rows_per_unit is even and divides H:
  // both halves then run the same number of steps and meet at every workgroup barrier)
  const int rows_h = HV == 2 ? p.rows_per_unit / 2 : p.rows_per_unit;
  const int ybeg = ys * p.rows_per_unit + half * rows_h;
  const int yend = min(ybeg + rows_h, p.H);
  const int nsteps = yend - ybeg;

  const unsigned dybytes = (unsigned)((long)p.N * p.H * p.W * p.lddy * ES);
  const unsigned xbytes = (unsigned)((long)p.N * p.H * p.W * p.ldx * ES);
  const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, dybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, xbytes, 0x00020000);

  const int sub = lane / CPR, pc = lane % CPR;
  unsigned colb[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int gi = wave + 4 * q;
    if (gi < DYI) {
      const int row = gi * RPI + sub;
      colb[q] = (unsigned)(((x0 + row) * p.lddy + i0) * ES + split_src_chunk<M16>(row, pc) * 16);
    } else if (gi < GI) {
      const int row = (gi - DYI) * RPI + sub;
      const int gx = x0 - d + row;
      colb[q] = ((unsigned)gx < (unsigned)p.W) ? (unsigned)((gx * p.ldx + j0) * ES + split_src_chunk<M16>(row, pc) * 16) : FOOB;
    } else {
      colb[q] = FOOB;
    }
  }
  const bool five = (wave + 4 * (NQ - 1)) < GI;

  auto issue_x = [&](int slot, int yy) {
    const bool yok = (unsigned)yy < (unsigned)p.H;
    const unsigned rowbase = (unsigned)((long)(n * p.H + (yok ? yy : 0)) * p.W * p.ldx * ES);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int gi = wave + 4 * q;
      if (gi >= DYI && gi < GI) {
        const unsigned v = (yok && colb[q] != FOOB) ? colb[q] : FOOB;
        lds_dma16(xr, lds_base + NDY * DYB + slot * XB + (gi - DYI) * 1024, v, rowbase);
      }
    }
  };
  auto issue_dy = [&](int slot, int y) {
    const unsigned rowbase = (unsigned)((long)(n * p.H + y) * p.W * p.lddy * ES);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int gi = wave + 4 * q;
      if (gi < DYI) lds_dma16(dyr, lds_base + slot * DYB + gi * 1024, colb[q], rowbase);
    }
  };

  SplitAcc<M16> acc;
  split_zero(acc);
  const Split16Offs offs = split_offsets<M16>(lane, qj, d);
  // INORM: the 64 (scale, shift) pairs of this workgroup's input-channel tile live in LDS behind the ring (no vector-memory
  // operation may join the hand-counted DMA queue inside the row loop); a thread owns logical chunk tid & 7 of every 32nd
  // pixel of a row.  Rows / pixels outside the image were zero-filled by the DMA and stay zero (zero padding of the
  // ACTIVATION); the result is rounded through bf16 like a stored activation: bit-identical to the two-pass form.
  float* const ncst = reinterpret_cast<float*>(xring + R * XB);              // [2][64]
  if (INORM) {
    if (tid < 128) ncst[tid] = (tid < 64 ? p.in_scale : p.in_shift)[j0 + (tid & 63)];
    __syncthreads();
  }
  auto normalise_row = [&](int slot, int yy) {
    if ((unsigned)yy >= (unsigned)p.H) return;                               // a zero row (wave-uniform condition)
    const int lc = tid & 7;
    float nsc[8], nsh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { nsc[e] = ncst[lc * 8 + e]; nsh[e] = ncst[64 + lc * 8 + e]; }
#pragma unroll
    for (int i = 0; i < (XR + 31) / 32; ++i) {
      const int px = (tid >> 3) + 32 * i;
      const int gx = x0 - d + px;
      if (px < XR && (unsigned)gx < (unsigned)p.W) {
        unsigned char* a = xring + slot * XB + px * RB + (split_src_chunk<M16>(px, lc) << 4);
        float v[8];
        Chunk<bf16_t>::unpack(ld16(a), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], nsc[e], nsh[e]), 0.f);
        st16(a, Chunk<bf16_t>::pack(v));
      }
    }
  };

#ifdef UNETDC_WGRAD_STAMPS
  unsigned long long st_sum[5] = {0, 0, 0, 0, 0}, st_t2 = 0;
  const unsigned long long st_kernel0 = __builtin_amdgcn_s_memtime();
#endif
  for (int img = 0; img < ipu && n0 + img < p.N; ++img) {
  n = n0 + img;
  if (img > 0) raw_barrier();                            // the previous image's last fragments have been read by every wave
  for (int rho = 0; rho < 2 * d; ++rho) issue_x(rho, ybeg - d + rho);
  int gslot_x = 2 * d, gslot_dy = 0;
#pragma unroll
  for (int k = 0; k < PF; ++k) {
    if (k < nsteps) {
      issue_dy(gslot_dy, ybeg + k);
      issue_x(gslot_x, ybeg + k + d);
      gslot_dy = (gslot_dy + 1 == NDY) ? 0 : gslot_dy + 1;
      gslot_x = (gslot_x + 1 == R) ? 0 : gslot_x + 1;
    }
  }
  int sl0 = 0, sl1 = d, sl2 = 2 * d, sdy = 0;
  for (int s = 0; s < nsteps; ++s) {
#ifdef UNETDC_WGRAD_STAMPS
    STAMP(t0);
#endif
    if (PF == 2 && s + 1 < nsteps) {                     // the group of step s+1 may stay in flight (in-order retirement)
      if (five) wait_vmcnt<NQ>();
      else wait_vmcnt<NQ - 1>();
    } else {
      wait_vmcnt<0>();
    }
#ifdef UNETDC_WGRAD_STAMPS
    STAMP(t1);
#endif
    raw_barrier();
#ifdef UNETDC_WGRAD_STAMPS
    STAMP(t2);
    st_sum[0] += t1 - t0; st_sum[1] += t2 - t1; st_t2 = t2;
#endif
    if (INORM) {
      // the X rows that have landed with this step and were not normalised yet: at the first step of an image the 2d rows
      // issued up front and the row of group 0, afterwards the one new row y + d (ring slot sl2)
      if (s == 0) {
        for (int rho = 0; rho < 2 * d; ++rho) normalise_row(rho, ybeg - d + rho);
      }
      normalise_row(sl2, ybeg + s + d);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      raw_barrier();
    }
    auto issue_row = [&]() {
      if (s + PF < nsteps) {
        issue_dy(gslot_dy, ybeg + s + PF);
        issue_x(gslot_x, ybeg + s + PF + d);
        gslot_dy = (gslot_dy + 1 == NDY) ? 0 : gslot_dy + 1;
        gslot_x = (gslot_x + 1 == R) ? 0 : gslot_x + 1;
      }
    };
    const unsigned char* sdyp = smem + sdy * DYB;
    // (the input-normalising form keeps its DMA statements in front: behind the reads it measured 221.7 vs 214.5 us)
    if (INORM) {
      issue_row();
      split_step<TG>(acc, sdyp, xring + sl0 * XB, xring + sl1 * XB, xring + sl2 * XB, lane, qj, d, offs, []() {});
    } else {
      split_step<TG>(acc, sdyp, xring + sl0 * XB, xring + sl1 * XB, xring + sl2 * XB, lane, qj, d, offs, issue_row);
    }
#ifdef UNETDC_WGRAD_STAMPS
    {
      STAMP(t4);
      st_sum[3] += t4 - st_t2;                            // barrier -> end of the step (first reads + items)
      st_sum[4] += 1;
    }
#endif
    sl0 = (sl0 + 1 == R) ? 0 : sl0 + 1;
    sl1 = (sl1 + 1 == R) ? 0 : sl1 + 1;
    sl2 = (sl2 + 1 == R) ? 0 : sl2 + 1;
    sdy = (sdy + 1 == NDY) ? 0 : sdy + 1;
  }
  }                                                      // images of this unit
#ifdef UNETDC_WGRAD_STAMPS
  if (lane == 0 && blockIdx.x < 512) {
    unsigned long long* o = g_wgrad_stamps[blockIdx.x][half * 4 + wave];
    o[0] = st_sum[0]; o[1] = st_sum[1]; o[2] = st_kernel0; o[3] = st_sum[3]; o[4] = st_sum[4];
    o[2] = __builtin_amdgcn_s_memtime() - st_kernel0;     // loop time incl. prologue (before the exchange / slab stores)
  }
#endif

  // every DMA has landed (vmcnt(0) on the last step); join of tap 4, (paired form) of the two halves, and the slab stores
  split_finish<HV>(acc, smem, smem_all, half, p, unit, i0, j0, qj, tg, lane);
#endif  // __HIP_DEVICE_COMPILE__
}

template <int PF, bool M16, bool INORM = false, int HV = 1>
__global__ __launch_bounds__(256 * HV, 2) void wgrad_ring_split_kernel(const WgradFusedParams p) {
  // the tap group is wave-uniform: two specialisations of the whole body, so the 160 accumulator registers of a wave
  // never meet in a phi (a per-step branch made the allocator spill ~590 registers)
  if (__builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 7) & 1)) == 0) ring_split_body<PF, 0, M16, INORM, HV>(p);
  else ring_split_body<PF, 1, M16, INORM, HV>(p);
}

template <int TG, bool M16, int HV>
__device__ __forceinline__ void fused_split_body(const WgradFusedParams& p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using T = bf16_t;
  constexpr int SEG = FusedCfg<T>::SEG;
  constexpr int XR = SEG + 16;                         // X segment rows (x0-d .. x0+SEG-1+d, d <= 8)
  constexpr int ES = (int)sizeof(T);
  constexpr int RB = 64 * ES;                          // bytes per pixel row (64 channels)
  constexpr int CPR = RB / 16, RPI = 64 / CPR;         // lanes per row, rows per DMA instruction
  constexpr int DYI = SEG / RPI, XI = XR / RPI;        // DMA instructions: dY segment, one X segment
  constexpr int NSLOT = (DYI + 3 * XI + 3) / 4;        // per wave
  constexpr int DYB = SEG * RB, XB = XR * RB;
  constexpr int STAGE = DYB + 3 * XB;
  static_assert(SEG % RPI == 0 && XR % RPI == 0, "segment/instruction mismatch");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int half = HV == 2 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;      // paired form: see split_finish
  unsigned char* const smem = smem_all + half * p.half_lds;

  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = lds_addr_of(smem);
  const int qj = wave & 1;
  constexpr int tg = TG;
  // block -> (unit, it, jt); unit -> (image, x segment, y range)
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = p.itiles * p.jtiles;
  const int unit = L / tiles, trem = L - unit * tiles;
  const int it = trem / p.jtiles, jt = trem - it * p.jtiles;
  const int i0 = it * 64, j0 = jt * 64;
  const int segs = p.W / SEG;
  const int ys = unit % p.ysplit, strip = unit / p.ysplit;
  const int n = strip / segs, x0 = (strip - n * segs) * SEG;
  const int rows_h = HV == 2 ? p.rows_per_unit / 2 : p.rows_per_unit;
  const int ybeg = ys * p.rows_per_unit + half * rows_h;
  const int yend = min(ybeg + rows_h, p.H);

  const unsigned dybytes = (unsigned)((long)p.N * p.H * p.W * p.lddy * ES);
  const unsigned xbytes = (unsigned)((long)p.N * p.H * p.W * p.ldx * ES);
  const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, dybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, xbytes, 0x00020000);

  // ---- DMA slots of this wave: slot q -> global instruction index gi = wave + 4*q --------------
  //   gi < DYI            : dY segment, instruction gi
  //   gi = DYI + ky*XI + k: X segment ky, instruction k
  const int sub = lane / CPR, pc = lane % CPR;
  unsigned colb[NSLOT];          // byte offset inside the image row (pixel * ld + channel chunk) or FOOB
#pragma unroll
  for (int q = 0; q < NSLOT; ++q) {
    const int gi = wave + 4 * q;
    if (gi < DYI) {
      const int row = gi * RPI + sub;                                  // pixel x0 + row
      const int c = split_src_chunk<M16>(row, pc);
      colb[q] = (unsigned)(((x0 + row) * p.lddy + i0) * ES + c * 16);
    } else if (gi < DYI + 3 * XI) {
      const int k = (gi - DYI) % XI;
      const int row = k * RPI + sub;                                   // pixel x0 - d + row
      const int gx = x0 - p.d + row;
      const int c = split_src_chunk<M16>(row, pc);
      colb[q] = ((unsigned)gx < (unsigned)p.W) ? (unsigned)((gx * p.ldx + j0) * ES + c * 16) : FOOB;
    } else {
      colb[q] = FOOB;
    }
  }

  SplitAcc<M16> acc;
  split_zero(acc);
  const Split16Offs offs = split_offsets<M16>(lane, qj, p.d);

  auto issue = [&](int stage, int y) {
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) {
      const int gi = wave + 4 * q;                                     // wave-uniform
      if (gi < DYI) {
        const unsigned rowbase = (unsigned)((long)(n * p.H + y) * p.W * p.lddy * ES);
        lds_dma16(dyr, lds_base + stage * STAGE + gi * 1024, colb[q], rowbase);
      } else if (gi < DYI + 3 * XI) {
        const int ky = (gi - DYI) / XI, k = (gi - DYI) - ky * XI;
        const int yy = y + (ky - 1) * p.d;
        const bool yok = (unsigned)yy < (unsigned)p.H;
        const unsigned rowbase = (unsigned)((long)(n * p.H + (yok ? yy : 0)) * p.W * p.ldx * ES);
        const unsigned v = (yok && colb[q] != FOOB) ? colb[q] : FOOB;
        lds_dma16(xr, lds_base + stage * STAGE + DYB + ky * XB + k * 1024, v, rowbase);
      }
    }
  };

  const int nsteps = yend - ybeg;
  if (nsteps > 0) issue(0, ybeg);
  for (int s = 0; s < nsteps; ++s) {
    wait_vmcnt<0>();                                   // asm DMA + raw barrier: see lds_dma.h
    raw_barrier();
    auto issue_row = [&]() {
      if (s + 1 < nsteps) issue((s + 1) & 1, ybeg + s + 1);
    };
    const unsigned char* sdy = smem + (s & 1) * STAGE;
    const unsigned char* sx = sdy + DYB;
    split_step<TG>(acc, sdy, sx, sx + XB, sx + 2 * XB, lane, qj, p.d, offs, issue_row);
  }

  // every DMA has landed (vmcnt(0) on the last step); join of tap 4, (paired form) of the two halves, and the slab stores
  split_finish<HV>(acc, smem, smem_all, half, p, unit, i0, j0, qj, tg, lane);
#endif  // __HIP_DEVICE_COMPILE__
}

// tap-split form of the three-segment kernel (d = 4, 8): same staging, the wave roles of wgrad_ring_split_kernel
template <bool M16, int HV = 1>
__global__ __launch_bounds__(256 * HV, 2) void wgrad_fused_split_kernel(const WgradFusedParams p) {
  if (__builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 7) & 1)) == 0) fused_split_body<0, M16, HV>(p);
  else fused_split_body<1, M16, HV>(p);
}

// LDS bytes of the ring kernel, or 0 when the configuration does not leave room for two workgroups per CU
static int ring_lds(int d, int dtype, int pf) {
  const int es = dtype == UNETDC_BF16 ? 2 : 4, seg = dtype == UNETDC_BF16 ? 64 : 32;
  const int dyb = seg * 64 * es, xb = (seg + 16) * 64 * es;
  const int lds = (pf + 1) * dyb + (2 * d + pf + 1) * xb;
  return lds <= 80 * 1024 ? lds : 0;
}
static int ring_pf(int d, int dtype) {
  static int off = -1;                                   // UNETDC_WGRAD_RING=0: always the three-segment kernel (A/B)
  if (off < 0) { const char* e = getenv("UNETDC_WGRAD_RING"); off = (e && e[0] == '0') ? 1 : 0; }
  if (off || d > 2) return 0;
  if (ring_lds(d, dtype, 2)) return 2;
  if (ring_lds(d, dtype, 1)) return 1;
  return 0;
}

// ------------------------------------------------------------------------------------------------
static int fused_seg(int dtype) { return dtype == UNETDC_BF16 ? 64 : 32; }

bool wgrad_fused_supported(int N, int H, int W, int CI, int CJ, int lda, int ldb, int d, int ntaps, int stride,
                           int dtype) {
  if (ntaps != 9 || stride != 1 || d < 1 || d > 8) return false;
  if (CI % 64 != 0 || CJ % 64 != 0) return false;
  if (CI > 1024 || CJ > 1024) return false;
  if (W % fused_seg(dtype) != 0) return false;
  const long P = (long)N * H * W;
  const long minp = 32L * 1024;
  if (P < minp) return false;
  const long es = dtype == UNETDC_BF16 ? 2 : 4;
  return P * lda * es < (1L << 31) && P * ldb * es < (1L << 31);
}

static void fused_plan(int N, int H, int W, int CI, int CJ, int dtype, int& ysplit, int& rows) {
  const int strips = N * (W / fused_seg(dtype));
  const int tiles = (CI / 64) * (CJ / 64);
  int ys = 512 / (strips * tiles);                 // ~2 workgroups per CU
  if (ys < 1) ys = 1;
  if (ys > H / 8) ys = H / 8 > 0 ? H / 8 : 1;
  rows = (H + ys - 1) / ys;
  ysplit = (H + rows - 1) / rows;
}

long wgrad_fused_workspace_bytes(int N, int H, int W, int CI, int CJ, int dtype) {
  if (W < fused_seg(dtype) || W % fused_seg(dtype) != 0 || H < 1 || CI % 64 != 0 || CJ % 64 != 0) return 0;
  int ys, rows;
  fused_plan(N, H, W, CI, CJ, dtype, ys, rows);
  return (long)N * (W / fused_seg(dtype)) * ys * 9 * CI * CJ * 4;
}

// Fills the slabs; the caller reduces `units` slabs with wgrad_reduce_kernel.
bool wgrad_bnin_supported(int N, int H, int W, int CI, int CJ, int lda, int ldb, int d, int dtype) {
  static int splits = -1;
  if (splits < 0) { const char* e = getenv("UNETDC_WGRAD_SPLIT"); splits = (e && e[0] == '0') ? 0 : 1; }
  return dtype == UNETDC_BF16 && splits && wgrad_fused_supported(N, H, W, CI, CJ, lda, ldb, d, 9, 1, dtype) &&
         ring_pf(d, dtype) > 0;
}

int launch_wgrad_fused(const void* dy, int lddy, const void* x, int ldx, float* part, int N, int H, int W, int CI,
                       int CJ, int d, int dtype, int* units_out, hipStream_t stream, const float* in_scale,
                       const float* in_shift) {
  WgradFusedParams p{};
  p.dy = dy; p.x = x; p.part = part; p.N = N; p.H = H; p.W = W; p.CI = CI; p.CJ = CJ; p.lddy = lddy; p.ldx = ldx;
  p.d = d; p.in_scale = in_scale; p.in_shift = in_shift;
  fused_plan(N, H, W, CI, CJ, dtype, p.ysplit, p.rows_per_unit);
  p.itiles = CI / 64;
  p.jtiles = CJ / 64;
  const int units = N * (W / fused_seg(dtype)) * p.ysplit;
  *units_out = units;
  const long nwg = (long)units * p.itiles * p.jtiles;
  const int pf = ring_pf(d, dtype);
  static int split = -1;                                 // UNETDC_WGRAD_SPLIT=0: quadrant ring kernel (A/B)
  if (split < 0) { const char* e = getenv("UNETDC_WGRAD_SPLIT"); split = (e && e[0] == '0') ? 0 : 1; }
  // (the tap-split kernels run on v_mfma_f32_16x16x32_bf16: settled in round 3, profiles/r03_wgrad_m16_ab.txt; the 32x32x16
  //  instantiations are no longer built)
  if (in_scale && !(pf && dtype == UNETDC_BF16 && split)) {
    set_error("wgrad (bnin): the input-normalising form exists for the 16x16x32 tap-split ring kernel only");
    return UNETDC_EUNSUPPORTED;
  }
  // PAIRED FORM (round 4): 512-thread workgroups whose two halves walk the two halves of a unit's rows and meet in LDS
  // (split_finish): half as many fp32 slabs written and reduced.  The plan is the one above at a target of 256 workgroups
  // (one per CU: the two rings fill the LDS), i.e. every half does exactly the work a 256-thread workgroup did.
  // (three-segment staging, d = 4 / 8, waits for ALL its DMAs at every step: in lock step the two halves expose that wait
  //  together -- measured 155.4 vs 157.9 us at d = 4 but 175.3 vs 166.3 us at d = 8, profiles/r04_wgrad_pair_ab.txt)
  if (dtype == UNETDC_BF16 && split && (pf || d <= 4)) {
    const int strips = N * (W / fused_seg(dtype)), tiles = p.itiles * p.jtiles;
    int ys = 256 / (strips * tiles);
    if (ys < 1) ys = 1;
    if (ys > H / 16) ys = H / 16 > 0 ? H / 16 : 1;
    const int rows = (H + ys - 1) / ys;
    const int half_lds = pf ? ring_lds(d, dtype, pf) + (in_scale ? 512 : 0)
                            : 2 * (fused_seg(dtype) * 128 + 3 * (fused_seg(dtype) + 16) * 128);
    if (rows % 2 == 0 && H % rows == 0 && 2 * half_lds <= 160 * 1024) {
      WgradFusedParams q = p;
      q.rows_per_unit = rows;
      q.ysplit = H / rows;
      q.half_lds = half_lds;
      int un = strips * q.ysplit;
      long nwg2 = (long)un * tiles;
      if (pf && q.ysplit == 1 && nwg2 > 256 && N > 1) {   // several images per workgroup (as in the unpaired form, target 256)
        int ipu = (int)((nwg2 + 255) / 256);
        if (ipu > N) ipu = N;
        q.imgs_per_unit = ipu;
        un = (W / fused_seg(dtype)) * ((N + ipu - 1) / ipu);
        nwg2 = (long)un * tiles;
      }
      const int lds2 = 160 * 1024;                         // two rings, and 144 KB of them for the hand-over at the end
      const void* fn2;
      if (!pf) fn2 = reinterpret_cast<const void*>(&wgrad_fused_split_kernel<true, 2>);
      else if (in_scale) fn2 = pf == 2 ? reinterpret_cast<const void*>(&wgrad_ring_split_kernel<2, true, true, 2>)
                                       : reinterpret_cast<const void*>(&wgrad_ring_split_kernel<1, true, true, 2>);
      else fn2 = pf == 2 ? reinterpret_cast<const void*>(&wgrad_ring_split_kernel<2, true, false, 2>)
                         : reinterpret_cast<const void*>(&wgrad_ring_split_kernel<1, true, false, 2>);
      if (const int rc_ = ensure_dynamic_lds(fn2, lds2, "paired wgrad kernel")) return rc_;
      *units_out = un;
      const dim3 g((unsigned)nwg2), b(512);
      if (!pf) hipLaunchKernelGGL((wgrad_fused_split_kernel<true, 2>), g, b, lds2, stream, q);
      else if (in_scale && pf == 2) hipLaunchKernelGGL((wgrad_ring_split_kernel<2, true, true, 2>), g, b, lds2, stream, q);
      else if (in_scale) hipLaunchKernelGGL((wgrad_ring_split_kernel<1, true, true, 2>), g, b, lds2, stream, q);
      else if (pf == 2) hipLaunchKernelGGL((wgrad_ring_split_kernel<2, true, false, 2>), g, b, lds2, stream, q);
      else hipLaunchKernelGGL((wgrad_ring_split_kernel<1, true, false, 2>), g, b, lds2, stream, q);
      note_kernel(!pf ? "wgrad_fused_split_kernel<16x16x32> paired"
                      : (in_scale ? (pf == 2 ? "wgrad_ring_split_kernel<2, 16x16x32> paired bnin" : "wgrad_ring_split_kernel<1, 16x16x32> paired bnin")
                                  : (pf == 2 ? "wgrad_ring_split_kernel<2, 16x16x32> paired" : "wgrad_ring_split_kernel<1, 16x16x32> paired")));
      return check_launch("paired wgrad kernel");
    }
  }
  if (pf && dtype == UNETDC_BF16 && split) {
    const int lds = ring_lds(d, dtype, pf) + (in_scale ? 512 : 0);
    long nwg = (long)units * p.itiles * p.jtiles;
    if (p.ysplit == 1 && nwg > 512 && N > 1) {           // more than two workgroups per CU: walk several images per workgroup
      int ipu = (int)((nwg + 511) / 512);
      if (ipu > N) ipu = N;
      p.imgs_per_unit = ipu;
      *units_out = (W / fused_seg(dtype)) * ((N + ipu - 1) / ipu);
      nwg = (long)*units_out * p.itiles * p.jtiles;
    }
    const void* fn = pf == 2 ? reinterpret_cast<const void*>(&wgrad_ring_split_kernel<2, true>)
                             : reinterpret_cast<const void*>(&wgrad_ring_split_kernel<1, true>);
    if (const int rc_ = ensure_dynamic_lds(fn, 80 * 1024, "wgrad_ring_split_kernel")) return rc_;
    if (in_scale) {
      const void* nfn = pf == 2 ? reinterpret_cast<const void*>(&wgrad_ring_split_kernel<2, true, true>)
                                : reinterpret_cast<const void*>(&wgrad_ring_split_kernel<1, true, true>);
      if (const int rc_ = ensure_dynamic_lds(nfn, 96 * 1024, "wgrad_ring_split_kernel bnin")) return rc_;
      if (pf == 2) hipLaunchKernelGGL((wgrad_ring_split_kernel<2, true, true>), dim3((unsigned)nwg), dim3(256), lds, stream, p);
      else hipLaunchKernelGGL((wgrad_ring_split_kernel<1, true, true>), dim3((unsigned)nwg), dim3(256), lds, stream, p);
      note_kernel(pf == 2 ? "wgrad_ring_split_kernel<2, 16x16x32> bnin" : "wgrad_ring_split_kernel<1, 16x16x32> bnin");
      return check_launch("wgrad_ring_split_kernel(bnin)");
    }
    if (pf == 2) hipLaunchKernelGGL((wgrad_ring_split_kernel<2, true>), dim3((unsigned)nwg), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((wgrad_ring_split_kernel<1, true>), dim3((unsigned)nwg), dim3(256), lds, stream, p);
    note_kernel(pf == 2 ? "wgrad_ring_split_kernel<2, 16x16x32>" : "wgrad_ring_split_kernel<1, 16x16x32>");
    return check_launch("wgrad_ring_split_kernel");
  }
  if (pf) {
    const int lds = ring_lds(d, dtype, pf);
    const void* fn;
    if (dtype == UNETDC_BF16) fn = pf == 2 ? reinterpret_cast<const void*>(&wgrad_ring_kernel<bf16_t, 2>)
                                            : reinterpret_cast<const void*>(&wgrad_ring_kernel<bf16_t, 1>);
    else fn = pf == 2 ? reinterpret_cast<const void*>(&wgrad_ring_kernel<float, 2>)
                      : reinterpret_cast<const void*>(&wgrad_ring_kernel<float, 1>);
    if (const int rc_ = ensure_dynamic_lds(fn, 80 * 1024, "wgrad_ring_kernel")) return rc_;
    if (dtype == UNETDC_BF16) {
      if (pf == 2) hipLaunchKernelGGL((wgrad_ring_kernel<bf16_t, 2>), dim3((unsigned)nwg), dim3(256), lds, stream, p);
      else hipLaunchKernelGGL((wgrad_ring_kernel<bf16_t, 1>), dim3((unsigned)nwg), dim3(256), lds, stream, p);
    } else {
      if (pf == 2) hipLaunchKernelGGL((wgrad_ring_kernel<float, 2>), dim3((unsigned)nwg), dim3(256), lds, stream, p);
      else hipLaunchKernelGGL((wgrad_ring_kernel<float, 1>), dim3((unsigned)nwg), dim3(256), lds, stream, p);
    }
    char nm[64];
    snprintf(nm, sizeof(nm), "wgrad_ring_kernel<%s, %d>", dtype == UNETDC_BF16 ? "__bf16" : "float", pf);
    note_kernel(nm);
    return check_launch("wgrad_ring_kernel");
  }
  const int es = dtype == UNETDC_BF16 ? 2 : 4;
  const int seg = fused_seg(dtype);
  const int lds = 2 * (seg * 64 * es + 3 * (seg + 16) * 64 * es);
  if (dtype == UNETDC_BF16 && split) {                   // d = 4, 8 in bf16: tap-split wave roles on the three-segment staging
    if (const int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&wgrad_fused_split_kernel<true>), lds, "wgrad_fused_split_kernel"))
      return rc_;
    hipLaunchKernelGGL(wgrad_fused_split_kernel<true>, dim3((unsigned)nwg), dim3(256), lds, stream, p);
    note_kernel("wgrad_fused_split_kernel<16x16x32>");
    return check_launch("wgrad_fused_split_kernel");
  }
  const void* fn = dtype == UNETDC_BF16 ? reinterpret_cast<const void*>(&wgrad_fused_kernel<bf16_t>)
                                        : reinterpret_cast<const void*>(&wgrad_fused_kernel<float>);
  if (const int rc_ = ensure_dynamic_lds(fn, lds, "wgrad_fused_kernel")) return rc_;
  if (dtype == UNETDC_BF16)
    hipLaunchKernelGGL(wgrad_fused_kernel<bf16_t>, dim3((unsigned)nwg), dim3(256), lds, stream, p);
  else
    hipLaunchKernelGGL(wgrad_fused_kernel<float>, dim3((unsigned)nwg), dim3(256), lds, stream, p);
  note_kernel(dtype == UNETDC_BF16 ? "wgrad_fused_kernel<__bf16>" : "wgrad_fused_kernel<float>");
  return check_launch("wgrad_fused_kernel");
}

}  // namespace unetdc
