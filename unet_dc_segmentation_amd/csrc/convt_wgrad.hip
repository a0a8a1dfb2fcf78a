// Tap-fused weight gradient of ConvTranspose2d(k = 2, s = 2) (bf16), round 4.
//
//   dW[ci][co][a][b] = sum_{n, y, x} X[n, y, x, ci] * dY[n, 2y + a, 2x + b, co]          (autograd of nn.ConvTranspose2d,
//                                                                                          models/model_2.py:20-29, 67-76)
// As a GEMM this is [Cin] x [4 * Cout] with K = the input pixels: the four taps are four COLUMN BLOCKS that share the X
// operand.  The per-tap kernel (wgrad_dma.hip) ran them as four GEMMs and staged X four times (303-520 TFLOP/s on the four
// up-convolutions, 0.32 ms per step).  Here a workgroup owns 128 ci x 64 co x ALL FOUR taps:
//   * wave w of a half = tap w (a = w >> 1, b = w & 1): 8 x 4 accumulator tiles of 16 x 16 (128 registers); every wave reads
//     the same X fragments and its own tap's dY fragments -- X is staged once for four taps;
//   * K step = 32 input pixels of one image row (v_mfma_f32_16x16x32_bf16: one fragment each).  Per step the LDS stage holds
//     X as two [32 pixels][64 channels] images and dY as four TAP images [32 pixels][64 channels]: the dY row 2y + a is
//     fetched with a per-lane source pixel 2x + b, so in LDS every tap looks like a dense operand (LDS-DMA takes a per-lane
//     source address; every pixel is a whole 128-byte line, the two parities of a row are two DMA instructions over the same
//     lines) and the transposed fragment reads of wgrad_frag.h (Frag16) apply unchanged -- a stride-2 pixel walk in LDS
//     would put the four rows of a read into two bank windows instead of four;
//   * three-stage ring, DMAs from inline asm with counted waits (lds_dma.h): stage s + 2 is issued behind the barrier of step s;
//   * PAIRED K RANGES: 512 threads = two halves of four waves, each with its own ring and its own half of the workgroup's
//     pixels; the second half hands its accumulators to the first through LDS at the end, so a launch writes 256 slabs of
//     128 KB instead of 512 (the slabs are the accumulator state of the chip: see wgrad_fused.hip, split_finish).
// Slabs part[ks][t][ci][co] are summed by wgrad_reduce_kernel (fixed order: reproducible) into PyTorch's [Cin][Cout][2][2].
#include <stdio.h>
#include <stdlib.h>

#include "kernels.h"
#include "lds_dma.h"
#include "wgrad_frag.h"

namespace unetdc {

struct ConvtWgradParams {
  const void* x;     // [N*H*W][ldx]      input of the up-convolution, channels -> ci
  const void* dy;    // [N*2H*2W][lddy]   gradient of its output, channels -> co
  float* part;       // [ksplit][4][CI][CJ]
  int N, H, W, CI, CJ, ldx, lddy;
  int ksplit, steps_per_half, itiles, jtiles;
};

constexpr int CWG_STAGE = 24 * 1024;               // 2 X images + 4 tap images of 32 pixels x 128 bytes
constexpr int CWG_NS = 3;
constexpr int CWG_HALF = CWG_NS * CWG_STAGE;       // 72 KB per half
constexpr unsigned CWG_OOB = 0x80000000u;

__global__ __launch_bounds__(512, 2) void convt_wgrad_kernel(const ConvtWgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // = tap 2a + b
  unsigned char* const smem = smem_all + half * CWG_HALF;
  const unsigned lds_base = lds_addr_of(smem);

  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = p.itiles * p.jtiles;
  const int ks = L / tiles, trem = L - ks * tiles;
  const int it = trem / p.jtiles, jt = trem - it * p.jtiles;
  const int i0 = it * 128, j0 = jt * 64;
  const int nsteps = p.steps_per_half;
  const int s0 = (ks * 2 + half) * nsteps;                            // first 32-pixel step of this half

  const unsigned xbytes = (unsigned)((long)p.N * p.H * p.W * p.ldx * 2);
  const unsigned dybytes = (unsigned)((long)p.N * 4 * p.H * p.W * p.lddy * 2);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, dybytes, 0x00020000);

  // ---- DMA slots: a stage is 24 wave-instructions (8 pixel rows x 128 bytes each); wave w issues gi = w + 4 j, j = 0..5
  //   gi < 8 : X image gi >> 2, rows 8 (gi & 3) ..          gi >= 8: tap (gi - 8) >> 2, rows 8 ((gi - 8) & 3) ..
  const int sub = lane >> 3, pc = lane & 7;
  unsigned voff[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int gi = wave + 4 * j;
    const int r = ((gi < 8 ? gi : gi - 8) & 3) * 8 + sub;             // pixel of the step
    const unsigned ch = (unsigned)(Frag16::src_chunk(r, pc) * 16);
    if (gi < 8) voff[j] = (unsigned)((r * p.ldx + i0 + (gi >> 2) * 64) * 2) + ch;
    else voff[j] = (unsigned)(((2 * r + (((gi - 8) >> 2) & 1)) * p.lddy + j0) * 2) + ch;
  }
  const int HW = p.H * p.W;
  auto issue = [&](int stage, int S) {
    const int pp0 = S * 32;                                           // first input pixel of the step (one image row: W % 32 == 0)
    const int n = pp0 / HW, rem = pp0 - n * HW;
    const int y = rem / p.W, x0 = rem - y * p.W;
    const unsigned xs = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)pp0 * (unsigned)p.ldx * 2u));
    const unsigned dbase = (unsigned)(((n * 2 * p.H + 2 * y) * 2 * p.W + 2 * x0)) * (unsigned)p.lddy * 2u;
    const unsigned drow = (unsigned)(2 * p.W) * (unsigned)p.lddy * 2u;   // one output row further (a = 1)
    const unsigned d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)dbase);
    const unsigned d1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(dbase + drow));
    const unsigned sb = lds_base + (unsigned)__builtin_amdgcn_readfirstlane(stage) * CWG_STAGE;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int gi = wave + 4 * j;                                    // wave-uniform
      if (gi < 8) lds_dma16(xr, sb + gi * 1024, voff[j], xs);
      else lds_dma16(dyr, sb + gi * 1024, voff[j], ((gi - 8) >> 3) ? d1 : d0);
    }
  };

  // ---- fragment offsets (same image format for both operands): 16 channels from 16 c, 32 pixel rows
  int fo[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) fo[c][jj] = Frag16::rd_off(lane, 16 * c, 0, jj);

  f32x4 acc[8][4];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[c][j][e] = 0.f;

  issue(0, s0);
  if (nsteps > 1) issue(1, s0 + 1);
  int cur = 0, fill = 2;
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) wait_vmcnt<6>(); else wait_vmcnt<0>();       // this wave's pieces of stage s (stage s + 1 may stay in flight)
    raw_barrier();                                                    // everyone's; and everyone has issued the MFMAs of step s - 1
    const unsigned char* st = smem + cur * CWG_STAGE;
    const unsigned char* sd = st + 8192 + wave * 4096;
    bf16x8 fb[4], fa[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = Frag16::frag_at(sd, fo[j][0], fo[j][1]);
#pragma unroll
    for (int c = 0; c < 8; ++c) fa[c] = Frag16::frag_at(st + (c >> 2) * 4096, fo[c & 3][0], fo[c & 3][1]);
    if (s + 2 < nsteps) issue(fill, s0 + s + 2);                     // behind the fragment reads: in flight while the DMAs are accepted
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[c], fb[j], acc[c][j], 0, 0, 0);
    cur = cur + 1 == CWG_NS ? 0 : cur + 1;
    fill = fill + 1 == CWG_NS ? 0 : fill + 1;
  }

  // ---- the second half's accumulators join the first half's through LDS (fixed order), then the slab
  __syncthreads();                                                    // every DMA has landed (vmcnt(0) above), every fragment is read
  f32x4* ex = reinterpret_cast<f32x4*>(smem_all) + wave * 32 * 64 + lane;       // [tap][c][j][lane]: 128 KB
  if (half == 1) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) ex[(c * 4 + j) * 64] = acc[c][j];
  }
  __syncthreads();
  if (half == 1) return;
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 o = ex[(c * 4 + j) * 64];
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[c][j][v] += o[v];                 // element by element: no packed-fp32 VALU (build guard)
    }
  // accumulator element v of a 16 x 16 tile: row (ci) 4 * (lane >> 4) + v, column (co) lane & 15
  const int col = lane & 15, rq = lane >> 4;
  float* slab = p.part + ((long)ks * 4 + wave) * p.CI * p.CJ;
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* o = slab + (long)(i0 + 16 * c + 4 * rq) * p.CJ + j0 + 16 * j + col;
#pragma unroll
      for (int v = 0; v < 4; ++v) o[(long)v * p.CJ] = acc[c][j][v];
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ------------------------------------------------------------------------------------------------
static bool convt_wgrad_plan(int N, int H, int W, int CI, int CJ, int& ksplit, int& steps_per_half) {
  const long P = (long)N * H * W;
  if (W % 32 != 0 || CI % 128 != 0 || CJ % 64 != 0 || P % 64 != 0) return false;
  const long steps = P / 32;
  const int tiles = (CI / 128) * (CJ / 64);
  long ks = 256 / tiles;                          // one workgroup per CU (two rings of 72 KB)
  if (ks < 1) ks = 1;
  while (ks > 1 && (steps % (2 * ks) != 0 || steps / (2 * ks) < 8)) --ks;
  if (steps % (2 * ks) != 0 || steps / (2 * ks) < 2) return false;
  ksplit = (int)ks;
  steps_per_half = (int)(steps / (2 * ks));
  return true;
}

bool convt_wgrad_fused_supported(int N, int H, int W, int CI, int CJ, int ldx, int lddy, int dtype) {
  if (dtype != UNETDC_BF16) return false;
  int ks, sh;
  if (!convt_wgrad_plan(N, H, W, CI, CJ, ks, sh)) return false;
  const long P = (long)N * H * W;
  return P * ldx * 2 < (1L << 32) && 4 * P * lddy * 2 < (1L << 32) && ldx % 8 == 0 && lddy % 8 == 0;
}

long convt_wgrad_fused_workspace_bytes(int N, int H, int W, int CI, int CJ) {
  int ks, sh;
  if (!convt_wgrad_plan(N, H, W, CI, CJ, ks, sh)) return 0;
  return (long)ks * 4 * CI * CJ * 4;
}

// Fills ksplit slabs [4][CI][CJ]; the caller reduces them with wgrad_reduce_kernel.
int launch_convt_wgrad_fused(const void* x, int ldx, const void* dy, int lddy, float* part, int N, int H, int W, int CI, int CJ,
                             int* units_out, hipStream_t stream) {
  ConvtWgradParams p{};
  p.x = x; p.dy = dy; p.part = part; p.N = N; p.H = H; p.W = W; p.CI = CI; p.CJ = CJ; p.ldx = ldx; p.lddy = lddy;
  if (!convt_wgrad_plan(N, H, W, CI, CJ, p.ksplit, p.steps_per_half)) {
    set_error("convT wgrad (fused): unsupported shape");
    return UNETDC_EUNSUPPORTED;
  }
  p.itiles = CI / 128;
  p.jtiles = CJ / 64;
  constexpr int LDS = 2 * CWG_HALF;
  if (const int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&convt_wgrad_kernel), LDS, "convt_wgrad_kernel")) return rc_;
  *units_out = p.ksplit;
  const long nwg = (long)p.ksplit * p.itiles * p.jtiles;
  hipLaunchKernelGGL(convt_wgrad_kernel, dim3((unsigned)nwg), dim3(512), LDS, stream, p);
  note_kernel("convt_wgrad_kernel");
  return check_launch("convt_wgrad_kernel");
}

}  // namespace unetdc
