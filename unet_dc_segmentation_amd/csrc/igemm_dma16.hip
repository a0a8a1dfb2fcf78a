// bf16 implicit-GEMM convolution with LDS-DMA staging on the 16x16x32 MFMA shape.
//
// Same tiles, same LDS image, same pipeline as igemm_dma.hip (read that file first); what changes is the matrix
// instruction: v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16.  Why: the kernels are POWER bound,
// not issue bound -- with all-zero operands the identical instruction stream runs 27-33 % faster (1.02 -> 1.33
// PFLOP/s on 512->512, 1.24 -> 1.65 on the bottleneck layer; tools/op_bench.py ZERO=1), i.e. the chip lowers its
// clock under the switching activity of real data.  MI355X_MICROARCH.md (DVFS, item 7) measures ~1.12-1.15x the
// FLOP/s for the 16x16x32 shape at equal cycles per FLOP with operands re-read from LDS.  LDS traffic per FLOP is
// unchanged: a wave still owns a (TM*16) x 64 output tile and reads TM + 4 fragments per 32-channel group.
//
// Layouts (lane l, c = l & 15, rb = l >> 4):
//   A / B fragment : row (or output channel row) c of the 16-row tile, K elements 8*rb .. 8*rb+7 of the 32-group
//                    = 16-byte chunk 4*g + rb of the 128-byte LDS row (XOR-swizzled with (row>>1)&7 like before;
//                    the four 16-lane groups of a ds_read_b128 still land on 16 distinct bank quads);
//   accumulator    : acc[i][j][v] = out[pixel 16*i + 4*rb + v][channel 4*c + j]: the B rows are ordered so that
//                    the FOUR N tiles of a wave hold the four consecutive channels 4c..4c+3 -- a lane packs them
//                    into one 8-byte store and 16 lanes write one whole 128-byte pixel row.
//
// BLOCK ORDER OF THE M INDEX (quad_bpr > 0; strongly dilated layers, d a multiple of 16): a workgroup's 256 rows are one
// 16 x 16 pixel block of one image instead of 256 consecutive pixels of the flattened map.  A tap of dilation d moves such
// a block by whole blocks, so it lies either entirely inside the image or entirely outside: the block-level tap mask then
// skips EVERY padded tap-pixel pair -- d = 16 on the 32 x 32 bottleneck map runs 4 of 9 taps per block (6 of 9 with
// row-major blocks, which can only skip whole rows of taps) and 25 / 36 on a 64 x 64 map.  Only addresses change: an MFMA
// tile is still 16 consecutive pixels of one image row.
#include <stdio.h>
#include <stdlib.h>

#include "igemm_epilogue16.h"
#include "kernels.h"
#include "lds_dma.h"

namespace unetdc {

#define LDS_PTR16(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned OOB16 = 0x80000000u;

// WM x WN waves; each wave owns (TMT*16) x 64 outputs (TMT x 4 MFMA 16x16 tiles).
// NS = stages of the LDS ring.  The DMAs are issued from inline asm (lds_dma.h) and waited for by count: with NS = 3 the
// stage of step s + 2 goes out behind the barrier of step s and has TWO steps of MFMA work to land (L2 -> LDS takes
// 1.5-2 us for a 48-64 KB stage; one step of a 256 x 128 tile is 0.5-0.9 us of MFMA work); NS = 2 is the round-1 pipeline
// (issue s + 1 behind the barrier of step s, drain before the next barrier) for the tiles whose three stages do not fit.
template <int WM, int WN, int TMT, int NS>
__global__ __launch_bounds__(512, 2) void igemm_dma16_kernel(const IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TMT * 16, BN = WN * 64;
  constexpr int AI = BM / 8 / NW, BI = BN / 8 / NW;
  constexpr int ES = 2, KE = 64;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int PER = AI + BI;                     // DMA wave-instructions per wave and stage
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/wave mismatch");
  static_assert(NS == 2 || NS == 3, "ring depth");
  static_assert((NS - 2) * PER <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mblk = tile / p.nblocks, nblk = tile - mblk * p.nblocks;
  const int m0 = mblk * BM, n0 = nblk * BN;
  const int HoWo = p.Ho * p.Wo;
  const unsigned lds_base = lds_addr_of(smem);

  const unsigned xbytes = (unsigned)(((long)p.M / HoWo) * p.Hi * p.Wi * p.ldx * ES);
  const unsigned wbytes = (unsigned)((long)p.ntaps * p.Cout * p.Cin * ES);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, wbytes, 0x00020000);

  const bool p2 = p.wo_shift >= 0;
  const bool quad = p.quad_bpr > 0;
  auto decode = [&](int m, int& n, int& oy, int& ox) {
    if (quad) {
      const int blk = m >> 8, r = m & 255;
      n = blk / p.quad_bpi;
      const int b = blk - n * p.quad_bpi;
      const int qy = b / p.quad_bpr;
      oy = qy * 16 + (r >> 4);
      ox = (b - qy * p.quad_bpr) * 16 + (r & 15);
    } else if (p2) {
      n = m >> p.howo_shift;
      const int rem = m & (HoWo - 1);
      oy = rem >> p.wo_shift;
      ox = rem & (p.Wo - 1);
    } else {
      n = m / HoWo;
      const int rem = m - n * HoWo;
      oy = rem / p.Wo;
      ox = rem - oy * p.Wo;
    }
  };
  unsigned tapmask = 0;
  {
    int na, ya, xa, nb, yb, xb;
    const int mlast = (m0 + BM < p.M ? m0 + BM : p.M) - 1;
    decode(m0, na, ya, xa);
    decode(mlast, nb, yb, xb);
    int by0 = 0, by1 = p.Ho - 1, bx0 = 0, bx1 = p.Wo - 1;
    if (na == nb) {
      by0 = ya; by1 = yb;
      if (ya == yb || quad) { bx0 = xa; bx1 = xb; }       // block order: first and last pixel are opposite corners of the block
    }
    for (int t = 0; t < p.ntaps; ++t) {
      const int iy0 = by0 * p.stride + p.offy[t], iy1 = by1 * p.stride + p.offy[t];
      const int ix0 = bx0 * p.stride + p.offx[t], ix1 = bx1 * p.stride + p.offx[t];
      if (iy1 >= 0 && iy0 < p.Hi && ix1 >= 0 && ix0 < p.Wi) tapmask |= 1u << t;
    }
    tapmask = __builtin_amdgcn_readfirstlane(tapmask);
  }
  const int sub = lane >> 3, pc = lane & 7;
  int ys[AI], xs[AI];
  unsigned abase[AI];
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int row = (wave + NW * j) * 8 + sub;
    const int m = m0 + row;
    const int c = pc ^ ((row >> 1) & 7);
    if (m < p.M) {
      int n, oy, ox;
      decode(m, n, oy, ox);
      ys[j] = oy * p.stride;
      xs[j] = ox * p.stride;
      abase[j] = (unsigned)(((n * p.Hi + ys[j]) * p.Wi + xs[j]) * p.ldx * ES + c * 16);
    } else {
      ys[j] = -(1 << 28);
      xs[j] = 0;
      abase[j] = 0;
    }
  }
  // B rows: LDS row q of a 64-channel group = N tile (q >> 4), column (q & 15) holds output channel 4*(q & 15) + (q >> 4)
  unsigned bbase[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int lrow = (wave + NW * j) * 8 + sub;
    const int grp = lrow >> 6, q = lrow & 63;
    const int cc = (q & 15) * 4 + (q >> 4);
    const int c = pc ^ ((lrow >> 1) & 7);
    bbase[j] = (unsigned)((n0 + grp * 64 + cc) * p.Cin * ES + c * 16);
  }
  const int nkc = p.Cin / KE;
  const int nsteps = __popc(tapmask) * nkc;

  // fragment read offsets: row c of a tile, chunk 4*g + rb (swizzle is the same for every 16-row tile)
  const int c16 = lane & 15, rb = lane >> 4;
  const int swz = (c16 >> 1) & 7;
  int a_rd[2], b_rd[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int ch = ((4 * g + rb) ^ swz) << 4;
    a_rd[g] = (wm * TMT * 16 + c16) * 128 + ch;
    b_rd[g] = BM * 128 + (wn * 64 + c16) * 128 + ch;
  }

  f32x4 acc[TMT][4];
#pragma unroll
  for (int i = 0; i < TMT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  int lt = 0, lkc = 0;
  while (lt < p.ntaps && !((tapmask >> lt) & 1u)) ++lt;

  auto issue = [&](int stage) {
    const int dy = p.offy[lt], dx = p.offx[lt];
    const unsigned dbytes = (unsigned)((dy * p.Wi + dx) * p.ldx * ES + lkc * 128);
    const unsigned sbase = lds_base + (unsigned)__builtin_amdgcn_readfirstlane(stage) * STAGE;
#pragma unroll
    for (int j = 0; j < AI; ++j) {
      const int iy = ys[j] + dy, ix = xs[j] + dx;
      const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      lds_dma16(xr, sbase + (wave + NW * j) * 1024, ok ? abase[j] + dbytes : OOB16, 0u);
    }
    const unsigned wbytes_t = (unsigned)__builtin_amdgcn_readfirstlane((int)(lt * p.Cout * p.Cin * ES + lkc * 128));
#pragma unroll
    for (int j = 0; j < BI; ++j) lds_dma16(wr, sbase + BM * 128 + (wave + NW * j) * 1024, bbase[j], wbytes_t);
    // K order: 64-channel chunk OUTER, tap INNER.  The nine taps of one chunk re-read (shifted) the same 32 KB of
    // input, back to back, so they hit in the XCD's 4 MB L2; with the tap outer a workgroup streamed its whole
    // 256-pixel x Cin slab between two uses and every tap came from beyond L2 (PMC: 2.1x the algorithmic bytes).
    do { ++lt; } while (lt < p.ntaps && !((tapmask >> lt) & 1u));
    if (lt >= p.ntaps) {
      ++lkc;
      lt = 0;
      while (lt < p.ntaps && !((tapmask >> lt) & 1u)) ++lt;
    }
  };

  // stages 0 .. NS - 2 up front; iteration s: own pieces of stage s landed (the NS - 2 younger stages may stay in flight) ->
  // barrier (RAW for stage s, WAR for the buffer of step s - 1) -> issue stage s + NS - 1 into that buffer -> MFMAs of step s
#pragma unroll
  for (int k = 0; k < NS - 1; ++k)
    if (k < nsteps) issue(k);
  // OPPOSITE PHASES for the two waves of a SIMD (waves w and w + NW / 2): the first half issues its DMAs before its MFMAs, the
  // second half after them -- while one wave waits for the texture-address unit to take its DMA instructions the other one
  // feeds the matrix pipe (profiles/r03_dma16_step_decomposition.txt: in lock step the two costs add up, 0.45 + 0.21 us on
  // 0.70 us per step).  A stage issued late still has one whole step to land: three-stage ring only.
  const bool late = NS == 3 && wave >= NW / 2;
  int cur = 0, fill = NS - 1;
  for (int s = 0; s < nsteps; ++s) {
    if (NS > 2 && s + NS - 2 < nsteps) wait_vmcnt<(NS - 2) * PER>(); else wait_vmcnt<0>();
    raw_barrier();
    const bool more = s + NS - 1 < nsteps;
    const unsigned char* base = smem + cur * STAGE;
    {
      // ONE exposed LDS latency per step instead of two (round 4): the B fragments of BOTH 32-channel halves and the A
      // fragments of the first are read up front (in-order returns: the first MFMA waits for its own two operands only);
      // fa[i] is re-loaded with the second half's fragment right behind the four MFMAs that read it (A-fragment-major
      // order), so the second half starts with everything in registers.
      u32x4 fa[TMT], fb[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[0][j] = ld16(base + b_rd[0] + j * 16 * 128);
#pragma unroll
      for (int i = 0; i < TMT; ++i) fa[i] = ld16(base + a_rd[0] + i * 16 * 128);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[1][j] = ld16(base + b_rd[1] + j * 16 * 128);
      if (more && !late) issue(fill);              // behind the fragment reads: they are in flight while the DMAs are being accepted
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < TMT; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[g][j]),
                                                                acc[i][j], 0, 0, 0);
          if (g == 0) {
            fa[i] = ld16(base + a_rd[1] + i * 16 * 128);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
    if (more && late) issue(fill);
    cur = cur + 1 == NS ? 0 : cur + 1;
    fill = fill + 1 == NS ? 0 : fill + 1;
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------
  const int col = n0 + wn * 64 + 4 * c16;
  const unsigned ldob = (unsigned)(p.ldo * ES), ldyb = (unsigned)(p.bn_ldy * ES);
  bool tile_ok[TMT];
  unsigned voff[TMT], yoff[TMT];
  int ccol = col;
  unsigned row_bytes = ldob;
  if (p.mode == MODE_SHUFFLE) {
    const int ab = col / p.shuf_c;
    ccol = col - ab * p.shuf_c;
    row_bytes = 2 * ldob;
#pragma unroll
    for (int i = 0; i < TMT; ++i) {
      const int mb = m0 + (wm * TMT + i) * 16;
      tile_ok[i] = mb < p.M;
      int n, oy, ox;
      decode(mb, n, oy, ox);
      const unsigned pix = (unsigned)((n * 2 * p.Ho + 2 * oy + (ab >> 1)) * (2 * p.Wo) + 2 * ox + (ab & 1));
      voff[i] = (pix + 8u * rb) * ldob + (unsigned)(ccol * ES);
      yoff[i] = 0;
    }
  } else {
#pragma unroll
    for (int i = 0; i < TMT; ++i) {
      const int mb = m0 + (wm * TMT + i) * 16;
      tile_ok[i] = mb < p.M;
      int pix = mb;                                         // output pixel of the tile's first row (16 consecutive pixels either way)
      if (quad) {
        int n, oy, ox;
        decode(mb, n, oy, ox);
        pix = (n * p.Ho + oy) * p.Wo + ox;
      }
      voff[i] = (unsigned)(pix + 4 * rb) * ldob + (unsigned)(col * ES);
      yoff[i] = (unsigned)(pix + 4 * rb) * ldyb + (unsigned)(col * ES);
    }
  }
  float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
  switch (p.mode) {
    case MODE_STATS: epilogue16<MODE_STATS, TMT>(p, acc, tile_ok, voff, row_bytes, yoff, ldyb, ccol, s4, q4); break;
    case MODE_AFFINE_RELU: epilogue16<MODE_AFFINE_RELU, TMT>(p, acc, tile_ok, voff, row_bytes, yoff, ldyb, ccol, s4, q4); break;
    case MODE_BNBWD: epilogue16<MODE_BNBWD, TMT>(p, acc, tile_ok, voff, row_bytes, yoff, ldyb, ccol, s4, q4); break;
    default: epilogue16<MODE_STORE, TMT>(p, acc, tile_ok, voff, row_bytes, yoff, ldyb, ccol, s4, q4); break;
  }
  if (p.mode == MODE_STATS || p.mode == MODE_BNBWD) write_stat_rows16<WM, WN>(p, smem, s4, q4, mblk, n0, tid, wave, c16, rb);
#endif
}

// ------------------------------------------------------------------------------------------------
template <int WM, int WN, int TMT, int NS>
static int launch_dma16_cfg(IgemmParams& p, hipStream_t stream) {
  constexpr int BM = WM * TMT * 16, BN = WN * 64;
  constexpr int LDS = NS * (BM + BN) * 128;
  static_assert(LDS <= 160 * 1024, "ring does not fit");
  if (const int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&igemm_dma16_kernel<WM, WN, TMT, NS>), LDS, "igemm_dma16_kernel")) return rc_;
  p.mblocks = ceil_div(p.M, BM);
  p.nblocks = p.Cout / BN;
  // 16 x 16 pixel blocks as M tiles for strongly dilated 3 x 3 convolutions (header comment): every padded tap-pixel pair is skipped
  p.quad_bpr = p.quad_bpi = 0;
  if (BM == 256 && p.ntaps == 9 && p.stride == 1 && p.mode != MODE_SHUFFLE && p.Ho == p.Hi && p.Wo == p.Wi &&
      p.offy[8] >= 16 && p.offy[8] % 16 == 0 && p.offx[8] == p.offy[8] && p.Ho % 16 == 0 && p.Wo % 16 == 0 &&
      p.M % ((long)p.Ho * p.Wo) == 0) {
    p.quad_bpr = p.Wo / 16;
    p.quad_bpi = (p.Ho / 16) * p.quad_bpr;
  }
  const long nwg = (long)p.mblocks * p.nblocks;
  hipLaunchKernelGGL((igemm_dma16_kernel<WM, WN, TMT, NS>), dim3((unsigned)nwg), dim3(WM * WN * 64), LDS, stream, p);
  char nm[96];
  snprintf(nm, sizeof(nm), "igemm_dma16_kernel<%d, %d, %d>%s%s", WM, WN, TMT, NS == 3 ? " ring3" : "", p.quad_bpr ? " blocks16x16" : "");
  note_kernel(nm);
  return check_launch("igemm_dma16_kernel");
}

bool igemm_dma16_supported(const IgemmParams& p, int dtype) {
  if (dtype != UNETDC_BF16) return false;
  if (p.M % 16 != 0) return false;
  if (p.mode == MODE_SHUFFLE && p.Wo % 16 != 0) return false;
  return true;
}

// cfg: 1 = 256x256 (8 waves), 2 = 256x128 (8 waves), 3 = 256x64 (4 waves) -- chosen by launch_igemm_dma
int launch_igemm_dma16(IgemmParams& p, int cfg, hipStream_t stream) {
  // ring depth per tile (settled in round 3, profiles/r03_dma16_tile_configs.txt): three stages where they fit
  if (cfg == 1) return launch_dma16_cfg<2, 4, 8, 2>(p, stream);              // 3 x 64 KB does not fit
  if (cfg == 2) return launch_dma16_cfg<4, 2, 4, 3>(p, stream);
  return launch_dma16_cfg<4, 1, 4, 2>(p, stream);
}

}  // namespace unetdc
