// Halo-patch implicit GEMM for the narrow, huge-pixel layers (Cout = 64 / 128, dilation <= 2):
// dilated 3x3 conv forward and dgrad with the INPUT TILE STAGED ONCE IN LDS FOR ALL 9 TAPS.
//
// With per-tap reloads (igemm_dma.hip) a 256x64 tile moves (256+64)*128 B per 256*64*64 MACs =
// 80 B/clk/CU at full MFMA rate -- far beyond what L2 -> LDS sustains, so those layers ran at
// 350-500 TFLOP/s.  Here a workgroup owns an 8 x 32 pixel output tile; per 64-channel K chunk it
// DMAs the (8+2d) x (32+2d) input patch (zero outside the image via the descriptor range check)
// once, and the 9 taps read their A fragments from shifted rows of that patch.  Only the 8/16 KB
// weight tile changes per tap (double buffered).  Bytes per CU clock drop to ~25.
//
//   * patch rows are pixels in (py, px) row-major order, 128 B each, 16-byte chunks XOR-swizzled
//     with (row>>1)&7 on the source side (LDS-DMA is lane-linear);
//   * an MFMA 32-row tile is one image row of the tile (32 consecutive pixels): for tap (ky,kx)
//     lane r reads patch row (ty + ky*d)*PW + kx*d + r  -> consecutive rows, conflict-free;
//   * weights: per (tap, chunk) tile [BN][128 B], de-interleaved (even channels first) so each
//     lane owns two adjacent output channels (same epilogue as igemm_dma.hip);
//   * epilogues: +bias, +BatchNorm partial statistics (one row per tile = 256 pixels), folded
//     BN+ReLU (eval).
#include <stdio.h>
#include <stdlib.h>

#include "igemm_epilogue16.h"
#include "kernels.h"

namespace unetdc {

template <typename T> struct MmaH;
template <> struct MmaH<bf16_t> {
  __device__ static __forceinline__ void run(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                  acc, 0, 0, 0);
  }
};
template <> struct MmaH<float> {
  __device__ static __forceinline__ void run(f32x16& acc, const u32x4& a, const u32x4& b) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const unsigned int ua = a[s], ub = b[s];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bits_f32(ua), bits_f32(ub), acc, 0, 0, 0);
    }
  }
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned HOOB = 0x80000000u;
constexpr int TH = 8, TW = 32;                  // output tile in pixels
constexpr int MAXPJ = 14;                       // patch DMA instructions per wave (d <= 2, 4 waves: 54/4)

// WN = 1: 4 waves (4 x 1), BN = 64;   WN = 2: 8 waves (4 x 2), BN = 128.  Wave tile 64 x 64.
template <typename T, int WN>
__global__ __launch_bounds__(512, 2) void igemm_halo_kernel(const IgemmParams p, int d, int npatch_bufs, int rcp_pw) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = 4 * WN;
  constexpr int BN = 64 * WN;
  constexpr int BI = BN / 8 / NW;               // weight DMA instructions per wave per tap (= 2)
  constexpr int ES = (int)sizeof(T);
  constexpr int KE = 128 / ES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int PW = TW + 2 * d, PH = TH + 2 * d, PP = PH * PW;
  const int NPI = (PP + 7) / 8;                 // patch DMA wave-instructions
  const int PATCH = NPI * 1024;                 // bytes per patch buffer
  unsigned char* const bsm = smem + npatch_bufs * PATCH;       // 2 weight stages of BN*128 bytes

  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mtile = tile / p.nblocks, nblk = tile - mtile * p.nblocks;
  const int n0 = nblk * BN;
  // tile -> (image, y0, x0): shifts when the map is a power of two in both directions
  int img, y0, x0;
  if (p.wo_shift >= 0) {
    const int txs = p.wo_shift - 5, tis = p.howo_shift - 8;      // log2(tiles per row), log2(tiles per image)
    img = mtile >> tis;
    const int trem = mtile & ((1 << tis) - 1);
    y0 = (trem >> txs) * TH;
    x0 = (trem & ((1 << txs) - 1)) * TW;
  } else {
    const int tiles_x = p.Wo / TW, tiles_y = p.Ho / TH;
    img = mtile / (tiles_x * tiles_y);
    const int trem = mtile - img * tiles_x * tiles_y;
    y0 = (trem / tiles_x) * TH;
    x0 = (trem % tiles_x) * TW;
  }

  const unsigned xbytes = (unsigned)(((long)p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.ldx * ES);
  const unsigned wbytes = (unsigned)((long)9 * p.Cout * p.Cin * ES);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, wbytes, 0x00020000);

  // ---- patch rows this lane feeds: instruction slot j -> instruction (wave + NW*j), row 8*instr + lane/8
  const int sub = lane >> 3, pc = lane & 7;
  unsigned pbase[MAXPJ];
#pragma unroll
  for (int j = 0; j < MAXPJ; ++j) {
    const int pr = (wave + NW * j) * 8 + sub;
    const int py = (pr * rcp_pw) >> 16, px = pr - py * PW;       // exact for pr < 65536 / PW (host-checked)
    const int gy = y0 - d + py, gx = x0 - d + px;
    const bool ok = pr < PP && (unsigned)gy < (unsigned)p.Hi && (unsigned)gx < (unsigned)p.Wi;
    const int c = pc ^ ((pr >> 1) & 7);
    pbase[j] = ok ? (unsigned)((((img * p.Hi + gy) * p.Wi + gx) * p.ldx) * ES + c * 16) : HOOB;
  }
  unsigned bbase[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int lrow = (wave + NW * j) * 8 + sub;
    const int grp = lrow >> 6, q = lrow & 63;
    const int cc = (q & 31) * 2 + (q >> 5);
    const int c = pc ^ ((lrow >> 1) & 7);
    bbase[j] = (unsigned)((n0 + grp * 64 + cc) * p.Cin * ES + c * 16);
  }

  const int nkc = p.Cin / KE;
  const int r = lane & 31, h = lane >> 5;
  const int swz_r = (r >> 1) & 7;
  int b_rd[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) b_rd[g] = (wn * 64 + r) * 128 + (((2 * g + h) ^ swz_r) << 4);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto issue_patch = [&](int buf, int kc) {
    unsigned char* dst = smem + buf * PATCH;
#pragma unroll
    for (int j = 0; j < MAXPJ; ++j) {
      const int instr = wave + NW * j;
      if (instr < NPI) {                                     // wave-uniform
        const unsigned v = (pbase[j] == HOOB) ? HOOB : pbase[j] + (unsigned)(kc * 128);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, LDS_PTR(dst + instr * 1024), 16, v, 0, 0, 0);
      }
    }
  };
  auto issue_w = [&](int stage, int tap, int kc) {
    unsigned char* dst = bsm + stage * (BN * 128);
    const unsigned off = (unsigned)((tap * p.Cout * p.Cin) * ES + kc * 128);
#pragma unroll
    for (int j = 0; j < BI; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, LDS_PTR(dst + (wave + NW * j) * 1024), 16, bbase[j] + off, 0, 0, 0);
  };

  // step s = kc*9 + tap.  Weight tile of step s lives in stage s&1; patch of chunk kc in buffer
  // kc % npatch_bufs.
  const int nsteps = 9 * nkc;
  issue_patch(0, 0);
  issue_w(0, 0, 0);
  for (int s = 0; s < nsteps; ++s) {
    const int kc = s / 9, tap = s - kc * 9;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < nsteps) {
      const int kc1 = (s + 1) / 9, tap1 = (s + 1) - kc1 * 9;
      issue_w((s + 1) & 1, tap1, kc1);
      // next chunk's patch: prefetch right after its buffer is free
      if (npatch_bufs == 2) {
        if (tap == 0 && kc + 1 < nkc) issue_patch((kc + 1) & 1, kc + 1);
      }
    }
    const unsigned char* pb = smem + (npatch_bufs == 2 ? (kc & 1) : 0) * PATCH;
    const unsigned char* wb = bsm + (s & 1) * (BN * 128);
    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u32x4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int pr = ((wm * 2 + i) + ky * d) * PW + kx * d + r;
        a[i] = ld16(pb + pr * 128 + (((2 * g + h) ^ ((pr >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = ld16(wb + b_rd[g] + j * 32 * 128);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) MmaH<T>::run(acc[i][j], a[i], b[j]);
    }
    if (npatch_bufs == 1 && tap == 8 && kc + 1 < nkc) {
      __syncthreads();                                   // everyone is done with the single patch buffer
      issue_patch(0, kc + 1);
    }
  }

  // ---- epilogue (igemm_epilogue.h): an MFMA tile is 32 consecutive pixels of one image row ---------
  const int col = n0 + wn * 64 + 2 * r;
  const unsigned ldob = (unsigned)(p.ldo * ES), ldyb = (unsigned)(p.bn_ldy * ES);
  bool tile_ok[2];
  unsigned voff[2], yoff[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const unsigned rowpix = (unsigned)(((img * p.Ho + y0 + wm * 2 + mi) * p.Wo) + x0 + 4 * h);
    tile_ok[mi] = true;
    voff[mi] = rowpix * ldob + (unsigned)(col * ES);
    yoff[mi] = rowpix * ldyb + (unsigned)(col * ES);
  }
  float st[4] = {0.f, 0.f, 0.f, 0.f};
  switch (p.mode) {
    case MODE_STATS: epilogue_tiles<T, MODE_STATS, 2>(p, acc, tile_ok, voff, ldob, yoff, ldyb, col, st); break;
    case MODE_AFFINE_RELU: epilogue_tiles<T, MODE_AFFINE_RELU, 2>(p, acc, tile_ok, voff, ldob, yoff, ldyb, col, st); break;
    case MODE_BNBWD: epilogue_tiles<T, MODE_BNBWD, 2>(p, acc, tile_ok, voff, ldob, yoff, ldyb, col, st); break;
    default: epilogue_tiles<T, MODE_STORE, 2>(p, acc, tile_ok, voff, ldob, yoff, ldyb, col, st); break;
  }
  if (p.mode == MODE_STATS || p.mode == MODE_BNBWD) write_stat_rows<4, WN>(p, smem, st, mtile, n0, tid, wave, r, h);
#endif  // __HIP_DEVICE_COMPILE__
}

// ------------------------------------------------------------------------------------------------
// bf16 on the 16x16x32 MFMA shape (see igemm_dma16.hip for why): same patch, same weight ring, same taps.  A wave
// owns 2 image rows x 32 pixels x 64 channels = 4 M-tiles of 16 pixels (tile i: row i>>1, x offset 16*(i&1)) x 4
// N-tiles; fragments are chunk 4g + rb of the 128-byte LDS rows; epilogue igemm_epilogue16.h.
template <int WN>
__global__ __launch_bounds__(512, 2) void igemm_halo16_kernel(const IgemmParams p, int d, int npatch_bufs, int rcp_pw) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = 4 * WN;
  constexpr int BN = 64 * WN;
  constexpr int BI = BN / 8 / NW;
  constexpr int ES = 2, KE = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int PW = TW + 2 * d, PH = TH + 2 * d, PP = PH * PW;
  const int NPI = (PP + 7) / 8;
  const int PATCH = NPI * 1024;
  unsigned char* const bsm = smem + npatch_bufs * PATCH;

  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mtile = tile / p.nblocks, nblk = tile - mtile * p.nblocks;
  const int n0 = nblk * BN;
  int img, y0, x0;
  if (p.wo_shift >= 0) {
    const int txs = p.wo_shift - 5, tis = p.howo_shift - 8;
    img = mtile >> tis;
    const int trem = mtile & ((1 << tis) - 1);
    y0 = (trem >> txs) * TH;
    x0 = (trem & ((1 << txs) - 1)) * TW;
  } else {
    const int tiles_x = p.Wo / TW, tiles_y = p.Ho / TH;
    img = mtile / (tiles_x * tiles_y);
    const int trem = mtile - img * tiles_x * tiles_y;
    y0 = (trem / tiles_x) * TH;
    x0 = (trem % tiles_x) * TW;
  }

  const unsigned xbytes = (unsigned)(((long)p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.ldx * ES);
  const unsigned wbytes = (unsigned)((long)9 * p.Cout * p.Cin * ES);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, wbytes, 0x00020000);

  const int sub = lane >> 3, pc = lane & 7;
  unsigned pbase[MAXPJ];
#pragma unroll
  for (int j = 0; j < MAXPJ; ++j) {
    const int pr = (wave + NW * j) * 8 + sub;
    const int py = (pr * rcp_pw) >> 16, px = pr - py * PW;
    const int gy = y0 - d + py, gx = x0 - d + px;
    const bool ok = pr < PP && (unsigned)gy < (unsigned)p.Hi && (unsigned)gx < (unsigned)p.Wi;
    const int c = pc ^ ((pr >> 1) & 7);
    pbase[j] = ok ? (unsigned)((((img * p.Hi + gy) * p.Wi + gx) * p.ldx) * ES + c * 16) : HOOB;
  }
  // weight rows: LDS row q of a 64-channel group = N tile (q >> 4), column (q & 15) <-> output channel 4*(q & 15) + (q >> 4)
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
  const int c16 = lane & 15, rb = lane >> 4;
  int b_rd[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) b_rd[g] = (wn * 64 + c16) * 128 + (((4 * g + rb) ^ ((c16 >> 1) & 7)) << 4);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  auto issue_patch = [&](int buf, int kc) {
    unsigned char* dst = smem + buf * PATCH;
#pragma unroll
    for (int j = 0; j < MAXPJ; ++j) {
      const int instr = wave + NW * j;
      if (instr < NPI) {
        const unsigned v = (pbase[j] == HOOB) ? HOOB : pbase[j] + (unsigned)(kc * 128);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, LDS_PTR(dst + instr * 1024), 16, v, 0, 0, 0);
      }
    }
  };
  auto issue_w = [&](int stage, int tap, int kc) {
    unsigned char* dst = bsm + stage * (BN * 128);
    const unsigned off = (unsigned)((tap * p.Cout * p.Cin) * ES + kc * 128);
#pragma unroll
    for (int j = 0; j < BI; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, LDS_PTR(dst + (wave + NW * j) * 1024), 16, bbase[j] + off, 0, 0, 0);
  };

  const int nsteps = 9 * nkc;
  issue_patch(0, 0);
  issue_w(0, 0, 0);
  for (int s = 0; s < nsteps; ++s) {
    const int kc = s / 9, tap = s - kc * 9;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < nsteps) {
      const int kc1 = (s + 1) / 9, tap1 = (s + 1) - kc1 * 9;
      issue_w((s + 1) & 1, tap1, kc1);
      if (npatch_bufs == 2) {
        if (tap == 0 && kc + 1 < nkc) issue_patch((kc + 1) & 1, kc + 1);
      }
    }
    const unsigned char* pb = smem + (npatch_bufs == 2 ? (kc & 1) : 0) * PATCH;
    const unsigned char* wb = bsm + (s & 1) * (BN * 128);
    const int ky = tap / 3, kx = tap - ky * 3;
    int prow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) prow[i] = ((wm * 2 + (i >> 1)) + ky * d) * PW + kx * d + 16 * (i & 1) + c16;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      u32x4 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = ld16(pb + prow[i] * 128 + (((4 * g + rb) ^ ((prow[i] >> 1) & 7)) << 4));
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = ld16(wb + b_rd[g] + j * 16 * 128);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]),
                                                              acc[i][j], 0, 0, 0);
    }
    if (npatch_bufs == 1 && tap == 8 && kc + 1 < nkc) {
      __syncthreads();
      issue_patch(0, kc + 1);
    }
  }

  const int col = n0 + wn * 64 + 4 * c16;
  const unsigned ldob = (unsigned)(p.ldo * ES), ldyb = (unsigned)(p.bn_ldy * ES);
  bool tile_ok[4];
  unsigned voff[4], yoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned rowpix = (unsigned)(((img * p.Ho + y0 + wm * 2 + (i >> 1)) * p.Wo) + x0 + 16 * (i & 1) + 4 * rb);
    tile_ok[i] = true;
    voff[i] = rowpix * ldob + (unsigned)(col * ES);
    yoff[i] = rowpix * ldyb + (unsigned)(col * ES);
  }
  float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
  switch (p.mode) {
    case MODE_STATS: epilogue16<MODE_STATS, 4>(p, acc, tile_ok, voff, ldob, yoff, ldyb, col, s4, q4); break;
    case MODE_AFFINE_RELU: epilogue16<MODE_AFFINE_RELU, 4>(p, acc, tile_ok, voff, ldob, yoff, ldyb, col, s4, q4); break;
    case MODE_BNBWD: epilogue16<MODE_BNBWD, 4>(p, acc, tile_ok, voff, ldob, yoff, ldyb, col, s4, q4); break;
    default: epilogue16<MODE_STORE, 4>(p, acc, tile_ok, voff, ldob, yoff, ldyb, col, s4, q4); break;
  }
  if (p.mode == MODE_STATS || p.mode == MODE_BNBWD) write_stat_rows16<4, WN>(p, smem, s4, q4, mtile, n0, tid, wave, c16, rb);
#endif
}

// ------------------------------------------------------------------------------------------------
bool igemm_halo_supported(const IgemmParams& p, int dtype) {
  if (p.ntaps != 9 || p.stride != 1 || p.mode == MODE_SHUFFLE) return false;
  if (p.Ho != p.Hi || p.Wo != p.Wi) return false;
  const int d = p.offy[8];                       // taps are (ky-1)*d, (kx-1)*d
  if (d < 1 || d > 2) return false;
  if (p.offx[8] != d || p.offy[0] != -d || p.offx[0] != -d) return false;
  if (!(p.Cout == 64 || p.Cout == 128)) return false;
  if (p.Ho % TH != 0 || p.Wo % TW != 0) return false;
  if ((long)p.M < 256L * 512) return false;      // small maps: not worth a patch per tile
  const long es = dtype == UNETDC_BF16 ? 2 : 4;
  const long xbytes = ((long)p.M / ((long)p.Ho * p.Wo)) * p.Hi * p.Wi * p.ldx * es;
  const long obytes = (long)p.M * p.ldo * es, ybytes = p.mode == MODE_BNBWD ? (long)p.M * p.bn_ldy * es : 0;
  return xbytes < (1L << 31) && obytes < (1L << 32) && ybytes < (1L << 32);
}

static bool halo_mfma16() { return true; }            // bf16: the 16x16x32 form (the A/B against 32x32x16 was settled in round 1)

template <typename T, int WN>
static int launch_halo_cfg(IgemmParams& p, int d, hipStream_t stream) {
  const int PP = (TH + 2 * d) * (TW + 2 * d);
  const int patch = ((PP + 7) / 8) * 1024;
  const int es = (int)sizeof(T);
  const int nkc = p.Cin / (128 / es);
  // single patch buffer for the 4-wave config (2 workgroups per CU hide the reload); the 8-wave
  // config is alone on its CU, so it double-buffers the patch when there is more than one chunk
  const int nbuf = (WN == 2 && nkc > 1) ? 2 : 1;
  const int lds = nbuf * patch + 2 * (64 * WN) * 128;
  if (const int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&igemm_halo_kernel<T, WN>), 160 * 1024, "igemm_halo_kernel")) return rc_;
  p.mblocks = (int)((long)p.M / 256);
  p.nblocks = p.Cout / (64 * WN);
  const long nwg = (long)p.mblocks * p.nblocks;
  const int rcp_pw = (65536 + (TW + 2 * d) - 1) / (TW + 2 * d);     // pr / PW == (pr * rcp_pw) >> 16 for pr < 1024 (PW <= 36)
  char nm[96];
  if (sizeof(T) == 2 && halo_mfma16()) {
    if (const int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&igemm_halo16_kernel<WN>), 160 * 1024, "igemm_halo16_kernel"))
      return rc_;
    hipLaunchKernelGGL((igemm_halo16_kernel<WN>), dim3((unsigned)nwg), dim3(256 * WN), lds, stream, p, d, nbuf, rcp_pw);
    snprintf(nm, sizeof(nm), "igemm_halo16_kernel<%d>", WN);
    note_kernel(nm);
    return check_launch("igemm_halo16_kernel");
  }
  hipLaunchKernelGGL((igemm_halo_kernel<T, WN>), dim3((unsigned)nwg), dim3(256 * WN), lds, stream, p, d, nbuf, rcp_pw);
  snprintf(nm, sizeof(nm), "igemm_halo_kernel<%s, %d>", sizeof(T) == 2 ? "__bf16" : "float", WN);
  note_kernel(nm);
  return check_launch("igemm_halo_kernel");
}

int launch_igemm_halo(IgemmParams& p, int dtype, hipStream_t stream) {
  const int d = p.offy[8];
  if (p.Cout % 128 == 0)
    return dtype == UNETDC_BF16 ? launch_halo_cfg<bf16_t, 2>(p, d, stream) : launch_halo_cfg<float, 2>(p, d, stream);
  return dtype == UNETDC_BF16 ? launch_halo_cfg<bf16_t, 1>(p, d, stream) : launch_halo_cfg<float, 1>(p, d, stream);
}

}  // namespace unetdc
