// Second-generation implicit-GEMM convolution: same math and epilogues as igemm_conv.hip, but the
// A (shifted pixels) and B (weights) tiles travel HBM/L2 -> LDS by LDS-DMA
// (`buffer_load_dwordx4 ... offen lds`), never through VGPRs:
//   * no ds_write pass (register->LDS stores top out at ~79 B/clk/CU on gfx950 and made the
//     first-generation 128x128 tile LDS-pipe-bound at 23 % of the MFMA peak),
//   * zero padding for free: a lane whose shifted pixel falls outside the image gets a byte offset
//     beyond the buffer descriptor's range, and the hardware writes zeros into LDS for it
//     (verified on MI355X: tools/probes/dma_oob.hip),
//   * the LDS image of a DMA is lane-linear (base + lane*16 B): 8 lanes cover one 128-byte row, one
//     wave-instruction covers 8 rows; the bank-conflict XOR swizzle is therefore applied on the
//     SOURCE side (lane with physical chunk pc of row R fetches logical chunk pc ^ ((R>>1)&7)),
//     and the same XOR on the ds_read_b128 fragment reads,
//   * 256-pixel block tiles (8 waves for the wide configurations) halve the bytes per FLOP that
//     have to cross L2 -> LDS compared with 128x128.
// Pipeline: two LDS stages, ONE barrier per K-step:  wait(my DMAs of step s) ; barrier ;
// issue DMAs of step s+1 into the other stage ; ds_read + MFMA on stage s.
#include <stdio.h>
#include <stdlib.h>

#include "igemm_epilogue.h"
#include "kernels.h"

namespace unetdc {

template <typename T> struct MmaD;
template <> struct MmaD<bf16_t> {
  __device__ static __forceinline__ void run(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                  acc, 0, 0, 0);
  }
};
template <> struct MmaD<float> {
  __device__ static __forceinline__ void run(f32x16& acc, const u32x4& a, const u32x4& b) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const unsigned int ua = a[s], ub = b[s];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bits_f32(ua), bits_f32(ub), acc, 0, 0, 0);
    }
  }
};

template <typename T> __device__ __forceinline__ void store_pair_d(T* dst, float v0, float v1);
template <> __device__ __forceinline__ void store_pair_d<float>(float* dst, float v0, float v1) {
  *reinterpret_cast<float2*>(dst) = make_float2(v0, v1);
}
template <> __device__ __forceinline__ void store_pair_d<bf16_t>(bf16_t* dst, float v0, float v1) {
  bf16_t lo = (bf16_t)v0, hi = (bf16_t)v1;
  *reinterpret_cast<unsigned int*>(dst) = (unsigned int)__builtin_bit_cast(unsigned short, lo) |
                                          ((unsigned int)__builtin_bit_cast(unsigned short, hi) << 16);
}

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned OOB = 0x80000000u;       // byte offset beyond any descriptor range (< 2 GiB tensors)

// WM x WN waves; each wave owns (TM*32) x 64 outputs (TM x 2 MFMA 32x32 tiles).
template <typename T, int WM, int WN, int TM>
__global__ __launch_bounds__(512, 2) void igemm_dma_kernel(const IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource builtins exist only in the device pass
  constexpr int NW = WM * WN;                   // waves per block
  constexpr int BM = WM * TM * 32, BN = WN * 64;
  constexpr int AI = BM / 8 / NW, BI = BN / 8 / NW;   // DMA wave-instructions per wave per K-step
  constexpr int ES = (int)sizeof(T);
  constexpr int KE = 128 / ES;                  // channels per K-step
  constexpr int STAGE = (BM + BN) * 128;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/wave mismatch");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mblk = tile / p.nblocks, nblk = tile - mblk * p.nblocks;
  const int m0 = mblk * BM, n0 = nblk * BN;
  const int HoWo = p.Ho * p.Wo;

  // buffer descriptors (wave-uniform: built from kernel arguments only)
  const unsigned xbytes = (unsigned)(((long)p.M / HoWo) * p.Hi * p.Wi * p.ldx * ES);      // upper bound on the view
  const unsigned wbytes = (unsigned)((long)p.ntaps * p.Cout * p.Cin * ES);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, wbytes, 0x00020000);

  // pixel index -> (image, y, x): shifts when the map sizes are powers of two (every U-Net level), else divisions
  const bool p2 = p.wo_shift >= 0;
  auto decode = [&](int m, int& n, int& oy, int& ox) {
    if (p2) {
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
  // ---- taps that can touch the image for ANY row of this block (wave-uniform, conservative): a block is a
  // segment of one image row, or whole rows of one image, or spans images
  unsigned tapmask = 0;
  {
    int na, ya, xa, nb, yb, xb;
    const int mlast = (m0 + BM < p.M ? m0 + BM : p.M) - 1;
    decode(m0, na, ya, xa);
    decode(mlast, nb, yb, xb);
    int by0 = 0, by1 = p.Ho - 1, bx0 = 0, bx1 = p.Wo - 1;
    if (na == nb) {
      by0 = ya; by1 = yb;
      if (ya == yb) { bx0 = xa; bx1 = xb; }
    }
    for (int t = 0; t < p.ntaps; ++t) {
      const int iy0 = by0 * p.stride + p.offy[t], iy1 = by1 * p.stride + p.offy[t];
      const int ix0 = bx0 * p.stride + p.offx[t], ix1 = bx1 * p.stride + p.offx[t];
      if (iy1 >= 0 && iy0 < p.Hi && ix1 >= 0 && ix0 < p.Wi) tapmask |= 1u << t;
    }
    tapmask = __builtin_amdgcn_readfirstlane(tapmask);
  }
  // ---- per-lane description of the A rows this lane feeds (row = 8*instr + lane/8) -------------
  const int sub = lane >> 3, pc = lane & 7;
  int ys[AI], xs[AI];
  unsigned abase[AI];                            // byte offset of (pixel, swizzled chunk) at tap 0,0 / kc 0
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
  // B rows: LDS row lrow holds output channel grp*64 + 2*(q&31) + (q>>5) (even channels first)
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
  const int nsteps = __popc(tapmask) * nkc;

  // ---- fragment read offsets --------------------------------------------------------------------
  const int r = lane & 31, h = lane >> 5;
  const int swz_r = (r >> 1) & 7;
  int a_rd[4], b_rd[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int ch = ((2 * g + h) ^ swz_r) << 4;
    a_rd[g] = (wm * TM * 32 + r) * 128 + ch;
    b_rd[g] = BM * 128 + (wn * 64 + r) * 128 + ch;
  }

  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int lt = 0, lkc = 0;
  while (lt < p.ntaps && !((tapmask >> lt) & 1u)) ++lt;

  auto issue = [&](int stage) {
    const int dy = p.offy[lt], dx = p.offx[lt];
    const unsigned dbytes = (unsigned)((dy * p.Wi + dx) * p.ldx * ES + lkc * 128);
    unsigned char* sbase = smem + stage * STAGE;
#pragma unroll
    for (int j = 0; j < AI; ++j) {
      const int iy = ys[j] + dy, ix = xs[j] + dx;
      const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      const unsigned voff = ok ? abase[j] + dbytes : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, LDS_PTR(sbase + (wave + NW * j) * 1024), 16, voff, 0, 0, 0);
    }
    const unsigned wbytes_t = (unsigned)(lt * p.Cout * p.Cin * ES + lkc * 128);
#pragma unroll
    for (int j = 0; j < BI; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, LDS_PTR(sbase + BM * 128 + (wave + NW * j) * 1024), 16,
                                               bbase[j] + wbytes_t, 0, 0, 0);
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

  if (nsteps > 0) issue(0);
  for (int s = 0; s < nsteps; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my DMAs of step s have landed
    __syncthreads();                                       // ... everyone's; and stage (s+1)&1 is free
    if (s + 1 < nsteps) issue((s + 1) & 1);
    const unsigned char* base = smem + (s & 1) * STAGE;
    // fragments of 16-channel group g+1 are read while the MFMAs of group g run (double-buffered registers,
    // order pinned with sched_barrier): left alone, the compiler reads each fragment right before its first
    // MFMA and every group of 2-4 MFMAs then waits out the LDS latency
    u32x4 fa[2][TM], fb[2][2];
    auto load_g = [&](int buf, int g) {
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[buf][j] = ld16(base + b_rd[g] + j * 32 * 128);
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[buf][i] = ld16(base + a_rd[g] + i * 32 * 128);
    };
    load_g(0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g + 1 < 4) load_g((g + 1) & 1, g + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) MmaD<T>::run(acc[i][j], fa[g & 1][i], fb[g & 1][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: straight from the accumulators (igemm_epilogue.h) --------------------------------
  const int col = n0 + wn * 64 + 2 * r;
  float st[4] = {0.f, 0.f, 0.f, 0.f};
  // fast path: whole 32-row MFMA tiles inside the tensor (and, for the pixel shuffle, inside one image row)
  const bool fast = (p.M & 31) == 0 && (p.mode != MODE_SHUFFLE || (p.Wo & 31) == 0);
  if (fast) {
    const unsigned ldob = (unsigned)(p.ldo * ES), ldyb = (unsigned)(p.bn_ldy * ES);
    bool tile_ok[TM];
    unsigned voff[TM], yoff[TM];
    int ccol = col;
    unsigned row_bytes = ldob;
    if (p.mode == MODE_SHUFFLE) {
      const int ab = col / p.shuf_c;
      ccol = col - ab * p.shuf_c;
      row_bytes = 2 * ldob;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int mb = m0 + (wm * TM + mi) * 32;
        tile_ok[mi] = mb < p.M;
        int n, oy, ox;
        decode(mb, n, oy, ox);
        const unsigned pix = (unsigned)((n * 2 * p.Ho + 2 * oy + (ab >> 1)) * (2 * p.Wo) + 2 * ox + (ab & 1));
        voff[mi] = (pix + 8u * h) * ldob + (unsigned)(ccol * ES);
        yoff[mi] = 0;
      }
    } else {
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int mb = m0 + (wm * TM + mi) * 32;
        tile_ok[mi] = mb < p.M;
        voff[mi] = (unsigned)(mb + 4 * h) * ldob + (unsigned)(col * ES);
        yoff[mi] = (unsigned)(mb + 4 * h) * ldyb + (unsigned)(col * ES);
      }
    }
    switch (p.mode) {
      case MODE_STATS: epilogue_tiles<T, MODE_STATS, TM>(p, acc, tile_ok, voff, row_bytes, yoff, ldyb, ccol, st); break;
      case MODE_AFFINE_RELU: epilogue_tiles<T, MODE_AFFINE_RELU, TM>(p, acc, tile_ok, voff, row_bytes, yoff, ldyb, ccol, st); break;
      case MODE_BNBWD: epilogue_tiles<T, MODE_BNBWD, TM>(p, acc, tile_ok, voff, row_bytes, yoff, ldyb, ccol, st); break;
      default: epilogue_tiles<T, MODE_STORE, TM>(p, acc, tile_ok, voff, row_bytes, yoff, ldyb, ccol, st); break;
    }
  } else {
    // ragged shapes (M not a multiple of 32, or a pixel shuffle whose rows are not multiples of 32): per-row checks
    T* __restrict__ og = reinterpret_cast<T*>(p.out);
    const T* __restrict__ yg = reinterpret_cast<const T*>(p.bn_y);
    float k0a = 0.f, k0b = 0.f, k1a = 0.f, k1b = 0.f, mua = 0.f, mub = 0.f, rsa = 0.f, rsb = 0.f;
    int shuf_ab = 0, shuf_co = col;
    if (p.mode == MODE_AFFINE_RELU || p.mode == MODE_BNBWD) {
      k0a = p.scale[col]; k0b = p.scale[col + 1];
      k1a = p.shift[col]; k1b = p.shift[col + 1];
    } else if (p.mode == MODE_SHUFFLE) {
      shuf_ab = col / p.shuf_c;
      shuf_co = col - shuf_ab * p.shuf_c;
      if (p.bias) { k1a = p.bias[shuf_co]; k1b = p.bias[shuf_co + 1]; }
    } else if (p.bias) {
      k1a = p.bias[col]; k1b = p.bias[col + 1];
    }
    if (p.mode == MODE_BNBWD) {
      mua = p.bn_mean[col]; mub = p.bn_mean[col + 1];
      rsa = p.bn_rstd[col]; rsb = p.bn_rstd[col + 1];
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      const int mb = m0 + (wm * TM + mi) * 32;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = mb + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v0 = acc[mi][0][reg], v1 = acc[mi][1][reg];
        if (p.mode == MODE_AFFINE_RELU) {
          v0 = fmaxf(fmaf(v0, k0a, k1a), 0.f);
          v1 = fmaxf(fmaf(v1, k0b, k1b), 0.f);
          store_pair_d<T>(og + (long)m * p.ldo + col, v0, v1);
        } else if (p.mode == MODE_SHUFFLE) {
          v0 += k1a; v1 += k1b;
          int n, oy, ox;
          decode(m, n, oy, ox);
          const long dst = ((long)(n * 2 * p.Ho + 2 * oy + (shuf_ab >> 1)) * (2 * p.Wo) + 2 * ox + (shuf_ab & 1));
          store_pair_d<T>(og + dst * p.ldo + shuf_co, v0, v1);
        } else if (p.mode == MODE_BNBWD) {
          store_pair_d<T>(og + (long)m * p.ldo + col, v0, v1);
          float y0, y1;
          PairRaw<T>::unpack(PairRaw<T>::load(yg + (long)m * p.bn_ldy + col), y0, y1);
          const float g0 = fmaf(y0, k0a, k1a) > 0.f ? round_through<T>(v0) : 0.f;
          const float g1 = fmaf(y1, k0b, k1b) > 0.f ? round_through<T>(v1) : 0.f;
          st[0] += g0; st[1] = fmaf(g0, (y0 - mua) * rsa, st[1]);
          st[2] += g1; st[3] = fmaf(g1, (y1 - mub) * rsb, st[3]);
        } else {
          v0 += k1a; v1 += k1b;
          store_pair_d<T>(og + (long)m * p.ldo + col, v0, v1);
          if (p.mode == MODE_STATS) {
            const float t0 = round_through<T>(v0), t1 = round_through<T>(v1);
            st[0] += t0; st[1] = fmaf(t0, t0, st[1]);
            st[2] += t1; st[3] = fmaf(t1, t1, st[3]);
          }
        }
      }
    }
  }
  if (p.mode == MODE_STATS || p.mode == MODE_BNBWD) write_stat_rows<WM, WN>(p, smem, st, mblk, n0, tid, wave, r, h);
#endif  // __HIP_DEVICE_COMPILE__
}

// ------------------------------------------------------------------------------------------------
template <typename T, int WM, int WN, int TM>
static int launch_dma_cfg(IgemmParams& p, hipStream_t stream) {
  constexpr int BM = WM * TM * 32, BN = WN * 64;
  constexpr int LDS = 2 * (BM + BN) * 128;
  if (const int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&igemm_dma_kernel<T, WM, WN, TM>), LDS, "igemm_dma_kernel")) return rc_;
  p.mblocks = ceil_div(p.M, BM);
  p.nblocks = p.Cout / BN;
  const long nwg = (long)p.mblocks * p.nblocks;
  hipLaunchKernelGGL((igemm_dma_kernel<T, WM, WN, TM>), dim3((unsigned)nwg), dim3(WM * WN * 64), LDS, stream, p);
  char nm[96];
  snprintf(nm, sizeof(nm), "igemm_dma_kernel<%s, %d, %d, %d>", sizeof(T) == 2 ? "__bf16" : "float", WM, WN, TM);
  note_kernel(nm);
  return check_launch("igemm_dma_kernel");
}

// rows of BatchNorm partial statistics the chosen configuration produces (all use BM = 256)
int igemm_dma_mblocks(long M) { return ceil_div(M, 256); }

bool igemm_dma_supported(const IgemmParams& p, int dtype) {
  const long es = dtype == UNETDC_BF16 ? 2 : 4;
  const long HoWo = (long)p.Ho * p.Wo;
  const long xbytes = (p.M / HoWo) * p.Hi * p.Wi * p.ldx * es;
  const long wbytes = (long)p.ntaps * p.Cout * p.Cin * es;
  const long opix = p.mode == MODE_SHUFFLE ? 4L * p.M : (long)p.M;
  const long obytes = opix * p.ldo * es, ybytes = p.mode == MODE_BNBWD ? (long)p.M * p.bn_ldy * es : 0;
  return xbytes < (1L << 31) && wbytes < (1L << 31) && obytes < (1L << 32) && ybytes < (1L << 32) && p.M % HoWo == 0;
}

int launch_igemm_dma(IgemmParams& p, int dtype, hipStream_t stream) {
  static int force = -1;                        // UNETDC_DMA_CFG=A|B|C forces a tile configuration (experiments)
  if (force < 0) {
    const char* e = getenv("UNETDC_DMA_CFG");
    force = e ? (e[0] == 'A' ? 1 : (e[0] == 'B' ? 2 : (e[0] == 'C' ? 3 : 0))) : 0;
  }
  // 256x256 tiles only when they still give every CU a workgroup; small maps (bottleneck: 8192 pixels)
  // take 256x128 so that the grid covers the chip
  const long blocks_a = (long)((p.M + 255) / 256) * (p.Cout / 256);
  const bool use_a = (force == 1) || (force == 0 && p.Cout % 256 == 0 && p.M >= 256 * 64 && blocks_a >= 200);
  const bool use_b = (force == 2) || (force == 0 && !use_a && p.Cout % 128 == 0);
  if (igemm_dma16_supported(p, dtype)) {               // bf16: 16x16x32 MFMA shape (igemm_dma16.hip)
    int cfg = (use_a && p.Cout % 256 == 0) ? 1 : ((use_b && p.Cout % 128 == 0) ? 2 : 3);
    if (force == 0) {
      // two measured refinements of the rule above (MI355X, bs 8):
      //  * ConvTranspose2d forward with N = 4 * Cout >= 2048 on a small map (upconv4: 8192 pixels): 256 workgroups of
      //    256 x 256 cover the chip and stage 25 % fewer bytes per MFMA: 46 -> 39 us
      //  * 256 x 128 tiles that leave half of the CUs without a workgroup (bottleneck.0 dgrad: 128 workgroups) while
      //    256 x 64 tiles fill the chip: 101 -> 85 us
      const long blocks_b = (long)((p.M + 255) / 256) * (p.Cout / 128);
      if (cfg == 2 && p.mode == MODE_SHUFFLE && p.Cout % 256 == 0 && blocks_a >= 200) cfg = 1;
      else if (cfg == 2 && blocks_b >= 100 && blocks_b < 200) cfg = 3;
    }
    return launch_igemm_dma16(p, cfg, stream);
  }
  if (use_a && p.Cout % 256 == 0) {
    return dtype == UNETDC_BF16 ? launch_dma_cfg<bf16_t, 2, 4, 4>(p, stream) : launch_dma_cfg<float, 2, 4, 4>(p, stream);
  }
  if (use_b && p.Cout % 128 == 0) {
    return dtype == UNETDC_BF16 ? launch_dma_cfg<bf16_t, 4, 2, 2>(p, stream) : launch_dma_cfg<float, 4, 2, 2>(p, stream);
  }
  return dtype == UNETDC_BF16 ? launch_dma_cfg<bf16_t, 4, 1, 2>(p, stream) : launch_dma_cfg<float, 4, 1, 2>(p, stream);
}

}  // namespace unetdc
