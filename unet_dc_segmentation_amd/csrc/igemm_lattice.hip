// Lattice-halo implicit GEMM (bf16): dilated 3x3 convolution forward / dgrad, persistent workgroups.
//
//   out[n, Y, X, co] = sum_{ky,kx,ci} x[n, Y + (ky-1) d, X + (kx-1) d, ci] * w[ky*3+kx][co][ci]     (models/model_2.py:41-51)
//
// What round 1's kernels left on the table (SQ counters, profiles/r02_sq_baseline.json):
//   * the per-tap LDS-DMA kernel (igemm_dma16) stages 64 KB per 256x256x64 step, i.e. ~65 GB/s per CU at full MFMA rate --
//     the rate at which a CU can pull bytes from L2 into LDS at all (MI355X_MICROARCH.md, gather into LDS: 66-73 GB/s/CU);
//   * the halo-patch kernel (igemm_halo16) staged the input once per 9 taps but spent 5 VALU instructions per MFMA on
//     fragment addresses (VALU-bound), lost 24 % of its LDS cycles to bank conflicts of the x-shifted taps, drained the VM
//     counter at every tap and paid a full prologue + epilogue per 256-pixel tile (MFMA pipe 35 % busy).
// This kernel keeps the halo patch and removes those costs:
//   * DILATION LIVES IN THE GLOBAL ADDRESSES ONLY.  A d-dilated 3x3 conv is d*d independent dense 3x3 convs on the
//     sub-lattices x[py::d, px::d]; a workgroup owns an 8 x 32 tile of ONE sub-lattice and gathers its (8+2) x (32+2)
//     patch with stride d (LDS-DMA takes a per-lane source address; every pixel is a whole 128-byte line).  In LDS every
//     layer is a dense conv with halo 1: tap (ky, kx) reads patch rows shifted by ky, kx.
//   * NO ADDRESS ARITHMETIC IN THE TAP LOOP: the 9 taps are unrolled; a fragment read is `ds_read_b128 v, vbase offset:imm`
//     with 6 + 2 per-lane bases computed once; everything tap-dependent on the global side is a scalar offset (soffset).
//   * CONFLICT-FREE SHIFTED READS: the 16 rows of an MFMA tile are consecutive patch pixels starting at ANY column, and
//     ds_read_b128 serves lanes {0-3,12-15 | rb} and {4-11 | rb^1} in one pass; lanes 0-3,12-15 read the even pixels and
//     lanes 4-11 the odd ones (a permutation of the tile's rows, undone by the store addresses), the XOR key is a function
//     of the patch column: the two lane sets always sit in different 128-byte halves of the bank row, whatever the shift.
//   * COUNTED WAITS: weights travel through a 3-stage ring two taps ahead, the next patch (next K chunk or next tile) is
//     fetched in slices during the first taps; DMAs are issued from inline asm (lds_dma.h) so that neither
//     __syncthreads() nor the compiler's LDS-alias rule drains the VM counter; one raw s_barrier per tap.
//   * PERSISTENT: a workgroup walks a list of (tile, n-block) items; the pipeline runs across item boundaries (the next
//     item's patch and first weights are in flight during the epilogue stores of the current one).
//
// Tiles: 256 pixels x BN channels, waves WM x WN, a wave owns MT M-tiles of 16 pixels x 64 channels
//   BN = 64 : 4 waves (4 x 1), MT = 4, ONE patch buffer (2 workgroups per CU hide the reload)
//   BN = 128: 8 waves (4 x 2), MT = 4, two patch buffers
// LDS: [NPB patch buffers][3 weight stages of BN x 128 B][statistics scratch].
#include <stdio.h>
#include <stdlib.h>

#include "igemm_epilogue16.h"
#include "kernels.h"
#include "lds_dma.h"

namespace unetdc {

constexpr int LTH = 8, LTW = 32;                  // lattice tile
constexpr int LPH = LTH + 2, LPW = LTW + 2;       // patch (halo 1)
constexpr int LPP = LPH * LPW;                    // 340 patch pixels
constexpr int LPI = (LPP + 7) / 8;                // 43 DMA wave-instructions (8 pixel rows of 128 B each)
constexpr unsigned LOOB = 0x80000000u;

struct LatticeParams {
  int d;                      // dilation = lattice stride
  int Hs, Ws;                 // lattice size H / d, W / d
  int tiles_x, tiles_y;       // Ws / 32, Hs / 8
  int tiles_per_img;          // d * d * tiles_x * tiles_y
  int nblocks;                // Cout / BN
  int items;                  // N * tiles_per_img * nblocks
  int nkc;                    // Cin / 64
  unsigned mg_nblocks, mg_tpi, mg_tpp, mg_d, mg_tx;   // ceil(2^32 / divisor) for the item decode (exact for n * divisor < 2^32)
  int stat_rows;              // statistics rows with data = gridDim.x / nblocks (one per workgroup and n-block), 0: per-tile rows
  int mtiles;                 // N * tiles_per_img = M / 256: rows [stat_rows, mtiles) are written as zeros
};
// s_setprio 1 through the tap loops, 0 in the epilogues (settled in round 3, profiles/r03_setprio_ab.txt).
// The round-2/3 timing probes that removed barriers / waits / DMAs (UNETDC_LAT_DBG: invalid results) are gone from the library;
// tools/probes/ keeps the notes.

#if defined(__HIP_DEVICE_COMPILE__)
// patch slices issued at tap t: PJ slices in groups of SPT over the first taps
template <int PJ, int SPT> __device__ constexpr int lat_nsl(int t) {
  return t < 0 ? 0 : ((PJ - t * SPT) <= 0 ? 0 : ((PJ - t * SPT) < SPT ? (PJ - t * SPT) : SPT));
}
#endif

// INORM (the 4-wave single-buffer form): the input is the RAW conv output of the stage in front; that stage's BatchNorm + ReLU
// is applied ONCE per staged patch, in LDS, between the barrier behind which the patch has landed and the first tap -- the
// stand-alone normalisation pass of that stage and its activation tensor disappear (models/model_2.py:45-46 fused into :48).
// A thread owns one logical 16-byte chunk (8 channels: its 16 constants live in registers) of every 32nd patch pixel; padding
// pixels (outside the sub-lattice) were zero-filled by the DMA and stay zero, which is what zero padding of the ACTIVATION means.
// The result is rounded through bf16 like a stored activation: outputs are bit-identical to the two-pass form.
template <int WM, int WN, int MT, int NPB, int MODE, bool INORM = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void igemm_lattice_kernel(const IgemmParams p, const LatticeParams q) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(!INORM || (NPB == 1 && WM * WN == 4), "input normalisation: the 4-wave single-buffer form");
  constexpr int NW = WM * WN, BN = WN * 64;
  constexpr int BI = BN / 8 / NW;                 // weight DMA instructions per wave per tap
  constexpr int PJ = (LPI + NW - 1) / NW;         // patch DMA instructions per wave per chunk (uniform: padded)
  constexpr int SPT = (NW == 4) ? 3 : 2;          // patch slices per tap while prefetching
  constexpr int PBUF = PJ * NW * 1024;            // bytes per patch buffer
  constexpr int WST = BN * 128;                   // bytes per weight stage
  constexpr int OFF_W = NPB * PBUF, OFF_RED = OFF_W + 3 * WST;
  constexpr int NST = MT * 4;                     // epilogue stores per wave and tile
  static_assert(WM * MT == 16 && (MT % 2) == 0, "a workgroup owns 16 M-tiles = 8 rows x 32 pixels");
  static_assert(BN % (8 * NW) == 0, "weight rows / wave mismatch");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const unsigned lds_base = lds_addr_of(smem);
  const int G = gridDim.x;
  const int d = q.d;

  // The patch origin (-1, -1) of a border tile lies in front of the tensor; the scalar part of a DMA address (soffset)
  // cannot be negative, so the descriptor starts SH bytes early (those bytes are never touched: halo lanes outside the
  // image carry an out-of-range voffset and read zeros).
  const unsigned SH = (unsigned)((d * p.Wi + d) * p.ldx * 2);
  const unsigned xbytes = (unsigned)((long)(p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.ldx * 2) + SH;
  const unsigned wbytes = (unsigned)((long)9 * p.Cout * p.Cin * 2);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.x)) - SH, 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, wbytes, 0x00020000);

  // ---- per-lane constants --------------------------------------------------------------------------------------
  const int sub = lane >> 3, pc = lane & 7;
  // patch slice sl of this wave = DMA instruction wave + NW*sl = patch pixels 8*instr .. 8*instr+7 (row-major in the
  // 10 x 34 patch); this lane feeds pixel pr, physical chunk pc <- logical chunk pc ^ key(column)
  unsigned prel[PJ];                              // byte offset of the lane's 16-byte piece relative to the patch origin
  unsigned pflg[(PJ + 3) / 4];                    // 8 flag bits per slice: 1 top halo row, 2 bottom, 4 left halo column, 8 right,
                                                  //                       16 padding row of the buffer, 32 always
#pragma unroll
  for (int w4 = 0; w4 < (PJ + 3) / 4; ++w4) pflg[w4] = 0;
#pragma unroll
  for (int sl = 0; sl < PJ; ++sl) {
    const int pr = (wave + NW * sl) * 8 + sub;
    const int ppy = (pr * 1928) >> 16, ppx = pr - ppy * LPW;       // pr / 34 for pr < 1024
    const int ch = pc ^ ((ppx >> 1) & 7);
    prel[sl] = (unsigned)(((ppy * d) * p.Wi + ppx * d) * p.ldx * 2 + ch * 16);
    const unsigned f = (ppy == 0 ? 1u : 0u) | (ppy == LPH - 1 ? 2u : 0u) | (ppx == 0 ? 4u : 0u) | (ppx == LPW - 1 ? 8u : 0u) |
                       (pr >= LPP ? 16u : 0u) | 32u;
    pflg[sl >> 2] |= f << (8 * (sl & 3));
  }
  // weight rows: LDS row lrow of the stage = N tile (q >> 4) of its 64-channel group, column q & 15
  //   <-> output channel 4*(q & 15) + (q >> 4): the four N tiles of a lane hold four consecutive channels
  unsigned bbase[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int lrow = (wave + NW * j) * 8 + sub;
    const int grp = lrow >> 6, qq = lrow & 63;
    const int cc = (qq & 15) * 4 + (qq >> 4);
    const int c = pc ^ ((lrow >> 1) & 7);
    bbase[j] = (unsigned)((grp * 64 + cc) * p.Cin * 2 + c * 16);
  }
  // fragment read bases.  MFMA row m of a tile (supplied by lanes with c16 = m) is pixel PI(m) of the 16:
  // lanes 0-3,12-15 -> even pixels, lanes 4-11 -> odd pixels
  const int c16 = lane & 15, rb = lane >> 4;
  const int pi = (c16 < 4) ? 2 * c16 : (c16 < 12 ? 2 * (c16 - 4) + 1 : 2 * (c16 - 8));
  int aoff[3][2], boff[2];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int px = kx + pi;                     // patch column (mod 16: +16 for the right half keeps the key)
      aoff[kx][g] = (wm * (MT / 2) * LPW + px) * 128 + (((4 * g + rb) ^ ((px >> 1) & 7)) << 4);
    }
#pragma unroll
  for (int g = 0; g < 2; ++g) boff[g] = OFF_W + (wn * 64 + c16) * 128 + (((4 * g + rb) ^ ((c16 >> 1) & 7)) << 4);

  // ---- work items ---------------------------------------------------------------------------------------------------
  struct Item { int img, phy, phx, ly0, lx0, nblk, mtile; };
  // v_mul_hi_u32 is a VALU instruction: readfirstlane brings the (wave-uniform) quotient back to an SGPR, which is what
  // the scalar operands of the DMA statements need
  auto udiv = [](unsigned n, unsigned magic, unsigned dv) {
    return dv == 1 ? n : (unsigned)__builtin_amdgcn_readfirstlane((int)__umulhi(n, magic));
  };
  auto decode = [&](int item) {
    Item it;
    it.mtile = (int)udiv((unsigned)item, q.mg_nblocks, (unsigned)q.nblocks);
    it.nblk = item - it.mtile * q.nblocks;
    it.img = (int)udiv((unsigned)it.mtile, q.mg_tpi, (unsigned)q.tiles_per_img);
    int r = it.mtile - it.img * q.tiles_per_img;
    const int tpp = q.tiles_x * q.tiles_y;        // tiles per phase
    const int ph = (int)udiv((unsigned)r, q.mg_tpp, (unsigned)tpp);
    r -= ph * tpp;
    it.phy = (int)udiv((unsigned)ph, q.mg_d, (unsigned)d);
    it.phx = ph - it.phy * d;
    const int ty = (int)udiv((unsigned)r, q.mg_tx, (unsigned)q.tiles_x);
    it.ly0 = ty * LTH;
    it.lx0 = (r - ty * q.tiles_x) * LTW;
    return it;
  };
  const int first = xcd_remap(blockIdx.x, G);
  if (first >= q.items) return;                   // (grid <= items: never taken)

  // ---- DMA issue -----------------------------------------------------------------------------------------------------
  // scalar description of the patch of (item, K chunk): byte offset of its origin (+SH) and which borders it touches
  auto patch_base = [&](const Item& it, int kc) {
    return (unsigned)((((it.img * p.Hi + (it.ly0 - 1) * d + it.phy) * p.Wi + (it.lx0 - 1) * d + it.phx) * p.ldx) * 2 + kc * 128) + SH;
  };
  auto patch_edges = [&](const Item& it, bool valid) {
    return valid ? ((it.ly0 == 0 ? 1u : 0u) | (it.ly0 + LTH == q.Hs ? 2u : 0u) | (it.lx0 == 0 ? 4u : 0u) |
                    (it.lx0 + LTW == q.Ws ? 8u : 0u) | 16u)
                 : 32u;
  };
  auto issue_slice = [&](int sl, int buf, unsigned pbase, unsigned edges) {
    const int pr = (wave + NW * sl) * 8 + sub;
    const int ppy = (pr * 1928) >> 16, ppx = pr - ppy * LPW;       // pr / 34 for pr < 1024
    const int ch = pc ^ ((ppx >> 1) & 7);
    const unsigned rel = (unsigned)(((ppy * d) * p.Wi + ppx * d) * p.ldx * 2 + ch * 16);
    const unsigned f = ((ppy == 0 ? 1u : 0u) | (ppy == LPH - 1 ? 2u : 0u) | (ppx == 0 ? 4u : 0u) | (ppx == LPW - 1 ? 8u : 0u) |
                        (pr >= LPP ? 16u : 0u) | 32u) & edges;
    lds_dma16(xr, lds_base + buf * PBUF + (wave + NW * sl) * 1024, f ? LOOB : rel, pbase);
  };
  auto issue_w = [&](int stage, int tap, int kc, int nblk, bool valid) {
    const unsigned soff = (unsigned)(((tap * p.Cout + nblk * BN) * p.Cin + kc * 64) * 2);
#pragma unroll
    for (int j = 0; j < BI; ++j)
      lds_dma16(wr, lds_base + OFF_W + stage * WST + (wave + NW * j) * 1024, valid ? bbase[j] : LOOB, soff);
  };

  // INORM: the producing stage's (scale, shift) pairs sit in LDS behind the statistics scratch, [2][Cin] floats, written once:
  // a vector-memory load inside the item loop would make the compiler drain the VM queue (the previous item's epilogue
  // stores) in front of every patch -- that drain, not the arithmetic, made the first form of this pass cost 3.4 us per item
  float* const ncst = reinterpret_cast<float*>(smem + OFF_RED + NW * 512);
  if (INORM) {
    for (int i = tid; i < 2 * p.Cin; i += NW * 64) ncst[i] = (i < p.Cin ? p.in_scale[i] : p.in_shift[i - p.Cin]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the first barrier of the pipeline publishes it)
  }
  // INORM: normalise the patch of (item, K chunk kc) in place
  auto normalise_patch = [&](const Item& it, int kc) {
    const int lc = tid & 7;                       // logical chunk: channels kc * 64 + 8 lc .. + 7
    float nsc[8], nsh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { nsc[e] = ncst[kc * 64 + lc * 8 + e]; nsh[e] = ncst[p.Cin + kc * 64 + lc * 8 + e]; }
#pragma unroll
    for (int i = 0; i < (LPP + 31) / 32; ++i) {
      const int pr = (tid >> 3) + 32 * i;
      const int ppy = (pr * 1928) >> 16, ppx = pr - ppy * LPW;       // pr / 34
      const int ly = it.ly0 + ppy - 1, lx = it.lx0 + ppx - 1;
      if (pr < LPP && (unsigned)ly < (unsigned)q.Hs && (unsigned)lx < (unsigned)q.Ws) {
        unsigned char* a = smem + pr * 128 + ((lc ^ ((ppx >> 1) & 7)) << 4);
        float v[8];
        Chunk<bf16_t>::unpack(ld16(a), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], nsc[e], nsh[e]), 0.f);
        st16(a, Chunk<bf16_t>::pack(v));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    raw_barrier();
  };

  f32x4 acc[MT][4];
  float binit[4] = {0.f, 0.f, 0.f, 0.f};          // conv bias of this lane's four channels: the accumulators START from it
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = binit[j];
  };

  // one tap: fragments of both 32-channel halves, MT*4*2 MFMAs
  // (the 4-wave BatchNorm-backward variant also carries the prefetched saved outputs: there the two halves take turns
  //  in ONE fragment register set, everywhere else both halves are read up front)
  constexpr bool SPLIT_FRAGS = (MODE == MODE_BNBWD && NW == 4);
  u32x4 fa_live[2][MT];
  // `issue`: this tap's vector-memory statements (weights two taps ahead, patch slices, saved-output prefetch), placed BEHIND
  // the tap's first fragment reads so that those are in flight while the wave gets its DMA instructions accepted
  auto compute_tap = [&](int ky, int kx, int stage, const int (&ab)[3][2], auto&& issue) {
    if (SPLIT_FRAGS) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        u32x4 fa[MT], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = ld16(smem + boff[g] + stage * WST + j * 16 * 128);
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = ld16(smem + ab[kx][g] + ((i >> 1) + ky) * (LPW * 128) + (i & 1) * (16 * 128));
        if (g == 0) issue();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]),
                                                                acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    // A FRAGMENTS LIVE ACROSS TAPS (round 4): fa[g][i] is re-loaded right behind the four MFMAs that read it with the
    // fragment the NEXT tap needs there -- the patch is resident for the whole chunk, so those reads cross the tap barrier
    // and only the eight B fragments of a tap (its weights are only known to have landed behind the barrier) are read in
    // front of the MFMAs; the first tap of a chunk loads its A fragments itself (the patch has just landed / been normalised).
    u32x4 fb[2][4];
    const int t = ky * 3 + kx, nky = (t + 1) / 3, nkx = (t + 1) % 3;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[g][j] = ld16(smem + boff[g] + stage * WST + j * 16 * 128);
      if (t == 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          fa_live[g][i] = ld16(smem + ab[kx][g] + ((i >> 1) + ky) * (LPW * 128) + (i & 1) * (16 * 128));
      }
    }
    issue();
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa_live[g][i]),
                                                              __builtin_bit_cast(bf16x8, fb[g][j]), acc[i][j], 0, 0, 0);
        if (t < 8) fa_live[g][i] = ld16(smem + ab[nkx][g] + ((i >> 1) + nky) * (LPW * 128) + (i & 1) * (16 * 128));
        __builtin_amdgcn_sched_barrier(0);
      }
  };

  // ---- epilogue of one item -----------------------------------------------------------------------------------------------
  const unsigned ldob = (unsigned)(p.ldo * 2), ldyb = (unsigned)(p.bn_ldy * 2);
  const int xb = (rb == 0) ? 0 : (rb == 1 ? 1 : (rb == 2 ? 9 : 8));          // pixel of accumulator row 4*rb + v = xb + 2v
  // MODE_BNBWD: the consumer stage's saved conv outputs of this item, fetched YT taps before the item's last MFMA.  They
  // sit in the VM queue between the weight DMAs: the counted waits of the two taps after the fetch leave them in flight
  // (+NY), from the third tap on an in-order wait would require them -- so the fetch goes at tap 6 of the last chunk.
  constexpr int YT = 6, NY = (MODE == MODE_BNBWD) ? MT * 4 : 0;
  u32x2 ypre[MT][4];
  auto item_offsets = [&](const Item& it, unsigned (&voff)[MT], unsigned (&yoff)[MT]) {
    const int col = it.nblk * BN + wn * 64 + 4 * c16;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int gm = wm * MT + i;                                            // M tile of the workgroup: row gm >> 1, half gm & 1
      const int Y = (it.ly0 + (gm >> 1)) * d + it.phy, X = (it.lx0 + 16 * (gm & 1) + xb) * d + it.phx;
      const unsigned pix = (unsigned)((it.img * p.Ho + Y) * p.Wo + X);
      voff[i] = pix * ldob + (unsigned)(col * 2);
      yoff[i] = pix * ldyb + (unsigned)(col * 2);
    }
  };
  Epi16Consts ec;
  auto load_consts = [&](int nblk) {
    ec = epi16_consts<MODE>(p, nblk * BN + wn * 64 + 4 * c16);
    if (MODE == MODE_STORE || MODE == MODE_STATS) {        // "+ bias" modes: fold it into the accumulator init
#pragma unroll
      for (int k = 0; k < 4; ++k) { binit[k] = ec.k1[k]; ec.k1[k] = 0.f; }
    }
  };
  float tot_su = 0.f, tot_sq = 0.f;               // lanes tid < BN: running statistics of channel nblk * BN + tid
  auto epilogue = [&](const Item& it) {
    __builtin_amdgcn_s_setprio(0);
    bool tile_ok[MT];
    unsigned voff[MT], yoff[MT];
    item_offsets(it, voff, yoff);
#pragma unroll
    for (int i = 0; i < MT; ++i) tile_ok[i] = true;
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned rbytes = 2u * (unsigned)d * ldob, yrbytes = 2u * (unsigned)d * ldyb;   // accumulator rows are 2 lattice pixels apart
    epilogue16c<MODE, MT, true>(p, acc, tile_ok, voff, rbytes, yoff, yrbytes, ec, s4, q4, ypre);
    if (MODE == MODE_STATS || MODE == MODE_BNBWD) {
      // The per-tile sums are added up per WORKGROUP (gridDim.x is a multiple of nblocks, so a workgroup keeps its n-block
      // for all its tiles): one partial row per workgroup instead of one per tile -- at 8 x 512 x 512 that is 256-512 rows
      // instead of 8192, few enough for the BatchNorm finalisers to read directly (no colsum_stage launch in between).
      // Scratch of its own: the stage and patch buffers carry the next item's prefetch; the >= 9 tap barriers between two
      // tiles order the reuse.
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s4[k] += __shfl_xor(s4[k], 16, 64); q4[k] += __shfl_xor(q4[k], 16, 64);
        s4[k] += __shfl_xor(s4[k], 32, 64); q4[k] += __shfl_xor(q4[k], 32, 64);
      }
      float* red = reinterpret_cast<float*>(smem + OFF_RED);                 // [wave][4 k][2][16 c]
      if (rb == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          red[((wave * 4 + k) * 2 + 0) * 16 + c16] = s4[k];
          red[((wave * 4 + k) * 2 + 1) * 16 + c16] = q4[k];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      raw_barrier();
      if (tid < BN) {
        const int wn2 = tid >> 6, cc = tid & 63, c2 = cc >> 2, k = cc & 3;
        float su = 0.f, sq = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < WM; ++w2) {
          su += red[(((w2 * WN + wn2) * 4 + k) * 2 + 0) * 16 + c2];
          sq += red[(((w2 * WN + wn2) * 4 + k) * 2 + 1) * 16 + c2];
        }
        tot_su += su;                                       // tiles in the order this workgroup visits them: reproducible
        tot_sq += sq;
      }
    }
    __builtin_amdgcn_s_setprio(1);
    zero_acc();
  };

  // ---- the pipeline ----------------------------------------------------------------------------------------------------
  // flat step q = 9*chunk + tap over all (item, K chunk) pairs of this workgroup; weights of step q live in ring stage
  // tap % 3 and are issued at step q - 2; the patch of chunk c lives in buffer c % NPB.
  __builtin_amdgcn_s_setprio(1);
  Item cur = decode(first), nxt = cur;
  int item = first;
  load_consts(cur.nblk);
  zero_acc();
  {
    if (NPB == 1) { issue_w(0, 0, 0, cur.nblk, true); issue_w(1, 1, 0, cur.nblk, true); }
#pragma unroll
    for (int sl = 0; sl < PJ; ++sl) issue_slice(sl, 0, patch_base(cur, 0), patch_edges(cur, true));
    if (NPB == 2) { issue_w(0, 0, 0, cur.nblk, true); issue_w(1, 1, 0, cur.nblk, true); }
  }
  int cbuf = 0;
  bool boundary = false;                          // the previous chunk ended an item: its NST stores are younger than the prefetch
  for (;;) {
    for (int kc = 0; kc < q.nkc; ++kc) {
      const bool last_kc = kc + 1 == q.nkc;
      const int item_n = last_kc ? item + G : item, kc_n = last_kc ? 0 : kc + 1;
      const bool have_n = item_n < q.items;
      if (last_kc) nxt = have_n ? decode(item_n) : cur;
      // the patch the next chunk needs: same item, next 64 channels -- or the next item's first chunk
      // (readfirstlane: a select between two uniform values can come out of the compiler as a VGPR phi, which the
      //  scalar operand of a DMA statement cannot take; the DMAs that read it are a barrier and a wait further down)
      const unsigned pb_n = (unsigned)__builtin_amdgcn_readfirstlane((int)(last_kc ? patch_base(nxt, 0) : patch_base(cur, kc_n)));
      const unsigned pe_n = (unsigned)__builtin_amdgcn_readfirstlane((int)(last_kc ? patch_edges(nxt, have_n) : patch_edges(cur, true)));
      const int nblk_n = __builtin_amdgcn_readfirstlane(nxt.nblk);
      int ab[3][2];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int g = 0; g < 2; ++g) ab[kx][g] = aoff[kx][g] + (NPB == 2 ? cbuf * PBUF : 0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        // ---- wait for the weights of this step (and, at t = 0, the patch of this chunk) -------------------------------
        if (NPB == 2) {
          constexpr int nA = lat_nsl<PJ, SPT>(0);
          if (t == 0) { if (boundary) wait_vmcnt<BI + NST>(); else wait_vmcnt<BI>(); }
          else if (t == 1) { if (boundary) wait_vmcnt<BI + nA + NST>(); else wait_vmcnt<BI + nA>(); }
          else if (t == 2) wait_vmcnt<lat_nsl<PJ, SPT>(0) + BI + lat_nsl<PJ, SPT>(1)>();
          else if (t == 3) wait_vmcnt<lat_nsl<PJ, SPT>(1) + BI + lat_nsl<PJ, SPT>(2)>();
          else if (t == 4) wait_vmcnt<lat_nsl<PJ, SPT>(2) + BI + lat_nsl<PJ, SPT>(3)>();
          else if (t == 5) wait_vmcnt<lat_nsl<PJ, SPT>(3) + BI + lat_nsl<PJ, SPT>(4)>();
          else if (t == 6) wait_vmcnt<lat_nsl<PJ, SPT>(4) + BI + lat_nsl<PJ, SPT>(5)>();
          else if (t == 7) { if (NY && last_kc) wait_vmcnt<lat_nsl<PJ, SPT>(5) + BI + lat_nsl<PJ, SPT>(6) + NY>(); else wait_vmcnt<lat_nsl<PJ, SPT>(5) + BI + lat_nsl<PJ, SPT>(6)>(); }
          else { if (NY && last_kc) wait_vmcnt<lat_nsl<PJ, SPT>(6) + BI + lat_nsl<PJ, SPT>(7) + NY>(); else wait_vmcnt<lat_nsl<PJ, SPT>(6) + BI + lat_nsl<PJ, SPT>(7)>(); }
        } else {
          // single patch buffer: the slices of this chunk were issued after the previous chunk's last tap, behind
          // W(q) and W(q+1) and in front of the epilogue stores
          if (t == 0) { if (boundary) wait_vmcnt<NST>(); else wait_vmcnt<0>(); }
          else if (t > YT && NY) { if (last_kc) wait_vmcnt<BI + NY>(); else wait_vmcnt<BI>(); }
          else wait_vmcnt<BI>();
        }
        raw_barrier();
        if (INORM && t == 0) normalise_patch(cur, kc);       // the patch of this chunk has landed (every wave's pieces)
        // ---- prefetch: weights two steps ahead, patch slices of the next chunk -----------------------------------------
        auto issue = [&]() {
          if (t < 7) issue_w((t + 2) % 3, t + 2, kc, cur.nblk, true);
          else issue_w((t + 2) % 3, t + 2 - 9, kc_n, nblk_n, have_n);
          if (NPB == 2) {
#pragma unroll
            for (int u = 0; u < SPT; ++u)
              if (t * SPT + u < PJ) issue_slice(t * SPT + u, cbuf ^ 1, pb_n, pe_n);
          }
          if (MODE == MODE_BNBWD && t == YT && last_kc) {     // after this tap's DMAs: the y fetch is younger than W(q + 2)
            unsigned voff[MT], yoff[MT];
            item_offsets(cur, voff, yoff);
            epi16_prefetch_y<MT>(p, yoff, 2u * (unsigned)d * ldyb, ypre);
          }
        };
        compute_tap(t / 3, t % 3, t % 3, ab, issue);
      }
      if (NPB == 1) {
        raw_barrier();                            // every wave has issued the MFMAs of tap 8: the patch buffer is free
#pragma unroll
        for (int sl = 0; sl < PJ; ++sl) issue_slice(sl, 0, pb_n, pe_n);
      }
      boundary = last_kc;
      if (last_kc) epilogue(cur);
      cbuf ^= 1;
    }
    item += G;
    if (item >= q.items) break;
    if (nxt.nblk != cur.nblk) { load_consts(nxt.nblk); zero_acc(); }
    cur = nxt;
  }
  if ((MODE == MODE_STATS || MODE == MODE_BNBWD) && tid < BN) {
    constexpr int nrow = (MODE == MODE_BNBWD) ? 3 : 2;
    const int nblk0 = first - (int)udiv((unsigned)first, q.mg_nblocks, (unsigned)q.nblocks) * q.nblocks;
    const int r0 = (first - nblk0) / q.nblocks;              // this workgroup's row; its n-block never changed
    float* col = p.stats + nblk0 * BN + tid;
    float* row = col + (long)r0 * nrow * p.Cout;
    row[0] = tot_su;
    row[p.Cout] = tot_sq;
    if (nrow == 3) row[2 * p.Cout] = 0.f;
    // rows a per-tile writer would have produced: zeros, so that a reader of all M / 256 rows still gets the right sums
    for (int r = r0 + q.stat_rows; r < q.mtiles; r += q.stat_rows) {
      float* z = col + (long)r * nrow * p.Cout;
      z[0] = 0.f;
      z[p.Cout] = 0.f;
      if (nrow == 3) z[2 * p.Cout] = 0.f;
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// Wide-wave form: 4 waves, each owns 64 pixels x 128 channels (two 64-channel groups: 4 M tiles x 8 N tiles).
// SQ counters of the 64 x 64 wave tile: 0.5 ds_read_b128 per MFMA = 128 B/clk/CU with all four SIMDs busy, half of what the
// LDS array delivers, and the LDS-DMA writes share the array.  Here a wave reads 4 A + 8 B fragments for 32 MFMAs (0.375 per MFMA) and the patch is shared by
// 128 output channels.  To keep TWO workgroups per CU (4 waves each; the single patch buffer leaves the next chunk's
// patch exposed, the other workgroup covers it) the weight ring has two 16 KB stages: 44 KB patch + 32 KB + 4 KB scratch
// = 80 KB.  With two stages the weights of tap t+1 are issued behind the barrier of tap t (one tap = 64 MFMAs per wave
// = the same 1024 cycles of cover as two taps of the 3-stage form); the stage of flat step q is q & 1, and 9 taps per
// chunk flip the parity per chunk (two fragment base sets, selected at compile time inside the unrolled taps).
// Modes: STORE / STATS / AFFINE_RELU (the fused BatchNorm-backward epilogue keeps the 64-wide forms: its saved-output
// prefetch does not fit the register budget next to 128 accumulators).
// INORM (round 5, statistics mode): the input is the RAW conv output of the stage in front; its BatchNorm + ReLU is applied once
// per staged patch in LDS, as in igemm_lattice_kernel<..., INORM> -- but this form has no LDS left for the constants (80 KB
// exactly, two workgroups per CU), so the chunk index is WAVE-UNIFORM here (wave w owns logical chunks 2w, 2w + 1 of every
// pixel; a lane owns pixel lane + 64 i) and the 16 constants of a chunk come in through the scalar cache into SGPRs: no
// vector-memory load joins the hand-counted DMA queue and no LDS is spent.  16 pixels of a wave-instruction cover the 8 XOR
// keys x 2 bank halves exactly once: the 64 16-byte accesses of an instruction spread evenly over the banks.
template <int MODE, bool INORM = false>
__global__ __launch_bounds__(256, 2) void igemm_lattice_wide_kernel(const IgemmParams p, const LatticeParams q) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(!INORM || MODE == MODE_STATS, "input normalisation: the training forward");
  constexpr int WM = 4, WN = 1, MT = 4, NPB = 1, NG = 2;
  constexpr int NW = WM * WN, BN = WN * NG * 64;
  constexpr int BI = BN / 8 / NW;                 // weight DMA instructions per wave per tap
  constexpr int PJ = (LPI + NW - 1) / NW;         // patch DMA instructions per wave per chunk (uniform: padded)
  constexpr int SPT = (NW == 4) ? 3 : 2;          // patch slices per tap while prefetching
  constexpr int PBUF = PJ * NW * 1024;            // bytes per patch buffer
  constexpr int WST = BN * 128;                   // bytes per weight stage
  constexpr int OFF_W = NPB * PBUF, OFF_RED = OFF_W + 2 * WST;
  constexpr int NST = MT * 4 * NG;                // epilogue stores per wave and tile
  static_assert(WM * MT == 16 && (MT % 2) == 0, "a workgroup owns 16 M-tiles = 8 rows x 32 pixels");
  static_assert(BN % (8 * NW) == 0, "weight rows / wave mismatch");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const unsigned lds_base = lds_addr_of(smem);
  const int G = gridDim.x;
  const int d = q.d;

  // The patch origin (-1, -1) of a border tile lies in front of the tensor; the scalar part of a DMA address (soffset)
  // cannot be negative, so the descriptor starts SH bytes early (those bytes are never touched: halo lanes outside the
  // image carry an out-of-range voffset and read zeros).
  const unsigned SH = (unsigned)((d * p.Wi + d) * p.ldx * 2);
  const unsigned xbytes = (unsigned)((long)(p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.ldx * 2) + SH;
  const unsigned wbytes = (unsigned)((long)9 * p.Cout * p.Cin * 2);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.x)) - SH, 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, wbytes, 0x00020000);

  // ---- per-lane constants --------------------------------------------------------------------------------------
  const int sub = lane >> 3, pc = lane & 7;
  // patch slice sl of this wave = DMA instruction wave + NW*sl = patch pixels 8*instr .. 8*instr+7 (row-major in the
  // 10 x 34 patch); this lane feeds pixel pr, physical chunk pc <- logical chunk pc ^ key(column)
  // (this form recomputes the lane's piece offset and border flags per slice when it issues the DMA -- 11 slices per chunk,
  //  a few dozen VALU instructions against 576 MFMAs -- instead of holding 14 registers across the tap loop: with 128
  //  accumulators they were spilled, and a scratch reload drains the VM counter in front of the hand-placed DMAs)
  // weight rows: LDS row lrow of the stage = N tile (q >> 4) of its 64-channel group, column q & 15
  //   <-> output channel 4*(q & 15) + (q >> 4): the four N tiles of a lane hold four consecutive channels
  unsigned bbase[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int lrow = (wave + NW * j) * 8 + sub;
    const int grp = lrow >> 6, qq = lrow & 63;
    const int cc = (qq & 15) * 4 + (qq >> 4);
    const int c = pc ^ ((lrow >> 1) & 7);
    bbase[j] = (unsigned)((grp * 64 + cc) * p.Cin * 2 + c * 16);
  }
  // fragment read bases.  MFMA row m of a tile (supplied by lanes with c16 = m) is pixel PI(m) of the 16:
  // lanes 0-3,12-15 -> even pixels, lanes 4-11 -> odd pixels
  const int c16 = lane & 15, rb = lane >> 4;
  const int pi = (c16 < 4) ? 2 * c16 : (c16 < 12 ? 2 * (c16 - 4) + 1 : 2 * (c16 - 8));
  int aoff[3][2], boff[2];                        // boff: group 0; group 1 = + 64 * 128
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int px = kx + pi;                     // patch column (mod 16: +16 for the right half keeps the key)
      aoff[kx][g] = (wm * (MT / 2) * LPW + px) * 128 + (((4 * g + rb) ^ ((px >> 1) & 7)) << 4);
    }
#pragma unroll
  for (int g = 0; g < 2; ++g) boff[g] = OFF_W + (wn * 64 + c16) * 128 + (((4 * g + rb) ^ ((c16 >> 1) & 7)) << 4);

  // ---- work items ---------------------------------------------------------------------------------------------------
  struct Item { int img, phy, phx, ly0, lx0, nblk, mtile; };
  // v_mul_hi_u32 is a VALU instruction: readfirstlane brings the (wave-uniform) quotient back to an SGPR, which is what
  // the scalar operands of the DMA statements need
  auto udiv = [](unsigned n, unsigned magic, unsigned dv) {
    return dv == 1 ? n : (unsigned)__builtin_amdgcn_readfirstlane((int)__umulhi(n, magic));
  };
  auto decode = [&](int item) {
    Item it;
    it.mtile = (int)udiv((unsigned)item, q.mg_nblocks, (unsigned)q.nblocks);
    it.nblk = item - it.mtile * q.nblocks;
    it.img = (int)udiv((unsigned)it.mtile, q.mg_tpi, (unsigned)q.tiles_per_img);
    int r = it.mtile - it.img * q.tiles_per_img;
    const int tpp = q.tiles_x * q.tiles_y;        // tiles per phase
    const int ph = (int)udiv((unsigned)r, q.mg_tpp, (unsigned)tpp);
    r -= ph * tpp;
    it.phy = (int)udiv((unsigned)ph, q.mg_d, (unsigned)d);
    it.phx = ph - it.phy * d;
    const int ty = (int)udiv((unsigned)r, q.mg_tx, (unsigned)q.tiles_x);
    it.ly0 = ty * LTH;
    it.lx0 = (r - ty * q.tiles_x) * LTW;
    return it;
  };
  const int first = xcd_remap(blockIdx.x, G);
  if (first >= q.items) return;                   // (grid <= items: never taken)

  // ---- DMA issue -----------------------------------------------------------------------------------------------------
  // scalar description of the patch of (item, K chunk): byte offset of its origin (+SH) and which borders it touches
  auto patch_base = [&](const Item& it, int kc) {
    return (unsigned)((((it.img * p.Hi + (it.ly0 - 1) * d + it.phy) * p.Wi + (it.lx0 - 1) * d + it.phx) * p.ldx) * 2 + kc * 128) + SH;
  };
  auto patch_edges = [&](const Item& it, bool valid) {
    return valid ? ((it.ly0 == 0 ? 1u : 0u) | (it.ly0 + LTH == q.Hs ? 2u : 0u) | (it.lx0 == 0 ? 4u : 0u) |
                    (it.lx0 + LTW == q.Ws ? 8u : 0u) | 16u)
                 : 32u;
  };
  auto issue_slice = [&](int sl, int buf, unsigned pbase, unsigned edges) {
    int sub_here = sub;
    asm volatile("" : "+v"(sub_here));              // opaque: keeps the compiler from hoisting (and then spilling) the 11 results
    const int pr = (wave + NW * sl) * 8 + sub_here;
    const int ppy = (pr * 1928) >> 16, ppx = pr - ppy * LPW;       // pr / 34 for pr < 1024
    const int ch = pc ^ ((ppx >> 1) & 7);
    const unsigned rel = (unsigned)(((ppy * d) * p.Wi + ppx * d) * p.ldx * 2 + ch * 16);
    const unsigned f = ((ppy == 0 ? 1u : 0u) | (ppy == LPH - 1 ? 2u : 0u) | (ppx == 0 ? 4u : 0u) | (ppx == LPW - 1 ? 8u : 0u) |
                        (pr >= LPP ? 16u : 0u) | 32u) & edges;
    lds_dma16(xr, lds_base + buf * PBUF + (wave + NW * sl) * 1024, f ? LOOB : rel, pbase);
  };
  auto issue_w = [&](int stage, int tap, int kc, int nblk, bool valid) {
    const unsigned soff = (unsigned)(((tap * p.Cout + nblk * BN) * p.Cin + kc * 64) * 2);
#pragma unroll
    for (int j = 0; j < BI; ++j)
      lds_dma16(wr, lds_base + OFF_W + stage * WST + (wave + NW * j) * 1024, valid ? bbase[j] : LOOB, soff);
  };

  // INORM: normalise the patch of (item, K chunk kc) in place (see the note above the kernel)
  auto normalise_patch = [&](const Item& it, int kc) {
#pragma unroll 1
    for (int h2 = 0; h2 < 2; ++h2) {
      const int lc = 2 * wave + h2;                 // wave-uniform logical chunk: channels kc * 64 + 8 lc .. + 7
      float nsc[8], nsh[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { nsc[e] = p.in_scale[kc * 64 + lc * 8 + e]; nsh[e] = p.in_shift[kc * 64 + lc * 8 + e]; }
#pragma unroll 1
      for (int i = 0; i < (LPP + 63) / 64; ++i) {
        int lane_here = lane;
        asm volatile("" : "+v"(lane_here));         // opaque: nothing of this loop is hoisted across the tap loops (and then spilled)
        const int pr = lane_here + 64 * i;
        const int ppy = (pr * 1928) >> 16, ppx = pr - ppy * LPW;       // pr / 34
        const int ly = it.ly0 + ppy - 1, lx = it.lx0 + ppx - 1;
        if (pr < LPP && (unsigned)ly < (unsigned)q.Hs && (unsigned)lx < (unsigned)q.Ws) {
          unsigned char* a = smem + pr * 128 + ((lc ^ ((ppx >> 1) & 7)) << 4);
          float v[8];
          Chunk<bf16_t>::unpack(ld16(a), v);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], nsc[e], nsh[e]), 0.f);
          st16(a, Chunk<bf16_t>::pack(v));
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    raw_barrier();
  };
  // ACTIVATION WRITE-BACK (at the END of a chunk, when every wave has issued its last MFMA on the patch and before the next
  // chunk's patch overwrites it): the 8 x 32 interior of the normalised patch = this item's 256 pixels x 64 channels of the
  // activation relu(scale * y + shift), exactly the bytes a stand-alone normalisation pass would have stored.  Written by the
  // items of n-block 0 only (every n-block stages the same patch); 8 lanes = the 128-byte line of one pixel.  The LTH stores of a
  // lane are older than the patch slices issued right behind them, so the wait of the next tap 0 ("the patch has landed") covers
  // them: VMEM retires in order, no counted wait changes.
  auto write_back = [&](const Item& it, int kc) {
    if (p.act_out != nullptr && it.nblk == 0) {
      const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(p.act_out, 0, 0xFFFFFFFFu, 0x00020000);
      const unsigned ldab = (unsigned)(p.ld_act * 2);
      // scalar part of the address: the item's first pixel and this chunk's channels; per-lane part: (row, column, 16-byte piece)
      const unsigned sbase = (unsigned)__builtin_amdgcn_readfirstlane(
          (int)((unsigned)((it.img * p.Ho + it.ly0 * d + it.phy) * p.Wo + it.lx0 * d + it.phx) * ldab + (unsigned)(kc * 128)));
      int tid_here = tid;
      asm volatile("" : "+v"(tid_here));            // opaque: nothing of this block is hoisted across the tap loops
      const int c = tid_here & 7, colp = (tid_here >> 3) & 31, ppx = colp + 1;
      unsigned la = (unsigned)((LPW + ppx) * 128 + ((c ^ ((ppx >> 1) & 7)) << 4));       // patch row 1 (+ LPW * 128 per row)
      unsigned ga = (unsigned)(colp * d) * ldab + (unsigned)(c * 16);                      // lattice row 0 (+ d * Wo * ldab per row)
      const unsigned grow = (unsigned)(d * p.Wo) * ldab;
#pragma unroll 1
      for (int row = 0; row < LTH; ++row) {
        const u32x4 v = ld16(smem + la);
        __builtin_amdgcn_raw_buffer_store_b128(v, ar, ga, sbase, 0);
        la += LPW * 128;
        ga += grow;
      }
    }
  };

  f32x4 acc[NG][MT][4];
  float binit[NG][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};    // conv bias of this lane's channels: the accumulators START from it
  auto zero_acc = [&]() {
#pragma unroll
    for (int ng = 0; ng < NG; ++ng)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[ng][i][j][e] = binit[ng][j];
  };

  // one tap = four groups of 16 MFMAs, (k half g, channel group ng) = (0,0) (0,1) (1,0) (1,1): 4 A fragments of half g x 4 B
  // fragments of (g, ng).  FRAGMENT READS ARE ROLLED INTO THE MFMA STREAM (round 4).  The round-3 form read the 8 fragments
  // of a group in front of its MFMAs: the ISA showed four exposed LDS latencies per tap -- a full one behind the barrier, a
  // full one at g = 1, and the B fragments of the two ng = 1 groups issued with two MFMAs of cover -- about 600 cycles against
  // 1024 cycles of MFMA work.  Now every fragment register is re-loaded right behind the LAST MFMA that reads it, with the
  // operand it holds next: in the ng = 0 groups the MFMAs run B-fragment-major and fb[j] <- B(g, 1, j) after its four
  // MFMAs (12 MFMAs of cover); in the ng = 1 groups they run A-fragment-major and fa[i] <- A of the next half -- or of the
  // NEXT TAP's first half: the patch is resident for the whole chunk, so those reads cross the tap barrier; the B
  // fragments of (1, 0) slip into the last A batch of (0, 1) one MFMA apart.  What stays exposed is B(0, 0) behind the tap
  // barrier (the weights of a tap are only known to have landed there) and the A fragments of a chunk's first tap.
  // No extra registers: the A set is live across the barrier instead of dead there.
  u32x4 fa[MT];
  auto lda = [&](int ky, int kx, int g, int i) {
    return ld16(smem + aoff[kx][g] + ((i >> 1) + ky) * (LPW * 128) + (i & 1) * (16 * 128));
  };
  auto mma = [&](int ng, int i, int j, const u32x4& b) {
    acc[ng][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, b),
                                                            acc[ng][i][j], 0, 0, 0);
  };
  // `issue`: the DMA statements of this tap (next tap's weights).  They go BEHIND the tap's first fragment reads: the reads
  // are in flight while the wave gets its four DMA instructions accepted (60-100 cycles each), instead of starting after them.
  auto compute_tap = [&](int t, const int (&bb)[2], auto&& issue) {
    const int ky = t / 3, kx = t % 3, nky = (t + 1) / 3, nkx = (t + 1) % 3;
    u32x4 fb[4];
    if (t == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = lda(ky, kx, 0, i);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (g == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = ld16(smem + bb[0] + j * 16 * 128);
        issue();
      }
      // (g, 0): B-major, fb[j] <- B(g, 1, j) behind its last reader
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < MT; ++i) mma(0, i, j, fb[j]);
        fb[j] = ld16(smem + bb[g] + (4 + j) * 16 * 128);
        __builtin_amdgcn_sched_barrier(0);
      }
      // (g, 1): A-major, fa[i] <- A(next half / next tap's first half, i) behind its last reader
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          mma(1, i, j, fb[j]);
          if (g == 0 && i == MT - 1) {             // last batch of (0, 1): fb[j] is free, B(1, 0, j) moves in
            fb[j] = ld16(smem + bb[1] + j * 16 * 128);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (g == 0) fa[i] = lda(ky, kx, 1, i);
        else if (t < 8) fa[i] = lda(nky, nkx, 0, i);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- epilogue of one item -----------------------------------------------------------------------------------------------
  const unsigned ldob = (unsigned)(p.ldo * 2), ldyb = (unsigned)(p.bn_ldy * 2);
  const int xb = (rb == 0) ? 0 : (rb == 1 ? 1 : (rb == 2 ? 9 : 8));          // pixel of accumulator row 4*rb + v = xb + 2v
  // MODE_BNBWD: the consumer stage's saved conv outputs of this item, fetched YT taps before the item's last MFMA.  They
  // sit in the VM queue between the weight DMAs: the counted waits of the two taps after the fetch leave them in flight
  // (+NY), from the third tap on an in-order wait would require them -- so the fetch goes at tap 6 of the last chunk.
  // MODE_BNBWD: no prefetch of the saved outputs and no resident per-channel constants here (64 + 32 registers next to 128
  // accumulators): the epilogue loads both itself; that latency is what the second workgroup of the CU is for.
  auto item_offsets = [&](const Item& it, unsigned (&voff)[MT], unsigned (&yoff)[MT]) {
    const int col = it.nblk * BN + 4 * c16;                                  // group 0; group 1: + 64 channels
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int gm = wm * MT + i;                                            // M tile of the workgroup: row gm >> 1, half gm & 1
      const int Y = (it.ly0 + (gm >> 1)) * d + it.phy, X = (it.lx0 + 16 * (gm & 1) + xb) * d + it.phx;
      const unsigned pix = (unsigned)((it.img * p.Ho + Y) * p.Wo + X);
      voff[i] = pix * ldob + (unsigned)(col * 2);
      yoff[i] = pix * ldyb + (unsigned)(col * 2);
    }
  };
  Epi16Consts ec[NG];
  auto load_consts = [&](int nblk) {
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) {
      if (MODE == MODE_BNBWD) continue;                    // loaded per item in the epilogue
      ec[ng] = epi16_consts<MODE>(p, nblk * BN + ng * 64 + 4 * c16);
      if (MODE == MODE_STORE || MODE == MODE_STATS) {      // "+ bias" modes: fold it into the accumulator init
#pragma unroll
        for (int k = 0; k < 4; ++k) { binit[ng][k] = ec[ng].k1[k]; ec[ng].k1[k] = 0.f; }
      }
    }
  };
  float tot_su = 0.f, tot_sq = 0.f;               // lanes tid < BN: running statistics of channel nblk * BN + tid
  auto epilogue = [&](const Item& it) {
    __builtin_amdgcn_s_setprio(0);
    bool tile_ok[MT];
    unsigned voff[MT], yoff[MT];
    item_offsets(it, voff, yoff);
#pragma unroll
    for (int i = 0; i < MT; ++i) tile_ok[i] = true;
    const unsigned rbytes = 2u * (unsigned)d * ldob, yrbytes = 2u * (unsigned)d * ldyb;   // accumulator rows are 2 lattice pixels apart
    float s4[NG][4], q4[NG][4];
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { s4[ng][k] = 0.f; q4[ng][k] = 0.f; }
      unsigned vo[MT], yo[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) { vo[i] = voff[i] + (unsigned)(ng * 128); yo[i] = yoff[i] + (unsigned)(ng * 128); }   // 64 channels further
      if (MODE == MODE_BNBWD) {
        const Epi16Consts ecl = epi16_consts<MODE>(p, it.nblk * BN + ng * 64 + 4 * c16);
        epilogue16c<MODE, MT, false>(p, acc[ng], tile_ok, vo, rbytes, yo, yrbytes, ecl, s4[ng], q4[ng]);
      } else {
        epilogue16c<MODE, MT, false>(p, acc[ng], tile_ok, vo, rbytes, yo, yrbytes, ec[ng], s4[ng], q4[ng]);
      }
    }
    if (MODE == MODE_STATS || MODE == MODE_BNBWD) {
      // per-workgroup statistics rows as in igemm_lattice_kernel; scratch [wave][group][4 k][2][16 c]
      float* red = reinterpret_cast<float*>(smem + OFF_RED);
#pragma unroll
      for (int ng = 0; ng < NG; ++ng) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s4[ng][k] += __shfl_xor(s4[ng][k], 16, 64); q4[ng][k] += __shfl_xor(q4[ng][k], 16, 64);
          s4[ng][k] += __shfl_xor(s4[ng][k], 32, 64); q4[ng][k] += __shfl_xor(q4[ng][k], 32, 64);
        }
        if (rb == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            red[(((wave * NG + ng) * 4 + k) * 2 + 0) * 16 + c16] = s4[ng][k];
            red[(((wave * NG + ng) * 4 + k) * 2 + 1) * 16 + c16] = q4[ng][k];
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      raw_barrier();
      if (tid < BN) {
        const int ng2 = tid >> 6, cc = tid & 63, c2 = cc >> 2, k = cc & 3;
        float su = 0.f, sq = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < WM; ++w2) {
          su += red[(((w2 * NG + ng2) * 4 + k) * 2 + 0) * 16 + c2];
          sq += red[(((w2 * NG + ng2) * 4 + k) * 2 + 1) * 16 + c2];
        }
        tot_su += su;                                       // tiles in the order this workgroup visits them: reproducible
        tot_sq += sq;
      }
    }
    __builtin_amdgcn_s_setprio(1);
    zero_acc();
  };

  // ---- the pipeline ----------------------------------------------------------------------------------------------------
  // flat step q = 9*chunk + tap; weights of step q live in ring stage q & 1 and are issued behind the barrier of step q - 1;
  // the patch of a chunk is issued behind the last barrier of the previous chunk (single buffer).
  // Two independent workgroups share every SIMD; VALU / MFMA issue is arbitrated by priority, then age (MI355X_MICROARCH.md,
  // "Two waves per SIMD").  The tap loops run at priority 1 and the epilogue (300-600 VALU instructions per item) at 0, so
  // that a workgroup in its MFMA phase is not held up by its neighbour's epilogue: -0.05 ... -0.07 ms per training step,
  // three of three interleaved pairs (profiles/r03_setprio_ab.txt).
  __builtin_amdgcn_s_setprio(1);
  Item cur = decode(first), nxt = cur;
  int item = first;
  load_consts(cur.nblk);
  zero_acc();
  issue_w(0, 0, 0, cur.nblk, true);
#pragma unroll
  for (int sl = 0; sl < PJ; ++sl) issue_slice(sl, 0, patch_base(cur, 0), patch_edges(cur, true));
  int par = 0;                                    // parity of the flat step of this chunk's tap 0
  bool boundary = false;                          // the previous chunk ended an item: its NST stores are the youngest VMEM operations
  for (;;) {
    for (int kc = 0; kc < q.nkc; ++kc) {
      const bool last_kc = kc + 1 == q.nkc;
      const int item_n = last_kc ? item + G : item, kc_n = last_kc ? 0 : kc + 1;
      const bool have_n = item_n < q.items;
      if (last_kc) nxt = have_n ? decode(item_n) : cur;
      const unsigned pb_n = (unsigned)__builtin_amdgcn_readfirstlane((int)(last_kc ? patch_base(nxt, 0) : patch_base(cur, kc_n)));
      const unsigned pe_n = (unsigned)__builtin_amdgcn_readfirstlane((int)(last_kc ? patch_edges(nxt, have_n) : patch_edges(cur, true)));
      const int nblk_n = __builtin_amdgcn_readfirstlane(nxt.nblk);
      int bb0[2], bb1[2];                         // fragment bases of the even / odd taps of this chunk
#pragma unroll
      for (int g = 0; g < 2; ++g) { bb0[g] = boff[g] + par * WST; bb1[g] = boff[g] + (par ^ 1) * WST; }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        // W(t) was issued one tap ago and nothing younger exists -- except, at t = 0 behind an item boundary, the epilogue stores
        if (t == 0 && boundary) wait_vmcnt<NST>(); else wait_vmcnt<0>();
        raw_barrier();
        if (INORM && t == 0) normalise_patch(cur, kc);       // the patch of this chunk has landed (every wave's pieces)
        auto issue = [&]() {
          if (t < 8) issue_w((par + t + 1) & 1, t + 1, kc, cur.nblk, true);
          else issue_w((par + 9) & 1, 0, kc_n, nblk_n, have_n);
        };
        if (t & 1) compute_tap(t, bb1, issue); else compute_tap(t, bb0, issue);
      }
      raw_barrier();                              // every wave has issued the MFMAs of tap 8: the patch buffer is free
      if (INORM && p.act_out != nullptr) {
        write_back(cur, kc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();                            // every wave has READ its pieces: the next patch may overwrite them
      }
#pragma unroll
      for (int sl = 0; sl < PJ; ++sl) issue_slice(sl, 0, pb_n, pe_n);
      boundary = last_kc;
      if (last_kc) epilogue(cur);
      par ^= 1;
    }
    item += G;
    if (item >= q.items) break;
    if (nxt.nblk != cur.nblk) { load_consts(nxt.nblk); zero_acc(); }
    cur = nxt;
  }
  if ((MODE == MODE_STATS || MODE == MODE_BNBWD) && tid < BN) {
    constexpr int nrow = (MODE == MODE_BNBWD) ? 3 : 2;
    const int nblk0 = first - (int)udiv((unsigned)first, q.mg_nblocks, (unsigned)q.nblocks) * q.nblocks;
    const int r0 = (first - nblk0) / q.nblocks;              // this workgroup's row; its n-block never changed
    float* col = p.stats + nblk0 * BN + tid;
    float* row = col + (long)r0 * nrow * p.Cout;
    row[0] = tot_su;
    row[p.Cout] = tot_sq;
    if (nrow == 3) row[2 * p.Cout] = 0.f;
    for (int r = r0 + q.stat_rows; r < q.mtiles; r += q.stat_rows) {   // rows a per-tile writer would have produced: zeros
      float* z = col + (long)r * nrow * p.Cout;
      z[0] = 0.f;
      z[p.Cout] = 0.f;
      if (nrow == 3) z[2 * p.Cout] = 0.f;
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
static int lattice_enabled() {
  static int v = -1;                              // UNETDC_LATTICE=0: round-1 kernels (A/B measurements)
  if (v < 0) { const char* e = getenv("UNETDC_LATTICE"); v = (e && e[0] == '0') ? 0 : 1; }
  return v;
}

bool igemm_lattice_supported(const IgemmParams& p, int dtype) {
  if (!lattice_enabled() || dtype != UNETDC_BF16) return false;
  if (p.ntaps != 9 || p.stride != 1 || p.mode == MODE_SHUFFLE) return false;
  if (p.Ho != p.Hi || p.Wo != p.Wi) return false;
  const int d = p.offy[8];
  if (d < 1 || p.offx[8] != d || p.offy[0] != -d || p.offx[0] != -d) return false;
  if (p.Ho % d != 0 || p.Wo % d != 0) return false;
  if ((p.Ho / d) % LTH != 0 || (p.Wo / d) % LTW != 0) return false;
  if (p.Cin % 64 != 0 || p.Cout % 64 != 0) return false;
  const long HoWo = (long)p.Ho * p.Wo;
  if (p.M % HoWo != 0) return false;
  {                                               // a workgroup must keep its n-block: nblocks divides the persistent grid
    const bool wide = p.Cout % 128 == 0;
    const int nb = p.Cout / (wide ? 128 : 64);
    if ((256 * (wide ? 1 : 2)) % nb != 0) return false;
  }
  const long xbytes = (p.M / HoWo) * p.Hi * p.Wi * p.ldx * 2L;
  const long obytes = (long)p.M * p.ldo * 2, ybytes = p.mode == MODE_BNBWD ? (long)p.M * p.bn_ldy * 2 : 0;
  const long wbytes = 9L * p.Cout * p.Cin * 2;
  return xbytes < (1L << 31) && wbytes < (1L << 31) && obytes < (1L << 32) && ybytes < (1L << 32);
}

// input normalisation on load: the statistics forward of the 4-wave forms (64-channel n-blocks: constants in LDS, Cin <= 256;
// 128-channel n-blocks: constants through the scalar cache, any Cin)
bool igemm_lattice_bnin_supported(const IgemmParams& p, int dtype) {
  return igemm_lattice_supported(p, dtype) && p.mode == MODE_STATS && (p.Cout % 128 == 0 || p.Cin <= 256);
}

// the wide form can also store the normalised activation (IgemmParams::act_out)
bool igemm_lattice_bnin_writes_activation(const IgemmParams& p, int dtype) {
  return igemm_lattice_bnin_supported(p, dtype) && p.Cout % 128 == 0 && (long)p.M * p.Cin * 2 < (1L << 32);
}

static long lattice_grid(long items, int wgs_per_cu) {
  const long g = 256L * wgs_per_cu;
  return g > items ? items : g;
}

template <int WM, int WN, int MT, int NPB, int MODE, bool INORM = false>
static int launch_lattice_cfg(IgemmParams& p, const LatticeParams& q, int wgs_per_cu, hipStream_t stream) {
  constexpr int NW = WM * WN, BN = WN * 64;
  constexpr int PJ = (LPI + NW - 1) / NW;
  constexpr int LDS = NPB * PJ * NW * 1024 + 3 * BN * 128 + NW * 128 * 4 + (INORM ? 2048 : 0);   // INORM: + [2][Cin <= 256] constants
  if (const int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&igemm_lattice_kernel<WM, WN, MT, NPB, MODE, INORM>), LDS, "igemm_lattice_kernel")) return rc_;
  // persistent grid: as many workgroups as fit the chip at once (256 CUs), never more than there are items
  const long grid = lattice_grid(q.items, wgs_per_cu);
  if (grid % q.nblocks != 0 || q.stat_rows != (int)(grid / q.nblocks)) {
    set_error("igemm_lattice: grid %ld / nblocks %d / stat_rows %d inconsistent", grid, q.nblocks, q.stat_rows);
    return UNETDC_ELAUNCH;
  }
  hipLaunchKernelGGL((igemm_lattice_kernel<WM, WN, MT, NPB, MODE, INORM>), dim3((unsigned)grid), dim3(NW * 64), LDS, stream, p, q);
  char nm[96];
  snprintf(nm, sizeof(nm), "igemm_lattice_kernel<%d, %d, %d, %d, %d>%s", WM, WN, MT, NPB, MODE, INORM ? " bnin" : "");
  note_kernel(nm);
  return check_launch("igemm_lattice_kernel");
}

template <int MODE, bool INORM = false>
static int launch_lattice_wide_cfg(IgemmParams& p, const LatticeParams& q, hipStream_t stream) {
  constexpr int NW = 4, BN = 128, PJ = (LPI + NW - 1) / NW;
  constexpr int LDS = PJ * NW * 1024 + 2 * BN * 128 + NW * 2 * 128 * 4;       // 44 KB patch + 2 x 16 KB weights + 4 KB scratch = 80 KB
  if (const int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&igemm_lattice_wide_kernel<MODE, INORM>), LDS, "igemm_lattice_wide_kernel")) return rc_;
  const long grid = lattice_grid(q.items, 2);
  if (grid % q.nblocks != 0 || q.stat_rows != (int)(grid / q.nblocks)) {
    set_error("igemm_lattice_wide: grid %ld / nblocks %d / stat_rows %d inconsistent", grid, q.nblocks, q.stat_rows);
    return UNETDC_ELAUNCH;
  }
  hipLaunchKernelGGL((igemm_lattice_wide_kernel<MODE, INORM>), dim3((unsigned)grid), dim3(NW * 64), LDS, stream, p, q);
  char nm[96];
  snprintf(nm, sizeof(nm), "igemm_lattice_wide_kernel<%d>%s", MODE, INORM ? " bnin" : "");
  note_kernel(nm);
  return check_launch("igemm_lattice_wide_kernel");
}

template <int WM, int WN, int MT, int NPB>
static int launch_lattice_mode(IgemmParams& p, const LatticeParams& q, int wgs, hipStream_t stream) {
  switch (p.mode) {
    case MODE_STATS: return launch_lattice_cfg<WM, WN, MT, NPB, MODE_STATS>(p, q, wgs, stream);
    case MODE_AFFINE_RELU: return launch_lattice_cfg<WM, WN, MT, NPB, MODE_AFFINE_RELU>(p, q, wgs, stream);
    case MODE_BNBWD: return launch_lattice_cfg<WM, WN, MT, NPB, MODE_BNBWD>(p, q, wgs, stream);
    default: return launch_lattice_cfg<WM, WN, MT, NPB, MODE_STORE>(p, q, wgs, stream);
  }
}

int launch_igemm_lattice(IgemmParams& p, hipStream_t stream) {
  LatticeParams q{};
  q.d = p.offy[8];
  q.Hs = p.Ho / q.d;
  q.Ws = p.Wo / q.d;
  q.tiles_x = q.Ws / LTW;
  q.tiles_y = q.Hs / LTH;
  q.tiles_per_img = q.d * q.d * q.tiles_x * q.tiles_y;
  q.nkc = p.Cin / 64;
  auto magic = [](unsigned dv) { return dv <= 1 ? 0u : (unsigned)(((1ull << 32) + dv - 1) / dv); };
  const int nimg = (int)((long)p.M / ((long)p.Ho * p.Wo));
  q.mtiles = nimg * q.tiles_per_img;              // = M / 256
  const bool wide = p.Cout % 128 == 0;
  // 4 waves x (64 pixels x 128 channels), two workgroups per CU: the statistics, folded-BatchNorm and fused-BatchNorm-backward
  // forms; the plain-store instantiation spills 36 registers -- its reloads would drain the VM counter in front of the
  // hand-placed DMAs -- and keeps the 8-wave form (4 x 2 waves of 64 x 64, two patch buffers)
  const bool ww = wide && (p.mode == MODE_STATS || p.mode == MODE_AFFINE_RELU || p.mode == MODE_BNBWD);
  const int wgs = (wide && !ww) ? 1 : 2;          // workgroups per CU of the configurations
  q.nblocks = p.Cout / (wide ? 128 : 64);
  q.items = q.mtiles * q.nblocks;
  // statistics: one row per workgroup and n-block (the kernel zero-fills the rest of the M / 256 rows).  The grid is
  // min(items, 256 * wgs): a multiple of nblocks whenever nblocks divides 256 * wgs (every channel count of the networks).
  q.stat_rows = (int)(lattice_grid(q.items, wgs) / q.nblocks);
  p.mblocks = q.stat_rows;
  q.mg_nblocks = magic(q.nblocks); q.mg_tpi = magic(q.tiles_per_img); q.mg_tpp = magic(q.tiles_x * q.tiles_y);
  q.mg_d = magic(q.d); q.mg_tx = magic(q.tiles_x);
  p.nblocks = q.nblocks;
  if (ww) {
    if (p.mode == MODE_STATS && p.in_scale) return launch_lattice_wide_cfg<MODE_STATS, true>(p, q, stream);
    if (p.mode == MODE_STATS) return launch_lattice_wide_cfg<MODE_STATS>(p, q, stream);
    if (p.mode == MODE_BNBWD) return launch_lattice_wide_cfg<MODE_BNBWD>(p, q, stream);
    return launch_lattice_wide_cfg<MODE_AFFINE_RELU>(p, q, stream);
  }
  if (wide) return launch_lattice_cfg<4, 2, 4, 2, MODE_STORE>(p, q, 1, stream);   // (wide && !ww: the plain-store mode only)
  if (p.in_scale) return launch_lattice_cfg<4, 1, 4, 1, MODE_STATS, true>(p, q, 2, stream);      // igemm_lattice_bnin_supported
  return launch_lattice_mode<4, 1, 4, 1>(p, q, 2, stream);
}

}  // namespace unetdc
