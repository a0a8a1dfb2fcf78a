// Weight gradient of the composed decoder up-path (bf16, round 5; upconv_compose.hip has the algebra):
//
//   dW'[phase p][tap t][ci][co] = sum_{n, r, s} h[n, r + py + ty - 1, s + px + tx - 1, ci] * dY[n, 2r + py, 2s + px, co]
//
// i.e. per output phase a 2 x 2-tap weight gradient between LOW-RES tensors: X = h (the up-conv's input, zero outside the
// image) and the phase-p sub-lattice of dY (the gradient of decN.0's conv output).  16 blocks [2C][C]; dW3[:, :C] and dWT come out
// of them by two small GEMMs (upc_gemm_kernel<1>, <2>).
//
// Structure = convt_wgrad.hip with the operands' roles swapped: a workgroup owns ONE phase, 128 ci x 64 co, ALL FOUR taps;
// wave w = tap (ty, tx) = (w >> 1, w & 1) with 8 x 4 accumulator tiles of 16 x 16; every wave reads the same dY fragments and
// its own SHIFTED X fragments:
//   * K step = 32 low-res pixels of one row r of a 32-pixel-wide column strip.  The dY row 2r + py is fetched with a per-lane
//     source pixel 2 (s0 + i) + px, so in LDS the phase looks dense ([32 pixels][64 channels]);
//   * X is staged ONCE per row with its halo: [34 pixels s0 - 1 .. s0 + 32][128 channels] as two 64-channel images; tap (ty, tx)
//     reads row slot (r + py - 1 + ty) at pixel offset px + tx through Frag16 (conflict-free at every row offset).  A strip is
//     walked top to bottom, so each step brings in ONE new X row and one dY row: 12.7 KB per 128 MFMAs;
//   * four-slot rings, DMAs from inline asm with counted waits: group G(m) = {X row m + 1, dY row m} = 4 instructions per wave
//     (padded with out-of-range dummies), issued two steps ahead.
// Slabs part[unit][p * 4 + t][ci][co] (fp32, one per workgroup) are summed in unit order by upc_wgrad_reduce_kernel, which also
// writes the two bf16 layouts the decomposition GEMMs read.
#include <stdio.h>

#include "kernels.h"
#include "lds_dma.h"
#include "wgrad_frag.h"

namespace unetdc {

struct UpcWgradParams {
  const void* x;      // [N, H, W, ldx]         low-res tensor h, channels -> ci (2C)
  const void* dy;     // [N, 2H, 2W, lddy]      gradient of decN.0's conv output, channels -> co (C)
  float* part;        // [units][16][CI][CJ]
  int N, H, W, CI, CJ, ldx, lddy;
  int itiles, jtiles; // CI / 128, CJ / 64
  int strips;         // W / 32
  int ipu;            // images per unit (a workgroup walks ipu images of one column strip)
  int units;          // strips * ceil(N / ipu)
};

constexpr int UWG_XIMG = 40 * 128;                 // one 64-channel X image: 34 pixel rows (+ 6 of padding), 128 bytes each
constexpr int UWG_XSLOT = 2 * UWG_XIMG;            // 128 channels
constexpr int UWG_YSLOT = 32 * 128;
constexpr int UWG_NS = 4;
constexpr int UWG_DUMP = UWG_NS * (UWG_XSLOT + UWG_YSLOT);  // 1 KB that only the padding DMAs write (zeros, never read)
constexpr int UWG_LDS = UWG_DUMP + 1024;                    // 57 KB
constexpr unsigned UWG_OOB = 0x80000000u;

__global__ __launch_bounds__(256, 2) void upc_wgrad_kernel(const UpcWgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // = tap 2 ty + tx
  const unsigned lds_base = lds_addr_of(smem);
  unsigned char* const xs = smem;
  unsigned char* const ys = smem + UWG_NS * UWG_XSLOT;

  // workgroup -> (unit, phase, ci tile, co tile)
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = 4 * p.itiles * p.jtiles;
  const int unit = L / tiles, trem = L - unit * tiles;
  const int ph = trem / (p.itiles * p.jtiles), t2 = trem - ph * (p.itiles * p.jtiles);
  const int it = t2 / p.jtiles, jt = t2 - it * p.jtiles;
  const int py = ph >> 1, px = ph & 1;
  const int i0 = it * 128, j0 = jt * 64;
  const int strip = unit % p.strips, ug = unit / p.strips;
  const int s0 = strip * 32;
  const int n0 = ug * p.ipu, n1 = min(n0 + p.ipu, p.N);
  const int ty = wave >> 1, tx = wave & 1;

  const unsigned xbytes = (unsigned)((long)p.N * p.H * p.W * p.ldx * 2);
  const unsigned dybytes = (unsigned)((long)p.N * 4 * p.H * p.W * p.lddy * 2);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, dybytes, 0x00020000);

  // ---- DMA slots of one group: 16 wave-instructions (8 pixel rows x 128 bytes each); wave w issues gi = w + 4 j, j = 0..3
  //   gi < 10 : X image gi / 5, rows 8 (gi % 5) ..      10 <= gi < 14: dY rows 8 (gi - 10) ..      gi >= 14: dummy
  const int sub = lane >> 3, pc = lane & 7;
  unsigned voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int gi = wave + 4 * j;
    if (gi < 10) {
      const int img = gi / 5, row = (gi - 5 * img) * 8 + sub;        // image row = pixel s0 - 1 + row
      const int sx = s0 - 1 + row;
      const unsigned ch = (unsigned)(Frag16::src_chunk(row, pc) * 16);
      voff[j] = (row < 34 && (unsigned)sx < (unsigned)p.W) ? (unsigned)((sx * p.ldx + i0 + img * 64) * 2) + ch : UWG_OOB;
    } else if (gi < 14) {
      const int row = (gi - 10) * 8 + sub;                            // low-res pixel s0 + row of the phase's sub-lattice
      const unsigned ch = (unsigned)(Frag16::src_chunk(row, pc) * 16);
      voff[j] = (unsigned)(((2 * (s0 + row) + px) * p.lddy + j0) * 2) + ch;
    } else {
      voff[j] = UWG_OOB;
    }
  }
  // group g of image n: X row g (low-res row g + py - 1) -> X slot g & 3;  dY row g - 1 (low-res row g - 1) -> dY slot (g - 1) & 3
  auto issue = [&](int n, int g, bool with_dy) {
    const int xrow = g + py - 1;
    const bool xok = (unsigned)xrow < (unsigned)p.H;
    const unsigned xsoff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((n * p.H + (xok ? xrow : 0)) * p.W) * (unsigned)p.ldx * 2u));
    const int yr = g - 1;                                             // low-res row of the dY row of this group
    const unsigned ysoff = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)((unsigned)((n * 2 * p.H + 2 * (with_dy ? yr : 0) + py) * 2 * p.W) * (unsigned)p.lddy * 2u));
    const unsigned xb = lds_base + (unsigned)((g & 3) * UWG_XSLOT), yb = lds_base + (unsigned)(UWG_NS * UWG_XSLOT + ((g - 1) & 3) * UWG_YSLOT);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gi = wave + 4 * j;                                    // wave-uniform
      if (gi < 10) lds_dma16(xr, xb + (gi / 5) * UWG_XIMG + (gi % 5) * 1024, xok ? voff[j] : UWG_OOB, xsoff);
      else if (gi < 14) lds_dma16(dyr, yb + (gi - 10) * 1024, with_dy ? voff[j] : UWG_OOB, ysoff);
      else lds_dma16(dyr, lds_base + UWG_DUMP, UWG_OOB, 0u);          // padding: keeps the per-wave count of a group at 4
    }
  };

  // ---- fragment offsets: A = X at pixel offset px + tx (16 channels from 16 c of a 64-channel image), B = dY
  int fa_off[4][2], fb_off[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      fa_off[c][jj] = Frag16::rd_off(lane, 16 * c, px + tx, jj);
      fb_off[c][jj] = Frag16::rd_off(lane, 16 * c, 0, jj);
    }

  f32x4 acc[8][4];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[c][j][e] = 0.f;

  for (int n = n0; n < n1; ++n) {
    // groups 0 .. H: group g carries X row g and dY row g - 1; step m (low-res row m) needs groups <= m + 1
    issue(n, 0, false);
    issue(n, 1, true);
    issue(n, 2, p.H > 1);
    for (int m = 0; m < p.H; ++m) {
      // this wave's pieces of groups <= m + 1 (group m + 2 may stay in flight)
      if (m + 2 <= p.H) wait_vmcnt<4>(); else wait_vmcnt<0>();
      raw_barrier();                                                  // everyone's; and everyone has issued the MFMAs of step m - 1
      const unsigned char* sx = xs + ((m + ty) & 3) * UWG_XSLOT;     // X row m + ty  <->  low-res row m + py - 1 + ty
      const unsigned char* sd = ys + (m & 3) * UWG_YSLOT;
      bf16x8 fb[4], fa[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = Frag16::frag_at(sd, fb_off[j][0], fb_off[j][1]);
#pragma unroll
      for (int c = 0; c < 8; ++c) fa[c] = Frag16::frag_at(sx + (c >> 2) * UWG_XIMG, fa_off[c & 3][0], fa_off[c & 3][1]);
      if (m + 3 <= p.H) issue(n, m + 3, true);                        // X slot (m + 3) & 3 held row m - 1: last read at step m - 1
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[c], fb[j], acc[c][j], 0, 0, 0);
    }
    raw_barrier();                                                    // the rings are free for the next image's first groups
  }

  // accumulator element v of a 16 x 16 tile: row (ci) 4 * (lane >> 4) + v, column (co) lane & 15
  const int col = lane & 15, rq = lane >> 4;
  float* slab = p.part + (((long)unit * 16 + ph * 4 + wave) * p.CI) * p.CJ;
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

// dwbt[z][ci][co] = sum_units part[u][z][ci][co] (unit order: reproducible), as bf16; dwb[z][co][ci] = its transpose.
// One workgroup per (z, 64 ci x 64 co tile); the transpose goes through LDS so that both images are written in 128-byte runs.
__global__ __launch_bounds__(256) void upc_wgrad_reduce_kernel(const float* __restrict__ part, int units, bf16_t* __restrict__ dwb,
                                                               bf16_t* __restrict__ dwbt, int CI, int CJ) {
  __shared__ float tile[64][65];
  const int z = blockIdx.z, ci0 = blockIdx.y * 64, co0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long slab = 16L * CI * CJ;
  for (int r = ty; r < 64; r += 4) {
    const float* src = part + ((long)z * CI + ci0 + r) * CJ + co0 + tx;
    float s = 0.f;
    for (int u = 0; u < units; ++u) s += src[(long)u * slab];
    tile[r][tx] = s;
    dwbt[((long)z * CI + ci0 + r) * CJ + co0 + tx] = from_f32<bf16_t>(s);
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4)                                    // r = co, tx = ci
    dwb[((long)z * CJ + co0 + r) * CI + ci0 + tx] = from_f32<bf16_t>(tile[tx][r]);
}

// ------------------------------------------------------------------------------------------------
static bool upc_wgrad_plan(int N, int H, int W, int CI, int CJ, UpcWgradParams& p) {
  if (W % 32 != 0 || CI % 128 != 0 || CJ % 64 != 0 || H < 2) return false;
  p.itiles = CI / 128; p.jtiles = CJ / 64; p.strips = W / 32;
  const long tiles = 4L * p.itiles * p.jtiles;
  // ~512 workgroups (two per CU): a workgroup walks `ipu` images of its strip
  long ipu = (tiles * p.strips * N + 511) / 512;
  if (ipu < 1) ipu = 1;
  if (ipu > N) ipu = N;
  p.ipu = (int)ipu;
  p.units = p.strips * ((N + p.ipu - 1) / p.ipu);
  return true;
}

bool upc_wgrad_supported(int N, int H, int W, int CI, int CJ, int ldx, int lddy) {
  UpcWgradParams p{};
  if (!upc_wgrad_plan(N, H, W, CI, CJ, p)) return false;
  const long P = (long)N * H * W;
  return P * ldx * 2 < (1L << 31) && 4 * P * lddy * 2 < (1L << 31) && ldx % 8 == 0 && lddy % 8 == 0;
}

long upc_wgrad_workspace_bytes(int N, int H, int W, int CI, int CJ) {
  UpcWgradParams p{};
  if (!upc_wgrad_plan(N, H, W, CI, CJ, p)) return 0;
  return (long)p.units * 16 * CI * CJ * 4;
}

// dwb [16][CJ][CI], dwbt [16][CI][CJ] (bf16): the sixteen blocks of dW' in the two layouts launch_upc_decompose reads
int launch_upc_wgrad(const void* x, int ldx, const void* dy, int lddy, void* dwb, void* dwbt, void* workspace, long workspace_bytes,
                     int N, int H, int W, int CI, int CJ, hipStream_t stream) {
  UpcWgradParams p{};
  p.x = x; p.dy = dy; p.N = N; p.H = H; p.W = W; p.CI = CI; p.CJ = CJ; p.ldx = ldx; p.lddy = lddy;
  if (!upc_wgrad_plan(N, H, W, CI, CJ, p) || !upc_wgrad_supported(N, H, W, CI, CJ, ldx, lddy)) {
    set_error("upconv wgrad: unsupported shape");
    return UNETDC_EUNSUPPORTED;
  }
  const long need = (long)p.units * 16 * CI * CJ * 4;
  if (need > workspace_bytes) {
    set_error("upconv wgrad: workspace too small (%ld < %ld bytes)", workspace_bytes, need);
    return UNETDC_EWORKSPACE;
  }
  p.part = reinterpret_cast<float*>(workspace);
  if (const int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&upc_wgrad_kernel), UWG_LDS, "upc_wgrad_kernel")) return rc_;
  const long nwg = (long)p.units * 4 * p.itiles * p.jtiles;
  hipLaunchKernelGGL(upc_wgrad_kernel, dim3((unsigned)nwg), dim3(256), UWG_LDS, stream, p);
  note_kernel("upc_wgrad_kernel");
  int rc = check_launch("upc_wgrad_kernel");
  if (rc != UNETDC_OK) return rc;
  hipLaunchKernelGGL(upc_wgrad_reduce_kernel, dim3(CJ / 64, CI / 64, 16), dim3(256), 0, stream, p.part, p.units,
                     reinterpret_cast<bf16_t*>(dwb), reinterpret_cast<bf16_t*>(dwbt), CI, CJ);
  return check_launch("upc_wgrad_reduce_kernel");
}

}  // namespace unetdc
