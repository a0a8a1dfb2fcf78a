// Implicit-GEMM convolution on the gfx950 matrix cores (MFMA 32x32), NHWC activations.
//
//   out[m, n] = sum_{tap, k} X[pixel(m) shifted by tap, k] * Wp[tap][n][k]
//
//   M = output pixels (N*Ho*Wo, linear), N = output channels, K = ntaps * Cin.
//   Input pixel of output pixel (oy, ox) for tap t: (oy*stride + offy[t], ox*stride + offx[t]),
//   zero outside the image -> the reference's zero padding = dilation (models/model_2.py:41-44).
//
// One kernel serves
//   * dilated 3x3 conv forward      (9 taps, offsets (ky-1)*d)         nn.Conv2d  model_2.py:41-51
//   * dilated 3x3 conv dgrad        (same taps, flipped/transposed weights)  autograd of the above
//   * ConvTranspose2d(2,2,s=2) fwd  (1 tap, N = 4*Cout, pixel-shuffle store) model_2.py:20-29,67-76
//   * ConvTranspose2d dgrad         (4 taps, stride 2)
//
// Layout / hardware mapping (wave64, one 64x64 output tile per wave as 2x2 MFMA 32x32 tiles):
//   * K-step = 128 bytes of channels per pixel row (64 bf16 / 32 fp32): every global load is a
//     16-byte chunk, 8 consecutive lanes fetch one pixel's 128 contiguous bytes.
//   * LDS image: [rows][128 B], 16-byte chunk index XOR-swizzled with (row>>1)&7 so the
//     ds_read_b128 fragment reads (lane = row) are bank-conflict-free without padding.
//   * A 16-byte fragment feeds one v_mfma_f32_32x32x16_bf16 (bf16) or four
//     v_mfma_f32_32x32x2_f32 (fp32, exact fp32 FMA chain) -- same byte addressing for both types.
//   * MFMA N index j of N-tile ni maps to output channel 2j+ni (weights rows de-interleaved when
//     they are written to LDS), so each lane owns two ADJACENT channels per pixel and a half-wave
//     stores a full contiguous 128/256-byte pixel row straight from the accumulators.
//   * register-staged double buffering: global loads of step s+1 are in flight during the MFMAs
//     of step s; one barrier per K-step.
//   * taps whose shifted window misses the image for the whole block are skipped (bottleneck
//     d=16 on a 32x32 map: 5 of 9 taps for most blocks).
#include <stdlib.h>

#include <stdio.h>

#include "kernels.h"

namespace unetdc {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  __device__ static __forceinline__ void run(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                  acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  __device__ static __forceinline__ void run(f32x16& acc, const u32x4& a, const u32x4& b) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const unsigned int ua = a[s], ub = b[s];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bits_f32(ua), bits_f32(ub), acc, 0, 0, 0);
    }
  }
};

template <typename T> __device__ __forceinline__ void store_pair(T* dst, float v0, float v1);
template <> __device__ __forceinline__ void store_pair<float>(float* dst, float v0, float v1) {
  *reinterpret_cast<float2*>(dst) = make_float2(v0, v1);
}
template <> __device__ __forceinline__ void store_pair<bf16_t>(bf16_t* dst, float v0, float v1) {
  bf16_t lo = (bf16_t)v0, hi = (bf16_t)v1;
  *reinterpret_cast<unsigned int*>(dst) = (unsigned int)__builtin_bit_cast(unsigned short, lo) |
                                          ((unsigned int)__builtin_bit_cast(unsigned short, hi) << 16);
}

template <typename T, int WM, int WN>
__global__ __launch_bounds__(256, 2) void igemm_conv_kernel(const IgemmParams p) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int AI = BM / 32, BI = BN / 32;     // 16-byte chunks per thread per K-step
  constexpr int KE = 128 / (int)sizeof(T);      // channels per K-step
  constexpr int EPC = 16 / (int)sizeof(T);      // channels per 16-byte chunk
  constexpr int STAGE = (BM + BN) * 128;        // LDS bytes per pipeline stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mblk = tile / p.nblocks, nblk = tile - mblk * p.nblocks;
  const int m0 = mblk * BM, n0 = nblk * BN;
  const int c = tid & 7, r0 = tid >> 3;
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);

  // ---- per-thread description of the A rows (output pixels) it stages -------------------------
  int pix[AI], ys[AI], xs[AI];
  unsigned tapmask = 0;
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + r0 + 32 * i;
    if (m < p.M) {
      const int n = m / HoWo, rem = m - n * HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      ys[i] = oy * p.stride;
      xs[i] = ox * p.stride;
      pix[i] = (n * p.Hi + ys[i]) * p.Wi + xs[i];
    } else {
      ys[i] = -(1 << 28);
      xs[i] = 0;
      pix[i] = 0;
    }
    for (int t = 0; t < p.ntaps; ++t) {
      const int iy = ys[i] + p.offy[t], ix = xs[i] + p.offx[t];
      if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) tapmask |= 1u << t;
    }
  }
  {
    unsigned* sm_u = reinterpret_cast<unsigned*>(smem);
    if (tid == 0) sm_u[0] = 0;
    __syncthreads();
    if (tapmask) atomicOr(&sm_u[0], tapmask);
    __syncthreads();
    tapmask = sm_u[0];
    __syncthreads();
  }
  const int nkc = p.Cin / KE;
  const int nsteps = __popc(tapmask) * nkc;

  // ---- LDS addressing -------------------------------------------------------------------------
  const int swz_w = (r0 >> 1) & 7;                      // (row>>1)&7 is the same for rows r0+32i
  int a_wr[AI], b_wr[BI];
#pragma unroll
  for (int i = 0; i < AI; ++i) a_wr[i] = (r0 + 32 * i) * 128 + ((c ^ swz_w) << 4);
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int row = r0 + 32 * i, grp = row >> 6, cc = row & 63;
    const int lrow = grp * 64 + (cc & 1) * 32 + (cc >> 1);    // de-interleave: even channels first
    b_wr[i] = BM * 128 + lrow * 128 + ((c ^ ((lrow >> 1) & 7)) << 4);
  }
  const int r = lane & 31, h = lane >> 5;
  const int swz_r = (r >> 1) & 7;
  int a_rd[2][4], b_rd[2][4];
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = ((2 * g + h) ^ swz_r) << 4;
      a_rd[t2][g] = (wm * 64 + t2 * 32 + r) * 128 + ch;
      b_rd[t2][g] = BM * 128 + (wn * 64 + t2 * 32 + r) * 128 + ch;
    }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  u32x4 ra[AI], rb[BI];
  int lt = 0, lkc = 0;                                  // (tap, k-chunk) of the next step to load
  while (lt < p.ntaps && !((tapmask >> lt) & 1u)) ++lt;

  auto gload = [&]() {
    const int dy = p.offy[lt], dx = p.offx[lt];
    const int dpix = dy * p.Wi + dx;
    const int koff = lkc * KE + c * EPC;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int iy = ys[i] + dy, ix = xs[i] + dx;
      const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) v = ld16(xg + ((long)(pix[i] + dpix) * p.ldx + koff));
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i)
      rb[i] = ld16(wg + ((long)(lt * p.Cout + n0 + r0 + 32 * i) * p.Cin + koff));
    if (++lkc == nkc) {
      lkc = 0;
      do { ++lt; } while (lt < p.ntaps && !((tapmask >> lt) & 1u));
    }
  };
  auto lds_store = [&](int stage) {
    unsigned char* base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < AI; ++i) st16(base + a_wr[i], ra[i]);
#pragma unroll
    for (int i = 0; i < BI; ++i) st16(base + b_wr[i], rb[i]);
  };

  if (nsteps > 0) {
    gload();
    lds_store(0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const bool more = (s + 1 < nsteps);
    if (more) gload();
    const unsigned char* base = smem + (s & 1) * STAGE;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x4 a0 = ld16(base + a_rd[0][g]), a1 = ld16(base + a_rd[1][g]);
      const u32x4 b0 = ld16(base + b_rd[0][g]), b1 = ld16(base + b_rd[1][g]);
      Mma<T>::run(acc[0][0], a0, b0);
      Mma<T>::run(acc[0][1], a0, b1);
      Mma<T>::run(acc[1][0], a1, b0);
      Mma<T>::run(acc[1][1], a1, b1);
    }
    if (more) lds_store((s + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: straight from the accumulators -----------------------------------------------
  // acc[mi][ni][reg]: pixel row = (reg&3) + 8*(reg>>2) + 4*h, channel = n0 + wn*64 + 2*r + ni
  const int col = n0 + wn * 64 + 2 * r;
  T* __restrict__ og = reinterpret_cast<T*>(p.out);
  float k0a = 0.f, k0b = 0.f, k1a = 0.f, k1b = 0.f;     // per-channel constants
  int shuf_ab = 0, shuf_co = col;
  if (p.mode == MODE_AFFINE_RELU) {
    k0a = p.scale[col]; k0b = p.scale[col + 1];
    k1a = p.shift[col]; k1b = p.shift[col + 1];
  } else if (p.mode == MODE_SHUFFLE) {
    shuf_ab = col / p.shuf_c;
    shuf_co = col - shuf_ab * p.shuf_c;
    if (p.bias) { k1a = p.bias[shuf_co]; k1b = p.bias[shuf_co + 1]; }
  } else if (p.bias) {
    k1a = p.bias[col]; k1b = p.bias[col + 1];
  }
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = m0 + wm * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (m >= p.M) continue;
      float v0 = acc[mi][0][reg], v1 = acc[mi][1][reg];
      if (p.mode == MODE_AFFINE_RELU) {
        v0 = fmaxf(fmaf(v0, k0a, k1a), 0.f);
        v1 = fmaxf(fmaf(v1, k0b, k1b), 0.f);
        store_pair<T>(og + (long)m * p.ldo + col, v0, v1);
      } else if (p.mode == MODE_SHUFFLE) {
        v0 += k1a; v1 += k1b;
        const int n = m / HoWo, rem = m - n * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const long dst = ((long)(n * 2 * p.Ho + 2 * oy + (shuf_ab >> 1)) * (2 * p.Wo) + 2 * ox + (shuf_ab & 1));
        store_pair<T>(og + dst * p.ldo + shuf_co, v0, v1);
      } else {
        v0 += k1a; v1 += k1b;
        store_pair<T>(og + (long)m * p.ldo + col, v0, v1);
        if (p.mode == MODE_STATS) {
          // statistics of the values as stored (what the normalisation pass will read back)
          const float t0 = round_through<T>(v0), t1 = round_through<T>(v1);
          s0 += t0; q0 = fmaf(t0, t0, q0);
          s1 += t1; q1 = fmaf(t1, t1, q1);
        }
      }
    }
  }
  if (p.mode == MODE_STATS) {
    // lanes l and l+32 hold the same channels; then reduce over the WM waves that share columns.
    s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64);
    s1 += __shfl_xor(s1, 32, 64); q1 += __shfl_xor(q1, 32, 64);
    float* red = reinterpret_cast<float*>(smem);          // [wave][4][32]
    if (h == 0) {
      red[(wave * 4 + 0) * 32 + r] = s0;
      red[(wave * 4 + 1) * 32 + r] = q0;
      red[(wave * 4 + 2) * 32 + r] = s1;
      red[(wave * 4 + 3) * 32 + r] = q1;
    }
    __syncthreads();
    if (tid < BN) {
      const int wn2 = tid >> 6, c2 = tid & 63, r2 = c2 >> 1, e = c2 & 1;
      float su = 0.f, sq = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < WM; ++w2) {
        su += red[((w2 * WN + wn2) * 4 + e * 2 + 0) * 32 + r2];
        sq += red[((w2 * WN + wn2) * 4 + e * 2 + 1) * 32 + r2];
      }
      p.stats[((long)mblk * 2 + 0) * p.Cout + n0 + tid] = su;
      p.stats[((long)mblk * 2 + 1) * p.Cout + n0 + tid] = sq;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// defined in igemm_dma.hip (second-generation kernel, used whenever the tensors are < 2 GiB)
bool igemm_dma_supported(const IgemmParams& p, int dtype);
int launch_igemm_dma(IgemmParams& p, int dtype, hipStream_t stream);
// defined in igemm_halo.hip (LDS-staged input patch shared by all 9 taps; narrow layers, d <= 2)
bool igemm_halo_supported(const IgemmParams& p, int dtype);
int launch_igemm_halo(IgemmParams& p, int dtype, hipStream_t stream);

// Every configuration that can be asked for BatchNorm statistics uses 256-pixel M blocks, so the
// number of partial-statistics rows is a function of the pixel count only.
int igemm_mblocks(long M, int Cout) { (void)Cout; return ceil_div(M, 256); }

// UNETDC_IGEMM=legacy: first-generation register-staged kernel; =dma: per-tap LDS-DMA kernel only
// (no halo-patch kernel).  Default: best kernel per layer.  For A/B measurements in one binary.
static int igemm_choice() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("UNETDC_IGEMM");
    v = (e && e[0] == 'l') ? 1 : ((e && e[0] == 'd') ? 2 : 0);
  }
  return v;
}
static bool use_legacy() { return igemm_choice() == 1; }

template <typename T, int WM, int WN>
static int launch_cfg(IgemmParams& p, hipStream_t stream) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int LDS = 2 * (BM + BN) * 128;
  if (const int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&igemm_conv_kernel<T, WM, WN>), LDS, "igemm_conv_kernel")) return rc_;
  p.mblocks = ceil_div(p.M, BM);
  p.nblocks = p.Cout / BN;
  const long nwg = (long)p.mblocks * p.nblocks;
  hipLaunchKernelGGL((igemm_conv_kernel<T, WM, WN>), dim3((unsigned)nwg), dim3(256), LDS, stream, p);
  char nm[96];
  snprintf(nm, sizeof(nm), "igemm_conv_kernel<%s, %d, %d>", sizeof(T) == 2 ? "__bf16" : "float", WM, WN);
  note_kernel(nm);
  return check_launch("igemm_conv_kernel");
}

int launch_igemm(IgemmParams& p, int dtype, hipStream_t stream) {
  const int esz = dtype == UNETDC_BF16 ? 2 : 4;
  const int ke = 128 / esz;
  UNETDC_REQUIRE(dtype == UNETDC_F32 || dtype == UNETDC_BF16, "igemm: bad dtype %d", dtype);
  UNETDC_REQUIRE(p.x && p.w && p.out, "igemm: null tensor pointer");
  UNETDC_REQUIRE(p.M > 0 && p.Cin > 0 && p.Cout > 0, "igemm: empty problem");
  UNETDC_REQUIRE(p.Cin % ke == 0, "igemm: Cin=%d must be a multiple of %d for this dtype", p.Cin, ke);
  UNETDC_REQUIRE(p.Cout % 64 == 0, "igemm: Cout=%d must be a multiple of 64", p.Cout);
  UNETDC_REQUIRE(p.ldx % (16 / esz) == 0 && p.ldo % (16 / esz) == 0, "igemm: ld not 16-byte aligned");
  UNETDC_REQUIRE(((uintptr_t)p.x % 16 == 0) && ((uintptr_t)p.w % 16 == 0) && ((uintptr_t)p.out % 16 == 0),
                 "igemm: pointers must be 16-byte aligned");
  UNETDC_REQUIRE(p.ntaps >= 1 && p.ntaps <= 9, "igemm: ntaps out of range");
  UNETDC_REQUIRE((long)p.M < (1L << 31) - 512, "igemm: too many pixels");
  if (p.mode == MODE_SHUFFLE) UNETDC_REQUIRE(p.shuf_c % 64 == 0, "igemm: convT channels must be a multiple of 64");
  if (p.mode == MODE_STATS) UNETDC_REQUIRE(p.stats != nullptr, "igemm: stats buffer missing");
  if (p.mode == MODE_AFFINE_RELU) UNETDC_REQUIRE(p.scale && p.shift, "igemm: scale/shift missing");
  if (p.mode == MODE_BNBWD)
    UNETDC_REQUIRE(p.stats && p.bn_y && p.scale && p.shift && p.bn_mean && p.bn_rstd, "igemm: BN-backward inputs missing");
  {
    const long howo = (long)p.Ho * p.Wo;
    const bool p2 = (p.Wo & (p.Wo - 1)) == 0 && (howo & (howo - 1)) == 0;
    p.wo_shift = p2 ? __builtin_ctz((unsigned)p.Wo) : -1;
    p.howo_shift = p2 ? __builtin_ctzl((unsigned long)howo) : -1;
  }
  if (p.in_scale) {
    // input normalisation on load exists in the lattice kernel ONLY: no A/B switch may route this call to a kernel that would
    // read the raw tensor as if it were the activation
    if (!igemm_lattice_bnin_supported(p, dtype)) {
      set_error("igemm: input normalisation asked for a shape / configuration the lattice kernel does not take");
      return UNETDC_EUNSUPPORTED;
    }
    return launch_igemm_lattice(p, stream);
  }
  if (igemm_choice() == 0 && igemm_lattice_supported(p, dtype)) return launch_igemm_lattice(p, stream);
  if (igemm_choice() == 0 && igemm_halo_supported(p, dtype)) return launch_igemm_halo(p, dtype, stream);
  if (!use_legacy() && igemm_dma_supported(p, dtype)) return launch_igemm_dma(p, dtype, stream);
  if (p.mode == MODE_BNBWD) return UNETDC_EUNSUPPORTED;      // first-generation kernel: caller reduces separately
  const bool wide = (p.Cout % 128 == 0) && p.mode != MODE_STATS;     // statistics rows assume BM = 256
  if (dtype == UNETDC_BF16)
    return wide ? launch_cfg<bf16_t, 2, 2>(p, stream) : launch_cfg<bf16_t, 4, 1>(p, stream);
  return wide ? launch_cfg<float, 2, 2>(p, stream) : launch_cfg<float, 4, 1>(p, stream);
}

}  // namespace unetdc
