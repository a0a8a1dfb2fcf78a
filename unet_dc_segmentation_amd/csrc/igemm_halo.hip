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

template <typename T> __device__ __forceinline__ void store_pair_h(T* dst, float v0, float v1);
template <> __device__ __forceinline__ void store_pair_h<float>(float* dst, float v0, float v1) {
  *reinterpret_cast<float2*>(dst) = make_float2(v0, v1);
}
template <> __device__ __forceinline__ void store_pair_h<bf16_t>(bf16_t* dst, float v0, float v1) {
  bf16_t lo = (bf16_t)v0, hi = (bf16_t)v1;
  *reinterpret_cast<unsigned int*>(dst) = (unsigned int)__builtin_bit_cast(unsigned short, lo) |
                                          ((unsigned int)__builtin_bit_cast(unsigned short, hi) << 16);
}

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned HOOB = 0x80000000u;
constexpr int TH = 8, TW = 32;                  // output tile in pixels
constexpr int MAXPJ = 14;                       // patch DMA instructions per wave (d <= 2, 4 waves: 54/4)

// WN = 1: 4 waves (4 x 1), BN = 64;   WN = 2: 8 waves (4 x 2), BN = 128.  Wave tile 64 x 64.
template <typename T, int WN>
__global__ __launch_bounds__(512, 2) void igemm_halo_kernel(const IgemmParams p, int d, int npatch_bufs) {
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
  const int tiles_x = p.Wo / TW, tiles_y = p.Ho / TH;
  const int img = mtile / (tiles_x * tiles_y);
  const int trem = mtile - img * tiles_x * tiles_y;
  const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;

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
    const int py = pr / PW, px = pr - py * PW;
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

  // ---- epilogue -----------------------------------------------------------------------------------
  const int col = n0 + wn * 64 + 2 * r;
  T* __restrict__ og = reinterpret_cast<T*>(p.out);
  float k0a = 0.f, k0b = 0.f, k1a = 0.f, k1b = 0.f;
  if (p.mode == MODE_AFFINE_RELU) {
    k0a = p.scale[col]; k0b = p.scale[col + 1];
    k1a = p.shift[col]; k1b = p.shift[col + 1];
  } else if (p.mode == MODE_BNBWD) {
    k0a = p.scale[col]; k0b = p.scale[col + 1];
    k1a = p.shift[col]; k1b = p.shift[col + 1];
  } else if (p.bias) {
    k1a = p.bias[col]; k1b = p.bias[col + 1];
  }
  float mua = 0.f, mub = 0.f, rsa = 0.f, rsb = 0.f;
  if (p.mode == MODE_BNBWD) {
    mua = p.bn_mean[col]; mub = p.bn_mean[col + 1];
    rsa = p.bn_rstd[col]; rsb = p.bn_rstd[col + 1];
  }
  const T* __restrict__ yg = reinterpret_cast<const T*>(p.bn_y);
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const long rowpix = ((long)img * p.Ho + y0 + wm * 2 + mi) * p.Wo + x0;
    typename PairRaw<T>::raw_t yraw[16];
    if (p.mode == MODE_BNBWD) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        yraw[reg] = PairRaw<T>::load(yg + (rowpix + (reg & 3) + 8 * (reg >> 2) + 4 * h) * p.bn_ldy + col);
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int tx = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      float v0 = acc[mi][0][reg], v1 = acc[mi][1][reg];
      T* dst = og + (rowpix + tx) * p.ldo + col;
      if (p.mode == MODE_AFFINE_RELU) {
        v0 = fmaxf(fmaf(v0, k0a, k1a), 0.f);
        v1 = fmaxf(fmaf(v1, k0b, k1b), 0.f);
        store_pair_h<T>(dst, v0, v1);
      } else if (p.mode == MODE_BNBWD) {
        store_pair_h<T>(dst, v0, v1);
        float y0, y1;
        PairRaw<T>::unpack(yraw[reg], y0, y1);
        const float g0 = fmaf(y0, k0a, k1a) > 0.f ? round_through<T>(v0) : 0.f;
        const float g1 = fmaf(y1, k0b, k1b) > 0.f ? round_through<T>(v1) : 0.f;
        s0 += g0; q0 = fmaf(g0, (y0 - mua) * rsa, q0);
        s1 += g1; q1 = fmaf(g1, (y1 - mub) * rsb, q1);
      } else {
        v0 += k1a; v1 += k1b;
        store_pair_h<T>(dst, v0, v1);
        if (p.mode == MODE_STATS) {
          const float t0 = round_through<T>(v0), t1 = round_through<T>(v1);
          s0 += t0; q0 = fmaf(t0, t0, q0);
          s1 += t1; q1 = fmaf(t1, t1, q1);
        }
      }
    }
  }
  if (p.mode == MODE_STATS || p.mode == MODE_BNBWD) {
    const int nrow = (p.mode == MODE_BNBWD) ? 3 : 2;
    s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64);
    s1 += __shfl_xor(s1, 32, 64); q1 += __shfl_xor(q1, 32, 64);
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);             // [wave][4][32]
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
      for (int w2 = 0; w2 < 4; ++w2) {
        su += red[((w2 * WN + wn2) * 4 + e * 2 + 0) * 32 + r2];
        sq += red[((w2 * WN + wn2) * 4 + e * 2 + 1) * 32 + r2];
      }
      p.stats[((long)mtile * nrow + 0) * p.Cout + n0 + tid] = su;
      p.stats[((long)mtile * nrow + 1) * p.Cout + n0 + tid] = sq;
      if (nrow == 3) p.stats[((long)mtile * 3 + 2) * p.Cout + n0 + tid] = 0.f;
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
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
  return xbytes < (1L << 31);
}

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
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_halo_kernel<T, WN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(igemm_halo_kernel) failed: %s", hipGetErrorString(e));
      return UNETDC_ELAUNCH;
    }
    attr_done = true;
  }
  p.mblocks = (int)((long)p.M / 256);
  p.nblocks = p.Cout / (64 * WN);
  const long nwg = (long)p.mblocks * p.nblocks;
  hipLaunchKernelGGL((igemm_halo_kernel<T, WN>), dim3((unsigned)nwg), dim3(256 * WN), lds, stream, p, d, nbuf);
  char nm[96];
  snprintf(nm, sizeof(nm), "igemm_halo_kernel<%s, %d>", sizeof(T) == 2 ? "__bf16" : "float", WN);
  note_kernel(nm);
  return check_launch("igemm_halo_kernel");
}

int launch_igemm_halo(IgemmParams& p, int dtype, hipStream_t stream) {
  const int d = p.offy[8];
  if (p.Cout == 128)
    return dtype == UNETDC_BF16 ? launch_halo_cfg<bf16_t, 2>(p, d, stream) : launch_halo_cfg<float, 2>(p, d, stream);
  return dtype == UNETDC_BF16 ? launch_halo_cfg<bf16_t, 1>(p, d, stream) : launch_halo_cfg<float, 1>(p, d, stream);
}

}  // namespace unetdc
