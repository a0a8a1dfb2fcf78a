// Epilogue of the 16x16x32-MFMA implicit-GEMM kernels (igemm_dma16.hip, igemm_halo.hip): a wave owns TMT M-tiles of
// 16 pixels x 4 N-tiles of 16 channels; lane l (c = l & 15, rb = l >> 4) holds acc[i][j][v] = out[pixel 16i + 4rb + v]
// [channel 4c + j], packs the four channels into one 8-byte store, and 16 lanes write one 128-byte pixel row.
// Same modes and partial-row layout as igemm_epilogue.h.
#pragma once
#include "igemm_epilogue.h"

namespace unetdc {

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ u32x2 pack4_bf16(float a, float b, float c, float d) {
  u32x2 r;
  r[0] = OutPair<bf16_t>::pack(a, b);
  r[1] = OutPair<bf16_t>::pack(c, d);
  return r;
}
__device__ __forceinline__ void unpack4_bf16(const u32x2& r, float (&t)[4]) {
  const unsigned int u0 = r[0], u1 = r[1];
  t[0] = bits_f32(u0 << 16); t[1] = bits_f32(u0 & 0xffff0000u);
  t[2] = bits_f32(u1 << 16); t[3] = bits_f32(u1 & 0xffff0000u);
}

// per-channel constants of the four channels a lane owns (bias | scale, shift | + mean, rstd)
struct Epi16Consts {
  float k0[4], k1[4], mu[4], rs[4];
};
template <int MODE>
__device__ __forceinline__ Epi16Consts epi16_consts(const IgemmParams& p, int ccol) {
  Epi16Consts c;
#pragma unroll
  for (int k = 0; k < 4; ++k) { c.k0[k] = 0.f; c.k1[k] = 0.f; c.mu[k] = 0.f; c.rs[k] = 0.f; }
  if (MODE == MODE_AFFINE_RELU || MODE == MODE_BNBWD) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { c.k0[k] = p.scale[ccol + k]; c.k1[k] = p.shift[ccol + k]; }
  } else if (p.bias) {
#pragma unroll
    for (int k = 0; k < 4; ++k) c.k1[k] = p.bias[ccol + k];
  }
  if (MODE == MODE_BNBWD) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { c.mu[k] = p.bn_mean[ccol + k]; c.rs[k] = p.bn_rstd[ccol + k]; }
  }
  return c;
}

// the saved conv outputs MODE_BNBWD reads, fetched ahead of the epilogue (persistent kernels issue this a few taps before
// the last MFMA of an item so that the HBM latency is not exposed)
template <int TMT>
__device__ __forceinline__ void epi16_prefetch_y(const IgemmParams& p, const unsigned (&yoff)[TMT], unsigned yrow_bytes,
                                                 u32x2 (&ypre)[TMT][4]) {
  const __amdgpu_buffer_rsrc_t yrr = whole_buffer(p.bn_y);
#pragma unroll
  for (int i = 0; i < TMT; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) ypre[i][v] = __builtin_amdgcn_raw_buffer_load_b64(yrr, yoff[i], (unsigned)v * yrow_bytes, 0);
}

// TMT M-tiles of 16 rows x 4 N-tiles of 16 channels per wave -> global memory (bf16); constants preloaded;
// YPRE: MODE_BNBWD takes the saved conv outputs from `ypre` (epi16_prefetch_y) instead of loading them here
template <int MODE, int TMT, bool YPRE = false>
__device__ __forceinline__ void epilogue16c(const IgemmParams& p, f32x4 (&acc)[TMT][4], const bool (&tile_ok)[TMT],
                                            const unsigned (&voff)[TMT], unsigned row_bytes, const unsigned (&yoff)[TMT],
                                            unsigned yrow_bytes, const Epi16Consts& kc, float (&s)[4], float (&q)[4],
                                            const u32x2 (*ypre)[4] = nullptr) {
  const __amdgpu_buffer_rsrc_t orr = whole_buffer(p.out);
  const __amdgpu_buffer_rsrc_t yrr = whole_buffer(MODE == MODE_BNBWD ? p.bn_y : p.out);
  const float (&k0)[4] = kc.k0, (&k1)[4] = kc.k1, (&mu)[4] = kc.mu, (&rs)[4] = kc.rs;
  constexpr int GRP = TMT < 4 ? TMT : 4;                  // tiles whose saved-output loads are in flight together
#pragma unroll
  for (int i0 = 0; i0 < TMT; i0 += GRP) {
    u32x2 yraw[GRP][4];
    if (MODE == MODE_BNBWD && YPRE) {
#pragma unroll
      for (int ii = 0; ii < GRP; ++ii)
#pragma unroll
        for (int v = 0; v < 4; ++v) yraw[ii][v] = ypre[i0 + ii][v];
    } else if (MODE == MODE_BNBWD) {
#pragma unroll
      for (int ii = 0; ii < GRP; ++ii)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          yraw[ii][v] = tile_ok[i0 + ii] ? __builtin_amdgcn_raw_buffer_load_b64(yrr, yoff[i0 + ii], (unsigned)v * yrow_bytes, 0)
                                         : u32x2{0u, 0u};
    }
#pragma unroll
    for (int ii = 0; ii < GRP; ++ii) {
      const int i = i0 + ii;
      if (!tile_ok[i]) continue;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = acc[i][k][v];
        if (MODE == MODE_AFFINE_RELU) {
#pragma unroll
          for (int k = 0; k < 4; ++k) x[k] = fmaxf(fmaf(x[k], k0[k], k1[k]), 0.f);
        } else if (MODE != MODE_BNBWD) {
#pragma unroll
          for (int k = 0; k < 4; ++k) x[k] += k1[k];
        }
        const u32x2 pk = pack4_bf16(x[0], x[1], x[2], x[3]);
        __builtin_amdgcn_raw_buffer_store_b64(pk, orr, voff[i], (unsigned)v * row_bytes, 0);
        if (MODE == MODE_STATS) {
          float t[4];
          unpack4_bf16(pk, t);
#pragma unroll
          for (int k = 0; k < 4; ++k) { s[k] += t[k]; q[k] = fmaf(t[k], t[k], q[k]); }
        } else if (MODE == MODE_BNBWD) {
          float t[4], y[4];
          unpack4_bf16(pk, t);
          unpack4_bf16(yraw[ii][v], y);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float g = fmaf(y[k], k0[k], k1[k]) > 0.f ? t[k] : 0.f;
            s[k] += g;
            q[k] = fmaf(g, (y[k] - mu[k]) * rs[k], q[k]);
          }
        }
      }
    }
  }
}

// same, loading the per-channel constants first (one call per workgroup in the non-persistent kernels)
template <int MODE, int TMT>
__device__ __forceinline__ void epilogue16(const IgemmParams& p, f32x4 (&acc)[TMT][4], const bool (&tile_ok)[TMT],
                                           const unsigned (&voff)[TMT], unsigned row_bytes, const unsigned (&yoff)[TMT],
                                           unsigned yrow_bytes, int ccol, float (&s)[4], float (&q)[4]) {
  const Epi16Consts kc = epi16_consts<MODE>(p, ccol);
  epilogue16c<MODE, TMT>(p, acc, tile_ok, voff, row_bytes, yoff, yrow_bytes, kc, s, q);
}

// per-channel statistics of a workgroup (WM x WN waves) -> one partial row; red = LDS scratch of NW*128 floats
template <int WM, int WN>
__device__ __forceinline__ void write_stat_rows16(const IgemmParams& p, unsigned char* smem, float (&s4)[4], float (&q4)[4],
                                                  int mrow, int n0, int tid, int wave, int c16, int rb) {
  constexpr int BN = WN * 64;
  const int nrow = (p.mode == MODE_BNBWD) ? 3 : 2;
#pragma unroll
  for (int k = 0; k < 4; ++k) {                             // the four row groups of a lane column
    s4[k] += __shfl_xor(s4[k], 16, 64); q4[k] += __shfl_xor(q4[k], 16, 64);
    s4[k] += __shfl_xor(s4[k], 32, 64); q4[k] += __shfl_xor(q4[k], 32, 64);
  }
  __syncthreads();                                         // stage buffers are free
  float* red = reinterpret_cast<float*>(smem);             // [wave][4 k][2][16 c]
  if (rb == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      red[((wave * 4 + k) * 2 + 0) * 16 + c16] = s4[k];
      red[((wave * 4 + k) * 2 + 1) * 16 + c16] = q4[k];
    }
  }
  __syncthreads();
  if (tid < BN) {
    const int wn2 = tid >> 6, cc = tid & 63, c2 = cc >> 2, k = cc & 3;
    float su = 0.f, sq = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < WM; ++w2) {
      su += red[(((w2 * WN + wn2) * 4 + k) * 2 + 0) * 16 + c2];
      sq += red[(((w2 * WN + wn2) * 4 + k) * 2 + 1) * 16 + c2];
    }
    p.stats[((long)mrow * nrow + 0) * p.Cout + n0 + tid] = su;
    p.stats[((long)mrow * nrow + 1) * p.Cout + n0 + tid] = sq;
    if (nrow == 3) p.stats[((long)mrow * 3 + 2) * p.Cout + n0 + tid] = 0.f;
  }
}

#endif  // __HIP_DEVICE_COMPILE__

}  // namespace unetdc
