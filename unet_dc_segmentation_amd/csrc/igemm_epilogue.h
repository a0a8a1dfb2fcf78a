// Accumulator -> global-memory epilogues shared by the LDS-DMA implicit-GEMM kernels
// (igemm_dma.hip, igemm_halo.hip).
//
// What the first version cost (rocprofv3 SQ counters on ConvTranspose2d forward, 64 MFMAs per wave):
// ~2400 VALU + ~1400 SALU instructions per wave.  The epilogue loop was compiled with the mode
// dispatch INSIDE the 64-way unrolled store loop, a 64-bit multiply per address, and -- in the fused
// BatchNorm-backward mode -- an `s_waitcnt vmcnt(0)` in front of every store (on gfx9 stores count
// in vmcnt, so every store waited for the previous one).  Here
//   * the mode is a template argument (the kernels `switch` once, outside the loops);
//   * stores are `buffer_store_dword(x2)` with a per-lane 32-bit byte offset computed ONCE per MFMA
//     tile and the row inside the tile supplied as the instruction's scalar offset (row * row_bytes,
//     wave-uniform), so a store costs the value arithmetic and nothing else;
//   * bf16 pairs are converted with one v_cvt_pk_bf16_f32; the rounded values the statistics need
//     are unpacked from the packed word (two bit operations);
//   * the saved conv outputs of the BatchNorm-backward fusion are fetched with buffer loads, 16 in
//     flight per tile, in straight-line code so the compiler can count vmcnt instead of draining.
//
// BUILD NOTE (-fno-slp-vectorize, see build.py): with SLP vectorisation the channel-pair arithmetic of the
// BatchNorm-backward statistics becomes packed fp32 (v_pk_fma_f32 / v_pk_add_f32), and on MI355X that build was
// NOT run-to-run deterministic: about one workgroup in 8192 dropped (or wrongly kept) the contribution of a
// single pixel, always in the ODD channel of a pair and always in lanes 48-63.  The root cause is not identified
// (an isolated replay of the packed-FMA -> compare -> select sequence, tools/probes/pk_hazard.hip, is clean); the
// scalar build (0 v_pk_* instructions) was bit-identical over 36 runs of tools/det_op.py and ever since, and is
// as fast.  tests/test_gpu_ops.py guards it at the full layer sizes.
#pragma once
#include "kernels.h"

namespace unetdc {

#if defined(__HIP_DEVICE_COMPILE__)

typedef __attribute__((ext_vector_type(2))) float epi_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 epi_bf16x2;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t whole_buffer(const void* base) {
  // every true offset is < 4 GiB (checked on the host), so the range check never fires
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xFFFFFFFFu, 0x00020000);
}

// two adjacent channels of one pixel as a single memory transaction
template <typename T> struct OutPair;
template <> struct OutPair<bf16_t> {
  typedef unsigned int packed_t;
  static __device__ __forceinline__ packed_t pack(float v0, float v1) {
    const epi_f32x2 v = {v0, v1};
    const epi_bf16x2 b = __builtin_convertvector(v, epi_bf16x2);
    return __builtin_bit_cast(unsigned int, b);
  }
  static __device__ __forceinline__ void rounded(packed_t pk, float, float, float& t0, float& t1) {
    t0 = bits_f32(pk << 16);
    t1 = bits_f32(pk & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, packed_t pk) {
    __builtin_amdgcn_raw_buffer_store_b32(pk, r, voff, soff, 0);
  }
  static __device__ __forceinline__ packed_t load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
  }
  static __device__ __forceinline__ void unpack(packed_t pk, float& a, float& b) { rounded(pk, 0.f, 0.f, a, b); }
};
template <> struct OutPair<float> {
  typedef u32x2 packed_t;
  static __device__ __forceinline__ packed_t pack(float v0, float v1) {
    packed_t pk;
    pk[0] = f32_bits(v0);
    pk[1] = f32_bits(v1);
    return pk;
  }
  static __device__ __forceinline__ void rounded(packed_t, float v0, float v1, float& t0, float& t1) { t0 = v0; t1 = v1; }
  static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, packed_t pk) {
    __builtin_amdgcn_raw_buffer_store_b64(pk, r, voff, soff, 0);
  }
  static __device__ __forceinline__ packed_t load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  }
  static __device__ __forceinline__ void unpack(packed_t pk, float& a, float& b) {
    const unsigned int u0 = pk[0], u1 = pk[1];
    a = bits_f32(u0);
    b = bits_f32(u1);
  }
};

// One wave's TM x 2 accumulator tiles (32 x 32 each; the two tiles of a pair hold the even / odd channel of
// a channel pair, see the de-interleaved weight rows) -> global memory.
//   voff[mi]   byte offset in `out` of (tile row 4*h, this lane's channel pair) for tile mi
//   row_bytes  bytes between consecutive tile rows (uniform): ld * sizeof(T), twice that for the pixel shuffle
//   yoff/yrow_bytes  the same for the saved conv output read by MODE_BNBWD
//   tile_ok[mi] uniform: tile lies inside the tensor
//   ccol       channel index for the per-channel constants (bias / scale / shift / mean / rstd)
//   st[4]      running (sum, sumsq) or (S1, S2) of the even and the odd channel
template <typename T, int MODE, int TM>
__device__ __forceinline__ void epilogue_tiles(const IgemmParams& p, f32x16 (&acc)[TM][2], const bool (&tile_ok)[TM],
                                               const unsigned (&voff)[TM], unsigned row_bytes,
                                               const unsigned (&yoff)[TM], unsigned yrow_bytes, int ccol,
                                               float (&st)[4]) {
  typedef OutPair<T> OP;
  const __amdgpu_buffer_rsrc_t orr = whole_buffer(p.out);
  const __amdgpu_buffer_rsrc_t yrr = whole_buffer(MODE == MODE_BNBWD ? p.bn_y : p.out);
  float k0a = 0.f, k0b = 0.f, k1a = 0.f, k1b = 0.f, mua = 0.f, mub = 0.f, rsa = 0.f, rsb = 0.f;
  if (MODE == MODE_AFFINE_RELU || MODE == MODE_BNBWD) {
    k0a = p.scale[ccol]; k0b = p.scale[ccol + 1];
    k1a = p.shift[ccol]; k1b = p.shift[ccol + 1];
  } else if (p.bias) {
    k1a = p.bias[ccol]; k1b = p.bias[ccol + 1];
  }
  if (MODE == MODE_BNBWD) {
    mua = p.bn_mean[ccol]; mub = p.bn_mean[ccol + 1];
    rsa = p.bn_rstd[ccol]; rsb = p.bn_rstd[ccol + 1];
  }
  float s0 = st[0], q0 = st[1], s1 = st[2], q1 = st[3];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    if (!tile_ok[mi]) continue;
    typename OP::packed_t yraw[16];
    if (MODE == MODE_BNBWD) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        yraw[reg] = OP::load(yrr, yoff[mi], (unsigned)((reg & 3) + 8 * (reg >> 2)) * yrow_bytes);
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const unsigned soff = (unsigned)((reg & 3) + 8 * (reg >> 2)) * row_bytes;
      float v0 = acc[mi][0][reg], v1 = acc[mi][1][reg];
      if (MODE == MODE_AFFINE_RELU) {
        v0 = fmaxf(fmaf(v0, k0a, k1a), 0.f);
        v1 = fmaxf(fmaf(v1, k0b, k1b), 0.f);
        OP::store(orr, voff[mi], soff, OP::pack(v0, v1));
      } else if (MODE == MODE_BNBWD) {
        const typename OP::packed_t pk = OP::pack(v0, v1);
        OP::store(orr, voff[mi], soff, pk);
        float t0, t1, y0, y1;
        OP::rounded(pk, v0, v1, t0, t1);
        OP::unpack(yraw[reg], y0, y1);
        const float g0 = fmaf(y0, k0a, k1a) > 0.f ? t0 : 0.f;
        const float g1 = fmaf(y1, k0b, k1b) > 0.f ? t1 : 0.f;
        s0 += g0; q0 = fmaf(g0, (y0 - mua) * rsa, q0);
        s1 += g1; q1 = fmaf(g1, (y1 - mub) * rsb, q1);
      } else {
        v0 += k1a; v1 += k1b;
        const typename OP::packed_t pk = OP::pack(v0, v1);
        OP::store(orr, voff[mi], soff, pk);
        if (MODE == MODE_STATS) {
          float t0, t1;
          OP::rounded(pk, v0, v1, t0, t1);
          s0 += t0; q0 = fmaf(t0, t0, q0);
          s1 += t1; q1 = fmaf(t1, t1, q1);
        }
      }
    }
  }
  st[0] = s0; st[1] = q0; st[2] = s1; st[3] = q1;
}

// Block-level reduction of the per-wave statistics into one partial row per M-block:
// red[wave][4][32] in LDS (stage buffers are free by now), then one thread per channel.
template <int WM, int WN>
__device__ __forceinline__ void write_stat_rows(const IgemmParams& p, unsigned char* smem, float (&st)[4], int mrow,
                                                int n0, int tid, int wave, int r, int h) {
  constexpr int BN = WN * 64;
  const int nrow = (p.mode == MODE_BNBWD) ? 3 : 2;          // BN-backward partials carry a third (zero) row
  float s0 = st[0], q0 = st[1], s1 = st[2], q1 = st[3];
  s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64);
  s1 += __shfl_xor(s1, 32, 64); q1 += __shfl_xor(q1, 32, 64);
  __syncthreads();                                         // all waves are done with the stage buffers
  float* red = reinterpret_cast<float*>(smem);
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
    p.stats[((long)mrow * nrow + 0) * p.Cout + n0 + tid] = su;
    p.stats[((long)mrow * nrow + 1) * p.Cout + n0 + tid] = sq;
    if (nrow == 3) p.stats[((long)mrow * 3 + 2) * p.Cout + n0 + tid] = 0.f;
  }
}

#endif  // __HIP_DEVICE_COMPILE__

}  // namespace unetdc
