// LDS fragment readers of the weight-gradient GEMM (shared by wgrad.hip and wgrad_dma.hip).
// The LDS image of an operand is [pixel row][channels] (RB bytes per row) exactly as it arrives
// from HBM; TW = tile width in units of 64 channels.
//   * bf16: the MFMA wants 8 consecutive pixels (K) per lane for one channel, so fragments are read
//     with ds_read_b64_tr_b16 (4 pixel rows x 16 channels per 16-lane group, delivered transposed).
//     The 64-byte units of a row are XOR-swizzled with the pixel row so the 4 rows of one read land
//     in 4 distinct bank windows: RB = 128: unit ^= (row>>1)&1; RB >= 256: unit ^= row&3.
//   * fp32: v_mfma_f32_32x32x2_f32 takes one scalar per lane: plain ds_read_b32 of 32 consecutive
//     channels of one pixel row per half-wave (conflict-free, no swizzle).
#pragma once
#include "kernels.h"

namespace unetdc {

template <typename T, int TW> struct Frag;

template <int TW> struct Frag<bf16_t, TW> {
  static constexpr int RB = TW * 128;                     // bytes per pixel row in LDS
  __device__ static __forceinline__ int swz(int row) { return (TW == 1) ? ((row >> 1) & 1) : (row & 3); }
  // byte offset (within an operand's stage image) of 16-byte chunk `ch` of pixel row `row`
  __device__ static __forceinline__ int wr_off(int row, int ch) {
    const int byte = ch * 16, unit = byte >> 6, within = byte & 63;
    return row * RB + ((unit ^ swz(row)) << 6) + within;
  }
  // logical 16-byte chunk that must be fetched into physical chunk `pc` of pixel row `row`
  __device__ static __forceinline__ int src_chunk(int row, int pc) { return (((pc >> 2) ^ swz(row)) << 2) | (pc & 3); }
  // per-lane address of one transposed read: 32 channels from `cbase`, pixel rows krow0 + [0,16)
  __device__ static __forceinline__ int rd_off(int lane, int cbase, int krow0, int jj) {
    const int i = lane & 15, G = (lane >> 4) & 3;
    const int q = i >> 2, pp = i & 3, hi = G & 1, h = G >> 1;
    const int krow = krow0 + 8 * h + 4 * jj + q;
    const int byte = (cbase + 16 * hi + 4 * pp) * 2, unit = byte >> 6, within = byte & 63;
    return krow * RB + ((unit ^ swz(krow)) << 6) + within;
  }
  __device__ static __forceinline__ bf16x8 frag(const unsigned char* s, int lane, int cbase, int krow0) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(s + rd_off(lane, cbase, krow0, 0)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(s + rd_off(lane, cbase, krow0, 1)));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  }
  // consume 16 pixels: acc[i][j] += A(32 ch from ca+32i)^T . B(32 ch from cb+32j)
  template <int NI>
  __device__ static __forceinline__ void mma16n(f32x16 (&acc)[NI][2], const unsigned char* sa, const unsigned char* sb,
                                                int lane, int ca, int cb, int krow0) {
    bf16x8 fa[NI], fb[2];
#pragma unroll
    for (int t = 0; t < NI; ++t) fa[t] = frag(sa, lane, ca + 32 * t, krow0);
#pragma unroll
    for (int t = 0; t < 2; ++t) fb[t] = frag(sb, lane, cb + 32 * t, krow0);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
  }
  __device__ static __forceinline__ void mma16(f32x16 (&acc)[2][2], const unsigned char* sa, const unsigned char* sb,
                                               int lane, int ca, int cb, int krow0) {
    mma16n<2>(acc, sa, sb, lane, ca, cb, krow0);
  }
};

// 16x16x32 operands from a [pixel][64 channels] bf16 image (128-byte rows): a fragment = 16 channels x 32 pixel rows, lane l
// holds channel cbase + (l & 15), pixels krow0 + 8 (l >> 4) + j in element j -- the A and the B map of
// v_mfma_f32_16x16x32_bf16 alike.  Two ds_read_b64_tr_b16 per fragment: block jj of lane group G = pixel rows
// krow0 + 8 G + 4 jj + (0..3), 16 channels.  The two groups of a 32-lane half read rows 8 apart in the SAME columns, and the
// four rows of a block are consecutive: the image is swizzled in 16-byte chunks (through the DMA source addresses),
//   physical chunk = logical chunk ^ (((row >> 1) & 1) << 2) ^ (((row >> 3) & 1) << 1)
// so that the four rows of a block sit in four different 64-byte windows of the 256-byte bank row and the blocks of the two
// groups in different 32-byte halves of them -- conflict-free at every row offset (tap shifts move krow0 by kx * d).
struct Frag16 {
  static constexpr int RB = 128;
  __device__ static __forceinline__ int key(int row) { return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1); }
  // logical 16-byte chunk that must be fetched into physical chunk `pc` of pixel row `row`
  __device__ static __forceinline__ int src_chunk(int row, int pc) { return pc ^ key(row); }
  __device__ static __forceinline__ int rd_off(int lane, int cbase, int krow0, int jj) {
    const int i = lane & 15, G = lane >> 4, q = i >> 2, pp = i & 3;
    const int krow = krow0 + 8 * G + 4 * jj + q;
    const int byte = (cbase + 4 * pp) * 2;
    return krow * RB + (((byte >> 4) ^ key(krow)) << 4) + (byte & 15);
  }
  __device__ static __forceinline__ bf16x8 frag_at(const unsigned char* s, int off0, int off1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + off1));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  }
};

template <int TW> struct Frag<float, TW> {
  static constexpr int RB = TW * 256;
  __device__ static __forceinline__ int wr_off(int row, int ch) { return row * RB + ch * 16; }
  __device__ static __forceinline__ int src_chunk(int row, int pc) { (void)row; return pc; }
  template <int NI>
  __device__ static __forceinline__ void mma16n(f32x16 (&acc)[NI][2], const unsigned char* sa, const unsigned char* sb,
                                                int lane, int ca, int cb, int krow0) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kp = 0; kp < 8; ++kp) {
      const int krow = krow0 + 2 * kp + h;
      float fa[NI], fb[2];
#pragma unroll
      for (int t = 0; t < NI; ++t) fa[t] = *reinterpret_cast<const float*>(sa + krow * RB + (ca + 32 * t + r) * 4);
#pragma unroll
      for (int t = 0; t < 2; ++t) fb[t] = *reinterpret_cast<const float*>(sb + krow * RB + (cb + 32 * t + r) * 4);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }
  __device__ static __forceinline__ void mma16(f32x16 (&acc)[2][2], const unsigned char* sa, const unsigned char* sb,
                                               int lane, int ca, int cb, int krow0) {
    mma16n<2>(acc, sa, sb, lane, ca, cb, krow0);
  }
};

}  // namespace unetdc
