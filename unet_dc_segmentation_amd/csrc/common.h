// Shared device/host helpers for the gfx950 (CDNA4, wave64) U-Net kernels.
// All activation tensors are NHWC ("pixel-major"): element (n, y, x, c) lives at
// ((n*H + y)*W + x) * ld + c, where ld >= C lets a tensor be a channel slice of a wider
// buffer (the decoder's zero-copy concat buffers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UNETDC_F32 0
#define UNETDC_BF16 1

#define UNETDC_OK 0
#define UNETDC_EINVAL (-1)
#define UNETDC_ELAUNCH (-2)
#define UNETDC_EWORKSPACE (-3)

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

namespace unetdc {

void set_error(const char* fmt, ...);
int check_launch(const char* what);
// hipFuncSetAttribute(fn, MaxDynamicSharedMemorySize, bytes) once per (device, kernel) -- the attribute belongs to the kernel
// as loaded on ONE device, so the memo is keyed by the current device (a process that drives two GPUs sets it on both) and
// raised again if a later call asks for more.  Returns UNETDC_OK or UNETDC_ELAUNCH (message in unetdc_last_error()).
int ensure_dynamic_lds(const void* fn, int bytes, const char* name);
void note_kernel(const char* name);          // records the symbol of the MFMA kernel just launched

#define UNETDC_REQUIRE(cond, ...)                     \
  do {                                                \
    if (!(cond)) {                                    \
      unetdc::set_error(__VA_ARGS__);                 \
      return UNETDC_EINVAL;                           \
    }                                                 \
  } while (0)

// ---- element traits: 16-byte chunks -----------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int PER_CHUNK = 4;   // elements per 16-byte chunk
  static constexpr int BYTES = 4;
};
template <> struct Elem<bf16_t> {
  static constexpr int PER_CHUNK = 8;
  static constexpr int BYTES = 2;
};

// NOTE: never write __builtin_bit_cast(float, vec[i]) on an ext_vector element reached through a
// reference -- hipcc (ROCm 7.2) folds every i to element 0.  Go through a scalar first.
__device__ __forceinline__ float bits_f32(unsigned int u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned int f32_bits(float f) { return __builtin_bit_cast(unsigned int, f); }

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// round a float through the storage type (what a later kernel will read back)
template <typename T> __device__ __forceinline__ float round_through(float v) { return to_f32(from_f32<T>(v)); }

// two floats -> one word of two bf16 (low half = first), round to nearest even: ONE v_cvt_pk_bf16_f32.  (Converted one by
// one and joined with shifts, hipcc spends four instructions per pair: 16 instead of 4 per stored chunk.)
typedef __attribute__((ext_vector_type(2))) float cvt_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 cvt_bf16x2;
__device__ __forceinline__ unsigned int pack2_bf16(float lo, float hi) {
  const cvt_f32x2 v = {lo, hi};
  const cvt_bf16x2 b = __builtin_convertvector(v, cvt_bf16x2);
  return __builtin_bit_cast(unsigned int, b);
}

// A 16-byte chunk viewed as N elements of T, converted to/from fp32.
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int N = 4;
  __device__ static void unpack(const u32x4& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned int u = c[i];
      f[i] = bits_f32(u);
    }
  }
  __device__ static u32x4 pack(const float* f) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = f32_bits(f[i]);
    return c;
  }
};
template <> struct Chunk<bf16_t> {
  static constexpr int N = 8;
  __device__ static void unpack(const u32x4& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned int u = c[i];
      f[2 * i] = bits_f32(u << 16);
      f[2 * i + 1] = bits_f32(u & 0xffff0000u);
    }
  }
  __device__ static u32x4 pack(const float* f) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = pack2_bf16(f[2 * i], f[2 * i + 1]);
    return c;
  }
};

// element e (compile-time constant after unrolling) of a RAW 16-byte chunk, and its counterpart for building one: the
// same conversions as Chunk<T>::unpack / pack, one element at a time (bf16 elements are packed in pairs: the even one is
// parked in `prev` until the odd one arrives)
template <typename T> __device__ __forceinline__ float chunk_elem(const u32x4& c, int e);
template <> __device__ __forceinline__ float chunk_elem<float>(const u32x4& c, int e) {
  const unsigned int u = c[e];
  return bits_f32(u);
}
template <> __device__ __forceinline__ float chunk_elem<bf16_t>(const u32x4& c, int e) {
  const unsigned int u = c[e >> 1];
  return (e & 1) ? bits_f32(u & 0xffff0000u) : bits_f32(u << 16);
}
template <typename T> __device__ __forceinline__ void chunk_set(u32x4& c, int e, float v, float& prev);
template <> __device__ __forceinline__ void chunk_set<float>(u32x4& c, int e, float v, float& prev) {
  (void)prev;
  c[e] = f32_bits(v);
}
template <> __device__ __forceinline__ void chunk_set<bf16_t>(u32x4& c, int e, float v, float& prev) {
  if (e & 1) {
    c[e >> 1] = pack2_bf16(prev, v);
  } else {
    prev = v;
  }
}

// two adjacent elements of T as one memory transaction, kept raw so that several can be in flight
template <typename T> struct PairRaw;
template <> struct PairRaw<bf16_t> {
  typedef unsigned int raw_t;
  __device__ static __forceinline__ raw_t load(const bf16_t* p) { return *reinterpret_cast<const unsigned int*>(p); }
  __device__ static __forceinline__ void unpack(raw_t r, float& a, float& b) { a = bits_f32(r << 16); b = bits_f32(r & 0xffff0000u); }
};
template <> struct PairRaw<float> {
  typedef float2 raw_t;
  __device__ static __forceinline__ raw_t load(const float* p) { return *reinterpret_cast<const float2*>(p); }
  __device__ static __forceinline__ void unpack(raw_t r, float& a, float& b) { a = r.x; b = r.y; }
};

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }

// XCD-aware remap of a linear workgroup id: blocks b and b+8 share an XCD (observed round-robin
// placement; speed only, never correctness), so give each XCD one contiguous chunk of the tile
// list to keep neighbouring tiles in one L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + local;
}

// wave-level sum over the 64 lanes
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace unetdc
