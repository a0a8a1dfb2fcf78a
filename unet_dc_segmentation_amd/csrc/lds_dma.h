// LDS-DMA (`buffer_load_dwordx4 ... offen lds`) issued from INLINE ASM, and the waits that go with it.
//
// Why not the builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds): hipcc (ROCm 7.2) models it as a store to LDS that
// is pending on the VM counter, and SIInsertWaitcnts then protects every later LDS access it cannot disambiguate:
//   * `__syncthreads()` while a DMA is in flight compiles to `s_waitcnt vmcnt(0) ; s_barrier`;
//   * the first LDS read after a DMA issue gets an `s_waitcnt vmcnt(0)` in front of it (seen in the round-1 ring
//     kernel: wait(counted) ; barrier ; issue group s+2 ; **vmcnt(0)** ; ds_read_b64_tr_b16 ... -- the prefetch
//     distance the source asked for never existed in the binary).
// A DMA issued from an asm statement is invisible to that pass: its completion is OURS to count
// (`s_waitcnt vmcnt(N)`: loads, stores and DMAs retire in issue order, MI355X_MICROARCH.md), while the compiler keeps
// counting lgkmcnt for the plain LDS reads it does see.  Rules used by the kernels that include this header:
//   RAW  a staged buffer is read only after (1) every wave's counted vmcnt that covers its own pieces and (2) a
//        workgroup barrier behind those waits;
//   WAR  a buffer is re-filled only behind a barrier that every wave reaches after ISSUING the MFMAs that consumed
//        its fragments (a wave waits lgkmcnt for a fragment before the MFMA that reads it, so none of its LDS reads
//        of that buffer can still be outstanding);
//   the asm statements carry a "memory" clobber, so the compiler moves no LDS access across them.
#pragma once
#include "common.h"

namespace unetdc {

#if defined(__HIP_DEVICE_COMPILE__)

// byte address of a __shared__ object as the LDS-DMA engine wants it in M0
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(unsigned long)((__attribute__((address_space(3))) const unsigned char*)p);
}

// One wave-instruction: 64 lanes x 16 bytes, lane l lands at lds_dst + 16*l; lane source = base + voff + soff.
// voff beyond the descriptor's range (e.g. 0x80000000) -> the hardware writes zeros (soff is not range-checked).
// lds_dst and soff must be wave-uniform; M0 is saved and restored inside the statement (the compiler treats M0 as its
// own).  s_nop 2: with the two s_mov in front of it the DMA issues >= 5 wait states after the statement starts, which
// covers both "SALU writes M0 -> LDS-DMA" and "v_readfirstlane writes an SGPR -> VMEM reads it as soffset" (hipcc pads
// no hazard whose consumer sits inside an asm string).  "vcc" is clobbered only to keep the register allocator from
// handing vcc_lo to an "s" operand: MUBUF does not accept it as soffset.
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned lds_dst, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 2\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(lds_dst), "s"(rsrc), "s"(soff)
               : "memory", "vcc");
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// workgroup barrier that does not drain the VM counter (see the header comment: RAW / WAR rules)
__device__ __forceinline__ void raw_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

#endif  // __HIP_DEVICE_COMPILE__

}  // namespace unetdc
