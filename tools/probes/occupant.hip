// Stand-in for a communication kernel on a one-GPU box (round-3 verdict item 6c): `nwg` workgroups of `threads` threads with
// `lds` bytes of LDS each sit on the chip for `cycles` shader clocks, doing nothing, on a stream of their own -- what an RCCL
// ring kernel does to the RESIDENCY of the persistent compute grids (it takes wave slots, registers and LDS on nwg CUs).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/occupant.hip -o tools/probes/liboccupant.so
#include <hip/hip_runtime.h>

__global__ void occupant_kernel(unsigned long long cycles, int* sink) {
  extern __shared__ int s[];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int k = 0;
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {
    __builtin_amdgcn_s_sleep(16);
    ++k;
  }
  if (sink && threadIdx.x == 1023 && k < 0) *sink = s[0];
}

extern "C" int occupant_launch(int nwg, int threads, int lds, unsigned long long cycles, void* stream) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&occupant_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return -1;
    attr = true;
  }
  hipLaunchKernelGGL(occupant_kernel, dim3(nwg), dim3(threads), lds, (hipStream_t)stream, cycles, (int*)nullptr);
  return (int)hipGetLastError();
}
