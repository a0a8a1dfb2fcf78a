// Probe: does an out-of-range lane of `buffer_load_dwordx4 ... lds` write ZEROS into LDS or skip the write?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const void* src, float* out, int nbytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  float* f = (float*)smem;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) f[i] = 123.0f;      // sentinel
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, nbytes, 0x00020000);
  unsigned voff = (lane & 1) ? 0x80000000u : lane * 16;                    // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
  unsigned voff2 = (lane & 2) ? (unsigned)(nbytes + lane * 16) : lane * 16; // just past the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + 1024), 16, voff2, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += blockDim.x) out[i] = f[i];
}
int main() {
  const int n = 64 * 4;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = 1000.f + i;
  float *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 512 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o, n * 4);
  std::vector<float> r(512);
  hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
  printf("err=%s\n", hipGetErrorString(hipGetLastError()));
  for (int l = 0; l < 8; ++l) printf("lane %d: A=[%g %g %g %g]  B=[%g %g %g %g]\n", l, r[l*4], r[l*4+1], r[l*4+2], r[l*4+3], r[256+l*4], r[256+l*4+1], r[256+l*4+2], r[256+l*4+3]);
  return 0;
}
