// Probe: is the HIGH half of a v_pk_fma_f32 result safe to consume by a VALU compare a few instructions later?
//
// Background ("A hardware/compiler hazard worth recording"): an SLP-vectorised epilogue that did
//     v_pk_fma_f32 v[d:d+1], ...      ; n = y*scale + shift for an (even, odd) channel pair
//     <0..2 independent VALU instructions>
//     v_cmp_lt_f32 vcc, 0, v[d+1]     ; gate of the ODD channel
//     v_cndmask_b32 g, 0, t, vcc
// dropped or wrongly kept single contributions of the odd channel in lanes 48-63, about once per 8192 workgroups.
// This probe runs exactly that sequence with GAP = 0, 1, 2 filler instructions, many waves per SIMD, a stream of
// buffer stores in flight (as in the epilogue), and counts lanes whose gate disagrees with the scalar evaluation.
//
//   hipcc --offload-arch=gfx950 -O2 tools/probes/pk_hazard.hip -o /tmp/pk_hazard && /tmp/pk_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int GAP>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ in, float* __restrict__ sink,
                                             unsigned long long* __restrict__ bad, int iters) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  unsigned long long mism = 0;
  float y0 = in[(gid * 7 + 0) & 0xffff], y1 = in[(gid * 7 + 1) & 0xffff];
  const float s0 = in[(gid * 7 + 2) & 0xffff], s1 = in[(gid * 7 + 3) & 0xffff];
  const float h0 = in[(gid * 7 + 4) & 0xffff], h1 = in[(gid * 7 + 5) & 0xffff];
  for (int it = 0; it < iters; ++it) {
    const float t = 1.0f + (float)(it & 7);
    float g;
    // v[46:47] is preset to a value whose sign is the OPPOSITE of the expected result, so a stale read flips the gate
    const float expect1 = fmaf(y1, s1, h1);
    const float stale = expect1 > 0.f ? -1.0f : 1.0f;
    if (GAP == 0) {
      asm volatile(
          "v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\t"
          "v_mov_b32 v44, %5\n\tv_mov_b32 v45, %6\n\tv_mov_b32 v46, %8\n\tv_mov_b32 v47, %8\n\ts_nop 7\n\t"
          "v_pk_fma_f32 v[46:47], v[40:41], v[42:43], v[44:45]\n\t"
          "v_cmp_lt_f32 vcc, 0, v47\n\t"
          "s_nop 1\n\t"
          "v_cndmask_b32 %0, 0, %7, vcc\n\t"
          : "=v"(g)
          : "v"(y0), "v"(y1), "v"(s0), "v"(s1), "v"(h0), "v"(h1), "v"(t), "v"(stale)
          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
    } else if (GAP == 1) {
      asm volatile(
          "v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\t"
          "v_mov_b32 v44, %5\n\tv_mov_b32 v45, %6\n\tv_mov_b32 v46, %8\n\tv_mov_b32 v47, %8\n\ts_nop 7\n\t"
          "v_pk_fma_f32 v[46:47], v[40:41], v[42:43], v[44:45]\n\t"
          "v_add_f32 v48, v48, v48\n\t"
          "v_cmp_lt_f32 vcc, 0, v47\n\t"
          "s_nop 1\n\t"
          "v_cndmask_b32 %0, 0, %7, vcc\n\t"
          : "=v"(g)
          : "v"(y0), "v"(y1), "v"(s0), "v"(s1), "v"(h0), "v"(h1), "v"(t), "v"(stale)
          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "vcc");
    } else {
      asm volatile(
          "v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\t"
          "v_mov_b32 v44, %5\n\tv_mov_b32 v45, %6\n\tv_mov_b32 v46, %8\n\tv_mov_b32 v47, %8\n\ts_nop 7\n\t"
          "v_pk_fma_f32 v[46:47], v[40:41], v[42:43], v[44:45]\n\t"
          "v_add_f32 v48, v48, v48\n\t"
          "v_add_f32 v48, v48, v48\n\t"
          "v_cmp_lt_f32 vcc, 0, v47\n\t"
          "s_nop 1\n\t"
          "v_cndmask_b32 %0, 0, %7, vcc\n\t"
          : "=v"(g)
          : "v"(y0), "v"(y1), "v"(s0), "v"(s1), "v"(h0), "v"(h1), "v"(t), "v"(stale)
          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "vcc");
    }
    const float want = expect1 > 0.f ? t : 0.f;
    if (g != want) ++mism;
    sink[(size_t)gid * 4 + (it & 3)] = g;                  // keep stores in flight like the real epilogue
    y1 = -y1;                                              // alternate the sign so the stale value matters both ways
    y0 = y0 * 1.0001f;
  }
  if (mism) {
    atomicAdd(&bad[lane >> 4], 1ull);                      // which quarter of the wave
    atomicAdd(&bad[4], 1ull);
  }
}

int main() {
  const int nthreads = 256 * 256 * 16, iters = 2000;
  std::vector<float> h(65536);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *din, *sink;
  unsigned long long* bad;
  hipMalloc(&din, h.size() * 4);
  hipMalloc(&sink, (size_t)nthreads * 4 * 4);
  hipMalloc(&bad, 5 * 8);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int gap = 0; gap < 3; ++gap) {
    hipMemset(bad, 0, 5 * 8);
    if (gap == 0) hipLaunchKernelGGL(probe<0>, dim3(nthreads / 256), dim3(256), 0, 0, din, sink, bad, iters);
    if (gap == 1) hipLaunchKernelGGL(probe<1>, dim3(nthreads / 256), dim3(256), 0, 0, din, sink, bad, iters);
    if (gap == 2) hipLaunchKernelGGL(probe<2>, dim3(nthreads / 256), dim3(256), 0, 0, din, sink, bad, iters);
    hipDeviceSynchronize();
    unsigned long long r[5];
    hipMemcpy(r, bad, sizeof(r), hipMemcpyDeviceToHost);
    printf("gap %d: %s; threads with a wrong gate: %llu of %d (x %d iterations); quarters hit: %llu %llu %llu %llu\n", gap,
           hipGetErrorString(hipGetLastError()), r[4], nthreads, iters, r[0], r[1], r[2], r[3]);
  }
  return 0;
}
