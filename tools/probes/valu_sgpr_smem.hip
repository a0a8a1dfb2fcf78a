// Probe: is an SGPR written by a VALU instruction (v_readfirstlane / v_readlane) safe to use, immediately, as the
// SOFFSET of a scalar load on gfx950?
//
// Found while writing csrc/wgrad_rect.hip: a per-tap table in the kernel-argument segment, indexed by a tap number that
// had itself been loaded with a vector load, came back with the WRONG entries, deterministically.  hipcc (ROCm 7.2) had
// compiled the access to
//     global_load_ubyte v7, ... ; s_waitcnt vmcnt(0) ; v_readfirstlane_b32 s12, v7 ; s_lshl_b32 s12, s12, 2 ;
//     s_load_dword s11, s[6:7], s12 offset:0xb0
// LLVM's hazard recognizer pads "VALU writes SGPR -> VMEM reads it" (5 wait states) on gfx9 but treats
// "VALU writes SGPR -> SMEM reads it" as hazard-free after Southern Islands.  This probe measures it: each workgroup
// reads table[idx[block]] through exactly that sequence, with 0..7 s_nop wait states between the readfirstlane and the
// s_load, and counts mismatches.  (Relevant beyond wgrad_rect: a spilled SGPR is restored with v_readlane, i.e. by the
// VALU, so a kernel with SGPR spills can feed a just-restored SGPR to an SMEM/VMEM instruction -- the round-1 build whose
// fused BatchNorm-backward epilogue was not reproducible was the one variant WITH SGPR spills.)
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_sgpr_smem tools/probes/valu_sgpr_smem.hip && ./valu_sgpr_smem
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int NOPS>
__global__ void probe(const int* __restrict__ idx, const int* __restrict__ table, int* __restrict__ out, int prev_seed) {
  // a different, VALID offset sits in the destination SGPR first, so that a too-early read is visible as a wrong entry
  int got;
  const int i = idx[blockIdx.x];                  // vector load -> VGPR
  int prev = prev_seed * 4;
  asm volatile(
      "s_mov_b32 s40, %2\n\t"                     // old content of the offset register: entry `prev_seed`
      "s_nop 7\n\t"
      "v_lshlrev_b32 %1, 2, %1\n\t"
      "s_nop 1\n\t"
      "v_readfirstlane_b32 s40, %1\n\t"           // VALU writes the SGPR ...
      ".if %4 > 0\n\ts_nop %4 - 1\n\t.endif\n\t"
      "s_load_dword %0, %3, s40\n\t"              // ... SMEM reads it as soffset
      "s_waitcnt lgkmcnt(0)"
      : "=s"(got)
      : "v"(i), "s"(prev), "s"(table), "n"(NOPS)
      : "s40", "memory");
  if (threadIdx.x == 0) out[blockIdx.x] = got;
}

template <int NOPS> static int run(const int* didx, const int* dtab, int* dout, const std::vector<int>& hidx, int nblk) {
  hipMemset(dout, 0xff, nblk * sizeof(int));
  hipLaunchKernelGGL(probe<NOPS>, dim3(nblk), dim3(64), 0, 0, didx, dtab, dout, 3);
  std::vector<int> h(nblk);
  hipMemcpy(h.data(), dout, nblk * sizeof(int), hipMemcpyDeviceToHost);
  int bad = 0, stale = 0;
  for (int b = 0; b < nblk; ++b) {
    if (h[b] != 1000 + hidx[b]) ++bad;
    if (h[b] == 1000 + 3 && hidx[b] != 3) ++stale;
  }
  printf("wait states between v_readfirstlane and s_load: %d   wrong entries: %d of %d   (of which the STALE offset's entry: %d)\n",
         NOPS, bad, nblk, stale);
  return bad;
}

int main() {
  const int nblk = 4096, ntab = 64;
  std::vector<int> hidx(nblk), htab(ntab);
  for (int i = 0; i < ntab; ++i) htab[i] = 1000 + i;
  for (int b = 0; b < nblk; ++b) hidx[b] = (b * 7 + 5) % ntab;
  int *didx, *dtab, *dout;
  hipMalloc(&didx, nblk * sizeof(int)); hipMalloc(&dtab, ntab * sizeof(int)); hipMalloc(&dout, nblk * sizeof(int));
  hipMemcpy(didx, hidx.data(), nblk * sizeof(int), hipMemcpyHostToDevice);
  hipMemcpy(dtab, htab.data(), ntab * sizeof(int), hipMemcpyHostToDevice);
  int bad = 0;
  bad += run<0>(didx, dtab, dout, hidx, nblk);
  bad += run<1>(didx, dtab, dout, hidx, nblk);
  bad += run<2>(didx, dtab, dout, hidx, nblk);
  bad += run<3>(didx, dtab, dout, hidx, nblk);
  bad += run<4>(didx, dtab, dout, hidx, nblk);
  bad += run<5>(didx, dtab, dout, hidx, nblk);
  bad += run<8>(didx, dtab, dout, hidx, nblk);
  printf(bad ? "HAZARD OBSERVED\n" : "no hazard observed\n");
  return 0;
}
