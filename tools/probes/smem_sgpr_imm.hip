// Probe: does gfx950 apply BOTH offsets of an SMEM load that carries an SGPR offset AND an immediate one,
//     s_load_dword sD, s[base:base+1], sOFF offset:IMM        (address = base + sOFF + IMM) ?
//
// Background (csrc/wgrad_rect.hip, WgradRectParams): a per-tap table in the kernel-argument segment, indexed by a tap number
// that had itself been loaded from memory, came back with the WRONG entries, deterministically.  hipcc (ROCm 7.2) had compiled
// the access to
//     v_readfirstlane_b32 s12, v7 ; s_lshl_b32 s12, s12, 2 ; s_load_dword s11, s[6:7], s12 offset:0xb0
// tools/probes/valu_sgpr_smem.hip showed that the VALU -> SGPR -> SMEM path is NOT the cause (no hazard at 0..8 wait states)
// but it issued the load WITHOUT an immediate offset.  This probe issues the combined form for a table of 64 entries with
// IMM = 0, 16, 0xb0 and reports what came back: table[(sOFF + IMM) / 4] (both applied), table[sOFF / 4] (immediate dropped) or
// table[IMM / 4] (SGPR dropped).
//
//   hipcc --offload-arch=gfx950 -O3 -o smem_sgpr_imm tools/probes/smem_sgpr_imm.hip && ./smem_sgpr_imm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int IMM>
__global__ void probe(const int* __restrict__ idx, const int* __restrict__ table, int* __restrict__ out) {
  int got;
  const int off = __builtin_amdgcn_readfirstlane(idx[blockIdx.x] * 4);      // wave-uniform byte offset in an SGPR
  asm volatile(
      "s_nop 7\n\t"
      "s_load_dword %0, %2, %1 offset:%3\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=s"(got)
      : "s"(off), "s"(table), "n"(IMM)
      : "memory");
  if (threadIdx.x == 0) out[blockIdx.x] = got;
}

template <int IMM> static int run(const int* didx, const int* dtab, int* dout, const std::vector<int>& hidx, int nblk) {
  hipMemset(dout, 0xff, nblk * sizeof(int));
  hipLaunchKernelGGL(probe<IMM>, dim3(nblk), dim3(64), 0, 0, didx, dtab, dout);
  std::vector<int> h(nblk);
  hipMemcpy(h.data(), dout, nblk * sizeof(int), hipMemcpyDeviceToHost);
  int both = 0, sgpr_only = 0, imm_only = 0, other = 0;
  for (int b = 0; b < nblk; ++b) {
    if (h[b] == 1000 + hidx[b] + IMM / 4) ++both;
    else if (h[b] == 1000 + hidx[b]) ++sgpr_only;
    else if (h[b] == 1000 + IMM / 4) ++imm_only;
    else ++other;
  }
  printf("offset:%#x  base + sgpr + imm: %d   base + sgpr (immediate dropped): %d   base + imm (sgpr dropped): %d   other: %d   of %d\n",
         IMM, both, sgpr_only, imm_only, other, nblk);
  return nblk - both;
}

int main() {
  const int nblk = 4096, ntab = 256;
  std::vector<int> hidx(nblk), htab(ntab);
  for (int i = 0; i < ntab; ++i) htab[i] = 1000 + i;
  for (int b = 0; b < nblk; ++b) hidx[b] = (b * 7 + 5) % 64;
  int *didx, *dtab, *dout;
  hipMalloc(&didx, nblk * sizeof(int)); hipMalloc(&dtab, ntab * sizeof(int)); hipMalloc(&dout, nblk * sizeof(int));
  hipMemcpy(didx, hidx.data(), nblk * sizeof(int), hipMemcpyHostToDevice);
  hipMemcpy(dtab, htab.data(), ntab * sizeof(int), hipMemcpyHostToDevice);
  int bad = 0;
  bad += run<0>(didx, dtab, dout, hidx, nblk);
  bad += run<16>(didx, dtab, dout, hidx, nblk);
  bad += run<0xb0>(didx, dtab, dout, hidx, nblk);
  printf(bad ? "COMBINED FORM MISBEHAVES\n" : "combined form applies both offsets\n");
  return 0;
}
