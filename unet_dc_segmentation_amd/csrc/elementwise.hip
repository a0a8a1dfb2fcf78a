// HBM-bound kernels of the U-Net path: BatchNorm statistics/normalisation (+ReLU, +2x2 max-pool),
// their backward (with the pool scatter and skip-gradient sum folded in), the 1x1 sigmoid head,
// and weight re-packing.  All tensors NHWC, every global access is a 16-byte chunk; reductions
// are two-stage with fixed summation order (no float atomics -> bitwise reproducible).
#include <stdlib.h>

#include "kernels.h"

namespace unetdc {

// ================================================================================================
// column sums of a [nrows][L] fp32 matrix into R row groups: out[r][l] = sum_{i in group r} in[i][l]
// ================================================================================================
__global__ __launch_bounds__(256) void colsum_stage_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int nrows, int L, int rows_per_group) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cl, grp = blockIdx.y;
  const int rbeg = grp * rows_per_group;
  const int rend = min(rbeg + rows_per_group, nrows);
  float s = 0.f;
  if (col < L)
    for (int i = rbeg + rl; i < rend; i += 8) s += in[(long)i * L + col];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && col < L) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][cl];
    out[(long)grp * L + col] = t;
  }
}

// Reduce `nparts` rows to at most 64 rows written at parts + nparts*L (caller provides 64 spare rows).
// Returns pointer/rows of the reduced set through the out-params.
// `direct` = row count up to which the consumer kernel reads the rows itself (64 for the serial finalizers; the
// 32-lanes-per-channel BatchNorm finalizers take 512 rows = 16 loads per lane and save this launch).
int reduce_parts(const float* parts, int nparts, int L, const float** red_ptr, int* red_rows, hipStream_t stream,
                 int direct = 64) {
  if (nparts <= direct) {
    *red_ptr = parts;
    *red_rows = nparts;
    return UNETDC_OK;
  }
  float* out = const_cast<float*>(parts) + (long)nparts * L;
  const int rpg = (nparts + 63) / 64;
  const int groups = (nparts + rpg - 1) / rpg;
  hipLaunchKernelGGL(colsum_stage_kernel, dim3((L + 31) / 32, groups), dim3(256), 0, stream, parts, out, nparts, L, rpg);
  *red_ptr = out;
  *red_rows = groups;
  return check_launch("colsum_stage_kernel");
}

// ================================================================================================
// BatchNorm (train): finalize batch statistics  (nn.BatchNorm2d, models/model_2.py:45,52)
//   parts[i][0][c] = partial sum, parts[i][1][c] = partial sum of squares
//   scale = gamma*rstd, shift = beta - mean*scale; running stats use the UNBIASED variance.
// ================================================================================================
// A workgroup owns FOUR consecutive channels: thread t takes partial rows t, t + 256, ... (one 16-byte load per row and
// statistic -- at most two trips for the <= 512 rows the producers write, all loads in flight at once), fp64 from there
// on: wave shuffle, then the four waves in fixed order through LDS -> deterministic.  (The first form, a serial per-thread
// loop over the rows, cost ~17 us per launch in load latency alone; the second, 32 lanes per channel and 16 dependent trips,
// 6.5-13 us.)
constexpr int FIN_C = 4;

__device__ __forceinline__ double lane32_sum(double v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
  return v;
}

template <int NROW>
__device__ __forceinline__ void fin_rows(const float* __restrict__ parts, int nparts, int cs, int col, int nch,
                                         double (&acc)[NROW][FIN_C]) {
  const bool vec = nch == FIN_C && ((cs | col) & 3) == 0 && (reinterpret_cast<uintptr_t>(parts) & 15) == 0;
  for (int i = threadIdx.x; i < nparts; i += 256) {
#pragma unroll
    for (int w = 0; w < NROW; ++w) {
      const float* src = parts + ((long)i * NROW + w) * cs + col;
      float v[FIN_C] = {0.f, 0.f, 0.f, 0.f};
      if (vec) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
        for (int e = 0; e < FIN_C; ++e) v[e] = t[e];
      } else {
#pragma unroll
        for (int e = 0; e < FIN_C; ++e) if (e < nch) v[e] = src[e];
      }
#pragma unroll
      for (int e = 0; e < FIN_C; ++e) acc[w][e] += (double)v[e];
    }
  }
}

// totals of the workgroup in red[w * FIN_C + e] (valid after the call for every thread)
template <int NROW>
__device__ __forceinline__ void fin_block_sum(double (&acc)[NROW][FIN_C], double* red, double* tot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < NROW; ++w)
#pragma unroll
    for (int e = 0; e < FIN_C; ++e) {
      double v = acc[w][e];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) red[wave * NROW * FIN_C + w * FIN_C + e] = v;
    }
  __syncthreads();
  if (threadIdx.x < NROW * FIN_C) {
    const int k = threadIdx.x;
    tot[k] = ((red[k] + red[NROW * FIN_C + k]) + red[2 * NROW * FIN_C + k]) + red[3 * NROW * FIN_C + k];
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ parts, int nparts, double count,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float momentum,
                                                          float* running_mean, float* running_var, float* scale,
                                                          float* shift, float* mean_out, float* rstd_out, int C) {
  __shared__ double red[4 * 2 * FIN_C], tot[2 * FIN_C];
  const int c4 = blockIdx.x * FIN_C;
  const int nch = min(FIN_C, C - c4);
  double acc[2][FIN_C] = {};
  fin_rows<2>(parts, nparts, C, c4, nch, acc);
  fin_block_sum<2>(acc, red, tot);
  if ((int)threadIdx.x >= nch) return;
  const int c = c4 + threadIdx.x;
  const double s = tot[threadIdx.x], q = tot[FIN_C + threadIdx.x];
  const double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  mean_out[c] = (float)mean;
  rstd_out[c] = rstd;
  if (running_mean) {
    const double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// BatchNorm (eval) folded with the conv bias: y = relu(acc*scale + shift)
__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv,
                                      const float* __restrict__ conv_bias, float eps, float* scale, float* shift,
                                      int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float s = gamma[c] / sqrtf(rv[c] + eps);
  scale[c] = s;
  shift[c] = beta[c] + ((conv_bias ? conv_bias[c] : 0.f) - rm[c]) * s;
}

// ================================================================================================
// a = relu(scale*y + shift)  (or a = y when scale == null), optional 2x2 max-pool of a
//   F.max_pool2d(x, 2): models/model_2.py:59-61,64.   POOL: one thread per (2x2 window, chunk).
// ================================================================================================
template <typename T, bool POOL>
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const ApplyParams p) {
  constexpr int EPC = Chunk<T>::N;
  const int cpp = p.C / EPC;                                  // chunks per pixel
  const int Hq = POOL ? p.H / 2 : p.H, Wq = POOL ? p.W / 2 : p.W;
  const long total = (long)p.N * Hq * Wq * cpp;
  const T* __restrict__ yg = reinterpret_cast<const T*>(p.y);
  T* __restrict__ ag = reinterpret_cast<T*>(p.a);
  T* __restrict__ pg = reinterpret_cast<T*>(p.pooled);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int ch = (int)(idx % cpp);
    const long q = idx / cpp;
    const int c0 = ch * EPC;
    float sc[EPC], sh[EPC];
    if (p.scale) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { sc[e] = p.scale[c0 + e]; sh[e] = p.shift[c0 + e]; }
    }
    if (!POOL) {
      float v[EPC];
      Chunk<T>::unpack(ld16(yg + q * p.ldy + c0), v);
      if (p.scale) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
      }
      st16(ag + q * p.lda + c0, Chunk<T>::pack(v));
    } else {
      const int xq = (int)(q % Wq);
      const long t2 = q / Wq;
      const int yq = (int)(t2 % Hq), n = (int)(t2 / Hq);
      float best[EPC];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long pix = ((long)n * p.H + 2 * yq + (k >> 1)) * p.W + 2 * xq + (k & 1);
        float v[EPC];
        Chunk<T>::unpack(ld16(yg + pix * p.ldy + c0), v);
        if (p.scale) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] = round_through<T>(fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f));
          if (ag) st16(ag + pix * p.lda + c0, Chunk<T>::pack(v));
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) best[e] = (k == 0) ? v[e] : fmaxf(best[e], v[e]);
      }
      st16(pg + q * p.ldp + c0, Chunk<T>::pack(best));
    }
  }
}

// ================================================================================================
// BatchNorm+ReLU backward.  Incoming gradient dA of the activated output a = relu(scale*y+shift):
//     dA = dskip (+ scatter of dpool to the window arg-max when POOL)
//   dyhat = dA * [a > 0];  xhat = (y - mean)*rstd
//   reduce:   S1 = sum dyhat, S2 = sum dyhat*xhat, S3 = sum xhat      (per channel)
//   apply:    dy = k1*dyhat - k2 - k3*xhat,   k1 = gamma*rstd, k2 = k1*S1/M, k3 = k1*S2/M
//   dgamma = S2, dbeta = S1, dbias(conv) = sum dy = -k3*S3  (zero up to rounding, as in the reference)
// The pool arg-max is recomputed from y exactly as the forward pass saw it (values rounded through
// the storage type, first maximum in row-major window order: ATen's tie rule).
// ================================================================================================
// HEAD (non-pooled apply pass only): the incoming gradient is recomputed from the head's dprobs / probs / weight row
// (BnBwdParams::head_w) -- a template parameter, because the extra registers and the branch cost the plain form 40 % as a
// run-time switch
template <typename T, bool POOL, bool APPLY, bool HEAD = false>
__global__ __launch_bounds__(256) void bn_bwd_kernel(const BnBwdParams p) {
  static_assert(!HEAD || (!POOL && APPLY), "head form: non-pooled apply pass");
  constexpr int EPC = Chunk<T>::N;
  __shared__ float red[256 * 3 * EPC];
  const int cpp = p.C / EPC;
  const int tid = threadIdx.x;
  // thread -> fixed channel chunk; pixel lanes stride over the (window) list
  const int seg = (cpp < 256) ? cpp : 256;                 // chunk lanes per block row
  const int plane = 256 / seg;                             // pixel lanes
  const int chl = tid % seg, pl = tid / seg;
  const int nseg = (cpp + seg - 1) / seg;                  // gridDim.y
  const int ch = blockIdx.y * seg + chl;
  const bool active = (pl < plane) && (ch < cpp);
  const int c0 = ch * EPC;
  const int Hq = POOL ? p.H / 2 : p.H, Wq = POOL ? p.W / 2 : p.W;
  const long Q = (long)p.N * Hq * Wq;
  const T* __restrict__ yg = reinterpret_cast<const T*>(p.y);
  const T* __restrict__ sg = reinterpret_cast<const T*>(p.dskip);
  const T* __restrict__ dpg = reinterpret_cast<const T*>(p.dpool);
  T* __restrict__ dyg = reinterpret_cast<T*>(p.dy);
  (void)nseg;

  float sc[EPC], sh[EPC], mu[EPC], rs[EPC], k1[EPC], k2[EPC], k3[EPC], hw[EPC];
  float s1[EPC], s2[EPC], s3[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    s1[e] = s2[e] = s3[e] = 0.f;
    hw[e] = 0.f;
    if (active) {
      sc[e] = p.scale[c0 + e]; sh[e] = p.shift[c0 + e]; mu[e] = p.mean[c0 + e]; rs[e] = p.rstd[c0 + e];
      if (APPLY) { k1[e] = p.k1[c0 + e]; k2[e] = p.k2[c0 + e]; k3[e] = p.k3[c0 + e]; }
      if (HEAD) hw[e] = p.head_w[c0 + e];
    }
  }
  if (active) {
    for (long q = (long)blockIdx.x * plane + pl; q < Q; q += (long)gridDim.x * plane) {
      if (!POOL) {
        float yv[EPC], g[EPC];
        Chunk<T>::unpack(ld16(yg + q * p.ldy + c0), yv);
        if (HEAD) {                                        // gradient of the head's input, recomputed (see BnBwdParams)
          const float pr = p.head_probs[q];
          const float dz = p.head_dprobs[q] * pr * (1.f - pr);
#pragma unroll
          for (int e = 0; e < EPC; ++e) g[e] = round_through<T>(dz * hw[e]);
        } else {
          Chunk<T>::unpack(ld16(sg + q * p.lds + c0), g);
        }
        float out[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const float nrm = fmaf(yv[e], sc[e], sh[e]);
          const float gh = nrm > 0.f ? g[e] : 0.f;
          const float xh = (yv[e] - mu[e]) * rs[e];
          if (APPLY) out[e] = fmaf(k1[e], gh, -k2[e]) - k3[e] * xh;
          else { s1[e] += gh; s2[e] = fmaf(gh, xh, s2[e]); s3[e] += xh; }
        }
        if (APPLY) st16(dyg + q * p.lddy + c0, Chunk<T>::pack(out));
      } else {
        // The nine 16-byte chunks of a window (4 x y, 4 x dskip, dpool) stay RAW and every channel is taken out of them when
        // its turn comes: holding them unpacked cost 160 / 192 registers (3 / 2 waves per SIMD on a pass that lives on
        // memory-level parallelism).  Arithmetic and summation order per channel are unchanged.
        const int xq = (int)(q % Wq);
        const long t2 = q / Wq;
        const int yq = (int)(t2 % Hq), n = (int)(t2 / Hq);
        long pix[4];
        u32x4 yr[4], sr[4], dpr, outr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          pix[k] = ((long)n * p.H + 2 * yq + (k >> 1)) * p.W + 2 * xq + (k & 1);
          yr[k] = ld16(yg + pix[k] * p.ldy + c0);
        }
        dpr = ld16(dpg + q * p.ldp + c0);
        if (sg) {
#pragma unroll
          for (int k = 0; k < 4; ++k) sr[k] = ld16(sg + pix[k] * p.lds + c0);
        }
        float prev[4] = {0.f, 0.f, 0.f, 0.f};               // APPLY, bf16: the even element of a pair waits for the odd one
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float yk[4], nrm[4];
          float best = 0.f;
          int arg = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            yk[k] = chunk_elem<T>(yr[k], e);
            nrm[k] = fmaf(yk[k], sc[e], sh[e]);
            const float a = round_through<T>(fmaxf(nrm[k], 0.f));
            if (k == 0 || a > best) { best = a; arg = k; }
          }
          const float dpe = chunk_elem<T>(dpr, e);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float gin = (sg ? chunk_elem<T>(sr[k], e) : 0.f) + (arg == k ? dpe : 0.f);
            const float gh = nrm[k] > 0.f ? gin : 0.f;
            const float xh = (yk[k] - mu[e]) * rs[e];
            if (APPLY) {
              const float o = fmaf(k1[e], gh, -k2[e]) - k3[e] * xh;
              chunk_set<T>(outr[k], e, o, prev[k]);
            } else {
              s1[e] += gh; s2[e] = fmaf(gh, xh, s2[e]); s3[e] += xh;
            }
          }
        }
        if (APPLY) {
#pragma unroll
          for (int k = 0; k < 4; ++k) st16(dyg + pix[k] * p.lddy + c0, outr[k]);
        }
      }
    }
  }
  if (!APPLY) {
    // block reduction over pixel lanes (fixed order)
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      red[(tid * 3 + 0) * EPC + e] = s1[e];
      red[(tid * 3 + 1) * EPC + e] = s2[e];
      red[(tid * 3 + 2) * EPC + e] = s3[e];
    }
    __syncthreads();
    // one thread per (chunk lane, which, element)
    for (int i = tid; i < seg * 3 * EPC; i += 256) {
      const int cl = i / (3 * EPC), rem = i - cl * 3 * EPC, which = rem / EPC, e = rem - which * EPC;
      const int chn = blockIdx.y * seg + cl;
      if (chn >= cpp) continue;
      float s = 0.f;
      for (int q2 = 0; q2 < plane; ++q2) s += red[((q2 * seg + cl) * 3 + which) * EPC + e];
      p.parts[((long)blockIdx.x * 3 + which) * p.C + chn * EPC + e] = s;
    }
  }
}

// Two sources of partial sums: A = [nA][3][csA] rows of which columns c0A .. c0A + C - 1 belong to this stage (csA = C,
// c0A = 0 for the stage's own partial buffer; csA = 2C, c0A = C when the sums rode in the epilogue of the kernel that wrote
// the [pixels, 2C] concat gradient), B = [nB][3][C] (optional second producer: the pooled part of an encoder stage).
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ parts, int nparts, int cs, int c0,
                                                              const float* __restrict__ parts2, int nparts2,
                                                              double count, const float* __restrict__ gamma,
                                                              const float* __restrict__ rstd, float* dgamma,
                                                              float* dbeta, float* dbias, float* k1, float* k2,
                                                              float* k3, int C, int frozen) {
  __shared__ double red[4 * 3 * FIN_C], tot[3 * FIN_C];
  const int c4 = blockIdx.x * FIN_C;
  const int nch = min(FIN_C, C - c4);
  double acc[3][FIN_C] = {};
  fin_rows<3>(parts, nparts, cs, c0 + c4, nch, acc);
  if (nparts2 > 0) fin_rows<3>(parts2, nparts2, C, c4, nch, acc);
  fin_block_sum<3>(acc, red, tot);
  if ((int)threadIdx.x >= nch) return;
  const int c = c4 + threadIdx.x;
  const double s1 = tot[threadIdx.x], s2 = tot[FIN_C + threadIdx.x], s3 = tot[2 * FIN_C + threadIdx.x];
  const double a = (double)gamma[c] * (double)rstd[c];
  dgamma[c] = (float)s2;
  dbeta[c] = (float)s1;
  k1[c] = (float)a;
  if (frozen) {
    // statistics that do not depend on the batch (eval-mode BatchNorm under autograd: running mean / variance): the
    // normalisation is a fixed per-channel affine map, dy = gamma * rstd * dyhat, and the conv bias sees sum(dy)
    k2[c] = 0.f;
    k3[c] = 0.f;
    if (dbias) dbias[c] = (float)(a * s1);
    return;
  }
  k2[c] = (float)(a * s1 / count);
  k3[c] = (float)(a * s2 / count);
  if (dbias) dbias[c] = (float)(-(a * s2 / count) * s3);
}

// ================================================================================================
// Head: 1x1 conv (C -> OC) + sigmoid (models/model_2.py:32,79-80), probabilities out in NCHW fp32.
// C/EPC lanes cooperate on one pixel (C = 64: 8 lanes bf16 / 16 lanes fp32), shuffle-reduced.
// ================================================================================================
// NORM: `a` is the raw conv output of the producing stage and the BatchNorm + ReLU of that stage is applied on load,
// a = relu(scale * y + shift) rounded through the storage type (exactly the values a stand-alone normalisation pass
// would have stored): the activation tensor is never written or read.
template <typename T, bool NORM, int CPP>
__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadParams p) {
  constexpr int EPC = Chunk<T>::N;
  constexpr int UNR = 4;                              // pixels per lane group and trip: four 16-byte loads in flight
  const int cpp = CPP ? CPP : p.C / EPC;              // lanes per pixel (power of two, <= 64); CPP > 0: known at compile time
  const int ppb = 256 / cpp;
  const int tid = threadIdx.x, cl = tid % cpp, pl = tid / cpp;
  const long HW = (long)p.H * p.W, P = (long)p.N * HW;
  const T* __restrict__ ag = reinterpret_cast<const T*>(p.a);
  float nsc[EPC], nsh[EPC];
  if (NORM) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) { nsc[e] = p.bn_scale[cl * EPC + e]; nsh[e] = p.bn_shift[cl * EPC + e]; }
  }
  for (long pb = (long)blockIdx.x * ppb * UNR; pb < P; pb += (long)gridDim.x * ppb * UNR) {
    float v[UNR][EPC];
    long pix[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      pix[u] = pb + u * ppb + pl;
      if (pix[u] < P) Chunk<T>::unpack(ld16(ag + pix[u] * p.lda + cl * EPC), v[u]);
    }
    if (NORM) {
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[u][e] = round_through<T>(fmaxf(fmaf(v[u][e], nsc[e], nsh[e]), 0.f));
    }
    for (int oc = 0; oc < p.OC; ++oc) {
      float wv[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) wv[e] = p.w[oc * p.C + cl * EPC + e];
      const float bias = p.b[oc];
      // the four dot products of a trip are reduced TOGETHER, step by step: four independent shuffle chains instead of four
      // serial ones (a cross-lane step is an LDS-crossbar round trip; with a runtime lane count the loop was not unrolled
      // and the kernel ran at 2.9 TB/s on shuffle latency)
      float sv[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        sv[u] = 0.f;
        if (pix[u] < P) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) sv[u] = fmaf(v[u][e], wv[e], sv[u]);
        }
      }
      if (CPP) {
#pragma unroll
        for (int o = 1; o < (CPP ? CPP : 1); o <<= 1)
#pragma unroll
          for (int u = 0; u < UNR; ++u) sv[u] += __shfl_xor(sv[u], o, 64);
      } else {
        for (int o = 1; o < cpp; o <<= 1)
#pragma unroll
          for (int u = 0; u < UNR; ++u) sv[u] += __shfl_xor(sv[u], o, 64);
      }
      // every lane of the group holds the four sums: lane u finishes pixel u (sigmoid + store)
      if (cpp >= UNR) {
        float z = sv[0];
#pragma unroll
        for (int u = 1; u < UNR; ++u) z = (cl == u) ? sv[u] : z;
        if (cl < UNR) {
          const long px = pb + cl * ppb + pl;
          if (px < P) {
            const long n = px / HW, rem = px - n * HW;
            p.probs[(n * p.OC + oc) * HW + rem] = 1.f / (1.f + expf(-(z + bias)));
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (pix[u] < P && cl == 0) {
            const long n = pix[u] / HW, rem = pix[u] - n * HW;
            p.probs[(n * p.OC + oc) * HW + rem] = 1.f / (1.f + expf(-(sv[u] + bias)));
          }
      }
    }
  }
}

// dz = dp * p * (1-p);  dA[pix][c] = sum_oc dz[oc]*w[oc][c];  partial sums of dz*a (dW) and dz (db)
// parts layout: [gridDim.x][OC][C + 1]   (last column = bias gradient)
// LEAN (the training path of the networks: one output channel, activation recomputed from y, input gradient not stored): the
// saved output stays a RAW chunk and each channel is taken out of it in turn -- 158 -> fewer registers, i.e. every workgroup of
// the 1024 resident at once instead of three quarters of them.  Same arithmetic, same order.
template <typename T, bool BN, bool LEAN = false>
__global__ __launch_bounds__(256) void head_bwd_kernel(const HeadParams p) {
  static_assert(!LEAN || BN, "lean form: fused BatchNorm-backward sums");
  constexpr int EPC = Chunk<T>::N;
  __shared__ float red[256 * (EPC + 1)];
  __shared__ float red3[BN ? 256 * 3 * EPC : 1];
  const int cpp = p.C / EPC, ppb = 256 / cpp;
  const int tid = threadIdx.x, cl = tid % cpp, pl = tid / cpp;
  const long HW = (long)p.H * p.W, P = (long)p.N * HW;
  const T* __restrict__ ag = reinterpret_cast<const T*>(p.a);
  T* __restrict__ dag = reinterpret_cast<T*>(p.da);
  // BN: the gradient written here is the activation gradient of the stage that produced `a`; its BatchNorm-backward
  // sums (S1 = sum dyhat, S2 = sum dyhat*xhat, S3 = sum xhat: see bn_bwd_kernel) are taken from the value as STORED
  // (rounded through the storage type), so the stand-alone reduction pass over da and y is not needed.
  const T* __restrict__ yg = reinterpret_cast<const T*>(p.bn_y);
  float sc[EPC], sh[EPC], mu[EPC], rs[EPC], s1[EPC], s2[EPC], s3[EPC];
  if (BN) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int c = cl * EPC + e;
      sc[e] = p.bn_scale[c]; sh[e] = p.bn_shift[c]; mu[e] = p.bn_mean[c]; rs[e] = p.bn_rstd[c];
      s1[e] = s2[e] = s3[e] = 0.f;
    }
  }
  for (int oc = 0; oc < p.OC; ++oc) {
    const bool last = oc == p.OC - 1;
    float wv[EPC], gw[EPC], gb = 0.f;
#pragma unroll
    for (int e = 0; e < EPC; ++e) { wv[e] = p.w[oc * p.C + cl * EPC + e]; gw[e] = 0.f; }
    // two pixels per lane group and trip: every load of both is issued before the arithmetic of the first (one 16-byte load
    // in flight per lane at 3 waves per SIMD left the kernel latency bound at 3.4 TB/s)
    constexpr int U = 2;
    if (LEAN) {
      for (long pb = (long)blockIdx.x * ppb * U; pb < P; pb += (long)gridDim.x * ppb * U) {
        u32x4 yraw[U];
        float pr[U], dpv[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long px0 = pb + u * ppb + pl;
          ok[u] = px0 < P;
          const long px = ok[u] ? px0 : 0;
          pr[u] = p.probs[px];                             // OC == 1: [N, 1, H, W] is the pixel index itself
          dpv[u] = p.dprobs[px];
          yraw[u] = ld16(yg + px * p.bn_ldy + cl * EPC);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!ok[u]) continue;
          float dz = dpv[u] * pr[u] * (1.f - pr[u]);
          asm volatile("" : "+v"(dz));                     // a rounded product, as in the stored form: no contraction into `gb += dz`
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            const float yv = chunk_elem<T>(yraw[u], e);
            const float nrm = fmaf(yv, sc[e], sh[e]);
            const float av = round_through<T>(fmaxf(nrm, 0.f));
            gw[e] = fmaf(dz, av, gw[e]);
            const float g = round_through<T>(0.f + dz * wv[e]);
            const float gh = nrm > 0.f ? g : 0.f;
            const float xh = (yv - mu[e]) * rs[e];
            s1[e] += gh; s2[e] = fmaf(gh, xh, s2[e]); s3[e] += xh;
          }
          if (cl == 0) gb += dz;
        }
      }
    } else
    for (long pb = (long)blockIdx.x * ppb * U; pb < P; pb += (long)gridDim.x * ppb * U) {
      long pix[U];
      bool ok[U];
      u32x4 yraw[U], araw[U], draw[U];
      float pr[U], dpv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        pix[u] = pb + u * ppb + pl;
        ok[u] = pix[u] < P;
        const long px = ok[u] ? pix[u] : 0;
        const long n = px / HW, rem = px - n * HW;
        const long o = (n * p.OC + oc) * HW + rem;
        pr[u] = p.probs[o];
        dpv[u] = p.dprobs[o];
        if (BN && (last || !ag)) yraw[u] = ld16(yg + px * p.bn_ldy + cl * EPC);
        if (!BN || ag) araw[u] = ld16(ag + px * p.lda + cl * EPC);
        if (oc > 0) draw[u] = ld16(dag + px * p.ldda + cl * EPC);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;
        const float dz = dpv[u] * pr[u] * (1.f - pr[u]);
        float av[EPC], d[EPC], yv[EPC];
        if (BN && (last || !ag)) Chunk<T>::unpack(yraw[u], yv);
        if (!BN || ag) Chunk<T>::unpack(araw[u], av);
        else {                                           // the activation as the forward pass saw it, from the saved conv output
#pragma unroll
          for (int e = 0; e < EPC; ++e) av[e] = round_through<T>(fmaxf(fmaf(yv[e], sc[e], sh[e]), 0.f));
        }
        if (oc > 0) Chunk<T>::unpack(draw[u], d);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          gw[e] = fmaf(dz, av[e], gw[e]);
          d[e] = (oc > 0 ? d[e] : 0.f) + dz * wv[e];
        }
        if (cl == 0) gb += dz;
        if (dag) st16(dag + pix[u] * p.ldda + cl * EPC, Chunk<T>::pack(d));     // null: the consumer recomputes it (BnBwdParams::head_w)
        if (BN && last) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            const float g = round_through<T>(d[e]);
            const float nrm = fmaf(yv[e], sc[e], sh[e]);
            const float gh = nrm > 0.f ? g : 0.f;
            const float xh = (yv[e] - mu[e]) * rs[e];
            s1[e] += gh; s2[e] = fmaf(gh, xh, s2[e]); s3[e] += xh;
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[tid * (EPC + 1) + e] = gw[e];
    red[tid * (EPC + 1) + EPC] = gb;
    __syncthreads();
    for (int i = tid; i < p.C + 1; i += 256) {
      float s = 0.f;
      if (i < p.C) {
        const int c2 = i / EPC, e = i % EPC;
        for (int q = 0; q < ppb; ++q) s += red[(q * cpp + c2) * (EPC + 1) + e];
      } else {
        for (int q = 0; q < ppb; ++q) s += red[(q * cpp) * (EPC + 1) + EPC];
      }
      p.parts[((long)blockIdx.x * p.OC + oc) * (p.C + 1) + i] = s;
    }
    __syncthreads();
  }
  if (BN) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      red3[(tid * 3 + 0) * EPC + e] = s1[e];
      red3[(tid * 3 + 1) * EPC + e] = s2[e];
      red3[(tid * 3 + 2) * EPC + e] = s3[e];
    }
    __syncthreads();
    for (int i = tid; i < 3 * p.C; i += 256) {             // (which, channel): pixel lanes summed in lane order
      const int which = i / p.C, c = i - which * p.C, c2 = c / EPC, e = c - c2 * EPC;
      float s = 0.f;
      for (int q = 0; q < ppb; ++q) s += red3[((q * cpp + c2) * 3 + which) * EPC + e];
      p.bn_parts[((long)blockIdx.x * 3 + which) * p.C + c] = s;
    }
  }
}

__global__ __launch_bounds__(256) void head_bwd_finalize_kernel(const float* __restrict__ parts, int nparts, float* dw,
                                                                float* db, int OC, int C) {
  // 32 lanes per output element, rows l, l+32, ... each, fixed shuffle tree
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  const int L = OC * (C + 1);
  double s = 0.0;
  if (i < L)
    for (int q = l; q < nparts; q += 32) s += (double)parts[(long)q * L + i];
  s = lane32_sum(s);
  if (i >= L || l != 0) return;
  const int oc = i / (C + 1), c = i - oc * (C + 1);
  if (c < C) dw[oc * C + c] = (float)s;
  else db[oc] = (float)s;
}

// ================================================================================================
// per-channel sum over pixels of an NHWC tensor (ConvTranspose2d bias gradient)
// parts layout: [gridDim.x][C]
// ================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const void* __restrict__ x, int ldx, float* __restrict__ parts,
                                                          long P, int C) {
  constexpr int EPC = Chunk<T>::N;
  __shared__ float red[256 * EPC];
  const int cpp = C / EPC;
  const int tid = threadIdx.x;
  const int seg = (cpp < 256) ? cpp : 256, plane = 256 / seg;
  const int chl = tid % seg, pl = tid / seg;
  const int ch = blockIdx.y * seg + chl;
  const T* __restrict__ xg = reinterpret_cast<const T*>(x);
  float s[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) s[e] = 0.f;
  if (ch < cpp)
    for (long q = (long)blockIdx.x * plane + pl; q < P; q += (long)gridDim.x * plane) {
      float v[EPC];
      Chunk<T>::unpack(ld16(xg + q * ldx + ch * EPC), v);
#pragma unroll
      for (int e = 0; e < EPC; ++e) s[e] += v[e];
    }
#pragma unroll
  for (int e = 0; e < EPC; ++e) red[tid * EPC + e] = s[e];
  __syncthreads();
  for (int i = tid; i < seg * EPC; i += 256) {
    const int cl = i / EPC, e = i - cl * EPC;
    const int chn = blockIdx.y * seg + cl;
    if (chn >= cpp) continue;
    float t = 0.f;
    for (int q2 = 0; q2 < plane; ++q2) t += red[(q2 * seg + cl) * EPC + e];
    parts[(long)blockIdx.x * C + chn * EPC + e] = t;
  }
}

__global__ void channel_sum_finalize_kernel(const float* __restrict__ parts, int nparts, float* out, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int i = 0; i < nparts; ++i) s += (double)parts[(long)i * C + c];
  out[c] = (float)s;
}

// ================================================================================================
// Weight packing (fp32 PyTorch layouts -> K-contiguous "[tap][out][in]" images in the compute type)
// ================================================================================================
template <typename T>
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Co, int Ci) {
  const long n = (long)Co * Ci * 9;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    // i indexes the forward image [t][co][ci]
    const int ci = (int)(i % Ci);
    const long r = i / Ci;
    const int co = (int)(r % Co), t = (int)(r / Co);
    const float v = w[((long)co * Ci + ci) * 9 + t];
    wf[i] = from_f32<T>(v);
    // dgrad image [t'][ci][co] with t' = 8 - t (both spatial axes flipped)
    if (wd) wd[((long)(8 - t) * Ci + ci) * Co + co] = from_f32<T>(v);
  }
}

template <typename T>
__global__ void pack_convT2x2_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Ci, int Co) {
  const long n = (long)Ci * Co * 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    // i indexes the forward image [(ab)*Co + co][ci]
    const int ci = (int)(i % Ci);
    const long r = i / Ci;
    const int co = (int)(r % Co), ab = (int)(r / Co);
    const float v = w[((long)ci * Co + co) * 4 + ab];
    wf[i] = from_f32<T>(v);
    if (wd) wd[((long)ab * Ci + ci) * Co + co] = from_f32<T>(v);    // dgrad image [ab][ci][co]
  }
}

// All layers in ONE launch, as an LDS-tiled transpose: a device-resident table describes each tensor
// (21 per network); one workgroup handles a 32 x 32 (out x in) channel tile with all its taps, reads
// the fp32 parameter in contiguous 1152/512-byte rows and writes both K-contiguous images in >= 64-byte
// runs.  (Element-wise scattering of the transposed dgrad image cost 0.3 ms per step.)
struct PackDesc {
  const float* w;      // fp32 parameter, PyTorch layout
  void* wf;            // forward image
  void* wd;            // dgrad image (nullable)
  long begin;          // index of this tensor's first 32x32 tile in the launch
  int a, b;            // conv3x3: (Cout, Cin), taps 9;  convT: (Cin, Cout), taps 4
  int kind;            // 0 = conv3x3, 1 = convT2x2
  int pad;
};

template <typename T>
__global__ __launch_bounds__(256) void pack_many_kernel(const PackDesc* __restrict__ table, int n) {
  __shared__ float tile[32 * 289];
  const int tid = threadIdx.x;
  const long tix = blockIdx.x;
  int lo = 0, hi = n - 1;                      // last descriptor with begin <= tix
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].begin <= tix) lo = mid; else hi = mid - 1;
  }
  const PackDesc d = table[lo];
  const int local = (int)(tix - d.begin);
  T* wf = reinterpret_cast<T*>(d.wf);
  T* wd = reinterpret_cast<T*>(d.wd);
  if (d.kind == 0) {
    const int Co = d.a, Ci = d.b, tci = Ci / 32;
    const int co0 = (local / tci) * 32, ci0 = (local % tci) * 32;
    for (int i = tid; i < 32 * 288; i += 256) {            // w[co][ci0 .. ci0+31][9]: 288 contiguous floats
      const int co = i / 288, rem = i - co * 288;
      tile[co * 289 + rem] = d.w[((long)(co0 + co) * Ci + ci0) * 9 + rem];
    }
    __syncthreads();
    for (int i = tid; i < 9 * 1024; i += 256) {
      const int t = i >> 10, x = (i >> 5) & 31, y = i & 31;
      wf[((long)t * Co + co0 + x) * Ci + ci0 + y] = from_f32<T>(tile[x * 289 + y * 9 + t]);            // x = co, y = ci
      if (wd) wd[((long)(8 - t) * Ci + ci0 + x) * Co + co0 + y] = from_f32<T>(tile[y * 289 + x * 9 + t]);  // x = ci, y = co
    }
  } else {
    const int Ci = d.a, Co = d.b, tco = Co / 32;
    const int ci0 = (local / tco) * 32, co0 = (local % tco) * 32;
    for (int i = tid; i < 32 * 128; i += 256) {            // w[ci][co0 .. co0+31][4]: 128 contiguous floats
      const int ci = i >> 7, rem = i & 127;
      tile[ci * 129 + rem] = d.w[((long)(ci0 + ci) * Co + co0) * 4 + rem];
    }
    __syncthreads();
    for (int i = tid; i < 4 * 1024; i += 256) {
      const int ab = i >> 10, x = (i >> 5) & 31, y = i & 31;
      wf[((long)ab * Co + co0 + x) * Ci + ci0 + y] = from_f32<T>(tile[y * 129 + x * 4 + ab]);              // x = co, y = ci
      if (wd) wd[((long)ab * Ci + ci0 + x) * Co + co0 + y] = from_f32<T>(tile[x * 129 + y * 4 + ab]);        // x = ci, y = co
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static int grid_for(long work_items, int per_block) {
  long nb = (work_items + per_block - 1) / per_block;
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  return (int)nb;
}

int launch_bn_finalize(const float* parts, int nparts, long count, const float* gamma, const float* beta, float eps,
                       float momentum, float* rm, float* rv, float* scale, float* shift, float* mean, float* rstd,
                       int C, hipStream_t stream) {
  UNETDC_REQUIRE(parts && gamma && beta && scale && shift && mean && rstd, "bn_finalize: null pointer");
  UNETDC_REQUIRE(nparts > 0 && count > 0 && C > 0, "bn_finalize: empty problem");
  const float* rp; int rows;
  int rc = reduce_parts(parts, nparts, 2 * C, &rp, &rows, stream, 512);
  if (rc != UNETDC_OK) return rc;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + FIN_C - 1) / FIN_C), dim3(256), 0, stream, rp, rows, (double)count, gamma,
                     beta, eps, momentum, rm, rv, scale, shift, mean, rstd, C);
  return check_launch("bn_finalize_kernel");
}

// BatchNorm with FROZEN statistics under autograd (model.eval() with gradients enabled): the training-path kernels run with
// mean / rstd taken from the running buffers instead of the batch; the conv bias is added by the conv epilogue as in training
__global__ void bn_frozen_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                        const float* __restrict__ rm, const float* __restrict__ rv, float eps, float* scale,
                                        float* shift, float* mean, float* rstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float r = 1.f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * r;
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
  mean[c] = rm[c];
  rstd[c] = r;
}

int launch_bn_frozen_affine(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* scale,
                            float* shift, float* mean, float* rstd, int C, hipStream_t stream) {
  UNETDC_REQUIRE(gamma && beta && rm && rv && scale && shift && mean && rstd, "bn_frozen_affine: null pointer");
  hipLaunchKernelGGL(bn_frozen_affine_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, gamma, beta, rm, rv, eps, scale, shift,
                     mean, rstd, C);
  return check_launch("bn_frozen_affine_kernel");
}

int launch_bn_eval_affine(const float* gamma, const float* beta, const float* rm, const float* rv,
                          const float* conv_bias, float eps, float* scale, float* shift, int C, hipStream_t stream) {
  UNETDC_REQUIRE(gamma && beta && rm && rv && scale && shift, "bn_eval_affine: null pointer");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, gamma, beta, rm, rv, conv_bias,
                     eps, scale, shift, C);
  return check_launch("bn_eval_affine_kernel");
}

int launch_apply(ApplyParams& p, int dtype, hipStream_t stream) {
  const int epc = dtype == UNETDC_BF16 ? 8 : 4;
  UNETDC_REQUIRE(dtype == UNETDC_F32 || dtype == UNETDC_BF16, "bn_relu_apply: bad dtype %d", dtype);
  UNETDC_REQUIRE(p.y, "bn_relu_apply: null input");
  UNETDC_REQUIRE(p.C % epc == 0 && p.ldy % epc == 0, "bn_relu_apply: C/ld not chunk aligned");
  UNETDC_REQUIRE((p.scale == nullptr) == (p.shift == nullptr), "bn_relu_apply: scale/shift mismatch");
  const bool pool = p.pooled != nullptr;
  if (pool) UNETDC_REQUIRE(p.H % 2 == 0 && p.W % 2 == 0 && p.ldp % epc == 0, "bn_relu_apply: pooling needs even H, W");
  if (!pool) UNETDC_REQUIRE(p.a != nullptr, "bn_relu_apply: nothing to write");
  if (p.a) UNETDC_REQUIRE(p.lda % epc == 0, "bn_relu_apply: lda not chunk aligned");
  const long items = (long)p.N * (pool ? p.H / 2 : p.H) * (pool ? p.W / 2 : p.W) * (p.C / epc);
  const int nb = grid_for(items, 256);
  if (dtype == UNETDC_BF16) {
    if (pool) hipLaunchKernelGGL((bn_relu_apply_kernel<bf16_t, true>), dim3(nb), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((bn_relu_apply_kernel<bf16_t, false>), dim3(nb), dim3(256), 0, stream, p);
  } else {
    if (pool) hipLaunchKernelGGL((bn_relu_apply_kernel<float, true>), dim3(nb), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((bn_relu_apply_kernel<float, false>), dim3(nb), dim3(256), 0, stream, p);
  }
  return check_launch("bn_relu_apply_kernel");
}

// workspace floats needed by launch_bn_bwd: (nblk + 64) * 3 * C + 3 * C
static int bn_bwd_blocks(long Q, int cpp) {
  const int seg = cpp < 256 ? cpp : 256;
  const int plane = 256 / seg;
  long nb = (Q + (long)plane * 8 - 1) / ((long)plane * 8);     // >= 8 items per pixel lane
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  return (int)nb;
}

long bn_bwd_workspace_bytes(int N, int H, int W, int C, int pooled, int dtype) {
  const int epc = dtype == UNETDC_BF16 ? 8 : 4;
  const long Q = (long)N * (pooled ? H / 2 : H) * (pooled ? W / 2 : W);
  const int nb = bn_bwd_blocks(Q, C / epc);
  return ((long)(nb + 64) * 3 * C + 3 * C) * 4;
}

template <typename T>
static void launch_bn_bwd_k(const BnBwdParams& p, bool pool, bool apply, dim3 grid, hipStream_t stream) {
  if (pool) {
    if (apply) hipLaunchKernelGGL((bn_bwd_kernel<T, true, true>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((bn_bwd_kernel<T, true, false>), grid, dim3(256), 0, stream, p);
  } else {
    if (apply) hipLaunchKernelGGL((bn_bwd_kernel<T, false, true>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((bn_bwd_kernel<T, false, false>), grid, dim3(256), 0, stream, p);
  }
}

// Reduction pass only: fills `parts` ([nparts][3][C]) -- used when the producer kernel could not fuse it.
int launch_bn_bwd_reduce_only(BnBwdParams& p, float* parts, long parts_floats, int* nparts, int dtype,
                              hipStream_t stream) {
  const int epc = dtype == UNETDC_BF16 ? 8 : 4;
  UNETDC_REQUIRE(p.y && p.dskip && parts && nparts, "bn_bwd_reduce: null pointer");
  const int cpp = p.C / epc;
  const int seg = cpp < 256 ? cpp : 256;
  UNETDC_REQUIRE(p.C % epc == 0 && 256 % seg == 0, "bn_bwd_reduce: C=%d unsupported", p.C);
  const long Q = (long)p.N * p.H * p.W;
  // the caller sized `parts` for the fused-epilogue producer (one row per 256 pixels + 64 spare rows); this
  // stand-alone pass (grid-stride over pixels) takes as many workgroups as that buffer has rows for
  int nb = bn_bwd_blocks(Q, cpp);
  const long cap = parts_floats / (3L * p.C) - 64;
  if (nb > cap) nb = (int)cap;
  UNETDC_REQUIRE(nb >= 1, "bn_bwd_reduce: partial buffer too small");
  p.parts = parts;
  p.dpool = nullptr;
  const dim3 grid(nb, (cpp + seg - 1) / seg);
  if (dtype == UNETDC_BF16) launch_bn_bwd_k<bf16_t>(p, false, false, grid, stream);
  else launch_bn_bwd_k<float>(p, false, false, grid, stream);
  *nparts = nb;
  return check_launch("bn_bwd_kernel(reduce)");
}

int launch_bn_bwd(BnBwdParams& p, const float* gamma, float* dgamma, float* dbeta, float* dbias, void* workspace,
                  long workspace_bytes, const float* pre_parts, int pre_nparts, int dtype, hipStream_t stream, bool frozen) {
  const int epc = dtype == UNETDC_BF16 ? 8 : 4;
  UNETDC_REQUIRE(dtype == UNETDC_F32 || dtype == UNETDC_BF16, "bn_bwd: bad dtype %d", dtype);
  UNETDC_REQUIRE(p.y && p.dy && (p.dskip || p.dpool || p.head_w), "bn_bwd: null tensor");
  if (p.head_w)
    UNETDC_REQUIRE(!p.dskip && !p.dpool && p.head_dprobs && p.head_probs && pre_parts,
                   "bn_bwd (head): needs dprobs, probs and the sums from unetdc_head_bwd_bnstats, and no stored gradient");
  UNETDC_REQUIRE(p.scale && p.shift && p.mean && p.rstd && gamma && dgamma && dbeta && workspace, "bn_bwd: null pointer");
  UNETDC_REQUIRE(p.C % epc == 0 && p.ldy % epc == 0 && p.lddy % epc == 0, "bn_bwd: C/ld not chunk aligned");
  const bool pool = p.dpool != nullptr;
  if (pool) UNETDC_REQUIRE(p.H % 2 == 0 && p.W % 2 == 0 && p.ldp % epc == 0, "bn_bwd: pooling needs even H, W");
  if (!pool) UNETDC_REQUIRE(p.dskip != nullptr || p.head_w != nullptr, "bn_bwd: gradient missing");
  if (p.dskip) UNETDC_REQUIRE(p.lds % epc == 0, "bn_bwd: lds not chunk aligned");
  if (pre_parts) UNETDC_REQUIRE(!pool && pre_nparts > 0, "bn_bwd: precomputed partial sums need the non-pooled form");
  const int cpp = p.C / epc;
  const int seg = cpp < 256 ? cpp : 256;
  UNETDC_REQUIRE(256 % seg == 0, "bn_bwd: C=%d unsupported (C/%d must divide 256 or be a multiple of 256)", p.C, epc);
  const long Q = (long)p.N * (pool ? p.H / 2 : p.H) * (pool ? p.W / 2 : p.W);
  const int nb = bn_bwd_blocks(Q, cpp);
  const long need = ((long)(nb + 64) * 3 * p.C + 3 * p.C) * 4;
  if (need > workspace_bytes) {
    set_error("bn_bwd: workspace too small (%ld < %ld bytes)", workspace_bytes, need);
    return UNETDC_EWORKSPACE;
  }
  float* ws = reinterpret_cast<float*>(workspace);
  float* k = ws;                         // k1,k2,k3
  float* parts = ws + 3 * p.C;
  p.parts = parts;
  p.k1 = k; p.k2 = k + p.C; p.k3 = k + 2 * p.C;
  const dim3 grid(nb, (cpp + seg - 1) / seg);
  int rc;
  const float* rp; int rows;
  if (pre_parts) {
    // the reduction was fused into the epilogue of the kernel that produced the gradient
    rc = reduce_parts(pre_parts, pre_nparts, 3 * p.C, &rp, &rows, stream, 512);
  } else {
    if (dtype == UNETDC_BF16) launch_bn_bwd_k<bf16_t>(p, pool, false, grid, stream);
    else launch_bn_bwd_k<float>(p, pool, false, grid, stream);
    rc = check_launch("bn_bwd_kernel(reduce)");
    if (rc != UNETDC_OK) return rc;
    rc = reduce_parts(parts, nb, 3 * p.C, &rp, &rows, stream, 512);
  }
  if (rc != UNETDC_OK) return rc;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((p.C + FIN_C - 1) / FIN_C), dim3(256), 0, stream, rp, rows, p.C, 0,
                     (const float*)nullptr, 0, (double)p.N * p.H * p.W, gamma, p.rstd, dgamma, dbeta, dbias, k, k + p.C,
                     k + 2 * p.C, p.C, frozen ? 1 : 0);
  rc = check_launch("bn_bwd_finalize_kernel");
  if (rc != UNETDC_OK) return rc;
  const long items = Q * cpp;
  const dim3 grid2(grid_for(items, 256) > 4096 ? 4096 : grid_for(items, 256), (cpp + seg - 1) / seg);
  if (p.head_w) {
    if (dtype == UNETDC_BF16) hipLaunchKernelGGL((bn_bwd_kernel<bf16_t, false, true, true>), grid2, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((bn_bwd_kernel<float, false, true, true>), grid2, dim3(256), 0, stream, p);
    return check_launch("bn_bwd_kernel(apply, head)");
  }
  if (dtype == UNETDC_BF16) launch_bn_bwd_k<bf16_t>(p, pool, true, grid2, stream);
  else launch_bn_bwd_k<float>(p, pool, true, grid2, stream);
  return check_launch("bn_bwd_kernel(apply)");
}

// Finalisation alone: dgamma / dbeta / dbias and the three per-channel coefficients of the apply arithmetic
// (dy = k1 * dyhat - k2 - k3 * xhat) into coeffs[3][C], from sums a producer kernel left as partial rows -- for a consumer
// that applies them itself (first-layer weight gradient, first_conv.hip)
int launch_bn_bwd_coeffs(const float* pre_parts, int pre_nparts, long count, const float* gamma, const float* rstd, float* dgamma,
                         float* dbeta, float* dbias, float* coeffs, int C, hipStream_t stream) {
  UNETDC_REQUIRE(pre_parts && pre_nparts > 0 && gamma && rstd && dgamma && dbeta && coeffs && C > 0 && count > 0,
                 "bn_bwd_coeffs: null pointer / empty problem");
  const float* rp; int rows;
  int rc = reduce_parts(pre_parts, pre_nparts, 3 * C, &rp, &rows, stream, 512);
  if (rc != UNETDC_OK) return rc;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FIN_C - 1) / FIN_C), dim3(256), 0, stream, rp, rows, C, 0,
                     (const float*)nullptr, 0, (double)count, gamma, rstd, dgamma, dbeta, dbias, coeffs, coeffs + C, coeffs + 2 * C,
                     C, 0);
  return check_launch("bn_bwd_finalize_kernel");
}

static int head_blocks(long P, int cpp) {
  long nb = (P + (256 / cpp) * 8 - 1) / ((256 / cpp) * 8);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return (int)nb;
}

long head_bwd_workspace_bytes(int N, int H, int W, int C, int OC, int dtype) {
  const int epc = dtype == UNETDC_BF16 ? 8 : 4;
  return ((long)head_blocks((long)N * H * W, C / epc) + 64) * OC * (C + 1) * 4;
}

static int check_head(const HeadParams& p, int dtype) {
  const int epc = dtype == UNETDC_BF16 ? 8 : 4;
  UNETDC_REQUIRE(dtype == UNETDC_F32 || dtype == UNETDC_BF16, "head: bad dtype %d", dtype);
  UNETDC_REQUIRE((p.a || p.bn_y) && p.w && p.probs, "head: null pointer");
  const int cpp = p.C / epc;
  UNETDC_REQUIRE(p.C % epc == 0 && cpp >= 1 && cpp <= 64 && (cpp & (cpp - 1)) == 0,
                 "head: C=%d unsupported (C/%d must be a power of two <= 64)", p.C, epc);
  UNETDC_REQUIRE(p.lda % epc == 0 && p.OC >= 1, "head: bad lda/OC");
  return UNETDC_OK;
}

int launch_head_fwd(HeadParams& p, int dtype, hipStream_t stream) {
  int rc = check_head(p, dtype);
  if (rc != UNETDC_OK) return rc;
  UNETDC_REQUIRE(p.b != nullptr, "head: null bias");
  const int epc = dtype == UNETDC_BF16 ? 8 : 4;
  const int nb = grid_for((long)p.N * p.H * p.W, 4 * (256 / (p.C / epc)));     // UNR = 4 pixels per lane group and trip
  const bool norm = p.bn_scale != nullptr;             // BatchNorm + ReLU of the producing stage applied on load
  if (norm) UNETDC_REQUIRE(p.bn_shift != nullptr, "head_fwd_bn: null shift");
  // C = 64 (the networks' head): 8 (bf16) / 16 (fp32) lanes per pixel known at compile time; other widths: generic form
  const bool c64 = p.C == 64;
  if (dtype == UNETDC_BF16) {
    if (norm && c64) hipLaunchKernelGGL((head_fwd_kernel<bf16_t, true, 8>), dim3(nb), dim3(256), 0, stream, p);
    else if (norm) hipLaunchKernelGGL((head_fwd_kernel<bf16_t, true, 0>), dim3(nb), dim3(256), 0, stream, p);
    else if (c64) hipLaunchKernelGGL((head_fwd_kernel<bf16_t, false, 8>), dim3(nb), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((head_fwd_kernel<bf16_t, false, 0>), dim3(nb), dim3(256), 0, stream, p);
  } else {
    if (norm && c64) hipLaunchKernelGGL((head_fwd_kernel<float, true, 16>), dim3(nb), dim3(256), 0, stream, p);
    else if (norm) hipLaunchKernelGGL((head_fwd_kernel<float, true, 0>), dim3(nb), dim3(256), 0, stream, p);
    else if (c64) hipLaunchKernelGGL((head_fwd_kernel<float, false, 16>), dim3(nb), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((head_fwd_kernel<float, false, 0>), dim3(nb), dim3(256), 0, stream, p);
  }
  return check_launch("head_fwd_kernel");
}

int launch_head_bwd(HeadParams& p, float* dw, float* db, void* workspace, long workspace_bytes, int dtype,
                    hipStream_t stream, int* bn_nparts, long bn_parts_floats) {
  int rc = check_head(p, dtype);
  if (rc != UNETDC_OK) return rc;
  UNETDC_REQUIRE(p.dprobs && dw && db && workspace, "head_bwd: null pointer");
  const int epc = dtype == UNETDC_BF16 ? 8 : 4;
  // da == NULL: the gradient of the head's input is not stored (one output channel, fused BatchNorm-backward sums: the
  // stage's backward pass recomputes it, unetdc_bn_relu_bwd_head)
  UNETDC_REQUIRE(p.da || (p.OC == 1 && p.bn_y), "head_bwd: da may be NULL only with one output channel and the fused sums");
  UNETDC_REQUIRE(!p.da || p.ldda % epc == 0, "head_bwd: ldda not chunk aligned");
  int nb = head_blocks((long)p.N * p.H * p.W, p.C / epc);
  const bool bn = p.bn_y != nullptr;
  UNETDC_REQUIRE(bn || p.a, "head_bwd: null activation");
  if (bn) {
    UNETDC_REQUIRE(p.bn_scale && p.bn_shift && p.bn_mean && p.bn_rstd && p.bn_parts && bn_nparts,
                   "head_bwd_bnstats: null pointer");
    UNETDC_REQUIRE(p.bn_ldy % epc == 0 && p.bn_ldy >= p.C, "head_bwd_bnstats: ldy not chunk aligned");
    // the caller sized `parts` for a convolution epilogue (one row per 256 pixels + 64 spare rows): the grid-stride loop
    // of this kernel takes as many workgroups as that buffer has rows for
    const long cap = bn_parts_floats / (3L * p.C) - 64;
    if (nb > cap) nb = (int)cap;
    UNETDC_REQUIRE(nb >= 1, "head_bwd_bnstats: partial buffer too small");
    *bn_nparts = nb;
  }
  const long need = ((long)nb + 64) * p.OC * (p.C + 1) * 4;
  if (need > workspace_bytes) {
    set_error("head_bwd: workspace too small (%ld < %ld bytes)", workspace_bytes, need);
    return UNETDC_EWORKSPACE;
  }
  p.parts = reinterpret_cast<float*>(workspace);
  const bool lean = bn && !p.a && !p.da && p.OC == 1;
  if (dtype == UNETDC_BF16) {
    if (lean) hipLaunchKernelGGL((head_bwd_kernel<bf16_t, true, true>), dim3(nb), dim3(256), 0, stream, p);
    else if (bn) hipLaunchKernelGGL((head_bwd_kernel<bf16_t, true>), dim3(nb), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((head_bwd_kernel<bf16_t, false>), dim3(nb), dim3(256), 0, stream, p);
  } else {
    if (lean) hipLaunchKernelGGL((head_bwd_kernel<float, true, true>), dim3(nb), dim3(256), 0, stream, p);
    else if (bn) hipLaunchKernelGGL((head_bwd_kernel<float, true>), dim3(nb), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((head_bwd_kernel<float, false>), dim3(nb), dim3(256), 0, stream, p);
  }
  rc = check_launch("head_bwd_kernel");
  if (rc != UNETDC_OK) return rc;
  const int L = p.OC * (p.C + 1);
  const float* rp; int rows;
  rc = reduce_parts(p.parts, nb, L, &rp, &rows, stream);
  if (rc != UNETDC_OK) return rc;
  hipLaunchKernelGGL(head_bwd_finalize_kernel, dim3((L + 7) / 8), dim3(256), 0, stream, rp, rows, dw, db, p.OC, p.C);
  return check_launch("head_bwd_finalize_kernel");
}

int launch_pack_conv3x3(const float* w, void* wf, void* wd, int Co, int Ci, int dtype, hipStream_t stream) {
  UNETDC_REQUIRE(w && wf, "pack_conv3x3: null pointer");
  UNETDC_REQUIRE(dtype == UNETDC_F32 || dtype == UNETDC_BF16, "pack_conv3x3: bad dtype %d", dtype);
  const int nb = grid_for((long)Co * Ci * 9, 256);
  if (dtype == UNETDC_BF16)
    hipLaunchKernelGGL(pack_conv3x3_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, w, (bf16_t*)wf, (bf16_t*)wd, Co, Ci);
  else
    hipLaunchKernelGGL(pack_conv3x3_kernel<float>, dim3(nb), dim3(256), 0, stream, w, (float*)wf, (float*)wd, Co, Ci);
  return check_launch("pack_conv3x3_kernel");
}

int launch_pack_convT2x2(const float* w, void* wf, void* wd, int Ci, int Co, int dtype, hipStream_t stream) {
  UNETDC_REQUIRE(w && wf, "pack_convT2x2: null pointer");
  UNETDC_REQUIRE(dtype == UNETDC_F32 || dtype == UNETDC_BF16, "pack_convT2x2: bad dtype %d", dtype);
  const int nb = grid_for((long)Co * Ci * 4, 256);
  if (dtype == UNETDC_BF16)
    hipLaunchKernelGGL(pack_convT2x2_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, w, (bf16_t*)wf, (bf16_t*)wd, Ci, Co);
  else
    hipLaunchKernelGGL(pack_convT2x2_kernel<float>, dim3(nb), dim3(256), 0, stream, w, (float*)wf, (float*)wd, Ci, Co);
  return check_launch("pack_convT2x2_kernel");
}

static int channel_sum_blocks(long P) {
  long nb = (P + 63) / 64;
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  return (int)nb;
}

long channel_sum_workspace_bytes(long P, int C) { return ((long)channel_sum_blocks(P) + 64) * C * 4; }

int launch_channel_sum(const void* x, int ldx, float* out, void* workspace, long workspace_bytes, long P, int C,
                       int dtype, hipStream_t stream) {
  const int epc = dtype == UNETDC_BF16 ? 8 : 4;
  UNETDC_REQUIRE(dtype == UNETDC_F32 || dtype == UNETDC_BF16, "channel_sum: bad dtype %d", dtype);
  UNETDC_REQUIRE(x && out && workspace && P > 0, "channel_sum: null pointer / empty");
  UNETDC_REQUIRE(C % epc == 0 && ldx % epc == 0, "channel_sum: C/ld not chunk aligned");
  const int cpp = C / epc, seg = cpp < 256 ? cpp : 256;
  UNETDC_REQUIRE(256 % seg == 0, "channel_sum: C=%d unsupported", C);
  const int nb = channel_sum_blocks(P);
  const long need = ((long)nb + 64) * C * 4;
  if (need > workspace_bytes) {
    set_error("channel_sum: workspace too small (%ld < %ld bytes)", workspace_bytes, need);
    return UNETDC_EWORKSPACE;
  }
  float* parts = reinterpret_cast<float*>(workspace);
  const dim3 grid(nb, (cpp + seg - 1) / seg);
  if (dtype == UNETDC_BF16) hipLaunchKernelGGL(channel_sum_kernel<bf16_t>, grid, dim3(256), 0, stream, x, ldx, parts, P, C);
  else hipLaunchKernelGGL(channel_sum_kernel<float>, grid, dim3(256), 0, stream, x, ldx, parts, P, C);
  int rc = check_launch("channel_sum_kernel");
  if (rc != UNETDC_OK) return rc;
  const float* rp; int rows;
  rc = reduce_parts(parts, nb, C, &rp, &rows, stream);
  if (rc != UNETDC_OK) return rc;
  hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, rp, rows, out, C);
  return check_launch("channel_sum_finalize_kernel");
}

// Column sums out of MODE_STATS partial rows [nparts][2][L/2 ... ] (row 0 of each pair = sums of the stored values):
// out[c] = sum_rows parts[row][0][c0 + c].  Used for the ConvTranspose2d bias gradient, whose input (the first
// half of the decoder's concat gradient) is written by the dgrad kernel that produced `parts`.
__global__ __launch_bounds__(256) void stats_colsum_finalize_kernel(const float* __restrict__ parts, int nparts, int L, int c0,
                                                                    float* out, int C) {
  // 32 lanes per channel (rows l, l+32, ... each, fixed shuffle tree): a serial walk over the rows cost 15 us per launch
  const int c = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  double s = 0.0;
  if (c < C)
    for (int i = l; i < nparts; i += 32) s += (double)parts[(long)i * L + c0 + c];
  s = lane32_sum(s);
  if (c < C && l == 0) out[c] = (float)s;
}

// row_floats = floats per partial row (2 * ctotal for statistics rows, 3 * ctotal for BatchNorm-backward rows)
int launch_stats_colsum_rows(const float* parts, int nparts, int row_floats, int c0, int c, float* out, hipStream_t stream) {
  UNETDC_REQUIRE(parts && out && nparts > 0 && c0 >= 0 && c > 0 && c0 + c <= row_floats, "stats_colsum: bad arguments");
  const float* rp; int rows;
  int rc = reduce_parts(parts, nparts, row_floats, &rp, &rows, stream, 512);
  if (rc != UNETDC_OK) return rc;
  hipLaunchKernelGGL(stats_colsum_finalize_kernel, dim3((c + 7) / 8), dim3(256), 0, stream, rp, rows, row_floats, c0, out, c);
  return check_launch("stats_colsum_finalize_kernel");
}

int launch_stats_colsum(const float* parts, int nparts, int ctotal, int c0, int c, float* out, hipStream_t stream) {
  UNETDC_REQUIRE(c0 + c <= ctotal, "stats_colsum: bad column range");
  return launch_stats_colsum_rows(parts, nparts, 2 * ctotal, c0, c, out, stream);
}

int launch_pack_many(const void* table_dev, int n, long total_tiles, int dtype, hipStream_t stream) {
  UNETDC_REQUIRE(table_dev && n > 0 && total_tiles > 0 && total_tiles < (1L << 31), "pack_many: empty table");
  UNETDC_REQUIRE(dtype == UNETDC_F32 || dtype == UNETDC_BF16, "pack_many: bad dtype %d", dtype);
  const PackDesc* t = reinterpret_cast<const PackDesc*>(table_dev);
  if (dtype == UNETDC_BF16)
    hipLaunchKernelGGL(pack_many_kernel<bf16_t>, dim3((unsigned)total_tiles), dim3(256), 0, stream, t, n);
  else
    hipLaunchKernelGGL(pack_many_kernel<float>, dim3((unsigned)total_tiles), dim3(256), 0, stream, t, n);
  return check_launch("pack_many_kernel");
}

}  // namespace unetdc
