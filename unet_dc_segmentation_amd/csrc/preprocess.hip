// Input preprocessing on the GPU: the rolling-ball background correction and the resize that
// stand in front of the network in both entry points,
//   kernel = cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (radius, radius))      /root/reference/utils/data_loader.py:17
//   background = cv2.morphologyEx(channel, cv2.MORPH_OPEN, kernel)                                                :19
//   corrected = cv2.subtract(channel, background); cv2.normalize(corrected, None, 0, 255, cv2.NORM_MINMAX)        :20-21
//   im = cv2.resize(im, (512, 512), ...); im.astype(np.float32) / 255.0; HWC -> CHW       quantify_droplets_batch.py:43-46
// restated on uint8 exactly as OpenCV defines them (cv2 itself is not available to this build: see DESIGN.md, "parity
// unpinned against cv2, pinned against the numpy restatement in utils/data_loader.py"):
//   * structuring element row i of a k x k ellipse: r = k/2, dy = i - r, dx = round(r * sqrt(1 - dy^2/r^2)), columns
//     [max(r - dx, 0), min(r + dx + 1, k)), anchor (k/2, k/2); erosion / dilation read src(y + i - r, x + j - r), pixels
//     outside the image do not take part (OpenCV's default morphology border value);
//   * opening = dilate(erode(src)) with the SAME element (no reflection -- OpenCV's definition, also for even k);
//   * subtract saturates at 0; normalize: scale = 255 / (max - min) (0 when max == min), shift = -min * scale, both formed
//     in double and applied in float, result rounded half-to-even and saturated;
//   * resize: OpenCV's 8-bit INTER_LINEAR (11-bit fixed-point coefficients, the >>4 / >>16 / +2 >>2 vertical pass).
// Byte work on small images (a 1388 x 1040 RGB micrograph is 4.3 MB): a 64 x 64 output tile stages its (64+k)^2 patch in
// LDS and takes every row span of the element as two power-of-two windows (morph_kernel); HBM traffic is one read and one
// write per pass.
#include "kernels.h"

namespace unetdc {

constexpr int PP_TILE = 64;
constexpr int PP_MAXK = 128;
constexpr int PP_THREADS = 512;
constexpr int PP_OPT = PP_TILE * PP_TILE / PP_THREADS;      // outputs per thread

struct MorphSpans {
  short j1[PP_MAXK];
  short j2[PP_MAXK];
};

// Erosion / dilation of one channel plane of an interleaved HWC uint8 image (pixel (y, x) at (y*W + x)*cn + c) by the k x k
// ellipse given as row spans [j1[i], j2[i]).
//
// A 64 x 64 output tile stages its (64 + k)^2 patch in LDS once.  The extremum over a row span of length L is the extremum of
// TWO power-of-two windows, min(M_s[j1], M_s[j2 - 2^s]) with 2^s <= L < 2^(s+1), where M_s[x] = min(row[x .. x + 2^s - 1]);
// the plane is lifted from level s to level s + 1 IN PLACE (M_{s+1}[x] = min(M_s[x], M_s[x + 2^s]), a read phase, a barrier,
// a write phase) after every row whose span has floor(log2 L) = s has been consumed, so one plane serves all levels.  Per
// output pixel: 2 k LDS reads plus the share of log2(k) lifts, instead of one read per element pixel (pi k^2 / 4: 1960
// for the default k = 50) -- the first form of this kernel took 7.2 ms per pass on a 1388 x 1040 x 3 micrograph, 93 % of the
// time of the droplet-quantification flow.  min / max of bytes: exact, order-independent.
template <bool IS_MAX>
__global__ __launch_bounds__(PP_THREADS) void morph_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                           int H, int W, int cn, int k, const MorphSpans sp) {
  extern __shared__ unsigned char plane[];                  // (PP_TILE + k) x (PP_TILE + k), pitch P
  const int r = k / 2, P = PP_TILE + k;
  const int c = blockIdx.z;
  const int y0 = blockIdx.y * PP_TILE, x0 = blockIdx.x * PP_TILE;
  const int tid = threadIdx.x;
  const unsigned char ident = IS_MAX ? 0 : 255;             // pixels outside the image never win
  auto better = [](int a, int b) { return IS_MAX ? (a > b ? a : b) : (a < b ? a : b); };
  for (int i = tid; i < P * P; i += PP_THREADS) {
    const int py = i / P, px = i - py * P;
    const int gy = y0 + py - r, gx = x0 + px - r;
    plane[i] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? src[((long)gy * W + gx) * cn + c] : ident;
  }
  int acc[PP_OPT];
#pragma unroll
  for (int u = 0; u < PP_OPT; ++u) acc[u] = ident;
  int smax = 0;                                             // highest level any row needs (uniform)
  for (int i = 0; i < k; ++i) {
    const int L = sp.j2[i] - sp.j1[i];
    if (L > 0) { const int s = 31 - __builtin_clz((unsigned)L); smax = s > smax ? s : smax; }
  }
  __syncthreads();
  for (int s = 0; s <= smax; ++s) {
    const int win = 1 << s;
    for (int i = 0; i < k; ++i) {                           // rows whose span length L has floor(log2 L) == s
      const int j1 = sp.j1[i], j2 = sp.j2[i], L = j2 - j1;
      if (L <= 0 || (31 - __builtin_clz((unsigned)L)) != s) continue;
#pragma unroll
      for (int u = 0; u < PP_OPT; ++u) {
        const int o = tid + u * PP_THREADS, ty = o / PP_TILE, tx = o - ty * PP_TILE;
        const unsigned char* row = plane + (ty + i) * P + tx;          // patch row of src row y + i - r; column j <-> x + j - r
        acc[u] = better(acc[u], better(row[j1], row[j2 - win]));
      }
    }
    if (s == smax) break;
    // lift the plane to level s + 1 in place, in chunks of PP_THREADS * 8 bytes: a chunk reads itself and bytes to its right
    // (unwritten so far), so one barrier between its reads and its writes is enough; the last barrier closes the level
    for (int base = 0; base < P * P; base += PP_THREADS * 8) {
      int v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = base + q * PP_THREADS + tid;
        v[q] = 0;
        if (e < P * P) {
          const int col = e % P;
          v[q] = plane[e];
          if (col + win < P) v[q] = better(v[q], plane[e + win]);      // windows that would leave the row are never read
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = base + q * PP_THREADS + tid;
        if (e < P * P) plane[e] = (unsigned char)v[q];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < PP_OPT; ++u) {
    const int o = tid + u * PP_THREADS, ty = o / PP_TILE, tx = o - ty * PP_TILE;
    const int y = y0 + ty, x = x0 + tx;
    if (y < H && x < W) dst[((long)y * W + x) * cn + c] = (unsigned char)acc[u];
  }
}

// corrected = saturate(src - background); per-channel min / max of the result (integer atomics: exact).
// A thread takes CN consecutive 16-byte chunks (16 pixels): the channel of byte b of such a group is b % CN at compile time.
template <int CN>
__global__ __launch_bounds__(256) void subtract_minmax_kernel(const unsigned char* __restrict__ src,
                                                              const unsigned char* __restrict__ bg,
                                                              unsigned char* __restrict__ out, long npix,
                                                              int* __restrict__ mn, int* __restrict__ mx) {
  int lmn[CN], lmx[CN];
#pragma unroll
  for (int c = 0; c < CN; ++c) { lmn[c] = 255; lmx[c] = 0; }
  const long ngroups = npix / 16;                           // groups of 16 pixels = 16 * CN bytes
  for (long gI = (long)blockIdx.x * blockDim.x + threadIdx.x; gI < ngroups; gI += (long)gridDim.x * blockDim.x) {
    const long o = gI * 16 * CN;
#pragma unroll
    for (int q = 0; q < CN; ++q) {
      const u32x4 a = ld16(src + o + 16 * q), b = ld16(bg + o + 16 * q);
      u32x4 r;
#pragma unroll
      for (int wI = 0; wI < 4; ++wI) {
        const unsigned aw = a[wI], bw = b[wI];
        unsigned rw = 0;
#pragma unroll
        for (int bI = 0; bI < 4; ++bI) {
          const int ch = (16 * q + 4 * wI + bI) % CN;
          const int d = (int)((aw >> (8 * bI)) & 255u) - (int)((bw >> (8 * bI)) & 255u);
          const int v = d > 0 ? d : 0;
          rw |= (unsigned)v << (8 * bI);
          lmn[ch] = v < lmn[ch] ? v : lmn[ch];
          lmx[ch] = v > lmx[ch] ? v : lmx[ch];
        }
        r[wI] = rw;
      }
      st16(out + o + 16 * q, r);
    }
  }
  // tail: the last npix % 16 pixels, one per thread of block 0
  if (blockIdx.x == 0) {
    const long p = ngroups * 16 + threadIdx.x;
    if (p < npix) {
#pragma unroll
      for (int c = 0; c < CN; ++c) {
        const int d = (int)src[p * CN + c] - (int)bg[p * CN + c];
        const int v = d > 0 ? d : 0;
        out[p * CN + c] = (unsigned char)v;
        lmn[c] = v < lmn[c] ? v : lmn[c];
        lmx[c] = v > lmx[c] ? v : lmx[c];
      }
    }
  }
  // wave -> workgroup -> ONE atomic per workgroup and channel: same-address atomics retire at < 100 per microsecond, the
  // first form's one per wave (8.5 K of them on a 1388 x 1040 image) took 0.09 ms of this kernel's 0.1 ms
  __shared__ int red[2][CN][4];
#pragma unroll
  for (int c = 0; c < CN; ++c) {
    int a = lmn[c], b = lmx[c];
    for (int o = 32; o > 0; o >>= 1) {
      const int a2 = __shfl_xor(a, o, 64), b2 = __shfl_xor(b, o, 64);
      a = a2 < a ? a2 : a;
      b = b2 > b ? b2 : b;
    }
    if ((threadIdx.x & 63) == 0) { red[0][c][threadIdx.x >> 6] = a; red[1][c][threadIdx.x >> 6] = b; }
  }
  __syncthreads();
  if (threadIdx.x < CN) {
    const int c = threadIdx.x;
    int a = red[0][c][0], b = red[1][c][0];
    for (int wv = 1; wv < 4; ++wv) {
      a = red[0][c][wv] < a ? red[0][c][wv] : a;
      b = red[1][c][wv] > b ? red[1][c][wv] : b;
    }
    atomicMin(&mn[c], a);
    atomicMax(&mx[c], b);
  }
}

__global__ void minmax_init_kernel(int* mn, int* mx) {
  if (threadIdx.x < 4) { mn[threadIdx.x] = 255; mx[threadIdx.x] = 0; }
}

// cv2.normalize(NORM_MINMAX, 0..255) on uint8: dst = saturate(round_half_even(float(src) * (float)scale + (float)shift))
__global__ void normalize_kernel(unsigned char* __restrict__ img, long npix, int cn, const int* __restrict__ mn,
                                 const int* __restrict__ mx) {
  float sc[4], sh[4];
  for (int c = 0; c < cn; ++c) {
    const double smin = mn[c], smax = mx[c];
    const double scale = 255.0 * ((smax - smin) > 2.220446049250313e-16 ? 1.0 / (smax - smin) : 0.0);
    sc[c] = (float)scale;
    sh[c] = (float)(0.0 - smin * scale);
  }
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    for (int c = 0; c < cn; ++c) {
      const float v = __fadd_rn(__fmul_rn((float)img[p * cn + c], sc[c]), sh[c]);      // no fused multiply-add
      int q = (int)rintf(v);                                                             // round half to even
      q = q < 0 ? 0 : (q > 255 ? 255 : q);
      img[p * cn + c] = (unsigned char)q;
    }
  }
}

// OpenCV 8-bit INTER_LINEAR to (dh, dw), then /255 and HWC -> CHW float32.  xofs/yofs: source index of the first tap;
// xa/ya: the two 11-bit coefficients per destination index (host tables, see utils/data_loader.py:_linear_tables).
__global__ void resize_linear_chw_kernel(const unsigned char* __restrict__ src, int H, int W, int cn, float* __restrict__ dst,
                                         int dh, int dw, const int* __restrict__ xofs, const short* __restrict__ xa,
                                         const int* __restrict__ yofs, const short* __restrict__ ya) {
  const long n = (long)dh * dw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int dy = (int)(i / dw), dx = (int)(i - (long)dy * dw);
    const int sx0 = xofs[dx], sx1 = sx0 + 1 < W ? sx0 + 1 : W - 1;
    int sy0 = yofs[dy], sy1 = sy0 + 1;
    sy0 = sy0 < 0 ? 0 : (sy0 > H - 1 ? H - 1 : sy0);
    sy1 = sy1 < 0 ? 0 : (sy1 > H - 1 ? H - 1 : sy1);
    const int a0 = xa[2 * dx], a1 = xa[2 * dx + 1], b0 = ya[2 * dy], b1 = ya[2 * dy + 1];
    for (int c = 0; c < cn; ++c) {
      const int r0 = src[((long)sy0 * W + sx0) * cn + c] * a0 + src[((long)sy0 * W + sx1) * cn + c] * a1;   // horizontal pass
      const int r1 = src[((long)sy1 * W + sx0) * cn + c] * a0 + src[((long)sy1 * W + sx1) * cn + c] * a1;
      const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;                             // vertical pass
      const unsigned char u = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
      dst[((long)c * dh + dy) * dw + dx] = (float)u / 255.0f;
    }
  }
}

static int pp_grid(long n) {
  long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  return (int)(nb < 1 ? 1 : nb);
}

// two image planes (each padded to 16 bytes: the 16-byte accesses of subtract_minmax_kernel) + 64 bytes for the min / max words
static long pp_plane_bytes(int h, int w, int cn) { return ((long)h * w * cn + 15) & ~15L; }
long rolling_ball_workspace_bytes(int h, int w, int cn) { return 2L * pp_plane_bytes(h, w, cn) + 64; }

int launch_rolling_ball(const unsigned char* src, unsigned char* dst, int h, int w, int cn, int k, void* workspace,
                        long workspace_bytes, hipStream_t stream) {
  UNETDC_REQUIRE(src && dst && workspace && h > 0 && w > 0 && cn >= 1 && cn <= 4, "rolling_ball: bad arguments");
  UNETDC_REQUIRE(k >= 1 && k <= PP_MAXK, "rolling_ball: element size must be in [1, %d]", PP_MAXK);
  if (workspace_bytes < rolling_ball_workspace_bytes(h, w, cn)) {
    set_error("rolling_ball: workspace too small (%ld bytes)", workspace_bytes);
    return UNETDC_EWORKSPACE;
  }
  MorphSpans sp;
  {
    // cv::getStructuringElement(MORPH_ELLIPSE, Size(k, k))
    const int r = k / 2, c = k / 2;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < k; ++i) {
      const int dy = i - r;
      int j1 = 0, j2 = 0;
      if (abs(dy) <= r) {
        const int dx = (int)lrint(c * sqrt((double)(r * r - dy * dy) * inv_r2));
        j1 = c - dx > 0 ? c - dx : 0;
        j2 = c + dx + 1 < k ? c + dx + 1 : k;
      }
      sp.j1[i] = (short)j1;
      sp.j2[i] = (short)j2;
    }
  }
  unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
  unsigned char* eroded = ws;
  unsigned char* bg = ws + pp_plane_bytes(h, w, cn);
  const dim3 grid((w + PP_TILE - 1) / PP_TILE, (h + PP_TILE - 1) / PP_TILE, cn);
  const int lds = (PP_TILE + k) * (PP_TILE + k);              // <= 192 x 192 = 36 KB
  hipLaunchKernelGGL(morph_kernel<false>, grid, dim3(PP_THREADS), lds, stream, src, eroded, h, w, cn, k, sp);
  hipLaunchKernelGGL(morph_kernel<true>, grid, dim3(PP_THREADS), lds, stream, eroded, bg, h, w, cn, k, sp);
  // per-channel min / max (int[4] each) live in the 64 spare bytes behind the two planes (16-byte aligned): never inside a
  // plane -- for images smaller than 32 bytes a scratch at the head of `eroded` would reach into `bg`, which
  // subtract_minmax_kernel reads while the atomics write
  int* mn = reinterpret_cast<int*>(ws + 2L * pp_plane_bytes(h, w, cn));
  int* mx = mn + 4;
  hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(64), 0, stream, mn, mx);
  const long npix = (long)h * w;
  {
    const dim3 g(pp_grid(npix / 16 + 1) > 256 ? 256 : pp_grid(npix / 16 + 1));
    UNETDC_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0) && ((uintptr_t)bg % 16 == 0),
                   "rolling_ball: image pointers must be 16-byte aligned");
    if (cn == 1) hipLaunchKernelGGL(subtract_minmax_kernel<1>, g, dim3(256), 0, stream, src, bg, dst, npix, mn, mx);
    else if (cn == 2) hipLaunchKernelGGL(subtract_minmax_kernel<2>, g, dim3(256), 0, stream, src, bg, dst, npix, mn, mx);
    else if (cn == 3) hipLaunchKernelGGL(subtract_minmax_kernel<3>, g, dim3(256), 0, stream, src, bg, dst, npix, mn, mx);
    else hipLaunchKernelGGL(subtract_minmax_kernel<4>, g, dim3(256), 0, stream, src, bg, dst, npix, mn, mx);
  }
  hipLaunchKernelGGL(normalize_kernel, dim3(pp_grid(npix)), dim3(256), 0, stream, dst, npix, cn, mn, mx);
  return check_launch("rolling ball kernels");
}

int launch_resize_linear_chw(const unsigned char* src, int h, int w, int cn, float* dst, int dh, int dw, const int* xofs,
                             const short* xa, const int* yofs, const short* ya, hipStream_t stream) {
  UNETDC_REQUIRE(src && dst && xofs && xa && yofs && ya && h > 0 && w > 0 && dh > 0 && dw > 0 && cn >= 1 && cn <= 4,
                 "resize_linear_chw: bad arguments");
  hipLaunchKernelGGL(resize_linear_chw_kernel, dim3(pp_grid((long)dh * dw)), dim3(256), 0, stream, src, h, w, cn, dst, dh, dw,
                     xofs, xa, yofs, ya);
  return check_launch("resize_linear_chw_kernel");
}

}  // namespace unetdc
