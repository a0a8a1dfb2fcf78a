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
// Byte work on small images (a 1388 x 1040 RGB micrograph is 4.3 MB): a 32 x 32 output tile stages its (32+k)^2 patch in
// LDS and walks the element's row spans; HBM traffic is one read and one write per pass.
#include "kernels.h"

namespace unetdc {

constexpr int PP_TILE = 32;
constexpr int PP_MAXK = 128;

struct MorphSpans {
  short j1[PP_MAXK];
  short j2[PP_MAXK];
};

// one channel plane of an interleaved HWC uint8 image: pixel (y, x) at (y*W + x)*cn + c
template <bool IS_MAX>
__global__ __launch_bounds__(256) void morph_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                    int H, int W, int cn, int k, const MorphSpans sp) {
  extern __shared__ unsigned char patch[];                  // (PP_TILE + k) x (PP_TILE + k), pitch PP_TILE + k
  const int r = k / 2, P = PP_TILE + k;
  const int c = blockIdx.z;
  const int y0 = blockIdx.y * PP_TILE, x0 = blockIdx.x * PP_TILE;
  const unsigned char ident = IS_MAX ? 0 : 255;             // pixels outside the image never win
  for (int i = threadIdx.x; i < P * P; i += 256) {
    const int py = i / P, px = i - py * P;
    const int gy = y0 + py - r, gx = x0 + px - r;
    patch[i] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? src[((long)gy * W + gx) * cn + c] : ident;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < PP_TILE * PP_TILE; o += 256) {
    const int ty = o / PP_TILE, tx = o - ty * PP_TILE;
    const int y = y0 + ty, x = x0 + tx;
    if (y >= H || x >= W) continue;
    int v = ident;
    for (int i = 0; i < k; ++i) {
      const unsigned char* row = patch + (ty + i) * P + tx;   // patch row of src row y + i - r; column j <-> x + j - r
      for (int j = sp.j1[i]; j < sp.j2[i]; ++j) {
        const int s = row[j];
        v = IS_MAX ? (s > v ? s : v) : (s < v ? s : v);
      }
    }
    dst[((long)y * W + x) * cn + c] = (unsigned char)v;
  }
}

// corrected = saturate(src - background); per-channel min / max of the result (integer atomics: exact)
__global__ void subtract_minmax_kernel(const unsigned char* __restrict__ src, const unsigned char* __restrict__ bg,
                                       unsigned char* __restrict__ out, long npix, int cn, int* __restrict__ mn,
                                       int* __restrict__ mx) {
  int lmn[4] = {255, 255, 255, 255}, lmx[4] = {0, 0, 0, 0};
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    for (int c = 0; c < cn; ++c) {
      const int d = (int)src[p * cn + c] - (int)bg[p * cn + c];
      const int v = d > 0 ? d : 0;
      out[p * cn + c] = (unsigned char)v;
      lmn[c] = v < lmn[c] ? v : lmn[c];
      lmx[c] = v > lmx[c] ? v : lmx[c];
    }
  }
  for (int c = 0; c < cn; ++c) {
    int a = lmn[c], b = lmx[c];
    for (int o = 32; o > 0; o >>= 1) {
      const int a2 = __shfl_xor(a, o, 64), b2 = __shfl_xor(b, o, 64);
      a = a2 < a ? a2 : a;
      b = b2 > b ? b2 : b;
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mn[c], a); atomicMax(&mx[c], b); }
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

long rolling_ball_workspace_bytes(int h, int w, int cn) { return 2L * h * w * cn + 64; }

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
  unsigned char* bg = ws + (long)h * w * cn;
  const dim3 grid((w + PP_TILE - 1) / PP_TILE, (h + PP_TILE - 1) / PP_TILE, cn);
  const int lds = (PP_TILE + k) * (PP_TILE + k);
  hipLaunchKernelGGL(morph_kernel<false>, grid, dim3(256), lds, stream, src, eroded, h, w, cn, k, sp);
  hipLaunchKernelGGL(morph_kernel<true>, grid, dim3(256), lds, stream, eroded, bg, h, w, cn, k, sp);
  // per-channel min / max (int[4] each) live in the 64 spare bytes behind the two planes (16-byte aligned): never inside a
  // plane -- for images smaller than 32 bytes a scratch at the head of `eroded` would reach into `bg`, which
  // subtract_minmax_kernel reads while the atomics write
  int* mn = reinterpret_cast<int*>(ws + ((2L * h * w * cn + 15) & ~15L));
  int* mx = mn + 4;
  hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(64), 0, stream, mn, mx);
  const long npix = (long)h * w;
  hipLaunchKernelGGL(subtract_minmax_kernel, dim3(pp_grid(npix)), dim3(256), 0, stream, src, bg, dst, npix, cn, mn, mx);
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
