// Droplet quantification on the GPU: probabilities -> thresholded mask at the ORIGINAL image
// size -> 4-connected components -> per-droplet area / centroid sums, for
//   mask512 = (logits[i, 0].cpu().numpy() > thresh).astype(np.uint8)            /root/reference/quantify_droplets_batch.py:56
//   mask    = cv2.resize(mask512, (ow, oh), cv2.INTER_NEAREST)                                                     :57
//   lbl = label(bin_mask, connectivity=1); drop objects < min_area; lbl = label(lbl, connectivity=1)               :81-86
//   regionprops_table(lbl, ["label", "area", "equivalent_diameter", "centroid"])                                   :89-90
// What the reference's table needs from the device is, per kept droplet IN LABEL ORDER: its pixel count and the sums of
// its row / column coordinates (centroid = sums / area, equivalent_diameter = sqrt(4 area / pi): host, float64).
// skimage numbers the objects in raster order of their first pixel; a union-find whose root is the component's MINIMUM
// linear index gives exactly that order, and removing whole components does not change it.
//
//   mask_kernel      strict `>` on the fp32 probability (train_DC_focal.py:259 semantics), nearest-neighbour resize with
//                    cv2's index rule  src = min(floor(dst * src_size / dst_size), src_size - 1)
//   ccl_init/merge/compress   label equivalence by lock-free union-find (atomicMin towards the smaller root)
//   ccl_stats        integer atomics into per-root accumulators: exact and order-independent (bitwise reproducible)
//   ccl_count/scan/emit   roots with area >= min_area, compacted in increasing root order (three-level exclusive scan)
// All byte / integer work: HBM-bound and tiny next to the network (a 512 x 512 mask is 256 KB).
#include "kernels.h"

namespace unetdc {

__global__ void mask_kernel(const float* __restrict__ probs, int ph, int pw, float thresh, unsigned char* __restrict__ mask,
                            int oh, int ow, double fy, double fx) {
  const long n = (long)oh * ow;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / ow), x = (int)(i - (long)y * ow);
    int sy = (int)floor(y * fy), sx = (int)floor(x * fx);
    sy = sy < ph - 1 ? sy : ph - 1;
    sx = sx < pw - 1 ? sx : pw - 1;
    mask[i] = probs[(long)sy * pw + sx] > thresh ? 1 : 0;
  }
}

// parent pointers only ever decrease and every value ever stored in L[x] is an ancestor of x in the final forest, so a
// stale read costs extra hops, never correctness; the agent-scope relaxed loads read through to L2 anyway, where the
// atomicMin of the merges executes
__device__ __forceinline__ int ccl_find(const int* L, int x) {
  int p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != x) { x = p; p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  return x;
}

__device__ __forceinline__ void ccl_unite(int* L, int a, int b) {
  for (;;) {
    a = ccl_find(L, a);
    b = ccl_find(L, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }           // a < b: hang the larger root under the smaller
    const int old = atomicMin(&L[b], a);
    if (old == b) return;                                   // b was still a root: done
    b = old;                                                // somebody re-parented b meanwhile: continue from there
  }
}

__global__ void ccl_init_kernel(const unsigned char* __restrict__ mask, int* __restrict__ L, int* __restrict__ area,
                                unsigned long long* __restrict__ sy, unsigned long long* __restrict__ sx, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    L[i] = i;
    area[i] = 0;
    sy[i] = 0ull;
    sx[i] = 0ull;
    (void)mask;
  }
}

__global__ void ccl_merge_kernel(const unsigned char* __restrict__ mask, int* __restrict__ L, int h, int w) {
  const int n = h * w;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (!mask[i]) continue;
    const int y = i / w, x = i - y * w;
    if (x + 1 < w && mask[i + 1]) ccl_unite(L, i, i + 1);   // 4-connectivity: right and down neighbours
    if (y + 1 < h && mask[i + w]) ccl_unite(L, i, i + w);
  }
}

__global__ void ccl_stats_kernel(const unsigned char* __restrict__ mask, int* __restrict__ L, int* __restrict__ area,
                                 unsigned long long* __restrict__ sy, unsigned long long* __restrict__ sx, int h, int w) {
  const int n = h * w;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (!mask[i]) continue;
    const int r = ccl_find(L, i);
    L[i] = r;                                               // path compression (a root keeps L[r] = r)
    const int y = i / w, x = i - y * w;
    atomicAdd(&area[r], 1);
    atomicAdd(&sy[r], (unsigned long long)y);
    atomicAdd(&sx[r], (unsigned long long)x);
  }
}

// kept[i] = pixel i is the root of a component with area >= min_area.  Three-level exclusive scan over 1024-element
// blocks (block sums -> one workgroup scans them -> emit), all in fixed order.
constexpr int CCL_BLK = 1024;

__device__ __forceinline__ bool ccl_kept(const unsigned char* mask, const int* L, const int* area, int i, int min_area) {
  return mask[i] && L[i] == i && area[i] >= min_area;
}

__global__ __launch_bounds__(256) void ccl_count_kernel(const unsigned char* __restrict__ mask, const int* __restrict__ L,
                                                        const int* __restrict__ area, int n, int min_area,
                                                        int* __restrict__ blocksum) {
  __shared__ int red[4];
  const int b = blockIdx.x, base = b * CCL_BLK;
  int c = 0;
  for (int k = threadIdx.x; k < CCL_BLK; k += 256) {
    const int i = base + k;
    if (i < n && ccl_kept(mask, L, area, i, min_area)) ++c;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blocksum[b] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(1024) void ccl_scan_kernel(int* __restrict__ blocksum, int nblocks, int* __restrict__ total) {
  // one workgroup: exclusive scan of up to 1024 * items_per_thread block sums (serial per thread, then across threads)
  __shared__ int part[1024];
  const int per = (nblocks + 1023) / 1024;
  const int t = threadIdx.x, lo = t * per, hi = min(lo + per, nblocks);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += blocksum[i];
  part[t] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {                      // Hillis-Steele inclusive scan
    const int v = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;                                    // exclusive prefix of this thread's range
  for (int i = lo; i < hi; ++i) {
    const int v = blocksum[i];
    blocksum[i] = run;
    run += v;
  }
  if (t == 1023) *total = part[1023];
}

__global__ __launch_bounds__(256) void ccl_emit_kernel(const unsigned char* __restrict__ mask, const int* __restrict__ L,
                                                       const int* __restrict__ area, const unsigned long long* __restrict__ sy,
                                                       const unsigned long long* __restrict__ sx, int n, int min_area,
                                                       const int* __restrict__ blockoff, int max_out, int* __restrict__ out_area,
                                                       long long* __restrict__ out_sy, long long* __restrict__ out_sx,
                                                       int* __restrict__ out_root) {
  // one wave per block of 1024 pixels would suffice; keep it simple: thread 0..255 handle 4 consecutive pixels each and a
  // block-level exclusive scan of the 256 per-thread counts gives every kept root its rank
  __shared__ int cnt[256];
  const int b = blockIdx.x, base = b * CCL_BLK + threadIdx.x * 4;
  bool k[4];
  int c = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = base + j;
    k[j] = i < n && ccl_kept(mask, L, area, i, min_area);
    c += k[j] ? 1 : 0;
  }
  cnt[threadIdx.x] = c;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int v = threadIdx.x >= o ? cnt[threadIdx.x - o] : 0;
    __syncthreads();
    cnt[threadIdx.x] += v;
    __syncthreads();
  }
  int rank = blockoff[b] + cnt[threadIdx.x] - c;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (!k[j]) continue;
    const int i = base + j;
    if (rank < max_out) {
      out_area[rank] = area[i];
      out_sy[rank] = (long long)sy[i];
      out_sx[rank] = (long long)sx[i];
      if (out_root) out_root[rank] = i;
    }
    ++rank;
  }
}

static int ccl_grid(long n) {
  long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  return (int)nb;
}

int launch_mask_from_probs(const float* probs, int ph, int pw, float thresh, unsigned char* mask, int oh, int ow,
                           hipStream_t stream) {
  UNETDC_REQUIRE(probs && mask && ph > 0 && pw > 0 && oh > 0 && ow > 0, "mask_from_probs: bad arguments");
  // cv2.resize(..., INTER_NEAREST): x_ofs[x] = min(cvFloor(x * (src_w / dst_w)), src_w - 1) in double precision
  hipLaunchKernelGGL(mask_kernel, dim3(ccl_grid((long)oh * ow)), dim3(256), 0, stream, probs, ph, pw, thresh, mask, oh, ow,
                     (double)ph / (double)oh, (double)pw / (double)ow);
  return check_launch("mask_kernel");
}

// The reference calls cv2.resize(mask512, (ow, oh), cv2.INTER_NEAREST) (quantify_droplets_batch.py:57): the third positional
// parameter of cv2.resize is `dst`, so the flag is ignored and OpenCV's default 8-bit INTER_LINEAR runs on the {0,1} mask:
//   r = m[x0] * a0 + m[x1] * a1 (11-bit coefficients), v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2  in {0,1}
// Same tables as unetdc_resize_linear_u8_to_chw_f32 (utils/data_loader.py:linear_tables + the edge rule).
__global__ void mask_linear_kernel(const float* __restrict__ probs, int ph, int pw, float thresh, unsigned char* __restrict__ mask,
                                   int oh, int ow, const int* __restrict__ xofs, const short* __restrict__ xa,
                                   const int* __restrict__ yofs, const short* __restrict__ ya) {
  const long n = (long)oh * ow;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int dy = (int)(i / ow), dx = (int)(i - (long)dy * ow);
    const int sx0 = xofs[dx], sx1 = sx0 + 1 < pw ? sx0 + 1 : pw - 1;
    int sy0 = yofs[dy], sy1 = sy0 + 1;
    sy0 = sy0 < 0 ? 0 : (sy0 > ph - 1 ? ph - 1 : sy0);
    sy1 = sy1 < 0 ? 0 : (sy1 > ph - 1 ? ph - 1 : sy1);
    const int a0 = xa[2 * dx], a1 = xa[2 * dx + 1], b0 = ya[2 * dy], b1 = ya[2 * dy + 1];
    const int m00 = probs[(long)sy0 * pw + sx0] > thresh, m01 = probs[(long)sy0 * pw + sx1] > thresh;
    const int m10 = probs[(long)sy1 * pw + sx0] > thresh, m11 = probs[(long)sy1 * pw + sx1] > thresh;
    const int r0 = m00 * a0 + m01 * a1, r1 = m10 * a0 + m11 * a1;
    const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    mask[i] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

int launch_mask_from_probs_linear(const float* probs, int ph, int pw, float thresh, unsigned char* mask, int oh, int ow,
                                  const int* xofs, const short* xcoef, const int* yofs, const short* ycoef, hipStream_t stream) {
  UNETDC_REQUIRE(probs && mask && xofs && xcoef && yofs && ycoef && ph > 0 && pw > 0 && oh > 0 && ow > 0,
                 "mask_from_probs_linear: bad arguments");
  hipLaunchKernelGGL(mask_linear_kernel, dim3(ccl_grid((long)oh * ow)), dim3(256), 0, stream, probs, ph, pw, thresh, mask, oh, ow,
                     xofs, xcoef, yofs, ycoef);
  return check_launch("mask_linear_kernel");
}

long ccl_workspace_bytes(int h, int w) {
  const long n = (long)h * w, nb = (n + CCL_BLK - 1) / CCL_BLK;
  return n * (4 + 4 + 8 + 8) + (nb + 16) * 4 + 64;
}

int launch_ccl_stats(const unsigned char* mask, int h, int w, int min_area, void* workspace, long workspace_bytes,
                     int* out_count, int* out_area, long long* out_sumy, long long* out_sumx, int* out_root, int max_out,
                     hipStream_t stream) {
  UNETDC_REQUIRE(mask && workspace && out_count && out_area && out_sumy && out_sumx, "ccl_stats: null pointer");
  UNETDC_REQUIRE(h > 0 && w > 0 && (long)h * w < (1L << 30) && max_out >= 0, "ccl_stats: bad geometry");
  if (workspace_bytes < ccl_workspace_bytes(h, w)) {
    set_error("ccl_stats: workspace too small (%ld < %ld bytes)", workspace_bytes, ccl_workspace_bytes(h, w));
    return UNETDC_EWORKSPACE;
  }
  const int n = h * w, nb = (n + CCL_BLK - 1) / CCL_BLK;
  UNETDC_REQUIRE(nb <= 1024 * 1024, "ccl_stats: image too large");
  unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
  unsigned long long* sy = reinterpret_cast<unsigned long long*>(ws);
  unsigned long long* sx = sy + n;
  int* L = reinterpret_cast<int*>(sx + n);
  int* area = L + n;
  int* blocksum = area + n;
  const int g = ccl_grid(n);
  hipLaunchKernelGGL(ccl_init_kernel, dim3(g), dim3(256), 0, stream, mask, L, area, sy, sx, n);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3(g), dim3(256), 0, stream, mask, L, h, w);
  hipLaunchKernelGGL(ccl_stats_kernel, dim3(g), dim3(256), 0, stream, mask, L, area, sy, sx, h, w);
  hipLaunchKernelGGL(ccl_count_kernel, dim3(nb), dim3(256), 0, stream, mask, L, area, n, min_area, blocksum);
  hipLaunchKernelGGL(ccl_scan_kernel, dim3(1), dim3(1024), 0, stream, blocksum, nb, out_count);
  hipLaunchKernelGGL(ccl_emit_kernel, dim3(nb), dim3(256), 0, stream, mask, L, area, sy, sx, n, min_area, blocksum, max_out,
                     out_area, out_sumy, out_sumx, out_root);
  return check_launch("ccl kernels");
}

}  // namespace unetdc
