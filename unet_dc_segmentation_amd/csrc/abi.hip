// extern "C" entry points of libunetdc_hip.so (declared in include/unetdc_hip.h).
// Thin argument validation + geometry setup; the kernels live in the other translation units.
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <vector>

#include "../../include/unetdc_hip.h"
#include "kernels.h"

namespace unetdc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local char g_kernel[128] = "";
void note_kernel(const char* name) { snprintf(g_kernel, sizeof(g_kernel), "%s", name); }

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return UNETDC_ELAUNCH;
  }
  return UNETDC_OK;
}

int ensure_dynamic_lds(const void* fn, int bytes, const char* name) {
  struct Entry { const void* fn; int dev; int bytes; };
  static std::mutex mu;
  static std::vector<Entry> memo;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  Entry* hit = nullptr;
  for (Entry& e : memo)
    if (e.fn == fn && e.dev == dev) { hit = &e; break; }
  if (hit && hit->bytes >= bytes) return UNETDC_OK;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(%s, %d bytes of LDS) failed on device %d: %s", name, bytes, dev, hipGetErrorString(e));
    return UNETDC_ELAUNCH;
  }
  if (hit) hit->bytes = bytes;
  else memo.push_back(Entry{fn, dev, bytes});
  return UNETDC_OK;
}

static void taps3x3(int d, int* offy, int* offx) {
  for (int t = 0; t < 9; ++t) {
    offy[t] = (t / 3 - 1) * d;
    offx[t] = (t % 3 - 1) * d;
  }
}

}  // namespace unetdc

using namespace unetdc;

#define GEOM_CHECK(n, h, w)                                                                          \
  UNETDC_REQUIRE((n) > 0 && (h) > 0 && (w) > 0, "bad geometry n=%d h=%d w=%d", (int)(n), (int)(h), (int)(w))

extern "C" {

int unetdc_version(void) { return UNETDC_ABI_VERSION; }
const char* unetdc_last_error(void) { return g_err; }
const char* unetdc_last_kernel(void) { return g_kernel; }

int unetdc_pack_conv3x3(const float* w, void* w_fwd, void* w_dgrad, int cout, int cin, int dtype, unetdc_stream_t s) {
  return launch_pack_conv3x3(w, w_fwd, w_dgrad, cout, cin, dtype, (hipStream_t)s);
}
int unetdc_pack_convT2x2(const float* w, void* w_fwd, void* w_dgrad, int cin, int cout, int dtype, unetdc_stream_t s) {
  return launch_pack_convT2x2(w, w_fwd, w_dgrad, cin, cout, dtype, (hipStream_t)s);
}

int unetdc_pack_many(const unetdc_pack_desc* table_dev, int n, int64_t total_tiles, int dtype, unetdc_stream_t s) {
  static_assert(sizeof(unetdc_pack_desc) == 48, "unetdc_pack_desc layout");
  return launch_pack_many(table_dev, n, (long)total_tiles, dtype, (hipStream_t)s);
}

int unetdc_adam_step(const unetdc_adam_desc* table_dev, int n, int64_t total_blocks, const float* flat_grad, double lr,
                     double beta1, double beta2, double eps, int64_t step, double grad_scale, int dtype, unetdc_stream_t s) {
  static_assert(sizeof(unetdc_adam_desc) == 80, "unetdc_adam_desc layout");
  return launch_adam_step(table_dev, n, (long)total_blocks, flat_grad, lr, beta1, beta2, eps, (long)step, grad_scale, dtype,
                          (hipStream_t)s);
}

int unetdc_conv3x3_stats_rows(int64_t npixels, int cout) { return igemm_mblocks((long)npixels, cout); }

int unetdc_conv3x3_fwd(const void* x, int ldx, const void* w_fwd, const float* bias, const float* scale,
                       const float* shift, void* y, int ldy, float* stats_part, int* stats_rows, int n, int h, int w,
                       int cin, int cout, int dilation, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  UNETDC_REQUIRE(dilation >= 1, "conv3x3_fwd: dilation must be >= 1");
  UNETDC_REQUIRE(ldx >= cin && ldy >= cout, "conv3x3_fwd: ld smaller than channel count");
  IgemmParams p{};
  p.x = x; p.w = w_fwd; p.out = y; p.bias = bias; p.scale = scale; p.shift = shift; p.stats = stats_part;
  p.M = n * h * w; p.Ho = h; p.Wo = w; p.Hi = h; p.Wi = w; p.Cin = cin; p.Cout = cout; p.ldx = ldx; p.ldo = ldy;
  p.ntaps = 9; p.stride = 1;
  p.mode = scale ? MODE_AFFINE_RELU : (stats_part ? MODE_STATS : MODE_STORE);
  taps3x3(dilation, p.offy, p.offx);
  const int rc = launch_igemm(p, dtype, (hipStream_t)s);
  if (rc == UNETDC_OK && p.mode == MODE_STATS && stats_rows) *stats_rows = p.mblocks;   // rows that carry data (the rest are zeros)
  return rc;
}

// forward conv fed from the RAW output of the stage in front of it: that stage's BatchNorm + ReLU is applied per staged patch
int unetdc_conv3x3_bnin_supported(int n, int h, int w, int cin, int cout, int dilation, int dtype) {
  if (n <= 0 || h <= 0 || w <= 0 || dilation < 1) return 0;
  IgemmParams p{};
  p.M = n * h * w; p.Ho = h; p.Wo = w; p.Hi = h; p.Wi = w; p.Cin = cin; p.Cout = cout; p.ldx = cin; p.ldo = cout;
  p.ntaps = 9; p.stride = 1; p.mode = MODE_STATS;
  taps3x3(dilation, p.offy, p.offx);
  if (!igemm_lattice_bnin_supported(p, dtype)) return 0;
  if (igemm_lattice_bnin_writes_activation(p, dtype)) return 2;      // forward stores the activation: any weight-gradient kernel follows
  return wgrad_bnin_supported(n, h, w, cout, cin, cout, cin, dilation, dtype) ? 1 : 0;
}

int unetdc_conv3x3_fwd_bnin(const void* x_raw, int ldx, const float* in_scale, const float* in_shift, const void* w_fwd,
                            const float* bias, void* y, int ldy, float* stats_part, int* stats_rows, void* act_out, int ldact,
                            int n, int h, int w, int cin, int cout, int dilation, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  UNETDC_REQUIRE(dilation >= 1 && ldx >= cin && ldy >= cout, "conv3x3_fwd_bnin: bad dilation/ld");
  UNETDC_REQUIRE(in_scale && in_shift && stats_part, "conv3x3_fwd_bnin: null pointer");
  UNETDC_REQUIRE(act_out == nullptr || (ldact >= cin && ldact % 8 == 0 && (int64_t)n * h * w * ldact * 2 < (1LL << 32)),
                 "conv3x3_fwd_bnin: bad activation ld (or an activation tensor of 4 GiB and more)");
  IgemmParams p{};
  p.act_out = act_out; p.ld_act = ldact;
  p.x = x_raw; p.w = w_fwd; p.out = y; p.bias = bias; p.stats = stats_part; p.in_scale = in_scale; p.in_shift = in_shift;
  p.M = n * h * w; p.Ho = h; p.Wo = w; p.Hi = h; p.Wi = w; p.Cin = cin; p.Cout = cout; p.ldx = ldx; p.ldo = ldy;
  p.ntaps = 9; p.stride = 1; p.mode = MODE_STATS;
  taps3x3(dilation, p.offy, p.offx);
  if (!igemm_lattice_bnin_supported(p, dtype) || (act_out && !igemm_lattice_bnin_writes_activation(p, dtype))) {
    set_error("conv3x3_fwd_bnin: shape not supported by the input-normalising kernel%s (ask unetdc_conv3x3_bnin_supported)",
              act_out ? " that also stores the activation" : "");
    return UNETDC_EUNSUPPORTED;
  }
  const int rc = launch_igemm(p, dtype, (hipStream_t)s);
  if (rc == UNETDC_OK && stats_rows) *stats_rows = p.mblocks;
  return rc;
}

int unetdc_conv3x3_wgrad_bnin(const void* x_raw, int ldx, const float* in_scale, const float* in_shift, const void* dy,
                              int lddy, float* dw, void* workspace, int64_t workspace_bytes, int n, int h, int w, int cin,
                              int cout, int dilation, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  UNETDC_REQUIRE(dilation >= 1 && in_scale && in_shift, "conv3x3_wgrad_bnin: bad arguments");
  WgradParams p{};
  p.a = dy; p.b = x_raw; p.N = n; p.H = h; p.W = w; p.Hb = h; p.Wb = w; p.CI = cout; p.CJ = cin;
  p.lda = lddy; p.ldb = ldx; p.ntaps = 9; p.stride = 1; p.in_scale = in_scale; p.in_shift = in_shift;
  taps3x3(dilation, p.offy, p.offx);
  return launch_wgrad(p, dw, workspace, (long)workspace_bytes, dtype, (hipStream_t)s);
}

int unetdc_conv3x3_dgrad(const void* dy, int lddy, const void* w_dgrad, void* dx, int lddx, int n, int h, int w,
                         int cin, int cout, int dilation, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  UNETDC_REQUIRE(dilation >= 1, "conv3x3_dgrad: dilation must be >= 1");
  UNETDC_REQUIRE(lddy >= cout && lddx >= cin, "conv3x3_dgrad: ld smaller than channel count");
  IgemmParams p{};
  p.x = dy; p.w = w_dgrad; p.out = dx;
  p.M = n * h * w; p.Ho = h; p.Wo = w; p.Hi = h; p.Wi = w; p.Cin = cout; p.Cout = cin; p.ldx = lddy; p.ldo = lddx;
  p.ntaps = 9; p.stride = 1; p.mode = MODE_STORE;
  taps3x3(dilation, p.offy, p.offx);
  return launch_igemm(p, dtype, (hipStream_t)s);
}

// dgrad whose epilogue also sums the stored gradient per channel: colsum[c] = sum_pixels dx[:, c0 + c]
int64_t unetdc_conv3x3_dgrad_colsum_workspace(int n, int h, int w, int cin) {
  return ((int64_t)igemm_mblocks((long)n * h * w, cin) + 64) * 2 * cin * 4;
}

int unetdc_conv3x3_dgrad_colsum(const void* dy, int lddy, const void* w_dgrad, void* dx, int lddx, float* colsum, int c0,
                                int c, void* workspace, int64_t workspace_bytes, int n, int h, int w, int cin, int cout,
                                int dilation, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  UNETDC_REQUIRE(dilation >= 1, "conv3x3_dgrad_colsum: dilation must be >= 1");
  UNETDC_REQUIRE(lddy >= cout && lddx >= cin, "conv3x3_dgrad_colsum: ld smaller than channel count");
  UNETDC_REQUIRE(colsum && workspace && c0 >= 0 && c > 0 && c0 + c <= cin, "conv3x3_dgrad_colsum: bad column range");
  if (workspace_bytes < unetdc_conv3x3_dgrad_colsum_workspace(n, h, w, cin)) {
    set_error("conv3x3_dgrad_colsum: workspace too small (%ld bytes)", (long)workspace_bytes);
    return UNETDC_EWORKSPACE;
  }
  IgemmParams p{};
  p.x = dy; p.w = w_dgrad; p.out = dx;
  p.M = n * h * w; p.Ho = h; p.Wo = w; p.Hi = h; p.Wi = w; p.Cin = cout; p.Cout = cin; p.ldx = lddy; p.ldo = lddx;
  p.ntaps = 9; p.stride = 1; p.mode = MODE_STATS; p.stats = reinterpret_cast<float*>(workspace);
  taps3x3(dilation, p.offy, p.offx);
  int rc = launch_igemm(p, dtype, (hipStream_t)s);
  if (rc != UNETDC_OK) return rc;
  return launch_stats_colsum(p.stats, p.mblocks, cin, c0, c, colsum, (hipStream_t)s);       // rows the kernel wrote
}

// dgrad whose epilogue also produces the BatchNorm-backward partial sums of the stage that consumes dx
static int dgrad_bnstats_common(IgemmParams& p, const void* y_prev, int ldy_prev, const float* scale,
                                const float* shift, const float* mean, const float* rstd, float* parts,
                                int64_t parts_floats, int* nparts, int n, int h, int w, int c_prev, int dtype,
                                hipStream_t stream) {
  UNETDC_REQUIRE(y_prev && scale && shift && mean && rstd && parts && nparts, "dgrad_bnstats: null pointer");
  const int rows = igemm_mblocks((long)p.M, p.Cout);
  UNETDC_REQUIRE((int64_t)(rows + 64) * 3 * c_prev <= parts_floats, "dgrad_bnstats: partial buffer too small");
  p.mode = MODE_BNBWD;
  p.stats = parts; p.bn_y = y_prev; p.bn_ldy = ldy_prev; p.scale = scale; p.shift = shift; p.bn_mean = mean; p.bn_rstd = rstd;
  int rc = launch_igemm(p, dtype, stream);
  if (rc == UNETDC_OK) { *nparts = p.mblocks; return rc; }                   // rows that carry data (<= rows; the rest are zeros)
  if (rc != UNETDC_EUNSUPPORTED) return rc;
  // first-generation kernel selected: plain dgrad, then the standalone reduction pass
  p.mode = MODE_STORE; p.stats = nullptr;
  rc = launch_igemm(p, dtype, stream);
  if (rc != UNETDC_OK) return rc;
  BnBwdParams b{};
  b.dskip = p.out; b.lds = p.ldo; b.y = y_prev; b.ldy = ldy_prev; b.scale = scale; b.shift = shift; b.mean = mean;
  b.rstd = rstd; b.N = n; b.H = h; b.W = w; b.C = c_prev;
  return launch_bn_bwd_reduce_only(b, parts, (long)parts_floats, nparts, dtype, stream);
}

int unetdc_conv3x3_dgrad_bnstats(const void* dy, int lddy, const void* w_dgrad, void* dx, int lddx,
                                            const void* y_prev, int ldy_prev, const float* scale, const float* shift,
                                            const float* mean, const float* rstd, float* parts, int64_t parts_floats,
                                            int* nparts, int n, int h, int w, int cin, int cout, int dilation,
                                            int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  UNETDC_REQUIRE(dilation >= 1 && lddy >= cout && lddx >= cin, "conv3x3_dgrad_bnstats: bad dilation/ld");
  IgemmParams p{};
  p.x = dy; p.w = w_dgrad; p.out = dx;
  p.M = n * h * w; p.Ho = h; p.Wo = w; p.Hi = h; p.Wi = w; p.Cin = cout; p.Cout = cin; p.ldx = lddy; p.ldo = lddx;
  p.ntaps = 9; p.stride = 1;
  taps3x3(dilation, p.offy, p.offx);
  return dgrad_bnstats_common(p, y_prev, ldy_prev, scale, shift, mean, rstd, parts, parts_floats, nparts, n, h, w,
                              cin, dtype, (hipStream_t)s);
}

int unetdc_convT2x2_dgrad_bnstats(const void* dup, int lddup, const void* w_dgrad, void* dx, int lddx,
                                             const void* y_prev, int ldy_prev, const float* scale, const float* shift,
                                             const float* mean, const float* rstd, float* parts,
                                             int64_t parts_floats, int* nparts, int n, int h, int w, int cin, int cout,
                                             int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  UNETDC_REQUIRE(lddup >= cout && lddx >= cin, "convT2x2_dgrad_bnstats: ld smaller than channel count");
  IgemmParams p{};
  p.x = dup; p.w = w_dgrad; p.out = dx;
  p.M = n * h * w; p.Ho = h; p.Wo = w; p.Hi = 2 * h; p.Wi = 2 * w; p.Cin = cout; p.Cout = cin; p.ldx = lddup;
  p.ldo = lddx; p.ntaps = 4; p.stride = 2;
  for (int t = 0; t < 4; ++t) { p.offy[t] = t >> 1; p.offx[t] = t & 1; }
  return dgrad_bnstats_common(p, y_prev, ldy_prev, scale, shift, mean, rstd, parts, parts_floats, nparts, n, h, w,
                              cin, dtype, (hipStream_t)s);
}

int64_t unetdc_conv3x3_wgrad_workspace(int n, int h, int w, int cin, int cout, int dtype) {
  long b = wgrad_workspace_bytes((long)n * h * w, cout, cin, 9, dtype);
  // the tap-fused kernel may be chosen (launch_wgrad decides with wgrad_fused_supported; the query returns 0 for
  // shapes that kernel never takes, so the same condition governs both sides)
  const long f = wgrad_fused_workspace_bytes(n, h, w, cout, cin, dtype);
  if (f > b) b = f;
  for (int d = 1; d <= 64; d *= 2) {          // the valid-rectangle kernel (strongly dilated layers); dilation is not an argument here
    const long r = wgrad_rect_workspace_bytes(n, h, w, cout, cin, d);
    if (r > b) b = r;
  }
  return b;
}

int unetdc_conv3x3_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, void* workspace,
                         int64_t workspace_bytes, int n, int h, int w, int cin, int cout, int dilation, int dtype,
                         unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  UNETDC_REQUIRE(dilation >= 1, "conv3x3_wgrad: dilation must be >= 1");
  WgradParams p{};
  p.a = dy; p.b = x; p.N = n; p.H = h; p.W = w; p.Hb = h; p.Wb = w; p.CI = cout; p.CJ = cin;
  p.lda = lddy; p.ldb = ldx; p.ntaps = 9; p.stride = 1;
  taps3x3(dilation, p.offy, p.offx);
  return launch_wgrad(p, dw, workspace, (long)workspace_bytes, dtype, (hipStream_t)s);
}

int unetdc_conv3x3_first_stats_rows(int64_t npixels, int cin, int cout) {
  return first_conv_mblocks((long)npixels, cin, cout);
}

int unetdc_conv3x3_first_fwd(const float* x_nchw, const float* w, const float* bias, const float* scale,
                             const float* shift, void* y, int ldy, float* stats_part, int n, int h, int wd, int cin,
                             int cout, int dilation, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, wd);
  UNETDC_REQUIRE(dilation >= 1 && ldy >= cout, "conv3x3_first_fwd: bad dilation/ld");
  UNETDC_REQUIRE((scale == nullptr) == (shift == nullptr), "conv3x3_first_fwd: scale/shift mismatch");
  FirstParams p{};
  p.x = x_nchw; p.w = w; p.bias = bias; p.scale = scale; p.shift = shift; p.y = y; p.stats = stats_part;
  p.N = n; p.H = h; p.W = wd; p.Cin = cin; p.Cout = cout; p.ldy = ldy; p.dil = dilation;
  return launch_first_fwd(p, dtype, (hipStream_t)s);
}

int64_t unetdc_conv3x3_first_wgrad_workspace(int n, int h, int w, int cin, int cout) {
  return first_wgrad_workspace_bytes((long)n * h * w, cin, cout);
}

int unetdc_conv3x3_first_wgrad(const float* x_nchw, const void* dy, int lddy, float* dw, void* workspace,
                               int64_t workspace_bytes, int n, int h, int w, int cin, int cout, int dilation,
                               int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  FirstWgradParams p{};
  p.x = x_nchw; p.dy = dy; p.N = n; p.H = h; p.W = w; p.Cin = cin; p.Cout = cout; p.lddy = lddy; p.dil = dilation;
  return launch_first_wgrad(p, dw, workspace, (long)workspace_bytes, dtype, (hipStream_t)s);
}

int unetdc_conv3x3_first_wgrad_bn_supported(int n, int h, int w, int cin, int cout, int dilation, int dtype) {
  return first_wgrad_bn_supported(n, h, w, cin, cout, dilation, dtype) ? 1 : 0;
}

int unetdc_conv3x3_first_wgrad_bn(const float* x_nchw, const void* dz, int lddz, const void* y, int ldy, const float* scale,
                                  const float* shift, const float* mean, const float* rstd, const float* coeffs, float* dw,
                                  void* workspace, int64_t workspace_bytes, int n, int h, int w, int cin, int cout,
                                  int dilation, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  UNETDC_REQUIRE(y != nullptr, "first_wgrad_bn: null saved output");
  FirstWgradParams p{};
  p.x = x_nchw; p.dy = dz; p.N = n; p.H = h; p.W = w; p.Cin = cin; p.Cout = cout; p.lddy = lddz; p.dil = dilation;
  p.bn_y = y; p.bn_ldy = ldy; p.bn_scale = scale; p.bn_shift = shift; p.bn_mean = mean; p.bn_rstd = rstd; p.bn_k = coeffs;
  return launch_first_wgrad(p, dw, workspace, (long)workspace_bytes, dtype, (hipStream_t)s);
}

int unetdc_bn_relu_bwd_coeffs(const float* pre_parts, int pre_nparts, const float* gamma, const float* rstd, float* dgamma,
                              float* dbeta, float* dbias, float* coeffs, int n, int h, int w, int c, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  return launch_bn_bwd_coeffs(pre_parts, pre_nparts, (long)n * h * w, gamma, rstd, dgamma, dbeta, dbias, coeffs, c, (hipStream_t)s);
}

int unetdc_conv3x3_first_dgrad(const void* dy, int lddy, const float* w, float* dx_nchw, int n, int h, int wd, int cin,
                               int cout, int dilation, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, wd);
  return launch_first_dgrad(dy, lddy, w, dx_nchw, n, h, wd, cin, cout, dilation, dtype, (hipStream_t)s);
}

int unetdc_convT2x2_fwd(const void* x, int ldx, const void* w_fwd, const float* bias, void* up, int ldup, int n,
                        int h, int w, int cin, int cout, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  UNETDC_REQUIRE(ldx >= cin && ldup >= cout, "convT2x2_fwd: ld smaller than channel count");
  IgemmParams p{};
  p.x = x; p.w = w_fwd; p.out = up; p.bias = bias;
  p.M = n * h * w; p.Ho = h; p.Wo = w; p.Hi = h; p.Wi = w; p.Cin = cin; p.Cout = 4 * cout; p.ldx = ldx; p.ldo = ldup;
  p.ntaps = 1; p.stride = 1; p.mode = MODE_SHUFFLE; p.shuf_c = cout;
  p.offy[0] = 0; p.offx[0] = 0;
  return launch_igemm(p, dtype, (hipStream_t)s);
}

int unetdc_convT2x2_dgrad(const void* dup, int lddup, const void* w_dgrad, void* dx, int lddx, int n, int h, int w,
                          int cin, int cout, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  UNETDC_REQUIRE(lddup >= cout && lddx >= cin, "convT2x2_dgrad: ld smaller than channel count");
  IgemmParams p{};
  p.x = dup; p.w = w_dgrad; p.out = dx;
  p.M = n * h * w; p.Ho = h; p.Wo = w; p.Hi = 2 * h; p.Wi = 2 * w; p.Cin = cout; p.Cout = cin; p.ldx = lddup;
  p.ldo = lddx; p.ntaps = 4; p.stride = 2; p.mode = MODE_STORE;
  for (int t = 0; t < 4; ++t) { p.offy[t] = t >> 1; p.offx[t] = t & 1; }
  return launch_igemm(p, dtype, (hipStream_t)s);
}

int64_t unetdc_convT2x2_wgrad_workspace(int n, int h, int w, int cin, int cout, int dtype) {
  const long a = wgrad_workspace_bytes((long)n * h * w, cin, cout, 4, dtype);
  const long b = convt_wgrad_fused_workspace_bytes(n, h, w, cin, cout);          // the tap-fused kernel may be chosen
  return a > b ? a : b;
}

int unetdc_convT2x2_wgrad(const void* x, int ldx, const void* dup, int lddup, float* dw, void* workspace,
                          int64_t workspace_bytes, int n, int h, int w, int cin, int cout, int dtype,
                          unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  WgradParams p{};
  p.a = x; p.b = dup; p.N = n; p.H = h; p.W = w; p.Hb = 2 * h; p.Wb = 2 * w; p.CI = cin; p.CJ = cout;
  p.lda = ldx; p.ldb = lddup; p.ntaps = 4; p.stride = 2;
  for (int t = 0; t < 4; ++t) { p.offy[t] = t >> 1; p.offx[t] = t & 1; }
  return launch_wgrad(p, dw, workspace, (long)workspace_bytes, dtype, (hipStream_t)s);
}

int unetdc_bn_finalize(const float* stats_part, int rows, int64_t count, const float* gamma, const float* beta,
                       float eps, float momentum, float* running_mean, float* running_var, float* scale,
                       float* shift, float* mean, float* rstd, int c, unetdc_stream_t s) {
  return launch_bn_finalize(stats_part, rows, (long)count, gamma, beta, eps, momentum, running_mean, running_var,
                            scale, shift, mean, rstd, c, (hipStream_t)s);
}

int unetdc_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                          const float* running_var, const float* conv_bias, float eps, float* scale, float* shift,
                          int c, unetdc_stream_t s) {
  return launch_bn_eval_affine(gamma, beta, running_mean, running_var, conv_bias, eps, scale, shift, c,
                               (hipStream_t)s);
}

int unetdc_bn_relu_apply(const void* y, int ldy, const float* scale, const float* shift, void* a, int lda,
                         void* pooled, int ldp, int n, int h, int w, int c, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  ApplyParams p{};
  p.y = y; p.a = a; p.pooled = pooled; p.scale = scale; p.shift = shift;
  p.N = n; p.H = h; p.W = w; p.C = c; p.ldy = ldy; p.lda = lda; p.ldp = ldp;
  return launch_apply(p, dtype, (hipStream_t)s);
}

int64_t unetdc_bn_relu_bwd_workspace(int n, int h, int w, int c, int pooled, int dtype) {
  return bn_bwd_workspace_bytes(n, h, w, c, pooled, dtype);
}

int unetdc_bn_relu_bwd(const void* dskip, int ldskip, const void* dpool, int ldpool, const void* y, int ldy,
                       const float* scale, const float* shift, const float* mean, const float* rstd,
                       const float* gamma, void* dy, int lddy, float* dgamma, float* dbeta, float* dbias,
                       void* workspace, int64_t workspace_bytes, const float* pre_parts, int pre_nparts, int n, int h,
                       int w, int c, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  BnBwdParams p{};
  p.dskip = dskip; p.dpool = dpool; p.y = y; p.dy = dy; p.scale = scale; p.shift = shift; p.mean = mean; p.rstd = rstd;
  p.N = n; p.H = h; p.W = w; p.C = c; p.lds = ldskip; p.ldp = ldpool; p.ldy = ldy; p.lddy = lddy;
  return launch_bn_bwd(p, gamma, dgamma, dbeta, dbias, workspace, (long)workspace_bytes, pre_parts, pre_nparts, dtype,
                       (hipStream_t)s);
}

int unetdc_bn_relu_bwd_head(const float* dprobs, const float* probs, const float* head_w, const void* y, int ldy,
                            const float* scale, const float* shift, const float* mean, const float* rstd,
                            const float* gamma, void* dy, int lddy, float* dgamma, float* dbeta, float* dbias,
                            void* workspace, int64_t workspace_bytes, const float* pre_parts, int pre_nparts, int n, int h,
                            int w, int c, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  BnBwdParams p{};
  p.head_dprobs = dprobs; p.head_probs = probs; p.head_w = head_w;
  p.y = y; p.dy = dy; p.scale = scale; p.shift = shift; p.mean = mean; p.rstd = rstd;
  p.N = n; p.H = h; p.W = w; p.C = c; p.ldy = ldy; p.lddy = lddy;
  UNETDC_REQUIRE(head_w != nullptr, "bn_relu_bwd_head: null head weights");
  return launch_bn_bwd(p, gamma, dgamma, dbeta, dbias, workspace, (long)workspace_bytes, pre_parts, pre_nparts, dtype,
                       (hipStream_t)s);
}

int unetdc_bn_frozen_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                            float eps, float* scale, float* shift, float* mean, float* rstd, int c, unetdc_stream_t s) {
  return launch_bn_frozen_affine(gamma, beta, running_mean, running_var, eps, scale, shift, mean, rstd, c, (hipStream_t)s);
}

int unetdc_bn_relu_bwd_frozen(const void* dskip, int ldskip, const void* dpool, int ldpool, const void* y, int ldy,
                              const float* scale, const float* shift, const float* mean, const float* rstd,
                              const float* gamma, void* dy, int lddy, float* dgamma, float* dbeta, float* dbias,
                              void* workspace, int64_t workspace_bytes, const float* pre_parts, int pre_nparts, int n, int h,
                              int w, int c, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, w);
  BnBwdParams p{};
  p.dskip = dskip; p.dpool = dpool; p.y = y; p.dy = dy; p.scale = scale; p.shift = shift; p.mean = mean; p.rstd = rstd;
  p.N = n; p.H = h; p.W = w; p.C = c; p.lds = ldskip; p.ldp = ldpool; p.ldy = ldy; p.lddy = lddy;
  return launch_bn_bwd(p, gamma, dgamma, dbeta, dbias, workspace, (long)workspace_bytes, pre_parts, pre_nparts, dtype,
                       (hipStream_t)s, true);
}

int unetdc_head_fwd(const void* a, int lda, const float* w, const float* b, float* probs, int n, int h, int wd,
                    int c, int oc, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, wd);
  HeadParams p{};
  p.a = a; p.w = w; p.b = b; p.probs = probs; p.N = n; p.H = h; p.W = wd; p.C = c; p.OC = oc; p.lda = lda;
  return launch_head_fwd(p, dtype, (hipStream_t)s);
}

int unetdc_head_fwd_bn(const void* y, int ldy, const float* scale, const float* shift, const float* w, const float* b,
                       float* probs, int n, int h, int wd, int c, int oc, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, wd);
  UNETDC_REQUIRE(y && scale && shift, "head_fwd_bn: null pointer");
  HeadParams p{};
  p.a = y; p.w = w; p.b = b; p.probs = probs; p.N = n; p.H = h; p.W = wd; p.C = c; p.OC = oc; p.lda = ldy;
  p.bn_scale = scale; p.bn_shift = shift;
  return launch_head_fwd(p, dtype, (hipStream_t)s);
}

int64_t unetdc_head_bwd_workspace(int n, int h, int w, int c, int oc, int dtype) {
  return head_bwd_workspace_bytes(n, h, w, c, oc, dtype);
}

int unetdc_head_bwd(const float* dprobs, const float* probs, const void* a, int lda, const float* w, void* da,
                    int ldda, float* dw, float* db, void* workspace, int64_t workspace_bytes, int n, int h, int wd,
                    int c, int oc, int dtype, unetdc_stream_t s) {
  GEOM_CHECK(n, h, wd);
  HeadParams p{};
  p.a = a; p.w = w; p.probs = const_cast<float*>(probs); p.dprobs = dprobs; p.da = da;
  p.N = n; p.H = h; p.W = wd; p.C = c; p.OC = oc; p.lda = lda; p.ldda = ldda;
  return launch_head_bwd(p, dw, db, workspace, (long)workspace_bytes, dtype, (hipStream_t)s);
}

int unetdc_head_bwd_bnstats(const float* dprobs, const float* probs, const void* a, int lda, const float* w, void* da,
                            int ldda, float* dw, float* db, void* workspace, int64_t workspace_bytes, const void* y_prev,
                            int ldy_prev, const float* scale, const float* shift, const float* mean, const float* rstd,
                            float* parts, int64_t parts_floats, int* nparts, int n, int h, int wd, int c, int oc, int dtype,
                            unetdc_stream_t s) {
  GEOM_CHECK(n, h, wd);
  UNETDC_REQUIRE(y_prev && scale && shift && mean && rstd && parts && nparts, "head_bwd_bnstats: null pointer");
  HeadParams p{};
  p.a = a; p.w = w; p.probs = const_cast<float*>(probs); p.dprobs = dprobs; p.da = da;
  p.N = n; p.H = h; p.W = wd; p.C = c; p.OC = oc; p.lda = lda; p.ldda = ldda;
  p.bn_y = y_prev; p.bn_ldy = ldy_prev; p.bn_scale = scale; p.bn_shift = shift; p.bn_mean = mean; p.bn_rstd = rstd;
  p.bn_parts = parts;
  return launch_head_bwd(p, dw, db, workspace, (long)workspace_bytes, dtype, (hipStream_t)s, nparts, (long)parts_floats);
}

int64_t unetdc_focal_dice_loss_workspace(int nimg, int64_t hw) { return loss_workspace_bytes(nimg, (long)hw); }

int unetdc_focal_dice_loss_fwd(const float* probs, const float* target, float* loss_out, float* coef, void* workspace,
                               int64_t workspace_bytes, int nimg, int64_t hw, float alpha, float gamma, float ratio,
                               float smooth, unetdc_stream_t s) {
  return launch_loss_fwd(probs, target, loss_out, coef, workspace, (long)workspace_bytes, nimg, (long)hw, alpha, gamma,
                         ratio, smooth, (hipStream_t)s);
}

int unetdc_focal_dice_loss_bwd(const float* probs, const float* target, const float* coef, const float* grad_out,
                               float* dprobs, int nimg, int64_t hw, float alpha, float gamma, float ratio,
                               unetdc_stream_t s) {
  return launch_loss_bwd(probs, target, coef, grad_out, dprobs, nimg, (long)hw, alpha, gamma, ratio, (hipStream_t)s);
}

int64_t unetdc_channel_sum_workspace(int64_t npixels, int c) { return channel_sum_workspace_bytes((long)npixels, c); }

int unetdc_channel_sum(const void* x, int ldx, float* out, void* workspace, int64_t workspace_bytes,
                       int64_t npixels, int c, int dtype, unetdc_stream_t s) {
  return launch_channel_sum(x, ldx, out, workspace, (long)workspace_bytes, (long)npixels, c, dtype, (hipStream_t)s);
}

int unetdc_mask_from_probs(const float* probs, int ph, int pw, float thresh, uint8_t* mask, int oh, int ow,
                           unetdc_stream_t s) {
  return launch_mask_from_probs(probs, ph, pw, thresh, mask, oh, ow, (hipStream_t)s);
}

int unetdc_mask_from_probs_linear(const float* probs, int ph, int pw, float thresh, uint8_t* mask, int oh, int ow,
                                  const int32_t* xofs, const int16_t* xcoef, const int32_t* yofs, const int16_t* ycoef,
                                  unetdc_stream_t s) {
  return launch_mask_from_probs_linear(probs, ph, pw, thresh, mask, oh, ow, xofs, xcoef, yofs, ycoef, (hipStream_t)s);
}

int64_t unetdc_ccl_workspace(int h, int w) { return ccl_workspace_bytes(h, w); }

int unetdc_ccl_stats(const uint8_t* mask, int h, int w, int min_area, void* workspace, int64_t workspace_bytes,
                     int32_t* out_count, int32_t* out_area, int64_t* out_sumy, int64_t* out_sumx, int32_t* out_root,
                     int max_out, unetdc_stream_t s) {
  return launch_ccl_stats(mask, h, w, min_area, workspace, (long)workspace_bytes, out_count, out_area,
                          reinterpret_cast<long long*>(out_sumy), reinterpret_cast<long long*>(out_sumx), out_root, max_out,
                          (hipStream_t)s);
}

int64_t unetdc_rolling_ball_workspace(int h, int w, int channels) { return rolling_ball_workspace_bytes(h, w, channels); }

int unetdc_rolling_ball_u8(const uint8_t* src_hwc, uint8_t* dst_hwc, int h, int w, int channels, int ksize, void* workspace,
                           int64_t workspace_bytes, unetdc_stream_t s) {
  return launch_rolling_ball(src_hwc, dst_hwc, h, w, channels, ksize, workspace, (long)workspace_bytes, (hipStream_t)s);
}

int unetdc_resize_linear_u8_to_chw_f32(const uint8_t* src_hwc, int h, int w, int channels, float* dst_chw, int dh, int dw,
                                       const int32_t* xofs, const int16_t* xcoef, const int32_t* yofs, const int16_t* ycoef,
                                       unetdc_stream_t s) {
  return launch_resize_linear_chw(src_hwc, h, w, channels, dst_chw, dh, dw, xofs, xcoef, yofs, ycoef, (hipStream_t)s);
}

}  // extern "C"
