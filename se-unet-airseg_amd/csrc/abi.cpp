// extern "C" entry points of libseunet_hip.so (per-op part; the whole-network calls live in net.cpp).
// Thin argument marshalling only: every function validates, forwards to a launcher and returns a status.
#include "seunet_common.h"
#include "../../include/seunet_hip.h"

using namespace seunet;

static inline Dims D(seunet_dims d) { return Dims{d.n, d.d, d.h, d.w}; }
static inline hipStream_t S(seunet_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static int make_src(int nsrc, const void* const* src, const int* src_c, SrcList& l) {
  SEUNET_CHECK(nsrc >= 1 && nsrc <= 3 && src && src_c, "bad source list");
  l = SrcList{};
  l.n = nsrc;
  for (int i = 0; i < nsrc; ++i) { l.ptr[i] = src[i]; l.C[i] = src_c[i]; }
  return 0;
}

namespace seunet { extern unsigned long long* g_conv_debug; }

extern "C" {

// diagnostic hook (not part of the public header): device buffer of 8 u64 that -DSEUNET_STAMP builds add cycle sums to
int seunet_debug_set_buffer(void* p) { seunet::g_conv_debug = reinterpret_cast<unsigned long long*>(p); return 0; }

int seunet_version(void) { return 200; }
const char* seunet_last_error(void) { return get_error(); }

int seunet_init(int device) {
  int count = 0, prev = 0;
  SEUNET_HIP(hipGetDeviceCount(&count));
  SEUNET_CHECK(device >= 0 && device < count, "init: device %d out of range (%d visible)", device, count);
  SEUNET_HIP(hipGetDevice(&prev));
  SEUNET_HIP(hipSetDevice(device));
  const void* page = seunet::device_zero_page();
  (void)hipSetDevice(prev);
  SEUNET_CHECK(page != nullptr, "init: could not allocate the zero page on device %d", device);
  return 0;
}

int seunet_pack_cl(int dtype, const float* in, int c, void* out, int c_pad, seunet_dims dims, seunet_stream_t s) {
  SEUNET_CHECK(in && out, "pack_cl: null tensor");
  return launch_pack_cl(dtype, in, c, out, c_pad, D(dims), S(s));
}
int seunet_unpack_cl(int dtype, const void* in, int c, float* out, seunet_dims dims, seunet_stream_t s) {
  SEUNET_CHECK(in && out, "unpack_cl: null tensor");
  return launch_unpack_cl(dtype, in, c, out, D(dims), S(s));
}

size_t seunet_conv_wpack_bytes(int dtype, int taps, int cin, int cout) { return conv_wpack_bytes(dtype, taps, cin, cout); }
int seunet_conv_pack_weights(int dtype, const float* w, int taps, int cin, int cout, int tflip, void* wpack, seunet_stream_t s) {
  SEUNET_CHECK(w && wpack, "conv_pack_weights: null tensor");
  return launch_conv_pack_weights(dtype, w, taps, cin, cout, tflip, wpack, S(s));
}
int seunet_conv_stats_slots(int impl, int taps, int dilation, seunet_dims dims) {
  return impl == SEUNET_CONV_NAIVE ? epi_partials(D(dims)) : conv_stats_tiles(D(dims), taps, dilation);
}
int seunet_conv3d_fwd(int dtype, int impl, int taps, int dilation, int nsrc, const void* const* src, const int* src_c, int cin,
                      const void* weights, int tflip, const float* bias, int ndst, void* const* dst, const int* dst_c,
                      const int* dst_acc, double* stats_partial, seunet_dims dims, seunet_stream_t s) {
  SrcList sl;
  if (int e = make_src(nsrc, src, src_c, sl)) return e;
  SEUNET_CHECK(ndst >= 1 && ndst <= 3 && dst && dst_c && weights, "conv3d_fwd: bad destination list / weights");
  DstList dl{};
  dl.n = ndst;
  for (int i = 0; i < ndst; ++i) { dl.ptr[i] = dst[i]; dl.C[i] = dst_c[i]; dl.acc[i] = dst_acc ? dst_acc[i] : 0; }
  if (impl == SEUNET_CONV_NAIVE) {
    if (int e = launch_conv_naive(dtype, taps, dilation, sl, cin, (const float*)weights, tflip, bias, dl, D(dims), S(s))) return e;
    if (stats_partial) {
      SEUNET_CHECK(ndst == 1, "conv3d_fwd: statistics need a single destination");
      return launch_channel_stats(dtype, dst[0], dst_c[0], stats_partial, D(dims), S(s));
    }
    return 0;
  }
  return launch_conv_igemm(dtype, taps, dilation, sl, cin, weights, bias, dl, stats_partial, D(dims), S(s));
}
int seunet_conv3d_stream_supported(int dtype, int dilation, int src_c, int dst_c) { return conv_stream_supported(dtype, 27, dilation, src_c, dst_c) ? 1 : 0; }
size_t seunet_conv3d_stream_wpack_bytes(int src_c) { return conv_stream_wpack_bytes(src_c); }
int seunet_conv3d_stream_slots(int dilation, seunet_dims dims) { return conv_stream_slots(D(dims), dilation); }
int seunet_conv3d_stream_pack(int dtype, const float* w, int cin_w, int cout_w, int transpose_flip, int src_c, int dst_c, void* wpack,
                              seunet_stream_t s) {
  return launch_conv_stream_pack(dtype, w, cin_w, cout_w, transpose_flip, src_c, dst_c, wpack, S(s));
}
int seunet_conv3d_stream(int dtype, int dilation, const void* src, int src_c, const void* wpack, const float* bias, void* dst, int dst_c,
                         int dst_accumulate, double* stats_partial, seunet_dims dims, seunet_stream_t s) {
  return launch_conv_stream(dtype, dilation, src, src_c, wpack, bias, dst, dst_c, dst_accumulate, stats_partial, D(dims), S(s));
}
static int make_dst(int ndst, void* const* dst, const int* dst_c, const int* dst_acc, DstList& dl) {
  SEUNET_CHECK(ndst >= 1 && ndst <= 3 && dst_c, "bad destination list");
  dl = DstList{};
  dl.n = ndst;
  for (int i = 0; i < ndst; ++i) { dl.ptr[i] = dst ? dst[i] : nullptr; dl.C[i] = dst_c[i]; dl.acc[i] = dst_acc ? dst_acc[i] : 0; }
  return 0;
}
int seunet_conv3d_march_supported(int dtype, int dilation, int nsrc, const int* src_c, int ndst, const int* dst_c) {
  if (nsrc < 1 || nsrc > 3 || ndst < 1 || ndst > 3 || !src_c || !dst_c) return 0;
  SrcList sl{}; DstList dl{};
  sl.n = nsrc; dl.n = ndst;
  for (int i = 0; i < nsrc; ++i) sl.C[i] = src_c[i];
  for (int i = 0; i < ndst; ++i) dl.C[i] = dst_c[i];
  return conv_march_supported(dtype, 27, dilation, sl, dl) ? 1 : 0;
}
size_t seunet_conv3d_march_wpack_bytes(int cin, int cout) { return conv_march_wpack_bytes(cin, cout); }
int seunet_conv3d_march_slots(int dilation, int cin, int cout, seunet_dims dims) { return conv_march_slots(D(dims), dilation, cin, cout); }
int seunet_conv3d_march_pack(int dtype, const float* w, int cin_w, int cout_w, int transpose_flip, int cin, int cout, void* wpack,
                             seunet_stream_t s) {
  return launch_conv_march_pack(dtype, w, cin_w, cout_w, transpose_flip, cin, cout, wpack, S(s));
}
int seunet_conv3d_march(int dtype, int dilation, int nsrc, const void* const* src, const int* src_c, const void* wpack, const float* bias,
                        int ndst, void* const* dst, const int* dst_c, const int* dst_accumulate, double* stats_partial, seunet_dims dims,
                        seunet_stream_t s) {
  SrcList sl; DstList dl;
  if (int e = make_src(nsrc, src, src_c, sl)) return e;
  SEUNET_CHECK(dst != nullptr, "conv3d_march: null destination list");
  if (int e = make_dst(ndst, dst, dst_c, dst_accumulate, dl)) return e;
  return launch_conv_march(dtype, dilation, sl, wpack, bias, dl, stats_partial, D(dims), S(s));
}
int seunet_conv3d_wgrad_stream_supported(int dtype, int dilation, int x_c, int dy_c) { return wgrad_stream_supported(dtype, 27, dilation, x_c, dy_c) ? 1 : 0; }
size_t seunet_conv3d_wgrad_stream_workspace_bytes(int x_c, int dy_c, int dilation, seunet_dims dims) {
  return wgrad_stream_workspace_bytes(x_c, dy_c, dilation, D(dims));
}
int seunet_conv3d_wgrad_stream(int dtype, int dilation, const void* x, int x_c, int cin, const void* dy, int dy_c, int cout, float* dw,
                               void* workspace, size_t workspace_bytes, seunet_dims dims, seunet_stream_t s) {
  return launch_wgrad_stream(dtype, dilation, x, x_c, cin, dy, dy_c, cout, dw, workspace, workspace_bytes, D(dims), S(s));
}
size_t seunet_conv3d_wgrad_workspace_bytes(int taps, int cin, int cout) { return wgrad_workspace_bytes(taps, cin, cout); }
int seunet_conv3d_wgrad(int dtype, int impl, int taps, int dilation, int nsrc, const void* const* src, const int* src_c, int cin,
                        const void* dy, int cout, float* dw, void* workspace, size_t workspace_bytes, seunet_dims dims,
                        seunet_stream_t s) {
  SrcList sl;
  if (int e = make_src(nsrc, src, src_c, sl)) return e;
  SEUNET_CHECK(dy && dw, "conv3d_wgrad: null tensor");
  if (impl == SEUNET_CONV_NAIVE) return launch_wgrad_naive(dtype, taps, dilation, sl, cin, dy, cout, dw, D(dims), S(s));
  SEUNET_CHECK(workspace, "conv3d_wgrad: null workspace");
  if (impl == SEUNET_CONV_MARCH)
    return taps == 1 ? launch_wgrad_1x1(dtype, sl, cin, dy, cout, dw, workspace, workspace_bytes, D(dims), S(s))
                     : launch_wgrad_march(dtype, taps, dilation, sl, cin, dy, cout, dw, workspace, workspace_bytes, D(dims), S(s));
  return launch_wgrad(dtype, taps, dilation, sl, cin, dy, cout, dw, workspace, workspace_bytes, D(dims), S(s), impl != SEUNET_CONV_TILED);
}

int seunet_epilogue_slots(seunet_dims dims) { return epi_partials(D(dims)); }
int seunet_channel_stats(int dtype, const void* t, int c, double* partial, seunet_dims dims, seunet_stream_t s) {
  SEUNET_CHECK(t && partial, "channel_stats: null tensor");
  return launch_channel_stats(dtype, t, c, partial, D(dims), S(s));
}
int seunet_stats_finalize(const double* partial, int slots, int c, int n, long long count, float eps, int mode, float* out_a,
                          float* out_b, seunet_stream_t s) {
  SEUNET_CHECK(partial && out_a && out_b && slots >= 1 && count >= 1, "stats_finalize: bad argument");
  return launch_stats_finalize(partial, slots, c, n, count, eps, mode, out_a, out_b, S(s));
}

int seunet_gate_epilogue_fwd(int dtype, const void* raw, const float* mean, const float* rstd, int c, const float* w_se,
                             const float* w_se2, const float* w_side, const float* b_side, float slope, void* e_out,
                             float* side_out, float* level_map, int level_accumulate, const float* head_w, const float* drop,
                             int drop_stride, seunet_dims dims, seunet_stream_t s) {
  SEUNET_CHECK(raw && mean && rstd && w_se && w_side && b_side && e_out, "gate_epilogue_fwd: null tensor");
  SEUNET_CHECK(!level_map || head_w, "gate_epilogue_fwd: level_map needs head_w");
  SseParams p{w_se, w_se2, w_side, b_side, slope};
  SseHead h{side_out, level_map, level_accumulate, head_w, drop, drop_stride};
  return launch_sse_fwd(dtype, raw, mean, rstd, c, p, e_out, h, D(dims), S(s));
}
int seunet_gate_epilogue_bwd(int dtype, const void* raw, const float* mean, const float* rstd, int c, const float* w_se,
                             const float* w_se2, const float* w_side, const float* b_side, float slope, const void* g_e,
                             const float* g_side, const float* g_level, const float* head_w, const float* drop, int drop_stride,
                             const float* m1, const float* m2, void* draw_out, double* stat_partial, float* pgrad_partial,
                             seunet_dims dims, seunet_stream_t s) {
  SEUNET_CHECK(raw && mean && rstd && w_se && w_side && b_side, "gate_epilogue_bwd: null tensor");
  SEUNET_CHECK(!g_level || head_w, "gate_epilogue_bwd: g_level needs head_w");
  SseParams p{w_se, w_se2, w_side, b_side, slope};
  SseBwdIn g{g_e, g_side, g_level};
  SseHead h{nullptr, nullptr, 0, head_w, drop, drop_stride};
  return launch_sse_bwd(dtype, raw, mean, rstd, c, p, g, h, m1, m2, draw_out, stat_partial, pgrad_partial, D(dims), S(s));
}
int seunet_pgrad_reduce(const float* pgrad_partial, int records, int c, float* dw_se, float* dw_se2, float* dw_side,
                        float* db_side, float* dhead_w, seunet_stream_t s) {
  SEUNET_CHECK(pgrad_partial && records >= 1, "pgrad_reduce: bad argument");
  return launch_pgrad_reduce(pgrad_partial, records, c, dw_se, dw_se2, dw_side, db_side, dhead_w, S(s));
}
int seunet_cat_epilogue_fwd(int dtype, const void* raw, const float* mean, const float* rstd, const void* raw2,
                            const float* mean2, const float* rstd2, int c, float slope, void* out, seunet_dims dims,
                            seunet_stream_t s) {
  SEUNET_CHECK(raw && mean && rstd && out && (!raw2 || (mean2 && rstd2)), "cat_epilogue_fwd: null tensor");
  return launch_cat_fwd(dtype, raw, mean, rstd, raw2, mean2, rstd2, c, slope, out, D(dims), S(s));
}
int seunet_cat_epilogue_bwd(int dtype, const void* g_out, const void* raw, const float* mean, const float* rstd,
                            const void* raw2, const float* mean2, const float* rstd2, int c, float slope, const float* m1,
                            const float* m2, const float* m1b, const float* m2b, void* dx, void* dx2, double* stat_partial,
                            double* stat_partial2, seunet_dims dims, seunet_stream_t s) {
  SEUNET_CHECK(g_out && raw && mean && rstd, "cat_epilogue_bwd: null tensor");
  SEUNET_CHECK(!raw2 || (mean2 && rstd2), "cat_epilogue_bwd: second branch incomplete");
  return launch_cat_bwd(dtype, g_out, raw, mean, rstd, raw2, mean2, rstd2, c, slope, m1, m2, m1b, m2b, dx, dx2, stat_partial,
                        stat_partial2, D(dims), S(s));
}

int seunet_maxpool_fwd(int dtype, const void* in, int c, void* out, seunet_dims d, seunet_stream_t s) {
  SEUNET_CHECK(in && out, "maxpool_fwd: null tensor");
  return launch_maxpool_fwd(dtype, in, c, out, D(d), S(s));
}
int seunet_maxpool_bwd(int dtype, const void* in, const void* g_out, int c, void* g_in, int accumulate, seunet_dims d,
                       seunet_stream_t s) {
  SEUNET_CHECK(in && g_out && g_in, "maxpool_bwd: null tensor");
  return launch_maxpool_bwd(dtype, in, g_out, c, g_in, accumulate, D(d), S(s));
}
int seunet_upsample2_fwd(int dtype, const void* in, int c, void* out, seunet_dims d, seunet_stream_t s) {
  SEUNET_CHECK(in && out, "upsample2_fwd: null tensor");
  return launch_upsample2_fwd(dtype, in, c, out, D(d), S(s));
}
int seunet_upsample2_bwd(int dtype, const void* g_out, int c, void* g_in, int accumulate, seunet_dims d, seunet_stream_t s) {
  SEUNET_CHECK(g_out && g_in, "upsample2_bwd: null tensor");
  return launch_upsample2_bwd(dtype, g_out, c, g_in, accumulate, D(d), S(s));
}
int seunet_side_upsample(const float* side, int c, int scale, float* out, int c_total, int c_off, seunet_dims low,
                         seunet_stream_t s) {
  SEUNET_CHECK(side && out && scale >= 1, "side_upsample: bad argument");
  return launch_side_upsample(side, c, scale, out, c_total, c_off, D(low), S(s));
}

int seunet_head_fwd(const float* const* level_maps, int nlevels, const float* bias, float* pred, seunet_dims d, seunet_stream_t s) {
  SEUNET_CHECK(level_maps && bias && pred, "head_fwd: null tensor");
  return launch_head_fwd(level_maps, nlevels, bias, pred, D(d), S(s));
}
size_t seunet_head_bwd_tmp_floats(seunet_dims d) { return head_bwd_tmp_floats(D(d)); }
int seunet_head_bwd(const float* g_pred, float* const* g_levels, int nlevels, float* tmp, float* g_bias, seunet_dims d,
                    seunet_stream_t s) {
  SEUNET_CHECK(g_pred && g_levels && tmp, "head_bwd: null tensor");
  return launch_head_bwd(g_pred, g_levels, nlevels, tmp, g_bias, D(d), S(s));
}

int seunet_loss_partial_floats(void) { return loss_partials() * SEUNET_LOSS_NSUMS; }
int seunet_loss_sums(const float* pred, int apply_sigmoid, const float* target, const float* weight, const float* skel,
                     long long n, float* partial, double* sums, int terms, seunet_stream_t s) {
  SEUNET_CHECK(pred && target && partial && sums && n >= 1, "loss_sums: bad argument");
  SEUNET_CHECK(terms >= 0 && terms <= 7, "loss_sums: terms=%d is not a mask of SEUNET_LOSS_DICE | _GUL | _ATR", terms);
  return launch_loss_sums(pred, apply_sigmoid, target, weight, skel, n, partial, sums, S(s), terms);
}
int seunet_loss_value(const double* sums0, double c_dice0, double c_gul0, double c_atr0, const double* sums1, double c_dice1,
                      double c_gul1, double c_atr1, float* value, seunet_stream_t s) {
  SEUNET_CHECK(sums0 && value, "loss_value: null argument");
  return launch_loss_value(sums0, c_dice0, c_gul0, c_atr0, sums1, c_dice1, c_gul1, c_atr1, value, S(s));
}
int seunet_loss_grad(const float* pred, int apply_sigmoid, const float* target, const float* weight, const float* skel,
                     long long n, const double* sums, float c_dice, float c_gul, float c_atr, float g_scale,
                     const float* g_scale_dev, float* g_pred, seunet_stream_t s) {
  SEUNET_CHECK(pred && target && sums && g_pred && n >= 1, "loss_grad: bad argument");
  return launch_loss_grad(pred, apply_sigmoid, target, weight, skel, n, sums, c_dice, c_gul, c_atr, g_scale, g_scale_dev, g_pred, S(s));
}

int seunet_xbranch_moment_slots(seunet_dims dims) { return xbranch_moment_slots(D(dims)); }
int seunet_xbranch_moments(int dtype, const void* x_in, double* partial, seunet_dims dims, seunet_stream_t s) {
  SEUNET_CHECK(x_in && partial, "xbranch_moments: null argument");
  return launch_xbranch_moments(dtype, x_in, partial, D(dims), S(s));
}
int seunet_xbranch_stats(const double* partial, int slots, const float* w2, int c, int in_channel, int n, long long count,
                         float eps, float* mean2, float* rstd2, double* moments_out, seunet_stream_t s) {
  SEUNET_CHECK(partial && w2 && mean2 && rstd2 && slots >= 1 && c >= 1 && n >= 1 && count >= 1, "xbranch_stats: bad argument");
  return launch_xbranch_stats(partial, slots, w2, c, in_channel, n, count, eps, mean2, rstd2, moments_out, S(s));
}
int seunet_cat_epilogue_fwd_x(int dtype, const void* raw, const float* mean, const float* rstd, const void* x_in,
                              const float* w2, int in_channel, const float* mean2, const float* rstd2, int c, float slope,
                              void* out, seunet_dims dims, seunet_stream_t s) {
  SEUNET_CHECK(raw && mean && rstd && x_in && w2 && mean2 && rstd2 && out, "cat_epilogue_fwd_x: null argument");
  return launch_cat_fwd_x(dtype, raw, mean, rstd, x_in, w2, in_channel, mean2, rstd2, c, slope, out, D(dims), S(s));
}
int seunet_cat_epilogue_bwd_x(int dtype, const void* g_out, const void* raw, const float* mean, const float* rstd,
                              const void* x_in, const float* w2, int in_channel, const float* mean2, const float* rstd2,
                              int c, float slope, const float* m1, const float* m2, const float* m1b, const float* m2b,
                              void* dx, double* stat_partial, double* stat_partial2, double* xw_partial, seunet_dims dims,
                              seunet_stream_t s) {
  SEUNET_CHECK(g_out && raw && mean && rstd && x_in && w2 && mean2 && rstd2, "cat_epilogue_bwd_x: null argument");
  return launch_cat_bwd_x(dtype, g_out, raw, mean, rstd, x_in, w2, in_channel, mean2, rstd2, c, slope, m1, m2, m1b, m2b, dx,
                          stat_partial, stat_partial2, xw_partial, D(dims), S(s));
}
int seunet_cat_xgrad_finalize(const double* xw_partial, const double* stat_partial2, int slots, const double* moments, const float* w2,
                              int c, int in_channel, int n, float eps, float* dw, seunet_stream_t s) {
  SEUNET_CHECK(xw_partial && stat_partial2 && moments && w2 && dw && slots >= 1 && c >= 8 && n >= 1, "cat_xgrad_finalize: bad argument");
  return launch_cat_xgrad_finalize(xw_partial, stat_partial2, slots, moments, w2, c, in_channel, n, eps, dw, S(s));
}

size_t seunet_cc_workspace_bytes(int h, int w, int z) {
  if (h < 1 || w < 1 || z < 1) { fail("cc_workspace_bytes: bad dimensions"); return 0; }
  return cc_workspace_bytes(h, w, z);
}
int seunet_largest_component(const unsigned char* volume, int h, int w, int z, int rule, unsigned char* out, int* status_dev,
                             void* workspace, size_t workspace_bytes, seunet_stream_t s) {
  return launch_largest_component(volume, h, w, z, rule, out, status_dev, workspace, workspace_bytes, S(s));
}
size_t seunet_metric_out_bytes(int nbins) { return nbins < 1 ? 0 : metric_out_bytes(nbins); }
int seunet_metric_sums(const unsigned char* pred, const unsigned char* label, const unsigned char* skeleton, const int* parsing,
                       long long n, int nbins, void* out, size_t out_bytes, seunet_stream_t s) {
  return launch_metric_sums(pred, label, skeleton, parsing, n, nbins, out, out_bytes, S(s));
}

int seunet_crop_batch(const void* img, int img_dtype, const unsigned char* label, const void* weight, int weight_dtype,
                      const unsigned char* skeleton, int d, int h, int w, int cube, int ncrop, const int* starts, const int* aug,
                      double weight_exponent, int f64_math, float* data_out, float* label_out, float* weight_out, float* skel_out,
                      seunet_stream_t s) {
  return launch_crop_batch(img, img_dtype, label, weight, weight_dtype, skeleton, d, h, w, cube, ncrop, starts, aug, weight_exponent,
                           f64_math, data_out, label_out, weight_out, skel_out, S(s));
}
int seunet_hu_two_channel(const void* img, int img_dtype, long long nvox, int f64_math, float* out, seunet_stream_t s) {
  return launch_hu_two_channel(img, img_dtype, nvox, f64_math, out, S(s));
}

int seunet_window_gather(const float* volume, int c, int x, int y, int z, int cube, int nwin, const int* starts, float* out,
                         seunet_stream_t s) {
  return launch_window_gather(volume, c, x, y, z, cube, nwin, starts, out, S(s));
}
int seunet_window_accumulate(const float* logits, int apply_sigmoid, int nwin, const int* starts, int cube, double* acc, int x, int y,
                             int z, seunet_stream_t s) {
  return launch_window_accumulate(logits, apply_sigmoid, nwin, starts, cube, acc, x, y, z, S(s));
}
int seunet_window_finalize(const double* acc, int x, int y, int z, int cube, int nx, const int* xs, int ny, const int* ys, int nz,
                           const int* zs, int dup0, double* out, seunet_stream_t s) {
  return launch_window_finalize(acc, x, y, z, cube, nx, xs, ny, ys, nz, zs, dup0, out, S(s));
}

size_t seunet_dti_workspace_bytes(int h, int w, int z) {
  if (h < 1 || w < 1 || z < 1) { fail("dti_workspace_bytes: bad dimensions"); return 0; }
  return dti_workspace_bytes(h, w, z);
}
int seunet_dti(const double* pred, int h, int w, int z, double h_thresh, double l_thresh, int pred_dtype, unsigned char* out,
               void* workspace, size_t workspace_bytes, seunet_stream_t s) {
  return launch_dti(pred, h, w, z, h_thresh, l_thresh, pred_dtype, out, workspace, workspace_bytes, S(s));
}

int seunet_adamw_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                      const long long* counts, int n_tensors, double lr, double beta1, double beta2, double eps,
                      double weight_decay, int step, int maximize, seunet_stream_t s) {
  SEUNET_CHECK(n_tensors == 0 || (params && grads && exp_avg && exp_avg_sq && counts), "adamw_step: bad argument");
  return launch_adamw(params, grads, exp_avg, exp_avg_sq, counts, n_tensors, lr, beta1, beta2, eps, weight_decay, step,
                      maximize, S(s));
}

}  // extern "C"
