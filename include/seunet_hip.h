/* seunet_hip.h -- C ABI of libseunet_hip.so: the MI355X (gfx950) SE-UNet hot path.
 *
 * The reference (Beryl2000/SE-UNet-AirSeg) has no native / FFI layer: its boundary for this path is
 * the Python nn.Module surface of SE_UNet.py plus three loss functions in train.py.
 * This header is the C boundary a binding for that surface talks to; every entry point names the
 * reference code it replaces.  Conventions:
 *   - plain C, no torch types: raw device pointers, explicit shapes, a dtype enum, a hipStream_t;
 *   - every function returns 0 on success, non-zero on error (message: seunet_last_error(), thread
 *     local); nothing throws across the ABI; every tensor and workspace is caller-owned and the
 *     launches themselves neither allocate nor synchronise;
 *   - what the library keeps per process, all of it created lazily on first need and none of it per call:
 *       (1) per device, one 4-KB page of zeros (hipMalloc + hipMemset, i.e. one device synchronisation, the first time
 *           a kernel that pads through it runs on that device -- or up front by seunet_init);
 *       (2) per kernel instantiation, a bit mask of the devices on which hipFuncSetAttribute (LDS above 64 KB) has run;
 *       (3) the opt-in seunet_prof_* recorder (process-wide, off by default; the per-launch-group timer of bench.py);
 *       (4) diagnostic environment switches (SEUNET_NO_STREAM, SEUNET_NO_MARCH, SEUNET_STREAM_WGS, ...: INTEGRATION.md section 2),
 *           read once per process, for A/B timing only.
 *     Apart from these everything is parameterised by (pointers, stream) and calls are re-entrant across threads and
 *     streams; a graph object (seunet_net_forward_capture) is owned by the caller like any other handle;
 *   - activations inside the library are channels-last [N][D][H][W][C], C a multiple of 8, f32, bf16 or
 *     f16 (SEUNET_F32 / SEUNET_BF16 / SEUNET_F16); parameters, logits, losses and gradients of parameters are
 *     f32 in the PyTorch layouts of the reference's state_dict.
 */
#ifndef SEUNET_HIP_H
#define SEUNET_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef SEUNET_F32
#define SEUNET_F32 0
#define SEUNET_BF16 1
#define SEUNET_F16 2   /* IEEE half activation storage (BASELINE configs[4] "fp16+MFMA"): v_mfma_*_f16, same rate and bytes as bf16 */
#endif
#define SEUNET_CONV_MFMA 0   /* implicit-GEMM matrix-core kernels (default)                  */
#define SEUNET_CONV_NAIVE 1  /* one-thread-per-output HIP kernels (device-side cross-check)  */
#define SEUNET_CONV_MARCH 2  /* seunet_conv3d_wgrad only: force the marching weight-gradient kernel (taps 27) / the whole-GEMM
                                1x1x1 kernel (taps 1); SEUNET_CONV_MFMA picks them by itself for the layers and sizes they
                                win on; error when the layer is not served */
#define SEUNET_CONV_TILED 3  /* seunet_conv3d_wgrad only: the tiled weight-gradient kernel whatever the size                */

typedef void* seunet_stream_t; /* a hipStream_t */
typedef struct seunet_dims { int n, d, h, w; } seunet_dims;

int seunet_version(void);
const char* seunet_last_error(void);
/* Optional: create the per-device state (1) above for `device` now, so that no later call synchronises the device
 * (call it before stream capture or a timed region; the training / inference entry points work without it). */
int seunet_init(int device);

/* ---- layout: reference tensors are NCDHW f32 (SE_UNet.py:181 "x: 1 2 128 128 128") ---------------- */
int seunet_pack_cl(int dtype, const float* in_ncdhw, int c, void* out_cl, int c_pad, seunet_dims dims, seunet_stream_t s);
int seunet_unpack_cl(int dtype, const void* in_cl, int c, float* out_ncdhw, seunet_dims dims, seunet_stream_t s);

/* ---- nn.Conv3d 3x3x3 (dilation 1|2, padding = dilation) and 1x1x1; SE_UNet.py:15,42,57 -------------
 * src/dst lists realise torch.cat (SE_UNet.py:186,195,204,212,216,218,222,224,228) and its backward.
 * weights: SEUNET_CONV_MFMA -> buffer produced by seunet_conv_pack_weights; SEUNET_CONV_NAIVE -> the
 * PyTorch (Cout,Cin,k,k,k) f32 tensor.  transpose_flip=1 selects the data-gradient operator.
 * stats_partial (optional): [n][seunet_conv_stats_slots][cout][2] f64 partial (sum, sum of squares) for the
 * following InstanceNorm3d (SE_UNet.py:17,43,59).  f64 because var = E[x^2]-E[x]^2 must survive |mean| >> std
 * (the CPU reference accumulates in double); the network's gradient is ill-conditioned w.r.t. such errors. */
size_t seunet_conv_wpack_bytes(int dtype, int taps, int cin, int cout);
int seunet_conv_pack_weights(int dtype, const float* w, int taps, int cin, int cout, int transpose_flip, void* wpack, seunet_stream_t s);
int seunet_conv_stats_slots(int impl, int taps, int dilation, seunet_dims dims);
int seunet_conv3d_fwd(int dtype, int impl, int taps, int dilation, int nsrc, const void* const* src, const int* src_c,
                      int cin, const void* weights, int transpose_flip, const float* bias, int ndst, void* const* dst,
                      const int* dst_c, const int* dst_accumulate, double* stats_partial, seunet_dims dims, seunet_stream_t s);
/* Streaming variant of the 3x3x3 convolution for the small-channel full-resolution layers (ec1 / ec2 / ec3 / dc6 forward and
 * data gradient; SE_UNet.py:108-110,147): one source tensor of 8 / 16 / 32 channels, one destination of <= 32 (16 when the
 * source has 32) channels, bf16, dilation 1 | 2.  A workgroup marches along z with the input planes arriving by LDS-DMA
 * (csrc/conv_stream.hip).  Weights: seunet_conv3d_stream_pack (transpose_flip = 1: data-gradient operator).  stats_partial:
 * [n][seunet_conv3d_stream_slots][dst_c][2] f64 (optional), same meaning as for seunet_conv3d_fwd. */
int seunet_conv3d_stream_supported(int dtype, int dilation, int src_c, int dst_c);
size_t seunet_conv3d_stream_wpack_bytes(int src_c);
int seunet_conv3d_stream_slots(int dilation, seunet_dims dims);
int seunet_conv3d_stream_pack(int dtype, const float* w, int cin_w, int cout_w, int transpose_flip, int src_c, int dst_c, void* wpack,
                              seunet_stream_t s);
int seunet_conv3d_stream(int dtype, int dilation, const void* src, int src_c, const void* wpack, const float* bias, void* dst, int dst_c,
                         int dst_accumulate, double* stats_partial, seunet_dims dims, seunet_stream_t s);
/* The same reference op (nn.Conv3d 3x3x3, SE_UNet.py:15,57, with the torch.cat of :222,228 fused) for the 32- and 64-input-channel
 * layers of the fine levels, forward and data gradient, on the marching kernel (csrc/conv_march.hip): bf16 | f16, one or two
 * source tensors of equal channel count (32 or 64 channels together), destinations in multiples of 16 channels (32 or more
 * together; a null destination drops its channels), dilation 1 | 2.  Weights: seunet_conv3d_march_pack (transpose_flip = 1:
 * data-gradient operator).  bias / stats_partial ([n][seunet_conv3d_march_slots][cout][2] f64) belong to the forward form,
 * dst_accumulate (+=) to the data-gradient form. */
int seunet_conv3d_march_supported(int dtype, int dilation, int nsrc, const int* src_c, int ndst, const int* dst_c);
size_t seunet_conv3d_march_wpack_bytes(int cin, int cout);
int seunet_conv3d_march_slots(int dilation, int cin, int cout, seunet_dims dims);
int seunet_conv3d_march_pack(int dtype, const float* w, int cin_w, int cout_w, int transpose_flip, int cin, int cout, void* wpack,
                             seunet_stream_t s);
int seunet_conv3d_march(int dtype, int dilation, int nsrc, const void* const* src, const int* src_c, const void* wpack, const float* bias,
                        int ndst, void* const* dst, const int* dst_c, const int* dst_accumulate, double* stats_partial, seunet_dims dims,
                        seunet_stream_t s);
/* weight gradient of the same small-channel layers on the streaming structure (csrc/wgrad_stream.hip): x (x_c = 8 | 16 | 32
 * channels, cin of them carry weights), dy (dy_c channels, cout valid); dw: (cout, cin, 3, 3, 3) f32, overwritten. */
int seunet_conv3d_wgrad_stream_supported(int dtype, int dilation, int x_c, int dy_c);
size_t seunet_conv3d_wgrad_stream_workspace_bytes(int x_c, int dy_c, int dilation, seunet_dims dims);
int seunet_conv3d_wgrad_stream(int dtype, int dilation, const void* x, int x_c, int cin, const void* dy, int dy_c, int cout, float* dw,
                               void* workspace, size_t workspace_bytes, seunet_dims dims, seunet_stream_t s);
size_t seunet_conv3d_wgrad_workspace_bytes(int taps, int cin, int cout);
int seunet_conv3d_wgrad(int dtype, int impl, int taps, int dilation, int nsrc, const void* const* src, const int* src_c,
                        int cin, const void* dy, int cout, float* dw, void* workspace, size_t workspace_bytes,
                        seunet_dims dims, seunet_stream_t s);

/* ---- InstanceNorm3d statistics (eps, biased variance; SE_UNet.py:17,43,59) -------------------------- */
int seunet_epilogue_slots(seunet_dims dims);
int seunet_channel_stats(int dtype, const void* t, int c, double* partial, seunet_dims dims, seunet_stream_t s);
/* mode 0: (mean, rstd) ; mode 1: (sum/count, sumsq/count) */
int seunet_stats_finalize(const double* partial, int slots, int c, int n, long long count, float eps, int mode,
                          float* out_a, float* out_b, seunet_stream_t s);

/* ---- gated block epilogue: IN -> LeakyReLU -> gate(s) -> e, side = conv1x1(e); SE_UNet.py:24-35,68-82 --
 * w_se2 == NULL selects the one-gate SSEConv.  side_out: f32 [n][vox][2] (optional).  level_map: f32
 * [n][vox] head pre-activation map (optional): += head_w[k]*drop[n][k]*side[k] (SE_UNet.py:232-233). */
int seunet_gate_epilogue_fwd(int dtype, const void* raw, const float* mean, const float* rstd, int c, const float* w_se,
                             const float* w_se2, const float* w_side, const float* b_side, float slope, void* e_out,
                             float* side_out, float* level_map, int level_accumulate, const float* head_w,
                             const float* drop, int drop_stride, seunet_dims dims, seunet_stream_t s);
/* backward, two recompute passes (dxhat itself is never stored; the loss gradient has a large common-mode
 * part that the InstanceNorm backward cancels, so the per-(n,c) sums are taken in f64 and the centred
 * result is rounded exactly once):
 *   pass A (m1 == NULL): stat_partial f64 [n][slots][c][2] = sums of dxhat, dxhat*xhat; pgrad_partial f32
 *                        [n*slots][4c+4] = dw_se | dw_se2 | dw_side[2][c] | db_side[2] | dhead_w[2]
 *   seunet_stats_finalize(mode 1) -> m1, m2 ; seunet_pgrad_reduce -> parameter gradients
 *   pass B (m1, m2 given): draw_out = rstd*(dxhat - m1 - xhat*m2), gradient w.r.t. the raw conv output
 *                          (may alias g_e). */
int seunet_gate_epilogue_bwd(int dtype, const void* raw, const float* mean, const float* rstd, int c, const float* w_se,
                             const float* w_se2, const float* w_side, const float* b_side, float slope, const void* g_e,
                             const float* g_side, const float* g_level, const float* head_w, const float* drop,
                             int drop_stride, const float* m1, const float* m2, void* draw_out, double* stat_partial,
                             float* pgrad_partial, seunet_dims dims, seunet_stream_t s);
int seunet_pgrad_reduce(const float* pgrad_partial, int records, int c, float* dw_se, float* dw_se2, float* dw_side,
                        float* db_side, float* dhead_w, seunet_stream_t s);

/* ---- aggregation block: conv1x1 -> IN -> LeakyReLU (+ x-branch); SE_UNet.py:45-49,187,196,205 -------- */
int seunet_cat_epilogue_fwd(int dtype, const void* raw, const float* mean, const float* rstd, const void* raw2,
                            const float* mean2, const float* rstd2, int c, float slope, void* out, seunet_dims dims,
                            seunet_stream_t s);
/* same two-pass scheme; pass A (m1 == NULL) fills the f64 partials, pass B writes dx (may alias g_out) / dx2 */
int seunet_cat_epilogue_bwd(int dtype, const void* g_out, const void* raw, const float* mean, const float* rstd,
                            const void* raw2, const float* mean2, const float* rstd2, int c, float slope,
                            const float* m1, const float* m2, const float* m1b, const float* m2b, void* dx, void* dx2,
                            double* stat_partial, double* stat_partial2, seunet_dims dims, seunet_stream_t s);

/* Two-branch block whose second branch is the 1x1x1 conv of the <= 2-channel network input (x33 / x63 / x93,
 * SE_UNet.py:112,118,124,187,196,205).  That conv's output is never materialised: every pass recomputes
 * raw2[c] = w2[c][0]*x0 + w2[c][1]*x1 from x_in, the packed 8-channel input [N][D][H][W][8] of the level (16 B per voxel
 * instead of 2*c); its InstanceNorm statistics follow from the input's second moments (seunet_xbranch_moments ->
 * seunet_xbranch_stats, exact in f64; moments_out, optional, keeps the per-sample means of x0, x1, x0^2, x0 x1, x1^2 for
 * the backward pass).  The conv's weight gradient dW2[c][i] = sum_v draw2[c] * x[i] is what is left of O(1) terms that cancel
 * to ~1e-5 of their size at 128^3, so it is not accumulated from the f32 draw2: pass A of the backward also sums
 * dxhat2[c] * x[i] (one f64 record per block in xw_partial: n * seunet_epilogue_slots(dims) * c * 2 doubles) and
 * seunet_cat_xgrad_finalize forms dw (c, in_channel, 1, 1, 1) in f64 from those sums, stat_partial2 and the moments
 * (train.py:602 loss.backward() through SE_UNet.py:112,118,124).  w2: (c, in_channel) f32. */
int seunet_xbranch_moment_slots(seunet_dims dims);
int seunet_xbranch_moments(int dtype, const void* x_in, double* partial /* [n][slots][5] */, seunet_dims dims, seunet_stream_t s);
int seunet_xbranch_stats(const double* partial, int slots, const float* w2, int c, int in_channel, int n, long long count,
                         float eps, float* mean2, float* rstd2, double* moments_out /* [n][5] or NULL */, seunet_stream_t s);
int seunet_cat_epilogue_fwd_x(int dtype, const void* raw, const float* mean, const float* rstd, const void* x_in,
                              const float* w2, int in_channel, const float* mean2, const float* rstd2, int c, float slope,
                              void* out, seunet_dims dims, seunet_stream_t s);
/* pass A (m1 == NULL): f64 partials of both branches (+ xw_partial when given); pass B: dx (may alias g_out) */
int seunet_cat_epilogue_bwd_x(int dtype, const void* g_out, const void* raw, const float* mean, const float* rstd,
                              const void* x_in, const float* w2, int in_channel, const float* mean2, const float* rstd2,
                              int c, float slope, const float* m1, const float* m2, const float* m1b, const float* m2b,
                              void* dx, double* stat_partial, double* stat_partial2, double* xw_partial, seunet_dims dims,
                              seunet_stream_t s);
int seunet_cat_xgrad_finalize(const double* xw_partial, const double* stat_partial2, int slots, const double* moments,
                              const float* w2, int c, int in_channel, int n, float eps, float* dw, seunet_stream_t s);

/* ---- nn.MaxPool3d(2,2) SE_UNet.py:131-133 ; nn.Upsample(x2 trilinear align_corners) :136-138 ---------- */
int seunet_maxpool_fwd(int dtype, const void* in, int c, void* out, seunet_dims in_dims, seunet_stream_t s);
int seunet_maxpool_bwd(int dtype, const void* in, const void* g_out, int c, void* g_in, int accumulate,
                       seunet_dims in_dims, seunet_stream_t s);
int seunet_upsample2_fwd(int dtype, const void* in, int c, void* out, seunet_dims in_dims, seunet_stream_t s);
int seunet_upsample2_bwd(int dtype, const void* g_out, int c, void* g_in, int accumulate, seunet_dims in_dims,
                         seunet_stream_t s);
/* side map [n][vox_low][c] f32 -> NCDHW f32 channels [c_off, c_off+c) of an (n, c_total, ...) tensor */
int seunet_side_upsample(const float* side, int c, int scale, float* out_ncdhw, int c_total, int c_off,
                         seunet_dims low_dims, seunet_stream_t s);

/* ---- heads: dc0_0 / dc0_1 over the DropLayer-scaled side stack; SE_UNet.py:150-153,232-233 ------------ */
int seunet_head_fwd(const float* const* level_maps, int nlevels, const float* bias, float* pred, seunet_dims dims,
                    seunet_stream_t s);
size_t seunet_head_bwd_tmp_floats(seunet_dims dims);
int seunet_head_bwd(const float* g_pred, float* const* g_levels, int nlevels, float* tmp, float* g_bias,
                    seunet_dims dims, seunet_stream_t s);

/* ---- losses: dice_loss / general_union_loss_lib / atr_loss; train.py:51-76 ------------------------------
 * sums[7] (f64, device): see csrc/loss.hip.  The caller forms the loss from the sums (and all-reduces
 * them first under data parallelism), then calls seunet_loss_grad; the upstream gradient is
 * g_scale * (g_scale_dev ? *g_scale_dev : 1) so it can stay on the device.
 * terms: which losses the caller will form (SEUNET_LOSS_DICE | SEUNET_LOSS_GUL | SEUNET_LOSS_ATR; 0 = all): the sums of the
 * others are left 0 (the general-union term's pow is the expensive part of the pass; a Dice-only step skips it). */
#define SEUNET_LOSS_DICE 1
#define SEUNET_LOSS_GUL 2
#define SEUNET_LOSS_ATR 4
int seunet_loss_partial_floats(void);
int seunet_loss_sums(const float* pred, int apply_sigmoid, const float* target, const float* weight, const float* skel,
                     long long n, float* partial, double* sums, int terms, seunet_stream_t s);
/* loss value of one head (sums1 == NULL) or of a training stage's two heads from their sums, on the device:
 * f32(c_dice*dice + c_gul*gul + c_atr*atr of sums0) + f32(the same of sums1), each formed in f64 (the scalar arithmetic of
 * train.py:51-76 and the stage sums train.py:597-599,433-435,241-243 without a dozen one-element kernels). */
int seunet_loss_value(const double* sums0, double c_dice0, double c_gul0, double c_atr0, const double* sums1, double c_dice1,
                      double c_gul1, double c_atr1, float* value, seunet_stream_t s);
int seunet_loss_grad(const float* pred, int apply_sigmoid, const float* target, const float* weight, const float* skel,
                     long long n, const double* sums, float c_dice, float c_gul, float c_atr, float g_scale,
                     const float* g_scale_dev, float* g_pred, seunet_stream_t s);

/* ---- optimizer step: torch.optim.AdamW(model.parameters(), lr=0.0001).step() -----------------
 * Replaces optimizer.step() at train.py:247,439,603 (constructed at train.py:188,386,569 with PyTorch's default
 * betas=(0.9,0.999), eps=1e-8, weight_decay=0.01, amsgrad=False).  All n tensors of a step are updated by
 * ceil(n/24) launches; pointer arrays are HOST arrays of DEVICE pointers to contiguous f32 tensors of counts[i]
 * elements.  `step` counts from 1 (the value of state['step'] AFTER the increment).  lr is passed per call, so
 * torch.optim.lr_scheduler.MultiStepLR (train.py:189-191) keeps working on the host side.  Hyper-parameters are
 * doubles because PyTorch forms 1-beta, 1-lr*wd and the bias corrections in double before rounding to f32. */
int seunet_adamw_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                      const long long* counts, int n_tensors, double lr, double beta1, double beta2, double eps,
                      double weight_decay, int step, int maximize, seunet_stream_t s);

/* ---- input pipeline: one resident case -> the tensors of a training step, in one launch ------------------
 * Replaces, for a mini-batch of crops: crop extraction (data.py:645-664, :85-252), the two HU windows (data.py:667-677,
 * :286-299, :775-784 = prediction.py:39-49), (mask > 0) (data.py:675), weight ** (U + 2) * label + (1 - label)
 * (data.py:701, :389, :561) and the flip / rotate augmentation (data.py:40-67), and writes the layout train.py:582-592
 * builds: data_out (ncrop, 2, cube^3) [2048-window, 1500-window], label_out / weight_out / skel_out (ncrop, 1, cube^3), f32.
 * img: (d, h, w) HU (= file value - 1024, data.py:692), int16 or float32; label / skeleton: uint8 or NULL; weight: float16
 * (the LIB maps, lib_weight.py:50) / float32 / float64 or NULL.  starts: HOST (z, y, x) per crop; aug: HOST code per crop
 * (NULL = none): bit k = source axis k reversed, bit 3 = source axes 1 and 2 fed by output axes 2 and 1 (random_flip then
 * random_rotate composed; all random draws stay with the caller).  weight_exponent = U + 2.  f64_math: 1 = true division
 * in float64 then one rounding (int crops, data.py:286-299; prediction.py:40), 0 = float32 division (data.py:667-677).
 * cube % 32 == 0, ncrop <= 32 per call. */
#define SEUNET_IMG_I16 0
#define SEUNET_IMG_F32 1
#define SEUNET_W_F16 0
#define SEUNET_W_F32 1
#define SEUNET_W_F64 2
int seunet_crop_batch(const void* img, int img_dtype, const unsigned char* label, const void* weight, int weight_dtype,
                      const unsigned char* skeleton, int d, int h, int w, int cube, int ncrop, const int* starts, const int* aug,
                      double weight_exponent, int f64_math, float* data_out, float* label_out, float* weight_out, float* skel_out,
                      seunet_stream_t s);
/* whole-volume network input of the inference / validation loops (prediction.py:39-49,71-75; data.py:775-784,796-798):
 * out (2, nvox) f32 = [2048-window | 1500-window] */
int seunet_hu_two_channel(const void* img, int img_dtype, long long nvox, int f64_math, float* out, seunet_stream_t s);

/* ---- sliding-window assembly: the data movement of prediction.py:78-109 and of the validation /
 * test loops train.py:682-691, test.py:151-161 (window table: data.py:731-773), on the device.
 * volume: (c, x, y, z) f32 NCDHW of ONE case, resident in HBM (prediction.py:77 `x.cuda()`); starts: HOST array of
 * nwin (xl, yl, zl) triples, nwin <= 64 per call; windows are cube^3 (cube % 4 == 0).
 *   gather:      out (nwin, c, cube, cube, cube) f32 = volume[:, xl:xl+cube, yl:yl+cube, zl:zl+cube]   (prediction.py:102)
 *   accumulate:  acc[xl:.., yl:.., zl:..] += sigmoid(logits[k]) in float64, k in list order, no atomics   (:104-106)
 *   finalize:    out = acc / pred_num, pred_num rebuilt from the per-axis window starts (the windows are a product
 *                grid) plus dup0 extra copies of window 0 = (xs[0], ys[0], zs[0]) (data.py:764-765)          (:107,109) */
int seunet_window_gather(const float* volume, int c, int x, int y, int z, int cube, int nwin, const int* starts, float* out,
                         seunet_stream_t s);
int seunet_window_accumulate(const float* logits, int apply_sigmoid, int nwin, const int* starts, int cube, double* acc, int x, int y,
                             int z, seunet_stream_t s);
int seunet_window_finalize(const double* acc, int x, int y, int z, int cube, int nx, const int* xs, int ny, const int* ys, int nz,
                           const int* zs, int dup0, double* out, seunet_stream_t s);

/* ---- post-processing: double_threshold_iteration ----------------------------------------------
 * The reference carries three copies that differ in one line: prediction.py:13-37 keeps pred*255 in float64 (:19);
 * train.py:25-49 and test.py:18-42 (validation / test) round it to float32 (train.py:31, test.py:24), so values within
 * a float32 ulp of a threshold classify differently.  pred_dtype selects the copy: SEUNET_DTI_F64 | SEUNET_DTI_F32.
 * pred: (h, w, z) float64 probabilities on the device (the overlap-averaged volume, prediction.py:109, train.py:693);
 * out: h*w*z bytes, 1 where the reference's result is 1.0.  Reproduces the reference's single in-place raster-order
 * sweep bit for bit.  workspace: seunet_dti_workspace_bytes(h, w, z) bytes, caller-owned. */
#define SEUNET_DTI_F64 0
#define SEUNET_DTI_F32 1
size_t seunet_dti_workspace_bytes(int h, int w, int z);
int seunet_dti(const double* pred, int h, int w, int z, double h_thresh, double l_thresh, int pred_dtype, unsigned char* out,
               void* workspace, size_t workspace_bytes, seunet_stream_t s);

/* ---- largest component, hole filling, metric sums -------------------------------------------------------
 * seunet_largest_component: volume (h, w, z) bytes, non-zero = foreground (the thresholded prediction, prediction.py:110-116).
 *   rule SEUNET_CC_EVALUATION (train.py:749-757): out = the 26-connected component with the most voxels (ties: the one that
 *        appears LAST in raster order = cc3d's highest label, `sorted(..)[::-1]`); an empty volume gives an empty mask.
 *   rule SEUNET_CC_MAXIMUM_3D (util.py:58-75): as above; if that component has no voxel in the slices z//2, z//3, z//3*2 of
 *        the last axis the second component is taken; then scipy.ndimage.binary_fill_holes (6-connected background that
 *        does not reach the border becomes foreground).
 * out: h*w*z bytes of 0/1.  status_dev (device int, optional): 0 ok, 1 no component, 2 no second component (the reference
 * raises IndexError in both cases under maximum_3d).  Union-find with atomics on labels = minimum linear index: the
 * result is deterministic.  workspace: seunet_cc_workspace_bytes(h, w, z), caller-owned. */
#define SEUNET_CC_EVALUATION 0
#define SEUNET_CC_MAXIMUM_3D 1
size_t seunet_cc_workspace_bytes(int h, int w, int z);
int seunet_largest_component(const unsigned char* volume, int h, int w, int z, int rule, unsigned char* out, int* status_dev,
                             void* workspace, size_t workspace_bytes, seunet_stream_t s);
/* The integer sums every ATM'22 metric of metrics.py:14-78 is formed from, in one pass over 0/1 byte volumes pred / label /
 * skeleton and the int32 branch-parsing volume (label, skeleton, parsing optional):
 * out = u64[8] {sum(pred*label), sum(pred), sum(label), sum(pred*skeleton), sum(skeleton), 0, 0, 0}, then u32[nbins] counts of
 * skeleton*parsing (np.bincount, metrics.py:16-17), u32[nbins] counts of skeleton*parsing*pred (:19-21), int max id, int
 * overflow (an id >= nbins was met).  The caller forms the rounded percentages exactly like metrics.py. */
size_t seunet_metric_out_bytes(int nbins);
int seunet_metric_sums(const unsigned char* pred, const unsigned char* label, const unsigned char* skeleton, const int* parsing,
                       long long n, int nbins, void* out, size_t out_bytes, seunet_stream_t s);

/* ---- whole network: SE_UNet.forward (SE_UNet.py:181-238) and its backward ------------------------------- */
typedef struct seunet_net_desc {
  int batch, in_channel, n_classes;
  int d, h, w;          /* multiples of 8 */
  int width_mult;       /* 1 = reference widths 8/16/32/64 (SE_UNet.py:108-148) */
  int dtype;            /* activation storage: SEUNET_F32 | SEUNET_BF16 | SEUNET_F16 */
  int conv_impl;        /* SEUNET_CONV_MFMA | SEUNET_CONV_NAIVE */
  float negative_slope; /* 0.01 (nn.LeakyReLU default, SE_UNet.py:18) */
  float eps;            /* 1e-5 (nn.InstanceNorm3d default) */
} seunet_net_desc;

int seunet_net_param_count(const seunet_net_desc* desc);
/* name: state_dict key; shape: up to 5 extents (PyTorch layout), ndim 1 or 5 */
int seunet_net_param_info(const seunet_net_desc* desc, int index, char* name, int name_cap, int* shape5, int* ndim);
size_t seunet_net_workspace_bytes(const seunet_net_desc* desc);
/* params: seunet_net_param_count device pointers in registry order.  x: NCDHW f32.  drop1/drop2: DropLayer
 * scale tensors [batch][24] / [batch][12] (NULL = eval mode identity).  pred0/pred1: [batch][n_classes][d][h][w] f32 logits
 * (n_classes 1 .. 8; SE_UNet.py:100,150-151.  Every reference caller uses 1, which keeps the fused head form; more classes run
 * the heads on a general path that materialises the 2-channel side maps, csrc/classes.hip).
 * The workspace keeps everything the backward pass needs; pass the same buffer to seunet_net_backward.
 * pred0 == NULL: inference form (prediction.py:102-103 keeps only the decoder head's output): the encoder head, the side convs of
 * the twelve encoder blocks and their level maps are not evaluated; pred1 is bit-identical; no backward pass may follow. */
int seunet_net_forward(const seunet_net_desc* desc, const float* const* params, const float* x, const float* drop1,
                       const float* drop2, float* pred0, float* pred1, void* workspace, size_t workspace_bytes,
                       seunet_stream_t s);
/* The same forward pass recorded as a HIP graph on exactly these pointers (stream capture on `s`, which must not be the
 * null stream; nothing executes during the capture).  seunet_graph_launch replays it on any stream of the device: the
 * graph reads params / x / drop1 / drop2 and writes pred0 / pred1 / workspace at replay time, so the caller refreshes
 * the CONTENTS of those buffers between replays and keeps the buffers themselves alive and in place.  For the
 * whole-volume inference loop (prediction.py:78-109: the same network call per window, hundreds of times): one graph
 * launch instead of ~150 kernel launches.  Run the eager seunet_net_forward once on the device first (per-kernel
 * one-time attribute setup is not stream work).  seunet_graph_destroy(NULL) is a no-op. */
int seunet_net_forward_capture(const seunet_net_desc* desc, const float* const* params, const float* x, const float* drop1,
                               const float* drop2, float* pred0, float* pred1, void* workspace, size_t workspace_bytes,
                               seunet_stream_t s, void** graph_out);
int seunet_graph_launch(void* graph, seunet_stream_t s);
int seunet_graph_destroy(void* graph);
/* diagnostic: read one intermediate of the last forward that ran on `workspace` back as f32 (which = 0 raw conv output of block
 * `name` [NCDHW], 1 / 2 its InstanceNorm mean / rstd [N][C], 3 the block's output tensor [NCDHW]); *channels = its channel count.
 * `name` may be an x-branch (x33 / x63 / x93; which 0..2): with in_channel <= 2 its raw values exist nowhere and are recomputed
 * from the packed input by the device function the aggregation epilogue uses (params = the forward's parameter list; may be
 * NULL otherwise).  The product path does not call this (tests/flip_census.py and the same-choice gradient gate do). */
int seunet_net_read_tensor(const seunet_net_desc* desc, const float* const* params, const void* workspace, size_t workspace_bytes,
                           const char* name, int which, float* out, int* channels, seunet_stream_t s);
/* grads: device pointers in registry order, each overwritten (NULL = skip).  The dead block dc62
 * (SE_UNet.py:148,230) receives no gradient: its entry is never written. */
int seunet_net_backward(const seunet_net_desc* desc, const float* const* params, const float* g_pred0,
                        const float* g_pred1, const float* drop1, const float* drop2, float* const* grads,
                        void* workspace, size_t workspace_bytes, seunet_stream_t s);
/* the same, recording `decoder_done_event` (a hipEvent_t, or NULL) on the stream once the parameter gradients of the decoder
 * blocks (dc1 .. dc6, dc22, dc42) are final: a data-parallel caller starts reducing that part of the gradient buffer on another
 * stream while the encoder is still being differentiated (the exchange step that replaces DataParallel's reduce, train.py:577). */
int seunet_net_backward_ev(const seunet_net_desc* desc, const float* const* params, const float* g_pred0,
                           const float* g_pred1, const float* drop1, const float* drop2, float* const* grads,
                           void* workspace, size_t workspace_bytes, seunet_stream_t s, void* decoder_done_event);

/* ---- opt-in timing of the launch groups inside seunet_net_forward/backward (HIP events on the caller's
 * stream; process-wide, meant for one benchmarking thread at a time).  seunet_prof_report writes "tag<TAB>ms<TAB>count" lines and resets; it waits on
 * the recorded events, so call it outside any timed region. */
int seunet_prof_enable(int on);
int seunet_prof_enable_filtered(const char* tag_substring);   /* time only the launch groups whose tag contains it */
int seunet_prof_report(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SEUNET_HIP_H */
