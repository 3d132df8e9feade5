// Dice / General-Union / ATR losses (reference train.py:51-76), forward sums and backward, gfx950.
//
// All three losses are ratios of whole-batch sums, so one streaming pass produces the
// seven sums every loss needs and one streaming pass produces d(loss)/d(pred).  With
// apply_sigmoid=1 the kernels take raw logits (train.py:595-596 applies torch.sigmoid first) and the
// backward pass folds in sigmoid'.
//   sums[0] = sum p*t        sums[1] = sum p          sums[2] = sum t            (dice)
//   sums[3] = sum w*(p+1e-4)^0.7*t   sums[4] = sum w*(0.2 p + 0.8 t)              (general union)
//   sums[5] = sum w*(p*s)*s          sums[6] = sum w*(p*s + s)                    (ATR; s = skeleton)
#include "seunet_common.h"

namespace seunet {

static constexpr int LOSS_BLOCKS = 512;
int loss_partials() { return LOSS_BLOCKS; }

__global__ void __launch_bounds__(256)
loss_sums_kernel(const float* __restrict__ pred, int apply_sigmoid, const float* __restrict__ target,
                 const float* __restrict__ weight, const float* __restrict__ skel, long long n,
                 float* __restrict__ partial) {
  float s[SEUNET_LOSS_NSUMS];
#pragma unroll
  for (int k = 0; k < SEUNET_LOSS_NSUMS; ++k) s[k] = 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float p = pred[i];
    if (apply_sigmoid) p = 1.f / (1.f + expf(-p));
    const float t = target[i];
    const float w = weight ? weight[i] : 1.f;
    const float sk = skel ? skel[i] : 0.f;
    s[0] += p * t;
    s[1] += p;
    s[2] += t;
    s[3] += w * powf(p + 1e-4f, 0.7f) * t;
    s[4] += w * (0.2f * p + 0.8f * t);
    const float ps = p * sk;
    s[5] += w * ps * sk;
    s[6] += w * (ps + sk);
  }
  __shared__ float red[4][SEUNET_LOSS_NSUMS];
#pragma unroll
  for (int k = 0; k < SEUNET_LOSS_NSUMS; ++k) {
    float v = s[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < SEUNET_LOSS_NSUMS) {
    const int k = threadIdx.x;
    partial[blockIdx.x * SEUNET_LOSS_NSUMS + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
  }
}

__global__ void loss_sums_final_kernel(const float* __restrict__ partial, int blocks, double* __restrict__ sums) {
  const int k = threadIdx.x;
  if (k >= SEUNET_LOSS_NSUMS) return;
  double s = 0.0;
  for (int b = 0; b < blocks; ++b) s += (double)partial[b * SEUNET_LOSS_NSUMS + k];
  sums[k] = s;
}

// g_pred[i] = g_scale * ( c_dice*d dice/dp + c_gul*d gul/dp + c_atr*d atr/dp ) [* p(1-p)]
__global__ void __launch_bounds__(256)
loss_grad_kernel(const float* __restrict__ pred, int apply_sigmoid, const float* __restrict__ target,
                 const float* __restrict__ weight, const float* __restrict__ skel, long long n,
                 const double* __restrict__ sums, float c_dice, float c_gul, float c_atr, float g_scale,
                 const float* __restrict__ g_scale_dev, float* __restrict__ g_pred) {
  if (g_scale_dev) g_scale *= g_scale_dev[0];
  // dice = 1 - (2I+1)/(P+T+1)
  const float dA = (float)(2.0 * sums[0] + 1.0), dB = (float)(sums[1] + sums[2] + 1.0);
  // gul = 1 - (A+1)/(B+1)
  const float gA = (float)(sums[3] + 1.0), gB = (float)(sums[4] + 1.0);
  // atr = 1 - (C1+1)/(C2+1)
  const float aA = (float)(sums[5] + 1.0), aB = (float)(sums[6] + 1.0);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float p = pred[i];
    if (apply_sigmoid) p = 1.f / (1.f + expf(-p));
    const float t = target[i];
    float g = 0.f;
    if (c_dice != 0.f) g += c_dice * (-(2.f * t * dB - dA) / (dB * dB));
    if (c_gul != 0.f) {
      const float w = weight ? weight[i] : 1.f;
      const float dnum = (t != 0.f) ? 0.7f * w * t * powf(p + 1e-4f, -0.3f) : 0.f;
      g += c_gul * (-(dnum * gB - gA * 0.2f * w) / (gB * gB));
    }
    if (c_atr != 0.f) {
      const float w = weight ? weight[i] : 1.f;
      const float sk = skel ? skel[i] : 0.f;
      g += c_atr * (-(w * sk * sk * aB - aA * w * sk) / (aB * aB));
    }
    g *= g_scale;
    if (apply_sigmoid) g *= p * (1.f - p);
    g_pred[i] = g;
  }
}

int launch_loss_sums(const float* pred, int apply_sigmoid, const float* target, const float* weight,
                     const float* skel, long long n, float* partial, double* sums, hipStream_t s) {
  loss_sums_kernel<<<LOSS_BLOCKS, 256, 0, s>>>(pred, apply_sigmoid, target, weight, skel, n, partial);
  loss_sums_final_kernel<<<1, 64, 0, s>>>(partial, LOSS_BLOCKS, sums);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_loss_grad(const float* pred, int apply_sigmoid, const float* target, const float* weight,
                     const float* skel, long long n, const double* sums, float c_dice, float c_gul,
                     float c_atr, float g_scale, const float* g_scale_dev, float* g_pred, hipStream_t s) {
  long long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  loss_grad_kernel<<<(int)g, 256, 0, s>>>(pred, apply_sigmoid, target, weight, skel, n, sums, c_dice, c_gul,
                                           c_atr, g_scale, g_scale_dev, g_pred);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace seunet
