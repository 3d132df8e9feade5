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

static constexpr int LOSS_BLOCKS = 1024;   // 4 per CU; each thread handles n / (1024 * 256) elements, 4 per load
int loss_partials() { return LOSS_BLOCKS; }

// TERMS: bit 0 dice (sums 0-2), bit 1 general union (3-4), bit 2 ATR (5-6); sums of terms that are not asked for stay 0.
// (The pow of the general-union term costs more than everything else together, and a wave pays it as soon as ONE of its 256
// voxels is foreground: a Dice-only stage-1 step should not.)
template <int TERMS>
__device__ __forceinline__ void loss_terms(float p, int apply_sigmoid, float t, float w, float sk, float (&s)[SEUNET_LOSS_NSUMS]) {
  if (apply_sigmoid) p = 1.f / (1.f + expf(-p));
  if (TERMS & 1) {
    s[0] += p * t;
    s[1] += p;
    s[2] += t;
  }
  if (TERMS & 2) {
    if (t != 0.f) s[3] += w * powf(p + 1e-4f, 0.7f) * t;   // (the label is sparse: the pow is skipped where it is multiplied by 0)
    s[4] += w * (0.2f * p + 0.8f * t);
  }
  if (TERMS & 4) {
    const float ps = p * sk;
    s[5] += w * ps * sk;
    s[6] += w * (ps + sk);
  }
}

template <int TERMS>
__global__ void __launch_bounds__(256)
loss_sums_kernel(const float* __restrict__ pred, int apply_sigmoid, const float* __restrict__ target,
                 const float* __restrict__ weight, const float* __restrict__ skel, long long n,
                 float* __restrict__ partial) {
  float s[SEUNET_LOSS_NSUMS];
#pragma unroll
  for (int k = 0; k < SEUNET_LOSS_NSUMS; ++k) s[k] = 0.f;
  const bool vec = (n & 3) == 0 && ((reinterpret_cast<size_t>(pred) | reinterpret_cast<size_t>(target) |
                                     reinterpret_cast<size_t>(weight) | reinterpret_cast<size_t>(skel)) & 15) == 0;
  if (vec) {   // 16-byte loads
    const long long n4 = n >> 2;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
      const float4 p = reinterpret_cast<const float4*>(pred)[i], t = reinterpret_cast<const float4*>(target)[i];
      const float4 w = weight ? reinterpret_cast<const float4*>(weight)[i] : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 k = skel ? reinterpret_cast<const float4*>(skel)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      loss_terms<TERMS>(p.x, apply_sigmoid, t.x, w.x, k.x, s);
      loss_terms<TERMS>(p.y, apply_sigmoid, t.y, w.y, k.y, s);
      loss_terms<TERMS>(p.z, apply_sigmoid, t.z, w.z, k.z, s);
      loss_terms<TERMS>(p.w, apply_sigmoid, t.w, w.w, k.w, s);
    }
  } else {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
      loss_terms<TERMS>(pred[i], apply_sigmoid, target[i], weight ? weight[i] : 1.f, skel ? skel[i] : 0.f, s);
  }
  __shared__ float red[4][SEUNET_LOSS_NSUMS];
#pragma unroll
  for (int k = 0; k < SEUNET_LOSS_NSUMS; ++k) {
    float v = s[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += shfl_xor_settled(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < SEUNET_LOSS_NSUMS) {
    const int k = threadIdx.x;
    partial[blockIdx.x * SEUNET_LOSS_NSUMS + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
  }
}

// one wave per sum: lane l adds the block partials l, l+64, ... in f64, then a fixed-order butterfly
__global__ void __launch_bounds__(64 * SEUNET_LOSS_NSUMS)
loss_sums_final_kernel(const float* __restrict__ partial, int blocks, double* __restrict__ sums) {
  const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s = 0.0;
  for (int b = lane; b < blocks; b += 64) s += (double)partial[b * SEUNET_LOSS_NSUMS + k];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += shfl_xor_settled(s, off);
  if (lane == 0) sums[k] = s;
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
  auto grad1 = [&](float p, float t, float w, float sk) -> float {
    if (apply_sigmoid) p = 1.f / (1.f + expf(-p));
    float g = 0.f;
    if (c_dice != 0.f) g += c_dice * (-(2.f * t * dB - dA) / (dB * dB));
    if (c_gul != 0.f) {
      const float dnum = (t != 0.f) ? 0.7f * w * t * powf(p + 1e-4f, -0.3f) : 0.f;
      g += c_gul * (-(dnum * gB - gA * 0.2f * w) / (gB * gB));
    }
    if (c_atr != 0.f) g += c_atr * (-(w * sk * sk * aB - aA * w * sk) / (aB * aB));
    g *= g_scale;
    if (apply_sigmoid) g *= p * (1.f - p);
    return g;
  };
  const bool vec = (n & 3) == 0 && ((reinterpret_cast<size_t>(pred) | reinterpret_cast<size_t>(target) | reinterpret_cast<size_t>(weight) |
                                     reinterpret_cast<size_t>(skel) | reinterpret_cast<size_t>(g_pred)) & 15) == 0;
  if (vec) {
    const long long n4 = n >> 2;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
      const float4 p = reinterpret_cast<const float4*>(pred)[i], t = reinterpret_cast<const float4*>(target)[i];
      const float4 w = weight ? reinterpret_cast<const float4*>(weight)[i] : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 k = skel ? reinterpret_cast<const float4*>(skel)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      reinterpret_cast<float4*>(g_pred)[i] = make_float4(grad1(p.x, t.x, w.x, k.x), grad1(p.y, t.y, w.y, k.y),
                                                         grad1(p.z, t.z, w.z, k.z), grad1(p.w, t.w, w.w, k.w));
    }
  } else {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
      g_pred[i] = grad1(pred[i], target[i], weight ? weight[i] : 1.f, skel ? skel[i] : 0.f);
  }
}

int launch_loss_sums(const float* pred, int apply_sigmoid, const float* target, const float* weight,
                     const float* skel, long long n, float* partial, double* sums, hipStream_t s, int terms) {
  terms &= 7;
  if (terms == 0) terms = 7;
  switch (terms) {
    case 1: loss_sums_kernel<1><<<LOSS_BLOCKS, 256, 0, s>>>(pred, apply_sigmoid, target, weight, skel, n, partial); break;
    case 2: loss_sums_kernel<2><<<LOSS_BLOCKS, 256, 0, s>>>(pred, apply_sigmoid, target, weight, skel, n, partial); break;
    case 6: loss_sums_kernel<6><<<LOSS_BLOCKS, 256, 0, s>>>(pred, apply_sigmoid, target, weight, skel, n, partial); break;
    default: loss_sums_kernel<7><<<LOSS_BLOCKS, 256, 0, s>>>(pred, apply_sigmoid, target, weight, skel, n, partial); break;
  }
  loss_sums_final_kernel<<<1, 64 * SEUNET_LOSS_NSUMS, 0, s>>>(partial, LOSS_BLOCKS, sums);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

// loss value from the (already all-reduced) sums of one or two heads, formed on the device in f64 and rounded to f32 per
// head like the reference's scalar arithmetic (train.py:51-76 on whole-batch sums): value = f32(head 0) + f32(head 1)
__global__ void loss_value_kernel(const double* __restrict__ s0, double d0, double g0, double a0,
                                  const double* __restrict__ s1, double d1, double g1, double a1, float* __restrict__ value) {
  auto one = [](const double* s, double cd, double cg, double ca) -> float {
    double out = 0.0;
    if (cd != 0.0) out = out + cd * (1.0 - (2.0 * s[0] + 1.0) / (s[1] + s[2] + 1.0));
    if (cg != 0.0) out = out + cg * (1.0 - (s[3] + 1.0) / (s[4] + 1.0));
    if (ca != 0.0) out = out + ca * (1.0 - (s[5] + 1.0) / (s[6] + 1.0));
    return (float)out;
  };
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float v = one(s0, d0, g0, a0);
    if (s1 != nullptr) v = v + one(s1, d1, g1, a1);
    value[0] = v;
  }
}

int launch_loss_value(const double* sums0, double c_dice0, double c_gul0, double c_atr0, const double* sums1, double c_dice1,
                      double c_gul1, double c_atr1, float* value, hipStream_t s) {
  loss_value_kernel<<<1, 64, 0, s>>>(sums0, c_dice0, c_gul0, c_atr0, sums1, c_dice1, c_gul1, c_atr1, value);
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
