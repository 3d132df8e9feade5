// Fused multi-tensor AdamW step.  Reference: the training loops call
// torch.optim.AdamW(model.parameters(), lr=0.0001) with default betas / eps / weight_decay (train.py:188,386,569) and
// optimizer.step() once per iteration (train.py:245-247,437-439,601-603): 117 small tensors, which eager PyTorch
// updates with ~18 multi-tensor launches.  Here every tensor of a step is updated by a handful of launches (up to
// 24 tensors per launch, their pointers travel in the kernel arguments: no device-side table, no workspace).
//
// Per element, decoupled weight decay and bias-corrected moments (torch/optim/adamw.py, _single_tensor_adamw; the
// reference pins pytorch 2.0.1, environment.yml:78):
//     p  <- p * (1 - lr * wd)
//     m  <- b1 * m + (1 - b1) * g
//     v  <- b2 * v + (1 - b2) * g * g
//     p  <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// HBM-bound: 16 B read + 12 B written per parameter (6 MB of parameters: microseconds); f32 arithmetic.
#include "seunet_common.h"

namespace seunet {

constexpr int AW_MAX = 24;      // tensors per launch
constexpr int AW_CHUNK = 1024;  // elements per block
struct AdamwArgs {
  float* p[AW_MAX];
  const float* g[AW_MAX];
  float* m[AW_MAX];
  float* v[AW_MAX];
  long long count[AW_MAX];
  int first_block[AW_MAX + 1];   // prefix sums of ceil(count / AW_CHUNK)
  int n;
  float b1, b2, omb1, omb2, eps, decay, step_size, inv_sqrt_bc2, gsign;   // derived constants are formed in f64 on the host
};

__global__ void __launch_bounds__(256) adamw_kernel(AdamwArgs a) {
  // which tensor does this block belong to (n <= 24: linear scan on scalars)
  int t = 0;
  while (t + 1 < a.n && (int)blockIdx.x >= a.first_block[t + 1]) ++t;
  const long long base = (long long)((int)blockIdx.x - a.first_block[t]) * AW_CHUNK;
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  const long long cnt = a.count[t];
  const float decay = a.decay, step_size = a.step_size;
#pragma unroll
  for (int k = 0; k < AW_CHUNK / 256; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    if (i < cnt) {
      const float gi = a.gsign * g[i];
      const float pi = p[i] * decay;
      const float mi = a.b1 * m[i] + a.omb1 * gi;
      const float vi = a.b2 * v[i] + a.omb2 * gi * gi;
      const float denom = sqrtf(vi) * a.inv_sqrt_bc2 + a.eps;
      m[i] = mi;
      v[i] = vi;
      p[i] = pi - step_size * (mi / denom);
    }
  }
}

int launch_adamw(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 const long long* counts, int n, double lr, double beta1, double beta2, double eps, double weight_decay,
                 int step, int maximize, hipStream_t s) {
  SEUNET_CHECK(n >= 0 && step >= 1, "adamw: n=%d step=%d (step counts from 1)", n, step);
  SEUNET_CHECK(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && lr >= 0.0 && weight_decay >= 0.0,
               "adamw: invalid hyper-parameter");
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  for (int base = 0; base < n; base += AW_MAX) {
    AdamwArgs a{};
    a.n = n - base < AW_MAX ? n - base : AW_MAX;
    long long blocks = 0;
    for (int i = 0; i < a.n; ++i) {
      SEUNET_CHECK(params[base + i] && grads[base + i] && exp_avg[base + i] && exp_avg_sq[base + i] && counts[base + i] >= 0,
                   "adamw: tensor %d has a null pointer or a negative size", base + i);
      a.p[i] = params[base + i]; a.g[i] = grads[base + i]; a.m[i] = exp_avg[base + i]; a.v[i] = exp_avg_sq[base + i];
      a.count[i] = counts[base + i];
      a.first_block[i] = (int)blocks;
      blocks += (counts[base + i] + AW_CHUNK - 1) / AW_CHUNK;
      SEUNET_CHECK(blocks < (1LL << 31), "adamw: too many elements in one launch");
    }
    a.first_block[a.n] = (int)blocks;
    a.b1 = (float)beta1; a.b2 = (float)beta2; a.eps = (float)eps;
    a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
    a.decay = (float)(1.0 - lr * weight_decay);
    a.step_size = (float)(lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.gsign = maximize ? -1.f : 1.f;
    if (blocks > 0) adamw_kernel<<<(unsigned)blocks, 256, 0, s>>>(a);
  }
  SEUNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace seunet
