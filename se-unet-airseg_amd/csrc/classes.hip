// n_classes > 1 (reference SE_UNet.py:100,150-151: dc0_0 = Conv3d(24, n_classes, 1), dc0_1 = Conv3d(12, n_classes, 1)).
//
// No reference caller builds the network with more than one class, so the single-class path keeps its fused form (the gated
// blocks' epilogues accumulate head_w * drop * side straight into ONE level map per head and level, and their backward takes
// the level map's gradient).  With K classes the heads are rank-K in the side maps, and this general path trades bytes for
// reuse of the same kernels:
//   forward : the gated-block epilogue writes its 2-channel side map (f32, native resolution) instead of a level map;
//             `side_to_level_kernel` then accumulates  L_c += sum_k head_w[c][k] * drop[n][k] * side_k  for every class c;
//             the head kernel (interpolation of the level maps + bias) runs once per (sample, class);
//   backward: the head backward runs once per (sample, class); `level_to_side_grad_kernel` folds the K level-map gradients of a
//             block's level back into the gradient of its side map,  g_side_k = drop[n][k] * sum_c head_w[c][k] * g_c,  and sums
//             the head-weight gradient  d head_w[c][k] = sum_{n,v} g_c * drop[n][k] * side_k  (f64, one record per block, fixed
//             order); the block's backward passes then take `g_side` (the path the stand-alone block modules use).
#include "seunet_common.h"

namespace seunet {

static constexpr int CL_MAXK = 8;

struct ClassHead {
  const float* head_w;     // + 2 * m: this block's two weights of class 0; class c at + c * wstride
  int wstride;             // 24 | 12
  const float* drop;       // [N][drop_stride] + 2 * m, or null
  int drop_stride;
  int K;
};

// level[(c * N + n) * V + v]  (+)=  hw[c][0] * d0 * s0 + hw[c][1] * d1 * s1
__global__ void __launch_bounds__(256)
side_to_level_kernel(const float* __restrict__ side, ClassHead h, float* __restrict__ level, long long V, int N, int accumulate) {
  const int n = blockIdx.y;
  const float d0 = h.drop ? h.drop[n * h.drop_stride] : 1.f, d1 = h.drop ? h.drop[n * h.drop_stride + 1] : 1.f;
  float w0[CL_MAXK], w1[CL_MAXK];
#pragma unroll
  for (int c = 0; c < CL_MAXK; ++c) {
    w0[c] = c < h.K ? h.head_w[c * h.wstride] * d0 : 0.f;       // (same association as the fused epilogue: (w * drop) * side)
    w1[c] = c < h.K ? h.head_w[c * h.wstride + 1] * d1 : 0.f;
  }
  for (long long v = blockIdx.x * 256ll + threadIdx.x; v < V; v += (long long)gridDim.x * 256) {
    const float2 s = reinterpret_cast<const float2*>(side)[(long long)n * V + v];
#pragma unroll
    for (int c = 0; c < CL_MAXK; ++c) {
      if (c >= h.K) break;
      float* dst = level + ((long long)c * N + n) * V + v;
      const float t = w0[c] * s.x + w1[c] * s.y;
      *dst = accumulate ? *dst + t : t;
    }
  }
}

// g_side[(n * V + v) * 2 + k] = d_k * sum_c hw[c][k] * g_c[n][v];  partial[(n * P + block) * 2 * K + c * 2 + k] = sum_v g_c * d_k * side_k
// g_c[n][v] = glev[c * cstride + n * nstride + v]
__global__ void __launch_bounds__(256)
level_to_side_grad_kernel(const float* __restrict__ glev, long long cstride, long long nstride, const float* __restrict__ side,
                          ClassHead h, float* __restrict__ g_side, double* __restrict__ partial, long long V) {
  const int n = blockIdx.y, P = gridDim.x;
  const float d0 = h.drop ? h.drop[n * h.drop_stride] : 1.f, d1 = h.drop ? h.drop[n * h.drop_stride + 1] : 1.f;
  float w0[CL_MAXK], w1[CL_MAXK];
  double a0[CL_MAXK], a1[CL_MAXK];
#pragma unroll
  for (int c = 0; c < CL_MAXK; ++c) {
    w0[c] = c < h.K ? h.head_w[c * h.wstride] * d0 : 0.f;
    w1[c] = c < h.K ? h.head_w[c * h.wstride + 1] * d1 : 0.f;
    a0[c] = a1[c] = 0.0;
  }
  for (long long v = blockIdx.x * 256ll + threadIdx.x; v < V; v += (long long)P * 256) {
    const float2 s = reinterpret_cast<const float2*>(side)[(long long)n * V + v];
    float g0 = 0.f, g1 = 0.f;
#pragma unroll
    for (int c = 0; c < CL_MAXK; ++c) {
      if (c >= h.K) break;
      const float g = glev[(long long)c * cstride + (long long)n * nstride + v];
      g0 += w0[c] * g;
      g1 += w1[c] * g;
      a0[c] += (double)(g * d0) * (double)s.x;
      a1[c] += (double)(g * d1) * (double)s.y;
    }
    reinterpret_cast<float2*>(g_side)[(long long)n * V + v] = make_float2(g0, g1);
  }
  __shared__ double red[4][2 * CL_MAXK];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < CL_MAXK; ++c) {
    double r0 = a0[c], r1 = a1[c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { r0 += shfl_xor_settled(r0, off); r1 += shfl_xor_settled(r1, off); }
    if (lane == 0) { red[wave][2 * c] = r0; red[wave][2 * c + 1] = r1; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * h.K) {
    const int k = threadIdx.x;
    partial[((long long)n * P + blockIdx.x) * (2 * h.K) + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
  }
}

// d head_w[c * wstride + k] = fixed-order sum of the block records
__global__ void __launch_bounds__(64)
class_head_grad_reduce_kernel(const double* __restrict__ partial, int records, int K, int wstride, float* __restrict__ dhead) {
  const int k = threadIdx.x;
  if (k >= 2 * K) return;
  double s = 0.0;
  for (int r = 0; r < records; ++r) s += partial[(long long)r * (2 * K) + k];
  dhead[(k >> 1) * wstride + (k & 1)] = (float)s;
}

// out[c] = sum_n in[n * K + c]
__global__ void __launch_bounds__(64)
class_bias_grad_kernel(const float* __restrict__ in, int N, int K, float* __restrict__ out) {
  const int c = threadIdx.x;
  if (c >= K) return;
  double s = 0.0;
  for (int n = 0; n < N; ++n) s += (double)in[n * K + c];
  out[c] = (float)s;
}

int class_max() { return CL_MAXK; }
int class_grad_records(Dims d) { long long b = (d.vox() + 4095) / 4096; return (int)(b < 1 ? 1 : (b > 256 ? 256 : b)); }

int launch_side_to_level(const float* side, const float* head_w, int wstride, const float* drop, int drop_stride, int K, float* level,
                         int accumulate, Dims d, hipStream_t s) {
  SEUNET_CHECK(K >= 1 && K <= CL_MAXK, "n_classes %d: the general head path handles up to %d classes", K, CL_MAXK);
  ClassHead h{head_w, wstride, drop, drop_stride, K};
  dim3 grid((unsigned)class_grad_records(d) * 4, (unsigned)d.N);
  side_to_level_kernel<<<grid, 256, 0, s>>>(side, h, level, d.vox(), d.N, accumulate);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_level_to_side_grad(const float* glev, long long cstride, long long nstride, const float* side, const float* head_w, int wstride,
                              const float* drop, int drop_stride, int K, float* g_side, double* partial, float* dhead, Dims d,
                              hipStream_t s) {
  SEUNET_CHECK(K >= 1 && K <= CL_MAXK, "n_classes %d: the general head path handles up to %d classes", K, CL_MAXK);
  ClassHead h{head_w, wstride, drop, drop_stride, K};
  const int P = class_grad_records(d);
  dim3 grid((unsigned)P, (unsigned)d.N);
  level_to_side_grad_kernel<<<grid, 256, 0, s>>>(glev, cstride, nstride, side, h, g_side, partial, d.vox());
  if (dhead != nullptr) class_head_grad_reduce_kernel<<<1, 64, 0, s>>>(partial, P * d.N, K, wstride, dhead);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_class_bias_grad(const float* per_sample, int N, int K, float* out, hipStream_t s) {
  class_bias_grad_kernel<<<1, 64, 0, s>>>(per_sample, N, K, out);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace seunet
