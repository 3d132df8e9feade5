// HBM-bound "epilogue" kernels of the SE-UNet blocks (gfx950).
//
//   gated block   (reference SSEConv / SSEConv2, SE_UNet.py:24-35, 68-82):
//       raw conv output -> InstanceNorm -> LeakyReLU -> spatial gate(s) -> e ; side = conv1x1(e)
//   aggregation   (reference CATConv, SE_UNet.py:45-49, and the residual adds at :187,196,205):
//       raw -> InstanceNorm -> LeakyReLU (+ the same for the raw-input "x" branch)
//   and their backward passes (two-phase InstanceNorm backward).
//
// Thread mapping: one lane owns 8 consecutive channels (16 B bf16 / 32 B f32) of one voxel,
// LPV = C/8 consecutive lanes own one voxel, so every global access is a fully coalesced
// 16/32-byte-per-lane stream, software-pipelined one voxel ahead; the per-voxel channel dot products of the gates are
// LPV-lane reductions on DPP (quad_perm / row_half_mirror / row_mirror), the per-(n,c) InstanceNorm sums are f64:
// strided shuffle reductions + a fixed-order cross-wave sum (deterministic, no atomics).
// The "x" branch of an aggregation block (a 1x1x1 conv of the <= 2-channel network input) is recomputed per voxel
// instead of read, its statistics come from the input's second moments, and its weight gradient is accumulated in the
// backward pass B (XR / XW template modes below).
#include "seunet_common.h"
#include <type_traits>

namespace seunet {

// the value a store of type T keeps (round to nearest even for the 16-bit types, the same conversion store8 uses)
template <typename T> __device__ __forceinline__ float round_to(float v) {
  if constexpr (sizeof(T) == 4) return v;
  else return unpack_lo<T>(pack2<T>(v, 0.f));
}

// The spatial gates' sigmoid.  f32 storage (the 1e-3 parity mode): expf and an IEEE division.  16-bit storage: one v_exp_f32 and
// one v_rcp_f32 (1 ulp each) instead of ~25 instructions of range reduction and division fix-up -- the gate multiplies values that
// keep 8 / 11 mantissa bits; forward and both backward passes use the same function, so a gate is the same number everywhere.
template <typename T> __device__ __forceinline__ float gate_sigmoid(float z) {
  if constexpr (sizeof(T) == 4) return 1.f / (1.f + expf(-z));
  else return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * z));
}

static constexpr int EPI_THREADS = 256;

int epi_partials(Dims d) {
  // One partial record per block and sample.  128 voxels per block at least: the coarse levels (32^3, 16^3) were running
  // pass A on 4..32 blocks per sample, 32 dependent iterations each (36 us for a 16^3 x 64-channel tensor).  256 at most
  // (every record is summed again by the finalize kernels: 512 made those 30 % slower for nothing).
  // Wave quantisation: pass A of the one-gate blocks holds 3 workgroups per CU (162 VGPRs), i.e. 768 on the chip; 4 x 256 =
  // 1024 blocks ran as one full round and a third of a second one.  The count per launch (N x partials; the other passes
  // use 4 x as many) is therefore rounded down to a multiple of 768 once it exceeds it: 4 x 192 at the bench shape, measured
  // against 96 / 160 / 224 / 256 / 384 per sample (sum of the epilogue classes 5.64 ms vs 6.12 / 5.90 / 5.99 / 5.83 / 5.86).
  long long v = d.vox();
  long long p = v / 128;
  if (p < 1) p = 1;
  if (p > 256) p = 256;
  const long long n = d.N > 0 ? d.N : 1, round = 3 * 256;
  if (n * p >= round) {
    const long long q = (n * p / round) * round / n;
    if (q >= 1) p = q;
  }
  return (int)p;
}

// ----------------------------------------------------------------------------------
// generic per-(n,c) sum / sum of squares of a channels-last tensor
// ----------------------------------------------------------------------------------
template <typename T, int LPV>
__global__ void __launch_bounds__(EPI_THREADS)
channel_stats_kernel(const T* __restrict__ t, int C, double* __restrict__ partial, long long V) {
  const int n = blockIdx.y, P = gridDim.x;
  const int cg = threadIdx.x % LPV, vb = threadIdx.x / LPV;
  constexpr int VPB = EPI_THREADS / LPV;
  double s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.0; s2[j] = 0.0; }
  for (long long v = (long long)blockIdx.x * VPB + vb; v < V; v += (long long)P * VPB) {
    float x[8];
    load8(t + ((long long)n * V + v) * C + cg * 8, x);
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] += (double)x[j]; s2[j] += (double)x[j] * (double)x[j]; }
  }
  __shared__ double red[4][16][16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    double a = stride_sum_d<LPV>(s1[j]), b = stride_sum_d<LPV>(s2[j]);
    if (lane < LPV) { red[wave][lane][j] = a; red[wave][lane][8 + j] = b; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < LPV * 16; i += EPI_THREADS) {
    const int g = i / 16, k = i % 16;
    const double tot = ((red[0][g][k] + red[1][g][k]) + red[2][g][k]) + red[3][g][k];
    const int c = g * 8 + (k & 7);
    partial[(((long long)n * P + blockIdx.x) * C + c) * 2 + (k >> 3)] = tot;
  }
}

// one 256-thread block per (n, c): sums the f64 partial slots in a fixed order.
//   mode 0: (mean, rstd = 1/sqrt(biased var + eps))      [InstanceNorm3d forward]
//   mode 1: (sum/count, sumsq/count)                     [the two means of the InstanceNorm backward]
__device__ __forceinline__ void stats_finalize_body(int idx, const double* __restrict__ partial, int slots, int C, double inv_count,
                                                    float eps, int mode, float* __restrict__ out_a, float* __restrict__ out_b) {
  const int n = idx / C, c = idx % C;
  double s1 = 0.0, s2 = 0.0;
  for (int p = threadIdx.x; p < slots; p += 256) {
    const double* q = partial + (((long long)n * slots + p) * C + c) * 2;
    s1 += q[0];
    s2 += q[1];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    s1 += shfl_xor_settled(s1, off);
    s2 += shfl_xor_settled(s2, off);
  }
  __shared__ double w1[4], w2[4];
  if ((threadIdx.x & 63) == 0) { w1[threadIdx.x >> 6] = s1; w2[threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s1 = ((w1[0] + w1[1]) + w1[2]) + w1[3];
    s2 = ((w2[0] + w2[1]) + w2[2]) + w2[3];
    if (mode == 0) {
      const double mean = s1 * inv_count;
      double var = s2 * inv_count - mean * mean;
      if (var < 0.0) var = 0.0;
      out_a[idx] = (float)mean;
      out_b[idx] = (float)(1.0 / sqrt(var + (double)eps));
    } else {
      out_a[idx] = (float)(s1 * inv_count);
      out_b[idx] = (float)(s2 * inv_count);
    }
  }
}
__global__ void __launch_bounds__(256)
stats_finalize_kernel(const double* __restrict__ partial, int slots, int C, int N, double inv_count,
                      float eps, int mode, float* __restrict__ out_a, float* __restrict__ out_b) {
  stats_finalize_body(blockIdx.x, partial, slots, C, inv_count, eps, mode, out_a, out_b);
}

// ----------------------------------------------------------------------------------
// gated block, forward
// ----------------------------------------------------------------------------------
template <typename T, int LPV, bool G2>
__global__ void __launch_bounds__(EPI_THREADS)
sse_fwd_kernel(const T* __restrict__ raw, const float* __restrict__ mean,
               const float* __restrict__ rstd, int C, SseParams p, T* __restrict__ e_out,
               SseHead head, long long V) {
  const int n = blockIdx.y, P = gridDim.x;
  const int cg = threadIdx.x % LPV, vb = threadIdx.x / LPV;
  constexpr int VPB = EPI_THREADS / LPV;
  const int c0 = cg * 8;
  float mu[8], rs[8], wse[8], wse2[8], w20[8], w21[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mu[j] = mean[n * C + c0 + j];
    rs[j] = rstd[n * C + c0 + j];
    wse[j] = p.w_se[c0 + j];
    wse2[j] = G2 ? p.w_se2[c0 + j] : 0.f;
    w20[j] = p.w_side[c0 + j];
    w21[j] = p.w_side[C + c0 + j];
  }
  const float b20 = p.b_side[0], b21 = p.b_side[1], slope = p.slope;
  const bool want_side = head.side_out != nullptr || head.level_map != nullptr;
  float hw0 = 0.f, hw1 = 0.f;
  if (head.level_map) {
    hw0 = head.head_w[0] * (head.drop ? head.drop[n * head.drop_stride + 0] : 1.f);
    hw1 = head.head_w[1] * (head.drop ? head.drop[n * head.drop_stride + 1] : 1.f);
  }
  const long long stride = (long long)P * VPB;
  long long v = (long long)blockIdx.x * VPB + vb;
  Pack8<T> nx;   // software pipeline: voxel v + stride is loaded before voxel v is computed
  zero8p(nx);
  if (v < V) load8p(raw + ((long long)n * V + v) * C + c0, nx);
  for (; v < V; v += stride) {
    const long long vi = (long long)n * V + v;
    float x[8], a[8], e[8];
    unpack8(nx, x);
    if (v + stride < V) load8p(raw + (vi + stride) * C + c0, nx);
    float d1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (x[j] - mu[j]) * rs[j];
      a[j] = xh > 0.f ? xh : xh * slope;
      d1 += wse[j] * a[j];
    }
    const float g1 = gate_sigmoid<T>(group_sum<LPV>(d1));
    float d2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      e[j] = a[j] * g1;
      d2 += wse2[j] * e[j];
    }
    if (G2) {
      const float g2 = gate_sigmoid<T>(group_sum<LPV>(d2));
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] *= g2;
    }
    store8(e_out + vi * C + c0, e);
    if (want_side) {     // (block-uniform; off for the encoder blocks of an inference forward that discards the encoder head)
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { s0 += w20[j] * e[j]; s1 += w21[j] * e[j]; }
      s0 = group_sum<LPV>(s0) + b20;
      s1 = group_sum<LPV>(s1) + b21;
      if (cg == 0) {
        if (head.side_out) { head.side_out[vi * 2] = s0; head.side_out[vi * 2 + 1] = s1; }
        if (head.level_map) {
          const float t = hw0 * s0 + hw1 * s1;
          head.level_map[vi] = head.level_accumulate ? head.level_map[vi] + t : t;
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------------
// gated block, backward.  The gradient w.r.t. the normalised activation (dxhat) is recomputed from the
// saved raw conv output in both passes and never stored:
//   APPLY = false (pass A): per-(n,c) sums of dxhat and dxhat*xhat (f64: the loss gradient has a large
//                           common-mode part that InstanceNorm's backward cancels, so f32 sums are not
//                           enough) + the gate / side / head parameter gradients
//   APPLY = true  (pass B): draw = rstd * (dxhat - m1 - xhat * m2), rounded once, stored over g_e
// ----------------------------------------------------------------------------------
// LEVEL: the side gradient arrives as ONE value per voxel, the gradient g of the head's level map (training: always), so
//   d side_k = hw_k * g with hw_k = head weight x DropLayer scale of the sample.  Everything that is linear in it is then taken out
//   of the voxel loop: de += g * (w20 hw0 + w21 hw1) with the bracket formed once per thread, and the gradients of the side conv,
//   its bias and the head weights all follow from G[c] = sum_v g e[c] and sum_v g at the end of the block (d w2k[c] = hw_k G[c],
//   d b2k = hw_k sum g, d head_k = drop_k (sum_c w2k[c] G[c] + b2k sum g)) -- no per-voxel side values, no second accumulator set.
//   Pass A of the one-gate C = 32 block: 266 -> ~200 instructions per voxel group, under its HBM time.
template <typename T, int LPV, bool G2, bool APPLY, bool LEVEL>
__global__ void __launch_bounds__(EPI_THREADS, (APPLY || G2) ? 1 : 3)
sse_bwd_kernel(const T* __restrict__ raw, const float* __restrict__ mean,
               const float* __restrict__ rstd, int C, SseParams p, SseBwdIn g, SseHead head,
               const float* __restrict__ m1p, const float* __restrict__ m2p,
               T* dxhat_out, double* __restrict__ stat_partial,
               float* __restrict__ pgrad_partial, long long V) {
  const int n = blockIdx.y, P = gridDim.x;
  const int cg = threadIdx.x % LPV, vb = threadIdx.x / LPV;
  constexpr int VPB = EPI_THREADS / LPV;
  const int c0 = cg * 8;
  float mu[8], rs[8], wse[8], wse2[8], w20[8], w21[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mu[j] = mean[n * C + c0 + j];
    rs[j] = rstd[n * C + c0 + j];
    wse[j] = p.w_se[c0 + j];
    wse2[j] = G2 ? p.w_se2[c0 + j] : 0.f;
    w20[j] = p.w_side[c0 + j];
    w21[j] = p.w_side[C + c0 + j];
  }
  const float b20 = p.b_side[0], b21 = p.b_side[1], slope = p.slope;
  float dr0 = 1.f, dr1 = 1.f, hw0 = 0.f, hw1 = 0.f;
  if (g.g_level) {
    if (head.drop) { dr0 = head.drop[n * head.drop_stride]; dr1 = head.drop[n * head.drop_stride + 1]; }
    hw0 = head.head_w[0] * dr0;
    hw1 = head.head_w[1] * dr1;
  }
  // f64 sums live in thread-private LDS slots (pass A only): 32 fewer VGPRs than register accumulators, which is
  // the difference between 2 and 3 waves per SIMD for this latency-bound loop
  // bf16 activations: the thread's <= ~130 voxels are summed in f32 registers and converted once (the tensors keep 8
  // mantissa bits; gate = the bf16-autocast comparison), which frees the 32 KB of LDS slots -> twice the blocks per CU
  constexpr bool F64ACC = !APPLY && sizeof(T) == 4;   // (bf16 / f16 storage: f32 thread sums)
  __shared__ double acc64[F64ACC ? 16 : 1][F64ACC ? EPI_THREADS : 1];
  float fdx[8], fdxx[8];   // f32 staging of the f64 sums, flushed every 8 voxels
  float awse[8], awse2[8], aw20[8], aw21[8], am1[8], am2[8];
  int since_flush = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (F64ACC) acc64[j][threadIdx.x] = acc64[8 + j][threadIdx.x] = 0.0;
    fdx[j] = fdxx[j] = 0.f;
    awse[j] = awse2[j] = aw20[j] = aw21[j] = 0.f;
    am1[j] = APPLY ? m1p[n * C + c0 + j] : 0.f;
    am2[j] = APPLY ? m2p[n * C + c0 + j] : 0.f;
  }
  float adb0 = 0.f, adb1 = 0.f, adh0 = 0.f, adh1 = 0.f;
  float wc[8], sgl = 0.f;      // LEVEL: w20 hw0 + w21 hw1; sum of g (aw20 doubles as G)
#pragma unroll
  for (int j = 0; j < 8; ++j) wc[j] = w20[j] * hw0 + w21[j] * hw1;

  // software pipeline: the loads of voxel v + stride are issued before voxel v is computed
  const long long stride = (long long)P * VPB;
  long long v = (long long)blockIdx.x * VPB + vb;
  Pack8<T> nx, nde;
  float ngl = 0.f, ns0 = 0.f, ns1 = 0.f;
  zero8p(nx); zero8p(nde);
#define SSE_BWD_FETCH(vv)                                                                  \
  do {                                                                                     \
    const long long fi_ = (long long)n * V + (vv);                                         \
    load8p(raw + fi_ * C + c0, nx);                                                        \
    if (g.g_e) load8p(reinterpret_cast<const T*>(g.g_e) + fi_ * C + c0, nde);              \
    if (g.g_level) ngl = g.g_level[fi_];                                                   \
    else if (g.g_side) { ns0 = g.g_side[fi_ * 2]; ns1 = g.g_side[fi_ * 2 + 1]; }           \
  } while (0)
  if (v < V) SSE_BWD_FETCH(v);
  for (; v < V; v += stride) {
    const long long vi = (long long)n * V + v;
    float x[8], xh[8], a[8], b[8], e[8], de[8];
    unpack8(nx, x);
    unpack8(nde, de);
    const float gl = ngl, gs0 = ns0, gs1 = ns1;
    if (v + stride < V) SSE_BWD_FETCH(v + stride);
    float d1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xh[j] = (x[j] - mu[j]) * rs[j];
      a[j] = xh[j] > 0.f ? xh[j] : xh[j] * slope;
      d1 += wse[j] * a[j];
    }
    const float g1 = gate_sigmoid<T>(group_sum<LPV>(d1));
    float d2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { b[j] = a[j] * g1; d2 += wse2[j] * b[j]; }
    float g2 = 1.f;
    if (G2) g2 = gate_sigmoid<T>(group_sum<LPV>(d2));
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = G2 ? b[j] * g2 : b[j];

    // gradient arriving through the 2-channel side output
    float t2 = 0.f;
    if (LEVEL) {
      if (!APPLY) sgl += gl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        de[j] += gl * wc[j];
        if (!APPLY) aw20[j] += gl * e[j];
        t2 += de[j] * b[j];
      }
    } else {
      float ds0 = 0.f, ds1 = 0.f;
      if (g.g_level) {
        ds0 = hw0 * gl;
        ds1 = hw1 * gl;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { s0 += w20[j] * e[j]; s1 += w21[j] * e[j]; }
        s0 = group_sum<LPV>(s0) + b20;
        s1 = group_sum<LPV>(s1) + b21;
        if (!APPLY && cg == 0) { adh0 += gl * dr0 * s0; adh1 += gl * dr1 * s1; }
      } else if (g.g_side) {
        ds0 = gs0;
        ds1 = gs1;
      }
      if (cg == 0) { adb0 += ds0; adb1 += ds1; }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        de[j] += w20[j] * ds0 + w21[j] * ds1;
        aw20[j] += ds0 * e[j];
        aw21[j] += ds1 * e[j];
        t2 += de[j] * b[j];
      }
    }
    if (G2) {  // e = b * g2, g2 = sigmoid(<w_se2, b>)
      const float q2 = group_sum<LPV>(t2) * g2 * (1.f - g2);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        awse2[j] += q2 * b[j];
        de[j] = de[j] * g2 + q2 * wse2[j];  // now d/db
      }
    }
    float t1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t1 += de[j] * a[j];
    const float q1 = group_sum<LPV>(t1) * g1 * (1.f - g1);  // b = a * g1, g1 = sigmoid(<w_se, a>)
    float dxh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      awse[j] += q1 * a[j];
      const float da = de[j] * g1 + q1 * wse[j];
      dxh[j] = da * (xh[j] > 0.f ? 1.f : slope);
      if (APPLY) {
        dxh[j] = rs[j] * (dxh[j] - am1[j] - xh[j] * am2[j]);
      } else {
        fdx[j] += dxh[j];
        fdxx[j] += dxh[j] * xh[j];
      }
    }
    if (APPLY) store8(dxhat_out + vi * C + c0, dxh);
    else if (F64ACC && ++since_flush == 8) {
      since_flush = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc64[j][threadIdx.x] += (double)fdx[j];
        acc64[8 + j][threadIdx.x] += (double)fdxx[j];
        fdx[j] = fdxx[j] = 0.f;
      }
    }
  }
#undef SSE_BWD_FETCH
  if (APPLY) return;
  if (LEVEL) {     // aw20 holds G[c] = sum_v g e[c]: the side / bias / head gradients of this thread's voxels follow from it
    float h0 = cg == 0 ? b20 * sgl : 0.f, h1 = cg == 0 ? b21 * sgl : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float G = aw20[j];
      h0 += w20[j] * G;
      h1 += w21[j] * G;
      aw20[j] = hw0 * G;
      aw21[j] = hw1 * G;
    }
    adh0 = dr0 * h0;
    adh1 = dr1 * h1;
    adb0 = cg == 0 ? hw0 * sgl : 0.f;
    adb1 = cg == 0 ? hw1 * sgl : 0.f;
  }
  double sdx[8], sdxx[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sdx[j] = (F64ACC ? acc64[j][threadIdx.x] : 0.0) + (double)fdx[j];
    sdxx[j] = (F64ACC ? acc64[8 + j][threadIdx.x] : 0.0) + (double)fdxx[j];
  }

  // ---- block reduction (fixed order) ----
  __shared__ double redd[4][16][16];
  __shared__ float red[4][16][32];
  __shared__ float reds[4][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double r0 = stride_sum_d<LPV>(sdx[j]), r1 = stride_sum_d<LPV>(sdxx[j]);
    const float r2 = stride_sum<LPV>(awse[j]), r3 = stride_sum<LPV>(awse2[j]);
    const float r4 = stride_sum<LPV>(aw20[j]), r5 = stride_sum<LPV>(aw21[j]);
    if (lane < LPV) {
      redd[wave][lane][j] = r0;     redd[wave][lane][8 + j] = r1;
      red[wave][lane][j] = r2;      red[wave][lane][8 + j] = r3;
      red[wave][lane][16 + j] = r4; red[wave][lane][24 + j] = r5;
    }
  }
  {
    const float q0 = stride_sum<1>(adb0), q1 = stride_sum<1>(adb1);
    const float q2 = stride_sum<1>(adh0), q3 = stride_sum<1>(adh1);
    if (lane == 0) { reds[wave][0] = q0; reds[wave][1] = q1; reds[wave][2] = q2; reds[wave][3] = q3; }
  }
  __syncthreads();
  const long long rec = (long long)n * P + blockIdx.x;
  float* pg = pgrad_partial + rec * (4 * C + 4);
  for (int i = threadIdx.x; i < LPV * 16; i += EPI_THREADS) {
    const int gq = i / 16, k = i % 16;
    const double tot = ((redd[0][gq][k] + redd[1][gq][k]) + redd[2][gq][k]) + redd[3][gq][k];
    stat_partial[(rec * C + gq * 8 + (k & 7)) * 2 + (k >> 3)] = tot;
  }
  for (int i = threadIdx.x; i < LPV * 32; i += EPI_THREADS) {
    const int gq = i / 32, k = i % 32;
    const float tot = ((red[0][gq][k] + red[1][gq][k]) + red[2][gq][k]) + red[3][gq][k];
    pg[(k >> 3) * C + gq * 8 + (k & 7)] = tot;
  }
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    pg[4 * C + k] = ((reds[0][k] + reds[1][k]) + reds[2][k]) + reds[3][k];
  }
}

// sums the per-block parameter-gradient records; one wave per entry, f64, fixed order
__device__ __forceinline__ void pgrad_reduce_body(int blk, const float* __restrict__ pg, int records, int C, float* dw_se,
                                                  float* dw_se2, float* dw_side, float* db_side, float* dhead_w) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = blk * 4 + wave, K = 4 * C + 4;
  if (k >= K) return;
  double s = 0.0;
  for (int r = lane; r < records; r += 64) s += (double)pg[(long long)r * K + k];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += shfl_xor_settled(s, off);
  if (lane != 0) return;
  const float v = (float)s;
  if (k < C) { if (dw_se) dw_se[k] = v; }
  else if (k < 2 * C) { if (dw_se2) dw_se2[k - C] = v; }
  else if (k < 4 * C) { if (dw_side) dw_side[k - 2 * C] = v; }
  else if (k < 4 * C + 2) { if (db_side) db_side[k - 4 * C] = v; }
  else { if (dhead_w) dhead_w[k - 4 * C - 2] = v; }
}
__global__ void __launch_bounds__(256)
pgrad_reduce_kernel(const float* __restrict__ pg, int records, int C, float* dw_se, float* dw_se2,
                    float* dw_side, float* db_side, float* dhead_w) {
  pgrad_reduce_body(blockIdx.x, pg, records, C, dw_se, dw_se2, dw_side, db_side, dhead_w);
}
// what follows pass A of a gated block, in ONE launch: the two means of the InstanceNorm backward (blocks [0, N*C)) and the
// parameter-gradient records (the remaining blocks) -- two dependent 5-us launches on the critical path otherwise
__global__ void __launch_bounds__(256)
gate_bwd_finalize_kernel(const double* __restrict__ partial, int slots, int C, int N, double inv_count, float* __restrict__ m1,
                         float* __restrict__ m2, const float* __restrict__ pg, int records, float* dw_se, float* dw_se2,
                         float* dw_side, float* db_side, float* dhead_w) {
  if ((int)blockIdx.x < N * C) stats_finalize_body(blockIdx.x, partial, slots, C, inv_count, 0.f, 1, m1, m2);
  else pgrad_reduce_body((int)blockIdx.x - N * C, pg, records, C, dw_se, dw_se2, dw_side, db_side, dhead_w);
}

// ----------------------------------------------------------------------------------
// x-branch statistics without the x-branch tensor.  raw2 = W2 x is linear in the (<= 2-channel) input, so its per-(n,c)
// InstanceNorm statistics follow from the input's first and second moments per sample:
//     mean2[c] = sum_i W2[c][i] m_i,      var2[c] = sum_ij W2[c][i] W2[c][j] (M_ij - m_i m_j)        (f64)
// ----------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(EPI_THREADS)
input_moments_kernel(const T* __restrict__ xin, double* __restrict__ partial, long long V) {
  const int n = blockIdx.y, P = gridDim.x;
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};   // x0, x1, x0^2, x0 x1, x1^2
  for (long long v = (long long)blockIdx.x * EPI_THREADS + threadIdx.x; v < V; v += (long long)P * EPI_THREADS) {
    float x[8];
    load8(xin + ((long long)n * V + v) * 8, x);
    const double a = (double)x[0], b = (double)x[1];
    s[0] += a; s[1] += b; s[2] += a * a; s[3] += a * b; s[4] += b * b;
  }
  __shared__ double red[4][5];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    double r = s[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) r += shfl_xor_settled(r, off);
    if (lane == 0) red[wave][k] = r;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const int k = threadIdx.x;
    partial[((long long)n * P + blockIdx.x) * 5 + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
  }
}

// one 256-thread block per sample: fixed-order sum of the moment partials, then mean / rstd of every output channel
__global__ void __launch_bounds__(256)
xbranch_stats_kernel(const double* __restrict__ partial, int slots, const float* __restrict__ w2, int C, int ic,
                     double inv_count, float eps, float* __restrict__ mean2, float* __restrict__ rstd2,
                     double* __restrict__ moments_out) {
  const int n = blockIdx.x;
  __shared__ double red[4][5], tot[5];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < slots; b += 256)
#pragma unroll
    for (int k = 0; k < 5; ++k) s[k] += partial[((long long)n * slots + b) * 5 + k];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    double r = s[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) r += shfl_xor_settled(r, off);
    if (lane == 0) red[wave][k] = r;
  }
  __syncthreads();
  if (threadIdx.x < 5) tot[threadIdx.x] = (((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]) * inv_count;
  __syncthreads();
  // (kept for the backward pass: the x-branch weight gradient is formed from these and two sums per channel, xw_finalize_kernel)
  if (moments_out != nullptr && threadIdx.x < 5) moments_out[n * 5 + threadIdx.x] = tot[threadIdx.x];
  const double m0 = tot[0], m1 = tot[1];
  const double c00 = tot[2] - m0 * m0, c01 = tot[3] - m0 * m1, c11 = tot[4] - m1 * m1;
  for (int c = threadIdx.x; c < C; c += 256) {
    const double a = (double)w2[c * ic], b = ic > 1 ? (double)w2[c * ic + 1] : 0.0;
    const double mu = a * m0 + b * m1;
    double var = a * a * c00 + 2.0 * a * b * c01 + b * b * c11;
    if (var < 0.0) var = 0.0;
    mean2[n * C + c] = (float)mu;
    rstd2[n * C + c] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// ----------------------------------------------------------------------------------
// aggregation block (1x1x1 conv output -> IN -> LeakyReLU, optional second branch added)
// ----------------------------------------------------------------------------------
// XR (x-branch recompute): the second branch is the 1x1x1 conv of the <= 2-channel network input (x33 / x63 / x93,
//     SE_UNet.py:112,118,124).  Its raw output is never stored: `raw2` then points at the packed 8-channel INPUT
//     [N][V][8] and raw2[c] = w2x[c][0]*x0 + w2x[c][1]*x1 is recomputed per voxel (16 B read instead of 2C bytes);
//     its InstanceNorm statistics come from the input's second moments (xbranch_stats_kernel).
template <bool XR>
__device__ __forceinline__ void second_branch(const float (&in8)[8], const float (&wa)[8], const float (&wb)[8], float (&x2)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) x2[j] = XR ? wa[j] * in8[0] + wb[j] * in8[1] : in8[j];
}

template <typename T, int LPV, bool TWO, bool XR = false>
__global__ void __launch_bounds__(EPI_THREADS)
cat_fwd_kernel(const T* __restrict__ raw, const float* __restrict__ mean,
               const float* __restrict__ rstd, const T* __restrict__ raw2,
               const float* __restrict__ mean2, const float* __restrict__ rstd2, int C, float slope,
               T* __restrict__ out, long long V, const float* __restrict__ w2x = nullptr, int xic = 0) {
  const int n = blockIdx.y, P = gridDim.x;
  const int cg = threadIdx.x % LPV, vb = threadIdx.x / LPV;
  constexpr int VPB = EPI_THREADS / LPV;
  const int c0 = cg * 8;
  float mu[8], rs[8], mu2[8], rs2[8], wa[8], wb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mu[j] = mean[n * C + c0 + j]; rs[j] = rstd[n * C + c0 + j];
    mu2[j] = TWO ? mean2[n * C + c0 + j] : 0.f;
    rs2[j] = TWO ? rstd2[n * C + c0 + j] : 0.f;
    wa[j] = XR ? w2x[(c0 + j) * xic] : 0.f;
    wb[j] = (XR && xic > 1) ? w2x[(c0 + j) * xic + 1] : 0.f;
  }
  const long long stride = (long long)P * VPB;
  long long v = (long long)blockIdx.x * VPB + vb;
  Pack8<T> nx, nx2;   // software pipeline: voxel v + stride is loaded before voxel v is computed
  zero8p(nx); zero8p(nx2);
  if (v < V) {
    const long long o = ((long long)n * V + v) * C + c0;
    load8p(raw + o, nx);
    if (TWO) load8p(XR ? raw2 + ((long long)n * V + v) * 8 : raw2 + o, nx2);
  }
  for (; v < V; v += stride) {
    const long long o = ((long long)n * V + v) * C + c0;
    float x[8], in2[8], x2[8], y[8];
    unpack8(nx, x);
    if (TWO) { unpack8(nx2, in2); second_branch<XR>(in2, wa, wb, x2); }
    if (v + stride < V) {
      load8p(raw + o + stride * C, nx);
      if (TWO) load8p(XR ? raw2 + ((long long)n * V + v + stride) * 8 : raw2 + o + stride * C, nx2);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (x[j] - mu[j]) * rs[j];
      y[j] = xh > 0.f ? xh : xh * slope;
    }
    if (TWO) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = x2[j];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (x[j] - mu2[j]) * rs2[j];
        y[j] += xh > 0.f ? xh : xh * slope;
      }
    }
    store8(out + o, y);
  }
}

// The gradient that arrives through the 2x2x2 max-pool consuming this block's output (encoder: ec33 -> pool0, ec63 -> pool1,
// ec93 -> pool2) is added ON THE FLY in both passes instead of being scattered into g_out by a pooling-backward kernel first
// (a read-modify-write of the whole full-resolution gradient): window word of arg-max positions (cat_fwd_pool_kernel) + the
// pooled gradient, 20 bytes per voxel and 8 channels, shared by the eight voxels of a window through the caches.
struct PoolRef {
  const unsigned* argmax;     // [N][Vo][C/8], 3 bits per channel; null = no pool gradient
  const void* g_pool;         // [N][Vo][C]
  unsigned W, H, Wo, Ho;      // extents of THIS block's level, and of the pooled level
  unsigned mW, mH;            // floor(2^32 / W), floor(2^32 / H)
  long long Vo;
};
__device__ __forceinline__ unsigned div_small(unsigned n, unsigned d, unsigned m, unsigned& rem) {
  unsigned q = __umulhi(n, m);            // q <= n / d <= q + 1
  unsigned r = n - q * d;
  if (r >= d) { ++q; r -= d; }
  rem = r;
  return q;
}
// window position (0..7, z-y-x scan order) of voxel v and the index of its window
__device__ __forceinline__ void pool_locate(const PoolRef& pr, unsigned v, unsigned& kpos, unsigned& cv) {
  unsigned x, y;
  const unsigned t = div_small(v, pr.W, pr.mW, x);
  const unsigned z = div_small(t, pr.H, pr.mH, y);
  kpos = ((z & 1u) << 2) | ((y & 1u) << 1) | (x & 1u);
  cv = ((z >> 1) * pr.Ho + (y >> 1)) * pr.Wo + (x >> 1);
}

// APPLY = false: per-(n,c) f64 sums of dxhat, dxhat*xhat for one or two branches (nothing stored)
// APPLY = true : draw = rstd * (dxhat - m1 - xhat * m2) for each branch (dxhat_out may alias g_out)
// XW (pass A of a two-branch block whose second branch is a 1x1x1 conv of the <= 2-channel network input, the x33 / x63 /
//     x93 detail-injection convs): that conv's weight gradient dW2[c][i] = sum_v draw2[v][c] * x[v][i] is NOT accumulated
//     from draw2.  sum_v draw2 = 0 and sum_v draw2 * xhat2 = 0, and xhat2 is itself linear in x, so the sum is what is left of
//     O(1) terms that cancel to ~1e-5 of their size at 128^3; formed from the f32 draw2 (f32 m1 / m2 / mean / rstd, each a
//     systematic offset times the voxel count) it was 4.6e-2 off at 1 x 128^3 in fp32 mode -- and so is the fp32 reference.
//     Instead pass A also sums S_i[c] = sum_v dxhat2[v][c] * x_i[v] (dxhat2 = g * LeakyReLU'; one record per block), and
//     xw_finalize_kernel forms the gradient in f64 from S_i, sum_v dxhat2 and the input's first / second moments.
template <typename T, int LPV, bool TWO, bool APPLY, bool XW = false, bool XR = false>
__global__ void __launch_bounds__(EPI_THREADS)
cat_bwd_kernel(const T* g_out, const T* __restrict__ raw,
               const float* __restrict__ mean, const float* __restrict__ rstd,
               const T* __restrict__ raw2, const float* __restrict__ mean2,
               const float* __restrict__ rstd2, int C, float slope,
               const float* __restrict__ m1p, const float* __restrict__ m2p,
               const float* __restrict__ m1bp, const float* __restrict__ m2bp, T* dxhat_out,
               T* dxhat2_out, double* __restrict__ stat_partial,
               double* __restrict__ stat_partial2, long long V,
               const T* __restrict__ xin = nullptr, double* __restrict__ xw_partial = nullptr,
               const float* __restrict__ w2x = nullptr, int xic = 0, PoolRef pool = PoolRef{}) {
  const int n = blockIdx.y, P = gridDim.x;
  const int cg = threadIdx.x % LPV, vb = threadIdx.x / LPV;
  constexpr int VPB = EPI_THREADS / LPV;
  const int c0 = cg * 8;
  const bool pooled = pool.argmax != nullptr;                       // (uniform)
  const unsigned* pam = pool.argmax + (long long)n * pool.Vo * (C / 8) + cg;
  const T* pgp = reinterpret_cast<const T*>(pool.g_pool) + (long long)n * pool.Vo * C + c0;
  float mu[8], rs[8], mu2[8], rs2[8], a1[8], a2[8], b1[8], b2[8];
  typedef typename std::conditional<sizeof(T) == 2, float, double>::type SumT;   // (bf16: f32 thread sums, see sse_bwd_kernel)
  SumT s[4][8];
  SumT xw[8][2];
  float wa[8], wb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    xw[j][0] = xw[j][1] = 0.0;
    wa[j] = XR ? w2x[(c0 + j) * xic] : 0.f;
    wb[j] = (XR && xic > 1) ? w2x[(c0 + j) * xic + 1] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mu[j] = mean[n * C + c0 + j]; rs[j] = rstd[n * C + c0 + j];
    mu2[j] = TWO ? mean2[n * C + c0 + j] : 0.f;
    rs2[j] = TWO ? rstd2[n * C + c0 + j] : 0.f;
    a1[j] = APPLY ? m1p[n * C + c0 + j] : 0.f;
    a2[j] = APPLY ? m2p[n * C + c0 + j] : 0.f;
    b1[j] = (APPLY && TWO) ? m1bp[n * C + c0 + j] : 0.f;
    b2[j] = (APPLY && TWO) ? m2bp[n * C + c0 + j] : 0.f;
    s[0][j] = s[1][j] = s[2][j] = s[3][j] = 0.0;
  }
  const long long stride = (long long)P * VPB;
  long long v = (long long)blockIdx.x * VPB + vb;
  Pack8<T> ng, nx, nx2, npg;   // software pipeline: voxel v + stride is loaded before voxel v is computed
  unsigned nam = 0, nkpos = 0;
  zero8p(ng); zero8p(nx); zero8p(nx2); zero8p(npg);
  auto fetch_pool = [&](long long vv) __attribute__((always_inline)) {
    unsigned cv;
    pool_locate(pool, (unsigned)vv, nkpos, cv);
    nam = pam[(long long)cv * (C / 8)];
    load8p(pgp + (long long)cv * C, npg);
  };
  if (v < V) {
    const long long o = ((long long)n * V + v) * C + c0;
    load8p(g_out + o, ng);
    load8p(raw + o, nx);
    if (TWO) load8p(XR ? raw2 + ((long long)n * V + v) * 8 : raw2 + o, nx2);
    if (pooled) fetch_pool(v);
  }
  for (; v < V; v += stride) {
    const long long o = ((long long)n * V + v) * C + c0;
    float gy[8], x[8], in2[8], x2[8], d[8];
    unpack8(ng, gy);
    unpack8(nx, x);
    if (TWO) { unpack8(nx2, in2); second_branch<XR>(in2, wa, wb, x2); }
    if (pooled) {           // + the pooled gradient where this voxel was its window's maximum
      float gp[8];
      unpack8(npg, gp);
#pragma unroll
      for (int j = 0; j < 8; ++j) gy[j] += ((nam >> (3 * j)) & 7u) == nkpos ? gp[j] : 0.f;
    }
    if (v + stride < V) {   // a later voxel of this same thread: never written by anyone before it is read
      load8p(g_out + o + stride * C, ng);
      load8p(raw + o + stride * C, nx);
      if (TWO) load8p(XR ? raw2 + ((long long)n * V + v + stride) * 8 : raw2 + o + stride * C, nx2);
      if (pooled) fetch_pool(v + stride);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (x[j] - mu[j]) * rs[j];
      d[j] = gy[j] * (xh > 0.f ? 1.f : slope);
      if (APPLY) d[j] = rs[j] * (d[j] - a1[j] - xh * a2[j]);
      else { s[0][j] += (SumT)d[j]; s[1][j] += (SumT)d[j] * (SumT)xh; }
    }
    if (TWO) {
      float d2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (x2[j] - mu2[j]) * rs2[j];
        d2[j] = gy[j] * (xh > 0.f ? 1.f : slope);
        if (APPLY) d2[j] = rs2[j] * (d2[j] - b1[j] - xh * b2[j]);
        else { s[2][j] += (SumT)d2[j]; s[3][j] += (SumT)d2[j] * (SumT)xh; }
      }
      if (APPLY && !XR) store8(dxhat2_out + o, d2);
      if (XW && !APPLY) {
        float xi[8];
        if (XR) {
#pragma unroll
          for (int j = 0; j < 8; ++j) xi[j] = in2[j];
        } else {
          load8(xin + ((long long)n * V + v) * 8, xi);   // the packed 8-channel input voxel (16 / 32 B, shared by the LPV lanes)
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { xw[j][0] += (SumT)d2[j] * (SumT)xi[0]; xw[j][1] += (SumT)d2[j] * (SumT)xi[1]; }
      }
    }
    if (APPLY) store8(dxhat_out + o, d);  // may alias g_out (same element, read before write)
  }
  if (APPLY) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (XW) {   // block record [C][2] (f64), fixed-order sums
    __shared__ double redx[4][16][16];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const double r = stride_sum_d<LPV>((double)xw[j][i]);
        if (lane < LPV) redx[wave][lane][j * 2 + i] = r;
      }
    __syncthreads();
    double* rec = xw_partial + ((long long)n * P + blockIdx.x) * (C * 2);
    for (int e = threadIdx.x; e < LPV * 16; e += EPI_THREADS) {
      const int gq = e / 16, k = e % 16;
      rec[(gq * 8 + (k >> 1)) * 2 + (k & 1)] = ((redx[0][gq][k] + redx[1][gq][k]) + redx[2][gq][k]) + redx[3][gq][k];
    }
  }
  __shared__ double red[4][16][32];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double r = stride_sum_d<LPV>((double)s[q][j]);
      if (lane < LPV) red[wave][lane][q * 8 + j] = r;
    }
  __syncthreads();
  const long long rec = (long long)n * P + blockIdx.x;
  for (int i = threadIdx.x; i < LPV * 32; i += EPI_THREADS) {
    const int gq = i / 32, k = i % 32, q = k >> 3;
    const double tot = ((red[0][gq][k] + red[1][gq][k]) + red[2][gq][k]) + red[3][gq][k];
    const int c = gq * 8 + (k & 7);
    if (q < 2) stat_partial[(rec * C + c) * 2 + q] = tot;
    else if (TWO) stat_partial2[(rec * C + c) * 2 + (q - 2)] = tot;
  }
}

// ----------------------------------------------------------------------------------
// launchers
// ----------------------------------------------------------------------------------
#define SEUNET_LPV_SWITCH(LPVVAL, ...)                                                  \
  switch (LPVVAL) {                                                                       \
    case 1: { constexpr int LPV = 1; __VA_ARGS__; } break;                                       \
    case 2: { constexpr int LPV = 2; __VA_ARGS__; } break;                                       \
    case 4: { constexpr int LPV = 4; __VA_ARGS__; } break;                                       \
    case 8: { constexpr int LPV = 8; __VA_ARGS__; } break;                                       \
    case 16: { constexpr int LPV = 16; __VA_ARGS__; } break;                                     \
    default: return fail("unsupported channel count %d (need 8,16,32,64 or 128)", (LPVVAL)*8); \
  }

static int check_c(int C) {
  SEUNET_CHECK(C % 8 == 0 && C >= 8 && C <= 128 && (C & (C - 1)) == 0,
               "channel count %d must be a power of two in [8,128]", C);
  return 0;
}

int launch_channel_stats(int dtype, const void* t, int C, double* partial, Dims d, hipStream_t s) {
  if (int e = check_c(C)) return e;
  dim3 grid(epi_partials(d), d.N);
  SEUNET_LPV_SWITCH(C / 8, {
    SEUNET_DTYPE_SWITCH(dtype, channel_stats_kernel<T, LPV><<<grid, EPI_THREADS, 0, s>>>((const T*)t, C, partial, d.vox()));
  });
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_stats_finalize(const double* partial, int slots, int C, int N, long long count, float eps,
                          int mode, float* out_a, float* out_b, hipStream_t s) {
  stats_finalize_kernel<<<N * C, 256, 0, s>>>(partial, slots, C, N, 1.0 / (double)count, eps,
                                                      mode, out_a, out_b);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_sse_fwd(int dtype, const void* raw, const float* mean, const float* rstd, int C,
                   const SseParams& p, void* e_out, const SseHead& head, Dims d, hipStream_t s) {
  if (int e = check_c(C)) return e;
  dim3 grid(epi_partials(d) * 4, d.N);   // nothing is reduced here: enough blocks for full occupancy
  const bool g2 = p.w_se2 != nullptr;
  SEUNET_LPV_SWITCH(C / 8, {
    SEUNET_DTYPE_SWITCH(dtype, {
      if (g2) sse_fwd_kernel<T, LPV, true><<<grid, EPI_THREADS, 0, s>>>((const T*)raw, mean, rstd, C, p, (T*)e_out, head, d.vox());
      else sse_fwd_kernel<T, LPV, false><<<grid, EPI_THREADS, 0, s>>>((const T*)raw, mean, rstd, C, p, (T*)e_out, head, d.vox());
    });
  });
  SEUNET_LAUNCH_CHECK();
  return 0;
}

template <typename T, bool APPLY>
static int sse_bwd_t(const void* raw, const float* mean, const float* rstd, int C, const SseParams& p, const SseBwdIn& g,
                     const SseHead& head, const float* m1, const float* m2, void* out, double* stat_partial,
                     float* pgrad_partial, Dims d, hipStream_t s) {
  dim3 grid(epi_partials(d) * (APPLY ? 4 : 1), d.N);
  const bool g2 = p.w_se2 != nullptr;
  const bool level = g.g_level != nullptr;
#define SEUNET_SSE_BWD(G2V, LV) sse_bwd_kernel<T, LPV, G2V, APPLY, LV><<<grid, EPI_THREADS, 0, s>>>((const T*)raw, mean, rstd, C, p, g, head, m1, m2, (T*)out, stat_partial, pgrad_partial, d.vox())
  SEUNET_LPV_SWITCH(C / 8, {
    if (g2) { if (level) SEUNET_SSE_BWD(true, true); else SEUNET_SSE_BWD(true, false); }
    else { if (level) SEUNET_SSE_BWD(false, true); else SEUNET_SSE_BWD(false, false); }
  });
#undef SEUNET_SSE_BWD
  SEUNET_LAUNCH_CHECK();
  return 0;
}

// m1 == nullptr: pass A (sums + parameter-gradient records); otherwise pass B (writes draw_out)
int launch_sse_bwd(int dtype, const void* raw, const float* mean, const float* rstd, int C,
                   const SseParams& p, const SseBwdIn& g, const SseHead& head, const float* m1, const float* m2,
                   void* draw_out, double* stat_partial, float* pgrad_partial, Dims d, hipStream_t s) {
  if (int e = check_c(C)) return e;
  if (m1 == nullptr) {
    SEUNET_CHECK(stat_partial && pgrad_partial, "gate_epilogue_bwd pass A needs the partial buffers");
    SEUNET_DTYPE_SWITCH(dtype, return (sse_bwd_t<T, false>(raw, mean, rstd, C, p, g, head, nullptr, nullptr, nullptr, stat_partial, pgrad_partial, d, s)));
  }
  SEUNET_CHECK(m2 && draw_out, "gate_epilogue_bwd pass B needs m2 and the output tensor");
  SEUNET_DTYPE_SWITCH(dtype, return (sse_bwd_t<T, true>(raw, mean, rstd, C, p, g, head, m1, m2, draw_out, nullptr, nullptr, d, s)));
  return 1;
}

int launch_gate_bwd_finalize(const double* stat_partial, int slots, int C, int N, long long count, float* m1, float* m2,
                             const float* pgrad_partial, int records, float* dw_se, float* dw_se2, float* dw_side,
                             float* db_side, float* dhead_w, hipStream_t s) {
  gate_bwd_finalize_kernel<<<N * C + cdiv(4 * C + 4, 4), 256, 0, s>>>(stat_partial, slots, C, N, 1.0 / (double)count, m1, m2,
                                                                     pgrad_partial, records, dw_se, dw_se2, dw_side, db_side, dhead_w);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_pgrad_reduce(const float* pgrad_partial, int records, int C, float* dw_se, float* dw_se2,
                        float* dw_side, float* db_side, float* dhead_w, hipStream_t s) {
  pgrad_reduce_kernel<<<cdiv(4 * C + 4, 4), 256, 0, s>>>(pgrad_partial, records, C, dw_se, dw_se2,
                                                        dw_side, db_side, dhead_w);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

// dW2 (PyTorch layout (C, in_channel, 1, 1, 1)) of an x-branch conv, in f64 from sums (cat_bwd_kernel XW).  Per sample, with
// xc_k = x_k - mean(x_k), Cov = the input's 2 x 2 covariance (per-voxel mean), xhat2 = rs * sum_k w_k xc_k, rs^2 = 1 / (w' Cov w
// + eps), dxhat2 = g * LeakyReLU'(xhat2) and A_k = sum_v dxhat2 xc_k = S_k - mean(x_k) * sum_v dxhat2:
//     draw2 = rs * (dxhat2 - mean(dxhat2) - xhat2 * mean(dxhat2 * xhat2))             (InstanceNorm backward)
//     dW2_i = sum_v draw2 * x_i = rs * (A_i - rs^2 * (sum_k w_k A_k) * (sum_k w_k Cov_ki))
// (the mean(dxhat2) term drops out against sum_v xc_i = 0).  The cancellation between A_i and its projection on w happens in
// f64 here; the f32 inputs of the sums (dxhat2 = g or slope * g, x) enter only through products that are exact in f64.
// One block per output channel; fixed summation order (slots within a sample, then samples): bitwise reproducible.
__global__ void __launch_bounds__(256)
xw_finalize_kernel(const double* __restrict__ xw_partial, const double* __restrict__ stat_partial2, int slots, int C, int N,
                   const double* __restrict__ moments, const float* __restrict__ w2, int ic, double eps, float* __restrict__ dw) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x;
  __shared__ double red[4][3];
  const double wa = (double)w2[c * ic], wb = ic > 1 ? (double)w2[c * ic + 1] : 0.0;
  double g0 = 0.0, g1 = 0.0;
  for (int n = 0; n < N; ++n) {
    double s[3] = {0.0, 0.0, 0.0};       // S_0, S_1, sum dxhat2
    for (int r = threadIdx.x; r < slots; r += 256) {
      const long long rec = (long long)n * slots + r;
      s[0] += xw_partial[(rec * C + c) * 2 + 0];
      s[1] += xw_partial[(rec * C + c) * 2 + 1];
      s[2] += stat_partial2[(rec * C + c) * 2 + 0];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double r = s[k];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) r += shfl_xor_settled(r, off);
      if (lane == 0) red[wave][k] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double S0 = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
      const double S1 = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
      const double Sd = ((red[0][2] + red[1][2]) + red[2][2]) + red[3][2];
      const double* m = moments + n * 5;      // mean x0, mean x1, mean x0^2, mean x0 x1, mean x1^2
      const double c00 = m[2] - m[0] * m[0], c01 = m[3] - m[0] * m[1], c11 = m[4] - m[1] * m[1];
      const double A0 = S0 - m[0] * Sd, A1 = S1 - m[1] * Sd;
      double var = wa * wa * c00 + 2.0 * wa * wb * c01 + wb * wb * c11;
      if (var < 0.0) var = 0.0;
      const double rs2 = 1.0 / (var + eps), rs = sqrt(rs2);
      const double proj = wa * A0 + wb * A1;
      g0 += rs * (A0 - rs2 * proj * (wa * c00 + wb * c01));
      g1 += rs * (A1 - rs2 * proj * (wa * c01 + wb * c11));
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    dw[c * ic] = (float)g0;
    if (ic > 1) dw[c * ic + 1] = (float)g1;
  }
}

int launch_cat_xgrad_finalize(const double* xw_partial, const double* stat_partial2, int slots, const double* moments, const float* w2,
                              int C, int in_channel, int N, float eps, float* dw, hipStream_t s) {
  SEUNET_CHECK(in_channel >= 1 && in_channel <= 2, "cat_xgrad_finalize: in_channel %d (1 or 2)", in_channel);
  xw_finalize_kernel<<<C, 256, 0, s>>>(xw_partial, stat_partial2, slots, C, N, moments, w2, in_channel, (double)eps, dw);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

// diagnostic (seunet_net_read_tensor): the x-branch's raw values, recomputed by the same device function as the aggregation
// epilogue uses (second_branch<true>: same expression, same contraction), written as NCDHW f32
template <typename T>
__global__ void __launch_bounds__(256)
xbranch_values_kernel(const T* __restrict__ x_in, const float* __restrict__ w2x, int C, int xic, float* __restrict__ out, long long V,
                      long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % (C / 8));
    const long long nv = i / (C / 8);           // n * V + v
    const long long n = nv / V, v = nv % V;
    float wa[8], wb[8], in2[8], x2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      wa[j] = w2x[(cg * 8 + j) * xic];
      wb[j] = xic > 1 ? w2x[(cg * 8 + j) * xic + 1] : 0.f;
    }
    Pack8<T> px;
    load8p(x_in + nv * 8, px);
    unpack8(px, in2);
    second_branch<true>(in2, wa, wb, x2);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[(n * C + cg * 8 + j) * V + v] = x2[j];
  }
}
int launch_xbranch_values(int dtype, const void* x_in, const float* w2, int C, int in_channel, float* out, Dims d, hipStream_t s) {
  SEUNET_CHECK(in_channel >= 1 && in_channel <= 2 && C % 8 == 0, "xbranch_values: bad argument");
  const long long total = (long long)d.N * d.vox() * (C / 8);
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  SEUNET_DTYPE_SWITCH(dtype, xbranch_values_kernel<T><<<grid, 256, 0, s>>>((const T*)x_in, w2, C, in_channel, out, d.vox(), total));
  SEUNET_LAUNCH_CHECK();
  return 0;
}

// ---- two-branch aggregation block whose second branch is recomputed from the network input (XR) ---------------------
int xbranch_moment_slots(Dims d) { return epi_partials(d); }

int launch_xbranch_moments(int dtype, const void* x_in, double* partial, Dims d, hipStream_t s) {
  dim3 grid(xbranch_moment_slots(d), d.N);
  SEUNET_DTYPE_SWITCH(dtype, input_moments_kernel<T><<<grid, EPI_THREADS, 0, s>>>((const T*)x_in, partial, d.vox()));
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_xbranch_stats(const double* partial, int slots, const float* w2, int C, int in_channel, int N, long long count,
                         float eps, float* mean2, float* rstd2, double* moments_out, hipStream_t s) {
  SEUNET_CHECK(in_channel >= 1 && in_channel <= 2, "xbranch_stats: in_channel %d (1 or 2)", in_channel);
  xbranch_stats_kernel<<<N, 256, 0, s>>>(partial, slots, w2, C, in_channel, 1.0 / (double)count, eps, mean2, rstd2, moments_out);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_cat_fwd_x(int dtype, const void* raw, const float* mean, const float* rstd, const void* x_in, const float* w2,
                     int in_channel, const float* mean2, const float* rstd2, int C, float slope, void* out, Dims d,
                     hipStream_t s) {
  if (int e = check_c(C)) return e;
  SEUNET_CHECK(in_channel >= 1 && in_channel <= 2, "cat_epilogue_fwd_x: in_channel %d (1 or 2)", in_channel);
  dim3 grid(epi_partials(d) * 4, d.N);
  SEUNET_LPV_SWITCH(C / 8, {
    SEUNET_DTYPE_SWITCH(dtype, cat_fwd_kernel<T, LPV, true, true><<<grid, EPI_THREADS, 0, s>>>((const T*)raw, mean, rstd, (const T*)x_in, mean2, rstd2, C, slope, (T*)out, d.vox(), w2, in_channel));
  });
  SEUNET_LAUNCH_CHECK();
  return 0;
}

// The aggregation block's forward with the 2x2x2 max-pool that follows it in the encoder (SE_UNet.py:188-189, 197-198,
// 206-207: ec33 -> pool0, ec63 -> pool1, ec93 -> pool2) written by the same kernel: a thread owns one pooling window x 8
// channels, computes the block output of its eight voxels (same arithmetic as cat_fwd_kernel<.., true, true>), stores them and
// their maximum.  The pooled tensor costs one extra 1/8-size store instead of a second read of the full-resolution output.
// (Rounding is monotonic, so the maximum of the rounded values the separate kernel reads equals the rounded maximum.)
template <typename T, int LPV>
__global__ void __launch_bounds__(EPI_THREADS)
cat_fwd_pool_kernel(const T* __restrict__ raw, const float* __restrict__ mean, const float* __restrict__ rstd,
                    const T* __restrict__ xin, const float* __restrict__ mean2, const float* __restrict__ rstd2, int C, float slope,
                    T* __restrict__ out, T* __restrict__ pooled, int D, int H, int W, const float* __restrict__ w2x, int xic,
                    unsigned* __restrict__ argmax) {
  const int n = blockIdx.y, P = gridDim.x;
  const int cg = threadIdx.x % LPV, vb = threadIdx.x / LPV;
  constexpr int VPB = EPI_THREADS / LPV;
  const int c0 = cg * 8;
  float mu[8], rs[8], mu2[8], rs2[8], wa[8], wb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mu[j] = mean[n * C + c0 + j]; rs[j] = rstd[n * C + c0 + j];
    mu2[j] = mean2[n * C + c0 + j]; rs2[j] = rstd2[n * C + c0 + j];
    wa[j] = w2x[(c0 + j) * xic];
    wb[j] = xic > 1 ? w2x[(c0 + j) * xic + 1] : 0.f;
  }
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const long long V = (long long)D * H * W, Vo = (long long)Do * Ho * Wo;
  const long long stride = (long long)P * VPB;
  for (long long cv = (long long)blockIdx.x * VPB + vb; cv < Vo; cv += stride) {
    const int xo = (int)(cv % Wo);
    const int yo = (int)((cv / Wo) % Ho);
    const int zo = (int)(cv / ((long long)Wo * Ho));
    const long long v0 = ((long long)(2 * zo) * H + 2 * yo) * W + 2 * xo;
    Pack8<T> px[8], pi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {   // all sixteen loads of the window in flight
      const long long v = v0 + ((long long)(k >> 2) * H + ((k >> 1) & 1)) * W + (k & 1);
      load8p(raw + ((long long)n * V + v) * C + c0, px[k]);
      load8p(xin + ((long long)n * V + v) * 8, pi[k]);
    }
    float m[8];
    unsigned am = 0;      // 3 bits per channel: the window position (z, y, x scan order) of the FIRST maximum of the STORED values
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const long long v = v0 + ((long long)(k >> 2) * H + ((k >> 1) & 1)) * W + (k & 1);
      float x[8], in2[8], x2[8], y[8];
      unpack8(px[k], x);
      unpack8(pi[k], in2);
      second_branch<true>(in2, wa, wb, x2);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (x[j] - mu[j]) * rs[j];
        y[j] = xh > 0.f ? xh : xh * slope;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (x2[j] - mu2[j]) * rs2[j];
        y[j] += xh > 0.f ? xh : xh * slope;
      }
      store8(out + ((long long)n * V + v) * C + c0, y);
      // the maximum (and its position) of the values as stored: what a max-pool over the stored tensor sees (two different f32
      // values may round to the same 16-bit value; the reference's first-maximum rule then picks the earlier one)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float yr = round_to<T>(y[j]);
        if (yr > m[j]) { m[j] = yr; am = (am & ~(7u << (3 * j))) | ((unsigned)k << (3 * j)); }
      }
    }
    store8(pooled + ((long long)n * Vo + cv) * C + c0, m);
    if (argmax != nullptr) argmax[((long long)n * Vo + cv) * (C / 8) + cg] = am;
  }
}

int launch_cat_fwd_x_pool(int dtype, const void* raw, const float* mean, const float* rstd, const void* x_in, const float* w2,
                          int in_channel, const float* mean2, const float* rstd2, int C, float slope, void* out, void* pooled,
                          Dims d, hipStream_t s, unsigned* argmax) {
  if (int e = check_c(C)) return e;
  SEUNET_CHECK(in_channel >= 1 && in_channel <= 2, "cat_epilogue_fwd_x_pool: in_channel %d (1 or 2)", in_channel);
  SEUNET_CHECK(d.D % 2 == 0 && d.H % 2 == 0 && d.W % 2 == 0, "cat_epilogue_fwd_x_pool: odd extent");
  dim3 grid(epi_partials(d) * 4, d.N);
  SEUNET_LPV_SWITCH(C / 8, {
    SEUNET_DTYPE_SWITCH(dtype, cat_fwd_pool_kernel<T, LPV><<<grid, EPI_THREADS, 0, s>>>((const T*)raw, mean, rstd, (const T*)x_in, mean2, rstd2, C, slope, (T*)out, (T*)pooled, d.D, d.H, d.W, w2, in_channel, argmax));
  });
  SEUNET_LAUNCH_CHECK();
  return 0;
}

// m1 == nullptr: pass A (f64 sums of both branches and, if xw_partial is given, one record of the x-branch weight-gradient sums per
// block, see cat_bwd_kernel XW / XR); otherwise pass B: writes dx (may alias g_out)
int launch_cat_bwd_x(int dtype, const void* g_out, const void* raw, const float* mean, const float* rstd, const void* x_in,
                     const float* w2, int in_channel, const float* mean2, const float* rstd2, int C, float slope,
                     const float* m1, const float* m2, const float* m1b, const float* m2b, void* dx, double* stat_partial,
                     double* stat_partial2, double* xw_partial, Dims d, hipStream_t s, const unsigned* pool_argmax, const void* pool_g) {
  if (int e = check_c(C)) return e;
  SEUNET_CHECK(in_channel >= 1 && in_channel <= 2, "cat_epilogue_bwd_x: in_channel %d (1 or 2)", in_channel);
  PoolRef pr{};
  if (pool_argmax != nullptr) {
    SEUNET_CHECK(pool_g != nullptr && d.D % 2 == 0 && d.H % 2 == 0 && d.W % 2 == 0 && d.vox() < (1ll << 31),
                 "cat_epilogue_bwd_x: pooled gradient needs even extents below 2^31 voxels");
    pr.argmax = pool_argmax; pr.g_pool = pool_g;
    pr.W = (unsigned)d.W; pr.H = (unsigned)d.H; pr.Wo = (unsigned)d.W / 2; pr.Ho = (unsigned)d.H / 2;
    pr.mW = (unsigned)((1ull << 32) / (unsigned)d.W); pr.mH = (unsigned)((1ull << 32) / (unsigned)d.H);
    pr.Vo = d.vox() / 8;
  }
  const bool apply = m1 != nullptr;
  if (!apply) SEUNET_CHECK(stat_partial && stat_partial2, "cat_epilogue_bwd_x pass A needs the partial buffers");
  else SEUNET_CHECK(m2 && m1b && m2b && dx, "cat_epilogue_bwd_x pass B: missing argument");
  dim3 grid(epi_partials(d) * (apply ? 4 : 1), d.N);
  SEUNET_LPV_SWITCH(C / 8, {
    SEUNET_DTYPE_SWITCH(dtype, {
      if (apply) cat_bwd_kernel<T, LPV, true, true, false, true><<<grid, EPI_THREADS, 0, s>>>((const T*)g_out, (const T*)raw, mean, rstd, (const T*)x_in, mean2, rstd2, C, slope, m1, m2, m1b, m2b, (T*)dx, nullptr, nullptr, nullptr, d.vox(), nullptr, nullptr, w2, in_channel, pr);
      else if (xw_partial) cat_bwd_kernel<T, LPV, true, false, true, true><<<grid, EPI_THREADS, 0, s>>>((const T*)g_out, (const T*)raw, mean, rstd, (const T*)x_in, mean2, rstd2, C, slope, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stat_partial, stat_partial2, d.vox(), nullptr, xw_partial, w2, in_channel, pr);
      else cat_bwd_kernel<T, LPV, true, false, false, true><<<grid, EPI_THREADS, 0, s>>>((const T*)g_out, (const T*)raw, mean, rstd, (const T*)x_in, mean2, rstd2, C, slope, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stat_partial, stat_partial2, d.vox(), nullptr, nullptr, w2, in_channel, pr);
    });
  });
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_cat_fwd(int dtype, const void* raw, const float* mean, const float* rstd, const void* raw2,
                   const float* mean2, const float* rstd2, int C, float slope, void* out, Dims d,
                   hipStream_t s) {
  if (int e = check_c(C)) return e;
  dim3 grid(epi_partials(d) * 4, d.N);
  const bool two = raw2 != nullptr;
  SEUNET_LPV_SWITCH(C / 8, {
    SEUNET_DTYPE_SWITCH(dtype, {
      if (two) cat_fwd_kernel<T, LPV, true><<<grid, EPI_THREADS, 0, s>>>((const T*)raw, mean, rstd, (const T*)raw2, mean2, rstd2, C, slope, (T*)out, d.vox());
      else cat_fwd_kernel<T, LPV, false><<<grid, EPI_THREADS, 0, s>>>((const T*)raw, mean, rstd, nullptr, nullptr, nullptr, C, slope, (T*)out, d.vox());
    });
  });
  SEUNET_LAUNCH_CHECK();
  return 0;
}

template <typename T, bool APPLY>
static int cat_bwd_t(const void* g_out, const void* raw, const float* mean, const float* rstd, const void* raw2,
                     const float* mean2, const float* rstd2, int C, float slope, const float* m1, const float* m2,
                     const float* m1b, const float* m2b, void* dx, void* dx2, double* st, double* st2, Dims d, hipStream_t s) {
  dim3 grid(epi_partials(d) * (APPLY ? 4 : 1), d.N);
  const bool two = raw2 != nullptr;
  SEUNET_LPV_SWITCH(C / 8, {
    if (two) cat_bwd_kernel<T, LPV, true, APPLY><<<grid, EPI_THREADS, 0, s>>>((const T*)g_out, (const T*)raw, mean, rstd, (const T*)raw2, mean2, rstd2, C, slope, m1, m2, m1b, m2b, (T*)dx, (T*)dx2, st, st2, d.vox());
    else cat_bwd_kernel<T, LPV, false, APPLY><<<grid, EPI_THREADS, 0, s>>>((const T*)g_out, (const T*)raw, mean, rstd, nullptr, nullptr, nullptr, C, slope, m1, m2, nullptr, nullptr, (T*)dx, nullptr, st, nullptr, d.vox());
  });
  SEUNET_LAUNCH_CHECK();
  return 0;
}

// m1 == nullptr: pass A (f64 sums into stat_partial[2]); otherwise pass B (writes draw into dx / dx2)
int launch_cat_bwd(int dtype, const void* g_out, const void* raw, const float* mean, const float* rstd,
                   const void* raw2, const float* mean2, const float* rstd2, int C, float slope, const float* m1,
                   const float* m2, const float* m1b, const float* m2b, void* dx, void* dx2, double* stat_partial,
                   double* stat_partial2, Dims d, hipStream_t s) {
  if (int e = check_c(C)) return e;
  if (m1 == nullptr) {
    SEUNET_CHECK(stat_partial && (!raw2 || stat_partial2), "cat_epilogue_bwd pass A needs the partial buffers");
    SEUNET_DTYPE_SWITCH(dtype, return (cat_bwd_t<T, false>(g_out, raw, mean, rstd, raw2, mean2, rstd2, C, slope, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stat_partial, stat_partial2, d, s)));
  }
  SEUNET_CHECK(m2 && dx && (!raw2 || (m1b && m2b && dx2)), "cat_epilogue_bwd pass B: missing argument");
  SEUNET_DTYPE_SWITCH(dtype, return (cat_bwd_t<T, true>(g_out, raw, mean, rstd, raw2, mean2, rstd2, C, slope, m1, m2, m1b, m2b, dx, dx2, nullptr, nullptr, d, s)));
  return 1;
}

}  // namespace seunet
