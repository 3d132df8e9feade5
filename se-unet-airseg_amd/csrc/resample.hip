// Layout conversion, 2x2x2 max-pool, trilinear (align_corners=True) up-sampling and the two
// deep-supervision heads (gfx950).  All HBM-bound; one lane moves 8 channels (16/32 B).
//
//   max-pool      reference SE_UNet.py:131-133 (nn.MaxPool3d 2/2)
//   up-sampling   reference SE_UNet.py:136-138 (nn.Upsample x2 trilinear align_corners=True) and
//                 the per-block side-map up-sampling by 1/2/4/8 (SE_UNet.py:19,34,61,81)
//   heads         reference SE_UNet.py:150-153,232-233: 1x1x1 conv over the DropLayer-scaled stack of
//                 up-sampled side maps.  Both are linear, so the head weight and the DropLayer scale
//                 are applied to each side map at its native resolution (epilogue.hip accumulates a
//                 single-channel "level map" per resolution) and only those maps are interpolated.
#include "seunet_common.h"

namespace seunet {

// ---------------- trilinear index helpers (PyTorch align_corners=True semantics) -------------
__device__ __forceinline__ float ac_scale(int in, int out) {
  return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
}
__device__ __forceinline__ void ac_src(int o, float rs, int in, int& i0, int& i1, float& lam) {
  const float src = rs * (float)o;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  lam = src - (float)i0;
}
// output indices that can touch input index i
__device__ __forceinline__ void ac_range(int i, float rs, int out, int& lo, int& hi) {
  if (rs <= 0.f) { lo = 0; hi = out - 1; return; }
  lo = (int)floorf((float)(i - 1) / rs) - 1;
  hi = (int)ceilf((float)(i + 1) / rs) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}
__device__ __forceinline__ float ac_weight(int o, int i, float rs, int in) {
  int i0, i1; float lam;
  ac_src(o, rs, in, i0, i1, lam);
  return (i0 == i ? 1.f - lam : 0.f) + (i1 == i ? lam : 0.f);
}

// ---------------- layout ---------------------------------------------------------------------
template <typename T>
__global__ void pack_cl_kernel(const float* __restrict__ in, int C, T* __restrict__ out, int Cpad,
                               long long V, long long total) {
  // thread -> (n, group, v) with v fastest: coalesced f32 reads per channel plane
  const int G = Cpad / 8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long v = i % V;
    const int g = (int)((i / V) % G);
    const long long n = i / (V * G);
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = g * 8 + j;
      x[j] = c < C ? in[(n * C + c) * V + v] : 0.f;
    }
    store8(out + (n * V + v) * Cpad + g * 8, x);
  }
}

template <typename T>
__global__ void unpack_cl_kernel(const T* __restrict__ in, int C, float* __restrict__ out, long long V,
                                 long long total) {
  const int G = C / 8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long v = i % V;
    const int g = (int)((i / V) % G);
    const long long n = i / (V * G);
    float x[8];
    load8(in + (n * V + v) * C + g * 8, x);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[(n * C + g * 8 + j) * V + v] = x[j];
  }
}

// ---------------- max-pool 2x2x2 ---------------------------------------------------------------
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ in, int C, T* __restrict__ out, int D, int H,
                                   int W, long long total) {
  const int G = C / 8, Do = D / 2, Ho = H / 2, Wo = W / 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    long long r = i / G;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho); r /= Ho;
    const int zo = (int)(r % Do);
    const long long n = r / Do;
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int z = 2 * zo + (k >> 2), y = 2 * yo + ((k >> 1) & 1), x = 2 * xo + (k & 1);
      float v[8];
      load8(in + ((((n * D + z) * H + y) * W + x) * (long long)C) + g * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = v[j] > m[j] ? v[j] : m[j];
    }
    store8(out + ((((n * Do + zo) * Ho + yo) * Wo + xo) * (long long)C) + g * 8, m);
  }
}

// routes g_out to the FIRST maximum of each window in (z,y,x) scan order (PyTorch CPU semantics)
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ in, const T* __restrict__ g_out, int C,
                                   T* g_in, int accumulate, int D, int H, int W, long long total) {
  const int G = C / 8, Do = D / 2, Ho = H / 2, Wo = W / 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    long long r = i / G;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho); r /= Ho;
    const int zo = (int)(r % Do);
    const long long n = r / Do;
    float m[8], gy[8];
    int am[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; am[j] = 0; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int z = 2 * zo + (k >> 2), y = 2 * yo + ((k >> 1) & 1), x = 2 * xo + (k & 1);
      float v[8];
      load8(in + ((((n * D + z) * H + y) * W + x) * (long long)C) + g * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (v[j] > m[j]) { m[j] = v[j]; am[j] = k; }
    }
    load8(g_out + ((((n * Do + zo) * Ho + yo) * Wo + xo) * (long long)C) + g * 8, gy);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int z = 2 * zo + (k >> 2), y = 2 * yo + ((k >> 1) & 1), x = 2 * xo + (k & 1);
      T* p = g_in + ((((n * D + z) * H + y) * W + x) * (long long)C) + g * 8;
      float v[8];
      if (accumulate) load8(p, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (accumulate ? v[j] : 0.f) + (am[j] == k ? gy[j] : 0.f);
      store8(p, v);
    }
  }
}

// The same from the arg-max words the fused forward wrote (epilogue.hip cat_fwd_pool_kernel: 3 bits per channel, position of the
// first maximum of the stored values): the backward pass reads 4 bytes per window and 8 channels instead of the eight input
// voxels and the pooled output (pool0 at 4 x 128^3 x 32 channels: 0.60 GB less traffic of 1.75 GB).
template <typename T>
__global__ void maxpool_bwd_idx_kernel(const unsigned* __restrict__ argmax, const T* __restrict__ g_out, int C,
                                       T* g_in, int accumulate, int D, int H, int W, long long total) {
  const int G = C / 8, Do = D / 2, Ho = H / 2, Wo = W / 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    long long r = i / G;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho); r /= Ho;
    const int zo = (int)(r % Do);
    const long long n = r / Do;
    const unsigned am = argmax[i];
    float gy[8];
    load8(g_out + ((((n * Do + zo) * Ho + yo) * Wo + xo) * (long long)C) + g * 8, gy);
    Pack8<T> old[8];
    if (accumulate) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {     // the eight old values in flight together
        const int z = 2 * zo + (k >> 2), y = 2 * yo + ((k >> 1) & 1), x = 2 * xo + (k & 1);
        load8p(g_in + ((((n * D + z) * H + y) * W + x) * (long long)C) + g * 8, old[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int z = 2 * zo + (k >> 2), y = 2 * yo + ((k >> 1) & 1), x = 2 * xo + (k & 1);
      T* p = g_in + ((((n * D + z) * H + y) * W + x) * (long long)C) + g * 8;
      float v[8];
      if (accumulate) unpack8(old[k], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (accumulate ? v[j] : 0.f) + (((am >> (3 * j)) & 7u) == (unsigned)k ? gy[j] : 0.f);
      store8(p, v);
    }
  }
}

// ---------------- x2 trilinear up-sampling of feature maps ------------------------------------
template <typename T>
__global__ void upsample2_fwd_kernel(const T* __restrict__ in, int C, T* __restrict__ out, int D, int H,
                                     int W, long long total) {
  const int G = C / 8, Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const float rz = ac_scale(D, Do), ry = ac_scale(H, Ho), rx = ac_scale(W, Wo);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    long long r = i / G;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho); r /= Ho;
    const int zo = (int)(r % Do);
    const long long n = r / Do;
    int z0, z1, y0, y1, x0, x1; float lz, ly, lx;
    ac_src(zo, rz, D, z0, z1, lz);
    ac_src(yo, ry, H, y0, y1, ly);
    ac_src(xo, rx, W, x0, x1, lx);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int z = (k & 4) ? z1 : z0, y = (k & 2) ? y1 : y0, x = (k & 1) ? x1 : x0;
      const float w = ((k & 4) ? lz : 1.f - lz) * ((k & 2) ? ly : 1.f - ly) * ((k & 1) ? lx : 1.f - lx);
      float v[8];
      load8(in + ((((n * D + z) * H + y) * W + x) * (long long)C) + g * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += w * v[j];
    }
    store8(out + ((((n * Do + zo) * Ho + yo) * Wo + xo) * (long long)C) + g * 8, acc);
  }
}

// Tiled separable form (the one the network uses).  The kernel above issues 8 16-byte loads per stored 16 bytes and is bound
// by the texture path (0.36 ms for the 32-channel 64^3 -> 128^3 map, 1.7 TB/s).  Here a block produces the 2 x 2 output
// rows (zo, yo) in {2zb, 2zb+1} x {2yb, 2yb+1} over 128 fine x: phase A blends the <= 3 x 3 coarse (z, y) rows those four
// output rows draw on -- each coarse 16-byte piece is loaded ONCE per block -- into four z/y-interpolated coarse rows in LDS
// (f32); phase B interpolates along x from LDS and stores coalesced: ~1.2 loads per store instead of 8.
constexpr int UF_XF = 128, UF_XC = 68;   // fine x per block; coarse x a block can touch (128 * 63/127 + 2 < 68)
template <typename T>
__global__ void __launch_bounds__(256)
upsample2_fwd_tiled_kernel(const T* __restrict__ in, int C, T* __restrict__ out, int D, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float uf[];   // [4 rows][UF_XC][C]
  const int G = C / 8, Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const float rz = ac_scale(D, Do), ry = ac_scale(H, Ho), rx = ac_scale(W, Wo);
  const int xf0 = blockIdx.x * UF_XF;
  const int nxf = (Wo - xf0 < UF_XF) ? Wo - xf0 : UF_XF;
  const int yb = blockIdx.y, zb = blockIdx.z % D;
  const long long n = blockIdx.z / D;
  // coarse rows and weights of the two zo / yo of this block
  int zs[2][2], ys[2][2]; float lzs[2], lys[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    ac_src(2 * zb + k, rz, D, zs[k][0], zs[k][1], lzs[k]);
    ac_src(2 * yb + k, ry, H, ys[k][0], ys[k][1], lys[k]);
  }
  const int zc0 = zs[0][0], nzc = zs[1][1] - zc0 + 1;         // distinct coarse z: zc0 .. zc0 + nzc - 1 (<= 3)
  const int yc0 = ys[0][0], nyc = ys[1][1] - yc0 + 1;
  int xc0, xc1, t0, t1; float tl;
  ac_src(xf0, rx, W, xc0, t0, tl);
  ac_src(xf0 + nxf - 1, rx, W, t1, xc1, tl);
  const int nxc = xc1 - xc0 + 1;                               // <= UF_XC
  // phase A
  for (int item = threadIdx.x; item < nxc * G; item += 256) {
    const int g = item % G, xi = item / G;
    float acc[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[r][j] = 0.f;
    for (int zi = 0; zi < nzc; ++zi) {
      const int z = zc0 + zi;
      float wz[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) wz[k] = (zs[k][0] == z ? 1.f - lzs[k] : 0.f) + (zs[k][1] == z ? lzs[k] : 0.f);
      for (int yi = 0; yi < nyc; ++yi) {
        const int y = yc0 + yi;
        float v[8];
        load8(in + ((((n * D + z) * H + y) * W + (xc0 + xi)) * (long long)C) + g * 8, v);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float wy = (ys[k][0] == y ? 1.f - lys[k] : 0.f) + (ys[k][1] == y ? lys[k] : 0.f);
#pragma unroll
          for (int kz = 0; kz < 2; ++kz) {
            const float w = wz[kz] * wy;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[kz * 2 + k][j] += w * v[j];
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* q = uf + ((r * UF_XC + xi) * C) + g * 8;
      reinterpret_cast<float4*>(q)[0] = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
      reinterpret_cast<float4*>(q)[1] = make_float4(acc[r][4], acc[r][5], acc[r][6], acc[r][7]);
    }
  }
  __syncthreads();
  // phase B
  for (int item = threadIdx.x; item < 4 * nxf * G; item += 256) {
    const int g = item % G;
    int r2 = item / G;
    const int xo = xf0 + r2 % nxf, r = r2 / nxf;                // r = kz * 2 + ky
    int x0, x1; float lx;
    ac_src(xo, rx, W, x0, x1, lx);
    const float* q0 = uf + ((r * UF_XC + (x0 - xc0)) * C) + g * 8;
    const float* q1 = uf + ((r * UF_XC + (x1 - xc0)) * C) + g * 8;
    const float4 a0 = reinterpret_cast<const float4*>(q0)[0], b0 = reinterpret_cast<const float4*>(q0)[1];
    const float4 a1 = reinterpret_cast<const float4*>(q1)[0], b1 = reinterpret_cast<const float4*>(q1)[1];
    const float w0 = 1.f - lx;
    float o[8] = {w0 * a0.x + lx * a1.x, w0 * a0.y + lx * a1.y, w0 * a0.z + lx * a1.z, w0 * a0.w + lx * a1.w,
                  w0 * b0.x + lx * b1.x, w0 * b0.y + lx * b1.y, w0 * b0.z + lx * b1.z, w0 * b0.w + lx * b1.w};
    const int zo = 2 * zb + (r >> 1), yo = 2 * yb + (r & 1);
    store8(out + ((((n * Do + zo) * Ho + yo) * Wo + xo) * (long long)C) + g * 8, o);
  }
}

// z-marching form of the forward (the one the network uses for C = 32 / 64 / 128).  A block owns two fine rows x XF fine
// columns (XF * C = 4096) and a run of ZSF fine planes.  Each thread keeps its four (row, column, 8 channels) items'
// values at two consecutive COARSE planes in registers: walking the coarse planes once, it blends the new plane's four
// y/x corners from a small LDS image of the coarse rows (double-buffered, one barrier per coarse plane) and emits the
// fine planes between the two coarse planes as (1 - lz) * previous + lz * current.  Coarse rows are fetched ~2.3 times
// in all instead of ~9, the y/x blend is done once per coarse plane instead of once per fine plane, and every thread has
// the same amount of work.
constexpr int UFM_XC = 68, UFM_ZSF = 16;
template <typename T>
__global__ void __launch_bounds__(256)
upsample2_fwd_march_kernel(const T* __restrict__ in, int C, T* __restrict__ out, int D, int H, int W, int XF) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ufm_raw[];
  T* tile = reinterpret_cast<T*>(ufm_raw);                     // [2 buffers][3 coarse rows][UFM_XC][C]
  const int G = C / 8, Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const float rz = ac_scale(D, Do), ry = ac_scale(H, Ho), rx = ac_scale(W, Wo);
  const TileIdx3 tl = xcd_contiguous_tile3();                  // (round 4: neighbouring tiles on one XCD, the halo in one L2)
  const int xf0 = tl.x * XF;
  const int nxf = (Wo - xf0 < XF) ? Wo - xf0 : XF;
  const int yb = tl.y;
  const int nseg = (Do + UFM_ZSF - 1) / UFM_ZSF;
  const int seg = tl.z % nseg;
  const long long n = tl.z / nseg;
  const int zf_a = seg * UFM_ZSF, zf_b = zf_a + UFM_ZSF < Do ? zf_a + UFM_ZSF : Do;
  int i0, i1; float lam;
  // coarse rows / columns under the block
  int ya0, ya1, yb0, yb1; float lya, lyb;
  ac_src(2 * yb, ry, H, ya0, ya1, lya);
  ac_src(2 * yb + 1, ry, H, yb0, yb1, lyb);
  const int yc0 = ya0, nyc = yb1 - yc0 + 1;                    // <= 3
  int xc0, xc1, t0;
  ac_src(xf0, rx, W, xc0, t0, lam);
  ac_src(xf0 + nxf - 1, rx, W, t0, xc1, lam);
  const int nxc = xc1 - xc0 + 1;                               // <= UFM_XC
  const int buf_elems = 3 * UFM_XC * C;
  // this thread's four items: fine row, fine column, channel group -> four corner offsets in a tile buffer and weights
  constexpr int NI = 4;
  int o00[NI], o01[NI], o10[NI], o11[NI];
  float w00[NI], w01[NI], w10[NI], w11[NI];
  long long oofs[NI];
  bool live[NI];
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int item = threadIdx.x + 256 * it;
    const int g = item % G;
    const int r = item / G;
    const int fx = r % XF, fy = r / XF;                        // fy in {0, 1}
    live[it] = fx < nxf && fy < 2;
    const int xo = xf0 + (fx < nxf ? fx : 0);
    int x0, x1; float lx;
    ac_src(xo, rx, W, x0, x1, lx);
    const int y0 = fy ? yb0 : ya0, y1 = fy ? yb1 : ya1;
    const float ly = fy ? lyb : lya;
    o00[it] = ((y0 - yc0) * UFM_XC + (x0 - xc0)) * C + g * 8;
    o01[it] = ((y0 - yc0) * UFM_XC + (x1 - xc0)) * C + g * 8;
    o10[it] = ((y1 - yc0) * UFM_XC + (x0 - xc0)) * C + g * 8;
    o11[it] = ((y1 - yc0) * UFM_XC + (x1 - xc0)) * C + g * 8;
    w00[it] = (1.f - ly) * (1.f - lx); w01[it] = (1.f - ly) * lx; w10[it] = ly * (1.f - lx); w11[it] = ly * lx;
    oofs[it] = (((n * Do) * Ho + (2 * yb + (fy & 1))) * (long long)Wo + xo) * C + g * 8;   // + zo * Ho * Wo * C
  }
  const long long oplane = (long long)Ho * Wo * C;
  // coarse planes the segment needs
  ac_src(zf_a, rz, D, i0, i1, lam);
  const int zc_first = i0;
  ac_src(zf_b - 1, rz, D, i0, i1, lam);
  const int zc_last = i1;
  auto stage = [&](int zc, int buf) {          // coarse rows yc0.. of plane zc -> tile[buf]
    T* tb = tile + buf * buf_elems;
    for (int item = threadIdx.x; item < nyc * nxc * G; item += 256) {
      const int g = item % G;
      const int r = item / G;
      const int xi = r % nxc, yi = r / nxc;
      Pack8<T> v;
      load8p(in + ((((n * D + zc) * H + (yc0 + yi)) * (long long)W + (xc0 + xi)) * C) + g * 8, v);
      *reinterpret_cast<Pack8<T>*>(tb + (yi * UFM_XC + xi) * C + g * 8) = v;
    }
  };
  float pprev[NI][8], pcur[NI][8];
#pragma unroll
  for (int it = 0; it < NI; ++it)
#pragma unroll
    for (int j = 0; j < 8; ++j) pprev[it][j] = pcur[it][j] = 0.f;
  int zo = zf_a;
  stage(zc_first, 0);
  __syncthreads();
  int buf = 0;
  for (int zc = zc_first; zc <= zc_last; ++zc) {
    if (zc < zc_last) stage(zc + 1, buf ^ 1);                 // next coarse plane into the other buffer (read after the barrier)
    const T* tb = tile + buf * buf_elems;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      float a[8], b[8], c[8], d[8];
      Pack8<T> k;
      k = *reinterpret_cast<const Pack8<T>*>(tb + o00[it]); unpack8(k, a);
      k = *reinterpret_cast<const Pack8<T>*>(tb + o01[it]); unpack8(k, b);
      k = *reinterpret_cast<const Pack8<T>*>(tb + o10[it]); unpack8(k, c);
      k = *reinterpret_cast<const Pack8<T>*>(tb + o11[it]); unpack8(k, d);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pprev[it][j] = pcur[it][j];
        pcur[it][j] = w00[it] * a[j] + w01[it] * b[j] + w10[it] * c[j] + w11[it] * d[j];
      }
    }
    // fine planes between coarse planes zc - 1 and zc (or clamped onto zc)
    while (zo < zf_b) {
      ac_src(zo, rz, D, i0, i1, lam);
      const bool between = i1 == zc && i0 == zc - 1, on = i0 == zc && i1 == zc;
      if (!between && !on) break;
      const float wp = between ? 1.f - lam : 0.f, wc = between ? lam : 1.f;
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        if (!live[it]) continue;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = wp * pprev[it][j] + wc * pcur[it][j];
        store8(out + oofs[it] + zo * oplane, o);
      }
      ++zo;
    }
    __syncthreads();                                          // the staged plane is complete; this plane's buffer is free
    buf ^= 1;
  }
}

// gather form of the transposed interpolation: one lane per (input voxel, 8 channels)
template <typename T>
__global__ void upsample2_bwd_kernel(const T* __restrict__ g_out, int C, T* g_in, int accumulate, int D,
                                     int H, int W, long long total) {
  const int G = C / 8, Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const float rz = ac_scale(D, Do), ry = ac_scale(H, Ho), rx = ac_scale(W, Wo);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    long long r = i / G;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H); r /= H;
    const int z = (int)(r % D);
    const long long n = r / D;
    int zl, zh, yl, yh, xl, xh;
    ac_range(z, rz, Do, zl, zh);
    ac_range(y, ry, Ho, yl, yh);
    ac_range(x, rx, Wo, xl, xh);
    float acc[8];
    T* p = g_in + ((((n * D + z) * H + y) * W + x) * (long long)C) + g * 8;
    if (accumulate) load8(p, acc);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    }
    for (int zo = zl; zo <= zh; ++zo) {
      const float wz = ac_weight(zo, z, rz, D);
      if (wz == 0.f) continue;
      for (int yo = yl; yo <= yh; ++yo) {
        const float wy = ac_weight(yo, y, ry, H);
        if (wy == 0.f) continue;
        for (int xo = xl; xo <= xh; ++xo) {
          const float wx = ac_weight(xo, x, rx, W);
          if (wx == 0.f) continue;
          float v[8];
          load8(g_out + ((((n * Do + zo) * Ho + yo) * Wo + xo) * (long long)C) + g * 8, v);
          const float w = wz * wy * wx;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += w * v[j];
        }
      }
    }
    store8(p, acc);
  }
}

// Tiled, separable form of the same transposed interpolation (the form the network uses).  The gather above spends its
// time evaluating ~7^3 candidate weights and issuing up to 64 16-byte loads per lane.  The trilinear weight factorises,
// so a block first reduces z and y for the fine x-columns its coarse tile touches (<= 16 loads per item, coalesced
// along x and channels) into LDS, then reduces x from LDS (<= 4 reads per item): ~4x fewer L1 transactions and ~6x
// fewer weight evaluations, no temporary tensor in HBM.
constexpr int UB_TX = 16, UB_XF = 44;   // coarse x per block; fine x columns a tile can touch (2.0625 * 17 + 4 < 44)
template <typename T, int TY>
__global__ void __launch_bounds__(256)
upsample2_bwd_tiled_kernel(const T* __restrict__ g_out, int C, T* g_in, int accumulate, int D, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float us[];   // [TY][UB_XF][C]
  const int G = C / 8, Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const float rz = ac_scale(D, Do), ry = ac_scale(H, Ho), rx = ac_scale(W, Wo);
  const int x0 = blockIdx.x * UB_TX, y0 = blockIdx.y * TY;
  const int z = blockIdx.z % D;
  const long long n = blockIdx.z / D;
  const int x1 = (x0 + UB_TX < W ? x0 + UB_TX : W) - 1;
  int xf0, xf1, t0, t1;
  ac_range(x0, rx, Wo, xf0, t0);
  ac_range(x1, rx, Wo, t1, xf1);
  const int nxf = xf1 - xf0 + 1;          // <= UB_XF (checked by the launcher's choice of UB_TX / UB_XF)
  int zl, zh;
  ac_range(z, rz, Do, zl, zh);
  // the z weights of the block's plane and the y weights of its TY rows, evaluated once (not per item and candidate)
  __shared__ float wzs[12], wys[TY][12];
  __shared__ int yls[TY], yhs[TY];
  if (threadIdx.x < 12) wzs[threadIdx.x] = (zl + (int)threadIdx.x <= zh) ? ac_weight(zl + (int)threadIdx.x, z, rz, D) : 0.f;
  if (threadIdx.x >= 64 && threadIdx.x < 64 + TY * 12) {
    const int yi = (threadIdx.x - 64) / 12, k = (threadIdx.x - 64) % 12, y = y0 + yi;
    int yl = 0, yh = -1;
    if (y < H) ac_range(y, ry, Ho, yl, yh);
    wys[yi][k] = (yl + k <= yh) ? ac_weight(yl + k, y, ry, H) : 0.f;
    if (k == 0) { yls[yi] = yl; yhs[yi] = yh < yl + 11 ? yh : yl + 11; }
  }
  __syncthreads();
  const int zh_c = zh < zl + 11 ? zh : zl + 11;   // (the candidate range is at most 9 wide)
  // phase 1: reduce z and y
  for (int item = threadIdx.x; item < TY * nxf * G; item += 256) {
    const int g = item % G;
    int r = item / G;
    const int xi = r % nxf, yi = r / nxf;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const int yl = yls[yi], yh = yhs[yi];
    for (int zo = zl; zo <= zh_c; ++zo) {
      const float wz = wzs[zo - zl];
      if (wz == 0.f) continue;
      for (int yo = yl; yo <= yh; ++yo) {
        const float wy = wys[yi][yo - yl];
        if (wy == 0.f) continue;
        float v[8];
        load8(g_out + ((((n * Do + zo) * Ho + yo) * Wo + (xf0 + xi)) * (long long)C) + g * 8, v);
        const float w = wz * wy;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += w * v[j];
      }
    }
    float* q = us + (yi * UB_XF + xi) * C + g * 8;
    reinterpret_cast<float4*>(q)[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    reinterpret_cast<float4*>(q)[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
  __syncthreads();
  // phase 2: reduce x
  for (int item = threadIdx.x; item < TY * UB_TX * G; item += 256) {
    const int g = item % G;
    int r = item / G;
    const int xi = r % UB_TX, yi = r / UB_TX;
    const int x = x0 + xi, y = y0 + yi;
    if (x >= W || y >= H) continue;
    int xl, xh;
    ac_range(x, rx, Wo, xl, xh);
    T* p = g_in + ((((n * D + z) * H + y) * W + x) * (long long)C) + g * 8;
    float acc[8];
    if (accumulate) load8(p, acc);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    }
    for (int xo = xl; xo <= xh; ++xo) {
      const float wx = ac_weight(xo, x, rx, W);
      if (wx == 0.f) continue;
      const float* q = us + (yi * UB_XF + (xo - xf0)) * C + g * 8;
      const float4 a = reinterpret_cast<const float4*>(q)[0], b = reinterpret_cast<const float4*>(q)[1];
      acc[0] += wx * a.x; acc[1] += wx * a.y; acc[2] += wx * a.z; acc[3] += wx * a.w;
      acc[4] += wx * b.x; acc[5] += wx * b.y; acc[6] += wx * b.z; acc[7] += wx * b.w;
    }
    store8(p, acc);
  }
}

// z-marching form of the tiled kernel (the one the network uses when every coarse extent is >= 4).  The tiled kernel
// above gives each coarse plane its own block, so every fine plane is fetched by the ~2 coarse planes it feeds and the
// interpolation ranges are re-derived per block.  Here a block owns TY x 16 coarse (y, x) positions and a run of ZS
// coarse planes and walks the fine planes under it ONCE, in order: a fine plane is reduced over y (global loads -> LDS)
// and x (LDS -> registers) to the block's coarse footprint, and that partial is added to the two coarse planes it feeds
// (weights 1-lam / lam of ac_src), held in two rotating register accumulators; a coarse plane is stored when the walk
// has passed it.  Every range and weight is evaluated once per block.
constexpr int UM_TX = 16, UM_XF = 40, UM_K = 6;   // coarse x per block; fine x columns under them; candidates per coarse index
__device__ __forceinline__ void ac_exact(int i, float rs, int in, int out, int& lo, int& n, float (&w)[UM_K]) {
  // the contiguous fine indices o with ac_src(o).i0 == i or .i1 == i, and their weights (<= 5 for in >= 4)
  int l, h;
  ac_range(i, rs, out, l, h);
  int i0, i1; float lam;
  for (;; ++l) { ac_src(l, rs, in, i0, i1, lam); if (i1 >= i || l >= h) break; }
  for (;; --h) { ac_src(h, rs, in, i0, i1, lam); if (i0 <= i || h <= l) break; }
  lo = l;
  n = h - l + 1 < UM_K ? h - l + 1 : UM_K;
#pragma unroll
  for (int k = 0; k < UM_K; ++k) w[k] = k < n ? ac_weight(l + k, i, rs, in) : 0.f;
}
template <typename T, int TY>
__global__ void __launch_bounds__(256, 2)
upsample2_bwd_march_kernel(const T* __restrict__ g_out, int C, T* g_in, int accumulate, int D, int H, int W, int ZS) {
  extern __shared__ __attribute__((aligned(16))) float us[];   // [2][TY][UM_XF][C]: one buffer per fine plane, alternating
  __shared__ float wys[TY][UM_K];
  __shared__ int yls[TY], nys[TY];
  const int G = C / 8, Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const float rz = ac_scale(D, Do), ry = ac_scale(H, Ho), rx = ac_scale(W, Wo);
  const TileIdx3 tl = xcd_contiguous_tile3();                  // (round 4: neighbouring tiles on one XCD, the halo in one L2)
  const int x0 = tl.x * UM_TX, y0 = tl.y * TY;
  const int nseg = (D + ZS - 1) / ZS;
  const int seg = tl.z % nseg;
  const long long n = tl.z / nseg;
  const int zc0 = seg * ZS, zc1 = zc0 + ZS < D ? zc0 + ZS : D;
  const int x1 = (x0 + UM_TX < W ? x0 + UM_TX : W) - 1;
  int i0, i1; float lam;
  // fine x columns / fine planes that feed this block (exact: first index whose upper target reaches the block, last whose
  // lower target is still inside)
  int xf0, xf1, t;
  ac_range(x0, rx, Wo, xf0, t);
  for (;; ++xf0) { ac_src(xf0, rx, W, i0, i1, lam); if (i1 >= x0 || xf0 >= Wo - 1) break; }
  ac_range(x1, rx, Wo, t, xf1);
  for (;; --xf1) { ac_src(xf1, rx, W, i0, i1, lam); if (i0 <= x1 || xf1 <= xf0) break; }
  const int nxf = xf1 - xf0 + 1 < UM_XF ? xf1 - xf0 + 1 : UM_XF;
  int zf0, zf1;
  ac_range(zc0, rz, Do, zf0, t);
  for (;; ++zf0) { ac_src(zf0, rz, D, i0, i1, lam); if (i1 >= zc0 || zf0 >= Do - 1) break; }
  ac_range(zc1 - 1, rz, Do, t, zf1);
  for (;; --zf1) { ac_src(zf1, rz, D, i0, i1, lam); if (i0 <= zc1 - 1 || zf1 <= zf0) break; }
  if (threadIdx.x < TY) {
    const int y = y0 + (int)threadIdx.x;
    float w[UM_K];
    int lo = 0, cnt = 0;
    if (y < H) ac_exact(y, ry, H, Ho, lo, cnt, w);
    yls[threadIdx.x] = lo; nys[threadIdx.x] = cnt;
#pragma unroll
    for (int k = 0; k < UM_K; ++k) wys[threadIdx.x][k] = (y < H && k < cnt) ? w[k] : 0.f;
  }
  // this thread's two phase-2 items (coarse y, coarse x, 8 channels): x range and weights, output pointer
  constexpr int NI = 1;     // TY * 16 * G = 256 items
  float wx[NI][UM_K];
  int xrel[NI], qoff[NI];
  bool live[NI];
  long long pofs[NI];
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int item = threadIdx.x + 256 * it;
    const int g = item % G;
    const int r = item / G;
    const int xi = r % UM_TX, yi = r / UM_TX;
    const int x = x0 + xi, y = y0 + yi;
    live[it] = yi < TY && x < W && y < H;
    int lo = xf0, cnt = 0;
    if (live[it]) ac_exact(x, rx, W, Wo, lo, cnt, wx[it]);
    else {
#pragma unroll
      for (int k = 0; k < UM_K; ++k) wx[it][k] = 0.f;
    }
    // columns beyond the staged range carry zero weight; clamp the window into the stage so every read stays inside it
    xrel[it] = lo - xf0;
    if (xrel[it] < 0) xrel[it] = 0;
    if (xrel[it] > UM_XF - UM_K) xrel[it] = UM_XF - UM_K;
    if (live[it] && xrel[it] != lo - xf0) {   // re-derive the weights for the clamped window (border tiles only)
#pragma unroll
      for (int k = 0; k < UM_K; ++k) wx[it][k] = ac_weight(xf0 + xrel[it] + k, x, rx, W);
    }
    qoff[it] = ((yi < TY ? yi : 0) * UM_XF + xrel[it]) * C + g * 8;
    pofs[it] = (((n * D) * H + y) * (long long)W + x) * C + g * 8;   // + z * H * W * C
  }
  float accA[NI][8], accB[NI][8];
#pragma unroll
  for (int it = 0; it < NI; ++it)
#pragma unroll
    for (int j = 0; j < 8; ++j) { accA[it][j] = 0.f; accB[it][j] = 0.f; }
  ac_src(zf0, rz, D, i0, i1, lam);
  int cur = i0;                                   // accA <-> coarse plane cur, accB <-> cur + 1
  const long long plane = (long long)H * W * C;
  auto flush = [&](int z, float (&acc)[NI][8]) {
    if (z < zc0 || z >= zc1) return;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      if (!live[it]) continue;
      T* p = g_in + pofs[it] + z * plane;
      float o[8];
      if (accumulate) {
        load8(p, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += acc[it][j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = acc[it][j];
      }
      store8(p, o);
    }
  };
  __syncthreads();
  // phase-1 items of this thread (coarse row, fine column, 8 channels), fixed for the whole walk: TY * G = 16, so
  // TY * nxf * G <= 640 -> 3 per thread
  constexpr int N1 = 3, KY = 5;
  int gofs[N1];           // element offset of the item's first fine row inside a fine plane (< 2^31: checked by the launcher); < 0: no item
  int uofs[N1], ycnt[N1], yrow[N1];
#pragma unroll
  for (int it = 0; it < N1; ++it) {
    const int item = threadIdx.x + 256 * it;
    const int g = item % G;
    const int r = item / G;
    const int xi = r % nxf, yi = r / nxf;
    const bool on = item < TY * nxf * G;
    const int yy = on ? yi : 0;
    gofs[it] = on ? (yls[yy] * Wo + (xf0 + xi)) * C + g * 8 : -1;
    uofs[it] = (yy * UM_XF + xi) * C + g * 8;
    ycnt[it] = on ? (nys[yy] < KY ? nys[yy] : KY) : 0;
    yrow[it] = yy;
  }
  Pack8<T> raw[N1][KY];
  auto issue = [&](int zo) {
    const T* gz = g_out + ((n * Do + zo) * (long long)Ho) * Wo * C;
#pragma unroll
    for (int it = 0; it < N1; ++it)
#pragma unroll
      for (int k = 0; k < KY; ++k)
        if (k < ycnt[it])   // uniform base + 32-bit byte offset: one address register per load, not a 64-bit pair
          load8p(reinterpret_cast<const T*>(reinterpret_cast<const char*>(gz) + (unsigned)((gofs[it] + k * Wo * C) * (int)sizeof(T))), raw[it][k]);
  };
  issue(zf0);
  int buf = 0;
  for (int zo = zf0; zo <= zf1; ++zo) {
    ac_src(zo, rz, D, i0, i1, lam);
    if (i0 > cur) {                               // (the source index advances by at most one per fine plane: scale < 1)
      flush(cur, accA);
#pragma unroll
      for (int it = 0; it < NI; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) { accA[it][j] = accB[it][j]; accB[it][j] = 0.f; }
      cur = i0;
    }
    const float wA = (1.f - lam) + (i1 == i0 ? lam : 0.f), wB = i1 != i0 ? lam : 0.f;
    // phase 1: the fetched rows of fine plane zo reduced over y into this plane's LDS buffer
    float* ub = us + buf * (TY * UM_XF * C);
#pragma unroll
    for (int it = 0; it < N1; ++it) {
      if (gofs[it] < 0) continue;
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
      for (int k = 0; k < KY; ++k) {
        if (k < ycnt[it]) {
          float v[8];
          unpack8(raw[it][k], v);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += wys[yrow[it]][k] * v[j];
        }
      }
      float* q = ub + uofs[it];
      reinterpret_cast<float4*>(q)[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      reinterpret_cast<float4*>(q)[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
    __syncthreads();     // the one barrier per fine plane: the buffers alternate, so plane zo + 2 overwrites this one only
                         // after every thread has passed the barrier of plane zo + 1, i.e. finished reading it
    if (zo < zf1) issue(zo + 1);   // in flight during phase 2
    // phase 2: reduce x from LDS, add to the two coarse planes this fine plane feeds
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      float pz[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pz[j] = 0.f;
#pragma unroll
      for (int k = 0; k < UM_K; ++k) {
        const float w = wx[it][k];
        if (xrel[it] + k < nxf) {
          const float* q = ub + qoff[it] + k * C;
          const float4 a = reinterpret_cast<const float4*>(q)[0], b = reinterpret_cast<const float4*>(q)[1];
          pz[0] += w * a.x; pz[1] += w * a.y; pz[2] += w * a.z; pz[3] += w * a.w;
          pz[4] += w * b.x; pz[5] += w * b.y; pz[6] += w * b.z; pz[7] += w * b.w;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { accA[it][j] += wA * pz[j]; accB[it][j] += wB * pz[j]; }
    }
    buf ^= 1;
  }
  flush(cur, accA);
  flush(cur + 1, accB);
}

// ---------------- side map up-sampling to NCDHW (block-level API / tests only) ------------------
__global__ void side_upsample_kernel(const float* __restrict__ side, int C, int scale,
                                     float* __restrict__ out, int c_total, int c_off, int D, int H, int W,
                                     long long total) {
  const int Do = D * scale, Ho = H * scale, Wo = W * scale;
  const float rz = ac_scale(D, Do), ry = ac_scale(H, Ho), rx = ac_scale(W, Wo);
  const long long Vo = (long long)Do * Ho * Wo;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho); r /= Ho;
    const int zo = (int)(r % Do);
    const long long n = r / Do;
    int z0, z1, y0, y1, x0, x1; float lz, ly, lx;
    ac_src(zo, rz, D, z0, z1, lz);
    ac_src(yo, ry, H, y0, y1, ly);
    ac_src(xo, rx, W, x0, x1, lx);
    for (int c = 0; c < C; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int z = (k & 4) ? z1 : z0, y = (k & 2) ? y1 : y0, x = (k & 1) ? x1 : x0;
        const float w = ((k & 4) ? lz : 1.f - lz) * ((k & 2) ? ly : 1.f - ly) * ((k & 1) ? lx : 1.f - lx);
        acc += w * side[((((n * D + z) * H + y) * W + x) * (long long)C) + c];
      }
      out[(n * c_total + c_off + c) * Vo + (((long long)zo * Ho + yo) * Wo + xo)] = acc;
    }
  }
}

// ---------------- heads ---------------------------------------------------------------------------
struct HeadLevels {
  const float* map[4];  // level l has extents (D0>>l, H0>>l, W0>>l); null = level absent
};

// One thread = 4 consecutive x of one (n, z, y) row: the z / y source rows and weights of every level are shared by the
// four outputs (the per-voxel form spent ~260 vector instructions per voxel, 87 us per head at 4x128^3).
__global__ void __launch_bounds__(256)
head_fwd_kernel(HeadLevels lv, const float* __restrict__ bias, float* __restrict__ pred,
                int D, int H, int W, long long total4) {
  const int W4 = W >> 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int xq = (int)(r % W4); r /= W4;
    const int yo = (int)(r % H); r /= H;
    const int zo = (int)(r % D);
    const long long n = r / D;
    const long long o = ((n * D + zo) * H + yo) * (long long)W + 4 * xq;
    float acc[4];
    const float b0 = bias[0];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = b0;
    if (lv.map[0]) {
      const float4 v = *reinterpret_cast<const float4*>(lv.map[0] + o);
      acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
    }
#pragma unroll
    for (int l = 1; l < 4; ++l) {
      if (!lv.map[l]) continue;
      const int Dl = D >> l, Hl = H >> l, Wl = W >> l;
      int z0, z1, y0, y1; float lz, ly;
      ac_src(zo, ac_scale(Dl, D), Dl, z0, z1, lz);
      ac_src(yo, ac_scale(Hl, H), Hl, y0, y1, ly);
      const float* m = lv.map[l] + n * (long long)Dl * Hl * Wl;
      const float* r00 = m + ((long long)z0 * Hl + y0) * Wl;
      const float* r01 = m + ((long long)z0 * Hl + y1) * Wl;
      const float* r10 = m + ((long long)z1 * Hl + y0) * Wl;
      const float* r11 = m + ((long long)z1 * Hl + y1) * Wl;
      const float w00 = (1.f - lz) * (1.f - ly), w01 = (1.f - lz) * ly, w10 = lz * (1.f - ly), w11 = lz * ly;
      const float rx = ac_scale(Wl, W);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int x0, x1; float lx;
        ac_src(4 * xq + k, rx, Wl, x0, x1, lx);
        // same association as the 8-corner sum: (wz*wy)*wx per corner
        const float a0 = w00 * r00[x0] + w01 * r01[x0] + w10 * r10[x0] + w11 * r11[x0];
        const float a1 = w00 * r00[x1] + w01 * r01[x1] + w10 * r10[x1] + w11 * r11[x1];
        acc[k] += (1.f - lx) * a0 + lx * a1;
      }
    }
    *reinterpret_cast<float4*>(pred + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}
// Row form of the same head: a block per (n, z) plane, ONE WAVE per output row.  The per-voxel form above issues 8 four-byte
// gathers per level and output and decodes its 64-bit flat index per thread (~400 vector instructions per 4 voxels: 62 us per
// head at 4 x 128^3 against 12 us of HBM time).  Here everything that depends on (n, z, y) only is wave-uniform (scalar
// registers); the lanes first blend the four (z, y) source rows of a level with coalesced loads -- c[xc] = w00 r00[xc] + w01
// r01[xc] + w10 r10[xc] + w11 r11[xc], W >> l values -- into a per-wave LDS row, then every lane interpolates its outputs along
// x from that row with x indices / weights tabulated once per thread.  Same products and association as head_fwd_kernel.
constexpr int HF_RPW = 4;     // rows per wave: their loads are issued together (a wave that walks its rows one by one is a
                              // chain of dependent global load -> LDS -> read -> store latencies)
__global__ void __launch_bounds__(256)
head_fwd_rows_kernel(HeadLevels lv, const float* __restrict__ bias, float* __restrict__ pred, int D, int H, int W) {
  extern __shared__ float hrow[];                       // [4 waves][HF_RPW rows][W/2 + W/4 + W/8]
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int per_row = (W >> 1) + (W >> 2) + (W >> 3);
  float* const rb = hrow + wv * HF_RPW * per_row;
  const float b0 = bias[0];
  const int ybl = (H + 4 * HF_RPW - 1) / (4 * HF_RPW);           // y blocks per plane
  const int yb = blockIdx.x % ybl, zo = (blockIdx.x / ybl) % D, n = blockIdx.x / (ybl * D);
  const int ybase = (yb * 4 + wv) * HF_RPW;
  // the level-0 values of the first pass over x (requested first: they arrive while the coarse rows are blended)
  float2 v0[HF_RPW];
#pragma unroll
  for (int r = 0; r < HF_RPW; ++r) {
    const int yo = ybase + r < H ? ybase + r : H - 1;
    v0[r] = (lv.map[0] && 2 * lane < W) ? *reinterpret_cast<const float2*>(lv.map[0] + (((long long)n * D + zo) * H + yo) * W + 2 * lane)
                                        : make_float2(0.f, 0.f);
  }
  // x tables of this lane's first two outputs (x = 2 lane, 2 lane + 1); further passes (W > 128) recompute
  int tx0[3][2], tx1[3][2]; float tlx[3][2];
#pragma unroll
  for (int l = 1; l < 4; ++l)
#pragma unroll
    for (int k = 0; k < 2; ++k) ac_src(2 * lane + k, ac_scale(W >> l, W), W >> l, tx0[l - 1][k], tx1[l - 1][k], tlx[l - 1][k]);
#pragma unroll
  for (int l = 1; l < 4; ++l) {
    if (!lv.map[l]) continue;
    const int Dl = D >> l, Hl = H >> l, Wl = W >> l;
    int z0, z1; float lz;
    ac_src(zo, ac_scale(Dl, D), Dl, z0, z1, lz);
    const float* m = lv.map[l] + (long long)n * Dl * Hl * Wl;
    const int lbase = (l > 1 ? (W >> 1) : 0) + (l > 2 ? (W >> 2) : 0);
#pragma unroll
    for (int r = 0; r < HF_RPW; ++r) {
      const int yo = ybase + r < H ? ybase + r : H - 1;
      int y0, y1; float ly;
      ac_src(yo, ac_scale(Hl, H), Hl, y0, y1, ly);
      const float* r00 = m + (z0 * Hl + y0) * Wl;
      const float* r01 = m + (z0 * Hl + y1) * Wl;
      const float* r10 = m + (z1 * Hl + y0) * Wl;
      const float* r11 = m + (z1 * Hl + y1) * Wl;
      const float w00 = (1.f - lz) * (1.f - ly), w01 = (1.f - lz) * ly, w10 = lz * (1.f - ly), w11 = lz * ly;
      for (int xc = lane; xc < Wl; xc += 64)
        rb[r * per_row + lbase + xc] = w00 * r00[xc] + w01 * r01[xc] + w10 * r10[xc] + w11 * r11[xc];
    }
  }
  __builtin_amdgcn_wave_barrier();                      // LDS operations of one wave complete in order
#pragma unroll
  for (int r = 0; r < HF_RPW; ++r) {
    const int yo = ybase + r;
    if (yo >= H) break;                                 // (wave-uniform)
    const long long orow = (((long long)n * D + zo) * H + yo) * W;
    for (int x = 2 * lane; x < W; x += 128) {
      float acc[2] = {b0, b0};
      if (lv.map[0]) {
        const float2 v = x < 128 ? v0[r] : *reinterpret_cast<const float2*>(lv.map[0] + orow + x);
        acc[0] += v.x; acc[1] += v.y;
      }
      int lb = r * per_row;
#pragma unroll
      for (int l = 1; l < 4; ++l) {
        if (!lv.map[l]) continue;
        const int Wl = W >> l;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          int x0 = tx0[l - 1][k], x1 = tx1[l - 1][k]; float lx = tlx[l - 1][k];
          if (x >= 128) ac_src(x + k, ac_scale(Wl, W), Wl, x0, x1, lx);
          acc[k] += (1.f - lx) * rb[lb + x0] + lx * rb[lb + x1];
        }
        lb += Wl;
      }
      *reinterpret_cast<float2*>(pred + orow + x) = make_float2(acc[0], acc[1]);
    }
  }
}
// transposed 1-D interpolation along one axis of an f32 tensor viewed as [outer][O][inner] -> [outer][I][inner]
__global__ void up_transpose_axis_kernel(const float* __restrict__ in, float* __restrict__ out, int I, int O,
                                         long long inner, long long total) {
  const float rs = ac_scale(I, O);
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long in_i = idx % inner;
    const int i = (int)((idx / inner) % I);
    const long long outer = idx / (inner * I);
    int lo, hi;
    ac_range(i, rs, O, lo, hi);
    float acc = 0.f;
    for (int o = lo; o <= hi; ++o) {
      const float w = ac_weight(o, i, rs, I);
      if (w != 0.f) acc += w * in[(outer * O + o) * inner + in_i];
    }
    out[idx] = acc;
  }
}

// The x-axis pass of head_bwd for ALL levels of a head in one launch, plus the partial sums of the bias gradient: g_pred (the
// largest tensor on this path, 4 B per full-resolution voxel) is read once instead of once per level and once more for
// sum(g_pred).  The interpolation weights depend on the x index only: each block tabulates them once in LDS (range start
// + up to HB_K weights per output and level) and then streams HB_ROWS x-rows, one wave per row.  Same weights and the same
// summation order as up_transpose_axis_kernel (bitwise identical results).
constexpr int HB_ROWS = 32, HB_K = 24, HB_PAD = 24, HB_KA = 8, HB_KB = 24;
__global__ void __launch_bounds__(256)
head_bwd_x_multi_kernel(const float* __restrict__ g, float* __restrict__ t1a, float* __restrict__ t1b, float* __restrict__ t1c,
                        int nl, int W, long long rows, double* __restrict__ bias_part) {
  extern __shared__ float hsm[];                        // [4][W + HB_PAD] row buffers (pad = zeros), then per output the table: lo, n, HB_K weights
  const int RW = W + HB_PAD;
  float* tab = hsm + 4 * RW;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long row0 = (long long)blockIdx.x * HB_ROWS;
  // rows of up to 256 values: ALL HB_ROWS / 4 rows of this wave are requested up front (registers), before the weight
  // table is built: with one row in flight
  // per wave the kernel moved 0.9 TB/s (bytes in flight x waves / HBM latency), not the table look-ups' fault
  constexpr int RPW = HB_ROWS / 4;
  const bool pf = W <= 256;
  float pre[RPW][4];
  if (pf) {
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const long long row = row0 + wv + 4 * j;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int x = lane + 64 * k;
        pre[j][k] = (row < rows && x < W) ? g[row * W + x] : 0.f;
      }
    }
  }
  float* const outs[3] = {t1a, t1b, t1c};
  // table: outputs of level l occupy entries [ebase_l, ebase_l + W >> l)
  int ebase[4] = {0, 0, W >> 1, (W >> 1) + (W >> 2)};
  const int nent = (nl > 1 ? W >> 1 : 0) + (nl > 2 ? W >> 2 : 0) + (nl > 3 ? W >> 3 : 0);
  for (int e = threadIdx.x; e < nent; e += 256) {
    const int l = e < ebase[2] ? 1 : (e < ebase[3] ? 2 : 3);
    const int i = e - ebase[l], Wl = W >> l;
    const float rs = ac_scale(Wl, W);
    int lo, hi;
    ac_range(i, rs, W, lo, hi);
    float* t = tab + e * (HB_K + 2);
    int cnt = 0, first = lo;
    bool started = false;
    for (int o = lo; o <= hi; ++o) {
      const float w = ac_weight(o, i, rs, Wl);
      if (!started && w == 0.f) { first = o + 1; continue; }     // leading zero weights: skipped exactly like `if (w != 0)`
      started = true;
      if (cnt < HB_K) t[2 + cnt] = w;
      ++cnt;
    }
    t[0] = __int_as_float(first);
    t[1] = __int_as_float(cnt < HB_K ? cnt : HB_K);
  }
  for (int i = threadIdx.x; i < 4 * HB_PAD; i += 256) hsm[(i / HB_PAD) * RW + W + i % HB_PAD] = 0.f;
  __syncthreads();
  // Fast path (<= 128 entries, i.e. W <= 146): a lane owns the same two entries (lane, lane + 64) in every row, so their taps
  // live in REGISTERS for the whole block -- the per-row work is then 30-odd LDS reads and FMAs instead of table look-ups.
  // (Measured by elimination at 4 x 128^3: row loads + LDS writes + bias sums 12.8 us, weight table 5 us, taps + stores 21 us;
  // the taps are unbalanced -- 16 lanes carry the ~20-tap level-3 entries -- which is what is left to fix.)  Taps beyond an entry's count carry
  // weight 0 and read the zero pad behind the row; skipping a zero weight and adding 0 * v give the same bits.
  const int e0 = lane, e1 = lane + 64;
  float wa[HB_KA], wb[HB_KB];
  int fa = 0, fb = 0, ca = 0, cb = 0;
  {
    if (e0 < nent) { const float* t = tab + e0 * (HB_K + 2); fa = __float_as_int(t[0]); ca = __float_as_int(t[1]); }
    if (e1 < nent) { const float* t = tab + e1 * (HB_K + 2); fb = __float_as_int(t[0]); cb = __float_as_int(t[1]); }
#pragma unroll
    for (int k = 0; k < HB_KA; ++k) wa[k] = (e0 < nent && k < ca) ? tab[e0 * (HB_K + 2) + 2 + k] : 0.f;
#pragma unroll
    for (int k = 0; k < HB_KB; ++k) wb[k] = (e1 < nent && k < cb) ? tab[e1 * (HB_K + 2) + 2 + k] : 0.f;
  }
  const bool fast = nent <= 128 && __all(ca <= HB_KA && cb <= HB_KB && fa + HB_KA <= RW && fb + HB_KB <= RW);   // (wave-uniform)
  int ka = 0, kb = 0;                                    // taps to walk: the wave's maxima, rounded up to 4
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { ca = max(ca, shfl_xor_settled(ca, off)); cb = max(cb, shfl_xor_settled(cb, off)); }
  ka = (ca + 3) & ~3; kb = (cb + 3) & ~3;
  const int la = e0 < ebase[2] ? 1 : (e0 < ebase[3] ? 2 : 3), lb = e1 < ebase[2] ? 1 : (e1 < ebase[3] ? 2 : 3);
  float* const outa = e0 < nent ? (la == 1 ? t1a : (la == 2 ? t1b : t1c)) : nullptr;
  float* const outb = e1 < nent ? (lb == 1 ? t1a : (lb == 2 ? t1b : t1c)) : nullptr;
  const int cola = e0 - ebase[la], colb = e1 - ebase[lb], wla = W >> la, wlb = W >> lb;
  double sum = 0.0;
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int rr = wv + 4 * j;
    const long long row = row0 + rr;
    if (row >= rows) break;                              // (wave-uniform)
    float* rb = hsm + wv * RW;
    if (pf) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int x = lane + 64 * k;
        if (x < W) { rb[x] = pre[j][k]; sum += (double)pre[j][k]; }
      }
    } else {
      for (int x = lane; x < W; x += 64) { const float v = g[row * W + x]; rb[x] = v; sum += (double)v; }
    }
    __builtin_amdgcn_wave_barrier();                     // LDS operations of one wave complete in order
    if (fast) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int k = 0; k < HB_KA; ++k)
        if (k < ka) a0 += wa[k] * rb[fa + k];
#pragma unroll
      for (int k = 0; k < HB_KB; ++k)
        if (k < kb) a1 += wb[k] * rb[fb + k];
      if (outa) outa[row * wla + cola] = a0;
      if (outb) outb[row * wlb + colb] = a1;
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    for (int e = lane; e < nent; e += 64) {
      const int l = e < ebase[2] ? 1 : (e < ebase[3] ? 2 : 3);
      float* out = outs[l - 1];
      if (out == nullptr) continue;
      const float* t = tab + e * (HB_K + 2);
      const int first = __float_as_int(t[0]), cnt = __float_as_int(t[1]);
      float acc = 0.f;
      for (int k = 0; k < cnt; k += 4) {                 // four taps at a time: independent LDS reads, same order of summation
        float w[4], v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool in = k + j < cnt;
          w[j] = in ? t[2 + k + j] : 0.f;
          v[j] = in ? rb[first + k + j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (w[j] != 0.f) acc += w[j] * v[j];
      }
      out[row * (W >> l) + (e - ebase[l])] = acc;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (bias_part != nullptr) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += shfl_xor_settled(sum, off);
    __shared__ double wsum[4];
    if (lane == 0) wsum[wv] = sum;
    __syncthreads();
    if (threadIdx.x == 0) bias_part[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
  }
}

// deterministic two-stage sum of an f32 array (16-byte loads when the array allows it; f64 accumulation)
__global__ void __launch_bounds__(256) sum_stage1_kernel(const float* __restrict__ in, long long n,
                                                         double* __restrict__ partial) {
  double s = 0.0;
  if ((n & 3) == 0 && (reinterpret_cast<size_t>(in) & 15) == 0) {
    const long long n4 = n >> 2;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
      const float4 v = reinterpret_cast<const float4*>(in)[i];
      s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    }
  } else {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += (double)in[i];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += shfl_xor_settled(s, off);
  __shared__ double w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((w[0] + w[1]) + w[2]) + w[3];
}
__global__ void __launch_bounds__(64) sum_stage2_kernel(const double* __restrict__ partial, int n, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += shfl_xor_settled(s, off);
  if (threadIdx.x == 0) out[0] = (float)s;
}

// the same with 1024 threads for long partial arrays (fixed order: thread-strided sums, wave shuffles, 16 waves in order)
__global__ void __launch_bounds__(1024) sum_stage2_wide_kernel(const double* __restrict__ partial, int n, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += partial[i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += shfl_xor_settled(s, off);
  __shared__ double w[16];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += w[k];
    out[0] = (float)t;
  }
}

// ---------------- launchers -----------------------------------------------------------------------
static inline int grid_for(long long total, int block = 256) {
  long long g = (total + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

int launch_pack_cl(int dtype, const float* in, int C, void* out, int Cpad, Dims d, hipStream_t s) {
  SEUNET_CHECK(Cpad % 8 == 0 && Cpad >= C, "pack_cl: padded channel count %d invalid for C=%d", Cpad, C);
  const long long V = d.vox(), total = (long long)d.N * V * (Cpad / 8);
  SEUNET_DTYPE_SWITCH(dtype, pack_cl_kernel<T><<<grid_for(total), 256, 0, s>>>(in, C, (T*)out, Cpad, V, total));
  SEUNET_LAUNCH_CHECK();
  return 0;
}
int launch_pack_input(int dtype, const float* x, int in_channel, void* out, Dims d, hipStream_t s) {
  SEUNET_CHECK(in_channel >= 1 && in_channel <= 8, "in_channel %d unsupported (1..8)", in_channel);
  return launch_pack_cl(dtype, x, in_channel, out, 8, d, s);
}
int launch_unpack_cl(int dtype, const void* in, int C, float* out, Dims d, hipStream_t s) {
  SEUNET_CHECK(C % 8 == 0, "unpack_cl: C=%d must be a multiple of 8", C);
  const long long V = d.vox(), total = (long long)d.N * V * (C / 8);
  SEUNET_DTYPE_SWITCH(dtype, unpack_cl_kernel<T><<<grid_for(total), 256, 0, s>>>((const T*)in, C, out, V, total));
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_maxpool_fwd(int dtype, const void* in, int C, void* out, Dims d, hipStream_t s) {
  SEUNET_CHECK(C % 8 == 0 && d.D % 2 == 0 && d.H % 2 == 0 && d.W % 2 == 0, "maxpool: bad shape");
  const long long total = (long long)d.N * (d.D / 2) * (d.H / 2) * (d.W / 2) * (C / 8);
  SEUNET_DTYPE_SWITCH(dtype, maxpool_fwd_kernel<T><<<grid_for(total), 256, 0, s>>>((const T*)in, C, (T*)out, d.D, d.H, d.W, total));
  SEUNET_LAUNCH_CHECK();
  return 0;
}
int launch_maxpool_bwd(int dtype, const void* in, const void* g_out, int C, void* g_in, int accumulate,
                       Dims d, hipStream_t s) {
  SEUNET_CHECK(C % 8 == 0 && d.D % 2 == 0 && d.H % 2 == 0 && d.W % 2 == 0, "maxpool: bad shape");
  const long long total = (long long)d.N * (d.D / 2) * (d.H / 2) * (d.W / 2) * (C / 8);
  SEUNET_DTYPE_SWITCH(dtype, maxpool_bwd_kernel<T><<<grid_for(total), 256, 0, s>>>((const T*)in, (const T*)g_out, C, (T*)g_in, accumulate, d.D, d.H, d.W, total));
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_maxpool_bwd_idx(int dtype, const unsigned* argmax, const void* g_out, int C, void* g_in, int accumulate, Dims d, hipStream_t s) {
  SEUNET_CHECK(C % 8 == 0 && d.D % 2 == 0 && d.H % 2 == 0 && d.W % 2 == 0 && argmax, "maxpool_bwd_idx: bad argument");
  const long long total = (long long)d.N * (d.D / 2) * (d.H / 2) * (d.W / 2) * (C / 8);
  SEUNET_DTYPE_SWITCH(dtype, maxpool_bwd_idx_kernel<T><<<grid_for(total), 256, 0, s>>>(argmax, (const T*)g_out, C, (T*)g_in, accumulate, d.D, d.H, d.W, total));
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_upsample2_fwd(int dtype, const void* in, int C, void* out, Dims d, hipStream_t s) {
  SEUNET_CHECK(C % 8 == 0, "upsample2: C=%d must be a multiple of 8", C);
  const long long total = (long long)d.N * d.vox() * 8 * (C / 8);
  if ((C == 32 || C == 64 || C == 128) && d.D >= 2 && d.H >= 2 && d.W >= 2 && getenv("SEUNET_UP_TILED") == nullptr) {
    const int XF = 4096 / C;                               // fine columns per block: four (row, column, 8-channel) items per thread
    const int nseg = (2 * d.D + UFM_ZSF - 1) / UFM_ZSF;
    const size_t lds_m = (size_t)2 * 3 * UFM_XC * C * dtype_size(dtype);   // <= 102 KB except 128 channels in f32 (tiled kernel)
    if ((long long)d.N * nseg <= 65535 && d.H <= 65535 && lds_m <= 112 * 1024) {
      static unsigned long long cfg[3] = {0, 0, 0};
      if (lds_m > 48 * 1024 && dtype >= 0 && dtype < 3)
        SEUNET_DTYPE_SWITCH(dtype, if (int e = configure_kernel_lds(cfg[dtype], reinterpret_cast<const void*>(&upsample2_fwd_march_kernel<T>), 112 * 1024)) return e);
      dim3 grid((unsigned)((2 * d.W + XF - 1) / XF), (unsigned)d.H, (unsigned)(d.N * nseg));
      SEUNET_DTYPE_SWITCH(dtype, upsample2_fwd_march_kernel<T><<<grid, 256, lds_m, s>>>((const T*)in, C, (T*)out, d.D, d.H, d.W, XF));
      SEUNET_LAUNCH_CHECK();
      return 0;
    }
  }
  const size_t lds = (size_t)4 * UF_XC * C * sizeof(float);
  if (lds <= 144 * 1024 && (long long)d.N * d.D <= 65535 && d.H <= 65535) {   // up to 128 channels
    static unsigned long long configured[3] = {0, 0, 0};
    if (lds > 48 * 1024 && dtype >= 0 && dtype < 3)
      SEUNET_DTYPE_SWITCH(dtype, if (int e = configure_kernel_lds(configured[dtype], reinterpret_cast<const void*>(&upsample2_fwd_tiled_kernel<T>), 144 * 1024)) return e);
    dim3 grid((unsigned)((2 * d.W + UF_XF - 1) / UF_XF), (unsigned)d.H, (unsigned)((long long)d.N * d.D));
    SEUNET_DTYPE_SWITCH(dtype, upsample2_fwd_tiled_kernel<T><<<grid, 256, lds, s>>>((const T*)in, C, (T*)out, d.D, d.H, d.W));
    SEUNET_LAUNCH_CHECK();
    return 0;
  }
  SEUNET_DTYPE_SWITCH(dtype, upsample2_fwd_kernel<T><<<grid_for(total), 256, 0, s>>>((const T*)in, C, (T*)out, d.D, d.H, d.W, total));
  SEUNET_LAUNCH_CHECK();
  return 0;
}
template <typename T, int TY>
static void upsample2_bwd_tiled(const void* g_out, int C, void* g_in, int accumulate, Dims d, hipStream_t s) {
  dim3 grid((unsigned)((d.W + UB_TX - 1) / UB_TX), (unsigned)((d.H + TY - 1) / TY), (unsigned)((long long)d.N * d.D));
  const size_t lds = (size_t)TY * UB_XF * C * sizeof(float);
  upsample2_bwd_tiled_kernel<T, TY><<<grid, 256, lds, s>>>((const T*)g_out, C, (T*)g_in, accumulate, d.D, d.H, d.W);
}

template <typename T, int TY>
static int upsample2_bwd_march(const void* g_out, int C, void* g_in, int accumulate, Dims d, int ZS, hipStream_t s) {
  const size_t lds = (size_t)2 * 128 * UM_XF * sizeof(float);   // 40 KB: four workgroups per CU
  static unsigned long long cfg = 0;
  if (int e = configure_kernel_lds(cfg, (const void*)upsample2_bwd_march_kernel<T, TY>, (int)lds)) return e;
  const int nseg = (d.D + ZS - 1) / ZS;
  dim3 grid((unsigned)((d.W + UM_TX - 1) / UM_TX), (unsigned)((d.H + TY - 1) / TY), (unsigned)(d.N * nseg));
  upsample2_bwd_march_kernel<T, TY><<<grid, 256, lds, s>>>((const T*)g_out, C, (T*)g_in, accumulate, d.D, d.H, d.W, ZS);
  return 0;
}

int launch_upsample2_bwd(int dtype, const void* g_out, int C, void* g_in, int accumulate, Dims d,
                         hipStream_t s) {
  SEUNET_CHECK(C % 8 == 0, "upsample2: C=%d must be a multiple of 8", C);
  // z-marching kernel: TY * C = 128 (256 phase-2 items of 8 channels = 1 per thread), LDS = 2 x 128 * 40 floats = 40 KB
  // (16-bit storage only: the f32 parity mode would hold 200 registers of fetched rows; it stays on the tiled kernel)
  if (dtype_size(dtype) == 2 && (C == 32 || C == 64 || C == 128) && d.D >= 4 && d.H >= 4 && d.W >= 4 &&
      (long long)4 * d.H * d.W * C < (1LL << 31) && getenv("SEUNET_UP_TILED") == nullptr) {
    const int ZS = d.D >= 32 ? 8 : 4;
    const int nseg = (d.D + ZS - 1) / ZS;
    SEUNET_CHECK((long long)d.N * nseg <= 65535, "upsample2_bwd: batch too large");
    int rc = 0;
    SEUNET_DTYPE_SWITCH(dtype, { rc = C == 32 ? upsample2_bwd_march<T, 4>(g_out, C, g_in, accumulate, d, ZS, s)
                                    : C == 64 ? upsample2_bwd_march<T, 2>(g_out, C, g_in, accumulate, d, ZS, s)
                                              : upsample2_bwd_march<T, 1>(g_out, C, g_in, accumulate, d, ZS, s); });
    if (rc) return rc;
    SEUNET_LAUNCH_CHECK();
    return 0;
  }
  // tiled separable kernel: LDS = TY * 44 * C floats <= 45 KB with TY = 4 up to 64 channels, TY = 2 up to 128
  const bool tiled = C <= 128 && (long long)d.N * d.D <= 65535 && d.W >= 2;
  if (tiled) {
    SEUNET_DTYPE_SWITCH(dtype, { if (C <= 64) upsample2_bwd_tiled<T, 4>(g_out, C, g_in, accumulate, d, s); else upsample2_bwd_tiled<T, 2>(g_out, C, g_in, accumulate, d, s); });
    SEUNET_LAUNCH_CHECK();
    return 0;
  }
  const long long total = (long long)d.N * d.vox() * (C / 8);
  SEUNET_DTYPE_SWITCH(dtype, upsample2_bwd_kernel<T><<<grid_for(total), 256, 0, s>>>((const T*)g_out, C, (T*)g_in, accumulate, d.D, d.H, d.W, total));
  SEUNET_LAUNCH_CHECK();
  return 0;
}

// zero several small f32 arrays with ONE launch (the identically-zero conv1.bias gradients of a backward pass: a
// hipMemsetAsync each is a 5-us fill kernel, thirty times per step)
struct ZeroList { float* ptr[48]; int count[48]; int n; };
__global__ void multi_zero_kernel(ZeroList z) {
  float* p = z.ptr[blockIdx.x];
  for (int i = threadIdx.x; i < z.count[blockIdx.x]; i += blockDim.x) p[i] = 0.f;
}
int launch_multi_zero(float* const* ptrs, const int* counts, int n, hipStream_t s) {
  for (int base = 0; base < n; base += 48) {
    ZeroList z{};
    z.n = n - base < 48 ? n - base : 48;
    for (int i = 0; i < z.n; ++i) { z.ptr[i] = ptrs[base + i]; z.count[i] = counts[base + i]; }
    multi_zero_kernel<<<z.n, 256, 0, s>>>(z);
  }
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_side_upsample(const float* side, int C, int scale, float* out, int c_total, int c_off, Dims dl,
                         hipStream_t s) {
  const long long total = (long long)dl.N * dl.vox() * scale * scale * scale;
  side_upsample_kernel<<<grid_for(total), 256, 0, s>>>(side, C, scale, out, c_total, c_off, dl.D, dl.H, dl.W, total);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_head_fwd(const float* const* level_maps, int nlevels, const float* bias, float* pred, Dims d0,
                    hipStream_t s) {
  SEUNET_CHECK(nlevels >= 1 && nlevels <= 4, "head: nlevels=%d out of range", nlevels);
  HeadLevels lv;
  for (int l = 0; l < 4; ++l) lv.map[l] = l < nlevels ? level_maps[l] : nullptr;
  for (int l = 1; l < nlevels; ++l)
    SEUNET_CHECK((d0.D >> l) >= 1 && (d0.H >> l) >= 1 && (d0.W >> l) >= 1, "head: volume too small for level %d", l);
  SEUNET_CHECK(d0.W % 4 == 0, "head: W=%d must be a multiple of 4", d0.W);
  if (d0.W % 8 == 0 && d0.W <= 2048 && (long long)d0.D * d0.H * d0.W < (1ll << 31)) {   // row form: whole level rows, LDS rows fit, 32-bit in-sample offsets
    const size_t lds = (size_t)4 * HF_RPW * ((d0.W >> 1) + (d0.W >> 2) + (d0.W >> 3)) * sizeof(float);
    const int ybl = (d0.H + 4 * HF_RPW - 1) / (4 * HF_RPW);
    head_fwd_rows_kernel<<<d0.N * d0.D * ybl, 256, lds, s>>>(lv, bias, pred, d0.D, d0.H, d0.W);
  } else {
    const long long total4 = (long long)d0.N * d0.vox() / 4;
    head_fwd_kernel<<<grid_for(total4), 256, 0, s>>>(lv, bias, pred, d0.D, d0.H, d0.W, total4);
  }
  SEUNET_LAUNCH_CHECK();
  return 0;
}

// the y- (or z-) passes of all coarse levels of one head in ONE launch: blockIdx.y = job (one per level)
struct AxisJob { const float* in; float* out; int I, O; long long inner, total; };
struct AxisJobs { AxisJob j[3]; };
__global__ void __launch_bounds__(256) up_transpose_axis_multi_kernel(AxisJobs jobs) {
  const AxisJob jb = jobs.j[blockIdx.y];
  if (jb.total == 0) return;
  const float rs = ac_scale(jb.I, jb.O);
  // (32-bit index arithmetic: a 64-bit division costs ~100 vector instructions and this loop has three per output; the launcher
  // checks that every flat index of the job fits)
  const unsigned inner = (unsigned)jb.inner, I = (unsigned)jb.I, total = (unsigned)jb.total;
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const unsigned q = idx / inner;
    const unsigned in_i = idx - q * inner;
    const unsigned outer = q / I;
    const int i = (int)(q - outer * I);
    int lo, hi;
    ac_range(i, rs, jb.O, lo, hi);
    float acc = 0.f;
    // four taps at a time: their loads are independent and go out together (same taps, same order of summation)
    for (int o = lo; o <= hi; o += 4) {
      float w[4], v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        w[k] = o + k <= hi ? ac_weight(o + k, i, rs, jb.I) : 0.f;
        v[k] = w[k] != 0.f ? jb.in[(outer * (unsigned)jb.O + (unsigned)(o + k)) * inner + in_i] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (w[k] != 0.f) acc += w[k] * v[k];
    }
    jb.out[idx] = acc;
  }
}

size_t head_bwd_tmp_floats(Dims d0) {
  // x-pass outputs of the (<= 3) coarse levels (7/8 of a full-resolution map) + their y-pass outputs (<= 21/64) + bias partials (f64)
  const size_t v = (size_t)d0.N * d0.vox();
  return v + v / 2 + 64 + 2 * (((size_t)d0.N * d0.D * d0.H + HB_ROWS - 1) / HB_ROWS + 64);
}
int launch_head_bwd(const float* g_pred, float* const* g_levels, int nlevels, float* tmp, float* g_bias,
                    Dims d0, hipStream_t s) {
  SEUNET_CHECK(nlevels >= 1 && nlevels <= 4, "head: nlevels=%d out of range", nlevels);
  const long long rows = (long long)d0.N * d0.D * d0.H;
  const long long V = rows * d0.W;
  // workspace: t1[l] = [N][D0][H0][Wl] for l = 1..3, then t2 (one level at a time), then the bias partials
  float* t1[4] = {nullptr, nullptr, nullptr, nullptr};
  long long off = 0;
  for (int l = 1; l < nlevels; ++l) {
    if (g_levels[l]) t1[l] = tmp + off;
    off += rows * (d0.W >> l);
  }
  float* t2[4] = {nullptr, nullptr, nullptr, nullptr};      // y-pass outputs [N][D0][Hl][Wl], after the t1 region (7/8 V)
  long long off2 = V - V / 8;
  for (int l = 1; l < nlevels; ++l) {
    t2[l] = tmp + off2;
    off2 += (long long)d0.N * d0.D * (d0.H >> l) * (d0.W >> l);
  }
  double* part = reinterpret_cast<double*>(tmp + ((V + V / 2 + 64 + 1) & ~1ll));
  const int nblk = (int)((rows + HB_ROWS - 1) / HB_ROWS);
  SEUNET_CHECK(d0.W <= 1024, "head_bwd: W=%d too large", d0.W);
  SEUNET_CHECK(V / 2 < (1ll << 31), "head_bwd: %lld voxels per call exceed the 32-bit index range of the axis passes", V);
  const size_t lds = ((size_t)4 * (d0.W + HB_PAD) + (size_t)(d0.W - (d0.W >> 3)) * (HB_K + 2)) * sizeof(float);
  head_bwd_x_multi_kernel<<<nblk, 256, lds, s>>>(g_pred, t1[1], t1[2], t1[3], nlevels, d0.W, rows, g_bias ? part : nullptr);
  // y-pass of every level in one launch, then z-pass of every level in one launch
  AxisJobs jy{}, jz{};
  long long max_y = 0, max_z = 0;
  int njobs = 0;
  for (int l = 1; l < nlevels; ++l) {
    if (!g_levels[l]) continue;
    const int Dl = d0.D >> l, Hl = d0.H >> l, Wl = d0.W >> l;
    const long long ty = (long long)d0.N * d0.D * Hl * Wl, tz = (long long)d0.N * Dl * Hl * Wl;
    jy.j[njobs] = AxisJob{t1[l], t2[l], Hl, d0.H, (long long)Wl, ty};
    jz.j[njobs] = AxisJob{t2[l], g_levels[l], Dl, d0.D, (long long)Hl * Wl, tz};
    max_y = ty > max_y ? ty : max_y;
    max_z = tz > max_z ? tz : max_z;
    ++njobs;
  }
  if (njobs) {
    up_transpose_axis_multi_kernel<<<dim3((unsigned)grid_for(max_y), (unsigned)njobs), 256, 0, s>>>(jy);
    up_transpose_axis_multi_kernel<<<dim3((unsigned)grid_for(max_z), (unsigned)njobs), 256, 0, s>>>(jz);
  }
  if (g_bias) sum_stage2_wide_kernel<<<1, 1024, 0, s>>>(part, nblk, g_bias);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace seunet
