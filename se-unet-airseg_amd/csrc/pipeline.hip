// GPU input pipeline: what the reference's Dataset.__getitem__ does to ONE resident case between the file
// read and the tensors of the training step, for a whole mini-batch of crops in one launch:
//   crop extraction            data.py:645-664 (CropSegData.crop), :85-252 (*_sample helpers): img[z0:z0+n, y0:y0+n, x0:x0+n]
//   two HU windows             data.py:667-677 / 286-299 / 775-784 (= prediction.py:39-49): clip to [-1024,1024] -> (v+1024)/2048,
//                              clip to [-1000,500] -> (v+1000)/1500
//   label binarisation         data.py:675: (mask > 0)
//   LIB-weight exponentiation  data.py:701 / 389 / 561: weight ** (U + 2) * label + (1 - label), U drawn by the caller
//   flip / rotate augmentation data.py:40-67 (random_flip, random_rotate), composed by the caller into one signed axis map
// and writes the step's tensors directly: data (B, 2, n, n, n), label / weight / skeleton (B, 1, n, n, n), f32 NCDHW -- the
// layout train.py:587-592 builds with transpose(0,1) + cat -- so the four H2D copies per step (train.py:582-585) disappear.
// Random numbers stay on the host (the reference's numpy / random generators); the kernels are pure functions of them.
//
// HBM-bound byte work.  The reference's rotations swap the two fast axes (data.py:51-58), i.e. a transpose: tiles of
// 32 x 32 voxels go through LDS so that both the gather (along the source's contiguous axis) and the store (along the
// destination's) are coalesced.
#include "seunet_common.h"
#include <hip/hip_fp16.h>

namespace seunet {

#define SEUNET_MAX_CROPS 32
struct CropList {
  int n;
  int z[SEUNET_MAX_CROPS], y[SEUNET_MAX_CROPS], x[SEUNET_MAX_CROPS];   // crop origin in the case volume
  unsigned char aug[SEUNET_MAX_CROPS];                                  // bit0..2: reverse source axis 0..2, bit3: swap axes 1,2
};

enum { IMG_I16 = 0, IMG_F32 = 1 };
enum { W_F16 = 0, W_F32 = 1, W_F64 = 2 };

// the two HU windows.  f64_math: integer volumes divide in float64 (numpy true division of an int array,
// data.py:286-299 on the int16 crops; prediction.py:40 `astype(float)`) and the step rounds to f32 afterwards
// (`.float()`, train.py:582); float32 volumes divide in float32 (data.py:667-677, 775-784).  IEEE division both ways.
__device__ __forceinline__ void hu_windows(float v, int f64_math, float& c0, float& c1) {
  const float a = fminf(fmaxf(v, -1024.f), 1024.f), b = fminf(fmaxf(v, -1000.f), 500.f);
  if (f64_math) {
    c0 = (float)(((double)a + 1024.0) / 2048.0);
    c1 = (float)(((double)b + 1000.0) / 1500.0);
  } else {
    c0 = (a + 1024.f) / 2048.f;
    c1 = (b + 1000.f) / 1500.f;
  }
}

__device__ __forceinline__ float load_img(const void* img, int dtype, long long i) {
  return dtype == IMG_I16 ? (float)reinterpret_cast<const short*>(img)[i] : reinterpret_cast<const float*>(img)[i];
}

// weight ** e * label + (1 - label) for a binary label: 1 outside the mask, weight ** e inside.  numpy computes the power
// in the array's dtype with the python-float exponent cast to it (NEP 50): the LIB weights are stored as float16
// (lib_weight.py:50), whose power numpy evaluates as half(powf(float(w), float(half(e)))).  The f32 power is taken in
// double and rounded once (what a correctly rounded powf returns).
__device__ __forceinline__ float lib_weight(const void* w, int dtype, long long i, double e, float e16, float e32, bool fg) {
  if (!fg) return 1.0f;
  if (dtype == W_F16) {
    const float wf = __half2float(reinterpret_cast<const __half*>(w)[i]);
    return __half2float(__float2half_rn((float)pow((double)wf, (double)e16)));
  }
  if (dtype == W_F32) return (float)pow((double)reinterpret_cast<const float*>(w)[i], (double)e32);
  return (float)pow(reinterpret_cast<const double*>(w)[i], e);
}

// grid: (n/32 tiles along o1) * (n/32 tiles along o2), o0 = blockIdx.y, crop = blockIdx.z; block 32 x 8.
// Output voxel o = (o0, o1, o2) of crop k reads source voxel s (crop-local):  (t1, t2) = swap ? (o2, o1) : (o1, o2);
// s0 = r0 ? n-1-o0 : o0;  s1 = r1 ? n-1-t1 : t1;  s2 = r2 ? n-1-t2 : t2.
__global__ void __launch_bounds__(256)
crop_batch_kernel(const void* __restrict__ img, int img_dtype, const unsigned char* __restrict__ label, const void* __restrict__ weight,
                  int w_dtype, const unsigned char* __restrict__ skel, int D, int H, int W, int n, CropList cl, double wexp, float wexp16,
                  float wexp32, int f64_math, float* __restrict__ data_out, float* __restrict__ label_out,
                  float* __restrict__ weight_out, float* __restrict__ skel_out) {
  __shared__ float t_c0[32][33], t_c1[32][33], t_w[32][33], t_s[32][33];
  __shared__ unsigned char t_l[32][36];
  const int k = blockIdx.z, o0 = blockIdx.y;
  const int tiles = n >> 5;
  const int tile1 = (blockIdx.x / tiles) << 5, tile2 = (blockIdx.x % tiles) << 5;   // tile origin in (o1, o2)
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const unsigned a = cl.aug[k];
  const bool r0 = a & 1, r1 = a & 2, r2 = a & 4, swap = a & 8;
  const int s0 = r0 ? n - 1 - o0 : o0;
  const long long plane = ((long long)(cl.z[k] + s0) * H + cl.y[k]) * W + cl.x[k];
  // gather: tx walks the source's contiguous axis (s2)
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int row = ty + r;                      // index along the source's axis 1 within the tile
    // tile in source coordinates: axis-1 range = (swap ? tile2 : tile1), axis-2 range = (swap ? tile1 : tile2), then reversal
    const int t1 = (swap ? tile2 : tile1) + row, t2 = (swap ? tile1 : tile2) + tx;
    const int s1 = r1 ? n - 1 - t1 : t1, s2 = r2 ? n - 1 - t2 : t2;
    const long long si = plane + (long long)s1 * W + s2;
    float c0, c1;
    hu_windows(load_img(img, img_dtype, si), f64_math, c0, c1);
    t_c0[row][tx] = c0;
    t_c1[row][tx] = c1;
    const bool fg = label ? label[si] > 0 : false;
    t_l[row][tx] = fg ? 1 : 0;
    if (weight) t_w[row][tx] = lib_weight(weight, w_dtype, si, wexp, wexp16, wexp32, fg);
    if (skel) t_s[row][tx] = (float)skel[si];
  }
  __syncthreads();
  const long long vol = (long long)n * n * n;
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int o1 = tile1 + ty + r, o2 = tile2 + tx;
    // LDS element of output (o1, o2): source tile coordinates (t1 - base1, t2 - base2) with (t1, t2) = swap ? (o2, o1) : (o1, o2)
    const int a1 = swap ? tx : ty + r, a2 = swap ? ty + r : tx;
    const long long oi = ((long long)o0 * n + o1) * n + o2;
    data_out[(long long)k * 2 * vol + oi] = t_c0[a1][a2];
    data_out[(long long)k * 2 * vol + vol + oi] = t_c1[a1][a2];
    if (label_out) label_out[(long long)k * vol + oi] = (float)t_l[a1][a2];
    if (weight_out) weight_out[(long long)k * vol + oi] = t_w[a1][a2];
    if (skel_out) skel_out[(long long)k * vol + oi] = t_s[a1][a2];
  }
}

// whole-volume two-channel input of the inference / validation loops (prediction.py:39-49,71-75; data.py:775-784,796-798):
// out[0] = 2048-window, out[1] = 1500-window, (2, X, Y, Z) f32
__global__ void __launch_bounds__(256)
hu_two_channel_kernel(const void* __restrict__ img, int img_dtype, long long nvox, int f64_math, float* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= nvox) return;
  float c0, c1;
  hu_windows(load_img(img, img_dtype, i), f64_math, c0, c1);
  out[i] = c0;
  out[nvox + i] = c1;
}

int launch_crop_batch(const void* img, int img_dtype, const unsigned char* label, const void* weight, int w_dtype,
                      const unsigned char* skel, int D, int H, int W, int cube, int ncrop, const int* starts, const int* aug,
                      double weight_exponent, int f64_math, float* data_out, float* label_out, float* weight_out, float* skel_out,
                      hipStream_t s) {
  SEUNET_CHECK(img && data_out && starts, "crop_batch: null argument");
  SEUNET_CHECK(img_dtype == IMG_I16 || img_dtype == IMG_F32, "crop_batch: image dtype %d (0 = int16, 1 = float32)", img_dtype);
  SEUNET_CHECK(ncrop >= 1 && ncrop <= SEUNET_MAX_CROPS, "crop_batch: %d crops per call (1..%d)", ncrop, SEUNET_MAX_CROPS);
  SEUNET_CHECK(cube >= 32 && cube % 32 == 0, "crop_batch: cube %d must be a multiple of 32", cube);
  SEUNET_CHECK(!label_out || label, "crop_batch: label output without a label volume");
  SEUNET_CHECK(!weight_out || (weight && label), "crop_batch: weight output needs the weight AND the label volume (data.py:701)");
  SEUNET_CHECK(!weight || (w_dtype >= W_F16 && w_dtype <= W_F64), "crop_batch: weight dtype %d (0 = f16, 1 = f32, 2 = f64)", w_dtype);
  SEUNET_CHECK(!skel_out || skel, "crop_batch: skeleton output without a skeleton volume");
  CropList cl;
  cl.n = ncrop;
  for (int k = 0; k < ncrop; ++k) {
    cl.z[k] = starts[3 * k]; cl.y[k] = starts[3 * k + 1]; cl.x[k] = starts[3 * k + 2];
    SEUNET_CHECK(cl.z[k] >= 0 && cl.y[k] >= 0 && cl.x[k] >= 0 && cl.z[k] + cube <= D && cl.y[k] + cube <= H && cl.x[k] + cube <= W,
                 "crop %d at (%d,%d,%d) + %d leaves the %dx%dx%d volume", k, cl.z[k], cl.y[k], cl.x[k], cube, D, H, W);
    const int a = aug ? aug[k] : 0;
    SEUNET_CHECK(a >= 0 && a < 16, "crop %d: augmentation code %d (bits 0-2 reverse axes, bit 3 swaps axes 1 and 2)", k, a);
    cl.aug[k] = (unsigned char)a;
  }
  const float e32 = (float)weight_exponent;
  const float e16 = (float)(_Float16)weight_exponent;   // numpy rounds the python float straight to half (one rounding)
  const int tiles = cube / 32;
  crop_batch_kernel<<<dim3(tiles * tiles, cube, ncrop), 256, 0, s>>>(img, img_dtype, label, weight, w_dtype, skel, D, H, W, cube, cl,
                                                                       weight_exponent, e16, e32, f64_math, data_out, label_out,
                                                                       weight_out, skel_out);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_hu_two_channel(const void* img, int img_dtype, long long nvox, int f64_math, float* out, hipStream_t s) {
  SEUNET_CHECK(img && out && nvox >= 1, "hu_two_channel: bad argument");
  SEUNET_CHECK(img_dtype == IMG_I16 || img_dtype == IMG_F32, "hu_two_channel: image dtype %d (0 = int16, 1 = float32)", img_dtype);
  hu_two_channel_kernel<<<(unsigned)((nvox + 255) / 256), 256, 0, s>>>(img, img_dtype, nvox, f64_math, out);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace seunet
