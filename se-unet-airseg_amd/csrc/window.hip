// Sliding-window assembly on the device: the data movement of the reference's whole-volume loops
//   prediction.py:78-109  (batch 1, float64 host accumulators `pred`, `pred_num`)
//   train.py:682-691 + data.py:731-773 (validation: batches of windows, the window list padded with copies of window 0,
//   the copies accumulated too; test.py:151-161 is the same loop)
// as three small kernels around the network forward:
//   window_gather      x[:, :, xl:xr, yl:yr, zl:zr] for a batch of windows -> one (nwin, C, cube^3) f32 NCDHW tensor
//   window_accumulate  pred[xl:xr, yl:yr, zl:zr] += sigmoid(logits[k])   in float64, windows in list order
//   window_finalize    pred / pred_num, with pred_num rebuilt from the window table (integer counts, exact)
// All three are HBM-bound streaming kernels: 16-byte accesses along z (the contiguous axis of the volume).
// Windows of one batch may overlap, and float64 addition is order dependent in the last bit: accumulate launches one
// grid per window, in list order (stream order = the reference's `for i in range(len(pos))` order), no atomics.
#include "seunet_common.h"

namespace seunet {

#define SEUNET_MAX_WINDOWS 64
struct WinList { int n; int x[SEUNET_MAX_WINDOWS], y[SEUNET_MAX_WINDOWS], z[SEUNET_MAX_WINDOWS]; };

// one thread = 4 consecutive z voxels of one (window, channel, x, y) row.  cube % 4 == 0; rows of the volume need not
// be 16-byte aligned (Z is arbitrary, zl too), so the loads are scalar-merged by the hardware rather than forced wide.
__global__ void __launch_bounds__(256)
window_gather_kernel(const float* __restrict__ vol, int C, int X, int Y, int Z, int cube, WinList wl, float* __restrict__ out) {
  const int q = cube >> 2;
  const long long per_win = (long long)C * cube * cube * q;
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  const int k = blockIdx.y;
  if (idx >= per_win) return;
  const int zq = (int)(idx % q);
  long long r = idx / q;
  const int yy = (int)(r % cube); r /= cube;
  const int xx = (int)(r % cube);
  const int c = (int)(r / cube);
  const float* src = vol + (((long long)c * X + (wl.x[k] + xx)) * Y + (wl.y[k] + yy)) * Z + wl.z[k] + zq * 4;
  float4 v;
  v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3];
  reinterpret_cast<float4*>(out + (long long)k * C * cube * cube * cube)[idx] = v;
}

__global__ void __launch_bounds__(256)
window_accumulate_kernel(const float* __restrict__ logits, int apply_sigmoid, int cube, int x0, int y0, int z0, int Y, int Z,
                         double* __restrict__ acc) {
  const int q = cube >> 2;
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx >= (long long)cube * cube * q) return;
  const int zq = (int)(idx % q);
  const long long r = idx / q;
  const int yy = (int)(r % cube), xx = (int)(r / cube);
  const float4 v = reinterpret_cast<const float4*>(logits)[idx];
  float p[4] = {v.x, v.y, v.z, v.w};
  double* dst = acc + ((long long)(x0 + xx) * Y + (y0 + yy)) * Z + z0 + zq * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // torch.sigmoid in f32 (prediction.py:104, train.py:684), then the float64 add of `pred[...] += p` (numpy promotes)
    const float s = apply_sigmoid ? 1.0f / (1.0f + expf(-p[i])) : p[i];
    dst[i] += (double)s;
  }
}

struct AxisStarts { int n; int s[SEUNET_MAX_WINDOWS]; };

__device__ __forceinline__ int axis_count(const AxisStarts& a, int v, int cube) {
  int c = 0;
  for (int i = 0; i < a.n; ++i) c += (v >= a.s[i] && v < a.s[i] + cube) ? 1 : 0;
  return c;
}

// out = acc / pred_num.  The windows form a product grid (every (xl, yl, zl) combination: prediction.py:83-100,
// data.py:745-763), so pred_num(x, y, z) = cx(x) * cy(y) * cz(z); `dup0` extra copies of window 0 = (xs[0], ys[0], zs[0])
// (data.py:764-765) add dup0 inside that window.  A voxel no window covers cannot exist (the last window is shifted back).
__global__ void __launch_bounds__(256)
window_finalize_kernel(const double* __restrict__ acc, int X, int Y, int Z, int cube, AxisStarts ax, AxisStarts ay, AxisStarts az,
                       int dup0, double* __restrict__ out) {
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx >= (long long)X * Y * Z) return;
  const int z = (int)(idx % Z);
  const long long r = idx / Z;
  const int y = (int)(r % Y), x = (int)(r / Y);
  int cnt = axis_count(ax, x, cube) * axis_count(ay, y, cube) * axis_count(az, z, cube);
  if (dup0 > 0 && x >= ax.s[0] && x < ax.s[0] + cube && y >= ay.s[0] && y < ay.s[0] + cube && z >= az.s[0] && z < az.s[0] + cube)
    cnt += dup0;
  out[idx] = acc[idx] / (double)cnt;
}

static int fill_winlist(WinList& wl, int nwin, const int* starts, int cube, int X, int Y, int Z) {
  SEUNET_CHECK(nwin >= 1 && nwin <= SEUNET_MAX_WINDOWS, "window: %d windows per call (1..%d)", nwin, SEUNET_MAX_WINDOWS);
  SEUNET_CHECK(cube >= 4 && cube % 4 == 0, "window: cube %d must be a multiple of 4", cube);
  wl.n = nwin;
  for (int k = 0; k < nwin; ++k) {
    wl.x[k] = starts[3 * k]; wl.y[k] = starts[3 * k + 1]; wl.z[k] = starts[3 * k + 2];
    SEUNET_CHECK(wl.x[k] >= 0 && wl.y[k] >= 0 && wl.z[k] >= 0 && wl.x[k] + cube <= X && wl.y[k] + cube <= Y && wl.z[k] + cube <= Z,
                 "window %d at (%d,%d,%d) + %d leaves the %dx%dx%d volume", k, wl.x[k], wl.y[k], wl.z[k], cube, X, Y, Z);
  }
  return 0;
}

int launch_window_gather(const float* vol, int C, int X, int Y, int Z, int cube, int nwin, const int* starts, float* out, hipStream_t s) {
  SEUNET_CHECK(vol && out && starts && C >= 1, "window_gather: bad argument");
  WinList wl;
  if (int e = fill_winlist(wl, nwin, starts, cube, X, Y, Z)) return e;
  const long long per_win = (long long)C * cube * cube * (cube / 4);
  window_gather_kernel<<<dim3((unsigned)((per_win + 255) / 256), nwin), 256, 0, s>>>(vol, C, X, Y, Z, cube, wl, out);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_window_accumulate(const float* logits, int apply_sigmoid, int nwin, const int* starts, int cube, double* acc, int X, int Y,
                             int Z, hipStream_t s) {
  SEUNET_CHECK(logits && acc && starts, "window_accumulate: bad argument");
  WinList wl;
  if (int e = fill_winlist(wl, nwin, starts, cube, X, Y, Z)) return e;
  const long long per_win = (long long)cube * cube * (cube / 4);
  for (int k = 0; k < nwin; ++k)   // one grid per window, in list order: overlapping windows of a batch never race
    window_accumulate_kernel<<<(unsigned)((per_win + 255) / 256), 256, 0, s>>>(logits + (long long)k * cube * cube * cube, apply_sigmoid,
                                                                                cube, wl.x[k], wl.y[k], wl.z[k], Y, Z, acc);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_window_finalize(const double* acc, int X, int Y, int Z, int cube, int nx, const int* xs, int ny, const int* ys, int nz,
                           const int* zs, int dup0, double* out, hipStream_t s) {
  SEUNET_CHECK(acc && out && xs && ys && zs, "window_finalize: bad argument");
  SEUNET_CHECK(nx >= 1 && ny >= 1 && nz >= 1 && nx <= SEUNET_MAX_WINDOWS && ny <= SEUNET_MAX_WINDOWS && nz <= SEUNET_MAX_WINDOWS,
               "window_finalize: 1..%d window starts per axis", SEUNET_MAX_WINDOWS);
  SEUNET_CHECK(dup0 >= 0, "window_finalize: negative duplicate count");
  AxisStarts ax{nx, {}}, ay{ny, {}}, az{nz, {}};
  for (int i = 0; i < nx; ++i) ax.s[i] = xs[i];
  for (int i = 0; i < ny; ++i) ay.s[i] = ys[i];
  for (int i = 0; i < nz; ++i) az.s[i] = zs[i];
  // every voxel must be covered (the reference would divide by zero): starts begin at 0 and the last window ends at the extent
  auto covered = [&](const AxisStarts& a, int extent) {
    int end = 0;
    for (int i = 0; i < a.n; ++i) { if (a.s[i] > end) return false; end = a.s[i] + cube > end ? a.s[i] + cube : end; }
    return end >= extent;
  };
  SEUNET_CHECK(covered(ax, X) && covered(ay, Y) && covered(az, Z), "window_finalize: the window table leaves voxels uncovered");
  const long long n = (long long)X * Y * Z;
  window_finalize_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(acc, X, Y, Z, cube, ax, ay, az, dup0, out);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace seunet
