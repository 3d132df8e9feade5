// double_threshold_iteration on the GPU.  The reference has three copies that differ in ONE line:
// prediction.py:13-37 keeps pred*255 in float64 (:19); train.py:25-49 and test.py:18-42 (validation / test) round it to
// float32 (train.py:31, test.py:24), so that numpy compares in float32 against thresholds rounded to float32 (NEP 50;
// requirements.txt pins numpy 2.1.3).  Voxels within a float32 ulp of a threshold classify differently: `f32` selects the copy.
//
// What the reference computes: pred*255 is thresholded into strong (>= h*255) and weak (>= l*255, < h*255)
// voxels; gbin starts as the strong mask; then ONE in-place raster-order sweep (i outer, k inner -- the `while` body runs
// exactly once) turns a weak voxel on if any of its 26 neighbours (indices clamped to the volume, i.e. out-of-range
// neighbours add nothing new) is on AT THAT MOMENT: neighbours earlier in raster order have their updated value, later
// ones their original value.  The result depends on that visiting order, so it is reproduced exactly:
//
//   * rows (i, j) are bit-packed along k (64 voxels per word).  Inside a row the sweep is the carry recurrence
//         g[k] = a[k] | (weak[k] & g[k-1]),   a = strong | (weak & (ext | strong[k+1]))
//     where ext ORs the (k-1, k, k+1) bits of the 8 neighbouring rows; the recurrence is exactly the carry chain of the
//     integer addition a + (a | weak), so a row of 64 voxels costs a few 64-bit operations;
//   * row (i, j) needs the FINAL rows (i-1, j-1..j+1), (i, j-1) and the ORIGINAL rows (i, j+1), (i+1, *): all rows with
//     2i + j = t are independent and ordered after t-1, so one workgroup walks the skewed wavefront t = 0 .. 2(h-1)+(w-1)
//     with one thread per row and a barrier per step.  512^3: 1534 steps of <= 256 rows.
// Integer / bit work: the result is bit-identical to the reference loop (tests/test_postprocess_gpu.py).
#include "seunet_common.h"

namespace seunet {

typedef unsigned long long u64;

// strong / weak masks, bit-packed along k.  One thread per (row, word).
__global__ void __launch_bounds__(256)
dti_pack_kernel(const double* __restrict__ pred, long long rows, int z, int nw, double hs, double ls, int f32,
                u64* __restrict__ strong, u64* __restrict__ weak) {
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx >= rows * nw) return;
  const long long row = idx / nw;
  const int m = (int)(idx % nw);
  const double* p = pred + row * z;
  u64 s = 0, w = 0;
  const int k1 = (m * 64 + 64 < z) ? m * 64 + 64 : z;
  for (int k = m * 64; k < k1; ++k) {
    double v = p[k] * 255.0;                       // (pred*255 in float64, prediction.py:19)
    if (f32) v = (double)(float)v;                 // np.array(pred*255, dtype=np.float32), train.py:31 / test.py:24
    const bool st = v >= hs;
    const bool wk = !st && v >= ls;                // pred < h*255 and pred >= l*255 (prediction.py:28)
    s |= (u64)st << (k & 63);
    w |= (u64)wk << (k & 63);
  }
  strong[idx] = s;
  weak[idx] = w;
}

// bits (k-1, k, k+1) of a neighbouring row, for word m
__device__ __forceinline__ u64 dti_dilate(const u64* __restrict__ g, int m, int nw) {
  const u64 x = g[m];
  const u64 xl = m > 0 ? g[m - 1] : 0ull, xr = m + 1 < nw ? g[m + 1] : 0ull;
  return x | (x << 1) | (xl >> 63) | (x >> 1) | (xr << 63);
}

// the raster sweep, one workgroup, one thread per row of the current wavefront
__global__ void __launch_bounds__(1024)
dti_sweep_kernel(u64* g, const u64* __restrict__ weak, int h, int w, int nw) {
  const int steps = 2 * (h - 1) + (w - 1) + 1;
  for (int t = 0; t < steps; ++t) {
    int i_lo = (t - (w - 1) + 1) / 2;
    if (t - (w - 1) < 0) i_lo = 0;
    const int i_hi = t / 2 < h - 1 ? t / 2 : h - 1;
    for (int i = i_lo + (int)threadIdx.x; i <= i_hi; i += (int)blockDim.x) {
      const int j = t - 2 * i;
      u64* row = g + ((long long)i * w + j) * nw;
      const u64* wrow = weak + ((long long)i * w + j) * nw;
      u64 carry = 0;
      for (int m = 0; m < nw; ++m) {
        u64 ext = 0;
#pragma unroll
        for (int di = -1; di <= 1; ++di)
#pragma unroll
          for (int dj = -1; dj <= 1; ++dj) {
            if (di == 0 && dj == 0) continue;
            const int ii = i + di, jj = j + dj;
            if (ii < 0 || ii >= h || jj < 0 || jj >= w) continue;
            ext |= dti_dilate(g + ((long long)ii * w + jj) * nw, m, nw);
          }
        const u64 s = row[m], wk = wrow[m];
        const u64 s_next = m + 1 < nw ? row[m + 1] : 0ull;             // k+1 of this row: not visited yet -> original
        const u64 a = s | (wk & (ext | (s >> 1) | (s_next << 63)));
        const u64 p = a | wk;
        const u64 sum = a + p;
        const u64 c1 = sum < a ? 1ull : 0ull;
        const u64 sum2 = sum + carry;
        const u64 c2 = sum2 < sum ? 1ull : 0ull;
        const u64 into = sum2 ^ a ^ p;                                  // carry INTO each bit
        carry = c1 | c2;                                                // carry out of the word = g[63] of this word
        row[m] = (into >> 1) | (carry << 63);                          // carry OUT of bit k = g[k]
      }
    }
    __syncthreads();   // (workgroup-scope release/acquire: the rows written above are visible to the next wavefront)
  }
}

__global__ void __launch_bounds__(256)
dti_unpack_kernel(const u64* __restrict__ g, long long rows, int z, int nw, unsigned char* __restrict__ out) {
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx >= rows * z) return;
  const long long row = idx / z;
  const int k = (int)(idx % z);
  out[idx] = (unsigned char)((g[row * nw + (k >> 6)] >> (k & 63)) & 1ull);
}

size_t dti_workspace_bytes(int h, int w, int z) {
  const long long nw = (z + 63) / 64;
  return (size_t)(2 * (long long)h * w * nw * 8);
}

int launch_dti(const double* pred, int h, int w, int z, double h_thresh, double l_thresh, int pred_dtype, unsigned char* out,
               void* workspace, size_t ws_bytes, hipStream_t s) {
  SEUNET_CHECK(pred && out && workspace && h >= 1 && w >= 1 && z >= 1, "dti: bad argument");
  SEUNET_CHECK(pred_dtype == 0 || pred_dtype == 1, "dti: pred_dtype %d (0 = float64 copy of prediction.py, 1 = float32 copy of train.py / test.py)", pred_dtype);
  SEUNET_CHECK(ws_bytes >= dti_workspace_bytes(h, w, z), "dti: workspace too small");
  const int nw = (z + 63) / 64;
  const long long rows = (long long)h * w;
  u64* g = reinterpret_cast<u64*>(workspace);
  u64* weak = g + rows * nw;
  double hs = h_thresh * 255.0, ls = l_thresh * 255.0;         // (h_thresh*255, l_thresh*255 in float64, prediction.py:20,28)
  if (pred_dtype == 1) { hs = (double)(float)hs; ls = (double)(float)ls; }   // weak python scalars adopt the array's float32
  dti_pack_kernel<<<(unsigned)((rows * nw + 255) / 256), 256, 0, s>>>(pred, rows, z, nw, hs, ls, pred_dtype, g, weak);
  dti_sweep_kernel<<<1, 1024, 0, s>>>(g, weak, h, w, nw);
  dti_unpack_kernel<<<(unsigned)((rows * z + 255) / 256), 256, 0, s>>>(g, rows, z, nw, out);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace seunet
