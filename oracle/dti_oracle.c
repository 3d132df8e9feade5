/* CPU oracle for double_threshold_iteration.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference loop, statement by statement.  The reference has THREE copies of the function that
 * differ in one line: prediction.py:19 keeps pred*255 in float64; train.py:31 and test.py:24 (validation / test) round it to
 * float32 (`np.array(pred*255, dtype=np.float32)`), after which numpy (NEP 50, requirements.txt pins numpy 2.1.3) compares
 * in float32 with the thresholds h*255, l*255 rounded to float32 -- voxels within a float32 ulp of a threshold classify
 * differently.  `f32` selects the variant.  Then: bin = pred >= h*255, gbin = copy of bin, ONE raster-order in-place sweep
 * (the `while` compares gbin_pre.all() with gbin.all() after `gbin_pre = gbin` aliased the array, so its body runs
 * exactly once -- SURVEY Q11); a weak voxel (gbin == 0, l*255 <= pred < h*255) is switched on when any of the 26
 * neighbours, indices clamped to the volume (prediction.py:33), is non-zero at that moment.
 * Pinned by tests/golden/dti_known.npz, produced by running the reference's own function (oracle/make_golden_dti.py).
 * Built by oracle/Makefile into oracle/libdti_oracle.so (git-ignored); only tests/ load it. */
#include <stdlib.h>

static const int NEIGB[26][3] = {   /* prediction.py:14-17, same order */
    {-1, -1, 0}, {-1, 0, 0}, {-1, 1, 0}, {0, -1, 0}, {0, 1, 0}, {1, -1, 0}, {1, 0, 0}, {1, 1, 0}, {-1, -1, -1},
    {-1, 0, -1}, {-1, 1, -1}, {0, -1, -1}, {0, 0, -1}, {0, 1, -1}, {1, -1, -1}, {1, 0, -1}, {1, 1, -1},
    {-1, -1, 1}, {-1, 0, 1}, {-1, 1, 1}, {0, -1, 1}, {0, 0, 1}, {0, 1, 1}, {1, -1, 1}, {1, 0, 1}, {1, 1, 1}};

static int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

/* pred: h*w*z float64, out: h*w*z bytes (1 where the reference returns 1.0).  Returns 0, or 1 on allocation failure. */
int dti_oracle(const double* pred, int h, int w, int z, double h_thresh, double l_thresh, int f32, unsigned char* out) {
  const long long n = (long long)h * w * z;
  double* p = (double*)malloc((size_t)n * sizeof(double));
  if (!p) return 1;
  double hs = h_thresh * 255, ls = l_thresh * 255;
  if (f32) { hs = (double)(float)hs; ls = (double)(float)ls; }   /* weak python scalars take the array's float32 */
  for (long long i = 0; i < n; ++i) {
    p[i] = pred[i] * 255;
    if (f32) p[i] = (double)(float)p[i];                         /* train.py:31 / test.py:24 */
    out[i] = p[i] >= hs ? 1 : 0;
  }
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < w; ++j)
      for (int k = 0; k < z; ++k) {
        const long long c = ((long long)i * w + j) * z + k;
        if (out[c] == 0 && p[c] < hs && p[c] >= ls) {
          for (int q = 0; q < 26; ++q) {
            const int ii = clampi(i + NEIGB[q][0], h - 1), jj = clampi(j + NEIGB[q][1], w - 1), kk = clampi(k + NEIGB[q][2], z - 1);
            if (out[((long long)ii * w + jj) * z + kk]) { out[c] = 1; break; }
          }
        }
      }
  free(p);
  return 0;
}
