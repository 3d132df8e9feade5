// Largest 26-connected component, hole filling and the ATM'22 metric sums on the device.
//
// Reference (CPU, third-party code behind it):
//   util.py:58-75   maximum_3d: cc3d.connected_components(connectivity=26) -> component with the most voxels (ties: the
//                   HIGHEST label number, because `sorted(..., key=area)[::-1]` reverses a stable sort) -> if that
//                   component touches none of the slices z = Z//2, Z//3, Z//3*2 of the LAST axis, the second one ->
//                   scipy.ndimage.binary_fill_holes (background = 6-connected, holes = background not reaching the border)
//   train.py:749-757 evaluation_case: the same largest-component rule without the slice test and without hole filling;
//                   an empty prediction stays empty
//   metrics.py:14-78 sums over pred / label / skeleton and per-branch counts (bincount of skeleton * parsing)
// cc3d numbers components in order of first appearance in memory order, so "highest label number" = the component whose
// first voxel comes LAST in raster order = the largest minimum linear index.
//
// Algorithm: label-equivalence union-find (every voxel points to a smaller linear index of its component; roots point to
// themselves; a union is an atomicMin on the larger root), so the final label of a component IS its minimum linear index.
// Parents only ever decrease, which makes the structure robust on this chip's non-coherent L1/L2s: a stale parent read
// is still an ancestor, every link is made by a device-scope atomic on the true value, and the final compression runs
// in its own launch.  Rows are pre-linked per wavefront with ballots (a run of foreground voxels along z inside one
// 64-voxel wave span starts compressed), which removes the long serial chains along the contiguous axis.
// Integer / index work: results are bit-identical to the CPU reference (tests/test_components_gpu.py).
#include "seunet_common.h"
#include <algorithm>

namespace seunet {

typedef unsigned long long u64;

struct CcSel {            // device-side scalars of one call
  u64 best, second;       // (voxel count << 32) | root index ; 0 = none
  int touches;            // largest component has a voxel in one of the three test slices
  int chosen;             // root index of the selected component, -1 = none
  int status;             // 0 ok, 1 no component at all, 2 second component needed but absent
  int pad;
};

__device__ __forceinline__ int cc_find(const int* L, int i) {
  int p = L[i];
  while (p != i) { i = p; p = L[i]; }      // strictly decreasing chain: terminates even on stale reads
  return i;
}

__device__ __forceinline__ void cc_union(int* L, int a, int b) {
  bool done;
  do {
    a = cc_find(L, a);
    b = cc_find(L, b);
    if (a < b) { const int old = atomicMin(&L[b], a); done = old == b; b = old; }
    else if (b < a) { const int old = atomicMin(&L[a], b); done = old == a; a = old; }
    else done = true;
  } while (!done);
}

// L[i] = start of the z-run of voxel i inside its wave span (foreground), -1 (background).  INVERT labels the complement.
template <bool INVERT>
__global__ void __launch_bounds__(256)
cc_init_kernel(const unsigned char* __restrict__ vol, long long n, int Z, int* __restrict__ L) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool in = i < n;
  const bool fg = in && ((vol[in ? i : 0] != 0) != INVERT);
  const bool row_start = in && (i % Z) == 0;
  const u64 m = __ballot(fg), z0 = __ballot(row_start);
  // lane j starts a run if it is foreground and (j == 0 or lane j-1 is background or voxel j opens a row)
  const u64 starts = m & (~(m << 1) | z0 | 1ull);
  if (!in) return;
  if (!fg) { L[i] = -1; return; }
  const u64 below = starts & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
  const int s = 63 - __builtin_clzll(below);
  L[i] = (int)(i - lane + s);
}

// links to the neighbours that precede voxel i in raster order: 13 of the 26 (CONN26) or 3 of the 6 neighbours
template <bool CONN26>
__global__ void __launch_bounds__(256)
cc_merge_kernel(int* L, long long n, int H, int W, int Z) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  if (L[i] < 0) return;
  const int z = (int)(i % Z);
  const long long r = i / Z;
  const int y = (int)(r % W), x = (int)(r / W);
  // (0, 0, -1): already linked inside a wave span by cc_init_kernel, except across the span boundary
  if (z > 0 && (i & 63) == 0 && L[i - 1] >= 0) cc_union(L, (int)i, (int)(i - 1));
  if (CONN26) {
    for (int dx = -1; dx <= 0; ++dx)
      for (int dy = -1; dy <= (dx < 0 ? 1 : -1); ++dy) {
        const int xx = x + dx, yy = y + dy;
        if (xx < 0 || yy < 0 || yy >= W) continue;
        const long long base = ((long long)xx * W + yy) * Z;
        for (int dz = -1; dz <= 1; ++dz) {
          const int zz = z + dz;
          if (zz < 0 || zz >= Z) continue;
          if (L[base + zz] >= 0) cc_union(L, (int)i, (int)(base + zz));
        }
      }
  } else {
    if (y > 0 && L[i - Z] >= 0) cc_union(L, (int)i, (int)(i - Z));
    if (x > 0 && L[i - (long long)W * Z] >= 0) cc_union(L, (int)i, (int)(i - (long long)W * Z));
  }
}

__global__ void __launch_bounds__(256)
cc_compress_kernel(int* L, long long n) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  if (L[i] >= 0) L[i] = cc_find(L, (int)i);
}

// voxel count per root: one atomic per run of equal roots inside a wave
__global__ void __launch_bounds__(256)
cc_count_kernel(const int* __restrict__ L, long long n, unsigned int* __restrict__ cnt) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int r = i < n ? L[i] : -1;
  const int prev = dpp_settle(__shfl_up(r, 1, 64));
  const bool lead = r >= 0 && (lane == 0 || prev != r);
  const u64 leaders = __ballot(lead), fg = __ballot(r >= 0);
  if (!lead) return;
  // run = consecutive lanes from `lane` with foreground and no new leader
  const u64 after = (lane == 63) ? 0ull : ((leaders | ~fg) >> (lane + 1));
  const int len = after ? __builtin_ctzll(after) + 1 : 64 - lane;
  atomicAdd(&cnt[r], (unsigned int)len);
}

// pass 0: best = max key over roots; pass 1: second = max key over roots other than best
__global__ void __launch_bounds__(256)
cc_select_kernel(const int* __restrict__ L, const unsigned int* __restrict__ cnt, long long n, int pass, CcSel* sel) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  u64 key = 0;
  if (i < n && L[i] == (int)i) {
    key = ((u64)cnt[i] << 32) | (u64)(unsigned int)i;
    if (pass == 1 && key == sel->best) key = 0;
  }
  // wave maximum first: one atomic per wave
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const u64 o = shfl_xor_settled(key, off);
    key = o > key ? o : key;
  }
  if ((threadIdx.x & 63) == 0 && key) atomicMax(pass == 0 ? &sel->best : &sel->second, key);
}

// util.py:66-70: does the largest component reach z = Z//2, Z//3 or Z//3*2 (last axis)?
__global__ void __launch_bounds__(256)
cc_slices_kernel(const int* __restrict__ L, long long rows, int Z, CcSel* sel) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= rows * 3 || sel->best == 0) return;
  const int which = (int)(i / rows);
  const long long row = i % rows;
  const int z = which == 0 ? Z / 2 : (which == 1 ? Z / 3 : Z / 3 * 2);
  if (L[row * Z + z] == (int)(unsigned int)(sel->best & 0xffffffffull)) atomicOr(&sel->touches, 1);
}

__global__ void cc_choose_kernel(CcSel* sel, int rule) {
  if (threadIdx.x || blockIdx.x) return;
  sel->status = 0;
  if (sel->best == 0) { sel->chosen = -1; sel->status = 1; return; }
  sel->chosen = (int)(unsigned int)(sel->best & 0xffffffffull);
  if (rule == 1 && !sel->touches) {
    if (sel->second == 0) { sel->status = 2; return; }
    sel->chosen = (int)(unsigned int)(sel->second & 0xffffffffull);
  }
}

__global__ void __launch_bounds__(256)
cc_mask_kernel(const int* __restrict__ L, long long n, const CcSel* __restrict__ sel, unsigned char* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  out[i] = (sel->chosen >= 0 && L[i] == sel->chosen) ? 1 : 0;
}

// background components that reach the border of the volume (binary_fill_holes: they are NOT holes)
__global__ void __launch_bounds__(256)
cc_border_kernel(const int* __restrict__ L, int H, int W, int Z, unsigned int* __restrict__ flag) {
  const long long n = (long long)H * W * Z;
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % Z);
  const long long r = i / Z;
  const int y = (int)(r % W), x = (int)(r / W);
  if (!(x == 0 || x == H - 1 || y == 0 || y == W - 1 || z == 0 || z == Z - 1)) return;
  if (L[i] >= 0) flag[L[i]] = 1u;
}

__global__ void __launch_bounds__(256)
cc_fill_kernel(const int* __restrict__ L, const unsigned int* __restrict__ flag, long long n, unsigned char* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  if (L[i] >= 0 && flag[L[i]] == 0u) out[i] = 1;      // enclosed background becomes foreground
}

size_t cc_workspace_bytes(int H, int W, int Z) {
  const size_t n = (size_t)H * W * Z;
  return align_up(n * 4, 256) * 2 + 256;               // labels, counts / border flags, CcSel
}

int launch_largest_component(const unsigned char* vol, int H, int W, int Z, int rule, unsigned char* out, int* status_dev,
                             void* workspace, size_t ws_bytes, hipStream_t s) {
  SEUNET_CHECK(vol && out && workspace && H >= 1 && W >= 1 && Z >= 1, "largest_component: bad argument");
  SEUNET_CHECK(rule == 0 || rule == 1, "largest_component: rule %d (0 = evaluation_case, train.py:749-757; 1 = maximum_3d, util.py:58-75)", rule);
  const long long n = (long long)H * W * Z;
  SEUNET_CHECK(n < (1ll << 31), "largest_component: %lld voxels exceed the 32-bit label range", n);
  SEUNET_CHECK(ws_bytes >= cc_workspace_bytes(H, W, Z), "largest_component: workspace too small");
  unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
  int* L = reinterpret_cast<int*>(ws);
  unsigned int* cnt = reinterpret_cast<unsigned int*>(ws + align_up((size_t)n * 4, 256));
  CcSel* sel = reinterpret_cast<CcSel*>(ws + 2 * align_up((size_t)n * 4, 256));
  const unsigned blocks = (unsigned)((n + 255) / 256);
  SEUNET_HIP(hipMemsetAsync(cnt, 0, (size_t)n * 4, s));
  SEUNET_HIP(hipMemsetAsync(sel, 0, sizeof(CcSel), s));
  cc_init_kernel<false><<<blocks, 256, 0, s>>>(vol, n, Z, L);
  cc_merge_kernel<true><<<blocks, 256, 0, s>>>(L, n, H, W, Z);
  cc_compress_kernel<<<blocks, 256, 0, s>>>(L, n);
  cc_count_kernel<<<blocks, 256, 0, s>>>(L, n, cnt);
  cc_select_kernel<<<blocks, 256, 0, s>>>(L, cnt, n, 0, sel);
  if (rule == 1) {
    cc_select_kernel<<<blocks, 256, 0, s>>>(L, cnt, n, 1, sel);
    const long long rows = (long long)H * W;
    cc_slices_kernel<<<(unsigned)((rows * 3 + 255) / 256), 256, 0, s>>>(L, rows, Z, sel);
  }
  cc_choose_kernel<<<1, 64, 0, s>>>(sel, rule);
  cc_mask_kernel<<<blocks, 256, 0, s>>>(L, n, sel, out);
  if (rule == 1) {   // binary_fill_holes (util.py:73): label the complement with 6-connectivity, keep what reaches the border
    SEUNET_HIP(hipMemsetAsync(cnt, 0, (size_t)n * 4, s));
    cc_init_kernel<true><<<blocks, 256, 0, s>>>(out, n, Z, L);
    cc_merge_kernel<false><<<blocks, 256, 0, s>>>(L, n, H, W, Z);
    cc_compress_kernel<<<blocks, 256, 0, s>>>(L, n);
    cc_border_kernel<<<blocks, 256, 0, s>>>(L, H, W, Z, cnt);
    cc_fill_kernel<<<blocks, 256, 0, s>>>(L, cnt, n, out);
  }
  if (status_dev) SEUNET_HIP(hipMemcpyAsync(status_dev, &sel->status, sizeof(int), hipMemcpyDeviceToDevice, s));
  SEUNET_LAUNCH_CHECK();
  return 0;
}

// ---- metrics.py:14-78: the sums every ATM'22 metric is made of ------------------------------------------------------
//   sums[0] = sum(pred*label)  [1] = sum(pred)  [2] = sum(label)  [3] = sum(pred*skeleton)  [4] = sum(skeleton)
//   branch_label[id] = #{skeleton*parsing == id},  branch_pred[id] = #{skeleton*parsing*pred == id},  id < nbins
//   (np.bincount of metrics.py:16-22; id 0 is counted too and dropped by the caller's [1:]); max_id = largest id seen
// pred / label / skeleton are used as 0/1 masks multiplied together exactly like the reference's uint8 products, i.e. the
// caller passes arrays whose values are 0 or 1 (large_cd, the mask, `skeleton > 0`: train.py:755,763-764).
__global__ void __launch_bounds__(256)
metric_sums_kernel(const unsigned char* __restrict__ pred, const unsigned char* __restrict__ label, const unsigned char* __restrict__ skel,
                   const int* __restrict__ parsing, long long n, int nbins, u64* __restrict__ sums, unsigned int* __restrict__ branch_label,
                   unsigned int* __restrict__ branch_pred, int* __restrict__ max_id, int* __restrict__ overflow) {
  u64 s[5] = {0, 0, 0, 0, 0};
  int mx = 0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const unsigned p = pred[i], l = label ? label[i] : 0u, k = skel ? skel[i] : 0u;
    s[0] += p * l; s[1] += p; s[2] += l; s[3] += p * k; s[4] += k;
    if (parsing && k) {
      const long long id = (long long)k * parsing[i];
      if (id < 0 || id >= nbins) { *overflow = 1; continue; }
      atomicAdd(&branch_label[id], 1u);
      const long long pid = id * p;
      atomicAdd(&branch_pred[pid < nbins ? pid : 0], 1u);
      mx = id > mx ? (int)id : mx;
    }
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    u64 v = s[q];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += shfl_xor_settled(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sums[q], v);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { const int o = shfl_xor_settled(mx, off); mx = o > mx ? o : mx; }
  if ((threadIdx.x & 63) == 0 && mx) atomicMax(max_id, mx);
}

// out: 8 u64 (5 sums used), then nbins u32 label-branch counts, then nbins u32 pred-branch counts, then int max_id, int overflow
size_t metric_out_bytes(int nbins) { return 8 * 8 + (size_t)nbins * 4 * 2 + 16; }

int launch_metric_sums(const unsigned char* pred, const unsigned char* label, const unsigned char* skel, const int* parsing, long long n,
                       int nbins, void* out, size_t out_bytes, hipStream_t s) {
  SEUNET_CHECK(pred && out && n >= 1 && nbins >= 1, "metric_sums: bad argument");
  SEUNET_CHECK(!parsing || skel, "metric_sums: the branch counts need the skeleton (metrics.py:15)");
  SEUNET_CHECK(out_bytes >= metric_out_bytes(nbins), "metric_sums: output buffer too small");
  SEUNET_HIP(hipMemsetAsync(out, 0, metric_out_bytes(nbins), s));
  u64* sums = reinterpret_cast<u64*>(out);
  unsigned int* bl = reinterpret_cast<unsigned int*>(sums + 8);
  unsigned int* bp = bl + nbins;
  int* extra = reinterpret_cast<int*>(bp + nbins);          // [0] max id, [1] overflow
  const unsigned blocks = (unsigned)std::min<long long>((n + 255) / 256, 2048);
  metric_sums_kernel<<<blocks, 256, 0, s>>>(pred, label, skel, parsing, n, nbins, sums, bl, bp, extra, extra + 1);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace seunet
