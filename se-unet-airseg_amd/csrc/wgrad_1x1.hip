// Weight gradient of the 1x1x1 aggregation convolutions (backward of nn.Conv3d(kernel_size=1) at reference SE_UNet.py:42:
// ec63, ec93, dc22, dc42), 16-bit storage:   dW[co][ci] = sum over all voxels v of  X[v][ci] * dY[v][co]
//
// Why a kernel of its own: a 1x1x1 weight gradient has no spatial structure -- it is ONE GEMM with K = all voxels -- and is bound
// by reading X and dY once.  The tiled kernel (wgrad.hip) splits the (ci, co) plane into 32 x 32 combos that each stream their
// own slice of both tensors: for 128 -> 64 channels X is read twice and dY four times, in 64-byte slices of 256-byte voxel
// records (0.169 ms for ec63 against 0.07 ms of HBM time).  Here a workgroup owns ALL (ci, co) pairs and walks chunks of 64 /
// 128 consecutive voxels of the flat [N x D x H x W] index:
//   * the chunk of every source tensor and of dY arrives by LDS-DMA as one contiguous copy (voxel-major records, the 16-byte
//     pieces XOR-swizzled by voxel on the source side so that the transposing reads are conflict-free), two chunks in flight;
//   * the four waves split the 16-channel input blocks, every wave pairs its blocks with all output blocks:
//     v_mfma_f32_16x16x32, A = dY^T (16 channels x 32 voxels), B = X (32 voxels x 16 channels), both through
//     ds_read_b64_tr_b16; accumulators (<= 24 pairs per wave) live in registers for the whole launch;
//   * one slab per workgroup, fixed-order f64 slab sum (deterministic, no atomics).
#include "seunet_common.h"
#include <utility>
#include <type_traits>
#include <cstdlib>

namespace seunet {

typedef bf16_t w1b16x4 __attribute__((ext_vector_type(4)));
typedef bf16_t w1b16x8 __attribute__((ext_vector_type(8)));
typedef f16_t w1f16x8 __attribute__((ext_vector_type(8)));
typedef float w1f32x4 __attribute__((ext_vector_type(4)));

struct W1Args {
  const void* src[3]; int srcC[3]; int nsrc;      // X: up to three tensors of 32 or 64 channels each
  const void* dy; int cout;
  float* slab; const void* zero;
  long long nvox;                                  // N * D * H * W
  int nchunk;
  int nst;                                         // LDS stages (2..4): nst - 1 chunks in flight
};

static constexpr int W1_NW = 4;

__device__ __forceinline__ void w1_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void w1_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// piece permutation of voxel v in a record of np 16-byte pieces (np = 4, 8, 16): a transposing read takes, per 16 lanes, 4
// consecutive voxels x two adjacent pieces; 32 lanes = voxels v..v+3 and v+8..v+11 (see wgrad_march.hip).  XOR on the pair index.
__device__ __forceinline__ int w1_swz(int np, int v) {
  if (np == 4) return ((v >> 3) & 1) << 1;
  if (np == 8) return (((v >> 1) & 1) | (((v >> 3) & 1) << 1)) << 1;
  return ((v & 3) | (((v >> 3) & 1) << 2)) << 1;
}

template <typename T> __device__ __forceinline__ w1f32x4 w1_mfma(w1b16x8 a, w1b16x8 b, w1f32x4 c) {
  if constexpr (std::is_same<T, f16_t>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(w1f16x8, a), __builtin_bit_cast(w1f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ w1b16x8 w1_frag(unsigned addr0, unsigned addr1) {
  typedef __attribute__((address_space(3))) w1b16x4 lds_b4;
  const w1b16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(size_t)addr0);
  const w1b16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(size_t)addr1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// CIBW: 16-channel input blocks per wave; NCOB: 16-channel output blocks (all of them, every wave); CH: voxels per chunk
template <typename T, int CIBW, int NCOB, int CH>
__global__ void __launch_bounds__(256)
wgrad_1x1_kernel(W1Args a) {
  constexpr int KS = CH / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int cin = a.srcC[0] + (a.nsrc > 1 ? a.srcC[1] : 0) + (a.nsrc > 2 ? a.srcC[2] : 0);
  const int stage = (cin + a.cout) * 2 * CH;                 // one stage: the X regions of the sources back to back, then dY
  const int dump = a.nst * stage;                            // 1 KB landing area of the padding DMA instructions
  // region offsets of the sources inside a stage
  int roff[3];
  roff[0] = 0; roff[1] = a.srcC[0] * 2 * CH; roff[2] = roff[1] + (a.nsrc > 1 ? a.srcC[1] : 0) * 2 * CH;
  const int yoff0 = cin * 2 * CH;

  // ---- fragment addressing (per lane, chunk-invariant): X block c of this wave / dY block k, K-step ks, half r ----
  unsigned xoff[CIBW][KS][2], yoff[NCOB][KS][2];
#pragma unroll
  for (int c = 0; c < CIBW; ++c) {
    const int ch0 = (wave * CIBW + c) * 16;                  // first channel of the block in the concatenation
    int s = 0, cl = ch0;
    if (a.nsrc > 1 && cl >= a.srcC[0]) { cl -= a.srcC[0]; s = 1; if (a.nsrc > 2 && cl >= a.srcC[1]) { cl -= a.srcC[1]; s = 2; } }
    const int C = a.srcC[s], np = C / 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int v = 32 * ks + 8 * grp + 4 * r + q;
        xoff[c][ks][r] = (unsigned)(roff[s] + v * C * 2 + (((cl / 8 + (p >> 1)) ^ w1_swz(np, v)) * 16) + (p & 1) * 8);
      }
  }
  {
    const int npy = a.cout / 8;
#pragma unroll
    for (int k = 0; k < NCOB; ++k)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int v = 32 * ks + 8 * grp + 4 * r + q;
          yoff[k][ks][r] = (unsigned)(yoff0 + v * a.cout * 2 + (((k * 2 + (p >> 1)) ^ w1_swz(npy, v)) * 16) + (p & 1) * 8);
        }
  }

  // ---- DMA plan: instruction number id = wave + 4 * it of a stage image (1 KB each).  Per lane and instruction one packed word:
  //      bits 0-1 tensor (0..2 sources, 3 = dY), bits 4-19 byte offset of the 16-byte piece inside the tensor's chunk (pieces
  //      permuted), bits 20-27 voxel inside the chunk; ~0 = padding.  Every wave issues exactly LW instructions per chunk
  //      (LW = 8, 16 or 24 >= the real count: the vmcnt literal of the loop), the surplus ones copy the zero page to a dump. ----
  const int ni = stage / 1024;                               // (regions are multiples of 1 KB: channels % 32 == 0, CH >= 64)
  const int items = (ni + W1_NW - 1) / W1_NW;
  const int LW = items <= 8 ? 8 : (items <= 16 ? 16 : 24);   // (uniform)
  constexpr int MAXI = 24;
  unsigned plan[MAXI];
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int id = wave + W1_NW * it;
    const int byte = id * 1024 + lane * 16;
    int sel, C, rel;
    if (byte >= yoff0) { sel = 3; C = a.cout; rel = byte - yoff0; }
    else if (a.nsrc > 2 && byte >= roff[2]) { sel = 2; C = a.srcC[2]; rel = byte - roff[2]; }
    else if (a.nsrc > 1 && byte >= roff[1]) { sel = 1; C = a.srcC[1]; rel = byte - roff[1]; }
    else { sel = 0; C = a.srcC[0]; rel = byte; }
    const int sh = 31 - __clz(C * 2);                        // records are 64, 128 or 256 bytes
    const int v = rel >> sh, slot = (rel & (C * 2 - 1)) >> 4;
    const int piece = slot ^ w1_swz(C / 8, v);
    plan[it] = (it < items && id < ni) ? ((unsigned)sel | ((unsigned)((v << sh) + piece * 16) << 4) | ((unsigned)v << 20)) : 0xFFFFFFFFu;
  }
  const unsigned char* zero_page = reinterpret_cast<const unsigned char*>(a.zero) + lane * 16;
  const unsigned char* tb[4] = {reinterpret_cast<const unsigned char*>(a.src[0]), reinterpret_cast<const unsigned char*>(a.nsrc > 1 ? a.src[1] : a.src[0]),
                                reinterpret_cast<const unsigned char*>(a.nsrc > 2 ? a.src[2] : a.src[0]), reinterpret_cast<const unsigned char*>(a.dy)};
  const long long tstride[4] = {(long long)CH * a.srcC[0] * 2, (long long)CH * a.srcC[1] * 2, (long long)CH * a.srcC[2] * 2, (long long)CH * a.cout * 2};
  auto issue = [&](int chunk, int st) __attribute__((always_inline)) {
    const bool cok = chunk < a.nchunk;                       // wave-uniform
    const long long left = a.nvox - (long long)chunk * CH;   // voxels of the volume from this chunk on
    const unsigned char* cb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) cb[t] = tb[t] + (long long)(cok ? chunk : 0) * tstride[t];
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
      if (it < LW) {                                         // (uniform)
        const unsigned w = plan[it];
        const int sel = w & 3, v = (w >> 20) & 0xFF;
        const unsigned char* base = sel == 0 ? cb[0] : (sel == 1 ? cb[1] : (sel == 2 ? cb[2] : cb[3]));
        const bool ok = cok && w != 0xFFFFFFFFu && v < left;
        const unsigned char* gp = ok ? base + ((w >> 4) & 0xFFFF) : zero_page;
        const int id = wave + W1_NW * it;
        w1_dma16(gp, (it < items && id < ni) ? lds_base + (unsigned)(st * stage + id * 1024) : lds_base + (unsigned)dump);
      }
    }
  };

  w1f32x4 acc[CIBW][NCOB];
#pragma unroll
  for (int c = 0; c < CIBW; ++c)
#pragma unroll
    for (int k = 0; k < NCOB; ++k) acc[c][k] = w1f32x4{0.f, 0.f, 0.f, 0.f};

  // chunks blockIdx.x, blockIdx.x + gridDim.x, ...; PF = nst - 1 chunks are in flight ahead of the one being multiplied (the
  // launch is bound by HBM latency x bytes in flight: two stages of 24 KB per CU were 3.6 TB/s)
  const int nst = a.nst, PF = nst - 1;
  int chunk = blockIdx.x, st = 0, stp = 0;                    // stage of the current chunk / of the next one to issue
  for (int k = 0; k < PF; ++k) { issue(chunk + k * (int)gridDim.x, stp); stp = stp + 1 == nst ? 0 : stp + 1; }
  for (; chunk < a.nchunk; chunk += gridDim.x) {
    issue(chunk + PF * (int)gridDim.x, stp);                 // (beyond the last chunk: padding instructions, same count)
    stp = stp + 1 == nst ? 0 : stp + 1;
    // the current chunk's LW instructions are older than the PF * LW issued after them
    switch (PF * LW) {
      case 8: w1_wait_vm<8>(); break;
      case 16: w1_wait_vm<16>(); break;
      case 24: w1_wait_vm<24>(); break;
      case 32: w1_wait_vm<32>(); break;
      case 48: w1_wait_vm<48>(); break;
      default: w1_wait_vm<0>(); break;                       // (72: beyond the 6-bit counter; the launcher never asks for it)
    }
    __builtin_amdgcn_s_barrier();
    const unsigned sb = lds_base + (unsigned)(st * stage);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      w1b16x8 xf[CIBW], yf[NCOB];
#pragma unroll
      for (int c = 0; c < CIBW; ++c) xf[c] = w1_frag(sb + xoff[c][ks][0], sb + xoff[c][ks][1]);
#pragma unroll
      for (int k = 0; k < NCOB; ++k) yf[k] = w1_frag(sb + yoff[k][ks][0], sb + yoff[k][ks][1]);
#pragma unroll
      for (int c = 0; c < CIBW; ++c)
#pragma unroll
        for (int k = 0; k < NCOB; ++k) acc[c][k] = w1_mfma<T>(yf[k], xf[c], acc[c][k]);
    }
    __builtin_amdgcn_s_barrier();                            // every wave is done reading this stage before it is refilled
    st = st + 1 == nst ? 0 : st + 1;
  }
  w1_wait_vm<0>();

  // ---- slab of this workgroup: [wave][c][k][lane][4] ----
  float* out = a.slab + ((size_t)blockIdx.x * W1_NW + wave) * (CIBW * NCOB * 256) + lane * 4;
#pragma unroll
  for (int c = 0; c < CIBW; ++c)
#pragma unroll
    for (int k = 0; k < NCOB; ++k) *reinterpret_cast<w1f32x4*>(out + (c * NCOB + k) * 256) = acc[c][k];
}

// sum of the slabs, 16-way parallel in a fixed order, f64 -> dw (cout, cin)
__global__ void __launch_bounds__(256)
wgrad_1x1_reduce_kernel(const float* __restrict__ slab, int nslab, int cibw, int ncob, int cin, float* __restrict__ dw) {
  const int per = W1_NW * cibw * ncob * 256;
  const int el = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  const float* ptr = slab + e;
  double s0 = 0.0, s1 = 0.0;
  int k = part;
  for (; k + 16 < nslab; k += 32) {
    s0 += (double)ptr[(size_t)k * per];
    s1 += (double)ptr[(size_t)(k + 16) * per];
  }
  if (k < nslab) s0 += (double)ptr[(size_t)k * per];
  __shared__ double red[16][17];
  red[part][el] = s0 + s1;
  __syncthreads();
  if (part == 0) {
    double t = red[0][el];
#pragma unroll
    for (int j = 1; j < 16; ++j) t += red[j][el];
    const int comp = e & 3, lane = (e >> 2) & 63, pair = e >> 8;
    const int kk = pair % ncob, c = (pair / ncob) % cibw, wave = pair / (ncob * cibw);
    const int co = kk * 16 + 4 * (lane >> 4) + comp;
    const int ci = (wave * cibw + c) * 16 + (lane & 15);
    if (ci < cin) dw[(size_t)co * cin + ci] = (float)t;
  }
}

struct W1Cfg { int cibw, ncob, ch, nst; };
static bool wgrad_1x1_cfg(int dtype, const SrcList& x, int cin_logical, int cout, Dims d, bool size_gate, W1Cfg& c) {
  if (dtype_size(dtype) != 2 || x.n < 1 || x.n > 3 || cin_logical != x.total()) return false;
  for (int i = 0; i < x.n; ++i) if (x.C[i] != 32 && x.C[i] != 64) return false;
  if (cout != 32 && cout != 64 && cout != 128) return false;
  const int ncib = cin_logical / 16;
  c.cibw = (ncib + W1_NW - 1) / W1_NW;
  c.ncob = cout / 16;
  if (ncib != W1_NW * c.cibw || c.cibw * c.ncob > 24) return false;
  if (!((c.cibw == 1 && c.ncob == 2) || (c.cibw == 2 && c.ncob == 4) || (c.cibw == 3 && c.ncob == 8))) return false;   // the instantiated forms
  // chunk size and stages: as many bytes in flight as the LDS holds (<= 4 stages; the vmcnt counter is 6 bits: (nst - 1) x
  // padded instructions per chunk <= 48)
  c.ch = 128; c.nst = 0;
  for (int ch = 128; ch >= 64 && c.nst < 3; ch /= 2) {
    const int stage = (cin_logical + cout) * 2 * ch;
    const int items = (stage / 1024 + W1_NW - 1) / W1_NW, lw = items <= 8 ? 8 : (items <= 16 ? 16 : 24);
    if (items > 24) continue;
    int nst = (159 * 1024) / stage;
    if (nst > 4) nst = 4;
    while (nst > 2 && (nst - 1) * lw > 48) --nst;
    if (nst >= 2 && nst > c.nst) { c.nst = nst; c.ch = ch; }
  }
  if (c.nst < 2) return false;
  if (size_gate) {
    static const bool off = std::getenv("SEUNET_NO_WGRAD_1X1") != nullptr;   // (diagnostic switch for A/B timing)
    // where it beats the tiled kernel (isolated launches): many (ci, co) combos there, i.e. many re-reads -- ec63 (8 combos, 1 M
    // voxels) 0.104 vs 0.165 ms, ec93 (24 combos, 131 k voxels) 0.051 vs 0.085 ms; not dc42 (2 combos: 0.059 vs 0.039 ms) nor
    // dc22 (8 combos but 131 k voxels: 0.033 vs 0.023 ms, four chunks per workgroup do not amortise the pipeline fill)
    const int combos = cdiv(cin_logical, 32) * cdiv(cout, 32);
    const long long nv = (long long)d.N * d.vox();
    static const bool all = std::getenv("SEUNET_WGRAD_1X1_ALL") != nullptr;   // (diagnostic: every layer the kernel serves)
    if (off || !(all || combos >= 16 || (combos >= 8 && nv >= 500000))) return false;
  }
  return true;
}
bool wgrad_1x1_supported(int dtype, const SrcList& x, int cin_logical, int cout, Dims d) {
  W1Cfg c;
  return wgrad_1x1_cfg(dtype, x, cin_logical, cout, d, true, c);
}

template <typename T, int CIBW, int NCOB, int CH>
static int wgrad_1x1_launch(const W1Args& a, int grid, int lds, hipStream_t s) {
  static unsigned long long configured = 0;
  if (int e = configure_kernel_lds(configured, reinterpret_cast<const void*>(&wgrad_1x1_kernel<T, CIBW, NCOB, CH>), lds)) return e;
  wgrad_1x1_kernel<T, CIBW, NCOB, CH><<<grid, 256, lds, s>>>(a);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_wgrad_1x1(int dtype, const SrcList& x, int cin_logical, const void* dy, int cout, float* dw, void* workspace,
                     size_t ws_bytes, Dims d, hipStream_t s) {
  W1Cfg c;
  SEUNET_CHECK(wgrad_1x1_cfg(dtype, x, cin_logical, cout, d, false, c),
               "wgrad_1x1: 16-bit tensors, 1..3 sources of 32 or 64 channels (64, 128 or 192 together), 32 / 64 / 128 output channels only");
  SEUNET_CHECK(ws_bytes >= 256, "wgrad_1x1: workspace too small");
  W1Args a{};
  for (int i = 0; i < 3; ++i) { a.src[i] = i < x.n ? x.ptr[i] : nullptr; a.srcC[i] = i < x.n ? x.C[i] : 0; }
  a.nsrc = x.n;
  a.dy = dy; a.cout = cout;
  a.zero = device_zero_page();
  SEUNET_CHECK(a.zero != nullptr, "wgrad_1x1: cannot allocate the device zero page");
  a.slab = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + 256);
  a.nvox = (long long)d.N * d.vox();
  a.nchunk = (int)((a.nvox + c.ch - 1) / c.ch);
  const int grid = a.nchunk < 256 ? a.nchunk : 256;
  const size_t per = (size_t)W1_NW * c.cibw * c.ncob * 256;
  SEUNET_CHECK(ws_bytes >= 256 + (size_t)grid * per * sizeof(float), "wgrad_1x1: workspace too small");
  const int lds = (cin_logical + cout) * 2 * c.ch * c.nst + 1024;
  a.nst = c.nst;
  int e = -1;
  SEUNET_DTYPE_SWITCH(dtype, if constexpr (sizeof(T) == 2) {
    if (c.cibw == 1 && c.ch == 128) e = wgrad_1x1_launch<T, 1, 2, 128>(a, grid, lds, s);
    else if (c.cibw == 1) e = wgrad_1x1_launch<T, 1, 2, 64>(a, grid, lds, s);
    else if (c.cibw == 2 && c.ch == 128) e = wgrad_1x1_launch<T, 2, 4, 128>(a, grid, lds, s);
    else if (c.cibw == 2) e = wgrad_1x1_launch<T, 2, 4, 64>(a, grid, lds, s);
    else if (c.ch == 128) e = wgrad_1x1_launch<T, 3, 8, 128>(a, grid, lds, s);
    else e = wgrad_1x1_launch<T, 3, 8, 64>(a, grid, lds, s);
  });
  if (e) return e;
  wgrad_1x1_reduce_kernel<<<(unsigned)(per / 16), 256, 0, s>>>(a.slab, grid, c.cibw, c.ncob, cin_logical, dw);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace seunet
