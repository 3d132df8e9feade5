// Marching weight gradient of the 3x3x3 convolutions of the two fine levels (backward of nn.Conv3d at reference
// SE_UNet.py:57 / :15: ec4..ec6, dc3, dc4, dc5), dilation 1 or 2, 16-bit storage.
//   dW[co][ci][tap] = sum over (n, voxel v) of  X[n][v + off(tap)][ci] * dY[n][v][co]
//
// Why a second weight-gradient kernel: the tiled one (wgrad.hip) re-reads both operands from the LDS for every tap -- 0.57 KB
// of fragment reads per MFMA, which binds it to the LDS pipe on the wide layers exactly like the tiled convolution was.  Here
// the structure of conv_march.hip is applied to the weight gradient:
//   * a workgroup (4 waves, ONE per SIMD, the whole 512-entry register file each) owns a 4 x 32 (y, x) patch and marches
//     along z, x-stationary: step s stages ONE plane of X and pairs it with the dY planes s, s-1, s-2 (dz = -1, 0, +1);
//   * ALL 27 tap accumulators of a (16 output channels x 16 input channels) pair live in registers for the whole march and
//     across marches (v_mfma_f32_16x16x32: A = dY^T 16 channels x 32 voxels, B = X 32 voxels x 16 channels); a wave owns one or
//     two pairs (108 / 216 accumulator registers), the four waves cover 64 x 32, 32 x 64 or 32 x 32 channels;
//   * both operands need "8 voxels of one channel" per lane while the planes are voxel-major: ds_read_b64_tr_b16 delivers that
//     from the unmodified DMA image; an X row fragment (3 x-taps) feeds up to 9 x PW MFMAs, a dY row fragment 9 (3 dy x 3 dx
//     of its dz): 0.19 KB of LDS reads per MFMA;
//   * planes arrive by LDS-DMA two steps ahead (X: 3-slot ring, dY: 5-slot ring -- a dY plane is used by three steps),
//     counted vmcnt, one raw s_barrier per step; 16-byte pieces XOR-swizzled by x on the DMA source side so that the
//     transposing reads touch every bank once; a dY plane outside the march is a zero plane in the LDS (no branches);
//   * workgroups are persistent over their (sample, patch, z-segment) items; one slab of accumulators per workgroup, summed
//     by a second kernel in a fixed order in f64 (deterministic, no atomics) into the PyTorch (Cout, Cin, 3, 3, 3) layout.
#include "seunet_common.h"
#include <utility>
#include <type_traits>
#include <cstdlib>

namespace seunet {

typedef bf16_t wmb16x4 __attribute__((ext_vector_type(4)));
typedef bf16_t wmb16x8 __attribute__((ext_vector_type(8)));
typedef f16_t wmf16x8 __attribute__((ext_vector_type(8)));
typedef float wmf32x4 __attribute__((ext_vector_type(4)));

struct WmArgs {
  const void* src0; const void* src1;   // X: one or two tensors of srcC channels each (virtual concatenation)
  int srcC, nsrc;
  const void* dy; int cout;
  float* slab;
  int N, D, H, W;
  int nyb, nxb, nseg, zsteps;           // patches, z segments per parity class, dY planes per segment
  int nco;                              // output-channel combos (blockIdx.y = ci combo * nco + co combo)
  int items;                            // work items: N x DIL parity classes x nseg x nyb x nxb
};

static constexpr int WM_TX = 32, WM_RY = 4, WM_HXP = 36, WM_NW = 4, WM_XRING = 3, WM_YRING = 5;

template <int NCB, int NOB, int DIL> struct WmGeo {
  static constexpr int XC = 16 * NCB, YC = 16 * NOB;                  // channels of X / dY per workgroup
  static constexpr int NPX = XC / 8, NPY = YC / 8, VBX = XC * 2, VBY = YC * 2;
  static constexpr int PW = NCB * NOB / WM_NW;                        // (co block, ci block) pairs per wave
  static constexpr int HX = WM_TX + 2 * DIL, HY = WM_RY + 2 * DIL;
  static constexpr int ROWBX = WM_HXP * VBX, ROWBY = WM_TX * VBY;
  static constexpr int NIX = (HY * WM_HXP * NPX + 63) / 64, PLBX = NIX * 1024, ITEMSX = (NIX + WM_NW - 1) / WM_NW;
  static constexpr int NIY = WM_RY * WM_TX * NPY / 64, PLBY = NIY * 1024, ITEMSY = NIY / WM_NW;
  static constexpr int TOT = ITEMSX + ITEMSY;                         // DMA instructions per wave and step
  static constexpr int YOFF = WM_XRING * PLBX, ZERO = YOFF + WM_YRING * PLBY, DUMP = ZERO + PLBY, LDS = DUMP + 1024;
  static_assert(PW == 1 || PW == 2, "pairs per wave");
  static_assert(NIY % WM_NW == 0, "dY plane instructions split evenly");
  static_assert(HX <= WM_HXP, "row pitch");
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert((HY - 1) * ROWBX + WM_HXP * VBX < 65536 && (WM_RY - 1) * ROWBY + WM_TX * VBY < 65536, "immediates");
};

__device__ __forceinline__ void wm_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// through a buffer descriptor (round 4; see conv_march.hip march_dma16_buf): base + scalar plane offset + the lane's constant offset;
// a lane beyond num_records -- a padding voxel, or any lane of a plane outside the march (zero records) -- writes ZEROS into the LDS
typedef unsigned int wmu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wm_dma16_buf(unsigned voff, wmu32x4 rsrc, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wm_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// piece permutation of the voxel at column v.  A transposing read takes, per 16 lanes, 4 consecutive voxels x 32 bytes (two
// adjacent pieces); 32 lanes = the voxels v..v+3 and v+8..v+11.  64-byte records: the four voxels already sit in different
// banks, bit 3 of v separates the two groups.  128-byte records: voxels v and v+2 share their banks -> bit 1 of v moves the
// piece pair, bit 3 separates the groups.  (XOR acts on the pair index: bit 0 of the piece stays.)
template <int NP> __device__ __forceinline__ int wm_swz(int v) {
  if constexpr (NP == 4) return ((v >> 3) & 1) << 1;
  else return ((((v >> 1) & 1) | (((v >> 3) & 1) << 1))) << 1;
}

template <int N, typename F> __device__ __forceinline__ void wm_for(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
    (f(std::integral_constant<int, I>{}), ...);
  }(std::make_integer_sequence<int, N>{});
}

// The matrix instruction as inline asm with the accumulator pinned to the accumulator half of the register file ("+a": D = C
// in place).  Through the builtin the register allocator spread the 216 accumulator registers over both halves and then
// spilled fragments; the asm leaves the vector half to the fragments and addresses.  Its operands are ordinary data
// dependencies (the compiler still waits for the LDS reads that produce them); an accumulator is only read back after the
// last march (behind explicit wait states).
template <typename T> __device__ __forceinline__ void wm_mfma(wmf32x4& c, wmb16x8 a, wmb16x8 b) {
  if constexpr (std::is_same<T, f16_t>::value) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// 8 voxels of one channel: two transposing reads of 4 voxels x 16 channels each (lane i of a 16-lane group receives channel i)
__device__ __forceinline__ wmb16x8 wm_frag(unsigned addr0, unsigned addr1) {
  typedef __attribute__((address_space(3))) wmb16x4 lds_b4;
  const wmb16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(size_t)addr0);
  const wmb16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(size_t)addr1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// processing order of the X rows: dilation 2 pairs X row hy with the dY rows hy, hy-2, hy-4 -> even rows first, then odd
template <int DIL, int HY> __host__ __device__ constexpr int wm_row(int i) {
  if constexpr (DIL == 1) return i;
  else return i < (HY + 1) / 2 ? 2 * i : 2 * (i - (HY + 1) / 2) + 1;
}

// the v-th y-tap (oldest dY row first: ty = 2, 1, 0) that pairs X row hy with a dY row inside the patch
template <int DIL> __host__ __device__ constexpr int wm_tap_y(int hy, int v) {
  for (int ty = 2; ty >= 0; --ty) {
    const int r = hy - DIL * ty;
    if (r >= 0 && r < WM_RY) { if (v == 0) return ty; --v; }
  }
  return -1;
}
template <int DIL, int HY, int PW> __host__ __device__ constexpr int wm_row_mfmas(int i) {
  int n = 0;
  for (int ty = 0; ty < 3; ++ty) { const int r = wm_row<DIL, HY>(i) - DIL * ty; n += (r >= 0 && r < WM_RY) ? 9 * PW : 0; }
  return n;
}
template <int DIL, int HY, int PW> __host__ __device__ constexpr int wm_row_frags(int i) {
  return 3 + (wm_row<DIL, HY>(i) < WM_RY ? 3 * PW : 0);
}

template <typename T, int NCB, int NOB, int DIL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
wgrad_march_kernel(WmArgs a) {
  using Geo = WmGeo<NCB, NOB, DIL>;
  constexpr int PW = Geo::PW, NPX = Geo::NPX, NPY = Geo::NPY, VBX = Geo::VBX, VBY = Geo::VBY, HX = Geo::HX, HY = Geo::HY;
  constexpr int RY = WM_RY, ROWBX = Geo::ROWBX, ROWBY = Geo::ROWBY, PLBX = Geo::PLBX, PLBY = Geo::PLBY;
  constexpr int NIX = Geo::NIX, ITEMSX = Geo::ITEMSX, ITEMSY = Geo::ITEMSY, TOT = Geo::TOT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cib = wave % NCB, cog = wave / NCB;            // input-channel block; output-channel blocks cog * PW + k
  const int grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int cic = blockIdx.y / a.nco, coc = blockIdx.y % a.nco;
  const int ci0 = cic * Geo::XC, co0 = coc * Geo::YC;

  // the zero dY plane
  for (int i = tid * 16; i < PLBY; i += 256 * 16) *reinterpret_cast<uint4*>(smem + Geo::ZERO + i) = make_uint4(0, 0, 0, 0);

  // ---- fragment addressing (lane part; rows are immediates, ring slots are added per step) ----
  unsigned xoff[3][2], yoff[PW][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int hx = DIL * dx + 8 * grp + 4 * r + q;
      xoff[dx][r] = (unsigned)(hx * VBX + (((cib * 2 + (p >> 1)) ^ wm_swz<NPX>(hx)) * 16) + (p & 1) * 8);
    }
#pragma unroll
  for (int k = 0; k < PW; ++k)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int v = 8 * grp + 4 * r + q;
      yoff[k][r] = (unsigned)(v * VBY + ((((cog * PW + k) * 2 + (p >> 1)) ^ wm_swz<NPY>(v)) * 16) + (p & 1) * 8);
    }

  wmf32x4 acc[PW][27];
#pragma unroll
  for (int k = 0; k < PW; ++k)
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[k][t] = wmf32x4{0.f, 0.f, 0.f, 0.f};

  const long long xplane = (long long)a.H * a.W * a.srcC * (long long)sizeof(T);
  const long long yplane = (long long)a.H * a.W * a.cout * (long long)sizeof(T);

  for (int item = blockIdx.x; item < a.items; item += gridDim.x) {
    int t = item;
    const int xb = t % a.nxb; t /= a.nxb;
    const int yb = t % a.nyb; t /= a.nyb;
    const int seg = t % a.nseg; t /= a.nseg;
    const int pz = t % DIL;
    const int n = t / DIL;
    const int x0 = xb * WM_TX, y0 = yb * RY;
    const int nplanes = (a.D - pz + DIL - 1) / DIL;
    const int q0 = seg * a.zsteps;
    const int Z = min(a.zsteps, nplanes - q0);            // dY planes of this march (may be <= 0 for a ragged last segment)
    const int nsteps = Z > 0 ? Z + 2 : 0;                 // X planes q0-1 .. q0+Z

    const unsigned char* x0_n = reinterpret_cast<const unsigned char*>(a.src0) + (long long)n * a.D * xplane;
    const unsigned char* x1_n = reinterpret_cast<const unsigned char*>(a.nsrc > 1 ? a.src1 : a.src0) + (long long)n * a.D * xplane;
    const unsigned char* dy_n = reinterpret_cast<const unsigned char*>(a.dy) + (long long)n * a.D * yplane;
    // ONE descriptor for the X operand of this sample (base = the lower of the one or two source pointers, a lane of the other source
    // adds the distance between the tensors; the launcher checked that it all fits 32 bits), one for dY
    const unsigned char* xbase = x1_n < x0_n ? x1_n : x0_n;
    const unsigned off_x0 = (unsigned)(x0_n - xbase), off_x1 = (unsigned)(x1_n - xbase);
    const unsigned xspan = (off_x0 > off_x1 ? off_x0 : off_x1) + (unsigned)((long long)a.D * xplane);
    const unsigned yspan = (unsigned)((long long)a.D * yplane);
    unsigned xlo, xhi, ylo, yhi;
    {
      const unsigned long long u = reinterpret_cast<unsigned long long>(xbase), w = reinterpret_cast<unsigned long long>(dy_n);
      xlo = __builtin_amdgcn_readfirstlane((unsigned)u); xhi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) & 0xFFFFu;
      ylo = __builtin_amdgcn_readfirstlane((unsigned)w); yhi = __builtin_amdgcn_readfirstlane((unsigned)(w >> 32)) & 0xFFFFu;
    }
    // ---- DMA plan of this item: the lane's byte offset from the descriptor base inside a z-plane; ~0 = padding ----
    unsigned dox[ITEMSX], doy[ITEMSY];
#pragma unroll
    for (int it = 0; it < ITEMSX; ++it) {
      const int id = wave + WM_NW * it;
      const int L = id * 64 + lane;
      const int v = L / NPX, sl = L % NPX;
      const int hy = v / WM_HXP, hx = v % WM_HXP;
      const int ch = ci0 + ((sl ^ wm_swz<NPX>(hx)) * 8);
      const int y = y0 - DIL + hy, x = x0 - DIL + hx;
      const bool s1 = ch >= a.srcC;
      const int cl = s1 ? ch - a.srcC : ch;
      const bool ok = id < NIX && hx < HX && hy < HY && y >= 0 && y < a.H && x >= 0 && x < a.W;
      dox[it] = ok ? (unsigned)(((y * a.W + x) * a.srcC + cl) * (int)sizeof(T)) + (s1 ? off_x1 : off_x0) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int it = 0; it < ITEMSY; ++it) {
      const int L = (wave + WM_NW * it) * 64 + lane;
      const int v = L / NPY, sl = L % NPY;
      const int r = v / WM_TX, xx = v % WM_TX;
      const int ch = co0 + ((sl ^ wm_swz<NPY>(xx)) * 8);
      const int y = y0 + r, x = x0 + xx;
      doy[it] = (y < a.H && x < a.W) ? (unsigned)(((y * a.W + x) * a.cout + ch) * (int)sizeof(T)) : 0xFFFFFFFFu;
    }

    // DMA instruction `it` of the planes of step s (X plane -> slot xs, dY plane s -> slot ys); every wave issues exactly TOT
    // instructions per step (padding instructions land in the dump area)
    // (the plane-level part -- validity, 64-bit plane base, LDS slot base -- is computed ONCE per step by `plane_of` and handed
    // to the items: with one wave per SIMD every scalar instruction takes an issue slot from the MFMA stream, and the items
    // sit in different scheduling regions, so the compiler recomputed it for each of them)
    struct PlaneRef { unsigned xsoff, ysoff, xrec, yrec; unsigned xlds, ylds; bool yok; };
    auto plane_of = [&](int s, int xs, int ys) __attribute__((always_inline)) -> PlaneRef {
      PlaneRef r;
      const int pl = q0 - 1 + s;
      const int z = pz + DIL * pl;
      const bool xok = pl >= 0 && z < a.D && s < nsteps;              // wave-uniform
      r.xsoff = (unsigned)((long long)(xok ? z : 0) * xplane);
      r.xrec = xok ? xspan : 0u;
      r.yok = s < Z;                                                  // wave-uniform
      r.ysoff = (unsigned)((long long)(r.yok ? pz + DIL * (q0 + s) : 0) * yplane);
      r.yrec = r.yok ? yspan : 0u;
      r.xlds = lds_base + (unsigned)(xs * PLBX);
      r.ylds = lds_base + (unsigned)(Geo::YOFF + ys * PLBY);
      return r;
    };
    auto issue_item = [&](const PlaneRef& r, auto it_c) __attribute__((always_inline)) {
      constexpr int it = decltype(it_c)::value;
      if constexpr (it < ITEMSX) {
        const bool real = wave + WM_NW * it < NIX;                    // wave-uniform
        wmu32x4 rs;
        rs.x = xlo; rs.y = xhi; rs.z = real ? r.xrec : 0u; rs.w = 0x00020000u;
        wm_dma16_buf(dox[it], rs, r.xsoff, real ? r.xlds + (unsigned)((wave + WM_NW * it) * 1024) : lds_base + (unsigned)Geo::DUMP);
      } else {
        constexpr int iy = it - ITEMSX;
        wmu32x4 rs;
        rs.x = ylo; rs.y = yhi; rs.z = r.yrec; rs.w = 0x00020000u;
        wm_dma16_buf(doy[iy], rs, r.ysoff, r.yok ? r.ylds + (unsigned)((wave + WM_NW * iy) * 1024) : lds_base + (unsigned)Geo::DUMP);
      }
    };

    // fragment addresses of a step: X plane in slot xs; dY planes s, s-1, s-2 (dz = 0, 1, 2) in slots ys, ys-1, ys-2 -- a plane
    // outside the march reads the zero plane
    unsigned xa[3][2], ya[3][PW][2];
    auto set_addr = [&](int s, int xs, int ys) __attribute__((always_inline)) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int r = 0; r < 2; ++r) xa[dx][r] = lds_base + (unsigned)(xs * PLBX) + xoff[dx][r];
#pragma unroll
      for (int tz = 0; tz < 3; ++tz) {
        const int d = s - tz;
        int sl = ys - tz; sl = sl < 0 ? sl + WM_YRING : sl;
        const unsigned pb = lds_base + (unsigned)((d >= 0 && d < Z) ? Geo::YOFF + sl * PLBY : Geo::ZERO);
#pragma unroll
        for (int k = 0; k < PW; ++k)
#pragma unroll
          for (int r = 0; r < 2; ++r) ya[tz][k][r] = pb + yoff[k][r];
      }
    };
    wmb16x8 xf[2][3];
    wmb16x8 yw[RY][3][PW];
    // fragment f of row i (processing order) of the current addresses: f < 3: X, x-tap f; else dY row hy of plane tz, pair k
    auto load_frag = [&](auto i_c, auto f_c) __attribute__((always_inline)) {
      constexpr int i = decltype(i_c)::value, f = decltype(f_c)::value;
      constexpr int hy = wm_row<DIL, HY>(i);
      if constexpr (f < 3) xf[i & 1][f] = wm_frag(xa[f][0] + hy * ROWBX, xa[f][1] + hy * ROWBX);
      else {
        constexpr int tz = (f - 3) / PW, k = (f - 3) % PW;
        yw[hy][tz][k] = wm_frag(ya[tz][k][0] + hy * ROWBY, ya[tz][k][1] + hy * ROWBY);
      }
    };
    // MFMA m of row i: (the m / 9PW-th valid y-tap, oldest dY row first; dz; x-tap; pair)
    auto mfma_one = [&](auto i_c, auto m_c) __attribute__((always_inline)) {
      constexpr int i = decltype(i_c)::value, m = decltype(m_c)::value;
      constexpr int hy = wm_row<DIL, HY>(i);
      constexpr int ty = wm_tap_y<DIL>(hy, m / (9 * PW));
      constexpr int rem = m % (9 * PW), tz = rem / (3 * PW), dx = (rem / PW) % 3, k = rem % PW;
      constexpr int r = hy - DIL * ty;
      static_assert(r >= 0 && r < RY, "row pairing");
      wm_mfma<T>(acc[k][(tz * 3 + ty) * 3 + dx], yw[r][tz][k], xf[i & 1][dx]);
    };
    // one row: the fragments of row `nx` are requested one by one between equal shares of row i's MFMAs (hand-placed:
    // sched_barrier fences keep the order), so that every read has most of a row of MFMAs to land
    auto row = [&](auto i_c, auto nx_c, auto&& extra) __attribute__((always_inline)) {
      constexpr int i = decltype(i_c)::value, nx = decltype(nx_c)::value;
      constexpr int NF = wm_row_frags<DIL, HY, PW>(nx), NM = wm_row_mfmas<DIL, HY, PW>(i);
      extra();
      wm_for<NF>([&](auto g_c) __attribute__((always_inline)) {
        constexpr int g = decltype(g_c)::value;
        load_frag(nx_c, g_c);
        wm_for<(g + 1) * NM / NF - g * NM / NF>([&](auto j_c) __attribute__((always_inline)) {
          mfma_one(i_c, std::integral_constant<int, g * NM / NF + decltype(j_c)::value>{});
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    };

    // Vector-memory operations of a wave, in program order: [prologue: planes 0 and 1], then per step s exactly TOT: the planes
    // of step s + 2 (issued during the rows of step s).  The step ends with the synchronisation FOR THE NEXT STEP, placed
    // before the MFMAs of its last row: all LDS reads of the step are done (lgkmcnt(0)), at most the TOT instructions of this
    // step are outstanding (planes s + 1 have landed), one barrier (every wave's part is in, every wave is done reading the
    // slots of step s -- which the DMA of step s + 1 overwrites); then the first row of step s + 1 is requested and the last
    // row's MFMAs cover its latency.
    wm_wait_vm<0>();
    __syncthreads();                     // the previous item's readers are done; the zero plane is written
    if (nsteps == 0) continue;           // (block-uniform)
    {
      const PlaneRef p0 = plane_of(0, 0, 0), p1 = plane_of(1, 1, 1);
      wm_for<TOT>([&](auto it_c) __attribute__((always_inline)) { issue_item(p0, it_c); });
      wm_for<TOT>([&](auto it_c) __attribute__((always_inline)) { issue_item(p1, it_c); });
    }
    wm_wait_vm<TOT>();
    __builtin_amdgcn_s_barrier();
    int xs = 0, ys = 0;                  // ring slots of step s: s % 3, s % 5
    set_addr(0, 0, 0);
    wm_for<wm_row_frags<DIL, HY, PW>(0)>([&](auto f_c) __attribute__((always_inline)) { load_frag(std::integral_constant<int, 0>{}, f_c); });
    for (int s = 0; s < nsteps; ++s) {
      const int xs2 = xs + 2 >= WM_XRING ? xs + 2 - WM_XRING : xs + 2;
      const int ys2 = ys + 2 >= WM_YRING ? ys + 2 - WM_YRING : ys + 2;
      const PlaneRef pr = plane_of(s + 2, xs2, ys2);
      wm_for<HY - 1>([&](auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        row(i_c, std::integral_constant<int, i + 1>{}, [&]() __attribute__((always_inline)) {
          wm_for<TOT>([&](auto it_c) __attribute__((always_inline)) {
            constexpr int it = decltype(it_c)::value;
            if constexpr ((it * (HY - 1)) / TOT == i) issue_item(pr, it_c);
          });
        });
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wm_wait_vm<TOT>();
      __builtin_amdgcn_s_barrier();
      xs = xs + 1 == WM_XRING ? 0 : xs + 1;
      ys = ys + 1 == WM_YRING ? 0 : ys + 1;
      set_addr(s + 1, xs, ys);
      row(std::integral_constant<int, HY - 1>{}, std::integral_constant<int, 0>{}, []() {});
    }
  }
  // (wait states between the last MFMA and the reads of its result below)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  wm_wait_vm<0>();

  // ---- slab of this workgroup: [pair = wave * PW + k][tap][lane][4] ----
  float* out = a.slab + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (WM_NW * PW) + wave * PW) * (27 * 256) + lane * 4;
#pragma unroll
  for (int k = 0; k < PW; ++k)
#pragma unroll
    for (int t = 0; t < 27; ++t) *reinterpret_cast<wmf32x4*>(out + (k * 27 + t) * 256) = acc[k][t];
}

// sum of the slabs, 16-way parallel in a fixed order, f64 -> dw (cout, cin, 27)
__global__ void __launch_bounds__(256)
wgrad_march_reduce_kernel(const float* __restrict__ slab, int nslab, int ncb, int pw, int xc, int yc, int nco,
                          int cin_w, float* __restrict__ dw) {
  const int per = WM_NW * pw * 27 * 256;
  const int el = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  const int combo = blockIdx.y;
  const float* ptr = slab + (size_t)combo * nslab * per + e;
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
    const int comp = e & 3, lane = (e >> 2) & 63, tap = (e >> 8) % 27, pair = e / (27 * 256);
    const int wave = pair / pw, kk = pair % pw;
    const int cib = wave % ncb, cob = (wave / ncb) * pw + kk;
    const int co = (combo % nco) * yc + cob * 16 + 4 * (lane >> 4) + comp;
    const int ci = (combo / nco) * xc + cib * 16 + (lane & 15);
    dw[((size_t)co * cin_w + ci) * 27 + tap] = (float)t;
  }
}

struct WmCfg { int ncb, nob; };
static bool wgrad_march_cfg(int dtype, int taps, int dil, const SrcList& x, int cin_logical, int cout, Dims d, bool size_gate, WmCfg& c) {
  if (dtype_size(dtype) != 2 || taps != 27 || (dil != 1 && dil != 2)) return false;
  if (x.n < 1 || x.n > 2 || (x.n == 2 && x.C[0] != x.C[1])) return false;
  if (cin_logical != x.total() || cin_logical % 32 || cout % 32 || x.C[0] % 8) return false;
  {
    // the plane DMA addresses a sample through 32-bit buffer offsets: one sample of dY, and the one or two X tensors of a sample
    // INCLUDING the distance between them, must fit (the network's plan places the halves of a concatenation next to each other;
    // anything larger goes to the tiled kernel)
    const long long esz = 2, xs = (long long)d.D * d.H * d.W * x.C[0] * esz, ys = (long long)d.D * d.H * d.W * cout * esz;
    long long dist = 0;
    if (x.n == 2) {
      dist = reinterpret_cast<const unsigned char*>(x.ptr[1]) - reinterpret_cast<const unsigned char*>(x.ptr[0]);
      if (dist < 0) dist = -dist;
    }
    if (dist + xs >= 0xFFFFFFFFll || ys >= 0xFFFFFFFFll) return false;
  }
  if (size_gate) {
    // where it beats the tiled kernel (isolated launches, 4 samples): the fine levels (rows of >= 32 voxels, >= 48^3); every
    // dilation-2 layer down to 16^3 (the tiled kernel works on parity sub-lattices there: 32^3 64 -> 64 0.057 vs 0.080 ms, 16^3
    // 128 -> 128 0.058 vs 0.075 ms); 256-channel inputs (dc1: 0.118 vs 0.134 ms).  Dilation 1 with <= 128 input channels on the
    // coarse levels is a tie and stays where it was.
    static const bool off = std::getenv("SEUNET_NO_WGRAD_MARCH") != nullptr;
    const bool fine = d.W >= 32 && (long long)d.D * d.H * d.W >= 48LL * 48 * 48;
    static const bool no_coarse = std::getenv("SEUNET_MARCH_NO_COARSE") != nullptr;   // (diagnostic switch for A/B timing)
    if (off || !(fine || (!no_coarse && (dil == 2 || cin_logical >= 256)))) return false;
  }
  if (cin_logical % 64 == 0) c = {4, 2};
  else if (cout % 64 == 0) c = {2, 4};
  else c = {2, 2};
  return true;
}

bool wgrad_march_supported(int dtype, int taps, int dil, const SrcList& x, int cin_logical, int cout, Dims d) {
  WmCfg c;
  return wgrad_march_cfg(dtype, taps, dil, x, cin_logical, cout, d, true, c);
}

template <typename T, int NCB, int NOB, int DIL>
static int wgrad_march_launch(const WmArgs& a, dim3 grid, hipStream_t s) {
  using Geo = WmGeo<NCB, NOB, DIL>;
  static unsigned long long configured = 0;
  if (int e = configure_kernel_lds(configured, reinterpret_cast<const void*>(&wgrad_march_kernel<T, NCB, NOB, DIL>), Geo::LDS)) return e;
  wgrad_march_kernel<T, NCB, NOB, DIL><<<grid, 256, Geo::LDS, s>>>(a);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_wgrad_march(int dtype, int taps, int dil, const SrcList& x, int cin_logical, const void* dy, int cout,
                       float* dw, void* workspace, size_t ws_bytes, Dims d, hipStream_t s) {
  WmCfg c;
  SEUNET_CHECK(wgrad_march_cfg(dtype, taps, dil, x, cin_logical, cout, d, false, c),
               "wgrad_march: 16-bit 3x3x3 layers with 32k input and 32k output channels (one tensor or two equal halves) only");
  SEUNET_CHECK(ws_bytes >= 256, "wgrad_march: workspace too small");
  const int xc = 16 * c.ncb, yc = 16 * c.nob, pw = c.ncb * c.nob / WM_NW;
  WmArgs a{};
  a.src0 = x.ptr[0]; a.src1 = x.n > 1 ? x.ptr[1] : nullptr; a.srcC = x.C[0]; a.nsrc = x.n;
  a.dy = dy; a.cout = cout;
  a.slab = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + 256);
  a.N = d.N; a.D = d.D; a.H = d.H; a.W = d.W;
  a.nyb = cdiv(d.H, WM_RY); a.nxb = cdiv(d.W, WM_TX);
  a.nco = cout / yc;
  const int combos = (cin_logical / xc) * a.nco;
  const int gmax = 256 / combos < 1 ? 1 : 256 / combos;             // one workgroup per CU in total
  const int nplanes = cdiv(d.D, dil);
  const int per_seg = d.N * dil * a.nyb * a.nxb;
  int nseg = 1;
  while (per_seg * nseg < gmax && nplanes / (nseg * 2) >= 8) nseg *= 2;
  a.nseg = nseg; a.zsteps = cdiv(nplanes, nseg);
  a.items = per_seg * nseg;
  const int G = a.items < gmax ? a.items : gmax;
  const size_t per = (size_t)WM_NW * pw * 27 * 256;
  SEUNET_CHECK(ws_bytes >= 256 + (size_t)combos * G * per * sizeof(float), "wgrad_march: workspace too small");
  dim3 grid(G, combos);
  int e = -1;
  SEUNET_DTYPE_SWITCH(dtype, if constexpr (sizeof(T) == 2) {
    if (c.ncb == 4 && dil == 1) e = wgrad_march_launch<T, 4, 2, 1>(a, grid, s);
    else if (c.ncb == 4) e = wgrad_march_launch<T, 4, 2, 2>(a, grid, s);
    else if (c.nob == 4 && dil == 1) e = wgrad_march_launch<T, 2, 4, 1>(a, grid, s);
    else if (c.nob == 4) e = wgrad_march_launch<T, 2, 4, 2>(a, grid, s);
    else if (dil == 1) e = wgrad_march_launch<T, 2, 2, 1>(a, grid, s);
    else e = wgrad_march_launch<T, 2, 2, 2>(a, grid, s);
  });
  if (e) return e;
  wgrad_march_reduce_kernel<<<dim3((unsigned)(per / 16), combos), 256, 0, s>>>(a.slab, G, c.ncb, pw, xc, yc, a.nco, cin_logical, dw);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace seunet
