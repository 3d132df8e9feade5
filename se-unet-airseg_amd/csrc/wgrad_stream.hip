// Streaming weight gradient of the small-channel full-resolution 3x3x3 layers (ec1 / ec2 / ec3 / dc6; backward of nn.Conv3d
// at SE_UNet.py:15):   dW[co][ci][dz,dy,dx] = sum over (n, z, y, x) of  X[n][z+dz*d][y+dy*d][x+dx*d][ci] * dY[n][z][y][x][co]
//
// The tiled kernel (wgrad.hip) accumulates 32 x 32 (ci x co) tiles per tap; with 8..32 channels 50-87 % of every MFMA is
// padding and the launch is MFMA-bound at 0.36-0.43 ms against an HBM floor of 0.05-0.13 ms.  Here:
//   * v_mfma_f32_16x16x32_bf16 with K = the 32 voxels of one x-row: a 16 (ci) x 16 (co) tile per MFMA, no padding for 16/32
//     channels; 8 input channels fold two x-taps into the 16 rows (voxels x, x+d are adjacent 16-byte pieces);
//   * the same z-march as conv_stream.hip: a workgroup owns an 8 x 32 (y, x) patch, the X planes (with halo) and the dY
//     planes arrive by LDS-DMA two steps ahead (counted vmcnt + one raw barrier per step); X plane s meets the dY planes
//     s-1, s, s+1 (the three dz taps), so an X fragment is read once per (row, dy, dx) and feeds three MFMAs;
//   * both operands need "8 voxels of one channel" per lane while LDS holds [voxel][channel]: ds_read_b64_tr_b16 delivers
//     exactly that from the unmodified images;
//   * 8 waves = 2 groups of (dy, dx, channel-block) units x 4 row pairs; every wave keeps its 16 x 16 accumulators in
//     registers for the whole march; at the end the four row-pair partials are summed through LDS in a fixed order and the
//     workgroup writes ONE compact slab [27][ci][co]; a second kernel sums the slabs (f64, fixed order: deterministic).
#include "seunet_common.h"
#include <utility>
#include <type_traits>

namespace seunet {

typedef bf16_t bf16x4w __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x8w __attribute__((ext_vector_type(8)));
typedef float f32x4w __attribute__((ext_vector_type(4)));

struct WsArgs {
  const void* x; const void* dy; float* slab;
  int N, D, H, W;
  int nyb, nxb, nzseg, zsteps;
};

static constexpr int WS_TY = 8, WS_TX = 32, WS_NW = 8, WS_PF = 2;

template <int CIN, int COUT, int DIL> struct WsGeo {
  static constexpr bool XF = CIN == 8;                          // two x-taps folded into the 16 tile rows
  static constexpr int NP = CIN / 8;
  static constexpr int HX = WS_TX + 2 * DIL + (XF ? DIL : 0), HY = WS_TY + 2 * DIL;
  static constexpr int NVP = HX * HY, G = (NVP + 63) / 64;
  static constexpr int PS = G * 1024, XPLANE = NP * PS;
  static constexpr int YPLANE = WS_TY * WS_TX * COUT * 2, NDY = YPLANE / 1024;   // dY plane: natural [row][x][co]
  static constexpr int RX = WS_PF + 2, RY = WS_PF + 4;
  static constexpr int XITEMS = (NP * G + WS_NW - 1) / WS_NW, YITEMS = (NDY + WS_NW - 1) / WS_NW;
  static constexpr int LW = XITEMS + YITEMS;
  static constexpr int CB = CIN >= 16 ? CIN / 16 : 1, OB = COUT >= 16 ? COUT / 16 : 1;
  static constexpr int AU = XF ? 6 : 9 * CB;                    // A-units: (dy, dx[, ci block]) or (dy, dx pair)
  static constexpr int AUG = (AU + 1) / 2;                      // per wave group
  static constexpr int YOFF = RX * XPLANE, DUMP = YOFF + RY * YPLANE + 64;   // (+64 B: a fragment read may run 16 B past a plane)
  static constexpr int STAGE = WS_NW * 8 * 1024;               // the final cross-wave sum: [8 waves][8 tiles][64 lanes][16 B]
  static constexpr int LDS = DUMP + 1024 > STAGE ? DUMP + 1024 : STAGE;
  static constexpr int SLAB = 27 * CIN * COUT;                  // floats per workgroup
};

__device__ __forceinline__ void ws_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// through a buffer descriptor (round 4; see conv_stream.hip stream_dma16_buf): base + scalar plane offset + the lane's constant
// offset; a lane beyond num_records -- a padding voxel, or any lane of a plane outside the march (zero records) -- writes ZEROS
typedef unsigned int wsu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ws_dma16_buf(unsigned voff, wsu32x4 rsrc, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void ws_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <typename T, int CIN, int COUT, int DIL>
__global__ void __launch_bounds__(512, 2)
wgrad_stream_kernel(WsArgs a) {
  using Geo = WsGeo<CIN, COUT, DIL>;
  constexpr bool XF = Geo::XF;
  constexpr int NP = Geo::NP, HX = Geo::HX, NVP = Geo::NVP, G = Geo::G, PS = Geo::PS, XPLANE = Geo::XPLANE, YPLANE = Geo::YPLANE;
  constexpr int NDY = Geo::NDY, RX = Geo::RX, RY = Geo::RY, XITEMS = Geo::XITEMS, YITEMS = Geo::YITEMS, LW = Geo::LW;
  constexpr int CB = Geo::CB, OB = Geo::OB, AU = Geo::AU, AUG = Geo::AUG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) bf16x4w lds_b4;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave & 1, rp = wave >> 1;                        // A-unit group, row pair (rows 2 rp, 2 rp + 1)
  int t;
  {
    const int nt = gridDim.x, b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int xb = t % a.nxb, yb = t / a.nxb;
  const int seg = blockIdx.y / DIL, pz = blockIdx.y % DIL;
  const int n = blockIdx.z;
  const int x0 = xb * WS_TX, y0 = yb * WS_TY;
  const int q0 = seg * a.zsteps;
  const int nplanes = (a.D - pz + DIL - 1) / DIL;
  const int q1 = min(q0 + a.zsteps, nplanes);
  const int nsteps = q1 - q0 + 2;                                  // X planes q0-1 .. q1
  const long long xplane_bytes = (long long)a.H * a.W * CIN * 2, yplane_bytes = (long long)a.H * a.W * COUT * 2;
  const unsigned char* x_n = reinterpret_cast<const unsigned char*>(a.x) + (long long)n * a.D * xplane_bytes;
  const unsigned char* y_n = reinterpret_cast<const unsigned char*>(a.dy) + (long long)n * a.D * yplane_bytes;
  const unsigned xspan = (unsigned)((long long)a.D * xplane_bytes), yspan = (unsigned)((long long)a.D * yplane_bytes);   // (< 2^32: launcher)
  unsigned xlo, xhi, ylo, yhi;
  {
    const unsigned long long u = reinterpret_cast<unsigned long long>(x_n), w = reinterpret_cast<unsigned long long>(y_n);
    xlo = __builtin_amdgcn_readfirstlane((unsigned)u); xhi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) & 0xFFFFu;
    ylo = __builtin_amdgcn_readfirstlane((unsigned)w); yhi = __builtin_amdgcn_readfirstlane((unsigned)(w >> 32)) & 0xFFFFu;
  }

  // ---- DMA plans (per lane, march-invariant) ----
  unsigned xoff[XITEMS], xlds[XITEMS];
#pragma unroll
  for (int it = 0; it < XITEMS; ++it) {
    const int id = wave + WS_NW * it;
    const int p = id / G, gi = id % G;
    const int v = gi * 64 + lane;
    const int hy = v / HX, hx = v % HX;
    const int y = y0 - DIL + hy, x = x0 - DIL + hx;
    const bool ok = id < NP * G && v < NVP && y >= 0 && y < a.H && x >= 0 && x < a.W;
    xoff[it] = ok ? (unsigned)(((y * a.W + x) * CIN + p * 8) * 2) : 0xFFFFFFFFu;
    xlds[it] = (unsigned)(p * PS + gi * 1024);
  }
  unsigned yoff[YITEMS];
#pragma unroll
  for (int it = 0; it < YITEMS; ++it) {
    const int id = wave + WS_NW * it;
    const int b = id * 1024 + lane * 16;                           // byte inside the plane image [row][x][co]
    const int v = b / (COUT * 2), cbyte = b % (COUT * 2);
    const int row = v / WS_TX, xx = v % WS_TX;
    const bool ok = id < NDY && y0 + row < a.H && x0 + xx < a.W;
    yoff[it] = ok ? (unsigned)((((y0 + row) * a.W + x0 + xx) * COUT) * 2 + cbyte) : 0xFFFFFFFFu;
  }
  auto dma_x = [&](int s, int slot, auto it_c) __attribute__((always_inline)) {
    constexpr int it = decltype(it_c)::value;
    const bool real = wave + WS_NW * it < NP * G;
    const int pl = q0 - 1 + s, z = pz + DIL * pl;
    const bool zok = real && pl >= 0 && z < a.D && s < nsteps;
    wsu32x4 rs;
    rs.x = xlo; rs.y = xhi; rs.z = zok ? xspan : 0u; rs.w = 0x00020000u;
    ws_dma16_buf(xoff[it], rs, (unsigned)((long long)(zok ? z : 0) * xplane_bytes),
                 real ? lds_base + (unsigned)(slot * XPLANE) + xlds[it] : lds_base + (unsigned)Geo::DUMP);
  };
  auto dma_y = [&](int jd, int slot, auto it_c) __attribute__((always_inline)) {   // dY plane q0 - 2 + jd; zero outside [q0, q1)
    constexpr int it = decltype(it_c)::value;
    const bool real = wave + WS_NW * it < NDY;
    const int pl = q0 - 2 + jd, z = pz + DIL * pl;
    const bool zok = real && pl >= q0 && pl < q1;
    wsu32x4 rs;
    rs.x = ylo; rs.y = yhi; rs.z = zok ? yspan : 0u; rs.w = 0x00020000u;
    ws_dma16_buf(yoff[it], rs, (unsigned)((long long)(zok ? z : 0) * yplane_bytes),
                 real ? lds_base + (unsigned)(Geo::YOFF + slot * YPLANE + (wave + WS_NW * it) * 1024) : lds_base + (unsigned)Geo::DUMP);
  };
  auto dma_group = [&](int s, int xslot, int jd, int yslot, int part) __attribute__((always_inline)) {
    // part 0 / 1: first / second half of the instructions (spread over the two rows of a step); part 2: everything
    [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
      (((part == 2 || (I & 1) == part) ? dma_x(s, xslot, std::integral_constant<int, I>{}) : (void)0), ...);
    }(std::make_integer_sequence<int, XITEMS>{});
    [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
      (((part == 2 || (I & 1) == part) ? dma_y(jd, yslot, std::integral_constant<int, I>{}) : (void)0), ...);
    }(std::make_integer_sequence<int, YITEMS>{});
  };

  // ---- fragment addressing (ds_read_b64_tr_b16): 16-lane group g = k-group (voxels 8g..8g+7), lane li = 4q + p supplies the
  //      address of voxel 8g + q (+4 for the second read), columns 4p..4p+3 of the 16-column block ----
  const int g = lane >> 4, li = lane & 15, fq = li >> 2, fp = li & 3;
  // X: columns = ci (16 per block; XF: ci 0..7 of voxel x and of voxel x + DIL).  Offset of (row 0, dx index 0, block 0):
  const int xfrag0 = XF ? (((8 * g + fq) + DIL * (fp >> 1)) * 16 + (fp & 1) * 8)
                        : ((fp >> 1) * PS + (8 * g + fq) * 16 + (fp & 1) * 8);
  // dY: columns = co; offset of (row 0, block 0)
  const int yfrag0 = ((8 * g + fq) * COUT + 4 * fp) * 2;

  f32x4w acc[AUG][3][OB];
#pragma unroll
  for (int u = 0; u < AUG; ++u)
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
      for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[u][dz][ob][e] = 0.f;

  auto read_x = [&](const unsigned char* p) __attribute__((always_inline)) -> bf16x8w {   // second read: + 4 voxels = + 64 B
    const bf16x4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)p);
    const bf16x4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(p + 64));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto read_y = [&](const unsigned char* p) __attribute__((always_inline)) -> bf16x8w {   // + 4 voxels = + 4 * COUT * 2 B
    const bf16x4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)p);
    const bf16x4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(p + 8 * COUT));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  // ---- one step: X plane s (slot xs) against the dY planes jd = s (dz = +1), s + 1 (dz = 0), s + 2 (dz = -1) ----
  auto compute = [&](int s, int xs, int ys0, int xs_pf, int ys_pf) __attribute__((always_inline)) {
    const unsigned char* xp = smem + xs * XPLANE + xfrag0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int w = 2 * rp + r;                                       // output row
      bf16x8w bf[3][OB];
#pragma unroll
      for (int dzi = 0; dzi < 3; ++dzi) {                              // dz = dzi - 1  <->  dY jd = s + 2 - dzi ... (dz = +1: jd = s)
        int ysl = ys0 + (2 - dzi);
        ysl = ysl >= RY ? ysl - RY : ysl;
        const unsigned char* yp = smem + Geo::YOFF + ysl * YPLANE + yfrag0 + (w * WS_TX) * COUT * 2;
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) bf[dzi][ob] = read_y(yp + ob * 32);
      }
      // the prefetch two steps ahead, half of the instructions after each row's dY reads
      dma_group(s + WS_PF, xs_pf, s + 2 + WS_PF, ys_pf, r);
#pragma unroll
      for (int u = 0; u < AUG; ++u) {
        const int au = grp * AUG + u;                                  // wave-uniform
        if (au >= AU) continue;
        int dyi, dxo, cb;
        if constexpr (XF) { dyi = au >> 1; dxo = 2 * (au & 1); cb = 0; }
        else { dyi = au / (3 * CB); dxo = (au / CB) % 3; cb = au % CB; }
        const bf16x8w af = read_x(xp + ((w + DIL * dyi) * HX + DIL * dxo) * 16 + (XF ? 0 : cb * 2 * PS));
#pragma unroll
        for (int dzi = 0; dzi < 3; ++dzi)
#pragma unroll
          for (int ob = 0; ob < OB; ++ob)
            if constexpr (std::is_same<T, f16_t>::value) {   // (the transposing reads are type-agnostic 16-bit patterns)
              typedef f16_t f16x8w __attribute__((ext_vector_type(8)));
              acc[u][dzi][ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8w, af), __builtin_bit_cast(f16x8w, bf[dzi][ob]), acc[u][dzi][ob], 0, 0, 0);
            } else {
              acc[u][dzi][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[dzi][ob], acc[u][dzi][ob], 0, 0, 0);
            }
      }
    }
  };

  // ---- the march (see conv_stream.hip for the vmcnt arithmetic; no stores in flight here) ----
  int ysl = 0;
  [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {   // the two dY planes ahead of group 0
    ((dma_y(0, 0, std::integral_constant<int, I>{}), dma_y(1, 1, std::integral_constant<int, I>{})), ...);
  }(std::make_integer_sequence<int, YITEMS>{});
#pragma unroll
  for (int k = 0; k < WS_PF; ++k) dma_group(k, k, k + 2, k + 2, 2);
  ws_wait_vm<(WS_PF - 1) * LW>();
  __builtin_amdgcn_s_barrier();
  int xs = 0, xs_pf = WS_PF, ys_pf = 2 + WS_PF;
  for (int s = 0; s < nsteps; ++s) {
    if (s > 0) {
      ws_wait_vm<(WS_PF - 1) * LW>();
      __builtin_amdgcn_s_barrier();
    }
    compute(s, xs, ysl, xs_pf, ys_pf);
    xs = xs == RX - 1 ? 0 : xs + 1;
    xs_pf = xs_pf == RX - 1 ? 0 : xs_pf + 1;
    ysl = ysl == RY - 1 ? 0 : ysl + 1;
    ys_pf = ys_pf == RY - 1 ? 0 : ys_pf + 1;
  }
  ws_wait_vm<0>();
  __syncthreads();          // every DMA has landed, every wave is done reading: the rings are dead

  // ---- sum the four row-pair partials of each group (fixed order) and write the workgroup's slab [27][CIN][COUT] ----
  float* slab = a.slab + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (size_t)Geo::SLAB;
  f32x4w* stage = reinterpret_cast<f32x4w*>(smem);                    // [wave][tile in round][lane]
  constexpr int TILES = AUG * 3 * OB, ROUND = 8;
#pragma unroll
  for (int t0 = 0; t0 < TILES; t0 += ROUND) {
#pragma unroll
    for (int k = 0; k < ROUND; ++k) {
      const int tl = t0 + k;
      if (tl < TILES) stage[(wave * ROUND + k) * 64 + lane] = acc[tl / (3 * OB)][(tl / OB) % 3][tl % OB];
    }
    __syncthreads();
    // 2 groups x ROUND tiles x 64 lanes sums of 4 partials: 1024 float4 results, 2 per thread
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + 512 * j;
      const int gg = idx >> 9, k = (idx >> 6) & 7, ln = idx & 63;
      const int tl = t0 + k, u = tl / (3 * OB), dzi = (tl / OB) % 3, ob = tl % OB;
      const int au = gg * AUG + u;
      if (tl < TILES && au < AU) {
        f32x4w sum = stage[((gg + 0) * ROUND + k) * 64 + ln];
#pragma unroll
        for (int p = 1; p < 4; ++p) sum += stage[((gg + 2 * p) * ROUND + k) * 64 + ln];
        const int co = ob * 16 + (ln & 15);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cl = 4 * (ln >> 4) + e;                            // tile row
          int dyi, dxi, ci;
          if constexpr (XF) { dyi = au >> 1; dxi = 2 * (au & 1) + (cl >> 3); ci = cl & 7; }
          else { dyi = au / (3 * CB); dxi = (au / CB) % 3; ci = (au % CB) * 16 + cl; }
          if (dxi < 3 && ci < CIN && co < COUT) slab[(((dzi * 3 + dyi) * 3 + dxi) * CIN + ci) * COUT + co] = sum[e];
        }
      }
    }
    __syncthreads();
  }
}

// dW (PyTorch (cout, cin_w, 3, 3, 3)) = f64 fixed-order sum of the slabs; 16 slab elements per block, 16 partial sums each
__global__ void __launch_bounds__(256)
wgrad_stream_reduce_kernel(const float* __restrict__ slab, int nslab, int per, int cin, int cout, int cin_w, int cout_w,
                           float* __restrict__ dw) {
  const int el = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  double s0 = 0.0, s1 = 0.0;
  if (e < per) {
    const float* p = slab + e;
    int k = part;
    for (; k + 16 < nslab; k += 32) { s0 += (double)p[(size_t)k * per]; s1 += (double)p[(size_t)(k + 16) * per]; }
    if (k < nslab) s0 += (double)p[(size_t)k * per];
  }
  __shared__ double red[16][17];
  red[part][el] = s0 + s1;
  __syncthreads();
  if (part == 0 && e < per) {
    double tsum = red[0][el];
#pragma unroll
    for (int q = 1; q < 16; ++q) tsum += red[q][el];
    const int co = e % cout, ci = (e / cout) % cin, tap = e / (cout * cin);
    if (ci < cin_w && co < cout_w) dw[((size_t)co * cin_w + ci) * 27 + tap] = (float)tsum;
  }
}

static bool ws_shape_ok(int cin, int cout, int dil) {
  if (cin == 32 && cout == 16) return dil == 1;          // (dilation 2 would need 161 KB of LDS)
  return (cin == 8 && (cout == 8 || cout == 16)) || (cin == 16 && cout == 32) || (cin == 16 && cout == 16);
}
bool wgrad_stream_supported(int dtype, int taps, int dil, int x_c, int dy_c) {
  return (dtype == SEUNET_BF16 || dtype == SEUNET_F16) && taps == 27 && (dil == 1 || dil == 2) && ws_shape_ok(x_c, dy_c, dil);
}
static int ws_zsteps(Dims d, int dil) {
  // long marches: few slabs, the pipeline fill amortised; aim at >= 256 workgroups (one per CU)
  const int planes = cdiv(d.D, dil);
  const int base = cdiv(d.H, WS_TY) * cdiv(d.W, WS_TX) * d.N * dil;
  int segs = (256 + base - 1) / base;
  if (segs < 1) segs = 1;
  if (segs > cdiv(planes, 8)) segs = cdiv(planes, 8);
  if (segs < 1) segs = 1;
  return cdiv(planes, segs);
}
static int ws_slabs(Dims d, int dil) {
  const int planes = cdiv(d.D, dil);
  return cdiv(d.H, WS_TY) * cdiv(d.W, WS_TX) * cdiv(planes, ws_zsteps(d, dil)) * dil * d.N;
}
size_t wgrad_stream_workspace_bytes(int x_c, int dy_c, int dil, Dims d) {
  return (size_t)ws_slabs(d, dil) * 27 * x_c * dy_c * sizeof(float);
}

template <typename T, int CIN, int COUT, int DIL>
static int ws_launch_t(const WsArgs& a, dim3 grid, hipStream_t s) {
  using Geo = WsGeo<CIN, COUT, DIL>;
  static_assert(Geo::LDS <= 160 * 1024, "wgrad_stream: LDS budget");
  static unsigned long long configured = 0;
  if (int e = configure_kernel_lds(configured, reinterpret_cast<const void*>(&wgrad_stream_kernel<T, CIN, COUT, DIL>), Geo::LDS)) return e;
  wgrad_stream_kernel<T, CIN, COUT, DIL><<<grid, WS_NW * 64, Geo::LDS, s>>>(a);
  SEUNET_LAUNCH_CHECK();
  return 0;
}
template <int CIN, int COUT, int DIL>
static int ws_launch(int dtype, const WsArgs& a, dim3 grid, hipStream_t s) {
  return dtype == SEUNET_F16 ? ws_launch_t<f16_t, CIN, COUT, DIL>(a, grid, s) : ws_launch_t<bf16_t, CIN, COUT, DIL>(a, grid, s);
}

// x: [N][D][H][W][x_c] (x_c = 8 | 16 | 32, cin_w leading channels carry weights); dy: [N][D][H][W][dy_c];
// dw: (cout_w, cin_w, 3, 3, 3) f32, overwritten.
int launch_wgrad_stream(int dtype, int dil, const void* x, int x_c, int cin_w, const void* dy, int dy_c, int cout_w, float* dw,
                        void* workspace, size_t ws_bytes, Dims d, hipStream_t s) {
  SEUNET_CHECK(wgrad_stream_supported(dtype, 27, dil, x_c, dy_c), "wgrad_stream: unsupported shape (%d x %d channels, dilation %d)", x_c, dy_c, dil);
  SEUNET_CHECK(x && dy && dw && workspace && cin_w >= 1 && cin_w <= x_c && cout_w >= 1 && cout_w <= dy_c, "wgrad_stream: bad argument");
  SEUNET_CHECK(ws_bytes >= wgrad_stream_workspace_bytes(x_c, dy_c, dil, d), "wgrad_stream: workspace too small");
  SEUNET_CHECK((long long)d.D * d.H * d.W * (x_c > dy_c ? x_c : dy_c) * 2 < 0xFFFFFFFFll,
               "wgrad_stream: one sample exceeds the 32-bit buffer offsets of this kernel");
  WsArgs a{};
  a.x = x; a.dy = dy; a.slab = reinterpret_cast<float*>(workspace);
  a.N = d.N; a.D = d.D; a.H = d.H; a.W = d.W;
  const int planes = cdiv(d.D, dil);
  a.zsteps = ws_zsteps(d, dil);
  a.nzseg = cdiv(planes, a.zsteps);
  a.nyb = cdiv(d.H, WS_TY); a.nxb = cdiv(d.W, WS_TX);
  SEUNET_CHECK(d.N <= 65535, "wgrad_stream: batch too large");
  dim3 grid(a.nyb * a.nxb, a.nzseg * dil, d.N);
  int e = 1;
  if (x_c == 8 && dy_c == 8) e = dil == 1 ? ws_launch<8, 8, 1>(dtype, a, grid, s) : ws_launch<8, 8, 2>(dtype, a, grid, s);
  else if (x_c == 8 && dy_c == 16) e = dil == 1 ? ws_launch<8, 16, 1>(dtype, a, grid, s) : ws_launch<8, 16, 2>(dtype, a, grid, s);
  else if (x_c == 16 && dy_c == 16) e = dil == 1 ? ws_launch<16, 16, 1>(dtype, a, grid, s) : ws_launch<16, 16, 2>(dtype, a, grid, s);
  else if (x_c == 16 && dy_c == 32) e = dil == 1 ? ws_launch<16, 32, 1>(dtype, a, grid, s) : ws_launch<16, 32, 2>(dtype, a, grid, s);
  else if (x_c == 32 && dy_c == 16) e = ws_launch<32, 16, 1>(dtype, a, grid, s);
  if (e) return e;
  const int per = 27 * x_c * dy_c;
  wgrad_stream_reduce_kernel<<<cdiv(per, 16), 256, 0, s>>>(a.slab, grid.x * grid.y * grid.z, per, x_c, dy_c, cin_w, cout_w, dw);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace seunet
