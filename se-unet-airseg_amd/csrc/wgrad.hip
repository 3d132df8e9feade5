// Weight gradient of the 3x3x3 / 1x1x1 convolutions on gfx950 matrix cores.
//   dW[co][ci][tap] = sum over (n, voxel v) of  X[n][v + off(tap)][ci] * dY[n][v][co]
// (backward of nn.Conv3d at reference SE_UNet.py:15,42,57; X may be a fused channel concatenation.)
//
//   GEMM view   M = input channels (32 per workgroup), N = output channels (32), K = voxels
//   workgroup   256 threads = 4 waves, persistent over spatial tiles, two workgroups per CU (one computes while the other's
//               tile is in flight); the 27 taps are split over the 4 waves (7/7/7/6), each wave keeping one 32x32 f32
//               accumulator per tap in registers across ALL its tiles
//   staging     LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write phase.  LDS image planar,
//               [16-B piece of the 32 channels][voxel][16 B]; padding voxels / channels read a zero page.  The DMA plan is
//               per workgroup: packed halo coordinates parked in LDS, scalar tensor bases, ~16 vector instructions per DMA
//   MFMA        bf16: v_mfma_f32_32x32x16_bf16, K-step = 16 consecutive x-voxels.  Both operands need "8 voxels of
//               one channel" per lane while the data is [voxel][channel]: gfx950's transposing LDS read
//               ds_read_b64_tr_b16 delivers exactly that from the unmodified image (a 16-lane group reads a
//               4-voxel x 16-channel block -> conflict-free), tap shifts only move the row address (tap base register +
//               immediate).  The phase is pipelined by hand: fragments are requested two MFMAs ahead.
//               f32 : v_mfma_f32_32x32x2_f32 (K-step = 2 voxels, one element per lane, plain loads)
//   output      each workgroup stores its accumulators once to a slab (1x1x1: after summing its four waves in LDS); a
//               second kernel sums the slabs 16-way parallel in a fixed order (deterministic, no atomics, f64) straight
//               into the PyTorch (Cout,Cin,3,3,3) layout
#include "seunet_common.h"
#include <utility>
#include <cstdlib>
#include <type_traits>

namespace seunet {

extern unsigned long long* g_conv_debug;   // conv_igemm.hip

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));

struct WgArgs {
  const void* src0; const void* src1; const void* src2;
  int srcC0, srcC1, srcC2;
  int cum1, cum2;
  int cin;                 // logical input channels
  const void* dy; int cout;
  float* slab;
  const void* zero;        // >= 16 zero bytes: source of padding voxels / channels for the LDS-DMA
  int N, D, H, W;
  int tx, ty, tz;          // tile counts per sample
  int co_tiles;
  unsigned long long* debug;   // diagnostic builds only (-DSEUNET_STAMP)
};

// spatial tile (z, y; x is always 32) on the (sub-)lattice.  Dilation 2 runs on the 8 parity sub-lattices
// (voxel = 2*lattice + parity), so the halo is one lattice voxel for every dilation.
template <typename T> struct WgTile;
template <> struct WgTile<bf16_t> { static constexpr int TZ = 2, TY = 4; };
template <> struct WgTile<f16_t> { static constexpr int TZ = 2, TY = 4; };
template <> struct WgTile<float> { static constexpr int TZ = 1, TY = 4; };

__device__ __forceinline__ const void* wg_uniform_ptr(const void* p) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
}

// NW = waves per workgroup: the 27 taps are dealt round-robin to the waves (NT = ceil(27/NW) accumulators each).
// Staging is LDS-DMA (global_load_lds_dwordx4): no staging registers and no ds_write phase, so two workgroups fit
// on a CU and one computes while the other's tile is in flight.  The LDS image is planar,
// [16-byte piece of the 32 channels][voxel][16 B]: one DMA wave-instruction writes 1 KB contiguous (64 consecutive
// voxels of one piece, the piece -- hence the source tensor -- being wave-uniform); plane strides are = 64 (mod 256)
// bytes so the transposing reads (4 voxels x 4 pieces per 32 lanes) touch every LDS bank once.
#ifdef SEUNET_STAMP
#define WSTAMP(i) do { const unsigned long long _t = __builtin_readcyclecounter(); ph[i] += _t - t_last; t_last = _t; } while (0)
#else
#define WSTAMP(i) do {} while (0)
#endif

template <typename T, int TAPS, int DIL, int NW>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 4)   // two workgroups per CU
wgrad_kernel(WgArgs a) {
#ifdef SEUNET_STAMP
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_readcyclecounter();
#endif
  constexpr int HALO = (TAPS == 27) ? 1 : 0;
  constexpr int STEP = (TAPS == 27) ? DIL : 1;
  constexpr int TZ = WgTile<T>::TZ, TY = WgTile<T>::TY, TX = 32;
  constexpr int HZ = TZ + 2 * HALO, HY = TY + 2 * HALO, HX = TX + 2 * HALO;
  constexpr int NVH = HZ * HY * HX, NVT = TZ * TY * TX;
  constexpr int EPP = 16 / sizeof(T);   // elements per 16-byte piece
  constexpr int PPV = 32 / EPP;         // pieces per voxel (32 channels): 4 (bf16) / 8 (f32)
  constexpr int XG = (NVH + 63) / 64, YG = (NVT + 63) / 64;               // 64-voxel groups
  constexpr int X_ITEMS = (XG * PPV + NW - 1) / NW, Y_ITEMS = (YG * PPV + NW - 1) / NW;   // DMA instructions per wave
  constexpr int XPL = (XG * 64 + 4) * 16, YPL = (YG * 64 + 4) * 16;       // plane strides in bytes, = 64 (mod 256)
  constexpr int NT = (TAPS == 27) ? (27 + NW - 1) / NW : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xs = smem;                 // [PPV][XG*64 (+4)][16 B]
  unsigned char* ys = smem + PPV * XPL;     // [PPV][YG*64 (+4)][16 B]
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int combo = blockIdx.y;
  const int ci0 = (combo / a.co_tiles) * 32, co0 = (combo % a.co_tiles) * 32;
  const long long V = (long long)a.D * a.H * a.W;
  const int tiles_per_par = a.tx * a.ty * a.tz;
  const int tiles_per_sample = tiles_per_par * STEP * STEP * STEP;
  const int total_tiles = tiles_per_sample * a.N;

  int tapoff[NT];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    int tap = wave + NW * ti;
    if (tap > TAPS - 1) tap = TAPS - 1;
    if (TAPS == 27) tapoff[ti] = (((tap / 9) * HALO) * HY + ((tap / 3) % 3) * HALO) * HX + (tap % 3) * HALO;
    else tapoff[ti] = 0;
  }
  f32x16 acc[NT];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;
  // 16-bit operand addressing (ds_read_b64_tr_b16; second read: +4 voxels = +64 B)
  const unsigned char* abase[NT];   // X image: lane offset + this wave's tap ti (1x1x1: + this wave's first row)
  const unsigned char* bbase;       // dY image
  {
    // 16-bit storage: v_mfma_f32_16x16x32 (K = the 32 voxels of a row; the chip holds a higher clock on this shape than on
    // 32x32x16).  16-lane group grp = k-group (voxels 8 grp .. 8 grp + 7, second read + 4 voxels); lane
    // li = 4q + p supplies voxel 8 grp + q, channels 4p .. 4p + 3 of a 16-channel block (block b: + 2 b planes)
    const int grp = lane >> 4, li = lane & 15;
    const int pl = (li & 3) >> 1;                                         // 16-byte piece (plane) inside the 16-channel block
    const int vo = 8 * grp + (li >> 2);                                   // voxel within the 32-voxel row
    const int sub = (li & 1) * 8;                                         // byte offset inside the piece
    const int wrow = (TAPS == 27) ? 0 : wave;                             // 1x1x1: wave w takes rows w, w + NW, ...
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
      abase[ti] = xs + (PPV == 4 ? pl : 0) * XPL + ((wrow / TY * HY + wrow % TY) * HX + tapoff[ti] + vo) * 16 + sub;
    bbase = ys + (PPV == 4 ? pl : 0) * YPL + (wrow * TX + vo) * 16 + sub;
  }

  // ---- DMA plan.  Wave-instruction k of a wave covers piece (wave + NW*(k % PV)) of voxel group k / PV (PV = PPV/NW
  // pieces per wave), 64 consecutive voxels of the LDS image per instruction.  Everything that does not depend on the
  // tile is computed once per workgroup: the lane's halo coordinates of each voxel group (one packed register per
  // group) and, per piece of this wave, the source tensor / channel (scalars).  Per tile and instruction that leaves a
  // range test of the three coordinates, one 64-bit multiply-add and the zero-page select: ~16 vector instructions
  // where recomputing the plan from scratch took ~50 -- the DMA issue phase was as long as the MFMA phase (stamps).
  static_assert(PPV % NW == 0, "pieces per voxel must be a multiple of the wave count");
  constexpr int PV = PPV / NW;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  // the lane's halo coordinates of its XG voxel groups, 16 bits each (hz: bits 0-1, hy: 2-4, hx: 5-10, bit 15: beyond the
  // halo tile), parked in a thread-private 32-byte LDS slot and re-read per tile: kept in registers they stay live
  // across the MFMA phase, and the spills that follow reload through scratch behind s_waitcnt vmcnt(0) -- i.e. behind
  // every DMA already in flight
  static_assert(XG <= 16 && HZ <= 4 && HY <= 8 && HX <= 64, "packed halo coordinates");
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4* cslot = reinterpret_cast<u32x4*>(smem + PPV * (XPL + YPL) + tid * 32);
  {
    unsigned w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < XG; ++g) {
      const int vox = g * 64 + lane;
      const int hx = vox % HX, r2 = vox / HX;
      const int hy = r2 % HY, hz = r2 / HY;
      const unsigned c = vox < NVH ? ((unsigned)hz | ((unsigned)hy << 2) | ((unsigned)hx << 5)) : 0x8000u;
      w[g >> 1] |= c << (16 * (g & 1));
    }
    u32x4 lo, hi;
    lo.x = w[0]; lo.y = w[1]; lo.z = w[2]; lo.w = w[3]; hi.x = w[4]; hi.y = w[5]; hi.z = w[6]; hi.w = w[7];
    cslot[0] = lo; cslot[1] = hi;
  }
  const unsigned char* xbase[PV]; long long xstride[PV]; bool xvalid[PV];   // per piece of this wave (scalars)
  const unsigned char* ybase[PV]; bool yvalid[PV];
#pragma unroll
  for (int j = 0; j < PV; ++j) {
    const int piece = wave_s + NW * j;
    const int ch0 = ci0 + piece * EPP;
    const void* sp = a.src0; int sC = a.srcC0, c = ch0;
    if (ch0 >= a.cum2) { sp = a.src2; sC = a.srcC2; c = ch0 - a.cum2; }
    else if (ch0 >= a.cum1) { sp = a.src1; sC = a.srcC1; c = ch0 - a.cum1; }
    xvalid[j] = ch0 < a.cin;
    xbase[j] = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(sp) + c);
    xstride[j] = (long long)sC * (long long)sizeof(T);
    const int cy0 = co0 + piece * EPP;
    yvalid[j] = cy0 < a.cout;
    ybase[j] = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(a.dy) + cy0);
  }
  const long long ystride = (long long)a.cout * (long long)sizeof(T);
  const unsigned char* zero_page = reinterpret_cast<const unsigned char*>(a.zero);

  auto stage = [&](int tile) __attribute__((always_inline)) {
    const int n = tile / tiles_per_sample;
    int t = tile % tiles_per_sample;
    const int par = t / tiles_per_par; t %= tiles_per_par;
    const int bx = t % a.tx; t /= a.tx;
    const int by = t % a.ty;
    const int bz = t / a.ty;
    const int px = par % STEP, py = (par / STEP) % STEP, pz = par / (STEP * STEP);
    const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ;
    // halo coordinate h is inside the volume iff lo <= h <= hi (scalars): lattice index l = origin - HALO + h needs
    // l >= 0 and STEP*l + parity < dim, i.e. l <= floor((dim - 1 - parity) / STEP)  (= (dim - 1 - parity + STEP) / STEP - 1)
    const int zlo = z0 >= HALO ? 0 : HALO - z0, ylo = y0 >= HALO ? 0 : HALO - y0, xlo = x0 >= HALO ? 0 : HALO - x0;
    const int zmax = (a.D - 1 - pz + STEP) / STEP - 1, ymax = (a.H - 1 - py + STEP) / STEP - 1, xmax = (a.W - 1 - px + STEP) / STEP - 1;
    int zsp = zmax - (z0 - HALO) - zlo, ysp = ymax - (y0 - HALO) - ylo, xsp = xmax - (x0 - HALO) - xlo;   // hi - lo
    const bool any = zsp >= 0 && ysp >= 0 && xsp >= 0;
    if (!any) zsp = ysp = xsp = 0;
    // voxel index of halo coordinate (0,0,0); negative on the low borders (those lanes are masked before any access)
    const long long origin = ((long long)(STEP * (z0 - HALO) + pz) * a.H + (STEP * (y0 - HALO) + py)) * a.W + (STEP * (x0 - HALO) + px);
    const long long nofs = (long long)n * V;
    const u32x4 clo = cslot[0], chi = cslot[1];
    const unsigned cw[8] = {clo.x, clo.y, clo.z, clo.w, chi.x, chi.y, chi.z, chi.w};
#pragma unroll
    for (int k = 0; k < X_ITEMS; ++k) {
      const int j = k % PV, g = k / PV;
      if (g < XG) {
        const unsigned c = cw[g >> 1] >> (16 * (g & 1));
        const unsigned cz = c & 3u, cy = (c >> 2) & 7u, cx = (c >> 5) & 63u;
        const bool ok = (int)(any & xvalid[j]) & (int)((c & 0x8000u) == 0) & (int)((cz - (unsigned)zlo) <= (unsigned)zsp) &
                        (int)((cy - (unsigned)ylo) <= (unsigned)ysp) & (int)((cx - (unsigned)xlo) <= (unsigned)xsp);
        const unsigned rel = (unsigned)STEP * ((cz * (unsigned)a.H + cy) * (unsigned)a.W + cx);
        const unsigned char* sb = xbase[j] + (nofs + origin) * xstride[j];   // scalar
        const unsigned char* gp = sb + (unsigned long long)rel * (unsigned long long)(unsigned)xstride[j];
        gp = ok ? gp : zero_page;
        __builtin_amdgcn_global_load_lds((glb_void*)gp, (lds_void*)(xs + (wave_s + NW * j) * XPL + g * 1024), 16, 0, 0);
      }
    }
    // dY tile: voxel (group * 64 + lane) -> (lz, ly, lx) by shifts (TX = 32, TY = 4)
    const long long yorigin = ((long long)(STEP * z0 + pz) * a.H + (STEP * y0 + py)) * a.W + (STEP * x0 + px);
    const int yzsp = zmax - z0, yysp = ymax - y0, yxsp = xmax - x0;
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));   // per-tile copy: keeps the (tile-invariant) offsets below from being hoisted and spilled
#pragma unroll
    for (int k = 0; k < Y_ITEMS; ++k) {
      const int j = k % PV, g = k / PV;
      if (g < YG) {
        const int vox = g * 64 + lane_t;
        const int lx = vox % TX, r2 = vox / TX;
        const int ly = r2 % TY, lz = r2 / TY;
        const bool ok = (int)yvalid[j] & (int)(vox < NVT) & (int)(lz <= yzsp) & (int)(ly <= yysp) & (int)(lx <= yxsp);
        const unsigned rel = (unsigned)STEP * (unsigned)((lz * a.H + ly) * a.W + lx);
        const unsigned char* sb = ybase[j] + (nofs + yorigin) * ystride;     // scalar
        const unsigned char* gp = sb + (unsigned long long)rel * (unsigned long long)(unsigned)ystride;
        gp = ok ? gp : zero_page;
        __builtin_amdgcn_global_load_lds((glb_void*)gp, (lds_void*)(ys + (wave_s + NW * j) * YPL + g * 1024), 16, 0, 0);
      }
    }
  };

  WSTAMP(0);   // prologue
  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    __syncthreads();   // previous tile's reads are done
    WSTAMP(1);   // barrier 1
    stage(tile);
    WSTAMP(2);   // DMA issue
    __syncthreads();   // (waits vmcnt(0): the DMA of every wave has landed)
    WSTAMP(3);   // DMA landing + barrier 2
    if constexpr (sizeof(T) == 2) {
      // MFMA phase, software-pipelined by hand.  One step = one MFMA = (row, 16-voxel K-step kk, tap ti).  Left to the
      // compiler this became read -> s_waitcnt lgkmcnt(0) -> MFMA per step on one set of fragment registers (every MFMA
      // behind a full LDS latency: the phase ran at half speed) plus a separate address register per step.  Here the A
      // fragments of step s + 2 and the B fragments of the next (row, kk) are requested before the MFMA of step s
      // issues (rotating buffers, compile-time indices), and a step's address is "per-tap base register + immediate".
      typedef __attribute__((address_space(3))) bf16x4 lds_b4;
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      constexpr int ROWS = (TAPS == 27) ? TZ * TY : TZ * TY / NW;    // 1x1x1: the rows are split over the waves
      constexpr int NSTEP = ROWS * NT;                               // one step = (row, tap): 2 x 2 MFMAs of 16x16x32
      auto a_off = [](int ri) constexpr {   // byte offset of a row inside the X image, relative to the tap base
        const int row = (TAPS == 27) ? ri : ri * NW;                 // (1x1x1: + wave, folded into the base)
        return ((row / TY * HY + row % TY) * HX) * 16;
      };
      auto b_off = [](int ri) constexpr {
        const int row = (TAPS == 27) ? ri : ri * NW;
        return (row * TX) * 16;
      };
      bf16x4 abuf[3][2][2], bbuf[2][2][2];   // [rotating buffer][16-channel block][first / second 4 voxels]
      auto load_a = [&](auto step_c) __attribute__((always_inline)) {
        constexpr int st = decltype(step_c)::value;
        if constexpr (st < NSTEP) {
          constexpr int ti = st % NT, ri = st / NT;
#pragma unroll
          for (int blk = 0; blk < 2; ++blk) {
            abuf[st % 3][blk][0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(abase[ti] + a_off(ri) + blk * 2 * XPL));
            abuf[st % 3][blk][1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(abase[ti] + a_off(ri) + blk * 2 * XPL + 64));
          }
        }
      };
      auto load_b = [&](auto ri_c) __attribute__((always_inline)) {
        constexpr int ri = decltype(ri_c)::value;
        if constexpr (ri < ROWS) {
#pragma unroll
          for (int blk = 0; blk < 2; ++blk) {
            bbuf[ri % 2][blk][0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(bbase + b_off(ri) + blk * 2 * YPL));
            bbuf[ri % 2][blk][1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(bbase + b_off(ri) + blk * 2 * YPL + 64));
          }
        }
      };
      load_b(std::integral_constant<int, 0>{});
      load_a(std::integral_constant<int, 0>{});
      load_a(std::integral_constant<int, 1>{});
      [&]<int... ST>(std::integer_sequence<int, ST...>) __attribute__((always_inline)) {
        ([&]() __attribute__((always_inline)) {
          constexpr int ti = ST % NT, ri = ST / NT;
          load_a(std::integral_constant<int, ST + 2>{});
          if constexpr (ti == 0) load_b(std::integral_constant<int, ri + 1>{});
          __builtin_amdgcn_sched_barrier(0);   // (the reads stay ahead of this step's MFMAs)
#pragma unroll
          for (int ab = 0; ab < 2; ++ab) {
            const bf16x8 afr = __builtin_shufflevector(abuf[ST % 3][ab][0], abuf[ST % 3][ab][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
              const bf16x8 bfr = __builtin_shufflevector(bbuf[ri % 2][bb][0], bbuf[ri % 2][bb][1], 0, 1, 2, 3, 4, 5, 6, 7);
              const int q = 4 * (2 * ab + bb);           // block (ab, bb) lives in elements q .. q + 3 of acc[ti]
              f32x4 c = {acc[ti][q], acc[ti][q + 1], acc[ti][q + 2], acc[ti][q + 3]};
              if constexpr (std::is_same<T, f16_t>::value) {   // (the transposing LDS read is type-agnostic: 16-bit patterns)
                typedef f16_t f16x8 __attribute__((ext_vector_type(8)));
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, afr), __builtin_bit_cast(f16x8, bfr), c, 0, 0, 0);
              } else {
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr, c, 0, 0, 0);
              }
              acc[ti][q] = c[0]; acc[ti][q + 1] = c[1]; acc[ti][q + 2] = c[2]; acc[ti][q + 3] = c[3];
            }
          }
          __builtin_amdgcn_sched_barrier(0);   // keep the issue order written here
        }(), ...);
      }(std::make_integer_sequence<int, NSTEP>{});
    } else {
      for (int row = 0; row < TZ * TY; ++row) {
        if (TAPS == 1 && (row % NW) != wave) continue;   // 1x1x1: rows are split over the waves
        const int lz = row / TY, ly = row % TY;
        const int xrow = (lz * HY + ly) * HX;             // tap (0,0,0) voxel index of this row in the halo tile
        const int yrow = row * TX;
        const int pl = col >> 2, sub = (col & 3) * 4;   // f32: channel col lives in plane col/4
#pragma unroll 4
        for (int kk = 0; kk < 16; ++kk) {
          const int xo = 2 * kk + h;
          const float bfr = *reinterpret_cast<const float*>(ys + pl * YPL + (yrow + xo) * 16 + sub);
#pragma unroll
          for (int ti = 0; ti < NT; ++ti) {
            const float afr = *reinterpret_cast<const float*>(xs + pl * XPL + (xrow + tapoff[ti] + xo) * 16 + sub);
            acc[ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr, bfr, acc[ti], 0, 0, 0);
          }
        }
      }
    }
    WSTAMP(4);   // MFMA rows of this tile
  }
  // ---- store this workgroup's accumulators: slab[(combo*G + wg) (x4 waves for 1x1)][tap][ci][co] ----
  if (TAPS == 27) {
    float* out = a.slab + ((size_t)combo * gridDim.x + blockIdx.x) * (27 * 1024);
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
      const int tap = wave + NW * ti;
      if (tap < 27) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // f32: 32x32 accumulator (row = ci, col = co).  16-bit: four 16x16 blocks, element r = 4 (2 a + b) + i is
          // ci = 16 a + 4 (lane >> 4) + i, co = 16 b + (lane & 15)
          const int row = sizeof(T) == 4 ? (r & 3) + 8 * (r >> 2) + 4 * h : 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3);
          const int cc = sizeof(T) == 4 ? col : 16 * ((r >> 2) & 1) + (lane & 15);
          out[(tap * 32 + row) * 32 + cc] = acc[ti][r];
        }
      }
    }
  } else {
    // 1x1x1: the waves hold partial sums of the same 32 x 32 tile (the rows of a tile are split over them): summed here
    // through LDS in wave order, so that the reduce kernel reads one slab per workgroup instead of four
    __syncthreads();   // the last tile's reads are done: the images are dead
    float* part = reinterpret_cast<float*>(smem);   // [NW][1024]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = sizeof(T) == 4 ? (r & 3) + 8 * (r >> 2) + 4 * h : 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3);
      const int cc = sizeof(T) == 4 ? col : 16 * ((r >> 2) & 1) + (lane & 15);
      part[wave * 1024 + row * 32 + cc] = acc[0][r];
    }
    __syncthreads();
    float* out = a.slab + ((size_t)combo * gridDim.x + blockIdx.x) * 1024;
    for (int e = tid; e < 1024; e += NW * 64) {
      float t = part[e];
#pragma unroll
      for (int w = 1; w < NW; ++w) t += part[w * 1024 + e];
      out[e] = t;
    }
  }
#ifdef SEUNET_STAMP
  WSTAMP(5);   // slab store
  if (a.debug != nullptr && lane == 0) {
    const size_t w = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NW + wave;
    for (int i = 0; i < 6; ++i) a.debug[w * 12 + i] = ph[i];
    for (int i = 6; i < 12; ++i) a.debug[w * 12 + i] = 0;
  }
#endif
}

// dW (PyTorch layout) = fixed-order sum of the slabs.  A 256-thread block owns 16 consecutive slab elements
// (tap, ci, co); its 16 thread groups each sum every 16th slab (64-B coalesced reads, 16-fold shorter dependent
// chains than one thread per element: that version averaged 37 us per launch, ~1 ms per training step), and the 16
// partial sums are combined in a fixed order through LDS.  f64 throughout; deterministic, no atomics.
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ slab, int nslab, int taps, int cin_w, int cout_w, int co_tiles,
                    float* __restrict__ dw) {
  const int per = taps * 1024;
  const int el = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;             // element inside one slab (per is a multiple of 16)
  const int combo = blockIdx.y;
  const float* p = slab + (size_t)combo * nslab * per + e;
  double s0 = 0.0, s1 = 0.0;
  int k = part;
  for (; k + 16 < nslab; k += 32) {
    s0 += (double)p[(size_t)k * per];
    s1 += (double)p[(size_t)(k + 16) * per];
  }
  if (k < nslab) s0 += (double)p[(size_t)k * per];
  __shared__ double red[16][17];
  red[part][el] = s0 + s1;
  __syncthreads();
  if (part == 0) {
    double t = red[0][el];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red[q][el];
    const int tap = e >> 10, ci = (combo / co_tiles) * 32 + ((e >> 5) & 31), co = (combo % co_tiles) * 32 + (e & 31);
    if (ci < cin_w && co < cout_w) dw[((size_t)co * cin_w + ci) * taps + tap] = (float)t;
  }
}

// persistent workgroups per (ci, co) combo == slabs the reduce kernel has to sum; ~2 resident workgroups per CU in total
static inline int wgrad_groups(int taps, int combos, int total_tiles) {
  static const int per_chip = [] { const char* e = getenv("SEUNET_WGRAD_WGS"); return e ? atoi(e) : 512; }();   // (diagnostic)
  int g = per_chip / combos;   // two resident workgroups per CU
  if (g < 16) g = 16;
  if (g > 512) g = 512;
  if (g > total_tiles) g = total_tiles;
  if (g < 1) g = 1;
  (void)taps;
  return g;
}

size_t wgrad_workspace_bytes(int taps, int cin, int cout) {
  const int combos = cdiv(cin, 32) * cdiv(cout, 32);
  const int g = 512 / combos < 16 ? 16 : (512 / combos > 512 ? 512 : 512 / combos);
  size_t bytes = 256 + (size_t)combos * g * taps * 1024 * sizeof(float);   // 256 zero bytes + slabs
  if (taps == 1) {   // the whole-GEMM 1x1x1 kernel (wgrad_1x1.hip): 256 workgroups x all 16 x 16 pairs
    const size_t b1 = 256 + (size_t)256 * (size_t)(cdiv(cin, 64) * 4) * (size_t)cdiv(cout, 16) * 256 * sizeof(float);
    if (b1 > bytes) bytes = b1;
  }
  return bytes;
}

template <typename T, int TAPS, int DIL, int NW>
static int wgrad_launch_one(const WgArgs& a, dim3 grid, hipStream_t s) {
  constexpr int HALO = (TAPS == 27) ? 1 : 0;
  constexpr int TZ = WgTile<T>::TZ, TY = WgTile<T>::TY;
  constexpr int PPV = 32 / (16 / (int)sizeof(T));
  constexpr int NVH = (TZ + 2 * HALO) * (TY + 2 * HALO) * (32 + 2 * HALO), NVT = TZ * TY * 32;
  constexpr int LDS = PPV * ((((NVH + 63) / 64) * 64 + 4) + (((NVT + 63) / 64) * 64 + 4)) * 16 + NW * 64 * 32;   // + packed halo coordinates
  static unsigned long long configured = 0;   // per instantiation: devices on which the LDS limit was raised
  if (int e = configure_kernel_lds(configured, reinterpret_cast<const void*>(&wgrad_kernel<T, TAPS, DIL, NW>), LDS)) return e;
  wgrad_kernel<T, TAPS, DIL, NW><<<grid, NW * 64, LDS, s>>>(a);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

// x: input activation (may be a concatenation), cin_logical leading channels carry weights;
// dy: gradient w.r.t. the raw conv output, [N][V][cout]; dw: (cout, cin_logical, taps) f32, overwritten.
int launch_wgrad(int dtype, int taps, int dil, const SrcList& x, int cin_logical, const void* dy, int cout,
                 float* dw, void* workspace, size_t ws_bytes, Dims d, hipStream_t s, bool allow_march) {
  SEUNET_CHECK(taps == 27 || taps == 1, "wgrad: taps=%d unsupported", taps);
  SEUNET_CHECK(taps == 1 || dil == 1 || dil == 2, "wgrad: dilation %d unsupported", dil);
  SEUNET_CHECK(x.n >= 1 && x.n <= 3, "wgrad: 1..3 sources");
  SEUNET_CHECK(cout % 8 == 0 && cin_logical >= 1 && cin_logical <= x.total(), "wgrad: bad channel counts");
  SEUNET_CHECK(ws_bytes >= wgrad_workspace_bytes(taps, cin_logical, cout), "wgrad: workspace too small");
  if (allow_march && taps == 1 && wgrad_1x1_supported(dtype, x, cin_logical, cout, d))   // aggregation convs: wgrad_1x1.hip
    return launch_wgrad_1x1(dtype, x, cin_logical, dy, cout, dw, workspace, ws_bytes, d, s);
  if (allow_march && wgrad_march_supported(dtype, taps, dil, x, cin_logical, cout, d))     // wide layers of the fine levels: wgrad_march.hip
    return launch_wgrad_march(dtype, taps, dil, x, cin_logical, dy, cout, dw, workspace, ws_bytes, d, s);
  WgArgs a{};
  a.src0 = x.ptr[0]; a.srcC0 = x.C[0];
  a.src1 = x.n > 1 ? x.ptr[1] : nullptr; a.srcC1 = x.n > 1 ? x.C[1] : 0;
  a.src2 = x.n > 2 ? x.ptr[2] : nullptr; a.srcC2 = x.n > 2 ? x.C[2] : 0;
  a.cin = cin_logical;
  a.cum1 = x.n > 1 ? x.C[0] : x.total();
  a.cum2 = x.n > 2 ? x.C[0] + x.C[1] : x.total();
  a.dy = dy; a.cout = cout;
  a.zero = device_zero_page();
  SEUNET_CHECK(a.zero != nullptr, "wgrad: cannot allocate the device zero page");
  a.slab = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + 256);
  a.N = d.N; a.D = d.D; a.H = d.H; a.W = d.W;
  const int tz = dtype_size(dtype) == 2 ? WgTile<bf16_t>::TZ : WgTile<float>::TZ;
  const int ty = dtype_size(dtype) == 2 ? WgTile<bf16_t>::TY : WgTile<float>::TY;
  const int st = taps == 27 ? dil : 1;
  a.tx = cdiv(cdiv(d.W, st), 32); a.ty = cdiv(cdiv(d.H, st), ty); a.tz = cdiv(cdiv(d.D, st), tz);
  a.co_tiles = cdiv(cout, 32);
  a.debug = g_conv_debug;
  const int combos = cdiv(cin_logical, 32) * a.co_tiles;
  const int G = wgrad_groups(taps, combos, a.tx * a.ty * a.tz * st * st * st * d.N);
  dim3 grid(G, combos);
  int e;
  SEUNET_DTYPE_SWITCH(dtype, {
    if (taps == 1) e = wgrad_launch_one<T, 1, 1, 4>(a, grid, s);
    else if (dil == 1) e = wgrad_launch_one<T, 27, 1, 4>(a, grid, s);
    else e = wgrad_launch_one<T, 27, 2, 4>(a, grid, s);
  });
  if (e) return e;
  wgrad_reduce_kernel<<<dim3(taps * 1024 / 16, combos), 256, 0, s>>>(a.slab, G, taps, cin_logical, cout,
                                                                            a.co_tiles, dw);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// naive weight gradient: one thread per weight element (device-side cross-check, small sizes only)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void wgrad_naive_kernel(WgArgs a, int taps, int dil, float* __restrict__ dw, long long total) {
  const long long V = (long long)a.D * a.H * a.W;
  const int t3 = taps == 27 ? 3 : 1, c = taps == 27 ? 1 : 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const int ci = (int)((i / taps) % a.cin);
    const int co = (int)(i / ((long long)taps * a.cin));
    const int dz = (tap / (t3 * t3) - c) * dil, dy = ((tap / t3) % t3 - c) * dil, dx = (tap % t3 - c) * dil;
    const void* sp = a.src0; int sC = a.srcC0, cc = ci;
    if (ci >= a.cum2) { sp = a.src2; sC = a.srcC2; cc = ci - a.cum2; }
    else if (ci >= a.cum1) { sp = a.src1; sC = a.srcC1; cc = ci - a.cum1; }
    double s = 0.0;
    for (int n = 0; n < a.N; ++n)
      for (int z = 0; z < a.D; ++z) {
        const int zz = z + dz;
        if ((unsigned)zz >= (unsigned)a.D) continue;
        for (int y = 0; y < a.H; ++y) {
          const int yy = y + dy;
          if ((unsigned)yy >= (unsigned)a.H) continue;
          for (int x = 0; x < a.W; ++x) {
            const int xx = x + dx;
            if ((unsigned)xx >= (unsigned)a.W) continue;
            const float xv = to_f32(reinterpret_cast<const T*>(sp)[(n * V + ((long long)zz * a.H + yy) * a.W + xx) * sC + cc]);
            const float gv = to_f32(reinterpret_cast<const T*>(a.dy)[(n * V + ((long long)z * a.H + y) * a.W + x) * a.cout + co]);
            s += (double)xv * (double)gv;
          }
        }
      }
    dw[i] = (float)s;
  }
}

int launch_wgrad_naive(int dtype, int taps, int dil, const SrcList& x, int cin_logical, const void* dy, int cout,
                       float* dw, Dims d, hipStream_t s) {
  WgArgs a{};
  a.src0 = x.ptr[0]; a.srcC0 = x.C[0];
  a.src1 = x.n > 1 ? x.ptr[1] : nullptr; a.srcC1 = x.n > 1 ? x.C[1] : 0;
  a.src2 = x.n > 2 ? x.ptr[2] : nullptr; a.srcC2 = x.n > 2 ? x.C[2] : 0;
  a.cin = cin_logical;
  a.cum1 = x.n > 1 ? x.C[0] : x.total();
  a.cum2 = x.n > 2 ? x.C[0] + x.C[1] : x.total();
  a.dy = dy; a.cout = cout;
  a.N = d.N; a.D = d.D; a.H = d.H; a.W = d.W;
  const long long total = (long long)cout * cin_logical * taps;
  const int grid = (int)((total + 63) / 64);
  SEUNET_DTYPE_SWITCH(dtype, wgrad_naive_kernel<T><<<grid, 64, 0, s>>>(a, taps, dil, dw, total));
  SEUNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace seunet
