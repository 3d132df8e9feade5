// Streaming 3x3x3 convolution for the small-channel full-resolution layers (reference nn.Conv3d at SE_UNet.py:15 for
// ec1 / ec2 / ec3 / dc6, forward and data gradient): 8..32 input channels, <= 32 output channels, dilation 1 or 2.
//
// Why a second conv kernel: on these layers the tiled implicit-GEMM kernel (conv_igemm.hip) spends ~8x its MFMA time per
// tile on fixed work (index plan, first fetch at full HBM latency, statistics tree, stores) with two workgroups per CU to
// hide it -- 0.29..0.69 ms per launch against HBM floors of 0.05..0.15 ms (profiles/r01_g_*).  Here a workgroup owns an
// 8 x 32 (y, x) patch and MARCHES along z:
//   * input-stationary in z: step s stages ONE input plane and adds its contribution to the three output planes s-1, s,
//     s+1 (accumulators of three planes live in registers, roles rotate with a 3x unrolled loop), so every LDS fragment
//     read feeds 3 (dz) MFMAs instead of one, and the finished plane s-1 is stored while the march goes on; 8 waves, one
//     output row each, two per SIMD.  (Round 4, measured: every variant but the 8-channel one needs > 128 registers per lane,
//     so ONE workgroup is resident per CU and its two waves per SIMD run in phase -- the barrier lines them up -- and hide
//     nothing of each other; hence marches as long as the volume allows, see stream_zsteps, and DESIGN 4 for the timeline);
//   * the planes arrive by LDS-DMA (buffer_load_dwordx4 ... lds through one descriptor per sample, inline asm so that hipcc
//     neither counts nor drains them; padding lanes and planes outside the march read as zeros) into a ring two or three
//     steps ahead of their use: counted s_waitcnt vmcnt(N) + one raw s_barrier per step;
//   * weights live in registers for the whole march (27 taps x 16 B per lane), no weight LDS, no K-chunk loop;
//   * InstanceNorm partial sums are carried per lane across the march (f32) and reduced once per workgroup (f64), not once
//     per 512-voxel tile;
//   * finished rows of the 32-channel tiles leave through a wave-private LDS row buffer as contiguous stores (round 4).
// MFMA shapes by channel counts (weights = A operand, so a lane owns one voxel and runs of 4 consecutive channels):
//   CIN 16 -> COUT <= 32 : v_mfma_f32_32x32x16 (K = the 16 channels of one tap)                       ec3 fwd, dc6 dgrad, ec2 dgrad
//   CIN 32 -> COUT <= 16 : v_mfma_f32_16x16x32 (K = the 32 channels of one tap)                       dc6 fwd, ec3 dgrad
//   CIN  8 -> COUT <= 16 : v_mfma_f32_16x16x32 with the three x-taps folded into K (3 x 8 channels of the contiguous
//                          voxels x-1, x, x+1 + 8 zero weights): 9 MFMAs per 16 voxels instead of 27   ec1 fwd, ec2 fwd
//   CIN 16 -> COUT <= 16 : the same with two x-taps per MFMA (2 x 16 channels): 18 MFMAs per 16 voxels       ec2 dgrad
// Dilation 2 marches each z-parity class separately (planes z, z+2, ...) and keeps the natural layout in the plane (halo 2).
// LDS image of a plane: planar [16-B piece of the channels][voxel of the halo patch][16 B]; fragment reads are ds_read_b128
// of consecutive voxels (conflict-free), DMA instructions write 1 KB contiguous.
#include "seunet_common.h"
#include <utility>
#include <type_traits>

namespace seunet {

extern unsigned long long* g_conv_debug;   // conv_igemm.hip (diagnostic builds)

typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
typedef f16_t f16x8s __attribute__((ext_vector_type(8)));
// fragments are 8 x 16-bit patterns; the matrix instruction follows the storage type (bf16 | f16)
template <typename T> __device__ __forceinline__ f32x16 st_mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (std::is_same<T, f16_t>::value) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8s, a), __builtin_bit_cast(f16x8s, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ f32x4 st_mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (std::is_same<T, f16_t>::value) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8s, a), __builtin_bit_cast(f16x8s, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

struct StreamArgs {
  const void* src; const void* wpack; const float* bias;
  void* dst; int dstC; int dacc; int cout; int stage;
  double* stats; const void* zero;
  unsigned long long* debug;       // diagnostic builds only (-DSEUNET_STAMP): [workgroup][wave][12] cycle sums per phase
  int N, D, H, W;
  int nyb, nxb, nzseg, zsteps;      // patches, z segments (per parity class), output planes per segment
};

#ifndef SEUNET_STREAM_ROWS
#define SEUNET_STREAM_ROWS 8
#endif
static constexpr int ST_TY = SEUNET_STREAM_ROWS, ST_TX = 32, ST_NW = SEUNET_STREAM_ROWS;   // 8 waves: one output row each, two waves per SIMD

template <int CIN, int COUTP, bool XFOLD, int DIL, bool DACC = false> struct StreamGeo {
  static constexpr int NP = CIN / 8;
  static constexpr int VPK = XFOLD ? 32 / CIN : 1;             // x-taps folded into the K = 32 of one 16x16x32 MFMA (CIN 8: 4 slots, CIN 16: 2)
  static constexpr int NDX = XFOLD ? (3 + VPK - 1) / VPK : 3, NTAP = 9 * NDX;
  static constexpr int HX = ST_TX + 2 * DIL + (NDX * VPK - 3) * DIL, HY = ST_TY + 2 * DIL;   // (+ the columns the padding slots read)
  static constexpr int NVP = HX * HY, G = (NVP + 63) / 64;
  static constexpr int PS = G * 1024, PLANE = NP * PS;
  static constexpr int OLDI = COUTP == 32 ? 2 : 1;             // DMA instructions per old destination row (32 voxels x <= 32 channels x 2 B <= 2 KB)
  // planes in flight ahead of the one being read: 3 where the 160 KB of LDS allow it; ring = those + the current one + the one
  // released by the previous step
  static constexpr int STG = COUTP == 32 ? 2048 : 0;           // output row staging per wave (32-channel tiles: 32 voxels x 64 B)
  static constexpr int PF = (5 * PLANE + 1280 + ST_NW * STG + (DACC ? 5 * ST_NW * OLDI * 1024 : 0) <= 160 * 1024) ? 3 : 2;
  static constexpr int RING = PF + 2;
  static constexpr int ITEMS = (NP * G + ST_NW - 1) / ST_NW;     // DMA wave-instructions per wave and plane
  static constexpr int NB = COUTP == 32 ? 32 : 16, NBX = ST_TX / NB;
  static constexpr int STORES = COUTP == 32 ? 4 : NBX;           // store wave-instructions per wave and step (direct and staged form alike)
  static constexpr int BIAS = RING * PLANE + 1024;          // f32 bias table (32-channel tiles keep it here instead of 16 registers per lane)
  static constexpr int STAGE = BIAS + 256;                     // output rows on their way out: [8 waves][STG]
  static constexpr int OLD = STAGE + ST_NW * STG;              // old destination rows (gradient accumulation): [ring slots][8 waves][OLDI KB]
  static constexpr int LDS = OLD + (DACC ? RING * ST_NW * OLDI * 1024 : 0);   // ring + a 1-KB dump for the padding DMA instructions (+ old rows)
};

__device__ __forceinline__ void stream_dma16(const void* gsrc, unsigned lds_dst) {
  // one LDS-DMA wave-instruction: 64 lanes x 16 B, LDS destination = lds_dst + 16 * lane (M0 carries the base)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// The same through a buffer descriptor (round 4): the 16 bytes of a lane come from base + soffset + voffset, and a lane whose
// voffset lies beyond num_records writes ZEROS into the LDS (probed on gfx950: scripts/probes/blds_oob.hip).  So the padding of a
// plane image needs neither a zero page nor a per-lane pointer select nor a 64-bit address: the lane's offset inside a plane is
// a constant (0xFFFFFFFF for padding voxels), the plane's offset inside the sample a scalar, a plane outside the volume a
// descriptor of zero records.  One vector register, no vector arithmetic per DMA instruction.
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_dma16_buf(unsigned voff, u32x4s rsrc, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
// -DSEUNET_STREAM_PROBE=1: no statistics; 2: no stores; 3: no MFMAs; 4: no fragment reads (timing by elimination, never shipped)
#ifndef SEUNET_STREAM_PROBE
#define SEUNET_STREAM_PROBE 0
#endif
#ifdef SEUNET_STAMP
#define SSTAMP(i) do { const unsigned long long _t = __builtin_readcyclecounter(); ph[i] += _t - t_last; t_last = _t; } while (0)
#else
#define SSTAMP(i) do {} while (0)
#endif
template <int N> __device__ __forceinline__ void stream_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// FWD: bias + InstanceNorm partial sums (forward); !FWD: data gradient, optionally accumulating into the destination (DACC)
template <typename T, int CIN, int COUTP, bool XFOLD, int DIL, bool FWD, bool DACC>
__global__ void __launch_bounds__(ST_NW * 64, 2)
conv_stream_kernel(StreamArgs a) {
#ifdef SEUNET_STAMP
  unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_readcyclecounter();
#endif
  using Geo = StreamGeo<CIN, COUTP, XFOLD, DIL, DACC>;
  constexpr int NP = Geo::NP, HX = Geo::HX, NVP = Geo::NVP, G = Geo::G, PS = Geo::PS, PLANE = Geo::PLANE;
  constexpr int ITEMS = Geo::ITEMS, NB = Geo::NB, NBX = Geo::NBX, NDX = Geo::NDX, NTAP = Geo::NTAP;
  constexpr int ACCR = COUTP == 32 ? 16 : 4;
  typedef typename std::conditional<COUTP == 32, f32x16, f32x4>::type AccT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // patch / segment of this workgroup (XCD-contiguous order of the patches)
  int t;
  {
    const int nt = gridDim.x, b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int xb = t % a.nxb, yb = t / a.nxb;
  const int seg = blockIdx.y / DIL, pz = blockIdx.y % DIL;       // z = pz + DIL * (plane index in the parity class)
  const int n = blockIdx.z;
  const int x0 = xb * ST_TX, y0 = yb * ST_TY;
  const int q0 = seg * a.zsteps;                                   // first output plane (parity units)
  const int nplanes = (a.D - pz + DIL - 1) / DIL;                  // planes of this parity class
  const int q1 = min(q0 + a.zsteps, nplanes);
  const int nsteps = q1 - q0 + 2;                                  // input planes q0-1 .. q1
  const long long plane_bytes = (long long)a.H * a.W * CIN * (long long)sizeof(T);
  const unsigned char* src_n = reinterpret_cast<const unsigned char*>(a.src) + (long long)n * a.D * plane_bytes;

  // ---- DMA plan: item it of this wave = (piece, 64-voxel group) number wave + 8 * it ----
  unsigned doff[ITEMS];            // byte offset inside a z-plane of this lane's 16 bytes; 0xFFFFFFFF = padding (zero page)
  unsigned dlds[ITEMS];            // LDS byte offset inside a plane image (wave-uniform)
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int id = wave + ST_NW * it;
    const int p = id / G, gi = id % G;
    const int v = gi * 64 + lane;
    const int hy = v / HX, hx = v % HX;
    const int y = y0 - DIL + hy, x = x0 - DIL + hx;
    const bool ok = id < NP * G && v < NVP && y >= 0 && y < a.H && x >= 0 && x < a.W;
    doff[it] = ok ? (unsigned)(((y * a.W + x) * CIN + p * 8) * (int)sizeof(T)) : 0xFFFFFFFFu;
    dlds[it] = (unsigned)(p * PS + gi * 1024);
  }
  SSTAMP(8);       // (prologue) DMA plan
  const unsigned char* zero_page = reinterpret_cast<const unsigned char*>(a.zero) + lane * 16;     // (old destination rows only)
  // source descriptor of this sample: base (scalar registers), num_records = the sample's bytes (0 for a plane outside the volume)
  unsigned src_lo, src_hi;
  {
    const unsigned long long u = reinterpret_cast<unsigned long long>(src_n);
    src_lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    src_hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) & 0xFFFFu;
  }
  const unsigned sample_bytes = (unsigned)((long long)a.D * plane_bytes);      // < 2^32 (checked by the launcher)
  // plane of step s -> ring slot `slot` (= s % Geo::RING).  The plane-level part (validity, offset of the plane inside the sample,
  // LDS slot base) is computed once per step (`plane_of`) and handed to the items
  struct PlaneRef { unsigned soff; unsigned nrec; unsigned lds; };
  auto plane_of = [&](int s, int slot) __attribute__((always_inline)) -> PlaneRef {
    PlaneRef r;
    const int pl = q0 - 1 + s;                                      // plane index in the parity class
    const int z = pz + DIL * pl;
    const bool ok = pl >= 0 && z < a.D && s < nsteps;               // wave-uniform
    r.soff = (unsigned)((long long)(ok ? z : 0) * plane_bytes);
    r.nrec = ok ? sample_bytes : 0u;
    r.lds = lds_base + (unsigned)(slot * PLANE);
    return r;
  };
  auto dma_item = [&](const PlaneRef& r, auto it_c) __attribute__((always_inline)) {
    constexpr int it = decltype(it_c)::value;
    if constexpr (it < ITEMS) {
      // every wave issues exactly ITEMS instructions per plane (the vmcnt arithmetic of the march counts on it): an item
      // number beyond the plane's NP * G pieces reads nothing (zero records) into the dump area
      const bool real = wave + ST_NW * it < NP * G;                 // wave-uniform
      u32x4s rs;
      rs.x = src_lo; rs.y = src_hi; rs.z = real ? r.nrec : 0u; rs.w = 0x00020000u;
      stream_dma16_buf(doff[it], rs, r.soff, real ? r.lds + dlds[it] : lds_base + (unsigned)(Geo::RING * PLANE));
    }
  };
  auto dma_plane = [&](int s, int slot) __attribute__((always_inline)) {
    const PlaneRef r = plane_of(s, slot);
    [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
      (dma_item(r, std::integral_constant<int, I>{}), ...);
    }(std::make_integer_sequence<int, ITEMS>{});
  };

  // gradient accumulation (DACC): the destination row this wave will finish at step s arrives by one more DMA instruction,
  // issued with the plane of step s (two steps ahead) into a wave-private 1-KB slot; no register load sits in the march
  const int old_row_bytes = ST_TX * a.dstC * (int)sizeof(T);        // <= OLDI KB
  auto dma_old = [&](int s, int slot) __attribute__((always_inline)) {
    if constexpr (DACC) {
      const int q = q0 + s - 2, z = pz + DIL * q, y = y0 + wave;
      const bool rok = q >= q0 && q < q1 && y < a.H;                // wave-uniform
      const unsigned char* row = reinterpret_cast<const unsigned char*>(a.dst) +
                                 ((((long long)n * a.D + (rok ? z : 0)) * a.H + (rok ? y : 0)) * a.W + x0) * a.dstC * (long long)sizeof(T);
#pragma unroll
      for (int k = 0; k < Geo::OLDI; ++k) {
        const int byte = k * 1024 + lane * 16;
        const int xv = x0 + byte / (a.dstC * (int)sizeof(T));
        const unsigned char* gp = (rok && byte < old_row_bytes && xv < a.W) ? row + byte : zero_page;
        stream_dma16(gp, lds_base + (unsigned)(Geo::OLD + ((slot * ST_NW + wave) * Geo::OLDI + k) * 1024));
      }
    }
  };

  // ---- weights: registers for the whole march.  wpack: [tap][lane][8 elements] ----
  bf16x8 wreg[NTAP];
  {
    const uint4* wp = reinterpret_cast<const uint4*>(a.wpack) + lane;
#pragma unroll
    for (int k = 0; k < NTAP; ++k) wreg[k] = __builtin_bit_cast(bf16x8, wp[k * 64]);
  }
  // ---- fragment geometry ----
  // 16x16x32 (COUTP 16): lane = (voxel n = lane & 15, k-group g = lane >> 4): piece g of the voxel (XFOLD: the only piece of voxel + g)
  // 32x32x16 (COUTP 32): lane = (voxel n = lane & 31, k-half h = lane >> 5): piece h of the voxel
  const int fn = COUTP == 32 ? (lane & 31) : (lane & 15);
  const int fg = COUTP == 32 ? (lane >> 5) : (lane >> 4);
  // output row of this wave (two waves per SIMD: while one waits for fragments, stores or the barrier the other issues MFMAs)
  const int ra = wave;
  // LDS byte offset of the lane's fragment for input row index ri = 0..2 (y = ra + DIL * (ri - 1)), x-block xbk, tap dx:
  //   ((ra + DIL*ri) * HX + xbk*NB + fn + DIL*(dx+1)) * 16 + piece * PS      (halo origin = -DIL in y and x)
  const int frag0 = XFOLD ? ((ra * HX + fn + DIL * (fg / NP)) * 16 + (fg % NP) * PS) : ((ra * HX + fn) * 16 + fg * PS);

  AccT acc[3][NBX];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int b = 0; b < NBX; ++b)
#pragma unroll
      for (int e = 0; e < ACCR; ++e) acc[i][b][e] = 0.f;

  // channel of accumulator register e of this lane
  auto chan = [&](int e) __attribute__((always_inline)) -> int {
    if constexpr (COUTP == 32) return (e & 3) + 8 * (e >> 2) + 4 * fg;
    else return 4 * fg + e;
  };
  constexpr int SR = FWD ? ACCR : 1;
  constexpr int BR = (FWD && COUTP == 16) ? ACCR : 1;     // bias in registers (16-channel tiles) or in LDS (32-channel tiles)
  float bias_r[BR];
#pragma unroll
  for (int e = 0; e < BR; ++e) { const int c = chan(e); bias_r[e] = (FWD && a.bias != nullptr && c < a.cout) ? a.bias[c] : 0.f; }
  if constexpr (FWD && COUTP == 32) {
    if (tid < 32) reinterpret_cast<float*>(smem + Geo::BIAS)[tid] = (a.bias != nullptr && tid < a.cout) ? a.bias[tid] : 0.f;
    __syncthreads();
  }
  // running InstanceNorm sums of this lane (f32: one value per plane of the march, at most ~130 per lane and channel at 128^3; the per-lane totals are combined in
  // f64.  bf16 activations keep 8 mantissa bits, the gate for this mode is the bf16-autocast comparison)
  float s1[SR], s2[SR];
#pragma unroll
  for (int e = 0; e < SR; ++e) { s1[e] = 0.f; s2[e] = 0.f; }

  // destination: buffer descriptor over this sample (range-checked 32-bit offsets).  The lane's part of a store offset -- row y
  // of this wave, voxel x, run of 4 channels; 0x80000000 = outside the volume / beyond the last channel: dropped by the range
  // check -- is a constant of the march (one register per store instruction), the plane's part a scalar, a plane outside the march
  // a descriptor of zero records: no vector arithmetic per store (round 4; the marching kernel's scheme)
  const long long dst_sample = (long long)a.D * a.H * a.W * a.dstC * (long long)sizeof(T);
  unsigned char* const dst_n = reinterpret_cast<unsigned char*>(a.dst) + (long long)n * dst_sample;
  const int dst_plane = __builtin_amdgcn_readfirstlane(a.H * a.W * a.dstC * (int)sizeof(T));
  bool okl[NBX];
  unsigned lofs[NBX][ACCR / 4];
#pragma unroll
  for (int b = 0; b < NBX; ++b) {
    const int y = y0 + ra, x = x0 + b * NB + fn;
    okl[b] = y < a.H && x < a.W;
#pragma unroll
    for (int pc = 0; pc < ACCR / 4; ++pc) {
      const int c0 = chan(4 * pc);
      lofs[b][pc] = (okl[b] && c0 < a.cout) ? (unsigned)(((y * a.W + x) * a.dstC + c0) * (int)sizeof(T)) : 0x80000000u;
    }
  }

  // Staged stores (round 4), 32-channel tiles.  A lane of an MFMA tile owns one voxel and runs of 4 channels, so its natural store is
  // 8 bytes at a stride of one 64-byte voxel record: 64 different segments per store instruction, which the address path of the
  // CU takes one by one -- timing by elimination put 0.085 ms of the 0.31 ms of the ec3 forward on the stores.  When the
  // destination holds exactly 32 channels the finished row (32 voxels x 64 B = 2 KB, contiguous in memory) goes through a
  // wave-private LDS buffer instead: 8-byte writes in the tile's layout (16-byte slots XOR-swizzled by the voxel number, so that
  // every bank pair takes 4 lanes), 8-byte reads in memory order, four stores of 512 contiguous bytes.  ec3 forward 0.299 ->
  // 0.255 ms, dc6 data gradient 0.32 -> 0.294.  (The 16-channel tiles keep the direct form: their records are 32 bytes, one
  // instruction already covers 512 dense bytes, and staging cost them 10 %.  16-byte stores -- two per row -- were tried from
  // registers after a v_permlane32_swap exchange and from this buffer with ds_read_b128; both wrote wrong rows now and then, which
  // turned out to be the counted wait of the march, not the stores: see there.  With the wait fixed they are correct and no faster
  // than the four 8-byte stores, which stay.)
  const bool staged = COUTP == 32 && a.stage != 0 && a.dstC == COUTP && a.cout == COUTP;         // wave-uniform
  constexpr int NSTG = COUTP == 32 ? 4 : 2, SPB = 8; // 8-byte pieces per lane of the row image
  const unsigned stg_base = (unsigned)(Geo::STAGE + wave * Geo::STG);
  unsigned stg_w[NBX][ACCR / 4];      // LDS byte address of the lane's 8-byte pieces
  unsigned stg_r[NSTG];               // LDS byte address of the lane's 16 bytes of the row image
  unsigned stg_o[NSTG];               // their offset inside a destination plane (0x80000000: outside the volume)
#pragma unroll
  for (int b = 0; b < NBX; ++b)
#pragma unroll
    for (int pc = 0; pc < ACCR / 4; ++pc) {
      const int v = b * NB + fn;
      if constexpr (COUTP == 32) stg_w[b][pc] = stg_base + (unsigned)(v * 64 + 16 * (pc ^ ((v >> 1) & 3)) + 8 * fg);
      else stg_w[b][pc] = stg_base + (unsigned)(v * 32 + 8 * fg);
    }
#pragma unroll
  for (int k = 0; k < NSTG; ++k) {
    const int g = k * 64 * SPB + lane * SPB;                      // byte of the row image
    const int v = g / (COUTP * 2);
    if constexpr (COUTP == 32) stg_r[k] = stg_base + (unsigned)(v * 64 + 16 * (((g >> 4) & 3) ^ ((v >> 1) & 3)) + (g & 8));
    else stg_r[k] = stg_base + (unsigned)g;
    const int y = y0 + ra, x = x0 + v;
    stg_o[k] = (y < a.H && x < a.W) ? (unsigned)(((y * a.W + x0) * COUTP) * (int)sizeof(T) + g) : 0x80000000u;
  }
  SSTAMP(9);       // (prologue) weight loads issued, fragment / store geometry
  float mk[NBX];
#pragma unroll
  for (int b = 0; b < NBX; ++b) mk[b] = okl[b] ? 1.f : 0.f;
  // ---- one step: input plane of step s (ring slot `slot`) -> accumulators; PH = s % 3 ----
  // The fragments of input row ri + 1 are requested before the MFMAs of row ri issue (two register sets), and the prefetch
  // DMA instructions of the plane Geo::PF steps ahead are spread over the rows.
  auto compute = [&](int s, int slot, int slot_pf, auto ph_c) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_c)::value;
    const unsigned char* pl = smem + slot * PLANE;
    const PlaneRef pref = plane_of(s + Geo::PF, slot_pf);
    bf16x8 fr[2][NDX][NBX];
    auto load_row = [&](auto ri_c) __attribute__((always_inline)) {
      constexpr int ri = decltype(ri_c)::value;
      if constexpr (ri < 3) {
#pragma unroll
        for (int dxi = 0; dxi < NDX; ++dxi)
#pragma unroll
          for (int b = 0; b < NBX; ++b) {
            const int off = frag0 + ((DIL * ri) * HX + b * NB + DIL * Geo::VPK * dxi) * 16;
            if constexpr (SEUNET_STREAM_PROBE == 4) fr[ri & 1][dxi][b] = wreg[(ri * NDX + dxi) % NTAP];
            else fr[ri & 1][dxi][b] = *reinterpret_cast<const bf16x8*>(pl + off);
          }
      }
    };
    load_row(std::integral_constant<int, 0>{});
    AccT cinit;
    if constexpr (FWD && COUTP == 32) {   // 16 bias values: four 16-byte reads of the LDS table per step (no registers to keep them)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(smem + Geo::BIAS + (8 * q + 4 * fg) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) cinit[4 * q + e] = bq[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < ACCR; ++e) cinit[e] = (FWD && COUTP == 16) ? bias_r[BR == 1 ? 0 : e] : 0.f;
    }
    [&]<int... RI>(std::integer_sequence<int, RI...>) __attribute__((always_inline)) {
      ([&]() __attribute__((always_inline)) {
        constexpr int ri = RI;                             // input row y = ra + DIL * (ri - 1): tap dy = ri - 1
        load_row(std::integral_constant<int, ri + 1>{});
        [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
          (((I % 3) == ri ? dma_item(pref, std::integral_constant<int, I>{}) : (void)0), ...);
        }(std::make_integer_sequence<int, ITEMS>{});
        if (ri == 2) dma_old(s + Geo::PF, slot_pf);
        __builtin_amdgcn_sched_barrier(0);                 // (the next row's reads stay ahead of this row's MFMAs)
#pragma unroll
        for (int dxi = 0; dxi < NDX; ++dxi)
#pragma unroll
          for (int dz = -1; dz <= 1; ++dz) {               // output plane s - dz
            const int ai = (PH - dz + 3) % 3;
            const int tap = ((dz + 1) * 3 + ri) * NDX + dxi;
            // the set that the previous step finished starts over here (tap dz = -1 of the first row): its first MFMA takes
            // the bias vector (forward) / zero as C, so nothing is ever zeroed and the epilogue adds no bias
            const bool first = dz == -1 && ri == 0 && dxi == 0;
#pragma unroll
            for (int b = 0; b < NBX; ++b) {
              if constexpr (SEUNET_STREAM_PROBE == 3) {
                if (first) acc[ai][b] = cinit;
                acc[ai][b][0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4s, fr[ri & 1][dxi][b]).x);   // (keeps the reads alive)
              } else if constexpr (COUTP == 32) acc[ai][b] = st_mfma32<T>(wreg[tap], fr[ri & 1][dxi][b], first ? cinit : acc[ai][b]);
              else acc[ai][b] = st_mfma16<T>(wreg[tap], fr[ri & 1][dxi][b], first ? cinit : acc[ai][b]);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }(), ...);
    }(std::make_integer_sequence<int, 3>{});
  };

  // ---- epilogue of the finished output plane q = (q0 - 1 + s) - 1 held in acc[(PH + 2) % 3] ----
  auto finish = [&](int s, int slot, auto ph_c) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_c)::value;
    constexpr int ai = (PH + 2) % 3;
    const int q = q0 + s - 2;                                       // parity-class plane index
    const int z = pz + DIL * q;
    const bool zok = q >= q0 && q < q1;                             // wave-uniform
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst_n, 0, zok ? (int)dst_sample : 0, 0x00020000);
    const int soff = __builtin_amdgcn_readfirstlane(zok ? z * dst_plane : 0);
#pragma unroll
    for (int b = 0; b < NBX; ++b) {
      float v[ACCR];
#pragma unroll
      for (int e = 0; e < ACCR; ++e) v[e] = acc[ai][b][e];       // (bias included: it was the C operand of the set's first MFMA)
      if constexpr (FWD && SEUNET_STREAM_PROBE != 1) {
        // branch-free (a lane outside the volume or a plane outside the march adds v * 0; a.stats == nullptr: the sums are simply
        // never stored): fma(v, 1, s) and fma(v * 1, v, s) round exactly like s + v and fma(v, v, s)
        const float m = zok ? mk[b] : 0.f;
#pragma unroll
        for (int e = 0; e < ACCR; ++e) { s1[e] = fmaf(v[e], m, s1[e]); s2[e] = fmaf(v[e] * m, v[e], s2[e]); }
      }
#ifdef SEUNET_STAMP
#pragma unroll
      for (int e = 0; e < SR; ++e) asm volatile("" : "+v"(s1[e]), "+v"(s2[e]));
      SSTAMP(6);   // statistics
#endif
      // runs of 4 consecutive channels -> 8-byte pieces
      u32x2s u[ACCR / 4];
#pragma unroll
      for (int pc = 0; pc < ACCR / 4; ++pc) {
        float w4[4] = {v[4 * pc], v[4 * pc + 1], v[4 * pc + 2], v[4 * pc + 3]};
        if constexpr (DACC) {
          const int c0 = chan(4 * pc);
          const u32x2s o = *reinterpret_cast<const u32x2s*>(smem + Geo::OLD + (slot * ST_NW + wave) * Geo::OLDI * 1024 +
                                                             ((b * NB + fn) * a.dstC + c0) * (int)sizeof(T));
          w4[0] += unpack_lo<T>(o.x); w4[1] += unpack_hi<T>(o.x);
          w4[2] += unpack_lo<T>(o.y); w4[3] += unpack_hi<T>(o.y);
        }
        u[pc].x = pack2<T>(w4[0], w4[1]);
        u[pc].y = pack2<T>(w4[2], w4[3]);
      }
      if (!staged) {
#pragma unroll
        for (int pc = 0; pc < ACCR / 4; ++pc) {
          if constexpr (SEUNET_STREAM_PROBE == 2) asm volatile("" :: "v"(u[pc]));
          else __builtin_amdgcn_raw_buffer_store_b64(u[pc], rd, lofs[b][pc], soff, 0);
        }
      } else {
#pragma unroll
        for (int pc = 0; pc < ACCR / 4; ++pc)
          *reinterpret_cast<u32x2s*>(smem + stg_w[b][pc]) = u[pc];
      }
      SSTAMP(7);     // packing + store issue
    }
    if (staged) {
      // (LDS instructions of one wave execute in order: the reads below see the writes above, and the next step's writes come
      // after these reads.  The compiler is told so: the two views of the buffer have different vector types)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k = 0; k < NSTG; ++k) {
        const u32x2s qv = *reinterpret_cast<const u32x2s*>(smem + stg_r[k]);
        if constexpr (SEUNET_STREAM_PROBE == 2) asm volatile("" :: "v"(qv));
        else __builtin_amdgcn_raw_buffer_store_b64(qv, rd, stg_o[k], soff, 0);
      }
      asm volatile("" ::: "memory");
    }
  };

  // ---- the march ----
  // VMEM issue order of a wave: [prologue: DMA(0) .. DMA(PF-1)]  then per step s: DMA(s+PF) (inside compute), stores(s).
  // At the top of step s >= 1 plane s must have landed; it was issued in step s - PF (or the prologue), and the only LOADS
  // younger than it are DMA(s+1) .. DMA(s+PF-1): the wait allows (PF-1) * LW outstanding operations and counts NO store as
  // outstanding.  Round 3 (and this round at first) added the stores issued since -- correct if operations retire in issue
  // order, and they do not: with 16-byte stores about one launch in a hundred passed the wait before its plane had landed
  // (wrong output rows, never reproducibly), because a store can retire ahead of an older LDS-DMA load and lower the count
  // early.  The 8-byte stores never showed it in thousands of runs; the count no longer relies on it (same speed: the stores
  // of the previous step are acknowledged within the step).  Then one barrier: every wave's part of the plane is in LDS, and
  // every wave has finished reading the slot that this step's prefetch overwrites (ring = PF + 2 slots).
  constexpr int LW = ITEMS + (DACC ? Geo::OLDI : 0);   // DMA instructions per wave and step (padded to the same count in every wave)
#pragma unroll
  for (int k = 0; k < Geo::PF; ++k) { dma_plane(k, k); dma_old(k, k); }
  // The weight loads are ordinary (compiler-visible) loads, and the compiler waits for a load at its first use -- which is inside
  // the march.  Where it does not peel the first step (the accumulating variants) it put `s_waitcnt vmcnt(4) / (1) / (0)` between the
  // MFMAs of EVERY step: each step then drained the plane prefetch it had just issued (the counter is in issue order), i.e. the
  // whole HBM latency was exposed per step (found in the ISA, round 4).  One use of every fragment here makes it wait once, before
  // the march (it also drains the prologue's planes: a one-time cost).
#pragma unroll
  for (int k = 0; k < NTAP; ++k) asm volatile("" :: "v"(wreg[k]));
  SSTAMP(10);      // (prologue) first planes requested, weights arrived
  stream_wait_vm<(Geo::PF - 1) * LW>();    // plane 0 has landed (this wave's part)
  __builtin_amdgcn_s_barrier();
  SSTAMP(0);   // prologue: plans, weights, first planes
  int slot = 0, slot_pf = Geo::PF;         // s % Geo::RING, (s + Geo::PF) % Geo::RING
  for (int s0 = 0; s0 < nsteps; s0 += 3) {
    [&]<int... PH>(std::integer_sequence<int, PH...>) __attribute__((always_inline)) {
      ([&]() __attribute__((always_inline)) {
        const int s = s0 + PH;
        if (s < nsteps) {
          if (s > 0) {
            stream_wait_vm<(Geo::PF - 1) * LW>();
            SSTAMP(1);   // counted wait for the plane (and the stores of two steps ago)
            __builtin_amdgcn_s_barrier();
            SSTAMP(2);   // barrier
          }
          compute(s, slot, slot_pf, std::integral_constant<int, PH>{});
          SSTAMP(3);     // fragment reads, DMA issue, MFMA issue
#ifdef SEUNET_STAMP
          asm volatile("v_mov_b32 %0, %0" : "+v"(acc[(PH + 2) % 3][NBX - 1][ACCR - 1]));
          SSTAMP(4);     // the finished set's last MFMA has written its result
#endif
          finish(s, slot, std::integral_constant<int, PH>{});
          SSTAMP(5);     // epilogue: sums, packing, store issue
          slot = slot == Geo::RING - 1 ? 0 : slot + 1;
          slot_pf = slot_pf == Geo::RING - 1 ? 0 : slot_pf + 1;
        }
      }(), ...);
    }(std::make_integer_sequence<int, 3>{});
  }
  // ---- InstanceNorm partial sums of this workgroup: un-shift in f64, reduce over the lanes that hold the same channels,
  //      then over the four waves (fixed order), one record per workgroup ----
  if (FWD && a.stats != nullptr) {
    stream_wait_vm<0>();
    __syncthreads();             // the ring is dead: reuse its first bytes
    double* red = reinterpret_cast<double*>(smem);     // [8 waves][COUTP][2]
#pragma unroll
    for (int e = 0; e < ACCR; ++e) {
      double t1 = (double)s1[FWD ? e : 0];
      double t2 = (double)s2[FWD ? e : 0];
      constexpr int GROUP = COUTP == 32 ? 32 : 16;
#pragma unroll
      for (int off = 1; off < GROUP; off <<= 1) { t1 += shfl_xor_settled(t1, off); t2 += shfl_xor_settled(t2, off); }
      if (fn == 0) {
        const int c = chan(e);
        red[(wave * COUTP + c) * 2] = t1;
        red[(wave * COUTP + c) * 2 + 1] = t2;
      }
    }
    __syncthreads();
    if (tid < COUTP * 2) {
      const int c = tid >> 1, k = tid & 1;
      if (c < a.cout) {
        double tot = 0.0;
#pragma unroll
        for (int wv = 0; wv < ST_NW; ++wv) tot += red[(wv * COUTP + c) * 2 + k];      // fixed order
        const long long slot = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        const long long slots = (long long)gridDim.y * gridDim.x;
        a.stats[(((long long)n * slots + slot) * a.cout + c) * 2 + k] = tot;
      }
    }
  } else {
    stream_wait_vm<0>();         // no DMA may outlive the workgroup's LDS allocation
  }
#ifdef SEUNET_STAMP
  if (a.debug != nullptr && lane == 0) {
    const size_t w = (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * ST_NW + wave;
    for (int i = 0; i < 12; ++i) a.debug[w * 12 + i] = ph[i];
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// weight packing: PyTorch (Cout, Cin, 3, 3, 3) f32 -> [tap][lane][8] MFMA A-operand fragments
// ------------------------------------------------------------------------------------------------------------------
struct StreamPackArgs { const float* w; void* out; int cin_w, cout_w, tflip, cin_e, cout_e, coutp, xfold, np; };

static constexpr int ST_PACK_MAX = 8;          // layers per multi-layer launch (the network has four forward and three data-gradient ones)
struct StreamPackList { StreamPackArgs j[ST_PACK_MAX]; int n; };

// blockIdx.x = packed tap index k; one wave writes the 64 fragments of that tap
template <typename T>
__device__ __forceinline__ void conv_stream_pack_body(const StreamPackArgs& p, int k) {
  const int lane = threadIdx.x;
  const int vpk = p.xfold ? 4 / p.np : 1;       // x-tap slots per MFMA
  const int ndx = p.xfold ? (3 + vpk - 1) / vpk : 3;
  if (k >= 9 * ndx) return;                   // ((dz*3 + dy) * ndx + dxi)
  const int dxi = k % ndx, zy = k / ndx;
  const int row = p.coutp == 32 ? (lane & 31) : (lane & 15);     // output channel
  const int grp = p.coutp == 32 ? (lane >> 5) : (lane >> 4);     // k-group: 8 elements
  T* out = reinterpret_cast<T*>(p.out) + ((size_t)k * 64 + lane) * 8;
  for (int j = 0; j < 8; ++j) {
    int ci, dx;
    if (p.xfold) { ci = 8 * (grp % p.np) + j; dx = vpk * dxi + grp / p.np; }   // K = (x-tap slot, channel); slots >= 3: zero padding
    else { ci = 8 * grp + j; dx = dxi; }
    float v = 0.f;
    if (row < p.cout_e && ci < p.cin_e && dx < 3) {
      const int tap = zy * 3 + dx;
      // effective operator W_e[co][ci][tap]; the data gradient swaps the channel roles and mirrors the taps
      v = p.tflip ? p.w[((long long)ci * p.cin_w + row) * 27 + (26 - tap)] : p.w[((long long)row * p.cin_w + ci) * 27 + tap];
    }
    out[j] = from_f32<T>(v);
  }
}
template <typename T>
__global__ void __launch_bounds__(64)
conv_stream_pack_kernel(StreamPackArgs p) { conv_stream_pack_body<T>(p, blockIdx.x); }
// every streaming layer of a pass in one launch: blockIdx.y = layer
template <typename T>
__global__ void __launch_bounds__(64)
conv_stream_pack_multi_kernel(StreamPackList l) { conv_stream_pack_body<T>(l.j[blockIdx.y], blockIdx.x); }

// which variant serves (padded source channels, destination channels); 0 = none
static int stream_variant(int dtype, int taps, int dil, int src_c, int dst_c) {
  if ((dtype != SEUNET_BF16 && dtype != SEUNET_F16) || taps != 27 || (dil != 1 && dil != 2)) return 0;
  if (src_c == 8 && dst_c <= 16 && dil == 1) return 3;      // x-folded, 4 slots
  if (src_c == 16 && dst_c <= 16) return 4;                 // x-folded, 2 slots
  if (src_c == 16 && dst_c <= 32) return 1;
  if (src_c == 32 && dst_c <= 16) return 2;
  return 0;
}
bool conv_stream_supported(int dtype, int taps, int dil, int src_c, int dst_c) { return stream_variant(dtype, taps, dil, src_c, dst_c) != 0; }

size_t conv_stream_wpack_bytes(int src_c) { return (size_t)27 * 64 * 16; (void)src_c; }

static int stream_zsteps(Dims d, int dil) {
  // Output planes per workgroup.  Every workgroup pays a prologue (plan, weights, first planes: 8 % of a wave's time at 34 steps,
  // round-4 stamps) and two halo steps, and the kernels with more than 128 registers per lane (every variant but the 8-channel
  // one) have ONE workgroup resident per CU, so nothing hides them: the march is as long as the volume allows while a batch of
  // FOUR samples still gives every CU a workgroup (measured at 1024, 512 and 256 workgroups per launch on 4 x 128^3: ec3 forward
  // 0.308 / 0.282 / 0.274 ms, dc6 forward 0.219 / 0.204 / 0.190, ec2 forward 0.091 / 0.090 / 0.082 -- also for the 8-channel
  // variants).  Round 3 cut the axis into 32-plane marches (1024 workgroups of 34 steps on that batch); this gives 256 of 130
  // at dilation 1, 512 of 66 at dilation 2.  At least 8 planes per march.  The split is a function of the SAMPLE's extents only:
  // the statistics records, hence the bits of a sample's result, must not depend on the batch it sits in (the data-parallel
  // equivalence tests and the window loop rely on that), so a batch of one 128^3 sample fills a quarter of the chip here.
  static const int target = [] { const char* e = std::getenv("SEUNET_STREAM_WGS"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 256; }();
  const int planes = cdiv(d.D, dil);
  const long long base = (long long)cdiv(d.H, ST_TY) * cdiv(d.W, ST_TX) * dil * 4;
  long long segs = (target + base - 1) / base;
  const int cap = cdiv(planes, 8);
  if (segs > cap) segs = cap;
  if (segs < 1) segs = 1;
  return cdiv(planes, (int)segs);
}
int conv_stream_slots(Dims d, int dil) {
  const int planes = cdiv(d.D, dil);
  const int zs = stream_zsteps(d, dil);
  return cdiv(d.H, ST_TY) * cdiv(d.W, ST_TX) * cdiv(planes, zs) * dil;
}

static int stream_pack_args(int dtype, const float* w, int cin_w, int cout_w, int tflip, int src_c, int dst_c, void* wpack, StreamPackArgs& p) {
  const int cin_e = tflip ? cout_w : cin_w, cout_e = tflip ? cin_w : cout_w;
  const int var = stream_variant(dtype, 27, 1, src_c, dst_c);
  SEUNET_CHECK(var != 0 && w && wpack, "conv_stream_pack: unsupported shape (%d -> %d channels)", src_c, dst_c);
  SEUNET_CHECK(cin_e <= src_c && cout_e <= dst_c, "conv_stream_pack: weight (%d -> %d) exceeds the tensors (%d -> %d)", cin_e, cout_e, src_c, dst_c);
  p = StreamPackArgs{w, wpack, cin_w, cout_w, tflip, cin_e, cout_e, var == 1 ? 32 : 16, (var == 3 || var == 4) ? 1 : 0, src_c / 8};
  return 0;
}
int launch_conv_stream_pack(int dtype, const float* w, int cin_w, int cout_w, int tflip, int src_c, int dst_c, void* wpack, hipStream_t s) {
  StreamPackArgs p{};
  if (int e = stream_pack_args(dtype, w, cin_w, cout_w, tflip, src_c, dst_c, wpack, p)) return e;
  if (dtype == SEUNET_F16) conv_stream_pack_kernel<f16_t><<<27, 64, 0, s>>>(p);
  else conv_stream_pack_kernel<bf16_t><<<27, 64, 0, s>>>(p);
  SEUNET_LAUNCH_CHECK();
  return 0;
}
// all streaming layers of a pass in one launch (the weights change every step; seven 6-us launches per step otherwise)
int launch_conv_stream_pack_multi(int dtype, const StreamPackJob* jobs, int n, hipStream_t s) {
  for (int i0 = 0; i0 < n; i0 += ST_PACK_MAX) {
    StreamPackList l{};
    l.n = n - i0 < ST_PACK_MAX ? n - i0 : ST_PACK_MAX;
    for (int i = 0; i < l.n; ++i) {
      const StreamPackJob& j = jobs[i0 + i];
      if (int e = stream_pack_args(dtype, j.w, j.cin_w, j.cout_w, j.tflip, j.src_c, j.dst_c, j.wpack, l.j[i])) return e;
    }
    if (dtype == SEUNET_F16) conv_stream_pack_multi_kernel<f16_t><<<dim3(27, l.n), 64, 0, s>>>(l);
    else conv_stream_pack_multi_kernel<bf16_t><<<dim3(27, l.n), 64, 0, s>>>(l);
    SEUNET_LAUNCH_CHECK();
  }
  return 0;
}

template <typename T, int CIN, int COUTP, bool XFOLD, int DIL, bool FWD, bool DACC>
static int stream_launch_acc(const StreamArgs& a, dim3 grid, hipStream_t s) {
  using Geo = StreamGeo<CIN, COUTP, XFOLD, DIL, DACC>;
  static unsigned long long configured = 0;
  if (int e = configure_kernel_lds(configured, reinterpret_cast<const void*>(&conv_stream_kernel<T, CIN, COUTP, XFOLD, DIL, FWD, DACC>), Geo::LDS)) return e;
  conv_stream_kernel<T, CIN, COUTP, XFOLD, DIL, FWD, DACC><<<grid, ST_NW * 64, Geo::LDS, s>>>(a);
  SEUNET_LAUNCH_CHECK();
  return 0;
}
template <typename T, int CIN, int COUTP, bool XFOLD, int DIL>
static int stream_launch_one(const StreamArgs& a, dim3 grid, hipStream_t s) {
  if (a.bias != nullptr || a.stats != nullptr) {
    SEUNET_CHECK(!a.dacc, "conv_stream: accumulation into the destination is a data-gradient feature (no bias, no statistics)");
    return stream_launch_acc<T, CIN, COUTP, XFOLD, DIL, true, false>(a, grid, s);
  }
  return a.dacc ? stream_launch_acc<T, CIN, COUTP, XFOLD, DIL, false, true>(a, grid, s)
                : stream_launch_acc<T, CIN, COUTP, XFOLD, DIL, false, false>(a, grid, s);
}

template <typename T>
static int stream_dispatch(int var, int dil, const StreamArgs& a, dim3 grid, hipStream_t s) {
  if (var == 3) return stream_launch_one<T, 8, 16, true, 1>(a, grid, s);
  if (var == 4) return dil == 1 ? stream_launch_one<T, 16, 16, true, 1>(a, grid, s) : stream_launch_one<T, 16, 16, true, 2>(a, grid, s);
  if (var == 1) return dil == 1 ? stream_launch_one<T, 16, 32, false, 1>(a, grid, s) : stream_launch_one<T, 16, 32, false, 2>(a, grid, s);
  return dil == 1 ? stream_launch_one<T, 32, 16, false, 1>(a, grid, s) : stream_launch_one<T, 32, 16, false, 2>(a, grid, s);
}

// src: [N][D][H][W][src_c] (src_c = 8 | 16 | 32); dst: [N][D][H][W][dst_c], cout valid output channels written
// (dst_c % 8 == 0, every channel < dst_c is written: channels >= the weight's Cout are zero + bias 0)
int launch_conv_stream(int dtype, int dil, const void* src, int src_c, const void* wpack, const float* bias, void* dst, int dst_c,
                       int dst_accumulate, double* stats, Dims d, hipStream_t s) {
  const int var = stream_variant(dtype, 27, dil, src_c, dst_c);
  SEUNET_CHECK(var != 0, "conv_stream: unsupported shape (%d -> %d channels, dilation %d, dtype %d)", src_c, dst_c, dil, dtype);
  SEUNET_CHECK(src && wpack && dst && dst_c % 8 == 0, "conv_stream: bad argument");
  SEUNET_CHECK((long long)d.vox() * dst_c * 2 < (1LL << 31) && (long long)d.H * d.W * src_c * 2 < (1LL << 31) &&
               (long long)d.vox() * src_c * 2 < (1LL << 32),
               "conv_stream: one sample exceeds the 32-bit offsets of this kernel");
  StreamArgs a{};
  a.src = src; a.wpack = wpack; a.bias = bias; a.dst = dst; a.dstC = dst_c; a.dacc = dst_accumulate; a.cout = dst_c;
  a.stats = stats; a.zero = device_zero_page();
  a.debug = g_conv_debug;
  static const bool no_stage = std::getenv("SEUNET_STREAM_NO_STAGE") != nullptr;     // (diagnostic switch for A/B timing)
  a.stage = no_stage ? 0 : 1;
  SEUNET_CHECK(a.zero != nullptr, "conv_stream: no zero page on this device");
  a.N = d.N; a.D = d.D; a.H = d.H; a.W = d.W;
  const int planes = cdiv(d.D, dil);
  a.zsteps = stream_zsteps(d, dil);
  a.nzseg = cdiv(planes, a.zsteps);
  a.nyb = cdiv(d.H, ST_TY); a.nxb = cdiv(d.W, ST_TX);
  SEUNET_CHECK(d.N <= 65535 && a.nzseg * dil <= 65535, "conv_stream: grid too large");
  dim3 grid(a.nyb * a.nxb, a.nzseg * dil, d.N);
  return dtype == SEUNET_F16 ? stream_dispatch<f16_t>(var, dil, a, grid, s) : stream_dispatch<bf16_t>(var, dil, a, grid, s);
}

}  // namespace seunet
