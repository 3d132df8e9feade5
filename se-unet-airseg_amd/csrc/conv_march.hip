// Marching 3x3x3 convolution for the 32- and 64-input-channel layers of the two fine levels (reference nn.Conv3d at
// SE_UNet.py:57 / :15 for ec4..ec6, dc4, dc5 forward; the data gradients of ec4..ec6, dc3, dc4, dc5), dilation 1 or 2.
//
// Why a third conv kernel: on these layers the tiled implicit-GEMM kernel (conv_igemm.hip) is bound by the LDS, not by the
// matrix pipes -- every MFMA needs 1.25 KB of fragment reads plus its share of a 67 KB register -> LDS staging round per
// 16-channel chunk, and the per-tile prologue / epilogue (first fetch, statistics, stores) is covered only by the second
// workgroup of the CU: 57 % busy matrix pipes (profiles/r02_pmc_dc5.md).  Here the structure of conv_stream.hip is taken to
// wide layers:
//   * a workgroup (4 waves, ONE per SIMD, the whole 512-entry register file each) owns an RY x 32 (y, x) patch and marches
//     along z, input-stationary: step s stages ONE input plane and adds its contribution to the output planes s+1, s, s-1
//     (three accumulator sets in registers, roles rotated by a 3x unrolled loop);
//   * the WEIGHTS live in registers for the whole march: a wave owns 16 output channels (v_mfma_f32_16x16x32: A = 16
//     output channels x 32 input channels of one tap, B = 32 channels x 16 voxels), i.e. 27 x CIN/32 fragments = 108 / 216
//     registers; the waves of a workgroup split the output channels (N groups) and, when there are fewer than four N
//     groups, the rows of the patch.  No weight ever touches the LDS, no K-chunk loop, no weight staging;
//   * every fragment read from the LDS (16 voxels x 32 channels) feeds 9 MFMAs (3 dz x up to 3 dy): 0.11 KB of LDS reads
//     per MFMA instead of 1.25 KB;
//   * input planes arrive by LDS-DMA (global_load_lds_dwordx4, counted vmcnt, one raw s_barrier per step) two steps ahead
//     into a 3-slot ring.  LDS image of a plane: voxel-major [row][x][CIN x 2 B] (row pitch 36 voxels), so a DMA
//     instruction reads whole 64 / 128-byte voxel records (full cache lines at 64 channels); the 16-byte pieces of a voxel
//     are XOR-swizzled by x (on the SOURCE side of the DMA and on the read) so that the fragment reads stay (nearly)
//     conflict-free, and every fragment address is "lane base + immediate";
//   * the finished plane is written lazily: a row of the oldest accumulator set is converted and stored right before the
//     step that re-initialises it (the first MFMA of a row takes the bias vector / zero as its C operand, so nothing is
//     ever zeroed), i.e. the epilogue of plane j runs under the MFMAs of plane j+3; gradient accumulation (+=) reads the
//     old destination rows through a second LDS-DMA ring one step ahead;
//   * InstanceNorm partial sums per lane in f32 inside a plane, in f64 (LDS slots) across the march and across lanes, one
//     record per workgroup.
// One workgroup per CU: a 4 x 128^3 batch at 64 -> 32 channels is exactly 256 marches of 128 planes.
#include "seunet_common.h"
#include <utility>
#include <type_traits>

namespace seunet {

typedef bf16_t mbf16x8 __attribute__((ext_vector_type(8)));
typedef f16_t mf16x8 __attribute__((ext_vector_type(8)));
typedef float mf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int mu32x2 __attribute__((ext_vector_type(2)));

// The matrix instruction as inline asm with the accumulator pinned to the accumulator half of the register file ("+a": D = C,
// in place) and the weights / voxels in the other half.  Through the builtin the register allocator moved accumulator sets
// between the two halves (D != C, copies through v_accvgpr_*) until a 377-register kernel no longer fitted 512.  Hazards: the
// asm's operands are ordinary data dependencies (the compiler still waits for the LDS reads that produce them); an accumulator
// is read by vector instructions (the lazy epilogue) a whole step after its last MFMA, and re-initialised after that read.
// WA = 1: the weight fragment lives in the accumulator half too (K-step 1 of the 64-channel layers: 216 weight registers).
template <typename T, int WA> __device__ __forceinline__ void mm16_acc(mf32x4& c, mbf16x8 a, mbf16x8 b) {
  if constexpr (std::is_same<T, f16_t>::value) {
    if constexpr (WA) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "a"(a), "v"(b));
    else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  } else {
    if constexpr (WA) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "a"(a), "v"(b));
    else asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  }
}
// first MFMA of an accumulator: C = the bias vector (accumulator half, like D) or the constant 0
template <typename T, int WA, bool BIAS> __device__ __forceinline__ void mm16_init(mf32x4& c, mbf16x8 a, mbf16x8 b, const mf32x4& c0) {
  if constexpr (std::is_same<T, f16_t>::value) {
    if constexpr (BIAS) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&a"(c) : "v"(a), "v"(b), "a"(c0));
    else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
  } else {
    if constexpr (BIAS) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&a"(c) : "v"(a), "v"(b), "a"(c0));
    else asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
  }
}

template <typename T> __device__ __forceinline__ mf32x4 mm16b(mbf16x8 a, mbf16x8 b, mf32x4 c) {
  if constexpr (std::is_same<T, f16_t>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mf16x8, a), __builtin_bit_cast(mf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

struct MarchArgs {
  const void* src0; const void* src1;   // one or two source tensors of srcC channels each (virtual concatenation)
  int srcC, nsrc;
  const void* wpack; const float* bias;
  void* dst0; void* dst1; void* dst2;    // channel split of the output (each a multiple of 16 channels; null = dropped)
  int dstC0, dstC1, dstC2;
  int dcum1, dcum2;                      // first output channel of destination 1 / 2
  int dacc0, dacc1, dacc2;
  int cout;                              // total output channels (multiple of 16)
  double* stats; const void* zero;
  int N, D, H, W;
  int nyb, nxb, nseg, zsteps, nblk;      // patches, z segments (per parity class), output planes per segment, N blocks
  int buf;                               // the source(s) of a sample fit one 32-bit buffer descriptor (see march_dma16_buf)
};

static constexpr int MA_TX = 32, MA_NW = 4, MA_HXP = 36, MA_PF = 2, MA_RING = 3;
#ifndef SEUNET_MARCH_PFD
#define SEUNET_MARCH_PFD 1
#endif
static constexpr int MA_PFD = SEUNET_MARCH_PFD;   // fragment prefetch distance in units

template <int KS, int NGW, int RYW, int DIL, int MODE> struct MarchGeo {
  static constexpr int RGW = MA_NW / NGW, RY = RYW * RGW;
  static constexpr int NP = 4 * KS, VB = NP * 16;                    // 16-B pieces / bytes per voxel
  static constexpr int HX = MA_TX + 2 * DIL, HY = RY + 2 * DIL, HYW = RYW + 2 * DIL;
  static constexpr int ROWB = MA_HXP * VB;                           // LDS row pitch in bytes
  static constexpr int NI = (HY * MA_HXP * NP + 63) / 64;            // DMA wave-instructions per plane
  static constexpr int PLB = NI * 1024;
  static constexpr int ITEMS = (NI + MA_NW - 1) / MA_NW;             // per wave (padded: every wave issues the same count)
  static constexpr int OLDN = MODE == 2 ? RYW : 0;                   // old-row DMA instructions per wave and step
  static constexpr int STORES = RYW * 2;                             // store instructions per wave and step
  static constexpr int TOT = ITEMS + OLDN + STORES;                  // vector-memory operations per wave and step
  static constexpr int DUMP = MA_RING * PLB;                         // 1 KB landing area of the padding DMA instructions
  static constexpr int OLD = DUMP + 1024;                            // [2 slots][4 waves][RYW rows][32 voxels][32 B]
  static constexpr int STAT = OLD + (MODE == 2 ? 2 * MA_NW * RYW * 1024 : 0);   // forward: f64 march totals, [8 values][256 lanes]
  static constexpr int LDS = STAT + (MODE == 0 ? 8 * 256 * 8 : 0);
  static_assert(HX <= MA_HXP, "row pitch");
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert((HYW - 1) * ROWB + 2 * 16 * VB < 65536, "fragment immediates must fit the 16-bit offset field");
};

__device__ __forceinline__ void march_dma16(const void* gsrc, unsigned lds_dst) {
  // one LDS-DMA wave-instruction: 64 lanes x 16 B, LDS destination = lds_dst + 16 * lane (M0 carries the base)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// The same through a buffer descriptor (round 4, single-source launches): base + scalar offset + per-lane offset, and a lane whose
// offset lies beyond num_records writes ZEROS into the LDS (probed on gfx950: scripts/probes/blds_oob.hip) -- the padding of a
// plane image needs no zero page, no per-lane pointer select, no validity mask and no 64-bit address.  In the main loop of the
// dc5 data gradient that is 1.25 -> 1.06 other instructions per MFMA and 65 -> 12 scalar-register spill reads.
typedef unsigned int mu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void march_dma16_buf(unsigned voff, mu32x4 rsrc, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void march_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int NP> __device__ __forceinline__ int march_swz(int hx) {
  // piece permutation of the voxel at halo column hx (an involution applied on the DMA source side and on the read)
  if constexpr (NP == 4) return ((hx >> 3) & 1) << 1;
  else return (hx >> 1) & 7;
}

template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
    (f(std::integral_constant<int, I>{}), ...);
  }(std::make_integer_sequence<int, N>{});
}

// MODE 0: forward (bias + InstanceNorm partial sums); 1: data gradient; 2: data gradient with accumulation (+=)
// BUF: one source tensor whose sample is < 4 GB: the plane DMA goes through a buffer descriptor (march_dma16_buf)
template <typename T, int KS, int NGW, int RYW, int DIL, int MODE, bool BUF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
conv_march_kernel(MarchArgs a) {
  using Geo = MarchGeo<KS, NGW, RYW, DIL, MODE>;
  constexpr int RGW = Geo::RGW, RY = Geo::RY, NP = Geo::NP, VB = Geo::VB, HX = Geo::HX, HY = Geo::HY, HYW = Geo::HYW;
  static_assert(RYW >= 2 && HYW - 1 > 2 * DIL, "epilogue schedule");
  constexpr int ROWB = Geo::ROWB, NI = Geo::NI, PLB = Geo::PLB, ITEMS = Geo::ITEMS;
  constexpr int NTAP = 27;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ng = wave % NGW, rg = wave / NGW;           // N group (16 output channels) / row group of this wave
  const int n16 = lane & 15, g = lane >> 4;              // voxel of the 16-block / k-group (inputs), channel quad (outputs)
  // patch (XCD-contiguous order), z segment, parity class, N block
  int t;
  {
    const int nt = gridDim.x, b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int xb = t % a.nxb, yb = t / a.nxb;
  const int nb = blockIdx.y % a.nblk;
  const int segpz = blockIdx.y / a.nblk;
  const int pz = segpz % DIL, seg = segpz / DIL;
  const int n = blockIdx.z;
  const int x0 = xb * MA_TX, y0 = yb * RY;
  const int nplanes = (a.D - pz + DIL - 1) / DIL;        // planes of this parity class
  const int q0 = seg * a.zsteps;
  const int Z = min(a.zsteps, nplanes - q0);             // output planes of this march (>= 1 by construction of the grid)
  const int ncompute = Z + 2;                            // input planes q0-1 .. q0+Z

  // ---- weights: registers for the whole march.  wpack: [16-channel group][tap][K-step][lane][8 elements] ----
  const int g16 = nb * NGW + ng;
  mbf16x8 wreg[NTAP * KS];
  {
    const uint4* wp = reinterpret_cast<const uint4*>(a.wpack) + (size_t)g16 * (NTAP * KS * 64) + lane;
#pragma unroll
    for (int k = 0; k < NTAP * KS; ++k) wreg[k] = __builtin_bit_cast(mbf16x8, wp[k * 64]);
    // 64-channel inputs: 216 weight registers + fragments + addresses do not fit the 256 architectural VGPRs, and left alone the
    // allocator parks weights in the accumulator half and copies them back before every use (170 v_accvgpr_read per step of the
    // dc5 forward: the forward's 2.68 of 4 busy SIMDs against the data gradient's 2.98).  The matrix instruction takes its A
    // operand from either half, so the fragments of K-step 1 are PINNED to the accumulator half here (an empty asm with an "a"
    // constraint; the MFMAs stay builtins, so the scheduler still interleaves the epilogue): 511 -> 140 copies per three steps,
    // the rest being the epilogue's reads of finished accumulators (round 4; counted on the ISA of every KS = 2 configuration)
    if constexpr (KS == 2) {
#pragma unroll
      for (int k = 0; k < NTAP * KS; ++k)
        if ((k % KS) == 1) asm volatile("" : "+a"(wreg[k]));
    }
  }

  const long long plane_bytes = (long long)a.H * a.W * a.srcC * (long long)sizeof(T);
  const unsigned char* zero_page = reinterpret_cast<const unsigned char*>(a.zero) + lane * 16;
  const unsigned char* src0_n = reinterpret_cast<const unsigned char*>(a.src0) + (long long)n * a.D * plane_bytes;
  const unsigned char* src1_n = reinterpret_cast<const unsigned char*>(a.nsrc > 1 ? a.src1 : a.src0) + (long long)n * a.D * plane_bytes;
  // plane of step s -> ring slot; every wave issues exactly ITEMS instructions (the vmcnt arithmetic counts on it).  The
  // plane-level part (validity, 64-bit plane bases, LDS slot base) is computed once per step (`plane_of`) and handed to the
  // items: they sit in different scheduling regions, so the compiler recomputed it for each of them, and with one wave per
  // SIMD every scalar instruction takes an issue slot from the MFMA stream
  // BUF: ONE descriptor for the sample -- base = the lower of the (one or two) source pointers, a lane of the other source adds the
  // distance between the two tensors (the launcher checked that distance + sample fit 32 bits; net.cpp keeps dc5's two sources
  // next to each other in the arena), num_records = everything up to the end of the upper tensor's sample
  unsigned s_lo = 0, s_hi = 0, sample_bytes = 0, off_s0 = 0, off_s1 = 0;
  if constexpr (BUF) {
    const unsigned char* base = src1_n < src0_n ? src1_n : src0_n;
    off_s0 = (unsigned)(src0_n - base);
    off_s1 = (unsigned)(src1_n - base);
    const unsigned long long u = reinterpret_cast<unsigned long long>(base);
    s_lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    s_hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) & 0xFFFFu;
    sample_bytes = (off_s0 > off_s1 ? off_s0 : off_s1) + (unsigned)((long long)a.D * plane_bytes);
  }
  // ---- DMA plan: item it of this wave = instruction number wave + 4 * it of the plane image ----
  unsigned doff[ITEMS];            // byte offset inside a z-plane of the source; bit 31 = second source (BUF: + that source's distance from the base); ~0 = padding
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int id = wave + MA_NW * it;
    const int L = id * 64 + lane;
    const int v = L / NP, sl = L % NP;
    const int hy = v / MA_HXP, hx = v % MA_HXP;
    const int p = sl ^ march_swz<NP>(hx);
    const int ch = p * 8;
    const int y = y0 - DIL + hy, x = x0 - DIL + hx;
    const bool s1 = ch >= a.srcC;
    const int cl = s1 ? ch - a.srcC : ch;
    const bool ok = id < NI && hx < HX && hy < HY && y >= 0 && y < a.H && x >= 0 && x < a.W && (!s1 || a.nsrc > 1);
    if constexpr (BUF) doff[it] = ok ? (unsigned)(((y * a.W + x) * a.srcC + cl) * (int)sizeof(T)) + (s1 ? off_s1 : off_s0) : 0xFFFFFFFFu;
    else doff[it] = ok ? ((unsigned)(((y * a.W + x) * a.srcC + cl) * (int)sizeof(T)) | (s1 ? 0x80000000u : 0u)) : 0xFFFFFFFFu;
  }

  struct PlaneRef { const unsigned char* b0; const unsigned char* b1; unsigned lds; bool ok; unsigned soff; unsigned nrec; };
  auto plane_of = [&](int s, int slot) __attribute__((always_inline)) -> PlaneRef {
    PlaneRef r;
    const int pl = q0 - 1 + s;
    const int z = pz + DIL * pl;
    r.ok = pl >= 0 && z < a.D && s < ncompute;                     // wave-uniform
    const long long zb = (long long)(r.ok ? z : 0) * plane_bytes;  // (scalar)
    r.b0 = src0_n + zb; r.b1 = src1_n + zb;
    r.lds = lds_base + (unsigned)(slot * PLB);
    r.soff = (unsigned)zb;                                          // (BUF: < 2^32, checked by the launcher)
    r.nrec = r.ok ? sample_bytes : 0u;
    return r;
  };
  auto dma_item = [&](const PlaneRef& r, auto it_c) __attribute__((always_inline)) {
    constexpr int it = decltype(it_c)::value;
    if constexpr (it < ITEMS) {
      const bool real = wave + MA_NW * it < NI;                     // wave-uniform
      if constexpr (BUF) {      // (doff: the lane's offset inside a plane, 0xFFFFFFFF for padding; a padding item reads zero records)
        mu32x4 rs;
        rs.x = s_lo; rs.y = s_hi; rs.z = real ? r.nrec : 0u; rs.w = 0x00020000u;
        march_dma16_buf(doff[it], rs, r.soff, real ? r.lds + (unsigned)((wave + MA_NW * it) * 1024) : lds_base + (unsigned)Geo::DUMP);
        return;
      }
      const unsigned d = doff[it];
      const unsigned char* gp = ((d & 0x80000000u) ? r.b1 : r.b0) + (d & 0x7FFFFFFFu);
      gp = (r.ok && real && d != 0xFFFFFFFFu) ? gp : zero_page;
      march_dma16(gp, real ? r.lds + (unsigned)((wave + MA_NW * it) * 1024) : lds_base + (unsigned)Geo::DUMP);
    }
  };

  // ---- destination of this wave's 16 channels ----
  const int co0 = g16 * 16;
  void* dptr = a.dst0; int dC = a.dstC0, dch = co0, dacc = a.dacc0;
  if (co0 >= a.dcum2) { dptr = a.dst2; dC = a.dstC2; dch = co0 - a.dcum2; dacc = a.dacc2; }
  else if (co0 >= a.dcum1) { dptr = a.dst1; dC = a.dstC1; dch = co0 - a.dcum1; dacc = a.dacc1; }
  dC = __builtin_amdgcn_readfirstlane(dC); dch = __builtin_amdgcn_readfirstlane(dch); dacc = __builtin_amdgcn_readfirstlane(dacc);
  const long long dst_sample = (long long)a.D * a.H * a.W * dC * (long long)sizeof(T);
  unsigned char* dbase = dptr ? reinterpret_cast<unsigned char*>(dptr) + (long long)n * dst_sample : nullptr;
  {
    const unsigned long long u = reinterpret_cast<unsigned long long>(dbase);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    dbase = reinterpret_cast<unsigned char*>(((unsigned long long)hi << 32) | lo);
  }
  const int dst_records = __builtin_amdgcn_readfirstlane(dbase ? (int)dst_sample : 0);

  // gradient accumulation: the 32 voxels x 32 B of a destination row of this wave arrive by one DMA instruction per row,
  // issued one step ahead of the step that finishes the row
  auto dma_old = [&](int s_fin, int oslot) __attribute__((always_inline)) {     // the rows that step s_fin will finish
    if constexpr (MODE == 2) {
      const int xx = x0 + (lane >> 1);
#pragma unroll
      for (int r = 0; r < RYW; ++r) {
        const int j = r == 0 ? s_fin - 2 : s_fin - 3;               // (row 0 of a plane is finished one step ahead of its other rows)
        const int z = pz + DIL * (q0 + j);
        const bool jok = j >= 0 && j < Z && dacc != 0 && dbase != nullptr;        // wave-uniform
        const int y = y0 + rg * RYW + r;
        const bool ok = jok && y < a.H && xx < a.W;
        const unsigned char* gp = ok ? dbase + (((long long)z * a.H + y) * a.W + xx) * dC * (long long)sizeof(T) + (dch + 8 * (lane & 1)) * (long long)sizeof(T)
                                     : zero_page;
        march_dma16(gp, lds_base + (unsigned)(Geo::OLD + ((oslot * MA_NW + wave) * RYW + r) * 1024));
      }
    }
  };

  // ---- fragment addressing: byte offset of this lane's 16 bytes for (K-step, x-tap, 16-voxel block); row + immediate ----
  unsigned foff[KS][3][2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int hx = 16 * b + DIL * dx + n16;
        foff[ks][dx][b] = (unsigned)((rg * RYW * MA_HXP + hx) * VB + (((4 * ks + g) ^ march_swz<NP>(hx)) * 16));
      }

  mf32x4 acc[3][RYW][2];
  // (no initialisation: the first MFMA of every (set, row, block) takes `cinit` as its C operand)
  mf32x4 cinit;
#pragma unroll
  for (int e = 0; e < 4; ++e) cinit[e] = (MODE == 0 && a.bias != nullptr) ? a.bias[co0 + 4 * g + e] : 0.f;

  // InstanceNorm sums of this lane's 4 channels: f32 inside a plane, f64 across the march
  // f32 inside a plane (16 values per lane), f64 across the march: the march totals live in the LDS (one slot per lane and value;
  // the 216-weight-register forward variant has no 16 registers for them), so that sums of f32 values are added in f64 all the
  // way -- exact in practice, hence independent of how the planes are cut into marches (the batch size decides that)
  float s1[4], s2[4];
  double* stot = reinterpret_cast<double*>(smem + Geo::STAT) + tid;     // value k at stot[k * 256]
#pragma unroll
  for (int e = 0; e < 4; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if constexpr (MODE == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) stot[k * 256] = 0.0;
  }

  const int yrow0 = y0 + rg * RYW;
  const bool okx0 = x0 + n16 < a.W, okx1 = x0 + 16 + n16 < a.W;
  // store offsets: the lane's part (x, channel quad; beyond every sample when x is outside the volume) in a register, the
  // row's part (z, y: wave-uniform) in the scalar offset of the store; a row outside the volume or a plane outside the march
  // stores through a descriptor of zero records (dropped)
  const unsigned lx0 = okx0 ? (unsigned)(((x0 + n16) * dC + dch + 4 * g) * (int)sizeof(T)) : 0x80000000u;
  const unsigned lx1 = okx1 ? (unsigned)(((x0 + 16 + n16) * dC + dch + 4 * g) * (int)sizeof(T)) : 0x80000000u;
  const int row_pitch = __builtin_amdgcn_readfirstlane(a.W * dC * (int)sizeof(T));

  // ---- lazy epilogue of one (row, 16-voxel block) of output plane j held in accumulator set AI; s = the step that runs it ----
  auto finish_blk = [&](int s, int j, auto ai_c, auto r_c, auto b_c) __attribute__((always_inline)) {
    constexpr int AI = decltype(ai_c)::value, r = decltype(r_c)::value, b = decltype(b_c)::value;
    const bool jok = j >= 0 && j < Z;                                   // wave-uniform
    const int z = pz + DIL * (q0 + j);
    const int y = yrow0 + r;
    const bool rowok = jok && y < a.H;                                  // wave-uniform
    const __amdgpu_buffer_rsrc_t rdr = __builtin_amdgcn_make_buffer_rsrc(dbase, 0, rowok ? dst_records : 0, 0x00020000);
    const int soff = __builtin_amdgcn_readfirstlane(rowok ? (z * a.H + y) * row_pitch : 0);
    mf32x4 v = acc[AI][r][b];
    // the accumulator leaves the accumulator half HERE (one copy, at the epilogue's place in the unit): with several vector
    // uses the compiler otherwise copies every accumulator out right after its last MFMA and keeps a whole set in VGPRs
    asm volatile("" : "+v"(v));
    if constexpr (MODE == 2) {
      const mu32x2 o = *reinterpret_cast<const mu32x2*>(smem + Geo::OLD + (((s & 1) * MA_NW + wave) * RYW + r) * 1024 +
                                                        (16 * b + n16) * 32 + g * 8);
      v[0] += unpack_lo<T>(o.x); v[1] += unpack_hi<T>(o.x);
      v[2] += unpack_lo<T>(o.y); v[3] += unpack_hi<T>(o.y);
    }
    if constexpr (MODE == 0) {
      const bool okl = rowok && (b == 0 ? okx0 : okx1);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float vm = okl ? v[e] : 0.f; s1[e] += vm; s2[e] = fmaf(vm, vm, s2[e]); }   // (select, not multiply: the sets of the first steps hold garbage)
      // (pins the sums to this unit: left alone the compiler sinks every step's statistics to the end of the 3-step loop body and
      // keeps all the epilogue values in registers until then)
      asm volatile("" : "+v"(s1[0]), "+v"(s1[1]), "+v"(s1[2]), "+v"(s1[3]), "+v"(s2[0]), "+v"(s2[1]), "+v"(s2[2]), "+v"(s2[3]));
    }
    mu32x2 u;
    u.x = pack2<T>(v[0], v[1]);
    u.y = pack2<T>(v[2], v[3]);
    __builtin_amdgcn_raw_buffer_store_b64(u, rdr, b == 0 ? lx0 : lx1, soff, 0);
  };
  auto flush_stats = [&]() __attribute__((always_inline)) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        stot[e * 256] += (double)s1[e];
        stot[(4 + e) * 256] += (double)s2[e];
        s1[e] = 0.f; s2[e] = 0.f;
      }
    }
  };

  // ---- one step: input plane of step s (ring slot `slot`) -> the three accumulator sets; PH = s % 3 ----
  // Unit u = (input row hi, K-step ks, x-tap dx): two fragments (the 16-voxel blocks), up to 18 MFMAs.  The fragments of unit
  // u + 1 are requested before the MFMAs of unit u issue (two register sets of 2 fragments); the DMA instructions of the plane
  // two steps ahead and the lazy epilogue are spread over the units.  Epilogue schedule of step s: row r >= 1 of output s - 3
  // (accumulator set PH, about to be re-initialised by this step's tap dz = 0) in the units of input row r - 1, i.e. at least
  // one unit before the first MFMA that overwrites it; row 0 of output s - 2 (set PH + 1, complete since input row 2 DIL of this
  // step) in the last units of the step, before the next step re-initialises it.
  auto compute = [&](int s, int slot, int slot_pf, auto ph_c) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_c)::value;
    constexpr int NU = HYW * KS * 3;
    const unsigned char* pl = smem + slot * PLB;
    const PlaneRef pref = plane_of(s + MA_PF, slot_pf);
    // fragments are requested PFD units ahead of their MFMAs (2 was measured: no change on any layer, the waves do not wait
    // for fragments -- SQ_WAIT_INST_LDS is 1 % of the wave cycles)
    constexpr int PFD = MA_PFD;
    mbf16x8 fr[PFD + 1][2];
    auto load_unit = [&](auto u_c) __attribute__((always_inline)) {
      constexpr int u = decltype(u_c)::value;
      if constexpr (u < NU) {
        constexpr int hi = u / (3 * KS), ks = (u / 3) % KS, dx = u % 3;
#pragma unroll
        for (int b = 0; b < 2; ++b)
          fr[u % (PFD + 1)][b] = *reinterpret_cast<const mbf16x8*>(pl + foff[ks][dx][b] + hi * ROWB);
      }
    };
    static_for<PFD>([&](auto k_c) __attribute__((always_inline)) { load_unit(k_c); });
    if constexpr (MODE == 2) dma_old(s + 1, (s + 1) & 1);   // (first in the step: the wait at the top of step s + 1 counts on it)
    static_for<NU>([&](auto u_c) __attribute__((always_inline)) {
      constexpr int u = decltype(u_c)::value;
      constexpr int hi = u / (3 * KS), ks = (u / 3) % KS, dx = u % 3;
      load_unit(std::integral_constant<int, u + PFD>{});
      static_for<ITEMS>([&](auto it_c) __attribute__((always_inline)) {
        constexpr int it = decltype(it_c)::value;
        if constexpr ((it * NU) / ITEMS == u) dma_item(pref, it_c);
      });
      if constexpr (ks == KS - 1 && dx >= 1) {
        if constexpr (hi + 1 < RYW)
          finish_blk(s, s - 3, std::integral_constant<int, PH>{}, std::integral_constant<int, hi + 1>{}, std::integral_constant<int, dx - 1>{});
        if constexpr (hi == HYW - 1)
          finish_blk(s, s - 2, std::integral_constant<int, (PH + 1) % 3>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, dx - 1>{});
      }
      if constexpr (hi == RYW && ks == 0 && dx == 0) flush_stats();
#ifdef SEUNET_MARCH_ASM_MFMA
      __builtin_amdgcn_sched_barrier(0);                 // (the next unit's reads stay ahead of this unit's MFMAs)
#endif
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int r = hi - DIL * dy;                    // output row (of the wave's RYW) fed by input row hi through tap dy
          if (r < 0 || r >= RYW) continue;
#pragma unroll
          for (int dz = 0; dz < 3; ++dz) {                // output plane s - dz
            const int ai = (PH - dz + 3) % 3;
            const int tap = (dz * 3 + dy) * 3 + dx;
#ifndef SEUNET_MARCH_ASM_MFMA
            if (dz == 0 && dy == 0 && dx == 0 && ks == 0) acc[ai][r][b] = mm16b<T>(wreg[tap * KS + ks], fr[u % (PFD + 1)][b], cinit);
            else acc[ai][r][b] = mm16b<T>(wreg[tap * KS + ks], fr[u % (PFD + 1)][b], acc[ai][r][b]);
#else
            if (dz == 0 && dy == 0 && dx == 0 && ks == 0) mm16_init<T, 0, MODE == 0>(acc[ai][r][b], wreg[tap * KS + ks], fr[u % (PFD + 1)][b], cinit);
            else mm16_acc<T, (ks == 1)>(acc[ai][r][b], wreg[tap * KS + ks], fr[u % (PFD + 1)][b]);
#endif
          }
        }
#ifndef SEUNET_MARCH_ASM_MFMA
      // one scheduling region per unit: the next unit's two fragment reads first, then the unit's MFMAs with the vector work of
      // the epilogue / DMA addressing dealt between them (an MFMA holds the SIMD's issue for 8 of its 16 cycles)
      {
        constexpr int NM = 6 * ((hi >= 2 * DIL && hi < RYW) ? 3 : ((hi >= DIL && hi < RYW + DIL) ? 2 : 1));
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- the march ----
  // Vector-memory operations of a wave, in program order: [prologue: DMA(0), DMA(1)], then per step s exactly TOT of them:
  // old rows (MODE 2, first), ITEMS plane instructions for plane s + 2, STORES stores.  At the top of step s >= 1 plane s
  // (issued in step s - 2, or in the prologue) must have landed, with accumulation the old rows of step s (issued first in
  // step s - 1) too: the only LOADS younger than those are the ITEMS plane instructions of step s - 1, so the wait allows
  // ITEMS outstanding operations and counts NO store as outstanding (round 4: until then it allowed TOT - OLDN, i.e. also the
  // stores of step s - 1 -- right only if stores and LDS-DMA loads retire in issue order, and they need not: the streaming conv's
  // 16-byte-store experiments passed such a wait before their plane had landed, DESIGN 4.  Same speed here).  Then one barrier:
  // every wave's part of the plane is in the LDS, and every wave has finished reading the slot (plane s - 1) that this step's
  // prefetch overwrites.
  static_for<MA_PF>([&](auto k_c) __attribute__((always_inline)) {
    constexpr int k = decltype(k_c)::value;
    const PlaneRef r = plane_of(k, k);
    static_for<ITEMS>([&](auto it_c) __attribute__((always_inline)) { dma_item(r, it_c); });
  });
  march_wait_vm<(MA_PF - 1) * ITEMS>();    // plane 0 has landed (this wave's part)
  __builtin_amdgcn_s_barrier();
  // Every step of the loop is a compute step and the trip count is padded to whole 3-step rounds, so that the loop body has no
  // conditional path (the accumulator sets stay in place: no copies, no merges); steps beyond the last input plane march over
  // zero planes (DMA from the zero page).  The rows 1.. of the LAST output plane Z - 1 (set (Z - 1) % 3) are written by the
  // lazy epilogue of step Z + 2 when the padding reaches that step; when Z + 2 is a multiple of 3 the loop ends before it and a
  // tail without MFMAs writes them (then (Z - 1) % 3 == 0: one variant of the tail).  launch_conv_march() prefers such
  // segment lengths.
  const int nrounds = (ncompute + 2) / 3;
  int slot = 0, slot_pf = MA_PF % MA_RING;
  for (int rd3 = 0; rd3 < nrounds; ++rd3) {
    static_for<3>([&](auto ph_c) __attribute__((always_inline)) {
      constexpr int PH = decltype(ph_c)::value;
      const int s = 3 * rd3 + PH;
      if (PH > 0 || rd3 > 0) {
        march_wait_vm<ITEMS>();
        __builtin_amdgcn_s_barrier();
      }
      compute(s, slot, slot_pf, ph_c);
      slot = slot == MA_RING - 1 ? 0 : slot + 1;
      slot_pf = slot_pf == MA_RING - 1 ? 0 : slot_pf + 1;
    });
  }
  march_wait_vm<0>();            // no DMA may outlive the workgroup's LDS allocation
  if (3 * nrounds == ncompute) {   // (wave-uniform) the tail: rows 1.. of output Z - 1, held in set 0, as step Z + 2 would have
    // (accumulation: a wave reads only the old rows it fetched itself, and they have landed: vmcnt(0) above)
    static_for<RYW - 1>([&](auto r_c) __attribute__((always_inline)) {
      constexpr int r = decltype(r_c)::value + 1;
      finish_blk(ncompute, ncompute - 3, std::integral_constant<int, 0>{}, std::integral_constant<int, r>{}, std::integral_constant<int, 0>{});
      finish_blk(ncompute, ncompute - 3, std::integral_constant<int, 0>{}, std::integral_constant<int, r>{}, std::integral_constant<int, 1>{});
    });
    flush_stats();
  }

  // ---- InstanceNorm partial sums of this workgroup: over the 16 lanes that hold the same channels, then over the row
  //      groups (fixed order), one record per workgroup ----
  if constexpr (MODE == 0) {
    if (a.stats != nullptr) {
      __syncthreads();           // the ring is dead: reuse its first bytes
      double* red = reinterpret_cast<double*>(smem);     // [4 waves][16 channels][2]
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double t1 = stot[e * 256], t2 = stot[(4 + e) * 256];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { t1 += shfl_xor_settled(t1, off); t2 += shfl_xor_settled(t2, off); }
        if (n16 == 0) {
          red[(wave * 16 + 4 * g + e) * 2] = t1;
          red[(wave * 16 + 4 * g + e) * 2 + 1] = t2;
        }
      }
      __syncthreads();
      if (tid < NGW * 32) {
        const int k = tid & 1, c = (tid >> 1) & 15, gi = tid >> 5;     // channel c of N group gi
        double tot = 0.0;
#pragma unroll
        for (int rr = 0; rr < RGW; ++rr) tot += red[((rr * NGW + gi) * 16 + c) * 2 + k];      // fixed order
        const int co = (nb * NGW + gi) * 16 + c;
        const long long slots = (long long)gridDim.x * a.nseg * DIL;
        const long long sl = (long long)segpz * gridDim.x + blockIdx.x;
        if (co < a.cout) a.stats[(((long long)n * slots + sl) * a.cout + co) * 2 + k] = tot;
      }
    }
  }
}

#ifndef SEUNET_MARCH_PROBE   /* (scripts/march_probe.sh compiles single instantiations of the kernel above) */
// ------------------------------------------------------------------------------------------------------------------
// weight packing: PyTorch (Cout, Cin, 3, 3, 3) f32 -> [16-channel group][tap][K-step][lane][8] MFMA A-operand fragments
// ------------------------------------------------------------------------------------------------------------------
struct MarchPackArgs { const float* w; void* out; int cin_w, cout_w, tflip, cin_e, cout_e, ks; };

template <typename T>
__global__ void __launch_bounds__(64)
conv_march_pack_kernel(MarchPackArgs p) {
  // blockIdx.x = (group * 27 + tap) * KS + ks; one wave writes the 64 fragments
  const int lane = threadIdx.x;
  const int ks = blockIdx.x % p.ks, gt = blockIdx.x / p.ks;
  const int tap = gt % 27, grp = gt / 27;
  const int co = grp * 16 + (lane & 15);
  const int kg = lane >> 4;
  T* out = reinterpret_cast<T*>(p.out) + ((size_t)blockIdx.x * 64 + lane) * 8;
  for (int j = 0; j < 8; ++j) {
    const int ci = 32 * ks + 8 * kg + j;
    float v = 0.f;
    if (co < p.cout_e && ci < p.cin_e)
      v = p.tflip ? p.w[((long long)ci * p.cin_w + co) * 27 + (26 - tap)] : p.w[((long long)co * p.cin_w + ci) * 27 + tap];
    out[j] = from_f32<T>(v);
  }
}

// all the weight tensors of a pass in one launch: blockIdx.y = list entry
struct MarchPackList { MarchPackArgs e[12]; int blocks[12]; };
template <typename T>
__global__ void __launch_bounds__(64)
conv_march_pack_multi_kernel(MarchPackList l) {
  const MarchPackArgs& p = l.e[blockIdx.y];
  if ((int)blockIdx.x >= l.blocks[blockIdx.y]) return;
  const int lane = threadIdx.x;
  const int ks = blockIdx.x % p.ks, gt = blockIdx.x / p.ks;
  const int tap = gt % 27, grp = gt / 27;
  const int co = grp * 16 + (lane & 15);
  const int kg = lane >> 4;
  T* out = reinterpret_cast<T*>(p.out) + ((size_t)blockIdx.x * 64 + lane) * 8;
  for (int j = 0; j < 8; ++j) {
    const int ci = 32 * ks + 8 * kg + j;
    float v = 0.f;
    if (co < p.cout_e && ci < p.cin_e)
      v = p.tflip ? p.w[((long long)ci * p.cin_w + co) * 27 + (26 - tap)] : p.w[((long long)co * p.cin_w + ci) * 27 + tap];
    out[j] = from_f32<T>(v);
  }
}

// ---- configuration by channel counts -------------------------------------------------------------------------------
// mode as in the kernel: 0 forward, 1 data gradient, 2 data gradient with accumulation
struct MarchCfg { int ks, ngw, ryw; };
static bool march_cfg(int dtype, int taps, int dil, int cin_e, int cout_e, int mode, MarchCfg& c) {
  if ((dtype != SEUNET_BF16 && dtype != SEUNET_F16) || taps != 27 || (dil != 1 && dil != 2)) return false;
  if (cin_e != 32 && cin_e != 64) return false;
  if (cout_e < 32 || cout_e % 32 != 0) return false;
  c.ks = cin_e / 32;
  if (cout_e % 64 == 0) { c.ngw = 4; c.ryw = c.ks == 1 ? 8 : 4; }
  else { c.ngw = 2; c.ryw = (c.ks == 2 && (dil == 2 || mode == 2)) ? 2 : 4; }   // (64-channel planes of 8 + 4 rows do not fit three times)
  return true;
}
bool conv_march_supported(int dtype, int taps, int dil, const SrcList& src, const DstList& dst) {
  MarchCfg c;
  if (src.n < 1 || src.n > 2 || dst.n < 1 || dst.n > 3) return false;
  if (src.n == 2 && src.C[0] != src.C[1]) return false;
  for (int i = 0; i < src.n; ++i) if (src.C[i] % 8 != 0) return false;
  for (int i = 0; i < dst.n; ++i) if (dst.C[i] % 16 != 0) return false;
  return march_cfg(dtype, taps, dil, src.total(), dst.total(), 0, c);
}
size_t conv_march_wpack_bytes(int cin_e, int cout_e) { return (size_t)(cout_e / 16) * 27 * (cin_e / 32) * 64 * 16; }

static int march_patch_rows(const MarchCfg& c) { return c.ryw * (MA_NW / c.ngw); }
// output planes per march: the fewest steps on the critical path -- rounds of workgroups over the chip's 256 CUs (one
// workgroup per CU) x steps of the longest march, a march of Z planes running Z + 2 steps rounded up to a multiple of 3
static int march_zsteps(int planes, long long wg_per_seg) {
  int best = planes;
  long long best_cost = -1;
  for (int zs = planes; zs >= 1 && zs * 8 >= planes; --zs) {
    const int segs = (planes + zs - 1) / zs;
    const long long rounds = (wg_per_seg * segs + 255) / 256;
    const long long cost = rounds * ((zs + 2 + 2) / 3 * 3);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = zs; }
  }
  return best;
}
int conv_march_slots(Dims d, int dil, int cin_e, int cout_e) {
  MarchCfg c;
  if (!march_cfg(SEUNET_BF16, 27, dil, cin_e, cout_e, 0, c)) return 0;
  const int ry = march_patch_rows(c);
  const int npatch = cdiv(d.H, ry) * cdiv(d.W, MA_TX);
  const int planes = cdiv(d.D, dil);
  const int nblk = cout_e / (16 * c.ngw);
  const int zs = march_zsteps(planes, (long long)npatch * d.N * nblk * dil);
  return npatch * cdiv(planes, zs) * dil;
}

int launch_conv_march_pack(int dtype, const float* w, int cin_w, int cout_w, int tflip, int cin_e, int cout_e, void* wpack, hipStream_t s) {
  MarchCfg c;
  SEUNET_CHECK(march_cfg(dtype, 27, 1, cin_e, cout_e, 0, c) && w && wpack, "conv_march_pack: unsupported shape (%d -> %d channels)", cin_e, cout_e);
  const int we_in = tflip ? cout_w : cin_w, we_out = tflip ? cin_w : cout_w;
  SEUNET_CHECK(we_in <= cin_e && we_out <= cout_e, "conv_march_pack: weight (%d -> %d) exceeds the tensors (%d -> %d)", we_in, we_out, cin_e, cout_e);
  MarchPackArgs p{w, wpack, cin_w, cout_w, tflip, we_in, we_out, c.ks};
  const int blocks = (cout_e / 16) * 27 * c.ks;
  if (dtype == SEUNET_F16) conv_march_pack_kernel<f16_t><<<blocks, 64, 0, s>>>(p);
  else conv_march_pack_kernel<bf16_t><<<blocks, 64, 0, s>>>(p);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_conv_march_pack_multi(int dtype, const MarchPackJob* jobs, int n, hipStream_t s) {
  for (int base = 0; base < n; base += 12) {
    MarchPackList l{};
    const int m = n - base < 12 ? n - base : 12;
    int maxb = 0;
    for (int i = 0; i < m; ++i) {
      const MarchPackJob& j = jobs[base + i];
      MarchCfg c;
      SEUNET_CHECK(march_cfg(dtype, 27, 1, j.cin_e, j.cout_e, 0, c) && j.w && j.wpack, "conv_march_pack: unsupported shape (%d -> %d channels)", j.cin_e, j.cout_e);
      const int we_in = j.tflip ? j.cout_w : j.cin_w, we_out = j.tflip ? j.cin_w : j.cout_w;
      SEUNET_CHECK(we_in <= j.cin_e && we_out <= j.cout_e, "conv_march_pack: weight (%d -> %d) exceeds the tensors (%d -> %d)", we_in, we_out, j.cin_e, j.cout_e);
      l.e[i] = MarchPackArgs{j.w, j.wpack, j.cin_w, j.cout_w, j.tflip, we_in, we_out, c.ks};
      l.blocks[i] = (j.cout_e / 16) * 27 * c.ks;
      maxb = l.blocks[i] > maxb ? l.blocks[i] : maxb;
    }
    if (dtype == SEUNET_F16) conv_march_pack_multi_kernel<f16_t><<<dim3(maxb, m), 64, 0, s>>>(l);
    else conv_march_pack_multi_kernel<bf16_t><<<dim3(maxb, m), 64, 0, s>>>(l);
  }
  SEUNET_LAUNCH_CHECK();
  return 0;
}

template <typename T, int KS, int NGW, int RYW, int DIL, int MODE, bool BUF>
static int march_launch_buf(const MarchArgs& a, dim3 grid, hipStream_t s) {
  using Geo = MarchGeo<KS, NGW, RYW, DIL, MODE>;
  static unsigned long long configured = 0;
  if (int e = configure_kernel_lds(configured, reinterpret_cast<const void*>(&conv_march_kernel<T, KS, NGW, RYW, DIL, MODE, BUF>), Geo::LDS)) return e;
  conv_march_kernel<T, KS, NGW, RYW, DIL, MODE, BUF><<<grid, MA_NW * 64, Geo::LDS, s>>>(a);
  SEUNET_LAUNCH_CHECK();
  return 0;
}
// the data gradients always have one source (the gradient of the raw conv output); the forward has one or two
template <typename T, int KS, int NGW, int RYW, int DIL, int MODE>
static int march_launch(const MarchArgs& a, dim3 grid, hipStream_t s) {
  if constexpr (MODE == 0) {
    if (!a.buf) return march_launch_buf<T, KS, NGW, RYW, DIL, MODE, false>(a, grid, s);   // two sources too far apart for one descriptor
  }
  return march_launch_buf<T, KS, NGW, RYW, DIL, MODE, true>(a, grid, s);
}
template <typename T, int KS, int NGW, int RYW, int DIL>
static int march_launch_mode(int mode, const MarchArgs& a, dim3 grid, hipStream_t s) {
  if (mode == 0) return march_launch<T, KS, NGW, RYW, DIL, 0>(a, grid, s);
  if (mode == 1) return march_launch<T, KS, NGW, RYW, DIL, 1>(a, grid, s);
  return march_launch<T, KS, NGW, RYW, DIL, 2>(a, grid, s);
}
template <typename T, int DIL>
static int march_launch_cfg(const MarchCfg& c, int mode, const MarchArgs& a, dim3 grid, hipStream_t s) {
  if (c.ks == 1 && c.ngw == 4) return march_launch_mode<T, 1, 4, 8, DIL>(mode, a, grid, s);
  if (c.ks == 2 && c.ngw == 4) return march_launch_mode<T, 2, 4, 4, DIL>(mode, a, grid, s);
  if (c.ks == 1) return march_launch_mode<T, 1, 2, 4, DIL>(mode, a, grid, s);
  if constexpr (DIL == 1) {
    if (mode == 0) return march_launch<T, 2, 2, 4, 1, 0>(a, grid, s);
    if (mode == 1) return march_launch<T, 2, 2, 4, 1, 1>(a, grid, s);
    return march_launch<T, 2, 2, 2, 1, 2>(a, grid, s);
  } else {
    return march_launch_mode<T, 2, 2, 2, DIL>(mode, a, grid, s);
  }
}

// src: one or two [N][D][H][W][C] tensors (32 or 64 channels together); dst: channel split of the output (multiples of 16)
int launch_conv_march(int dtype, int dil, const SrcList& src, const void* wpack, const float* bias, const DstList& dst, double* stats,
                      Dims d, hipStream_t s) {
  SEUNET_CHECK(conv_march_supported(dtype, 27, dil, src, dst),
               "conv_march: unsupported shape (%d -> %d channels, dilation %d, dtype %d)", src.total(), dst.total(), dil, dtype);
  SEUNET_CHECK(wpack != nullptr, "conv_march: null weights");
  bool any_acc = false;
  for (int i = 0; i < dst.n; ++i) {
    any_acc |= dst.acc[i] != 0 && dst.ptr[i] != nullptr;
    SEUNET_CHECK((long long)d.vox() * dst.C[i] * 2 < (1LL << 31), "conv_march: one sample of destination %d exceeds the 32-bit offsets of this kernel", i);
  }
  SEUNET_CHECK((long long)d.H * d.W * src.C[0] * 2 < (1LL << 31), "conv_march: one plane of the source exceeds the 31-bit offsets of this kernel");
  const bool fwd_probe = bias != nullptr || stats != nullptr;
  const long long sample_src = (long long)d.vox() * src.C[0] * 2;
  long long span = sample_src;           // bytes from the lower source's sample to the end of the upper source's sample
  if (src.n > 1) {
    const long long dist = reinterpret_cast<const char*>(src.ptr[1]) - reinterpret_cast<const char*>(src.ptr[0]);
    span += dist < 0 ? -dist : dist;
  }
  const bool buf_ok = span < (1LL << 32);
  SEUNET_CHECK(buf_ok || (fwd_probe && src.n > 1), "conv_march: the source of one sample exceeds the 32-bit offsets of this kernel");
  const bool fwd = bias != nullptr || stats != nullptr;
  SEUNET_CHECK(!(fwd && any_acc), "conv_march: accumulation into the destination is a data-gradient feature (no bias, no statistics)");
  SEUNET_CHECK(!stats || dst.n == 1, "conv_march: statistics need a single destination");
  MarchArgs a{};
  a.src0 = src.ptr[0]; a.src1 = src.n > 1 ? src.ptr[1] : nullptr; a.srcC = src.C[0]; a.nsrc = src.n;
  a.wpack = wpack; a.bias = bias;
  a.dst0 = dst.ptr[0]; a.dstC0 = dst.C[0]; a.dacc0 = dst.acc[0];
  a.dst1 = dst.n > 1 ? dst.ptr[1] : nullptr; a.dstC1 = dst.n > 1 ? dst.C[1] : 0; a.dacc1 = dst.n > 1 ? dst.acc[1] : 0;
  a.dst2 = dst.n > 2 ? dst.ptr[2] : nullptr; a.dstC2 = dst.n > 2 ? dst.C[2] : 0; a.dacc2 = dst.n > 2 ? dst.acc[2] : 0;
  a.cout = dst.total();
  a.dcum1 = dst.n > 1 ? dst.C[0] : a.cout;
  a.dcum2 = dst.n > 2 ? dst.C[0] + dst.C[1] : a.cout;
  a.stats = stats; a.zero = device_zero_page();
  a.buf = buf_ok ? 1 : 0;
  SEUNET_CHECK(a.zero != nullptr, "conv_march: no zero page on this device");
  a.N = d.N; a.D = d.D; a.H = d.H; a.W = d.W;
  const int mode = fwd ? 0 : (any_acc ? 2 : 1);
  MarchCfg c;
  march_cfg(dtype, 27, dil, src.total(), dst.total(), mode, c);
  const int ry = march_patch_rows(c);
  a.nyb = cdiv(d.H, ry); a.nxb = cdiv(d.W, MA_TX);
  a.nblk = a.cout / (16 * c.ngw);
  const int planes = cdiv(d.D, dil);
  a.zsteps = march_zsteps(planes, (long long)a.nyb * a.nxb * d.N * a.nblk * dil);
  a.nseg = cdiv(planes, a.zsteps);
  SEUNET_CHECK(d.N <= 65535 && (long long)a.nseg * dil * a.nblk <= 65535, "conv_march: grid too large");
  dim3 grid(a.nyb * a.nxb, a.nseg * dil * a.nblk, d.N);
  if (dtype == SEUNET_F16) return dil == 1 ? march_launch_cfg<f16_t, 1>(c, mode, a, grid, s) : march_launch_cfg<f16_t, 2>(c, mode, a, grid, s);
  return dil == 1 ? march_launch_cfg<bf16_t, 1>(c, mode, a, grid, s) : march_launch_cfg<bf16_t, 2>(c, mode, a, grid, s);
}

#endif  // SEUNET_MARCH_PROBE

}  // namespace seunet
