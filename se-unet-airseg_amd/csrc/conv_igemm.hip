// Implicit-GEMM convolution for gfx950 matrix cores: 3x3x3 (dilation 1 or 2, "same" zero padding) and
// 1x1x1, forward and data-gradient (a data-gradient is the same kernel run on flipped/transposed
// weights).  Reference ops: nn.Conv3d at SE_UNet.py:15,42,57 (+ torch.cat at :186,195,204,212,216,218,
// 222,224,228 which is fused here as a multi-pointer channel concatenation).
//
//   GEMM view     output channels x voxels, K = taps x input channels.  The WEIGHTS are the MFMA's A operand, so the
//                 accumulators hold the transposed tile: a lane owns one voxel (x = lane % 32) and runs of 4
//                 consecutive output channels -- what a channels-last store wants
//   workgroup     256 threads = 4 waves; output tile 4(z) x 4(y) x 32(x) voxels; wave w owns z-slice w,
//                 i.e. four 32-voxel x-rows, times all output-channel columns of the tile (32 or 64)
//   dilation 2    decomposes into 8 independent dilation-1 problems on the parity sub-lattices
//                 (voxel = 2*lattice + parity): the tile lives on one sub-lattice, so the halo is 1 lattice
//                 voxel (39 KB tile) instead of 2 voxels (74 KB), and the rest of the kernel is unchanged
//   MFMA          bf16: v_mfma_f32_32x32x16_bf16 (K-step = 16 channels of one tap)
//                 f32 : v_mfma_f32_32x32x2_f32   (exact f32 FMA chain; the 1e-3 parity mode)
//   LDS           input halo tile, planar: [16-B piece of the 32-B channel chunk][6*6*34 voxels][16 B]; weight slab
//                 [tap][k-half][column][16 B].  Every fragment address is "lane base + immediate": no swizzle, no
//                 address arithmetic in the K loop, conflict-free reads and writes
//   K loop        chunks of 32 bytes of channels (16 bf16 / 8 f32).  The NEXT chunk's tile and weights are fetched
//                 into registers while the current chunk's 27 taps of MFMAs run, and written to LDS after the barrier.
//                 The fetches are buffer_load_dwordx4 through wave-uniform descriptors: 32-bit offsets, hardware range
//                 check returns zeros for padding voxels / channels; lane pairs fetch the two pieces of one voxel
//                 (32 contiguous bytes) unless a concatenation boundary splits a chunk.  The MFMA loop is pipelined
//                 by hand: the fragments of tap t+1 are requested before the MFMAs of tap t issue
//   epilogue      + bias; per-(n,c) InstanceNorm partial sums by a DPP reduce-scatter over the 32 voxel lanes (f32 tree
//                 for bf16 activations, exact f64 in the f32 parity mode) and a fixed-order cross-wave sum behind one
//                 LDS-only barrier; stores straight from the accumulators when every destination has <= 16 channels,
//                 otherwise through a wave-private ds_write_b64 restage so that each lane stores 16 contiguous bytes;
//                 optionally += (gradient accumulation) and split over up to three destination tensors (backward of
//                 the fused concatenation)
//   grid          blockIdx.x is remapped so that each XCD (private L2) owns a contiguous run of tiles
#include "seunet_common.h"
#include <utility>
#include <type_traits>

namespace seunet {

typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef f16_t f16x8 __attribute__((ext_vector_type(8)));
// LDS fragments are read as 8 x 16-bit patterns (bf16x8); the matrix instruction is chosen by the storage type
template <typename T> __device__ __forceinline__ float __attribute__((ext_vector_type(16)))
mfma32_16bit(bf16x8 a, bf16x8 b, float __attribute__((ext_vector_type(16))) c) {
  if constexpr (std::is_same<T, f16_t>::value)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static constexpr int CV_TZ = 4, CV_TY = 4, CV_TX = 32;

struct ConvKArgs {
  const void* src0; const void* src1; const void* src2;
  int srcC0, srcC1, srcC2;
  int cum1, cum2;            // first virtual channel of source 1 / 2
  int cin;                   // valid input channels
  const void* wpack;
  const float* bias;
  void* dst0; void* dst1; void* dst2;
  int dstC0, dstC1, dstC2;
  int dcum1, dcum2;
  int dacc0, dacc1, dacc2;
  int cout;
  double* stats;
  int N, D, H, W;
  int tx, ty, tz;            // tile counts (on the sub-lattice when dilated)
  int nchunks;
  int pair;                  // 1: both 16-B pieces of every channel chunk come from one source tensor (lane-pair staging)
  int direct;                // 1: every destination has <= 16 channels (stores straight from the accumulators)
  int ncb;                   // column blocks per tile (gridDim.x = tiles * ncb, column block fastest)
  unsigned long long* debug;   // diagnostic builds only (-DSEUNET_STAMP): per-phase cycle sums
};

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { static constexpr int KC = 16, KSTEPS = 1; };
template <> struct Frag<f16_t> { static constexpr int KC = 16, KSTEPS = 1; };
template <> struct Frag<float> { static constexpr int KC = 8, KSTEPS = 4; };

// cross-lane fetches for f32 and f64 values: DPP (row-local patterns) and ds_swizzle xor 16 (within 32 lanes)
template <int CTRL> __device__ __forceinline__ float xlane_dpp(float v) { return dpp_fetch<CTRL>(v); }
template <int CTRL> __device__ __forceinline__ double xlane_dpp(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)dpp_fetch_bits<CTRL>((int)(unsigned)u);
  const unsigned hi = (unsigned)dpp_fetch_bits<CTRL>((int)(unsigned)(u >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float xlane_swz16(float v) {
  return dpp_settle(__builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F)));
}
__device__ __forceinline__ double xlane_swz16(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)dpp_settle(__builtin_amdgcn_ds_swizzle((int)(unsigned)u, 0x401F));
  const unsigned hi = (unsigned)dpp_settle(__builtin_amdgcn_ds_swizzle((int)(unsigned)(u >> 32), 0x401F));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

#ifdef SEUNET_STAMP
#define STAMP(i) do { const unsigned long long _t = __builtin_readcyclecounter(); ph[i] += _t - t_last; t_last = _t; } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

template <typename T, int NSUB, int TAPS, int DIL>
__global__ void __launch_bounds__(256, (TAPS == 1 && NSUB == 1) ? 3 : 1)   // 1x1x1 is bandwidth bound: keep 3 workgroups per CU
conv_igemm_kernel(ConvKArgs a) {
#ifdef SEUNET_STAMP
  unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_readcyclecounter();
#endif
  constexpr int KC = Frag<T>::KC, KSTEPS = Frag<T>::KSTEPS;
  constexpr int HALO = (TAPS == 27) ? 1 : 0;
  constexpr int STEP = (TAPS == 27) ? DIL : 1;              // voxel stride of the (sub-)lattice
  constexpr int HZ = CV_TZ + 2 * HALO, HY = CV_TY + 2 * HALO, HX = CV_TX + 2 * HALO;
  constexpr int NVH = HZ * HY * HX;
  constexpr int NCOL = 32 * NSUB;
  constexpr int T3 = (TAPS == 27) ? 3 : 1;
  constexpr int IN_ITEMS = (NVH * 2 + 255) / 256;           // 16-B pieces per thread (LDS region padded to IN_ITEMS*4 KB)
  constexpr int W_TOTAL = TAPS * NCOL * 2;                  // 16-byte pieces of one weight slab
  constexpr int W_ITEMS = (W_TOTAL + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* in_tile = smem;
  unsigned char* w_tile = smem + IN_ITEMS * 4096 + 128;   // after the two (64-B padded) input planes

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  // XCD-aware order: blocks b, b+8, b+16.. share an XCD; give each XCD a contiguous run of (tile, column block) pairs
  // with the column block fastest, so the workgroups that read the SAME input tile (one per 32/64-column block of the
  // outputs) run back to back on one XCD and the second finds the tile in that XCD's L2
  int t;
  {
    const int nt = gridDim.x, b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int ntile = t % a.ncb;
  t /= a.ncb;
  const int tiles_per_sample = gridDim.x / a.ncb;
  const int tile_slot = t;
  const int bx = t % a.tx; t /= a.tx;
  const int by = t % a.ty; t /= a.ty;
  const int bz = t % a.tz;
  const int par = t / a.tz;                                  // parity class 0..STEP^3-1
  const int px = par % STEP, py = (par / STEP) % STEP, pz = par / (STEP * STEP);
  const int x0 = bx * CV_TX, y0 = by * CV_TY, z0 = bz * CV_TZ;   // lattice coordinates
  const int n = blockIdx.z;
  const long long V = (long long)a.D * a.H * a.W;

  // ---- per-thread staging plan (chunk independent) ----
  // A staging instruction moves 64 16-byte pieces.  Paired mode (a.pair; whenever the two pieces of every 32-B channel
  // chunk come from the same source tensor): lanes (2i, 2i+1) take both pieces of one voxel -- 32 contiguous bytes, 32
  // voxels per instruction -- which halves the cache lines an instruction touches; the fetch phase is bound by the
  // texture path at about one line per clock.  Otherwise (a concatenation boundary inside a chunk) the piece is
  // wave-uniform (wave & 1) and an instruction covers 64 consecutive voxels.
  const bool pair = a.pair != 0;                                   // wave-uniform
  const int piece = pair ? (lane & 1) : (wave & 1);
  const int vox0 = pair ? wave * 32 + (lane >> 1) : (wave >> 1) * 64 + lane;   // voxel of item k: vox0 + 128 * k
  constexpr unsigned INVALID = 0xFFFFFFFFu;
  unsigned vofs[IN_ITEMS];
#pragma unroll
  for (int k = 0; k < IN_ITEMS; ++k) {
    const int vox = vox0 + 128 * k;
    const int hx = vox % HX;
    const int r2 = vox / HX;
    const int hy = r2 % HY, hz = r2 / HY;
    const int lz = z0 - HALO + hz, ly = y0 - HALO + hy, lx = x0 - HALO + hx;
    const int gz = STEP * lz + pz, gy = STEP * ly + py, gx = STEP * lx + px;
    const bool ok = vox < NVH && lz >= 0 && ly >= 0 && lx >= 0 && gz < a.D && gy < a.H && gx < a.W;
    vofs[k] = ok ? (unsigned)((gz * a.H + gy) * a.W + gx) : INVALID;
  }
  STAMP(8);   // index plan
  // LDS image of the halo tile: planar, [16-B piece of the 32-B chunk][voxel][16 B] -- a staging instruction writes 1 KB
  // contiguous, an MFMA fragment read (32 consecutive voxels of one piece) is 512 B contiguous: both conflict-free with
  // no swizzle, so every fragment address is "lane base + compile-time offset" (the offset field of ds_read)
  constexpr int PLANE = IN_ITEMS * 2048 + 64;   // (+64 B: in paired mode even / odd lanes write different planes; keeps them on different banks)
  const int lds_in0 = piece * PLANE + vox0 * 16;   // + k * 2048
  // fragment bases of this lane: voxel (z-slice of the wave, x = col), k-half h
  const unsigned char* afrag0 = in_tile + (sizeof(T) == 2 ? h * PLANE : 4 * h) + (wave * HY * HX + col) * 16;
  const unsigned char* wfrag0 = w_tile + (h * NCOL + col) * (sizeof(T) == 2 ? 16 : 4);
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(a.wpack) +
                               (size_t)ntile * a.nchunks * (size_t)(W_TOTAL * 16);
  u32x4 rin[IN_ITEMS], rw[W_ITEMS];
  auto uniform_ptr = [](const void* p) -> const void* {   // tell the compiler the descriptor base is wave-uniform
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
  };
  // The fetch of one chunk is split into its descriptors (scalar work, once per chunk) and IN_ITEMS + W_ITEMS single
  // wave-instructions, so that the K loop can issue them one per tap between the MFMAs: issued as one block they
  // hold the wave for ~1.5k cycles (the texture path takes 16+ cycles per 1-KB instruction) with the matrix pipe idle.
  __amdgpu_buffer_rsrc_t rs_in, rs_w;
  unsigned in_stride = 0;
  unsigned pofs = 0;        // paired mode: byte offset of this lane's piece inside the chunk
  bool pbad = false;        // paired mode: this lane's piece lies beyond the last input channel (reads zeros)
  auto fetch_setup = [&](int chunk) __attribute__((always_inline)) {
    const int chb = chunk * KC;                                   // first channel of the chunk
    const int ch0 = pair ? chb : chb + piece * (KC / 2);          // (unpaired: piece is wave-uniform)
    const void* sp = a.src0; int sC = a.srcC0, c = ch0;
    if (ch0 >= a.cum2) { sp = a.src2; sC = a.srcC2; c = ch0 - a.cum2; }
    else if (ch0 >= a.cum1) { sp = a.src1; sC = a.srcC1; c = ch0 - a.cum1; }
    const T* base = reinterpret_cast<const T*>(sp) + (long long)n * V * sC + c;
    const long long avail = ch0 < a.cin ? ((long long)V * sC - c) * (long long)sizeof(T) : 0;   // 0 records: all zeros
    rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_ptr(base)), 0,
                                              __builtin_amdgcn_readfirstlane((int)avail), 0x00020000);
    in_stride = __builtin_amdgcn_readfirstlane((unsigned)(sC * (int)sizeof(T)));
    if (pair) {
      pofs = (unsigned)piece * 16u;
      pbad = piece == 1 && chb + KC / 2 >= a.cin;
    }
    rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_ptr(wbase + (size_t)chunk * (W_TOTAL * 16))), 0,
                                             W_TOTAL * 16, 0x00020000);
  };
  auto fetch_item = [&](auto item_c) __attribute__((always_inline)) {   // item: 0..IN_ITEMS-1 input pieces, then W_ITEMS weight pieces
    constexpr int item = decltype(item_c)::value;
    if constexpr (item < IN_ITEMS) {
      // byte offset = voxel index x stride, formed at the point of issue from a fresh copy of the stride (the empty asm
      // keeps the compiler from hoisting all IN_ITEMS products out of the tap loop into 10 more live registers).
      // Padding voxels carry index 0xFFFFFFFF: the product wraps to 2^32 - stride (+ < stride), beyond any tensor the
      // 32-bit range check admits (launch_conv_igemm rejects >= 2^31-byte samples), so the hardware returns zeros.
      unsigned st = in_stride;
      asm volatile("" : "+s"(st));
      const unsigned off = pbad ? 0x80000000u : vofs[item] * st + pofs;
      rin[item] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
    } else if constexpr (item < IN_ITEMS + W_ITEMS) {
      constexpr int k = item - IN_ITEMS;
      rw[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)tid * 16u, k * 4096, 0);
    }
  };
  constexpr int F_ITEMS = IN_ITEMS + W_ITEMS;
  constexpr int F_PER_TAP = (F_ITEMS + TAPS - 1) / TAPS;
  auto fetch_range = [&](auto first_c) __attribute__((always_inline)) {   // items [first, first + F_PER_TAP)
    constexpr int first = decltype(first_c)::value;
    [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
      (fetch_item(std::integral_constant<int, first + I>{}), ...);
    }(std::make_integer_sequence<int, F_PER_TAP>{});
  };

  f32x16 acc[4][NSUB];
#pragma unroll
  for (int ms = 0; ms < 4; ++ms)
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;

  fetch_setup(0);
  [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
    (fetch_item(std::integral_constant<int, I>{}), ...);
  }(std::make_integer_sequence<int, F_ITEMS>{});
  STAMP(0);   // prologue: index plan + first prefetch issue
  for (int chunk = 0; chunk < a.nchunks; ++chunk) {
    __syncthreads();   // every wave is done reading the previous chunk's tiles
    STAMP(1);   // barrier 1
#pragma unroll
    for (int k = 0; k < IN_ITEMS; ++k) *reinterpret_cast<u32x4*>(in_tile + lds_in0 + k * 2048) = rin[k];
#pragma unroll
    for (int k = 0; k < W_ITEMS; ++k) *reinterpret_cast<u32x4*>(w_tile + (tid + 256 * k) * 16) = rw[k];
    STAMP(2);   // wait for the fetched registers + LDS writes
    __syncthreads();
    STAMP(3);   // barrier 2
    // Next chunk's fetch.  64 columns (one workgroup per CU, nothing else to fill the matrix pipe): issued one item
    // per tap between the MFMAs below (measured -9..-11 % on 128->64 @64^3).  32 columns (two workgroups per CU): as
    // one block here -- interleaved it ran 20 % slower (a wave blocked on a full vector-memory queue cannot issue
    // its MFMAs either, while as a block the other workgroup's MFMAs cover the issue time).
    constexpr bool INTERLEAVE = NSUB == 2;
    const bool has_next = chunk + 1 < a.nchunks;
    if (has_next) {
      fetch_setup(chunk + 1);
      if constexpr (!INTERLEAVE) {
        [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
          (fetch_item(std::integral_constant<int, I>{}), ...);
        }(std::make_integer_sequence<int, F_ITEMS>{});
      }
    }
    STAMP(4);   // prefetch issue

    // ---- MFMA over taps, software-pipelined by hand ----
    // One step = one tap (bf16) or one (tap, 2-channel K-step) (f32): 1 + ... fragments of weights (NSUB) and voxels (4
    // y-rows), 4 * NSUB MFMAs.  The fragments of step s + 1 are requested before the MFMAs of step s issue (two
    // register sets, compile-time indices); __builtin_amdgcn_sched_barrier keeps that order, so the live set is what is
    // written here -- left to the scheduler the loop either waited for each read (lgkmcnt(0) before most MFMAs) or, fully
    // unrolled, hoisted reads until the kernel no longer fitted two workgroups per CU.
    {
      constexpr int NST = TAPS * KSTEPS;
      typedef typename std::conditional<sizeof(T) == 2, bf16x8, float>::type FragT;
      FragT wf[2][NSUB], af[2][4];
      auto load_step = [&](auto st_c) __attribute__((always_inline)) {
        constexpr int st = decltype(st_c)::value;
        if constexpr (st < NST) {
          constexpr int tap = st / KSTEPS, ks = st % KSTEPS, b = st & 1;
          constexpr int tz3 = tap / (T3 * T3), ty3 = (tap / T3) % T3, tx3 = tap % T3;
          constexpr int voff = ((tz3 * HALO) * HY + ty3 * HALO) * HX + tx3 * HALO;
          if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns)
              wf[b][ns] = *reinterpret_cast<const bf16x8*>(wfrag0 + (tap * 2 * NCOL + ns * 32) * 16);
#pragma unroll
            for (int ms = 0; ms < 4; ++ms)
              af[b][ms] = *reinterpret_cast<const bf16x8*>(afrag0 + (voff + ms * HX) * 16);
          } else {
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns)
              wf[b][ns] = *reinterpret_cast<const float*>(wfrag0 + ((tap * 4 + ks) * 2 * NCOL + ns * 32) * 4);
#pragma unroll
            for (int ms = 0; ms < 4; ++ms)
              af[b][ms] = *reinterpret_cast<const float*>(afrag0 + (ks >> 1) * PLANE + 8 * (ks & 1) + (voff + ms * HX) * 16);
          }
        }
      };
      load_step(std::integral_constant<int, 0>{});
      [&]<int... ST>(std::integer_sequence<int, ST...>) __attribute__((always_inline)) {
        ([&]() __attribute__((always_inline)) {
          constexpr int b = ST & 1;
          load_step(std::integral_constant<int, ST + 1>{});
          __builtin_amdgcn_sched_barrier(0);   // (the reads stay ahead of this step's MFMAs)
          if constexpr (INTERLEAVE && (ST % KSTEPS) == 0) {
            if (has_next) fetch_range(std::integral_constant<int, (ST / KSTEPS) * F_PER_TAP>{});   // wave-uniform
          }
#pragma unroll
          for (int ms = 0; ms < 4; ++ms)
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) {
              if constexpr (sizeof(T) == 2)
                acc[ms][ns] = mfma32_16bit<T>(wf[b][ns], af[b][ms], acc[ms][ns]);
              else
                acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[b][ns], af[b][ms], acc[ms][ns], 0, 0, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
        }(), ...);
      }(std::make_integer_sequence<int, NST>{});
    }
    STAMP(5);   // MFMA block of this chunk
  }

  // ---- epilogue ----
  // The MFMAs ran with the weights as the A operand, so the accumulators are the TRANSPOSED tile: lane (col, h) holds
  // voxel x = col of the four y-rows ms, and register r of acc[ms][ns] is output channel
  //     ch(ns, r, h) = ns*32 + (r & 3) + 8*(r >> 2) + 4*h,
  // i.e. four runs of 4 consecutive channels per lane: exactly what a channels-last store wants (8-B / 16-B pieces,
  // the h = 0 / 1 lanes writing adjacent pieces), with no transposition through LDS and no workgroup barrier between
  // the K loop and the stores.
  const int gz_w = STEP * (z0 + wave) + pz;
  const int gx_l = STEP * (x0 + col) + px;
  unsigned vlin[4];   // linear voxel index of this lane in y-row ms; 0xFFFFFFFF outside the volume
  bool vok[4];
#pragma unroll
  for (int ms = 0; ms < 4; ++ms) {
    const int gy = STEP * (y0 + ms) + py;
    vok[ms] = gz_w < a.D && gy < a.H && gx_l < a.W;
    vlin[ms] = vok[ms] ? (unsigned)((gz_w * a.H + gy) * a.W + gx_l) : INVALID;
  }
  // (1) bias
  if (a.bias != nullptr) {
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = ntile * NCOL + ns * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float bias = co < a.cout ? a.bias[co] : 0.f;
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) acc[ms][ns][r] += bias;
      }
  }
  // (2) InstanceNorm partial sums of this wave's 128 voxels (fixed order, hence deterministic).  Per channel the
  //     deviations from one of its values (the wave's first voxel) are summed in f32 -- they are of the order of the
  //     standard deviation, so nothing cancels -- and the shift is undone in f64.  The 32 per-lane values (16 sums,
  //     16 sums of squares) are reduced over the 32 voxel lanes of each half-wave by a reduce-scatter on DPP:
  //     5 steps of "keep half of my values, add the partner's copy of them", partners l^16, l^8, l^7, l^2, l^1
  //     (ds_swizzle, row_ror:8, row_half_mirror, quad_perm), after which lane j of a half owns value j.
  constexpr int K_BYTES = (IN_ITEMS + W_ITEMS) * 4096 + 128;   // the K-loop tiles; slower waves may still be reading them
  constexpr int E_BYTES = 4 * 128 * (NCOL * (int)sizeof(T) + 16);   // the four waves' store stages (step 3)
  double* red = reinterpret_cast<double*>(smem + (K_BYTES > E_BYTES ? K_BYTES : E_BYTES));   // [4 waves][NCOL][2]
  if (a.stats != nullptr) {
    float cntf = 0.f;
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) cntf += vok[ms] ? 1.f : 0.f;
    cntf += dpp_fetch<0xB1>(cntf); cntf += dpp_fetch<0x4E>(cntf); cntf += dpp_fetch<0x141>(cntf); cntf += dpp_fetch<0x140>(cntf);
    cntf = dpp_settle(cntf);      // (see seunet_common.h: nothing may narrow EXEC right behind a DPP fetch)
    cntf += dpp_settle(__builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, cntf), 0x401F)));   // xor 16
    const int j = col, rj = j & 15;
    // partial-sum type: f32 for bf16 activations (the stored tensor keeps 8 bits of mantissa anyway); f64 in the f32
    // parity mode, whose gradients are ill-conditioned enough to see an f32 reduction tree
    typedef typename std::conditional<sizeof(T) == 4, double, float>::type R;
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
      float v0[16];
      R val[32];   // val[0..15] sums, val[16..31] sums of squares
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // the shift: this channel's value at the half-wave's first lane, y-row 0 (any value of the channel will do)
        v0[r] = dpp_settle(__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane & 32) * 4, __builtin_bit_cast(int, acc[0][ns][r]))));
        R p1 = 0, p2 = 0;
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
          const R dv = vok[ms] ? (R)acc[ms][ns][r] - (R)v0[r] : (R)0;   // (exact in the f64 mode)
          p1 += dv;
          p2 += dv * dv;
        }
        val[r] = p1; val[16 + r] = p2;
      }
      {  // reduce-scatter over the 32 lanes of each half
        const bool b4 = (j & 16) != 0, b3 = (j & 8) != 0, b2 = (j & 4) != 0, b1 = (j & 2) != 0, b0 = (j & 1) != 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const R keep = b4 ? val[16 + i] : val[i], send = b4 ? val[i] : val[16 + i];
          val[i] = keep + xlane_swz16(send);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const R keep = b3 ? val[8 + i] : val[i], send = b3 ? val[i] : val[8 + i];
          val[i] = keep + xlane_dpp<0x128>(send);   // row_ror:8 == lane ^ 8
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const R keep = b2 ? val[4 + i] : val[i], send = b2 ? val[i] : val[4 + i];
          val[i] = keep + xlane_dpp<0x141>(send);   // row_half_mirror == lane ^ 7
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const R keep = b1 ? val[2 + i] : val[i], send = b1 ? val[i] : val[2 + i];
          val[i] = keep + xlane_dpp<0x4E>(send);    // quad_perm [2,3,0,1] == lane ^ 2
        }
        {
          const R keep = b0 ? val[1] : val[0], send = b0 ? val[0] : val[1];
          val[0] = keep + xlane_dpp<0xB1>(send);    // quad_perm [1,0,3,2] == lane ^ 1
        }
      }
      // lane j now owns value j: sum (j < 16) or sum of squares (j >= 16) of channel register rj
      float sh = v0[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) sh = (rj == r) ? v0[r] : sh;
      const R mine = dpp_settle(val[0]);
      const R s_of_q = xlane_swz16(mine);   // lane j ^ 16: the sum that belongs to this lane's sum of squares
      const double d0 = (double)sh, dc = (double)cntf;
      const double tot = j < 16 ? (double)mine + dc * d0
                                : (double)mine + 2.0 * d0 * (double)s_of_q + dc * d0 * d0;
      const int cl = ns * 32 + (rj & 3) + 8 * (rj >> 2) + 4 * h;
      red[(wave * NCOL + cl) * 2 + (j >> 4)] = tot;
    }
  }
  STAMP(6);   // bias + statistics arithmetic

  // (3) stores.  Two paths, chosen per launch:
  //  * narrow destinations (<= 16 channels, a.direct): straight from the accumulators.  A lane writes its runs of 4
  //    channels as 8-B / 16-B pieces; the h = 0 / 1 lanes and neighbouring voxels together cover contiguous memory
  //    (voxel pitch 16-32 B), so the stores coalesce and neither LDS nor a barrier is needed.  Destinations are reached
  //    through buffer descriptors: 32-bit offsets, voxels outside the volume (offset 2^32 - ...) dropped by the range check.
  //  * wide destinations: the same stores would be 8-B pieces one voxel pitch (>= 64 B) apart, nothing for the write
  //    coalescer to merge (measured: 240 cycles per store instruction on 32-channel outputs).  So each wave restages
  //    its 128 voxels through LDS -- cheaply, because the transposed accumulators already hold runs of 4 channels: one
  //    ds_write_b64 / b128 per run into [voxel][NCOL channels] rows (row pitch +16 B: conflict-free) -- and stores whole
  //    16-B pieces, 4-8 adjacent lanes covering one voxel's contiguous channels.
  if (a.direct) {
    const int daccmask = a.dacc0 | (a.dacc1 << 1) | (a.dacc2 << 2);
    const int dC0 = a.dstC0, dC1 = a.dstC1, dC2 = a.dstC2;
    const unsigned long long dp0 = reinterpret_cast<unsigned long long>(a.dst0), dp1 = reinterpret_cast<unsigned long long>(a.dst1),
                             dp2 = reinterpret_cast<unsigned long long>(a.dst2);
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co8 = ntile * NCOL + ns * 32 + 8 * q;
        if (co8 >= a.cout) continue;
        // which destination: by arithmetic, not select chains over the argument struct's fields -- those the compiler
        // turns into a scratch lookup table whose reload waits (s_waitcnt vmcnt(0)) for every store issued so far
        const int w1 = co8 >= a.dcum1 ? 1 : 0, w2 = co8 >= a.dcum2 ? 1 : 0;
        const int dC = dC0 + w1 * (dC1 - dC0) + w2 * (dC2 - dC1);
        const int cl = co8 - w1 * (a.dcum1 - w2 * a.dcum1) - w2 * a.dcum2;
        const int dacc = (daccmask >> (w1 + w2)) & 1;
        const unsigned long long dpu = dp0 + (unsigned long long)w1 * (dp1 - dp0) + (unsigned long long)w2 * (dp2 - dp1);
        void* dpv = reinterpret_cast<void*>(dpu);
        if (dpv == nullptr) continue;
        T* dbase = reinterpret_cast<T*>(dpv) + (long long)n * V * dC + cl;
        const long long davail = ((long long)V * dC - cl) * (long long)sizeof(T);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(uniform_ptr(dbase)), 0, __builtin_amdgcn_readfirstlane((int)davail), 0x00020000);
        const unsigned dstride = __builtin_amdgcn_readfirstlane((unsigned)(dC * (int)sizeof(T)));
        const unsigned hofs = (unsigned)(4 * h * (int)sizeof(T));
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
          const unsigned off = vok[ms] ? vlin[ms] * dstride + hofs : 0x80000000u;   // beyond any admitted sample: dropped
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[ms][ns][4 * q + i];
          if constexpr (sizeof(T) == 2) {
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            if (dacc) {
              const u32x2 o = __builtin_amdgcn_raw_buffer_load_b64(rd, off, 0, 0);
              v[0] += unpack_lo<T>(o.x); v[1] += unpack_hi<T>(o.x);
              v[2] += unpack_lo<T>(o.y); v[3] += unpack_hi<T>(o.y);
            }
            u32x2 u;
            u.x = pack2<T>(v[0], v[1]);
            u.y = pack2<T>(v[2], v[3]);
            __builtin_amdgcn_raw_buffer_store_b64(u, rd, off, 0, 0);
          } else {
            if (dacc) {
              const u32x4 o = __builtin_amdgcn_raw_buffer_load_b128(rd, off, 0, 0);
              v[0] += __uint_as_float(o.x); v[1] += __uint_as_float(o.y); v[2] += __uint_as_float(o.z); v[3] += __uint_as_float(o.w);
            }
            u32x4 u;
            u.x = __float_as_uint(v[0]); u.y = __float_as_uint(v[1]); u.z = __float_as_uint(v[2]); u.w = __float_as_uint(v[3]);
            __builtin_amdgcn_raw_buffer_store_b128(u, rd, off, 0, 0);
          }
        }
      }
    }
    STAMP(10);   // store issue
  } else {
    __syncthreads();   // every wave has left the K loop: its tiles are dead, the stages below reuse that memory
    STAMP(7);   // barrier after the K loop
    constexpr int ROWB = NCOL * (int)sizeof(T) + 16;          // stage row pitch in bytes
    constexpr int PIECES = NCOL * (int)sizeof(T) / 16;        // 16-B pieces per voxel
    constexpr int CPP = 16 / (int)sizeof(T);                  // channels per piece
    unsigned char* stage = smem + wave * (128 * ROWB);
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
      for (int ms = 0; ms < 4; ++ms)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned char* sp = stage + (ms * 32 + col) * ROWB + (ns * 32 + 8 * q + 4 * h) * (int)sizeof(T);
          if constexpr (sizeof(T) == 2) {
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            u32x2 u;
            u.x = pack2<T>(acc[ms][ns][4 * q], acc[ms][ns][4 * q + 1]);
            u.y = pack2<T>(acc[ms][ns][4 * q + 2], acc[ms][ns][4 * q + 3]);
            *reinterpret_cast<u32x2*>(sp) = u;
          } else {
            u32x4 u;
            u.x = __float_as_uint(acc[ms][ns][4 * q]); u.y = __float_as_uint(acc[ms][ns][4 * q + 1]);
            u.z = __float_as_uint(acc[ms][ns][4 * q + 2]); u.w = __float_as_uint(acc[ms][ns][4 * q + 3]);
            *reinterpret_cast<u32x4*>(sp) = u;
          }
        }
    __builtin_amdgcn_wave_barrier();   // LDS operations of one wave complete in order
    STAMP(9);   // stage writes
    {
      // the lane's piece (hence destination tensor and channel offset) is the same for every item: item % PIECES == lane % PIECES
      const int piece = lane % PIECES;
      const int co0 = ntile * NCOL + piece * CPP;
      T* sdst = nullptr; int sdC = 0, sdacc = 0;
      if (co0 < a.cout) {
        void* dpv = a.dst0; int cl = co0;
        sdC = a.dstC0; sdacc = a.dacc0;
        if (co0 >= a.dcum2) { dpv = a.dst2; sdC = a.dstC2; sdacc = a.dacc2; cl = co0 - a.dcum2; }
        else if (co0 >= a.dcum1) { dpv = a.dst1; sdC = a.dstC1; sdacc = a.dacc1; cl = co0 - a.dcum1; }
        if (dpv != nullptr) sdst = reinterpret_cast<T*>(dpv) + cl;
      }
#pragma unroll
      for (int i = 0; i < 2 * PIECES; ++i) {
        const int item = lane + 64 * i;
        const int vl = item / PIECES;                          // voxel of the wave's 4 x 32 block
        const int gy = STEP * (y0 + (vl >> 5)) + py, gx = STEP * (x0 + (vl & 31)) + px;
        if (sdst != nullptr && gz_w < a.D && gy < a.H && gx < a.W) {
          u32x4 u = *reinterpret_cast<const u32x4*>(stage + vl * ROWB + piece * 16);
          u32x4* qp = reinterpret_cast<u32x4*>(sdst + ((long long)n * V + ((long long)gz_w * a.H + gy) * a.W + gx) * sdC);
          if (sdacc) {
            const u32x4 o = *qp;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float lo = unpack_lo<T>(u[e]) + unpack_lo<T>(o[e]);
                const float hi = unpack_hi<T>(u[e]) + unpack_hi<T>(o[e]);
                u[e] = pack2<T>(lo, hi);
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) u[e] = __float_as_uint(__uint_as_float(u[e]) + __uint_as_float(o[e]));
            }
          }
          *qp = u;
        }
      }
    }
    STAMP(10);   // stage reads + store issue

  }
  // (4) cross-wave statistics: one barrier that waits for the LDS writes only (a __syncthreads() would also wait for
  //     the global stores above to be acknowledged), then a fixed-order sum of the four wave partials
  if (a.stats != nullptr) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid < NCOL * 2) {
      const int c = tid >> 1, k = tid & 1;
      const int co = ntile * NCOL + c;
      if (co < a.cout) {
        const double tot = ((red[(0 * NCOL + c) * 2 + k] + red[(1 * NCOL + c) * 2 + k]) +
                            red[(2 * NCOL + c) * 2 + k]) + red[(3 * NCOL + c) * 2 + k];
        a.stats[(((long long)n * tiles_per_sample + tile_slot) * a.cout + co) * 2 + k] = tot;
      }
    }
  }
#ifdef SEUNET_STAMP
  STAMP(11);   // transpose: stage reads + global stores
  if (a.debug != nullptr && lane == 0) {   // one 12-slot record per wave, no contention
    const size_t w = (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave;
    for (int i = 0; i < 12; ++i) a.debug[w * 12 + i] = ph[i];
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// weight packing: PyTorch (Cout, Cin, kD, kH, kW) f32  ->  per (n-tile, K-chunk) LDS images
// transpose_flip = 1 builds the data-gradient operator (roles of Cin/Cout swapped, taps mirrored)
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void conv_pack_body(const float* __restrict__ w, int taps, int cin_w, int cout_w, int tflip,
                                               T* __restrict__ out, int cin_e, int cout_e, int nchunks, int ncol,
                                               long long total) {
  constexpr int KC = Frag<T>::KC;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    int ci_in_chunk, col;
    if (sizeof(T) == 2) {  // [tap][h][col][j]
      const int j = (int)(r % 8); r /= 8;
      col = (int)(r % ncol); r /= ncol;
      const int hh = (int)(r % 2); r /= 2;
      ci_in_chunk = 8 * hh + j;
    } else {               // [tap][ks][h][col]
      col = (int)(r % ncol); r /= ncol;
      const int hh = (int)(r % 2); r /= 2;
      const int ks = (int)(r % 4); r /= 4;
      ci_in_chunk = 2 * ks + hh;
    }
    const int tap = (int)(r % taps); r /= taps;
    const int chunk = (int)(r % nchunks);
    const int ntile = (int)(r / nchunks);
    const int ci = chunk * KC + ci_in_chunk, co = ntile * ncol + col;
    float v = 0.f;
    if (ci < cin_e && co < cout_e) {
      if (!tflip) v = w[((long long)co * cin_w + ci) * taps + tap];
      else v = w[((long long)ci * cin_w + co) * taps + (taps - 1 - tap)];
    }
    out[i] = from_f32<T>(v);
  }
}
template <typename T>
__global__ void conv_pack_kernel(const float* __restrict__ w, int taps, int cin_w, int cout_w, int tflip,
                                 T* __restrict__ out, int cin_e, int cout_e, int nchunks, int ncol,
                                 long long total) {
  conv_pack_body<T>(w, taps, cin_w, cout_w, tflip, out, cin_e, cout_e, nchunks, ncol, total);
}
// all the weight tensors of a forward (or backward) pass in one launch: blockIdx.y = list entry
struct PackEntry { const float* w; void* out; int taps, cin_w, cout_w, tflip, cin_e, cout_e, nchunks, ncol; long long total; };
struct PackList { PackEntry e[24]; };
template <typename T>
__global__ void conv_pack_multi_kernel(PackList l) {
  const PackEntry& e = l.e[blockIdx.y];
  conv_pack_body<T>(e.w, e.taps, e.cin_w, e.cout_w, e.tflip, reinterpret_cast<T*>(e.out), e.cin_e, e.cout_e, e.nchunks, e.ncol, e.total);
}

// N columns per workgroup.  Round 1 chose 64 columns (two MFMA column blocks per A fragment, one workgroup per CU) for
// >= 64 -> > 32 channel layers (128 -> 64 ran 10 % faster that way).  Since the column blocks of one tile run back to back on
// one XCD (the second finds the input tile in that L2), two 32-column workgroups per CU win everywhere: dc3 forward 0.47 ->
// 0.45 ms, data gradient 0.52 -> 0.47 ms, the 128 -> 64 1x1x1 blocks 0.14 -> 0.12 ms, the 16^3-level layers 0.033 -> 0.022 ms
// (17.53 -> 17.20 ms per step).  SEUNET_CONV_NCOL=64 restores the old rule for A/B timing.
static inline int conv_ncol(int cin_e, int cout_e) {
  static const bool wide = [] { const char* e = getenv("SEUNET_CONV_NCOL"); return e && atoi(e) == 64; }();
  return (wide && cout_e > 32 && cin_e >= 64) ? 64 : 32;
}
unsigned long long* g_conv_debug = nullptr;   // set by seunet_debug_set_buffer (diagnostic builds)
static inline int conv_kc(int dtype) { return dtype_size(dtype) == 2 ? 16 : 8; }

size_t conv_wpack_bytes(int dtype, int taps, int cin, int cout) {
  const int ncol = conv_ncol(cin, cout), ntiles = cdiv(cout, ncol), nchunks = cdiv(cin, conv_kc(dtype));
  return (size_t)ntiles * nchunks * taps * ncol * 32;
}

int launch_conv_pack_weights(int dtype, const float* w, int taps, int cin_w, int cout_w, int tflip,
                             void* wpack, hipStream_t s) {
  SEUNET_CHECK(taps == 27 || taps == 1, "conv pack: taps=%d unsupported", taps);
  const int cin_e = tflip ? cout_w : cin_w, cout_e = tflip ? cin_w : cout_w;
  const int ncol = conv_ncol(cin_e, cout_e), nchunks = cdiv(cin_e, conv_kc(dtype));
  const long long total = (long long)(conv_wpack_bytes(dtype, taps, cin_e, cout_e) / dtype_size(dtype));
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  SEUNET_DTYPE_SWITCH(dtype, conv_pack_kernel<T><<<grid, 256, 0, s>>>(w, taps, cin_w, cout_w, tflip, (T*)wpack, cin_e, cout_e, nchunks, ncol, total));
  SEUNET_LAUNCH_CHECK();
  return 0;
}

int launch_conv_pack_weights_multi(int dtype, const ConvPackJob* jobs, int n, hipStream_t s) {
  for (int base = 0; base < n; base += 24) {
    PackList l{};
    const int m = n - base < 24 ? n - base : 24;
    for (int i = 0; i < m; ++i) {
      const ConvPackJob& j = jobs[base + i];
      SEUNET_CHECK(j.taps == 27 || j.taps == 1, "conv pack: taps=%d unsupported", j.taps);
      PackEntry& e = l.e[i];
      e.w = j.w; e.out = j.wpack; e.taps = j.taps; e.cin_w = j.cin_w; e.cout_w = j.cout_w; e.tflip = j.tflip;
      e.cin_e = j.tflip ? j.cout_w : j.cin_w; e.cout_e = j.tflip ? j.cin_w : j.cout_w;
      e.ncol = conv_ncol(e.cin_e, e.cout_e); e.nchunks = cdiv(e.cin_e, conv_kc(dtype));
      e.total = (long long)(conv_wpack_bytes(dtype, j.taps, e.cin_e, e.cout_e) / dtype_size(dtype));
    }
    SEUNET_DTYPE_SWITCH(dtype, conv_pack_multi_kernel<T><<<dim3(64, m), 256, 0, s>>>(l));
  }
  SEUNET_LAUNCH_CHECK();
  return 0;
}

// partial-stat slots per sample == workgroups per sample (tiles x parity classes of the dilation)
int conv_stats_tiles(Dims d, int taps, int dil) {
  const int st = taps == 27 ? dil : 1;
  return cdiv(cdiv(d.D, st), CV_TZ) * cdiv(cdiv(d.H, st), CV_TY) * cdiv(cdiv(d.W, st), CV_TX) * st * st * st;
}

template <typename T, int NSUB, int TAPS, int DIL>
static int launch_one(const ConvKArgs& a, dim3 grid, hipStream_t s) {
  constexpr int HALO = (TAPS == 27) ? 1 : 0;
  constexpr int NVH = (CV_TZ + 2 * HALO) * (CV_TY + 2 * HALO) * (CV_TX + 2 * HALO);
  constexpr int LDS_K = ((NVH * 2 + 255) / 256 + (TAPS * 64 * NSUB + 255) / 256) * 4096 + 128;   // K-loop tiles, padded to whole staging rounds
  constexpr int LDS_E = 4 * 128 * (32 * NSUB * (int)sizeof(T) + 16);                       // the four waves' store stages
  constexpr int LDS = (LDS_K > LDS_E ? LDS_K : LDS_E) + 4 * 32 * NSUB * 16;                // + the statistics partials
  static unsigned long long configured = 0;   // per instantiation: devices on which the LDS limit was raised
  if (int e = configure_kernel_lds(configured, reinterpret_cast<const void*>(&conv_igemm_kernel<T, NSUB, TAPS, DIL>), LDS)) return e;
  conv_igemm_kernel<T, NSUB, TAPS, DIL><<<grid, 256, LDS, s>>>(a);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int launch_t(int taps, int dil, int nsub, const ConvKArgs& a, dim3 grid, hipStream_t s) {
  if (taps == 1) return nsub == 1 ? launch_one<T, 1, 1, 1>(a, grid, s) : launch_one<T, 2, 1, 1>(a, grid, s);
  if (dil == 1) return nsub == 1 ? launch_one<T, 1, 27, 1>(a, grid, s) : launch_one<T, 2, 27, 1>(a, grid, s);
  return nsub == 1 ? launch_one<T, 1, 27, 2>(a, grid, s) : launch_one<T, 2, 27, 2>(a, grid, s);
}

static int check_lists(const SrcList& src, const DstList& dst) {
  SEUNET_CHECK(src.n >= 1 && src.n <= 3 && dst.n >= 1 && dst.n <= 3, "conv: 1..3 sources/destinations");
  for (int i = 0; i < src.n; ++i)
    SEUNET_CHECK(src.C[i] > 0 && src.C[i] % 8 == 0 && src.ptr[i], "conv: source %d needs C %% 8 == 0 (got %d)", i, src.C[i]);
  for (int i = 0; i < dst.n; ++i)
    SEUNET_CHECK(dst.C[i] > 0 && dst.C[i] % 8 == 0, "conv: destination %d needs C %% 8 == 0 (got %d)", i, dst.C[i]);
  return 0;
}

int launch_conv_igemm(int dtype, int taps, int dil, const SrcList& src, int cin_logical, const void* wpack,
                      const float* bias, const DstList& dst, double* stats, Dims d, hipStream_t s) {
  if (int e = check_lists(src, dst)) return e;
  SEUNET_CHECK(cin_logical >= 1 && cin_logical <= src.total(), "conv: cin=%d exceeds the source channels %d", cin_logical, src.total());
  SEUNET_CHECK(taps == 27 || taps == 1, "conv: taps=%d unsupported", taps);
  SEUNET_CHECK(taps == 1 || dil == 1 || dil == 2, "conv: dilation %d unsupported", dil);
  ConvKArgs a{};
  a.src0 = src.ptr[0]; a.srcC0 = src.C[0];
  a.src1 = src.n > 1 ? src.ptr[1] : nullptr; a.srcC1 = src.n > 1 ? src.C[1] : 0;
  a.src2 = src.n > 2 ? src.ptr[2] : nullptr; a.srcC2 = src.n > 2 ? src.C[2] : 0;
  a.cin = cin_logical;
  a.cum1 = src.n > 1 ? src.C[0] : src.total();
  a.cum2 = src.n > 2 ? src.C[0] + src.C[1] : src.total();
  a.wpack = wpack; a.bias = bias;
  a.dst0 = dst.ptr[0]; a.dstC0 = dst.C[0]; a.dacc0 = dst.acc[0];
  a.dst1 = dst.n > 1 ? dst.ptr[1] : nullptr; a.dstC1 = dst.n > 1 ? dst.C[1] : 0; a.dacc1 = dst.n > 1 ? dst.acc[1] : 0;
  a.dst2 = dst.n > 2 ? dst.ptr[2] : nullptr; a.dstC2 = dst.n > 2 ? dst.C[2] : 0; a.dacc2 = dst.n > 2 ? dst.acc[2] : 0;
  a.cout = dst.total();
  a.dcum1 = dst.n > 1 ? dst.C[0] : a.cout;
  a.dcum2 = dst.n > 2 ? dst.C[0] + dst.C[1] : a.cout;
  a.stats = stats;
  a.N = d.N; a.D = d.D; a.H = d.H; a.W = d.W;
  const int st = taps == 27 ? dil : 1;
  a.tx = cdiv(cdiv(d.W, st), CV_TX); a.ty = cdiv(cdiv(d.H, st), CV_TY); a.tz = cdiv(cdiv(d.D, st), CV_TZ);
  a.nchunks = cdiv(a.cin, conv_kc(dtype));
  {  // lane-pair staging needs the concatenation boundaries on chunk boundaries (always true for f32: chunk = 8 channels)
    const int kc = conv_kc(dtype);
    a.pair = (a.cum1 % kc == 0 || a.cum1 >= a.cin) && (a.cum2 % kc == 0 || a.cum2 >= a.cin) ? 1 : 0;
  }
  a.direct = 1;   // voxel pitch of every destination <= 32 B (dilation 2 writes every other voxel: pitch doubles)
  for (int i = 0; i < dst.n; ++i) if (dst.C[i] * st > 16) a.direct = 0;
  a.debug = g_conv_debug;
  const int ncol = conv_ncol(a.cin, a.cout);
  a.ncb = cdiv(a.cout, ncol);
  dim3 grid(a.tx * a.ty * a.tz * st * st * st * a.ncb, 1, d.N);
  SEUNET_CHECK(d.N <= 65535, "conv: batch too large");
  // the staging loads use 32-bit byte offsets inside one sample of one source tensor
  for (int i = 0; i < src.n; ++i)
    SEUNET_CHECK((long long)d.vox() * src.C[i] * (long long)dtype_size(dtype) < (1LL << 31),
                 "conv: one sample of source %d is %lld bytes; the MFMA path addresses < 2^31 bytes per sample "
                 "(tile the volume, e.g. sliding_window_predict)", i, (long long)d.vox() * src.C[i] * (long long)dtype_size(dtype));
  for (int i = 0; i < dst.n; ++i)
    SEUNET_CHECK((long long)d.vox() * dst.C[i] * (long long)dtype_size(dtype) < (1LL << 31),
                 "conv: one sample of destination %d is %lld bytes; the MFMA path addresses < 2^31 bytes per sample",
                 i, (long long)d.vox() * dst.C[i] * (long long)dtype_size(dtype));
  SEUNET_DTYPE_SWITCH(dtype, return (launch_t<T>(taps, dil, ncol / 32, a, grid, s)));
  return 1;
}

// ------------------------------------------------------------------------------------------------
// naive direct convolution (one thread per output element).  Device-side cross-check for the MFMA
// kernel and selectable with SEUNET_CONV_IMPL=naive; never a CPU fallback.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void conv_naive_kernel(ConvKArgs a, const float* __restrict__ w, int taps, int dil, int tflip,
                                  int cin_w, long long total) {
  const long long V = (long long)a.D * a.H * a.W;
  const int t3 = taps == 27 ? 3 : 1, c = taps == 27 ? 1 : 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(i % a.cout);
    long long r = i / a.cout;
    const int x = (int)(r % a.W); r /= a.W;
    const int y = (int)(r % a.H); r /= a.H;
    const int z = (int)(r % a.D);
    const long long n = r / a.D;
    float acc = a.bias ? a.bias[co] : 0.f;
    for (int tap = 0; tap < taps; ++tap) {
      const int dz = (tap / (t3 * t3) - c) * dil, dy = ((tap / t3) % t3 - c) * dil, dx = (tap % t3 - c) * dil;
      const int zz = z + dz, yy = y + dy, xx = x + dx;
      if ((unsigned)zz >= (unsigned)a.D || (unsigned)yy >= (unsigned)a.H || (unsigned)xx >= (unsigned)a.W) continue;
      const long long vv = n * V + ((long long)zz * a.H + yy) * a.W + xx;
      for (int ci = 0; ci < a.cin; ++ci) {
        const void* sp = a.src0; int sC = a.srcC0, cc = ci;
        if (ci >= a.cum2) { sp = a.src2; sC = a.srcC2; cc = ci - a.cum2; }
        else if (ci >= a.cum1) { sp = a.src1; sC = a.srcC1; cc = ci - a.cum1; }
        const float xv = to_f32(reinterpret_cast<const T*>(sp)[vv * sC + cc]);
        const float wv = tflip ? w[((long long)ci * cin_w + co) * taps + (taps - 1 - tap)]
                               : w[((long long)co * cin_w + ci) * taps + tap];
        acc += xv * wv;
      }
    }
    void* dpv = a.dst0; int dC = a.dstC0, dacc = a.dacc0, cl = co;
    if (co >= a.dcum2) { dpv = a.dst2; dC = a.dstC2; dacc = a.dacc2; cl = co - a.dcum2; }
    else if (co >= a.dcum1) { dpv = a.dst1; dC = a.dstC1; dacc = a.dacc1; cl = co - a.dcum1; }
    if (dpv == nullptr) continue;
    T* q = reinterpret_cast<T*>(dpv) + (n * V + ((long long)z * a.H + y) * a.W + x) * dC + cl;
    if (dacc) acc += to_f32(*q);
    *q = from_f32<T>(acc);
  }
}

// Weights are in the PyTorch layout.  cin_logical = number of (leading) input channels that carry weights;
// tensors may be zero-padded beyond it (the packed network input).
int launch_conv_naive(int dtype, int taps, int dil, const SrcList& src, int cin_logical, const float* w,
                      int tflip, const float* bias, const DstList& dst, Dims d, hipStream_t s) {
  if (int e = check_lists(src, dst)) return e;
  SEUNET_CHECK(cin_logical >= 1 && cin_logical <= src.total(), "conv: cin=%d exceeds the source channels %d", cin_logical, src.total());
  ConvKArgs a{};
  a.src0 = src.ptr[0]; a.srcC0 = src.C[0];
  a.src1 = src.n > 1 ? src.ptr[1] : nullptr; a.srcC1 = src.n > 1 ? src.C[1] : 0;
  a.src2 = src.n > 2 ? src.ptr[2] : nullptr; a.srcC2 = src.n > 2 ? src.C[2] : 0;
  a.cin = cin_logical;
  a.cum1 = src.n > 1 ? src.C[0] : src.total();
  a.cum2 = src.n > 2 ? src.C[0] + src.C[1] : src.total();
  a.bias = bias;
  a.dst0 = dst.ptr[0]; a.dstC0 = dst.C[0]; a.dacc0 = dst.acc[0];
  a.dst1 = dst.n > 1 ? dst.ptr[1] : nullptr; a.dstC1 = dst.n > 1 ? dst.C[1] : 0; a.dacc1 = dst.n > 1 ? dst.acc[1] : 0;
  a.dst2 = dst.n > 2 ? dst.ptr[2] : nullptr; a.dstC2 = dst.n > 2 ? dst.C[2] : 0; a.dacc2 = dst.n > 2 ? dst.acc[2] : 0;
  a.cout = dst.total();
  a.dcum1 = dst.n > 1 ? dst.C[0] : a.cout;
  a.dcum2 = dst.n > 2 ? dst.C[0] + dst.C[1] : a.cout;
  a.N = d.N; a.D = d.D; a.H = d.H; a.W = d.W;
  const long long total = (long long)d.N * d.vox() * a.cout;
  const int grid = (int)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256);
  const int cin_w = tflip ? a.cout : cin_logical;  // the PyTorch weight's Cin extent
  SEUNET_DTYPE_SWITCH(dtype, conv_naive_kernel<T><<<grid, 256, 0, s>>>(a, w, taps, dil, tflip, cin_w, total));
  SEUNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace seunet
