// Implicit-GEMM convolution for gfx950 matrix cores: 3x3x3 (dilation 1 or 2, "same" zero padding) and
// 1x1x1, forward and data-gradient (a data-gradient is the same kernel run on flipped/transposed
// weights).  Reference ops: nn.Conv3d at SE_UNet.py:15,42,57 (+ torch.cat at :186,195,204,212,216,218,
// 222,224,228 which is fused here as a multi-pointer channel concatenation).
//
//   GEMM view     M = voxels, N = output channels, K = taps x input channels
//   workgroup     256 threads = 4 waves; output tile 4(z) x 4(y) x 32(x) voxels; wave w owns z-slice w,
//                 i.e. four 32-voxel x-rows, times all N columns of the tile (32 or 64)
//   dilation 2    decomposes into 8 independent dilation-1 problems on the parity sub-lattices
//                 (voxel = 2*lattice + parity): the tile lives on one sub-lattice, so the halo is 1 lattice
//                 voxel (39 KB tile) instead of 2 voxels (74 KB), and the rest of the kernel is unchanged
//   MFMA          bf16: v_mfma_f32_32x32x16_bf16 (K-step = 16 channels of one tap)
//                 f32 : v_mfma_f32_32x32x2_f32   (exact f32 FMA chain; the 1e-3 parity mode)
//   LDS           input halo tile [6*6*34 voxels][32 B = one K-chunk], 16-B slots XOR-swizzled by voxel bit 3
//                 so a ds_read_b128 of 16 consecutive voxels is conflict-free; weight slab
//                 [tap][k-half][column][8 x bf16] (one contiguous 16-B fragment per lane)
//   K loop        chunks of 32 bytes of channels (16 bf16 / 8 f32).  The NEXT chunk's tile and weights are
//                 fetched into registers while the current chunk's 27 taps of MFMAs run, and written to LDS
//                 after the barrier (split issue-early / write-late staging).  The fetches are
//                 buffer_load_dwordx4 through wave-uniform descriptors: 32-bit offsets, and the hardware
//                 range check returns zeros for padding voxels / channels, so there is no branch and no
//                 64-bit address arithmetic in the loop (each wave fetches one 16-B piece index of 64
//                 consecutive halo voxels per instruction)
//   epilogue      + bias; per-(n,c) InstanceNorm partial sums (shifted f32 sums per register row, f64 across rows,
//                 fixed-order cross-wave sum); each wave transposes its own 128 voxels through a private LDS
//                 stage (no workgroup barrier) so that every lane stores 16 B (8 channels of one voxel),
//                 optionally += (gradient accumulation) and split over up to three destination tensors
//                 (backward of the fused concatenation)
//   grid          blockIdx.x is remapped so that each XCD (private L2) owns a contiguous run of tiles
#include "seunet_common.h"
#include <utility>

namespace seunet {

typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
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
  unsigned long long* debug;   // diagnostic builds only (-DSEUNET_STAMP): per-phase cycle sums
};

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { static constexpr int KC = 16, KSTEPS = 1; };
template <> struct Frag<float> { static constexpr int KC = 8, KSTEPS = 4; };

__device__ __forceinline__ void store_vec8(float* q, const float (&v)[8], int acc) {
  if (acc) { float o[8]; load8(q, o);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] += v[j];
    store8(q, o);
  } else store8(q, v);
}
__device__ __forceinline__ void store_vec8(bf16_t* q, const float (&v)[8], int acc) {
  if (acc) { float o[8]; load8(q, o);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] += v[j];
    store8(q, o);
  } else store8(q, v);
}

#ifdef SEUNET_STAMP
#define STAMP(i) do { const unsigned long long _t = __builtin_readcyclecounter(); ph[i] += _t - t_last; t_last = _t; } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

template <typename T, int NSUB, int TAPS, int DIL>
__global__ void __launch_bounds__(256)
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
  unsigned char* w_tile = smem + IN_ITEMS * 4096;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  // XCD-aware order: blocks b, b+8, b+16.. share an XCD; give each XCD a contiguous run of tiles
  int t;
  {
    const int nt = gridDim.x, b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int tile_slot = t;
  const int bx = t % a.tx; t /= a.tx;
  const int by = t % a.ty; t /= a.ty;
  const int bz = t % a.tz;
  const int par = t / a.tz;                                  // parity class 0..STEP^3-1
  const int px = par % STEP, py = (par / STEP) % STEP, pz = par / (STEP * STEP);
  const int x0 = bx * CV_TX, y0 = by * CV_TY, z0 = bz * CV_TZ;   // lattice coordinates
  const int ntile = blockIdx.y, n = blockIdx.z;
  const long long V = (long long)a.D * a.H * a.W;

  // ---- per-thread staging plan (chunk independent) ----
  // staging slot L = tid + 256*k: wave-instruction (wave, k) moves piece (wave & 1) of the 64 consecutive halo
  // voxels starting at (2k + (wave >> 1)) * 64, so the piece -- hence the source tensor -- is wave-uniform.
  const int piece = wave & 1;
  constexpr unsigned INVALID = 0xFFFFFFFFu;
  unsigned vofs[IN_ITEMS];
#pragma unroll
  for (int k = 0; k < IN_ITEMS; ++k) {
    const int vox = (2 * k + (wave >> 1)) * 64 + lane;
    const int hx = vox % HX;
    const int r2 = vox / HX;
    const int hy = r2 % HY, hz = r2 / HY;
    const int lz = z0 - HALO + hz, ly = y0 - HALO + hy, lx = x0 - HALO + hx;
    const int gz = STEP * lz + pz, gy = STEP * ly + py, gx = STEP * lx + px;
    const bool ok = vox < NVH && lz >= 0 && ly >= 0 && lx >= 0 && gz < a.D && gy < a.H && gx < a.W;
    vofs[k] = ok ? (unsigned)((gz * a.H + gy) * a.W + gx) : INVALID;
  }
  STAMP(8);   // index plan
  const int lds_in0 = lane * 32 + (wave >> 1) * 2048 + 16 * (piece ^ ((lane >> 3) & 1));   // + k * 4096
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
  auto fetch_setup = [&](int chunk) __attribute__((always_inline)) {
    const int ch0 = chunk * KC + piece * (KC / 2);
    const void* sp = a.src0; int sC = a.srcC0, c = ch0;
    if (ch0 >= a.cum2) { sp = a.src2; sC = a.srcC2; c = ch0 - a.cum2; }
    else if (ch0 >= a.cum1) { sp = a.src1; sC = a.srcC1; c = ch0 - a.cum1; }
    const T* base = reinterpret_cast<const T*>(sp) + (long long)n * V * sC + c;
    const long long avail = ch0 < a.cin ? ((long long)V * sC - c) * (long long)sizeof(T) : 0;   // 0 records: all zeros
    rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_ptr(base)), 0,
                                              __builtin_amdgcn_readfirstlane((int)avail), 0x00020000);
    in_stride = __builtin_amdgcn_readfirstlane((unsigned)(sC * (int)sizeof(T)));
    rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_ptr(wbase + (size_t)chunk * (W_TOTAL * 16))), 0,
                                             W_TOTAL * 16, 0x00020000);
  };
  auto fetch_item = [&](auto item_c) __attribute__((always_inline)) {   // item: 0..IN_ITEMS-1 input pieces, then W_ITEMS weight pieces
    constexpr int item = decltype(item_c)::value;
    if constexpr (item < IN_ITEMS) {
      // byte offset = voxel index x stride, formed at the point of issue from a fresh copy of the stride (the empty asm
      // keeps the compiler from hoisting all IN_ITEMS products out of the tap loop into 10 more live registers).
      // Padding voxels carry index 0xFFFFFFFF: the product wraps to 2^32 - stride, beyond any tensor the 32-bit range
      // check admits (launch_conv_igemm rejects >= 2^31-byte samples), so the hardware returns zeros.
      unsigned st = in_stride;
      asm volatile("" : "+s"(st));
      rin[item] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, vofs[item] * st, 0, 0);
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
    for (int k = 0; k < IN_ITEMS; ++k) *reinterpret_cast<u32x4*>(in_tile + lds_in0 + k * 4096) = rin[k];
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

    // ---- MFMA over taps ----
    constexpr int ZUNROLL = NSUB == 2 ? T3 : 1;
    // 32 columns: the z loop stays rolled (fully unrolled the scheduler hoists LDS reads and the kernel no longer fits
    // two workgroups per CU); 64 columns run one workgroup per CU anyway and compile without scratch only unrolled
#pragma unroll ZUNROLL
    for (int tz3 = 0; tz3 < T3; ++tz3) {
     [&]<int... TYX>(std::integer_sequence<int, TYX...>) __attribute__((always_inline)) {
      ([&]() __attribute__((always_inline)) {
          constexpr int ty3 = TYX / T3, tx3 = TYX % T3;
          const int tap = tz3 * T3 * T3 + TYX;
          if (INTERLEAVE && has_next) {   // wave-uniform: the fetch items of this tap, with compile-time register indices
            if (tz3 == 0) fetch_range(std::integral_constant<int, TYX * F_PER_TAP>{});
            else if (tz3 == 1) fetch_range(std::integral_constant<int, (T3 * T3 + TYX) * F_PER_TAP>{});
            else fetch_range(std::integral_constant<int, (2 * T3 * T3 + TYX) * F_PER_TAP>{});
          }
          const int vbase = ((wave + tz3 * HALO) * HY + ty3 * HALO) * HX + tx3 * HALO + col;
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) {
            if constexpr (sizeof(T) == 2) {
              bf16x8 bfr[NSUB];
#pragma unroll
              for (int ns = 0; ns < NSUB; ++ns)
                bfr[ns] = *reinterpret_cast<const bf16x8*>(w_tile + ((tap * 2 + h) * NCOL + ns * 32 + col) * 16);
#pragma unroll
              for (int ms = 0; ms < 4; ++ms) {
                const int vox = vbase + ms * HX;
                const bf16x8 afr =
                    *reinterpret_cast<const bf16x8*>(in_tile + vox * 32 + 16 * (h ^ ((vox >> 3) & 1)));
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns)
                  acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr[ns], acc[ms][ns], 0, 0, 0);
              }
            } else {
              float bfr[NSUB];
#pragma unroll
              for (int ns = 0; ns < NSUB; ++ns)
                bfr[ns] = *reinterpret_cast<const float*>(w_tile + (((tap * 4 + ks) * 2 + h) * NCOL + ns * 32 + col) * 4);
#pragma unroll
              for (int ms = 0; ms < 4; ++ms) {
                const int vox = vbase + ms * HX;
                const float afr = *reinterpret_cast<const float*>(
                    in_tile + vox * 32 + 16 * ((ks >> 1) ^ ((vox >> 3) & 1)) + 4 * (2 * (ks & 1) + h));
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns)
                  acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr, bfr[ns], acc[ms][ns], 0, 0, 0);
              }
            }
          }
      }(), ...);
     }(std::make_integer_sequence<int, T3 * T3>{});
    }
    STAMP(5);   // MFMA block of this chunk
  }

  // ---- epilogue ----
  // (1) bias + f64 InstanceNorm partial sums from the accumulators
  const int gz_w = STEP * (z0 + wave) + pz;
  double s1[NSUB], s2[NSUB];
#pragma unroll
  for (int ns = 0; ns < NSUB; ++ns) {
    s1[ns] = 0.0; s2[ns] = 0.0;
    const int co = ntile * NCOL + ns * 32 + col;
    const bool cvalid = co < a.cout;
    const float bias = (a.bias != nullptr && cvalid) ? a.bias[co] : 0.f;
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
      const int gy = STEP * (y0 + ms) + py;
      const bool rowok = a.stats != nullptr && cvalid && gz_w < a.D && gy < a.H;
      // shifted sums: deviations from the row's first value are summed in f32 (no cancellation: they are of the
      // order of the standard deviation), the shift is undone in f64
      const float v0 = acc[ms][ns][0] + bias;
      float p1 = 0.f, p2 = 0.f, cnt = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gx = STEP * (x0 + (r & 3) + 8 * (r >> 2) + 4 * h) + px;
        const float val = acc[ms][ns][r] + bias;
        acc[ms][ns][r] = val;
        const bool ok = rowok && gx < a.W;
        const float dv = ok ? val - v0 : 0.f;
        p1 += dv;
        p2 += dv * dv;
        cnt += ok ? 1.f : 0.f;
      }
      const double d0 = (double)v0, dp1 = (double)p1, dc = (double)cnt;
      s1[ns] += dp1 + dc * d0;
      s2[ns] += (double)p2 + 2.0 * d0 * dp1 + dc * d0 * d0;
    }
  }
  STAMP(6);   // bias + statistics arithmetic
  __syncthreads();   // all waves are done with the K-loop tiles; LDS is reused below
  STAMP(7);   // barrier after the K loop
  // (2) statistics: per-wave partials through LDS, fixed-order sum.  This happens BEFORE the stores: a
  //     __syncthreads() also waits for vmcnt(0), i.e. a barrier after the stores would wait for HBM write latency.
  constexpr int STG = 64 * NCOL * 4;                       // bytes of one wave's transpose stage (64 voxels x NCOL f32)
  double* red = reinterpret_cast<double*>(smem + 4 * STG);  // [4 waves][NCOL][2]
  if (a.stats != nullptr) {
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
      const double u = s1[ns] + __shfl_xor(s1[ns], 32, 64);
      const double v = s2[ns] + __shfl_xor(s2[ns], 32, 64);
      if (h == 0) {
        red[(wave * NCOL + ns * 32 + col) * 2] = u;
        red[(wave * NCOL + ns * 32 + col) * 2 + 1] = v;
      }
    }
    __syncthreads();
    if (tid < NCOL * 2) {
      const int c = tid >> 1, k = tid & 1;
      const int co = ntile * NCOL + c;
      if (co < a.cout) {
        const double tot = ((red[(0 * NCOL + c) * 2 + k] + red[(1 * NCOL + c) * 2 + k]) +
                            red[(2 * NCOL + c) * 2 + k]) + red[(3 * NCOL + c) * 2 + k];
        a.stats[(((long long)n * gridDim.x + tile_slot) * a.cout + co) * 2 + k] = tot;
      }
    }
  }
  STAMP(9);   // cross-wave statistics
  // (3) wave-private transpose (no workgroup barrier): two passes of 2 x-rows (64 voxels) through this wave's own
  //     LDS stage, then every lane stores 8 channels (16/32 B) of one voxel
  // store-phase destination of this lane (its 8-channel group is the same for every item): selected once
  // (after the K loop, so it costs no registers there) so that no select chain / lookup table -- which the compiler
  // would place in scratch and reload behind s_waitcnt vmcnt(0) -- sits between the global stores
  constexpr int GRP = NCOL / 8;
  T* sdst = nullptr; int sdC = 0, sdacc = 0;
  {
    const int co0 = ntile * NCOL + (lane % GRP) * 8;
    if (co0 < a.cout) {
      void* dpv = a.dst0; int cl = co0;
      sdC = a.dstC0; sdacc = a.dacc0;
      if (co0 >= a.dcum2) { dpv = a.dst2; sdC = a.dstC2; sdacc = a.dacc2; cl = co0 - a.dcum2; }
      else if (co0 >= a.dcum1) { dpv = a.dst1; sdC = a.dstC1; sdacc = a.dacc1; cl = co0 - a.dcum1; }
      if (dpv != nullptr) sdst = reinterpret_cast<T*>(dpv) + cl;
    }
  }

  float* stage = reinterpret_cast<float*>(smem + wave * STG);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int xl = (r & 3) + 8 * (r >> 2) + 4 * h;
          stage[(m2 * 32 + xl) * NCOL + ns * 32 + col] = acc[pass * 2 + m2][ns][r];
        }
    __builtin_amdgcn_wave_barrier();   // LDS operations of one wave complete in order
    STAMP(10);   // transpose: stage writes
#pragma unroll
    for (int i = 0; i < GRP; ++i) {
      const int item = lane + 64 * i;            // 64 voxels x GRP groups; item % GRP == lane % GRP
      const int vl = item / GRP, grp = item % GRP;
      const int gy = STEP * (y0 + pass * 2 + (vl >> 5)) + py, gx = STEP * (x0 + (vl & 31)) + px;
      if (sdst != nullptr && gz_w < a.D && gy < a.H && gx < a.W) {
        float v[8];
        load8(stage + vl * NCOL + grp * 8, v);
        T* q = sdst + ((long long)n * V + ((long long)gz_w * a.H + gy) * a.W + gx) * sdC;
        store_vec8(q, v, sdacc);
      }
    }
    __builtin_amdgcn_wave_barrier();
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
__global__ void conv_pack_kernel(const float* __restrict__ w, int taps, int cin_w, int cout_w, int tflip,
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

// N columns per workgroup.  64 columns (two MFMA column blocks per A fragment, one workgroup per CU) only pay when
// the K loop is long; measured on MI355X (scripts/bench_conv.py): 32->64 channel convs run 20-25 % faster as two
// 32-column workgroups per CU, 128->64 runs 10 % faster with 64 columns.
static inline int conv_ncol(int cin_e, int cout_e) { return (cout_e > 32 && cin_e >= 64) ? 64 : 32; }
unsigned long long* g_conv_debug = nullptr;   // set by seunet_debug_set_buffer (diagnostic builds)
static inline int conv_kc(int dtype) { return dtype == SEUNET_BF16 ? 16 : 8; }

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
  if (dtype == SEUNET_BF16)
    conv_pack_kernel<bf16_t><<<grid, 256, 0, s>>>(w, taps, cin_w, cout_w, tflip, (bf16_t*)wpack, cin_e, cout_e, nchunks, ncol, total);
  else
    conv_pack_kernel<float><<<grid, 256, 0, s>>>(w, taps, cin_w, cout_w, tflip, (float*)wpack, cin_e, cout_e, nchunks, ncol, total);
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
  constexpr int LDS_K = ((NVH * 2 + 255) / 256 + (TAPS * 64 * NSUB + 255) / 256) * 4096;   // K-loop tiles, padded to whole staging rounds
  constexpr int LDS_E = 4 * 64 * 32 * NSUB * 4 + 4 * 32 * NSUB * 16;   // 4 wave-private transpose stages + stats partials
  constexpr int LDS = LDS_K > LDS_E ? LDS_K : LDS_E;
  static bool configured = false;  // per instantiation
  if (!configured) {
    SEUNET_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, NSUB, TAPS, DIL>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    configured = true;
  }
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
  a.debug = g_conv_debug;
  const int ncol = conv_ncol(a.cin, a.cout);
  dim3 grid(a.tx * a.ty * a.tz * st * st * st, cdiv(a.cout, ncol), d.N);
  SEUNET_CHECK(d.N <= 65535, "conv: batch too large");
  // the staging loads use 32-bit byte offsets inside one sample of one source tensor
  for (int i = 0; i < src.n; ++i)
    SEUNET_CHECK((long long)d.vox() * src.C[i] * (long long)dtype_size(dtype) < (1LL << 31),
                 "conv: one sample of source %d is %lld bytes; the MFMA path addresses < 2^31 bytes per sample "
                 "(tile the volume, e.g. sliding_window_predict)", i, (long long)d.vox() * src.C[i] * (long long)dtype_size(dtype));
  if (dtype == SEUNET_BF16) return launch_t<bf16_t>(taps, dil, ncol / 32, a, grid, s);
  return launch_t<float>(taps, dil, ncol / 32, a, grid, s);
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
  if (dtype == SEUNET_BF16)
    conv_naive_kernel<bf16_t><<<grid, 256, 0, s>>>(a, w, taps, dil, tflip, cin_w, total);
  else
    conv_naive_kernel<float><<<grid, 256, 0, s>>>(a, w, taps, dil, tflip, cin_w, total);
  SEUNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace seunet
