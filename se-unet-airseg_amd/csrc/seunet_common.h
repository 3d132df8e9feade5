// Shared declarations for the SE-UNet gfx950 library (internal; the public C ABI is
// include/seunet_hip.h).  Activations are channels-last  [N][D][H][W][C]  with C a
// multiple of 8, stored as f32 or bf16; all reductions and accumulators are f32/f64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#define SEUNET_F32 0
#define SEUNET_BF16 1
#define SEUNET_F16 2   /* IEEE half storage (BASELINE configs[4]); same MFMA rate and byte size as bf16 */

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

namespace seunet {

// ---- error plumbing (never throws across the ABI) -------------------------------
void set_error(const std::string& msg);
const char* get_error();
int fail(const char* fmt, ...);
bool prof_on();
void prof_mark(const char* tag, hipStream_t s);   // no-op unless seunet_prof_enable(1)

#define SEUNET_CHECK(cond, ...)                       \
  do {                                                \
    if (!(cond)) return ::seunet::fail(__VA_ARGS__);  \
  } while (0)

#define SEUNET_HIP(expr)                                                              \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess)                                                             \
      return ::seunet::fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                            __FILE__, __LINE__);                                      \
  } while (0)

#define SEUNET_LAUNCH_CHECK()                                                          \
  do {                                                                                \
    hipError_t _e = hipGetLastError();                                                \
    if (_e != hipSuccess)                                                             \
      return ::seunet::fail("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), \
                            __FILE__, __LINE__);                                      \
  } while (0)

static inline size_t dtype_size(int dtype) { return (dtype == SEUNET_BF16 || dtype == SEUNET_F16) ? 2 : 4; }
static inline bool dtype_ok(int dtype) { return dtype == SEUNET_F32 || dtype == SEUNET_BF16 || dtype == SEUNET_F16; }
// run `...` with T = float | bf16_t | f16_t
#define SEUNET_DTYPE_SWITCH(DT, ...)                                   \
  do {                                                                 \
    if ((DT) == SEUNET_BF16) { typedef bf16_t T; __VA_ARGS__; }        \
    else if ((DT) == SEUNET_F16) { typedef f16_t T; __VA_ARGS__; }     \
    else { typedef float T; __VA_ARGS__; }                             \
  } while (0)
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- small POD descriptors shared by launchers and the net executor ---------------
struct Dims {
  int N, D, H, W;
  long long vox() const { return (long long)D * H * W; }
};

struct SrcList {           // virtual channel concatenation of up to 3 tensors
  const void* ptr[3];
  int C[3];
  int n;
  int total() const { int t = 0; for (int i = 0; i < n; ++i) t += C[i]; return t; }
};

struct DstList {           // channel split of an output over up to 3 tensors
  void* ptr[3];            // may be null: that channel range is computed but dropped
  int C[3];
  int acc[3];              // 1: read-modify-write (+=), 0: overwrite
  int n;
  int total() const { int t = 0; for (int i = 0; i < n; ++i) t += C[i]; return t; }
};

// ---- device helpers ----------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ float bf16_bits_to_f32(unsigned int hi16) {
  return __uint_as_float(hi16 << 16);
}
__device__ __forceinline__ unsigned int f32_to_bf16_bits(float f) {
  bf16_t b = (bf16_t)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return (unsigned int)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }   // v_cvt_f16_f32: RNE, overflow -> inf
// two 16-bit elements <-> one dword, for either 16-bit storage type
template <typename T> __device__ __forceinline__ unsigned int pack2(float lo, float hi);
// (one v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 with two sources: RNE, NaN stays NaN; the scalar casts compiled to two
// conversions, a shift and an or per dword)
typedef float seunet_f32x2 __attribute__((ext_vector_type(2)));
typedef bf16_t seunet_bf16x2 __attribute__((ext_vector_type(2)));
typedef f16_t seunet_f16x2 __attribute__((ext_vector_type(2)));
template <> __device__ __forceinline__ unsigned int pack2<bf16_t>(float lo, float hi) {
  const seunet_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, seunet_bf16x2));
}
template <> __device__ __forceinline__ unsigned int pack2<f16_t>(float lo, float hi) {
  const seunet_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, seunet_f16x2));
}
template <typename T> __device__ __forceinline__ float unpack_lo(unsigned int u);
template <typename T> __device__ __forceinline__ float unpack_hi(unsigned int u);
template <> __device__ __forceinline__ float unpack_lo<bf16_t>(unsigned int u) { return bf16_bits_to_f32(u & 0xffffu); }
template <> __device__ __forceinline__ float unpack_hi<bf16_t>(unsigned int u) { return bf16_bits_to_f32(u >> 16); }
template <> __device__ __forceinline__ float unpack_lo<f16_t>(unsigned int u) { return (float)__builtin_bit_cast(f16_t, (unsigned short)(u & 0xffffu)); }
template <> __device__ __forceinline__ float unpack_hi<f16_t>(unsigned int u) { return (float)__builtin_bit_cast(f16_t, (unsigned short)(u >> 16)); }

// 8 consecutive channels <-> 8 floats
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const float4 a = reinterpret_cast<const float4*>(p)[0];
  const float4 b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  v[0] = bf16_bits_to_f32(u.x & 0xffffu); v[1] = bf16_bits_to_f32(u.x >> 16);
  v[2] = bf16_bits_to_f32(u.y & 0xffffu); v[3] = bf16_bits_to_f32(u.y >> 16);
  v[4] = bf16_bits_to_f32(u.z & 0xffffu); v[5] = bf16_bits_to_f32(u.z >> 16);
  v[6] = bf16_bits_to_f32(u.w & 0xffffu); v[7] = bf16_bits_to_f32(u.w >> 16);
}
__device__ __forceinline__ void load8(const f16_t* p, float (&v)[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  v[0] = unpack_lo<f16_t>(u.x); v[1] = unpack_hi<f16_t>(u.x); v[2] = unpack_lo<f16_t>(u.y); v[3] = unpack_hi<f16_t>(u.y);
  v[4] = unpack_lo<f16_t>(u.z); v[5] = unpack_hi<f16_t>(u.z); v[6] = unpack_lo<f16_t>(u.w); v[7] = unpack_hi<f16_t>(u.w);
}
__device__ __forceinline__ void store8(f16_t* p, const float (&v)[8]) {
  uint4 u;
  u.x = pack2<f16_t>(v[0], v[1]); u.y = pack2<f16_t>(v[2], v[3]); u.z = pack2<f16_t>(v[4], v[5]); u.w = pack2<f16_t>(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = u;
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  uint4 u;
  u.x = pack2<bf16_t>(v[0], v[1]); u.y = pack2<bf16_t>(v[2], v[3]); u.z = pack2<bf16_t>(v[4], v[5]); u.w = pack2<bf16_t>(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = u;
}

// the same 8 channels kept packed (4 or 8 registers): what a software-pipelined loop carries for the NEXT voxel
// while it computes the current one, so that two loads per lane are in flight instead of one
template <typename T> struct Pack8;
template <> struct Pack8<bf16_t> { uint4 u; };
template <> struct Pack8<float> { float4 a, b; };
template <> struct Pack8<f16_t> { uint4 u; };
__device__ __forceinline__ void load8p(const f16_t* p, Pack8<f16_t>& k) { k.u = *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void unpack8(const Pack8<f16_t>& k, float (&v)[8]) {
  v[0] = unpack_lo<f16_t>(k.u.x); v[1] = unpack_hi<f16_t>(k.u.x); v[2] = unpack_lo<f16_t>(k.u.y); v[3] = unpack_hi<f16_t>(k.u.y);
  v[4] = unpack_lo<f16_t>(k.u.z); v[5] = unpack_hi<f16_t>(k.u.z); v[6] = unpack_lo<f16_t>(k.u.w); v[7] = unpack_hi<f16_t>(k.u.w);
}
__device__ __forceinline__ void zero8p(Pack8<f16_t>& k) { k.u = make_uint4(0, 0, 0, 0); }
__device__ __forceinline__ void load8p(const bf16_t* p, Pack8<bf16_t>& k) { k.u = *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void load8p(const float* p, Pack8<float>& k) {
  k.a = reinterpret_cast<const float4*>(p)[0];
  k.b = reinterpret_cast<const float4*>(p)[1];
}
__device__ __forceinline__ void unpack8(const Pack8<bf16_t>& k, float (&v)[8]) {
  v[0] = bf16_bits_to_f32(k.u.x & 0xffffu); v[1] = bf16_bits_to_f32(k.u.x >> 16);
  v[2] = bf16_bits_to_f32(k.u.y & 0xffffu); v[3] = bf16_bits_to_f32(k.u.y >> 16);
  v[4] = bf16_bits_to_f32(k.u.z & 0xffffu); v[5] = bf16_bits_to_f32(k.u.z >> 16);
  v[6] = bf16_bits_to_f32(k.u.w & 0xffffu); v[7] = bf16_bits_to_f32(k.u.w >> 16);
}
__device__ __forceinline__ void unpack8(const Pack8<float>& k, float (&v)[8]) {
  v[0] = k.a.x; v[1] = k.a.y; v[2] = k.a.z; v[3] = k.a.w; v[4] = k.b.x; v[5] = k.b.y; v[6] = k.b.z; v[7] = k.b.w;
}
__device__ __forceinline__ void zero8p(Pack8<bf16_t>& k) { k.u = make_uint4(0, 0, 0, 0); }
__device__ __forceinline__ void zero8p(Pack8<float>& k) { k.a = k.b = make_float4(0.f, 0.f, 0.f, 0.f); }

// sum over the LPV (power of two, <= 16) consecutive lanes that share one voxel, on DPP (data-parallel primitives:
// a cross-lane operand fetched inside the VALU, a couple of cycles) instead of __shfl_xor (ds_bpermute: an LDS round
// trip of ~100 cycles in a dependent chain).  quad_perm [1,0,3,2] / [2,3,0,1] are the xor-1 / xor-2 butterflies;
// row_half_mirror (i <-> 7-i) and row_mirror (i <-> 15-i) pair up the already reduced quads / octets.
//
// DPP and the EXEC mask (found in round 4).  A DPP fetch with bound_ctrl returns 0 for a source lane that is disabled, and the
// DPP stage looks at EXEC late: where the compiler placed an s_and_saveexec (the `if (cg == 0)` that follows a group sum) right
// behind the last DPP instruction, the last 16 lanes of the wave occasionally saw the NARROWED mask -- their partner lanes read as
// 0 and the side value lost half of its channels.  Seen only while a second process kept the chip busy (two ranks on one GPU in the
// data-parallel tests; scripts/r4_stress3.py: 5 % of launches), never in the sums that are followed by more vector work.  The ISA
// lists the mirror case (a VALU write of EXEC needs 5 wait states before a DPP), not this one, and the hazard recognizer inserts
// nothing.  `dpp_settle` makes a value that came through DPP the operand of a plain vector move: the move cannot issue before the
// DPP instruction has completed, and a scalar write of EXEC behind it is ordered after it by the ordinary interlock.  Every
// reduction that ends in DPP passes its result through it.
// Workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest), and each XCD has its own L2.  Kernels whose
// neighbouring tiles share halo data re-number their tiles so that an XCD works on a contiguous run of them (tile t of `nt`
// for launch-order index b): the halo is then fetched into one L2 instead of several.
__device__ __forceinline__ int xcd_contiguous_tile(int b, int nt) {
  const int q = nt >> 3, r = nt & 7, xcd = b & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}
struct TileIdx3 { int x, y, z; };
__device__ __forceinline__ TileIdx3 xcd_contiguous_tile3() {   // the 3-D grid form: (x, y, z) of this workgroup's tile, x fastest
  const int b = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
  const int t = xcd_contiguous_tile(b, (int)(gridDim.x * gridDim.y * gridDim.z));
  TileIdx3 r;
  r.x = t % (int)gridDim.x;
  r.y = (t / (int)gridDim.x) % (int)gridDim.y;
  r.z = t / (int)(gridDim.x * gridDim.y);
  return r;
}
__device__ __forceinline__ float dpp_settle(float v) {
  asm volatile("v_mov_b32 %0, %0" : "+v"(v));
  return v;
}
__device__ __forceinline__ int dpp_settle(int v) {
  asm volatile("v_mov_b32 %0, %0" : "+v"(v));
  return v;
}
__device__ __forceinline__ unsigned long long dpp_settle(unsigned long long u) {
  unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
  asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(lo), "+v"(hi));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double dpp_settle(double v) {
  return __builtin_bit_cast(double, dpp_settle(__builtin_bit_cast(unsigned long long, v)));
}
// The same holds for the LDS cross-lane instructions (ds_bpermute behind __shfl_*, ds_swizzle): the compiler leaves them in
// flight across the s_and_saveexec of the `if (lane == 0)` that follows a wave reduction (the s_waitcnt sinks into the branch), and
// under the same conditions the first-layer block's statistics lost lanes (ec1.conv1.weight / ec1.conv_se.weight off by 1e-4 ..
// 7e-3 in 1 % of the steps of scripts/r4_stress.py).  Every cross-lane value is settled before it is used: the move needs the
// result, so the wait for it precedes any change of EXEC.
template <typename V> __device__ __forceinline__ V shfl_xor_settled(V v, int off) { return dpp_settle(__shfl_xor(v, off, 64)); }
template <int CTRL> __device__ __forceinline__ int dpp_fetch_bits(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ float dpp_fetch(float v) {
  return __builtin_bit_cast(float, dpp_fetch_bits<CTRL>(__builtin_bit_cast(int, v)));
}
template <int LPV> __device__ __forceinline__ float group_sum(float v) {
  static_assert(LPV == 1 || LPV == 2 || LPV == 4 || LPV == 8 || LPV == 16, "group_sum: LPV must be 1..16");
  if (LPV >= 2) v += dpp_fetch<0xB1>(v);    // quad_perm [1,0,3,2]
  if (LPV >= 4) v += dpp_fetch<0x4E>(v);    // quad_perm [2,3,0,1]
  if (LPV >= 8) v += dpp_fetch<0x141>(v);   // row_half_mirror
  if (LPV >= 16) v += dpp_fetch<0x140>(v);  // row_mirror
  return LPV >= 2 ? dpp_settle(v) : v;
}
// sum over all lanes with the same (lane % LPV): the lanes holding the same channels
template <int LPV> __device__ __forceinline__ float stride_sum(float v) {
#pragma unroll
  for (int off = 32; off >= LPV; off >>= 1) v += shfl_xor_settled(v, off);
  return v;
}
template <int LPV> __device__ __forceinline__ double stride_sum_d(double v) {
#pragma unroll
  for (int off = 32; off >= LPV; off >>= 1) v += shfl_xor_settled(v, off);
  return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
#endif  // __HIPCC__

// ---- launcher prototypes (one per kernel family; defined in the .hip files) ----------
// layout
int launch_pack_input(int dtype, const float* x_ncdhw, int in_channel, void* out_cl8, Dims d, hipStream_t s);
int launch_unpack_cl(int dtype, const void* in_cl, int C, float* out_ncdhw, Dims d, hipStream_t s);
int launch_pack_cl(int dtype, const float* in_ncdhw, int C, void* out_cl, int Cpad, Dims d, hipStream_t s);

// convolution (conv_igemm.hip)
size_t conv_wpack_bytes(int dtype, int taps, int cin, int cout);
int launch_conv_pack_weights(int dtype, const float* w_torch, int taps, int cin_w, int cout_w,
                             int transpose_flip, void* wpack, hipStream_t s);
int conv_stats_tiles(Dims d, int taps, int dil);   // partial-stat slots per sample written by the igemm kernel
struct ConvPackJob { const float* w; void* wpack; int taps, cin_w, cout_w, tflip; };
int launch_conv_pack_weights_multi(int dtype, const ConvPackJob* jobs, int n, hipStream_t s);
int launch_conv_igemm(int dtype, int taps, int dil, const SrcList& src, int cin_logical,
                      const void* wpack, const float* bias, const DstList& dst,
                      double* stats_partial, Dims d, hipStream_t s);
int launch_conv_naive(int dtype, int taps, int dil, const SrcList& src, int cin_logical,
                      const float* w_torch, int transpose_flip, const float* bias,
                      const DstList& dst, Dims d, hipStream_t s);

// streaming small-channel 3x3x3 convolution (conv_stream.hip): bf16, 8/16/32 source channels, <= 32 destination channels
bool conv_stream_supported(int dtype, int taps, int dil, int src_c, int dst_c);
size_t conv_stream_wpack_bytes(int src_c);
int conv_stream_slots(Dims d, int dil);
int launch_conv_stream_pack(int dtype, const float* w, int cin_w, int cout_w, int tflip, int src_c, int dst_c, void* wpack, hipStream_t s);
struct StreamPackJob { const float* w; void* wpack; int cin_w, cout_w, tflip, src_c, dst_c; };
int launch_conv_stream_pack_multi(int dtype, const StreamPackJob* jobs, int n, hipStream_t s);
int launch_conv_stream(int dtype, int dil, const void* src, int src_c, const void* wpack, const float* bias, void* dst, int dst_c,
                       int dst_accumulate, double* stats, Dims d, hipStream_t s);

// marching 3x3x3 convolution for 32 / 64 source channels (conv_march.hip): bf16 | f16, one or two equal sources, destinations
// in multiples of 16 channels; forward (bias, statistics) and data gradient (optionally accumulating)
bool conv_march_supported(int dtype, int taps, int dil, const SrcList& src, const DstList& dst);
size_t conv_march_wpack_bytes(int cin_e, int cout_e);
int conv_march_slots(Dims d, int dil, int cin_e, int cout_e);
int launch_conv_march_pack(int dtype, const float* w, int cin_w, int cout_w, int tflip, int cin_e, int cout_e, void* wpack, hipStream_t s);
struct MarchPackJob { const float* w; void* wpack; int cin_w, cout_w, tflip, cin_e, cout_e; };
int launch_conv_march_pack_multi(int dtype, const MarchPackJob* jobs, int n, hipStream_t s);   // every layer of a pass in one launch
int launch_conv_march(int dtype, int dil, const SrcList& src, const void* wpack, const float* bias, const DstList& dst, double* stats,
                      Dims d, hipStream_t s);

// streaming small-channel weight gradient (wgrad_stream.hip)
bool wgrad_stream_supported(int dtype, int taps, int dil, int x_c, int dy_c);
size_t wgrad_stream_workspace_bytes(int x_c, int dy_c, int dil, Dims d);
int launch_wgrad_stream(int dtype, int dil, const void* x, int x_c, int cin_w, const void* dy, int dy_c, int cout_w, float* dw,
                        void* workspace, size_t ws_bytes, Dims d, hipStream_t s);

// weight gradient (wgrad.hip)
size_t wgrad_workspace_bytes(int taps, int cin, int cout);
int launch_wgrad(int dtype, int taps, int dil, const SrcList& x, int cin_logical, const void* dy,
                 int cout, float* dw_torch, void* workspace, size_t ws_bytes, Dims d,
                 hipStream_t s, bool allow_march = true);
// marching weight gradient (wgrad_march.hip); `supported` includes the size gate launch_wgrad dispatches on
bool wgrad_march_supported(int dtype, int taps, int dil, const SrcList& x, int cin_logical, int cout, Dims d);
int launch_wgrad_march(int dtype, int taps, int dil, const SrcList& x, int cin_logical, const void* dy, int cout,
                       float* dw, void* workspace, size_t ws_bytes, Dims d, hipStream_t s);
// 1x1x1 weight gradient of the aggregation convolutions (wgrad_1x1.hip)
bool wgrad_1x1_supported(int dtype, const SrcList& x, int cin_logical, int cout, Dims d);
int launch_wgrad_1x1(int dtype, const SrcList& x, int cin_logical, const void* dy, int cout, float* dw, void* workspace,
                     size_t ws_bytes, Dims d, hipStream_t s);
int launch_wgrad_naive(int dtype, int taps, int dil, const SrcList& x, int cin_logical,
                       const void* dy, int cout, float* dw_torch, Dims d, hipStream_t s);

// normalisation / gates / cat (epilogue.hip)
int epi_partials(Dims d);         // partial slots per sample used by the epilogue kernels
int launch_channel_stats(int dtype, const void* t, int C, double* partial, Dims d, hipStream_t s);
int launch_stats_finalize(const double* partial, int slots, int C, int N, long long count,
                          float eps, int mode, float* out_a, float* out_b, hipStream_t s);
struct SseParams {
  const float* w_se;      // [C]
  const float* w_se2;     // [C] or null (one gate)
  const float* w_side;    // [2][C]
  const float* b_side;    // [2]
  float slope;
};
struct SseHead {          // how the 2-channel side output is consumed
  float* side_out;        // fp32 [N][V][2] or null
  float* level_map;       // fp32 [N][V] head pre-activation map of this level, or null
  int level_accumulate;   // 0: overwrite level_map, 1: +=
  const float* head_w;    // [2] head weights of this block's two channels
  const float* drop;      // [N][drop_stride] DropLayer scales (points at this block's channel 0) or null
  int drop_stride;
};
int launch_sse_fwd(int dtype, const void* raw, const float* mean, const float* rstd, int C,
                   const SseParams& p, void* e_out, const SseHead& head, Dims d, hipStream_t s);
struct SseBwdIn {
  const void* g_e;        // gradient w.r.t. e (T) or null
  const float* g_side;    // fp32 [N][V][2] gradient w.r.t. the side map, or null
  const float* g_level;   // fp32 [N][V] gradient w.r.t. the level map, or null
};
// partial parameter-gradient record per (sample, slot):  4*C + 4 floats
//   [0,C) dw_se  [C,2C) dw_se2  [2C,4C) dw_side[2][C]  [4C,4C+2) db_side  [4C+2,4C+4) dhead_w
// m1 == nullptr: pass A (f64 stat sums + parameter-gradient records); else pass B (writes draw_out, may alias g_e)
int launch_sse_bwd(int dtype, const void* raw, const float* mean, const float* rstd, int C,
                   const SseParams& p, const SseBwdIn& g, const SseHead& head, const float* m1,
                   const float* m2, void* draw_out, double* stat_partial, float* pgrad_partial,
                   Dims d, hipStream_t s);
int launch_gate_bwd_finalize(const double* stat_partial, int slots, int C, int N, long long count, float* m1, float* m2,
                             const float* pgrad_partial, int records, float* dw_se, float* dw_se2, float* dw_side,
                             float* db_side, float* dhead_w, hipStream_t s);
int launch_pgrad_reduce(const float* pgrad_partial, int records, int C, float* dw_se,
                        float* dw_se2, float* dw_side, float* db_side, float* dhead_w,
                        hipStream_t s);
int launch_cat_fwd(int dtype, const void* raw, const float* mean, const float* rstd,
                   const void* raw2, const float* mean2, const float* rstd2, int C, float slope,
                   void* out, Dims d, hipStream_t s);
int launch_cat_bwd(int dtype, const void* g_out, const void* raw, const float* mean,
                   const float* rstd, const void* raw2, const float* mean2, const float* rstd2,
                   int C, float slope, const float* m1, const float* m2, const float* m1b,
                   const float* m2b, void* dx, void* dx2, double* stat_partial,
                   double* stat_partial2, Dims d, hipStream_t s);

// pooling / interpolation / heads (resample.hip)
int xbranch_moment_slots(Dims d);
int launch_xbranch_moments(int dtype, const void* x_in, double* partial, Dims d, hipStream_t s);
int launch_xbranch_stats(const double* partial, int slots, const float* w2, int C, int in_channel, int N, long long count,
                         float eps, float* mean2, float* rstd2, double* moments_out, hipStream_t s);
int launch_cat_fwd_x(int dtype, const void* raw, const float* mean, const float* rstd, const void* x_in, const float* w2,
                     int in_channel, const float* mean2, const float* rstd2, int C, float slope, void* out, Dims d,
                     hipStream_t s);
int launch_cat_fwd_x_pool(int dtype, const void* raw, const float* mean, const float* rstd, const void* x_in, const float* w2,
                          int in_channel, const float* mean2, const float* rstd2, int C, float slope, void* out, void* pooled,
                          Dims d, hipStream_t s, unsigned* argmax = nullptr);
int launch_cat_bwd_x(int dtype, const void* g_out, const void* raw, const float* mean, const float* rstd, const void* x_in,
                     const float* w2, int in_channel, const float* mean2, const float* rstd2, int C, float slope,
                     const float* m1, const float* m2, const float* m1b, const float* m2b, void* dx, double* stat_partial,
                     double* stat_partial2, double* xw_partial, Dims d, hipStream_t s, const unsigned* pool_argmax = nullptr,
                     const void* pool_g = nullptr);   // pool_*: gradient of the max-pool consuming the block's output, added on the fly
int launch_cat_xgrad_finalize(const double* xw_partial, const double* stat_partial2, int slots, const double* moments, const float* w2,
                              int C, int in_channel, int N, float eps, float* dw, hipStream_t s);
int launch_xbranch_values(int dtype, const void* x_in, const float* w2, int C, int in_channel, float* out_ncdhw, Dims d, hipStream_t s);   // diagnostic
int launch_maxpool_fwd(int dtype, const void* in, int C, void* out, Dims din, hipStream_t s);
// max-pool backward from the arg-max words written by launch_cat_fwd_x_pool ([N][Vo][C/8] uint32, 3 bits per channel)
int launch_maxpool_bwd_idx(int dtype, const unsigned* argmax, const void* g_out, int C, void* g_in, int accumulate, Dims d, hipStream_t s);
int launch_maxpool_bwd(int dtype, const void* in, const void* g_out, int C, void* g_in,
                       int accumulate, Dims din, hipStream_t s);
int launch_upsample2_fwd(int dtype, const void* in, int C, void* out, Dims din, hipStream_t s);
int launch_upsample2_bwd(int dtype, const void* g_out, int C, void* g_in, int accumulate,
                         Dims din, hipStream_t s);
int launch_multi_zero(float* const* ptrs, const int* counts, int n, hipStream_t s);
// n_classes > 1: the general head path (classes.hip)
int class_max();
int class_grad_records(Dims d);
int launch_side_to_level(const float* side, const float* head_w, int wstride, const float* drop, int drop_stride, int K, float* level,
                         int accumulate, Dims d, hipStream_t s);
int launch_level_to_side_grad(const float* glev, long long cstride, long long nstride, const float* side, const float* head_w, int wstride,
                              const float* drop, int drop_stride, int K, float* g_side, double* partial, float* dhead, Dims d,
                              hipStream_t s);
int launch_class_bias_grad(const float* per_sample, int N, int K, float* out, hipStream_t s);
// connected components / metrics (components.hip)
size_t cc_workspace_bytes(int H, int W, int Z);
int launch_largest_component(const unsigned char* vol, int H, int W, int Z, int rule, unsigned char* out, int* status_dev,
                             void* workspace, size_t ws_bytes, hipStream_t s);
size_t metric_out_bytes(int nbins);
int launch_metric_sums(const unsigned char* pred, const unsigned char* label, const unsigned char* skel, const int* parsing, long long n,
                       int nbins, void* out, size_t out_bytes, hipStream_t s);
// input pipeline (pipeline.hip)
int launch_crop_batch(const void* img, int img_dtype, const unsigned char* label, const void* weight, int w_dtype,
                      const unsigned char* skel, int D, int H, int W, int cube, int ncrop, const int* starts, const int* aug,
                      double weight_exponent, int f64_math, float* data_out, float* label_out, float* weight_out, float* skel_out,
                      hipStream_t s);
int launch_hu_two_channel(const void* img, int img_dtype, long long nvox, int f64_math, float* out, hipStream_t s);
// sliding-window assembly (window.hip)
int launch_window_gather(const float* vol, int C, int X, int Y, int Z, int cube, int nwin, const int* starts, float* out, hipStream_t s);
int launch_window_accumulate(const float* logits, int apply_sigmoid, int nwin, const int* starts, int cube, double* acc, int X, int Y,
                             int Z, hipStream_t s);
int launch_window_finalize(const double* acc, int X, int Y, int Z, int cube, int nx, const int* xs, int ny, const int* ys, int nz,
                           const int* zs, int dup0, double* out, hipStream_t s);
size_t dti_workspace_bytes(int h, int w, int z);
int launch_dti(const double* pred, int h, int w, int z, double h_thresh, double l_thresh, int pred_dtype, unsigned char* out,
               void* workspace, size_t ws_bytes, hipStream_t s);
int launch_adamw(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 const long long* counts, int n, double lr, double beta1, double beta2, double eps, double weight_decay,
                 int step, int maximize, hipStream_t s);
// raises a kernel's dynamic-LDS limit once per (instantiation, device): under a lock, the device's bit is set on success only
int configure_kernel_lds(unsigned long long& mask, const void* fn, int bytes);
const void* device_zero_page();   // >= 256 zero bytes on the current device (allocated once per device, never freed)
int launch_side_upsample(const float* side, int C, int scale, float* out_ncdhw, int c_total,
                         int c_off, Dims dlow, hipStream_t s);
int launch_head_fwd(const float* const* level_maps, int nlevels, const float* bias, float* pred,
                    Dims d0, hipStream_t s);
int launch_head_bwd(const float* g_pred, float* const* g_levels, int nlevels, float* tmp,
                    float* g_bias, Dims d0, hipStream_t s);
size_t head_bwd_tmp_floats(Dims d0);

// losses (loss.hip)
#define SEUNET_LOSS_NSUMS 7
int loss_partials();
int launch_loss_sums(const float* pred, int apply_sigmoid, const float* target,
                     const float* weight, const float* skel, long long n, float* partial,
                     double* sums, hipStream_t s, int terms = 7);
int launch_loss_value(const double* sums0, double c_dice0, double c_gul0, double c_atr0, const double* sums1, double c_dice1,
                      double c_gul1, double c_atr1, float* value, hipStream_t s);
int launch_loss_grad(const float* pred, int apply_sigmoid, const float* target,
                     const float* weight, const float* skel, long long n, const double* sums,
                     float c_dice, float c_gul, float c_atr, float g_scale,
                     const float* g_scale_dev, float* g_pred, hipStream_t s);
}  // namespace seunet
