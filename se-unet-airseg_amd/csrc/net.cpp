// Whole-network executor: SE_UNet.forward (reference SE_UNet.py:181-238) and its backward, as one
// stream-ordered sequence of the kernels in this directory.  The graph is data (kOps below) and two small
// interpreters walk it forwards / backwards; nothing here allocates: every intermediate lives at a fixed
// offset of a caller-owned workspace whose layout is a pure function of seunet_net_desc (so the same
// layout is recomputed by the backward call).
#include "seunet_common.h"
#include "../../include/seunet_hip.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace seunet {
namespace {

// ---------------------------------------------------------------------------------------------------
// parameter registry == the reference's state_dict order (SE_UNet.py:108-151)
// ---------------------------------------------------------------------------------------------------
struct BlockDesc { const char* name; char kind; int cin; int cout; };  // cin -1 = in_channel; 'g' one gate, 'G' two, 'c' cat
const BlockDesc kBlocks[] = {
    {"ec1", 'g', -1, 8},   {"ec2", 'g', 8, 16},    {"ec3", 'g', 16, 32},  {"ec33", 'c', 56, 32},  {"x33", 'c', -1, 32},
    {"ec4", 'G', 32, 32},  {"ec5", 'G', 32, 32},   {"ec6", 'G', 32, 64},  {"ec63", 'c', 128, 64}, {"x63", 'c', -1, 64},
    {"ec7", 'G', 64, 64},  {"ec8", 'G', 64, 64},   {"ec9", 'G', 64, 64},  {"ec93", 'c', 192, 64}, {"x93", 'c', -1, 64},
    {"ec10", 'G', 64, 64}, {"ec11", 'G', 64, 64},  {"ec12", 'G', 64, 64}, {"ec123", 'c', 192, 64},
    {"dc1", 'G', 128, 64}, {"dc2", 'G', 64, 64},   {"dc22", 'c', 128, 64},
    {"dc3", 'G', 128, 64}, {"dc4", 'G', 64, 32},   {"dc42", 'c', 96, 32},
    {"dc5", 'g', 64, 32},  {"dc6", 'g', 32, 16},   {"dc62", 'c', 48, 16},
};
constexpr int kNumBlocks = sizeof(kBlocks) / sizeof(kBlocks[0]);

struct ParamInfo { std::string name; int shape[5]; int ndim; };

std::vector<ParamInfo> build_registry(const seunet_net_desc& d) {
  std::vector<ParamInfo> r;
  auto add5 = [&](const std::string& n, int a, int b, int k) { r.push_back({n, {a, b, k, k, k}, 5}); };
  auto add1 = [&](const std::string& n, int a) { r.push_back({n, {a, 0, 0, 0, 0}, 1}); };
  for (int i = 0; i < kNumBlocks; ++i) {
    const BlockDesc& b = kBlocks[i];
    const int ci = b.cin < 0 ? d.in_channel : b.cin * d.width_mult, co = b.cout * d.width_mult;
    const std::string n = b.name;
    if (b.kind == 'c') { add5(n + ".conv1.weight", co, ci, 1); continue; }
    add5(n + ".conv1.weight", co, ci, 3);
    add1(n + ".conv1.bias", co);
    add5(n + ".conv2.weight", 2, co, 1);
    add1(n + ".conv2.bias", 2);
    add5(n + ".conv_se.weight", 1, co, 1);
    if (b.kind == 'G') add5(n + ".conv_se2.weight", 1, co, 1);
  }
  add5("dc0_0.weight", d.n_classes, 24, 1);
  add1("dc0_0.bias", d.n_classes);
  add5("dc0_1.weight", d.n_classes, 12, 1);
  add1("dc0_1.bias", d.n_classes);
  return r;
}

int find_param(const std::vector<ParamInfo>& reg, const std::string& name) {
  for (size_t i = 0; i < reg.size(); ++i)
    if (reg[i].name == name) return (int)i;
  return -1;
}

// ---------------------------------------------------------------------------------------------------
// the graph
// ---------------------------------------------------------------------------------------------------
enum TId {
  T_X0, T_X1, T_X2,
  T_E0, T_E1A, T_E1_1, T_E1, T_E2IN, T_E2, T_E3A, T_E3_1, T_E3, T_E4IN, T_E4, T_E5A, T_E5_1, T_E5,
  T_E6IN, T_E6, T_E7A, T_E7_1, T_E7, T_E8, T_D0A, T_D0_1, T_D0, T_D1U, T_D1A, T_D1_1, T_D1, T_D2U, T_D2A, T_D2_1,
  T_COUNT
};
struct TDesc { int level; int cbase; };  // cbase 0 = padded network input (8 channels)
const TDesc kT[T_COUNT] = {
    {0, 0}, {1, 0}, {2, 0},
    {0, 8}, {0, 16}, {0, 32}, {0, 32}, {1, 32}, {1, 32}, {1, 32}, {1, 64}, {1, 64}, {2, 64}, {2, 64}, {2, 64}, {2, 64}, {2, 64},
    {3, 64}, {3, 64}, {3, 64}, {3, 64}, {3, 64}, {2, 64}, {2, 64}, {2, 64}, {2, 64}, {1, 64}, {1, 64}, {1, 32}, {1, 32}, {0, 32}, {0, 32}, {0, 16},
};

enum OpKind { OP_GATED, OP_CAT, OP_POOL, OP_UP };
struct OpDesc {
  OpKind kind; const char* name; int nsrc; int src[3]; int dst;
  int dil; int gates; int head; int m;   // gated: dilation, #gates, 0 = encoder head / 1 = decoder head, side slot
  const char* xname; int xsrc;          // cat: optional raw-input branch
};
const OpDesc kOps[] = {
    {OP_GATED, "ec1", 1, {T_X0, 0, 0}, T_E0, 1, 1, 0, 0, nullptr, 0},
    {OP_GATED, "ec2", 1, {T_E0, 0, 0}, T_E1A, 1, 1, 0, 1, nullptr, 0},
    {OP_GATED, "ec3", 1, {T_E1A, 0, 0}, T_E1_1, 2, 1, 0, 2, nullptr, 0},
    {OP_CAT, "ec33", 3, {T_E1_1, T_E0, T_E1A}, T_E1, 0, 0, 0, 0, "x33", T_X0},
    {OP_POOL, "pool0", 1, {T_E1, 0, 0}, T_E2IN, 0, 0, 0, 0, nullptr, 0},
    {OP_POOL, "pool0x", 1, {T_X0, 0, 0}, T_X1, 0, 0, 0, 0, nullptr, 0},
    {OP_GATED, "ec4", 1, {T_E2IN, 0, 0}, T_E2, 1, 2, 0, 3, nullptr, 0},
    {OP_GATED, "ec5", 1, {T_E2, 0, 0}, T_E3A, 2, 2, 0, 4, nullptr, 0},
    {OP_GATED, "ec6", 1, {T_E3A, 0, 0}, T_E3_1, 2, 2, 0, 5, nullptr, 0},
    {OP_CAT, "ec63", 3, {T_E3_1, T_E2, T_E3A}, T_E3, 0, 0, 0, 0, "x63", T_X1},
    {OP_POOL, "pool1", 1, {T_E3, 0, 0}, T_E4IN, 0, 0, 0, 0, nullptr, 0},
    {OP_POOL, "pool1x", 1, {T_X1, 0, 0}, T_X2, 0, 0, 0, 0, nullptr, 0},
    {OP_GATED, "ec7", 1, {T_E4IN, 0, 0}, T_E4, 1, 2, 0, 6, nullptr, 0},
    {OP_GATED, "ec8", 1, {T_E4, 0, 0}, T_E5A, 2, 2, 0, 7, nullptr, 0},
    {OP_GATED, "ec9", 1, {T_E5A, 0, 0}, T_E5_1, 2, 2, 0, 8, nullptr, 0},
    {OP_CAT, "ec93", 3, {T_E5_1, T_E4, T_E5A}, T_E5, 0, 0, 0, 0, "x93", T_X2},
    {OP_POOL, "pool2", 1, {T_E5, 0, 0}, T_E6IN, 0, 0, 0, 0, nullptr, 0},
    {OP_GATED, "ec10", 1, {T_E6IN, 0, 0}, T_E6, 1, 2, 0, 9, nullptr, 0},
    {OP_GATED, "ec11", 1, {T_E6, 0, 0}, T_E7A, 1, 2, 0, 10, nullptr, 0},
    {OP_GATED, "ec12", 1, {T_E7A, 0, 0}, T_E7_1, 1, 2, 0, 11, nullptr, 0},
    {OP_CAT, "ec123", 3, {T_E7_1, T_E6, T_E7A}, T_E7, 0, 0, 0, 0, nullptr, 0},
    {OP_UP, "up0", 1, {T_E7, 0, 0}, T_E8, 0, 0, 0, 0, nullptr, 0},
    {OP_GATED, "dc1", 2, {T_E8, T_E5, 0}, T_D0A, 1, 2, 1, 0, nullptr, 0},
    {OP_GATED, "dc2", 1, {T_D0A, 0, 0}, T_D0_1, 1, 2, 1, 1, nullptr, 0},
    {OP_CAT, "dc22", 2, {T_D0_1, T_D0A, 0}, T_D0, 0, 0, 0, 0, nullptr, 0},
    {OP_UP, "up1", 1, {T_D0, 0, 0}, T_D1U, 0, 0, 0, 0, nullptr, 0},
    {OP_GATED, "dc3", 2, {T_D1U, T_E3, 0}, T_D1A, 1, 2, 1, 2, nullptr, 0},
    {OP_GATED, "dc4", 1, {T_D1A, 0, 0}, T_D1_1, 1, 2, 1, 3, nullptr, 0},
    {OP_CAT, "dc42", 2, {T_D1_1, T_D1A, 0}, T_D1, 0, 0, 0, 0, nullptr, 0},
    {OP_UP, "up2", 1, {T_D1, 0, 0}, T_D2U, 0, 0, 0, 0, nullptr, 0},
    {OP_GATED, "dc5", 2, {T_D2U, T_E1, 0}, T_D2A, 1, 1, 1, 4, nullptr, 0},
    {OP_GATED, "dc6", 1, {T_D2A, 0, 0}, T_D2_1, 1, 1, 1, 5, nullptr, 0},
    // dc62 (SE_UNet.py:148,230) is dead: never evaluated, no gradient
};
constexpr int kNumOps = sizeof(kOps) / sizeof(kOps[0]);

inline bool is_input(int t) { return t <= T_X2; }
// levels on which the marching conv pays: full 32-voxel rows and enough (y, x) patches x planes for one workgroup per CU
// (dilation 2 already at 32^3: the tiled kernel runs it on eight 16^3 parity sub-lattices, where its tiles are mostly halo --
// measured on 4 x 32^3, 64 -> 64 channels: forward 0.043 vs 0.057 ms, data gradient 0.039 vs 0.059 ms; dilation 1 at that size
// stays on the tiled kernel, 0.036 vs 0.041 ms)
inline bool march_level(const Dims& d, int dil) {
  static const long long minvox = [] { const char* e = getenv("SEUNET_MARCH_MINVOX"); return e ? atoll(e) : 48LL * 48 * 48; }();
  static const bool no_coarse = getenv("SEUNET_MARCH_NO_COARSE") != nullptr;   // (diagnostic switch for A/B timing)
  return d.W >= 32 && (d.vox() >= minvox || (!no_coarse && dil == 2 && d.vox() >= 32LL * 32 * 32));
}

// ---------------------------------------------------------------------------------------------------
// workspace plan
// ---------------------------------------------------------------------------------------------------
struct OpRes {
  size_t raw = 0, raw2 = 0, mean = 0, rstd = 0, mean2 = 0, rstd2 = 0, xtot = 0, wp_f = 0, wp_d = 0, wp_x = 0;
  int cin = 0, cout = 0, taps = 0;
  bool need_dgrad = false;
  bool stream_f = false, stream_d = false, stream_w = false;   // forward / data gradient / weight gradient on the streaming kernels
  bool march_f = false, march_d = false;                       // forward / data gradient on the marching kernel (conv_march.hip)
  size_t pool_idx = 0; bool has_pool_idx = false;              // OP_POOL behind a fused aggregation block: arg-max words of the forward
  size_t side = 0;                                             // n_classes > 1: the block's 2-channel side map, f32 [N][V][2] (classes.hip)
};

struct Plan {
  seunet_net_desc d;
  size_t esz;
  Dims dims[4];
  int C[T_COUNT];
  size_t feat[T_COUNT], grad[T_COUNT];
  OpRes op[kNumOps];
  size_t lvl[2][4], glvl[2][4];
  size_t stats, stats2, pgrad, m1, m2, m1b, m2b, wgrad_ws, head_tmp, gx, gx_bytes = 0, xwp = 0, xmom = 0;
  size_t gside = 0, cls_part = 0, cls_bias = 0;                // n_classes > 1: side-map gradient of one block, head-gradient records, per-(sample, class) bias sums
  // x-branches (x33 / x63 / x93) recomputed from the <= 2-channel input instead of materialised (csrc/epilogue.hip, XR)
  bool fuse_x = false;
  bool use_stream = true;
  bool use_march = true;
  size_t wgrad_ws_bytes;
  size_t total;
  int stat_slots_max;

  size_t cur = 0;
  size_t take(size_t bytes) { size_t o = cur; cur = align_up(cur + bytes, 256); return o; }

  int init(const seunet_net_desc& desc) {
    d = desc;
    SEUNET_CHECK(d.batch >= 1 && d.in_channel >= 1 && d.in_channel <= 8, "net: batch/in_channel out of range");
    SEUNET_CHECK(d.n_classes >= 1 && d.n_classes <= class_max(), "net: n_classes=%d out of range (1 .. %d)", d.n_classes, class_max());
    SEUNET_CHECK(d.d % 8 == 0 && d.h % 8 == 0 && d.w % 8 == 0 && d.d >= 8 && d.h >= 8 && d.w >= 8,
                 "net: spatial extents (%d,%d,%d) must be multiples of 8", d.d, d.h, d.w);
    SEUNET_CHECK(d.width_mult == 1 || d.width_mult == 2, "net: width_mult %d unsupported (1 or 2)", d.width_mult);
    SEUNET_CHECK(dtype_ok(d.dtype), "net: dtype %d unsupported (SEUNET_F32 | SEUNET_BF16 | SEUNET_F16)", d.dtype);
    esz = dtype_size(d.dtype);
    for (int l = 0; l < 4; ++l) dims[l] = Dims{d.batch, d.d >> l, d.h >> l, d.w >> l};
    for (int t = 0; t < T_COUNT; ++t) C[t] = kT[t].cbase == 0 ? 8 : kT[t].cbase * d.width_mult;
    // (the two sources of a two-source marching layer are allocated next to each other: the kernel then reaches both through one
    // 32-bit buffer descriptor, conv_march.hip `BUF` and wgrad_march.hip)
    bool placed[T_COUNT] = {};
    for (int t = 0; t < T_COUNT; ++t) {
      const size_t bytes = (size_t)d.batch * dims[kT[t].level].vox() * C[t] * esz;
      if (!placed[t]) { feat[t] = take(bytes); placed[t] = true; }
      // (dc5 only.  The same placement for dc3 and dc1 cost their tiled forward 5 % -- conv_fwd:dc3 0.465 -> 0.49 ms, same box,
      // alternating runs -- and their sources are 1.2 GB apart at batch 4 anyway; beyond 4 GB the weight gradient of those two
      // layers takes the tiled kernel)
      const int mate = t == T_E1 ? T_D2U : -1;
      if (mate >= 0 && !placed[mate]) {
        feat[mate] = take((size_t)d.batch * dims[kT[mate].level].vox() * C[mate] * esz);
        placed[mate] = true;
      }
      grad[t] = is_input(t) ? 0 : take(bytes);
    }
    size_t gx_max = 0, wg_max = 0, xw_max = 0, xmom_max = 0;
    fuse_x = d.in_channel <= 2 && d.conv_impl != SEUNET_CONV_NAIVE;
    use_stream = getenv("SEUNET_NO_STREAM") == nullptr;   // (diagnostic switch for A/B timing; the default is on)
    use_march = getenv("SEUNET_NO_MARCH") == nullptr;     // (likewise)
    int slots_max = 1, cmax = 8;
    for (int i = 0; i < kNumOps; ++i) {
      const OpDesc& o = kOps[i];
      OpRes& r = op[i];
      if (o.kind == OP_POOL && i > 0 && kOps[i - 1].kind == OP_CAT && kOps[i - 1].xname && kOps[i - 1].dst == o.src[0] &&
          d.in_channel <= 2 && d.conv_impl != SEUNET_CONV_NAIVE && !is_input(o.src[0])) {
        // the aggregation block's forward writes this pool (and the position of each maximum, which the backward pass then
        // reads instead of the block output and the pooled tensor)
        const Dims& dl = dims[kT[o.dst].level];
        r.pool_idx = take((size_t)d.batch * dl.vox() * (C[o.src[0]] / 8) * 4);
        r.has_pool_idx = true;
      }
      if (o.kind == OP_POOL || o.kind == OP_UP) continue;
      const int lv = kT[o.dst].level;
      r.cout = C[o.dst];
      r.taps = o.kind == OP_GATED ? 27 : 1;
      int cin = 0;
      bool any_grad = false;
      for (int k = 0; k < o.nsrc; ++k) { cin += C[o.src[k]]; any_grad |= !is_input(o.src[k]); }
      if (o.nsrc == 1 && is_input(o.src[0])) cin = d.in_channel;
      r.cin = cin;
      r.need_dgrad = any_grad;
      const size_t act = (size_t)d.batch * dims[lv].vox() * r.cout * esz;
      r.raw = take(act);
      r.mean = take((size_t)d.batch * r.cout * 4);
      r.rstd = take((size_t)d.batch * r.cout * 4);
      // small-channel 3x3x3 layers with one source tensor (ec1 / ec2 / ec3 / dc6 at width 1) run on the streaming kernel
      // (the streaming kernels address one sample through 32-bit buffer offsets; a sample of 4 GB or more takes the general kernels)
      const bool stream_ok = use_stream && o.kind == OP_GATED && o.nsrc == 1 && d.conv_impl != SEUNET_CONV_NAIVE &&
                             (long long)dims[kT[o.dst].level].D * dims[kT[o.dst].level].H * dims[kT[o.dst].level].W * 32 * (long long)esz < 0xFFFFFFFFll;
      r.stream_f = stream_ok && conv_stream_supported(d.dtype, 27, o.dil, C[o.src[0]], r.cout);
      r.stream_d = stream_ok && r.need_dgrad && conv_stream_supported(d.dtype, 27, o.dil, r.cout, C[o.src[0]]);
      // 32 / 64-input-channel 3x3x3 layers of the levels that fill the chip with 32-voxel rows run on the marching kernel
      // (one workgroup per CU, weights in registers): dc5, dc4, ec4..ec6 and the data gradients of those and of dc3
      if (use_march && o.kind == OP_GATED && d.conv_impl != SEUNET_CONV_NAIVE && !r.stream_f && march_level(dims[lv], o.dil)) {
        SrcList sl{}; DstList dl{};
        sl.n = o.nsrc;
        for (int k = 0; k < o.nsrc; ++k) sl.C[k] = C[o.src[k]];
        dl.n = 1; dl.C[0] = r.cout;
        r.march_f = conv_march_supported(d.dtype, 27, o.dil, sl, dl);
        if (r.need_dgrad && !r.stream_d) {
          SrcList gs{}; DstList gd{};
          gs.n = 1; gs.C[0] = r.cout;
          gd.n = o.nsrc;
          for (int k = 0; k < o.nsrc; ++k) gd.C[k] = C[o.src[k]];
          r.march_d = conv_march_supported(d.dtype, 27, o.dil, gs, gd);
        }
      }
      r.wp_f = take(r.stream_f ? conv_stream_wpack_bytes(C[o.src[0]])
                               : (r.march_f ? conv_march_wpack_bytes(cin, r.cout) : conv_wpack_bytes(d.dtype, r.taps, r.cin, r.cout)));
      if (r.need_dgrad) r.wp_d = take(r.stream_d ? conv_stream_wpack_bytes(r.cout)
                                                 : (r.march_d ? conv_march_wpack_bytes(r.cout, cin) : conv_wpack_bytes(d.dtype, r.taps, r.cout, r.cin)));
      if (r.stream_f) slots_max = std::max(slots_max, conv_stream_slots(dims[lv], o.dil));
      if (r.march_f) slots_max = std::max(slots_max, conv_march_slots(dims[lv], o.dil, cin, r.cout));
      r.stream_w = stream_ok && wgrad_stream_supported(d.dtype, 27, o.dil, C[o.src[0]], r.cout);
      if (r.stream_w) wg_max = std::max(wg_max, wgrad_stream_workspace_bytes(C[o.src[0]], r.cout, o.dil, dims[lv]));
      wg_max = std::max(wg_max, wgrad_workspace_bytes(r.taps, r.cin, r.cout));
      if (o.xname) {
        r.mean2 = take((size_t)d.batch * r.cout * 4);
        r.rstd2 = take((size_t)d.batch * r.cout * 4);
        if (fuse_x) {
          xw_max = std::max(xw_max, (size_t)d.batch * epi_partials(dims[lv]) * r.cout * 2 * 8);
          r.xtot = take((size_t)d.batch * 5 * 8);      // the input's moments per sample, kept for the backward pass
          xmom_max = std::max(xmom_max, (size_t)d.batch * xbranch_moment_slots(dims[lv]) * 5 * 8);
        } else {
          r.raw2 = take(act);
          r.wp_x = take(conv_wpack_bytes(d.dtype, 1, d.in_channel, r.cout));
          gx_max = std::max(gx_max, act);
          wg_max = std::max(wg_max, wgrad_workspace_bytes(1, d.in_channel, r.cout));
        }
      }
      slots_max = std::max(slots_max, std::max(std::max(conv_stats_tiles(dims[lv], 27, 1), conv_stats_tiles(dims[lv], 27, 2)), epi_partials(dims[lv])));
      cmax = std::max(cmax, r.cout);
      if (d.n_classes > 1 && o.kind == OP_GATED) r.side = take((size_t)d.batch * dims[lv].vox() * 2 * 4);
    }
    stat_slots_max = slots_max;
    for (int h = 0; h < 2; ++h)
      for (int l = 0; l < 4; ++l) {
        lvl[h][l] = take((size_t)d.n_classes * d.batch * dims[l].vox() * 4);       // [class][N][V]
        glvl[h][l] = take((size_t)d.n_classes * d.batch * dims[l].vox() * 4);
      }
    if (d.n_classes > 1) {
      gside = take((size_t)d.batch * dims[0].vox() * 2 * 4);
      cls_part = take((size_t)d.batch * 256 * 2 * d.n_classes * 8);
      cls_bias = take((size_t)d.batch * d.n_classes * 4);
    }
    const size_t stat_bytes = (size_t)d.batch * slots_max * cmax * 2 * 8;   // f32 forward / f64 backward partials
    stats = take(stat_bytes);
    stats2 = take(stat_bytes);
    pgrad = take((size_t)d.batch * slots_max * (4 * cmax + 4) * 4);
    m1 = take((size_t)d.batch * cmax * 4);
    m2 = take((size_t)d.batch * cmax * 4);
    m1b = take((size_t)d.batch * cmax * 4);
    m2b = take((size_t)d.batch * cmax * 4);
    wgrad_ws_bytes = wg_max;
    wgrad_ws = take(wg_max);
    head_tmp = take(head_bwd_tmp_floats(dims[0]) * 4);
    gx = take(gx_max);
    gx_bytes = gx_max;
    xwp = take(xw_max);
    xmom = take(xmom_max);
    total = cur;
    return 0;
  }
};

struct Exec {
  Plan p;
  std::vector<ParamInfo> reg;
  unsigned char* ws = nullptr;
  const float* const* params = nullptr;
  hipStream_t s = nullptr;

  void mark(const std::string& tag) const { if (prof_on()) prof_mark(tag.c_str(), s); }
  void* at(size_t off) const { return ws + off; }
  float* fat(size_t off) const { return reinterpret_cast<float*>(ws + off); }
  double* dat(size_t off) const { return reinterpret_cast<double*>(ws + off); }
  const float* P(const std::string& name) const {
    const int i = find_param(reg, name);
    return i < 0 ? nullptr : params[i];
  }

  int setup(const seunet_net_desc* desc, const float* const* prm, void* workspace, size_t bytes, hipStream_t st) {
    SEUNET_CHECK(desc && prm && workspace, "net: null argument");
    if (int e = p.init(*desc)) return e;
    SEUNET_CHECK(bytes >= p.total, "net: workspace too small (%zu < %zu bytes)", bytes, p.total);
    SEUNET_CHECK((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "net: workspace must be 256-byte aligned");
    reg = build_registry(p.d);
    ws = reinterpret_cast<unsigned char*>(workspace);
    params = prm;
    s = st;
    for (size_t i = 0; i < reg.size(); ++i)
      SEUNET_CHECK(prm[i] != nullptr || reg[i].name.rfind("dc62", 0) == 0, "net: parameter %s is null", reg[i].name.c_str());
    return 0;
  }

  SrcList srcs(const OpDesc& o) const {
    SrcList l{};
    l.n = o.nsrc;
    for (int k = 0; k < o.nsrc; ++k) { l.ptr[k] = at(p.feat[o.src[k]]); l.C[k] = p.C[o.src[k]]; }
    return l;
  }

  // conv (+ InstanceNorm statistics) of one block: raw <- conv(src), (mean, rstd) <- stats(raw)
  int conv_and_stats(const std::string& nm, int taps, int dil, const SrcList& src, int cin, const float* w, const float* bias, size_t wp_off,
                     size_t raw_off, int cout, size_t mean_off, size_t rstd_off, const Dims& dm, bool stream = false, bool march = false) {
    DstList dst{};
    dst.n = 1; dst.ptr[0] = at(raw_off); dst.C[0] = cout; dst.acc[0] = 0;
    int slots;
    if (march) {
      mark("conv_fwd:" + nm);   // (weights were packed by pack_all_weights)
      if (int e = launch_conv_march(p.d.dtype, dil, src, at(wp_off), bias, dst, dat(p.stats), dm, s)) return e;
      slots = conv_march_slots(dm, dil, src.total(), cout);
    } else if (stream) {
      mark("conv_fwd:" + nm);   // (weights were packed by pack_all_weights)
      if (int e = launch_conv_stream(p.d.dtype, dil, src.ptr[0], src.C[0], at(wp_off), bias, at(raw_off), cout, 0, dat(p.stats), dm, s)) return e;
      slots = conv_stream_slots(dm, dil);
    } else if (p.d.conv_impl == SEUNET_CONV_NAIVE) {
      mark("conv_fwd:" + nm);
      if (int e = launch_conv_naive(p.d.dtype, taps, dil, src, cin, w, 0, bias, dst, dm, s)) return e;
      mark("stats");
      if (int e = launch_channel_stats(p.d.dtype, at(raw_off), cout, dat(p.stats), dm, s)) return e;
      slots = epi_partials(dm);
    } else {
      mark("conv_fwd:" + nm);   // (weights were packed by pack_all_weights)
      if (int e = launch_conv_igemm(p.d.dtype, taps, dil, src, cin, at(wp_off), bias, dst, dat(p.stats), dm, s)) return e;
      slots = conv_stats_tiles(dm, taps, dil);
    }
    mark("stats");
    return launch_stats_finalize(dat(p.stats), slots, cout, dm.N, dm.vox(), p.d.eps, 0, fat(mean_off), fat(rstd_off), s);
  }

  SseParams sse_params(const OpDesc& o) const {
    const std::string n = o.name;
    SseParams sp{};
    sp.w_se = P(n + ".conv_se.weight");
    sp.w_se2 = o.gates == 2 ? P(n + ".conv_se2.weight") : nullptr;
    sp.w_side = P(n + ".conv2.weight");
    sp.b_side = P(n + ".conv2.bias");
    sp.slope = p.d.negative_slope;
    return sp;
  }
  bool skip_enc_head = false;   // inference (prediction.py:102-103 discards pred0): the encoder head and its side maps are not evaluated

  SseHead sse_head(const OpDesc& o, const float* drop1, const float* drop2, bool first_of_level) const {
    SseHead h{};
    const int lv = kT[o.dst].level;
    h.side_out = nullptr;
    if (o.head == 0 && skip_enc_head) return h;     // (no level map: the epilogue skips the side conv altogether)
    if (p.d.n_classes > 1) {                        // general head path (classes.hip): the epilogue leaves the side map
      h.side_out = fat(p.op[&o - kOps].side);
      h.level_accumulate = first_of_level ? 0 : 1;    // (consumed by launch_side_to_level)
      return h;
    }
    h.level_map = fat(p.lvl[o.head][lv]);
    h.level_accumulate = first_of_level ? 0 : 1;
    h.head_w = (o.head == 0 ? P("dc0_0.weight") : P("dc0_1.weight")) + 2 * o.m;
    const float* dr = o.head == 0 ? drop1 : drop2;
    h.drop = dr ? dr + 2 * o.m : nullptr;
    h.drop_stride = o.head == 0 ? 24 : 12;
    return h;
  }

  // every conv weight of a pass repacked into its MFMA layout by one or two launches (the parameters change every step)
  int pack_all_weights(bool dgrad) {
    if (p.d.conv_impl == SEUNET_CONV_NAIVE) return 0;
    mark("pack_w");
    std::vector<ConvPackJob> jobs;
    std::vector<MarchPackJob> mjobs;
    std::vector<StreamPackJob> sjobs;
    for (int i = 0; i < kNumOps; ++i) {
      const OpDesc& o = kOps[i];
      if (o.kind != OP_GATED && o.kind != OP_CAT) continue;
      const OpRes& r = p.op[i];
      const std::string n = o.name;
      if ((!dgrad && r.stream_f) || (dgrad && r.stream_d)) {
        // PyTorch weight (cout, cin, 3,3,3); the data gradient reads it transposed / mirrored
        const int src_c = dgrad ? r.cout : p.C[o.src[0]], dst_c = dgrad ? p.C[o.src[0]] : r.cout;
        sjobs.push_back({P(n + ".conv1.weight"), at(dgrad ? r.wp_d : r.wp_f), r.cin, r.cout, dgrad ? 1 : 0, src_c, dst_c});
        continue;
      }
      if ((!dgrad && r.march_f) || (dgrad && r.march_d)) {
        int ctot = 0;
        for (int k = 0; k < o.nsrc; ++k) ctot += p.C[o.src[k]];
        mjobs.push_back({P(n + ".conv1.weight"), at(dgrad ? r.wp_d : r.wp_f), r.cin, r.cout, dgrad ? 1 : 0, dgrad ? r.cout : ctot,
                         dgrad ? ctot : r.cout});
        continue;
      }
      if (!dgrad) {
        jobs.push_back({P(n + ".conv1.weight"), at(r.wp_f), r.taps, r.cin, r.cout, 0});
        if (o.kind == OP_CAT && o.xname && !p.fuse_x) jobs.push_back({P(std::string(o.xname) + ".conv1.weight"), at(r.wp_x), 1, p.d.in_channel, r.cout, 0});
      } else if (r.need_dgrad) {
        jobs.push_back({P(n + ".conv1.weight"), at(r.wp_d), r.taps, r.cin, r.cout, 1});
      }
    }
    if (!sjobs.empty())
      if (int e = launch_conv_stream_pack_multi(p.d.dtype, sjobs.data(), (int)sjobs.size(), s)) return e;
    if (!mjobs.empty())
      if (int e = launch_conv_march_pack_multi(p.d.dtype, mjobs.data(), (int)mjobs.size(), s)) return e;
    return launch_conv_pack_weights_multi(p.d.dtype, jobs.data(), (int)jobs.size(), s);
  }

  int forward(const float* x, const float* drop1, const float* drop2, float* pred0, float* pred1) {
    skip_enc_head = pred0 == nullptr;
    if (int e = pack_all_weights(false)) return e;
    mark("pack_input");
    if (int e = launch_pack_input(p.d.dtype, x, p.d.in_channel, at(p.feat[T_X0]), p.dims[0], s)) return e;
    bool lvl_written[2][4] = {{false, false, false, false}, {false, false, false, false}};
    bool pool_done[kNumOps] = {};
    for (int i = 0; i < kNumOps; ++i) {
      const OpDesc& o = kOps[i];
      const OpRes& r = p.op[i];
      const std::string n = o.name;
      if (o.kind == OP_POOL) {
        if (pool_done[i]) continue;          // written by the aggregation block's epilogue (below)
        mark("pool_fwd:" + n);
        if (int e = launch_maxpool_fwd(p.d.dtype, at(p.feat[o.src[0]]), p.C[o.src[0]], at(p.feat[o.dst]), p.dims[kT[o.src[0]].level], s)) return e;
      } else if (o.kind == OP_UP) {
        mark("up_fwd:" + n);
        if (int e = launch_upsample2_fwd(p.d.dtype, at(p.feat[o.src[0]]), p.C[o.src[0]], at(p.feat[o.dst]), p.dims[kT[o.src[0]].level], s)) return e;
      } else if (o.kind == OP_GATED) {
        const int lv = kT[o.dst].level;
        if (int e = conv_and_stats(n, 27, o.dil, srcs(o), r.cin, P(n + ".conv1.weight"), P(n + ".conv1.bias"), r.wp_f, r.raw,
                                   r.cout, r.mean, r.rstd, p.dims[lv], r.stream_f, r.march_f)) return e;
        const SseHead hd = sse_head(o, drop1, drop2, !lvl_written[o.head][lv]);
        lvl_written[o.head][lv] = true;
        mark("epi_fwd:" + n);
        if (int e = launch_sse_fwd(p.d.dtype, at(r.raw), fat(r.mean), fat(r.rstd), r.cout, sse_params(o), at(p.feat[o.dst]), hd,
                                   p.dims[lv], s)) return e;
        if (p.d.n_classes > 1 && hd.side_out != nullptr) {
          const float* dr = o.head == 0 ? drop1 : drop2;
          if (int e = launch_side_to_level(fat(r.side), (o.head == 0 ? P("dc0_0.weight") : P("dc0_1.weight")) + 2 * o.m, o.head == 0 ? 24 : 12,
                                           dr ? dr + 2 * o.m : nullptr, o.head == 0 ? 24 : 12, p.d.n_classes, fat(p.lvl[o.head][lv]),
                                           hd.level_accumulate, p.dims[lv], s)) return e;
        }
      } else {  // OP_CAT
        const int lv = kT[o.dst].level;
        if (int e = conv_and_stats(n, 1, 1, srcs(o), r.cin, P(n + ".conv1.weight"), nullptr, r.wp_f, r.raw, r.cout, r.mean,
                                   r.rstd, p.dims[lv])) return e;
        if (o.xname && p.fuse_x) {
          // x-branch: statistics from the input's moments, values recomputed inside the epilogue (never stored)
          const float* w2 = P(std::string(o.xname) + ".conv1.weight");
          mark("stats");
          if (int e = launch_xbranch_moments(p.d.dtype, at(p.feat[o.xsrc]), dat(p.xmom), p.dims[lv], s)) return e;
          if (int e = launch_xbranch_stats(dat(p.xmom), xbranch_moment_slots(p.dims[lv]), w2, r.cout, p.d.in_channel, p.dims[lv].N,
                                           p.dims[lv].vox(), p.d.eps, fat(r.mean2), fat(r.rstd2), dat(r.xtot), s)) return e;
          mark("cat_fwd:" + n);
          // the max-pool that consumes this block (ec33 -> pool0, ec63 -> pool1, ec93 -> pool2) is written by the same kernel
          const bool pool_next = i + 1 < kNumOps && kOps[i + 1].kind == OP_POOL && kOps[i + 1].src[0] == o.dst &&
                                 getenv("SEUNET_NO_POOL_FUSE") == nullptr;
          if (pool_next) {
            if (int e = launch_cat_fwd_x_pool(p.d.dtype, at(r.raw), fat(r.mean), fat(r.rstd), at(p.feat[o.xsrc]), w2, p.d.in_channel,
                                              fat(r.mean2), fat(r.rstd2), r.cout, p.d.negative_slope, at(p.feat[o.dst]),
                                              at(p.feat[kOps[i + 1].dst]), p.dims[lv], s,
                                              p.op[i + 1].has_pool_idx ? reinterpret_cast<unsigned*>(at(p.op[i + 1].pool_idx)) : nullptr)) return e;
            pool_done[i + 1] = true;
          } else if (int e = launch_cat_fwd_x(p.d.dtype, at(r.raw), fat(r.mean), fat(r.rstd), at(p.feat[o.xsrc]), w2, p.d.in_channel,
                                       fat(r.mean2), fat(r.rstd2), r.cout, p.d.negative_slope, at(p.feat[o.dst]), p.dims[lv], s)) return e;
        } else {
          if (o.xname) {
            SrcList xs{};
            xs.n = 1; xs.ptr[0] = at(p.feat[o.xsrc]); xs.C[0] = 8;
            if (int e = conv_and_stats(o.xname, 1, 1, xs, p.d.in_channel, P(std::string(o.xname) + ".conv1.weight"), nullptr, r.wp_x,
                                       r.raw2, r.cout, r.mean2, r.rstd2, p.dims[lv])) return e;
          }
          mark("cat_fwd:" + n);
          if (int e = launch_cat_fwd(p.d.dtype, at(r.raw), fat(r.mean), fat(r.rstd), o.xname ? at(r.raw2) : nullptr,
                                     o.xname ? fat(r.mean2) : nullptr, o.xname ? fat(r.rstd2) : nullptr, r.cout,
                                     p.d.negative_slope, at(p.feat[o.dst]), p.dims[lv], s)) return e;
        }
      }
    }
    const float* enc[4] = {fat(p.lvl[0][0]), fat(p.lvl[0][1]), fat(p.lvl[0][2]), fat(p.lvl[0][3])};
    const float* dec[3] = {fat(p.lvl[1][0]), fat(p.lvl[1][1]), fat(p.lvl[1][2])};
    mark("head_fwd");
    if (p.d.n_classes > 1) {       // once per (sample, class): level maps [class][N][V], logits (N, K, D, H, W)
      const int K = p.d.n_classes, N = p.d.batch;
      const Dims d1{1, p.dims[0].D, p.dims[0].H, p.dims[0].W};
      for (int hd = 0; hd < 2; ++hd) {
        float* pred = hd == 0 ? pred0 : pred1;
        if (pred == nullptr) continue;
        const int nl = hd == 0 ? 4 : 3;
        for (int n = 0; n < N; ++n)
          for (int c = 0; c < K; ++c) {
            const float* lm[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int l = 0; l < nl; ++l) lm[l] = fat(p.lvl[hd][l]) + ((size_t)c * N + n) * p.dims[l].vox();
            if (int e = launch_head_fwd(lm, nl, (hd == 0 ? P("dc0_0.bias") : P("dc0_1.bias")) + c, pred + ((size_t)n * K + c) * d1.vox(), d1, s)) return e;
          }
      }
      mark("outside");
      return 0;
    }
    if (!skip_enc_head)
      if (int e = launch_head_fwd(enc, 4, P("dc0_0.bias"), pred0, p.dims[0], s)) return e;
    if (int e = launch_head_fwd(dec, 3, P("dc0_1.bias"), pred1, p.dims[0], s)) return e;
    mark("outside");
    return 0;
  }

  // gradient w.r.t. the raw conv output is in grad[dst]; produce weight gradient and input gradients
  int conv_backward(int i, const SrcList& x, float* const* grads, bool* written) {
    const OpDesc& o = kOps[i];
    const OpRes& r = p.op[i];
    const int lv = kT[o.dst].level;
    const std::string n = o.name;
    const int wi = find_param(reg, n + ".conv1.weight");
    if (grads[wi]) {
      mark("wgrad:" + n);
      if (r.stream_w) {
        if (int e = launch_wgrad_stream(p.d.dtype, o.dil, x.ptr[0], x.C[0], r.cin, at(p.grad[o.dst]), r.cout, r.cout, grads[wi],
                                        at(p.wgrad_ws), p.wgrad_ws_bytes, p.dims[lv], s)) return e;
      } else if (p.d.conv_impl == SEUNET_CONV_NAIVE) {
        if (int e = launch_wgrad_naive(p.d.dtype, r.taps, o.dil, x, r.cin, at(p.grad[o.dst]), r.cout, grads[wi], p.dims[lv], s)) return e;
      } else {
        if (int e = launch_wgrad(p.d.dtype, r.taps, o.dil, x, r.cin, at(p.grad[o.dst]), r.cout, grads[wi], at(p.wgrad_ws),
                                 p.wgrad_ws_bytes, p.dims[lv], s)) return e;
      }
    }
    if (!r.need_dgrad) return 0;
    SrcList gsrc{};
    gsrc.n = 1; gsrc.ptr[0] = at(p.grad[o.dst]); gsrc.C[0] = r.cout;
    DstList gd{};
    gd.n = o.nsrc;
    for (int k = 0; k < o.nsrc; ++k) {
      const int t = o.src[k];
      gd.C[k] = p.C[t];
      gd.ptr[k] = is_input(t) ? nullptr : at(p.grad[t]);
      gd.acc[k] = (!is_input(t) && written[t]) ? 1 : 0;
      if (!is_input(t)) written[t] = true;
    }
    const float* w = P(n + ".conv1.weight");
    mark("dgrad:" + n);
    if (r.stream_d)
      return launch_conv_stream(p.d.dtype, o.dil, at(p.grad[o.dst]), r.cout, at(r.wp_d), nullptr, gd.ptr[0], gd.C[0], gd.acc[0], nullptr,
                                p.dims[lv], s);
    if (r.march_d) return launch_conv_march(p.d.dtype, o.dil, gsrc, at(r.wp_d), nullptr, gd, nullptr, p.dims[lv], s);
    if (p.d.conv_impl == SEUNET_CONV_NAIVE)
      return launch_conv_naive(p.d.dtype, r.taps, o.dil, gsrc, r.cout, w, 1, nullptr, gd, p.dims[lv], s);
    return launch_conv_igemm(p.d.dtype, r.taps, o.dil, gsrc, r.cout, at(r.wp_d), nullptr, gd, nullptr, p.dims[lv], s);
  }

  hipEvent_t decoder_done = nullptr;   // recorded once every gradient of the decoder blocks (dc1 .. dc6, dc22, dc42) is final

  int backward(const float* g_pred0, const float* g_pred1, const float* drop1, const float* drop2, float* const* grads) {
    if (int e = pack_all_weights(true)) return e;
    bool written[T_COUNT];
    for (int t = 0; t < T_COUNT; ++t) written[t] = false;
    // heads: level gradients = transposed interpolation of the logit gradients
    {
      mark("head_bwd");
      float* ge[4] = {nullptr, fat(p.glvl[0][1]), fat(p.glvl[0][2]), fat(p.glvl[0][3])};
      float* gd[4] = {nullptr, fat(p.glvl[1][1]), fat(p.glvl[1][2]), nullptr};
      if (p.d.n_classes > 1) {     // once per (sample, class); the bias gradient of a class = its per-sample sums added up
        const int K = p.d.n_classes, N = p.d.batch;
        const Dims d1{1, p.dims[0].D, p.dims[0].H, p.dims[0].W};
        for (int hd = 0; hd < 2; ++hd) {
          const float* gp = hd == 0 ? g_pred0 : g_pred1;
          const int nl = hd == 0 ? 4 : 3;
          for (int n = 0; n < N; ++n)
            for (int c = 0; c < K; ++c) {
              float* gl[4] = {nullptr, nullptr, nullptr, nullptr};
              for (int l = 1; l < nl; ++l) gl[l] = fat(p.glvl[hd][l]) + ((size_t)c * N + n) * p.dims[l].vox();
              if (int e = launch_head_bwd(gp + ((size_t)n * K + c) * d1.vox(), gl, nl, fat(p.head_tmp), fat(p.cls_bias) + n * K + c, d1, s)) return e;
            }
          if (float* gb = grads[find_param(reg, hd == 0 ? "dc0_0.bias" : "dc0_1.bias")])
            if (int e = launch_class_bias_grad(fat(p.cls_bias), N, K, gb, s)) return e;
        }
      } else {
      if (int e = launch_head_bwd(g_pred0, ge, 4, fat(p.head_tmp), grads[find_param(reg, "dc0_0.bias")], p.dims[0], s)) return e;
      if (int e = launch_head_bwd(g_pred1, gd, 3, fat(p.head_tmp), grads[find_param(reg, "dc0_1.bias")], p.dims[0], s)) return e;
      }
    }
    std::vector<float*> zero_ptrs;
    std::vector<int> zero_counts;
    const unsigned* pool_am[T_COUNT] = {};     // per tensor: a max-pool gradient still to be added by its producer's backward
    const void* pool_g[T_COUNT] = {};
    for (int i = kNumOps - 1; i >= 0; --i) {
      const OpDesc& o = kOps[i];
      const OpRes& r = p.op[i];
      const std::string n = o.name;
      if (decoder_done && std::string(o.name) == "up0") {
        // the walk is in reverse forward order: everything after up0 (the decoder) has been differentiated.  Its parameter
        // gradients are final from here on (the heads' weights are not: the encoder blocks still add to dc0_0)
        SEUNET_HIP(hipEventRecord(decoder_done, s));
      }
      if (o.kind == OP_POOL || o.kind == OP_UP) {
        const int t = o.src[0];
        if (is_input(t)) continue;
        SEUNET_CHECK(written[o.dst], "net: internal: gradient of %s output missing", o.name);
        mark((o.kind == OP_POOL ? "pool_bwd:" : "up_bwd:") + n);
        if (o.kind == OP_POOL) {
          // (the index is valid when the fused forward wrote it: same condition as in the forward walk)
          static const bool no_fuse = getenv("SEUNET_NO_POOL_FUSE") != nullptr, no_defer = getenv("SEUNET_NO_POOL_DEFER") != nullptr;
          if (r.has_pool_idx && !no_fuse && !no_defer && written[t] && p.fuse_x) {
            // nothing to launch: the aggregation block that produced tensor t adds this gradient on the fly in both of its
            // backward passes (launch_cat_bwd_x, pool_* arguments) -- no read-modify-write of the full-resolution gradient
            pool_am[t] = reinterpret_cast<const unsigned*>(at(r.pool_idx));
            pool_g[t] = at(p.grad[o.dst]);
            continue;
          }
          if (r.has_pool_idx && !no_fuse) {
            if (int e = launch_maxpool_bwd_idx(p.d.dtype, reinterpret_cast<const unsigned*>(at(r.pool_idx)), at(p.grad[o.dst]), p.C[t],
                                               at(p.grad[t]), written[t] ? 1 : 0, p.dims[kT[t].level], s)) return e;
          } else if (int e = launch_maxpool_bwd(p.d.dtype, at(p.feat[t]), at(p.grad[o.dst]), p.C[t], at(p.grad[t]), written[t] ? 1 : 0,
                                                p.dims[kT[t].level], s)) return e;
        } else {
          if (int e = launch_upsample2_bwd(p.d.dtype, at(p.grad[o.dst]), p.C[t], at(p.grad[t]), written[t] ? 1 : 0,
                                           p.dims[kT[t].level], s)) return e;
        }
        written[t] = true;
        continue;
      }
      const int lv = kT[o.dst].level;
      const Dims& dm = p.dims[lv];
      const int P_slots = epi_partials(dm);
      if (o.kind == OP_GATED) {
        SseBwdIn g{};
        g.g_e = written[o.dst] ? at(p.grad[o.dst]) : nullptr;
        g.g_side = nullptr;
        g.g_level = lv == 0 ? (o.head == 0 ? g_pred0 : g_pred1) : fat(p.glvl[o.head][lv]);
        SseHead hd = sse_head(o, drop1, drop2, false);
        float* g_head = grads[find_param(reg, o.head == 0 ? "dc0_0.weight" : "dc0_1.weight")];
        if (p.d.n_classes > 1) {
          // general head path: the K level-map gradients of this level folded into the gradient of the block's side map (and the
          // head-weight gradient summed on the way); the block's passes then take g_side
          const int K = p.d.n_classes;
          const float* dr = o.head == 0 ? drop1 : drop2;
          const long long V = dm.vox();
          const float* glev = g.g_level;
          const long long cstride = lv == 0 ? V : (long long)dm.N * V, nstride = lv == 0 ? (long long)K * V : V;
          mark("epi_bwd:" + n);
          if (int e = launch_level_to_side_grad(glev, cstride, nstride, fat(r.side), (o.head == 0 ? P("dc0_0.weight") : P("dc0_1.weight")) + 2 * o.m,
                                                o.head == 0 ? 24 : 12, dr ? dr + 2 * o.m : nullptr, o.head == 0 ? 24 : 12, K, fat(p.gside),
                                                dat(p.cls_part), g_head ? g_head + 2 * o.m : nullptr, dm, s)) return e;
          g.g_level = nullptr;
          g.g_side = fat(p.gside);
          hd.side_out = nullptr;
          g_head = nullptr;          // (written above; the finalize kernel must not overwrite it)
        }
        mark("epi_bwd:" + n);   // pass A: f64 sums + parameter-gradient records
        if (int e = launch_sse_bwd(p.d.dtype, at(r.raw), fat(r.mean), fat(r.rstd), r.cout, sse_params(o), g, hd, nullptr, nullptr,
                                   nullptr, dat(p.stats), fat(p.pgrad), dm, s)) return e;
        mark("stats");
        const int i_se2 = o.gates == 2 ? find_param(reg, n + ".conv_se2.weight") : -1;
        if (int e = launch_gate_bwd_finalize(dat(p.stats), P_slots, r.cout, dm.N, dm.vox(), fat(p.m1), fat(p.m2), fat(p.pgrad),
                                             dm.N * P_slots, grads[find_param(reg, n + ".conv_se.weight")],
                                             i_se2 >= 0 ? grads[i_se2] : nullptr, grads[find_param(reg, n + ".conv2.weight")],
                                             grads[find_param(reg, n + ".conv2.bias")], g_head ? g_head + 2 * o.m : nullptr, s)) return e;
        mark("in_bwd:" + n);    // pass B: recompute dxhat, apply the InstanceNorm backward, store draw over g_e
        if (int e = launch_sse_bwd(p.d.dtype, at(r.raw), fat(r.mean), fat(r.rstd), r.cout, sse_params(o), g, hd, fat(p.m1), fat(p.m2),
                                   at(p.grad[o.dst]), nullptr, nullptr, dm, s)) return e;
        written[o.dst] = true;
        // conv1.bias feeds an affine-less InstanceNorm: its gradient is identically zero; zeroed in one
        // launch after the loop
        if (float* gb = grads[find_param(reg, n + ".conv1.bias")]) { zero_ptrs.push_back(gb); zero_counts.push_back(r.cout); }
        if (int e = conv_backward(i, srcs(o), grads, written)) return e;
      } else {  // OP_CAT
        SEUNET_CHECK(written[o.dst], "net: internal: gradient of %s output missing", o.name);
        mark("cat_bwd:" + n);   // pass A
        const float* mu2 = o.xname ? fat(r.mean2) : nullptr;
        const float* rs2 = o.xname ? fat(r.rstd2) : nullptr;
        const int xi = o.xname ? find_param(reg, std::string(o.xname) + ".conv1.weight") : -1;
        if (o.xname && p.fuse_x) {
          // x-branch recomputed from the input in both passes; pass A also sums what its weight gradient is formed from
          const float* w2 = P(std::string(o.xname) + ".conv1.weight");
          const void* xin = at(p.feat[o.xsrc]);
          if (int e = launch_cat_bwd_x(p.d.dtype, at(p.grad[o.dst]), at(r.raw), fat(r.mean), fat(r.rstd), xin, w2, p.d.in_channel, mu2, rs2,
                                       r.cout, p.d.negative_slope, nullptr, nullptr, nullptr, nullptr, nullptr, dat(p.stats), dat(p.stats2),
                                       grads[xi] ? dat(p.xwp) : nullptr, dm, s, pool_am[o.dst], pool_g[o.dst])) return e;
          mark("stats");
          if (grads[xi])
            if (int e = launch_cat_xgrad_finalize(dat(p.xwp), dat(p.stats2), P_slots, dat(r.xtot), w2, r.cout, p.d.in_channel, dm.N,
                                                  p.d.eps, grads[xi], s)) return e;
          if (int e = launch_stats_finalize(dat(p.stats), P_slots, r.cout, dm.N, dm.vox(), 0.f, 1, fat(p.m1), fat(p.m2), s)) return e;
          if (int e = launch_stats_finalize(dat(p.stats2), P_slots, r.cout, dm.N, dm.vox(), 0.f, 1, fat(p.m1b), fat(p.m2b), s)) return e;
          mark("in_bwd:" + n);    // pass B
          if (int e = launch_cat_bwd_x(p.d.dtype, at(p.grad[o.dst]), at(r.raw), fat(r.mean), fat(r.rstd), xin, w2, p.d.in_channel, mu2, rs2,
                                       r.cout, p.d.negative_slope, fat(p.m1), fat(p.m2), fat(p.m1b), fat(p.m2b), at(p.grad[o.dst]), nullptr,
                                       nullptr, nullptr, dm, s, pool_am[o.dst], pool_g[o.dst])) return e;
        } else {
          const void* r2 = o.xname ? at(r.raw2) : nullptr;
          if (int e = launch_cat_bwd(p.d.dtype, at(p.grad[o.dst]), at(r.raw), fat(r.mean), fat(r.rstd), r2, mu2, rs2, r.cout,
                                     p.d.negative_slope, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, dat(p.stats),
                                     dat(p.stats2), dm, s)) return e;
          mark("stats");
          if (int e = launch_stats_finalize(dat(p.stats), P_slots, r.cout, dm.N, dm.vox(), 0.f, 1, fat(p.m1), fat(p.m2), s)) return e;
          if (o.xname)
            if (int e = launch_stats_finalize(dat(p.stats2), P_slots, r.cout, dm.N, dm.vox(), 0.f, 1, fat(p.m1b), fat(p.m2b), s)) return e;
          mark("in_bwd:" + n);    // pass B
          if (int e = launch_cat_bwd(p.d.dtype, at(p.grad[o.dst]), at(r.raw), fat(r.mean), fat(r.rstd), r2, mu2, rs2, r.cout,
                                     p.d.negative_slope, fat(p.m1), fat(p.m2), o.xname ? fat(p.m1b) : nullptr,
                                     o.xname ? fat(p.m2b) : nullptr, at(p.grad[o.dst]), o.xname ? at(p.gx) : nullptr, nullptr, nullptr,
                                     dm, s)) return e;
          if (o.xname && grads[xi]) {
            mark(std::string("wgrad:") + o.xname);
            SrcList xs{};
            xs.n = 1; xs.ptr[0] = at(p.feat[o.xsrc]); xs.C[0] = 8;
            if (p.d.conv_impl == SEUNET_CONV_NAIVE) {
              if (int e = launch_wgrad_naive(p.d.dtype, 1, 1, xs, p.d.in_channel, at(p.gx), r.cout, grads[xi], dm, s)) return e;
            } else {
              if (int e = launch_wgrad(p.d.dtype, 1, 1, xs, p.d.in_channel, at(p.gx), r.cout, grads[xi], at(p.wgrad_ws),
                                       p.wgrad_ws_bytes, dm, s)) return e;
            }
          }
        }
        if (int e = conv_backward(i, srcs(o), grads, written)) return e;
      }
    }
    if (!zero_ptrs.empty()) {
      mark("stats");
      if (int e = launch_multi_zero(zero_ptrs.data(), zero_counts.data(), (int)zero_ptrs.size(), s)) return e;
    }
    mark("outside");
    return 0;
  }
};

}  // namespace
}  // namespace seunet

using namespace seunet;

extern "C" {

int seunet_net_param_count(const seunet_net_desc* desc) {
  if (!desc) return -1;
  return (int)build_registry(*desc).size();
}

int seunet_net_param_info(const seunet_net_desc* desc, int index, char* name, int name_cap, int* shape5, int* ndim) {
  SEUNET_CHECK(desc && name && shape5 && ndim, "param_info: null argument");
  const std::vector<ParamInfo> reg = build_registry(*desc);
  SEUNET_CHECK(index >= 0 && index < (int)reg.size(), "param_info: index %d out of range", index);
  snprintf(name, (size_t)name_cap, "%s", reg[index].name.c_str());
  for (int k = 0; k < 5; ++k) shape5[k] = reg[index].shape[k];
  *ndim = reg[index].ndim;
  return 0;
}

size_t seunet_net_workspace_bytes(const seunet_net_desc* desc) {
  if (!desc) return 0;
  Plan p;
  if (p.init(*desc)) return 0;
  return p.total;
}

int seunet_net_forward(const seunet_net_desc* desc, const float* const* params, const float* x, const float* drop1,
                       const float* drop2, float* pred0, float* pred1, void* workspace, size_t workspace_bytes,
                       seunet_stream_t s) {
  SEUNET_CHECK(x && pred1, "net_forward: null tensor");
  Exec ex;
  if (int e = ex.setup(desc, params, workspace, workspace_bytes, (hipStream_t)s)) return e;
  return ex.forward(x, drop1, drop2, pred0, pred1);
}

// ---- the forward pass as a HIP graph (inference loops call the same forward on the same buffers hundreds of times:
// one graph launch replaces ~150 kernel launches, and the dependent-launch gaps between the small kernels of the coarse
// levels shrink) ----------------------------------------------------------------------------------------------------
struct NetGraph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

int seunet_net_forward_capture(const seunet_net_desc* desc, const float* const* params, const float* x, const float* drop1,
                               const float* drop2, float* pred0, float* pred1, void* workspace, size_t workspace_bytes,
                               seunet_stream_t s, void** graph_out) {
  SEUNET_CHECK(x && pred1 && graph_out, "net_forward_capture: null argument");
  SEUNET_CHECK(s != nullptr, "net_forward_capture: stream capture needs a stream other than the null stream");
  SEUNET_CHECK(!prof_on(), "net_forward_capture: switch the launch-group timer off before capturing");
  *graph_out = nullptr;
  Exec ex;
  if (int e = ex.setup(desc, params, workspace, workspace_bytes, (hipStream_t)s)) return e;
  SEUNET_CHECK(device_zero_page() != nullptr, "net_forward_capture: zero page allocation failed");   // (not inside the capture)
  SEUNET_HIP(hipStreamBeginCapture((hipStream_t)s, hipStreamCaptureModeThreadLocal));
  const int rc = ex.forward(x, drop1, drop2, pred0, pred1);
  NetGraph* g = new NetGraph;
  const hipError_t e_end = hipStreamEndCapture((hipStream_t)s, &g->graph);
  if (rc != 0 || e_end != hipSuccess) {
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    if (rc != 0) return rc;    // (the message of the failing launch is already recorded)
    return fail("net_forward_capture: hipStreamEndCapture failed: %s", hipGetErrorString(e_end));
  }
  const hipError_t e_inst = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
  if (e_inst != hipSuccess) {
    (void)hipGraphDestroy(g->graph);
    delete g;
    return fail("net_forward_capture: hipGraphInstantiate failed: %s", hipGetErrorString(e_inst));
  }
  *graph_out = g;
  return 0;
}

int seunet_graph_launch(void* graph, seunet_stream_t s) {
  SEUNET_CHECK(graph != nullptr, "graph_launch: null graph");
  SEUNET_HIP(hipGraphLaunch(static_cast<NetGraph*>(graph)->exec, (hipStream_t)s));
  return 0;
}

int seunet_graph_destroy(void* graph) {
  if (!graph) return 0;
  NetGraph* g = static_cast<NetGraph*>(graph);
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
  return 0;
}

// Diagnostic read-back of one intermediate of the LAST forward that ran on `workspace` (nothing is recomputed): which = 0 the
// raw conv output of block `name` (before InstanceNorm; NCDHW f32, `channels` of them), 1 / 2 its per-(n, c) mean / rstd
// ([N][C] f32), 3 the block's output tensor (for an aggregation block: after the x-branch was added; NCDHW f32).  Used by the
// flip census (tests/flip_census.py): LeakyReLU sign / max-pool argmax disagreements with the float64 oracle.
int seunet_net_read_tensor(const seunet_net_desc* desc, const float* const* params, const void* workspace, size_t workspace_bytes,
                           const char* name, int which, float* out, int* channels, seunet_stream_t s) {
  SEUNET_CHECK(desc && workspace && name && out, "net_read_tensor: null argument");
  Plan p;
  if (int e = p.init(*desc)) return e;
  SEUNET_CHECK(workspace_bytes >= p.total, "net_read_tensor: workspace too small");
  const unsigned char* ws = reinterpret_cast<const unsigned char*>(workspace);
  for (int i = 0; i < kNumOps; ++i) {
    const OpDesc& o = kOps[i];
    if (o.kind != OP_GATED && o.kind != OP_CAT) continue;
    const OpRes& r = p.op[i];
    const int lv = kT[o.dst].level;
    if (o.xname && std::string(o.xname) == name) {   // the raw-input branch of an aggregation block, when it is materialised (in_channel > 2)
      SEUNET_CHECK(which >= 0 && which <= 2, "net_read_tensor: which=%d is not stored for an x-branch", which);
      if (channels) *channels = r.cout;
      if (which == 0 && p.fuse_x) {
        // in_channel <= 2: the branch is recomputed inside the aggregation epilogue and leaves no tensor: recompute it here by the
        // same device function, from the packed input in the workspace and the caller's weights
        SEUNET_CHECK(params != nullptr, "net_read_tensor: the recomputed x-branch %s needs the parameter list", name);
        const std::vector<ParamInfo> reg = build_registry(p.d);
        const int wi = find_param(reg, std::string(name) + ".conv1.weight");
        SEUNET_CHECK(wi >= 0 && params[wi], "net_read_tensor: no weight for %s", name);
        return launch_xbranch_values(p.d.dtype, ws + p.feat[o.xsrc], params[wi], r.cout, p.d.in_channel, out, p.dims[lv], (hipStream_t)s);
      }
      if (which == 0) return launch_unpack_cl(p.d.dtype, ws + r.raw2, r.cout, out, p.dims[lv], (hipStream_t)s);
      SEUNET_HIP(hipMemcpyAsync(out, ws + (which == 1 ? r.mean2 : r.rstd2), (size_t)p.d.batch * r.cout * 4, hipMemcpyDeviceToDevice, (hipStream_t)s));
      return 0;
    }
    if (std::string(o.name) != name) continue;
    if (channels) *channels = which == 3 ? p.C[o.dst] : r.cout;
    if (which == 0) return launch_unpack_cl(p.d.dtype, ws + r.raw, r.cout, out, p.dims[lv], (hipStream_t)s);
    if (which == 3) return launch_unpack_cl(p.d.dtype, ws + p.feat[o.dst], p.C[o.dst], out, p.dims[lv], (hipStream_t)s);
    SEUNET_CHECK(which == 1 || which == 2, "net_read_tensor: which=%d", which);
    SEUNET_HIP(hipMemcpyAsync(out, ws + (which == 1 ? r.mean : r.rstd), (size_t)p.d.batch * r.cout * 4, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return 0;
  }
  return fail("net_read_tensor: no block named %s", name);
}

int seunet_net_backward(const seunet_net_desc* desc, const float* const* params, const float* g_pred0,
                        const float* g_pred1, const float* drop1, const float* drop2, float* const* grads,
                        void* workspace, size_t workspace_bytes, seunet_stream_t s) {
  return seunet_net_backward_ev(desc, params, g_pred0, g_pred1, drop1, drop2, grads, workspace, workspace_bytes, s, nullptr);
}

int seunet_net_backward_ev(const seunet_net_desc* desc, const float* const* params, const float* g_pred0,
                           const float* g_pred1, const float* drop1, const float* drop2, float* const* grads,
                           void* workspace, size_t workspace_bytes, seunet_stream_t s, void* decoder_done_event) {
  SEUNET_CHECK(g_pred0 && g_pred1 && grads, "net_backward: null tensor");
  Exec ex;
  if (int e = ex.setup(desc, params, workspace, workspace_bytes, (hipStream_t)s)) return e;
  ex.decoder_done = reinterpret_cast<hipEvent_t>(decoder_done_event);
  return ex.backward(g_pred0, g_pred1, drop1, drop2, grads);
}

}  // extern "C"
