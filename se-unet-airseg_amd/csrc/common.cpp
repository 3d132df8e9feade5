// Error plumbing for the C ABI: a thread-local message, integer status codes, no exceptions.
#include "seunet_common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace seunet {
static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
const char* get_error() { return g_last_error.c_str(); }

int fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return 1;
}

// ---- opt-in per-launch-group timing with HIP events on the caller's stream (bench.py's roofline) ------
struct Prof {
  bool on = false;
  std::string filter;        // non-empty: only launch groups whose tag contains it are timed (two events per group)
  bool prev_match = false;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  std::vector<std::pair<std::string, size_t>> marks;
};
// process-wide on purpose: autograd runs the backward call on another thread than the forward call.
// Only the benchmark enables it, from one thread, around stream-ordered calls.
static Prof g_prof;

bool prof_on() { return g_prof.on; }

void prof_mark(const char* tag, hipStream_t s) {
  Prof& p = g_prof;
  if (!p.on) return;
  const char* rec_tag = tag;
  if (!p.filter.empty()) {   // record only the start of a matching group and the mark that ends it
    const bool m = strstr(tag, p.filter.c_str()) != nullptr;
    const bool ends_previous = p.prev_match;
    p.prev_match = m;
    if (!m && !ends_previous) return;
    if (!m) rec_tag = "(untimed)";
  }
  tag = rec_tag;
  if (p.used == p.pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    p.pool.push_back(e);
  }
  (void)hipEventRecord(p.pool[p.used], s);
  p.marks.emplace_back(tag, p.used);
  ++p.used;
}
// hipFuncSetAttribute is per device: each kernel instantiation keeps a bit mask of the devices it was configured on
// (one process per GPU is the normal case; a process driving several devices, e.g. nn.DataParallel, still works).
// Raise a kernel's dynamic-LDS limit once per (instantiation, device).  The attribute call runs under the lock and the
// device's bit is set only when it succeeded: a second host thread can neither launch before the limit is in place nor skip
// a call that failed.
int configure_kernel_lds(unsigned long long& mask, const void* fn, int bytes) {
  static std::mutex mu;
  int dev = 0;
  SEUNET_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
  if (bit && (mask & bit)) return 0;
  SEUNET_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  mask |= bit;
  return 0;
}

// ---- a per-device page of zeros: source of padding voxels / channels for the weight-gradient LDS-DMA -------------
// (one hipMalloc + hipMemset per device for the life of the process, instead of a fill kernel in every launch)
const void* device_zero_page() {
  static std::mutex mu;
  static void* page[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (page[dev] == nullptr) {
    void* p = nullptr;
    if (hipMalloc(&p, 4096) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 4096) != hipSuccess) { (void)hipFree(p); return nullptr; }
    page[dev] = p;
  }
  return page[dev];
}
}  // namespace seunet

extern "C" int seunet_prof_enable(int on) {
  seunet::g_prof.on = on != 0;
  seunet::g_prof.used = 0;
  seunet::g_prof.marks.clear();
  seunet::g_prof.filter.clear();
  seunet::g_prof.prev_match = false;
  return 0;
}

// Like seunet_prof_enable(1), but only launch groups whose tag contains `substr` are timed: a HIP event per launch group
// costs ~1.5 us of stream time, ~5 % of a training step when all ~350 groups of a step are marked.  bench.py times the
// step with the dominant-kernel candidates marked only, and fills its full kernel table from extra, untimed steps.
extern "C" int seunet_prof_enable_filtered(const char* substr) {
  seunet_prof_enable(1);
  seunet::g_prof.filter = substr ? substr : "";
  return 0;
}

// Writes "tag\tmilliseconds\tcount\n" lines (time from each mark to the next one, summed per tag) and
// resets the recorder.  Synchronises on the recorded events (call it outside the timed region).
extern "C" int seunet_prof_report(char* buf, size_t cap) {
  using namespace seunet;
  Prof& p = g_prof;
  std::map<std::string, std::pair<double, long>> agg;
  if (!p.marks.empty()) {
    if (hipEventSynchronize(p.pool[p.marks.back().second]) != hipSuccess) return fail("prof: event sync failed");
    for (size_t i = 0; i + 1 < p.marks.size(); ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.pool[p.marks[i].second], p.pool[p.marks[i + 1].second]) != hipSuccess) continue;
      auto& a = agg[p.marks[i].first];
      a.first += ms;
      a.second += 1;
    }
  }
  std::string out;
  char line[256];
  for (auto& kv : agg) {
    snprintf(line, sizeof(line), "%s\t%.6f\t%ld\n", kv.first.c_str(), kv.second.first, kv.second.second);
    out += line;
  }
  p.used = 0;
  p.marks.clear();
  if (buf && cap) snprintf(buf, cap, "%s", out.c_str());
  return 0;
}
