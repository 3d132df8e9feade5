// Error plumbing for the C ABI: a thread-local message, integer status codes, no exceptions.
#include "seunet_common.h"
#include <cstdarg>
#include <cstdio>

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
}  // namespace seunet
