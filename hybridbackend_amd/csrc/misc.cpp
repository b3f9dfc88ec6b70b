// Error channel and version string of libhbk_core.so.
#include "common.h"

namespace hbk {
namespace {
thread_local char g_last_error[1024] = "";
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace hbk

extern "C" const char* hbk_last_error(void) { return hbk::g_last_error; }

extern "C" const char* hbk_version(void) { return "hbk 0.1.0 gfx950"; }

// Host-side evaluation of the device's divide-free floor-mod / floor-div (common.h), so the
// magic-number arithmetic can be checked exhaustively without a GPU.
extern "C" int64_t hbk_host_floormod_i64(int64_t v, int64_t d) {
  if (d <= 0) return -1;
  hbk::FastDiv f = hbk::make_fastdiv((uint64_t)d);
  f.d = (uint64_t)d;
  return (int64_t)hbk::floormod_i64(v, f);
}

extern "C" uint64_t hbk_host_fastdiv_u64(uint64_t n, uint64_t d) {
  if (d == 0) return 0;
  hbk::FastDiv f = hbk::make_fastdiv(d);
  return hbk::fastdiv(n, f);
}
