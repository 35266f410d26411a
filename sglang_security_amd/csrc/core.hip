// Error reporting, device queries.
#include "common.h"

namespace sglk {

static thread_local char g_err[512] = "";

char* err_buf() { return g_err; }

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace sglk

extern "C" {

const char* sglk_last_error(void) { return sglk::err_buf(); }

int sglk_version(void) { return 100; }

int sglk_device_cu_count(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return sglk::fail(SGLK_EHIP, "hipGetDevice failed");
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return sglk::fail(SGLK_EHIP, "hipDeviceGetAttribute failed");
  return n;
}
}
