// Library-level entry points of libzshmc.so: error reporting, version,
// device query.  See include/zshmc.h.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace zshmc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int device_cu_count() {
  static thread_local int cached_dev = -1;
  static thread_local int cached_cu = 256;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return cached_cu;
  if (dev != cached_dev) {
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) ==
            hipSuccess &&
        cu > 0)
      cached_cu = cu;
    cached_dev = dev;
  }
  return cached_cu;
}

}  // namespace zshmc

extern "C" const char* zshmc_last_error(void) { return zshmc::g_err; }
extern "C" int zshmc_version(void) { return ZSHMC_VERSION; }
