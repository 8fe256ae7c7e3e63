// Library-level entry points of libzshmc.so: error reporting, version,
// device query.  See include/zshmc.h.
#include <stdarg.h>
#include <string.h>

#include "common.h"
#include "philox.h"

namespace zshmc {

static thread_local char g_err[512] = "";

static thread_local int tl_keep_parts = 0;
KeepSplitParts::KeepSplitParts() { ++tl_keep_parts; }
KeepSplitParts::~KeepSplitParts() { --tl_keep_parts; }
bool KeepSplitParts::active() { return tl_keep_parts > 0; }

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
extern "C" int zshmc_philox_rounds(void) { return ZS_PHILOX_ROUNDS; }

// hipMemsetAsync(0) on the caller's stream: the per-chain accumulators of the
// transition (kinetic energies) are cleared without an ATen fill kernel.
extern "C" int zshmc_zero(void* ptr, int64_t n_bytes, void* stream) {
  ZS_REQUIRE(ptr && n_bytes >= 0, "zshmc_zero: bad argument");
  if (n_bytes == 0) return ZSHMC_OK;
  return zshmc::check_hip(
      hipMemsetAsync(ptr, 0, (size_t)n_bytes,
                     reinterpret_cast<hipStream_t>(stream)),
      "hipMemsetAsync");
}
