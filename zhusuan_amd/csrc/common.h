// Shared host-side helpers of libzshmc.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/zshmc.h"

namespace zshmc {

void set_error(const char* fmt, ...);

inline int check_hip(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return ZSHMC_ERR_HIP;
  }
  return ZSHMC_OK;
}

// number of CUs of the current device (cached)
int device_cu_count();

#define ZS_REQUIRE(cond, ...)          \
  do {                                 \
    if (!(cond)) {                     \
      zshmc::set_error(__VA_ARGS__);   \
      return ZSHMC_ERR_BAD_ARG;        \
    }                                  \
  } while (0)

#define ZS_LAUNCH_CHECK(what)                                   \
  do {                                                          \
    int _rc = zshmc::check_hip(hipGetLastError(), what);        \
    if (_rc != ZSHMC_OK) return _rc;                            \
  } while (0)

constexpr int kWave = 64;  // gfx950 wavefront

// sum over the `width` consecutive lanes a lane belongs to (width = power
// of two <= 64); every lane of the group gets the total.
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = WIDTH / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace zshmc
