// Isolated cost of the fused kernel's two VALU hot spots at W waves/SIMD:
//   lf : the leapfrog trip (16 v_pk_fma_f32: r += eps*p ; p += nep*r)
//   rng: normal4 x4 (Philox4x32-7 + Box-Muller for 16 normals)
// Reports cycles per trip per SIMD (wall clock x assumed 2.0-2.4 GHz is
// avoided: uses s_memtime of one wave as the clock and the wall time).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../zhusuan_amd/csrc/philox.h"
typedef float f4 __attribute__((ext_vector_type(4)));
using namespace zshmc;

__global__ __launch_bounds__(256) void k_lf(float* out, float eps_in, int iters) {
  f4 r[4], p[4], nep[4];
  for (int k = 0; k < 4; ++k) {
    r[k] = f4{1.f, 2.f, 3.f, 4.f} * (float)(threadIdx.x + k);
    p[k] = f4{0.5f, 0.25f, 0.125f, 1.f} * (float)(k + 1);
    nep[k] = f4{-1e-3f, -2e-3f, -3e-3f, -4e-3f};
  }
  const float eps = eps_in;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      r[k] += eps * p[k];
      p[k] += nep[k] * r[k];
    }
  }
  f4 s = r[0] + r[1] + r[2] + r[3] + p[0] + p[1] + p[2] + p[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256) void k_rng(float* out, uint32_t k0, uint32_t k1, int iters) {
  f4 acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float z0, z1, z2, z3;
      normal4(k * 64 + (threadIdx.x & 63), blockIdx.x * 4 + (threadIdx.x >> 6), i, 0, k0, k1, z0, z1, z2, z3);
      acc += f4{z0, z1, z2, z3};
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w : {1, 2, 4, 8}) {
    const int blocks = 256 * w;  // w blocks of 4 waves per CU -> w waves/SIMD
    for (int which = 0; which < 2; ++which) {
      const int iters = which == 0 ? 20000 : 2000;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k_lf, dim3(blocks), dim3(256), 0, 0, out, 1e-3f, iters);
        else hipLaunchKernelGGL(k_rng, dim3(blocks), dim3(256), 0, 0, out, 1u, 2u, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // per SIMD: w waves x iters trips
      const double ns_per_trip = ms * 1e6 / ((double)iters * w);
      printf("%s waves/SIMD=%d: %.3f ms  %.1f ns per trip per SIMD  (= %.0f cycles @2.0GHz, %.0f @2.4GHz)\n",
             which == 0 ? "lf " : "rng", w, ms, ns_per_trip, ns_per_trip * 2.0, ns_per_trip * 2.4);
    }
  }
  return 0;
}
