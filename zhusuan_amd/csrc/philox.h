// Philox4x32-10 counter-based RNG for gfx950, bit-identical to
// oracle/philox.py.  Stands in for tf.random_normal / tf.random_uniform
// (reference zhusuan/hmc.py:22, :485; univariate.py:167, :389) whose
// TensorFlow Philox stream is keyed by graph state and cannot be reproduced.
//
// Counter mapping (DESIGN.md "RNG"):
//   momentum  : (d/4, global chain, iteration, STREAM_MOMENTUM | latent<<8)
//   MH uniform: (0,   global chain, iteration, STREAM_MH)
//   dist ops  : (i/4 lo, i/4 hi,    offset,    STREAM_DIST)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zshmc {

constexpr uint32_t kPhiloxM0 = 0xD2511F53u;
constexpr uint32_t kPhiloxM1 = 0xCD9E8D57u;
constexpr uint32_t kPhiloxW0 = 0x9E3779B9u;
constexpr uint32_t kPhiloxW1 = 0xBB67AE85u;

constexpr uint32_t kStreamMomentum = 0;
constexpr uint32_t kStreamMH = 1;
constexpr uint32_t kStreamDist = 2;

struct U4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1,
                                            uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)kPhiloxM0 * c0;
    const uint64_t p1 = (uint64_t)kPhiloxM1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += kPhiloxW0;
    k1 += kPhiloxW1;
  }
  return U4{c0, c1, c2, c3};
}

// uint32 -> [0,1) with 24 random bits (exact in float32)
__device__ __forceinline__ float u01(uint32_t x) {
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}
// uint32 -> (0,1]
__device__ __forceinline__ float u01_open_low(uint32_t x) {
  return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

// Box-Muller on the hardware transcendental units: v_log_f32 (log2),
// v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in revolutions, so the
// uniform is used directly -- no 2*pi multiply, no range reduction).
__device__ __forceinline__ void box_muller(uint32_t xa, uint32_t xb,
                                           float& z0, float& z1) {
  const float u1 = u01_open_low(xa);
  const float u2 = u01(xb);
  // -2 ln(u1) = (-2 ln 2) * log2(u1)
  const float r =
      __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
  z0 = r * __builtin_amdgcn_cosf(u2);
  z1 = r * __builtin_amdgcn_sinf(u2);
}

// four N(0,1) for latents 4g..4g+3 of one chain
__device__ __forceinline__ void normal4(uint32_t group, uint32_t chain,
                                        uint32_t iteration, uint32_t stream,
                                        uint32_t k0, uint32_t k1, float& z0,
                                        float& z1, float& z2, float& z3) {
  const U4 r = philox4x32_10(group, chain, iteration, stream, k0, k1);
  box_muller(r.x, r.y, z0, z1);
  box_muller(r.z, r.w, z2, z3);
}

__device__ __forceinline__ float uniform_chain(uint32_t chain,
                                               uint32_t iteration,
                                               uint32_t k0, uint32_t k1) {
  return u01(philox4x32_10(0u, chain, iteration, kStreamMH, k0, k1).x);
}

}  // namespace zshmc
