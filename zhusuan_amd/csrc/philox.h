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

// a ^ b ^ c in ONE VALU instruction: gfx950's v_bitop3_b32 with the truth
// table of the 3-input XOR (0x96).  hipcc otherwise emits two v_xor_b32, one
// of them with the SGPR round key as operand (the slow issue class,
// tools/instr_bench.hip): 40 of the 60 VALU instructions of a Philox call.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#ifdef ZS_NO_BITOP3  // A/B probe only
  return a ^ b ^ c;
#else
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#endif
}

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1,
                                            uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)kPhiloxM0 * c0;
    const uint64_t p1 = (uint64_t)kPhiloxM1 * c2;
    const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
    const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
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
//   radius: u1 = (float(xa) + 1) * 2^-32 in (0, 1] -- v_cvt_f32_u32 keeps the
//     full exponent range of the 32-bit word, so the small end of u1 (the
//     tail of the normal) is resolved down to 2^-32: |z| reaches 6.66 sigma
//     (24-bit uniforms stop at 5.77; TensorFlow's 23-bit ones at 5.65).
//     One convert + one fma (single rounding) instead of shift/convert/add/mul.
//   angle : the top 23 bits of xb dropped into the mantissa of a float in
//     [1, 2): v_sin/v_cos take revolutions and are periodic, so the "1 +"
//     costs nothing and no convert is needed (shift + or).
__device__ __forceinline__ float bm_radius_uniform(uint32_t x) {
  return __builtin_fmaf((float)x, 0x1p-32f, 0x1p-32f);
}
__device__ __forceinline__ float bm_angle_rev(uint32_t x) {
  return __builtin_bit_cast(float, (x >> 9) | 0x3f800000u);
}
__device__ __forceinline__ void box_muller(uint32_t xa, uint32_t xb,
                                           float& z0, float& z1) {
  const float u1 = bm_radius_uniform(xa);
  const float rev = bm_angle_rev(xb);
  // -2 ln(u1) = (-2 ln 2) * log2(u1)
  const float r =
      __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
  z0 = r * __builtin_amdgcn_cosf(rev);
  z1 = r * __builtin_amdgcn_sinf(rev);
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
