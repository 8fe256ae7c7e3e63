// Philox4x32-7 counter-based RNG for gfx950, bit-identical to
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

// Seven rounds: the fewest Random123 (Salmon et al., SC'11, table 2) lists as
// Crush-resistant for Philox4x32 (10 is its default, with a safety margin).
// The stream is this repository's own documented mapping -- TensorFlow's
// cannot be reproduced either way -- and the generator is 40 % of the fused
// kernel's VALU work: 7 instead of 10 rounds took the headline launch from
// 0.0955 to 0.0912 ms (profiles/archive/r03e_philox7_kbench.txt).  Shared bit for bit
// with oracle/philox.py (known-answer vectors for 7 AND 10 rounds pinned in
// tests/test_oracle_philox.py); -DZS_PHILOX_ROUNDS=10 rebuilds the old stream.
#ifndef ZS_PHILOX_ROUNDS
#define ZS_PHILOX_ROUNDS 7
#endif

struct U4 {
  uint32_t x, y, z, w;
};

// a ^ b ^ c in ONE VALU instruction: gfx950's v_bitop3_b32 with the truth
// table of the 3-input XOR (0x96).  hipcc otherwise emits two v_xor_b32, one
// of them with the SGPR round key as operand (the slow issue class,
// tools/instr_bench.hip): 40 of the 60 VALU instructions of a Philox call.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

__device__ __forceinline__ U4 philox4x32(uint32_t c0, uint32_t c1,
                                            uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < ZS_PHILOX_ROUNDS; ++r) {
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
  const U4 r = philox4x32(group, chain, iteration, stream, k0, k1);
  box_muller(r.x, r.y, z0, z1);
  box_muller(r.z, r.w, z2, z3);
}

// Two calls advanced round by round: a Philox call is two dependent chains
// (mad -> xor3 -> mad ...), and one wave issues in order, so a lone call
// leaves the VALU waiting on its own results (tools/instr_bench.hip,
// k_xor_mad_mix: 2.1 ns per instruction against 1.3 for the same mix without
// the dependency).  Interleaving two calls doubles the independent work
// between dependent instructions.
__device__ __forceinline__ void philox4x32_x2(
    uint32_t a0, uint32_t b0, uint32_t c1, uint32_t c2, uint32_t c3,
    uint32_t k0, uint32_t k1, U4& ra, U4& rb) {
  uint32_t a1 = c1, a2 = c2, a3 = c3, b1 = c1, b2 = c2, b3 = c3;
#pragma unroll
  for (int r = 0; r < ZS_PHILOX_ROUNDS; ++r) {
    const uint64_t pa0 = (uint64_t)kPhiloxM0 * a0;
    const uint64_t pb0 = (uint64_t)kPhiloxM0 * b0;
    const uint64_t pa1 = (uint64_t)kPhiloxM1 * a2;
    const uint64_t pb1 = (uint64_t)kPhiloxM1 * b2;
#ifndef ZS_RNG_NO_FENCE  // keep [4 products][4 xor3] as issued groups (A/B: -DZS_RNG_NO_FENCE)
    __builtin_amdgcn_sched_barrier(0);
#endif
    const uint32_t na0 = xor3((uint32_t)(pa1 >> 32), a1, k0);
    const uint32_t nb0 = xor3((uint32_t)(pb1 >> 32), b1, k0);
    const uint32_t na2 = xor3((uint32_t)(pa0 >> 32), a3, k1);
    const uint32_t nb2 = xor3((uint32_t)(pb0 >> 32), b3, k1);
    a1 = (uint32_t)pa1;
    b1 = (uint32_t)pb1;
    a3 = (uint32_t)pa0;
    b3 = (uint32_t)pb0;
    a0 = na0;
    b0 = nb0;
    a2 = na2;
    b2 = nb2;
    k0 += kPhiloxW0;
    k1 += kPhiloxW1;
#ifndef ZS_RNG_NO_FENCE
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
  ra = U4{a0, a1, a2, a3};
  rb = U4{b0, b1, b2, b3};
}

// eight N(0,1): latents 4ga..4ga+3 and 4gb..4gb+3 of one chain (same counter
// mapping as two normal4 calls: bit-identical results)
__device__ __forceinline__ void normal4x2(uint32_t ga, uint32_t gb,
                                          uint32_t chain, uint32_t iteration,
                                          uint32_t stream, uint32_t k0,
                                          uint32_t k1, float* za, float* zb) {
  U4 ra, rb;
  philox4x32_x2(ga, gb, chain, iteration, stream, k0, k1, ra, rb);
  box_muller(ra.x, ra.y, za[0], za[1]);
  box_muller(rb.x, rb.y, zb[0], zb[1]);
  box_muller(ra.z, ra.w, za[2], za[3]);
  box_muller(rb.z, rb.w, zb[2], zb[3]);
}

__device__ __forceinline__ float uniform_chain(uint32_t chain,
                                               uint32_t iteration,
                                               uint32_t k0, uint32_t k1) {
  return u01(philox4x32(0u, chain, iteration, kStreamMH, k0, k1).x);
}

}  // namespace zshmc
