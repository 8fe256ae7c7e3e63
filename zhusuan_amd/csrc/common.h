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

// While one of these lives on the calling thread, the split likelihood
// launches (lb_reduce_splits and its callers) leave their row-range partials
// in the workspace and launch no reduction: the caller's next kernel adds
// them itself (csrc/hmc_model_run.hip: the leapfrog step, csrc/model_step.h).
struct KeepSplitParts {
  KeepSplitParts();
  ~KeepSplitParts();
  static bool active();
};

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

// Sum over the 64 lanes of a wave on the DPP data path (no LDS round trip):
// inclusive scan inside each row of 16 lanes (row_shr 1,2,4,8), then the row
// totals are chained with row_bcast15 / row_bcast31 (gfx9 DPP broadcasts), so
// lane 63 holds the total, which is returned wave-uniform (in an SGPR).
// The summation tree differs from group_sum<64>'s xor butterfly, so the two
// agree to rounding, not bit for bit.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(
      0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float wave_total_dpp(float v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1, 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2, 3
  return __builtin_bit_cast(
      float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// four totals at once, the scan steps interleaved so that consecutive DPP
// instructions are independent (a DPP read of a just-written VGPR costs two
// wait states)
__device__ __forceinline__ void wave_total4_dpp(float& a, float& b, float& c,
                                                float& d) {
#define ZS_DPP_STEP(CTRL, MASK) \
  a = dpp_add<CTRL, MASK>(a);   \
  b = dpp_add<CTRL, MASK>(b);   \
  c = dpp_add<CTRL, MASK>(c);   \
  d = dpp_add<CTRL, MASK>(d);
  ZS_DPP_STEP(0x111, 0xf)
  ZS_DPP_STEP(0x112, 0xf)
  ZS_DPP_STEP(0x114, 0xf)
  ZS_DPP_STEP(0x118, 0xf)
  ZS_DPP_STEP(0x142, 0xa)
  ZS_DPP_STEP(0x143, 0xc)
#undef ZS_DPP_STEP
  a = __builtin_bit_cast(
      float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
  b = __builtin_bit_cast(
      float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 63));
  c = __builtin_bit_cast(
      float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), 63));
  d = __builtin_bit_cast(
      float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), 63));
}

// Four wave totals through the gfx950 half / row exchanges instead of four
// separate DPP scans (24 DPP + 8 moves + 8 adds + 4 readlanes -> 3 swaps +
// 3 adds + 4 DPP + 4 readlanes; DPP and SGPR-operand VALU ops issue at half
// the rate of plain ones, tools/instr_bench.hip):
//   v_permlane32_swap(a, b): lanes 32-63 of a <-> lanes 0-31 of b, so
//     a' + b' folds a onto lanes 0-31 and b onto lanes 32-63;
//   v_permlane16_swap(ab, cd): odd rows of ab <-> even rows of cd, so
//     ab' + cd' leaves one value per row of 16 lanes (rows: a, c, b, d);
//   one row_shr scan then finishes all four at once.
__device__ __forceinline__ void wave_total4_swap(float& a, float& b, float& c,
                                                 float& d) {
  // inline asm, operands updated in place; the two wait states the swaps
  // need after a VALU write of an operand are inside the string (hipcc's
  // builtin mis-pairs the two results when both feed one add)
  asm volatile(
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %1\n\t"
      "v_permlane32_swap_b32 %2, %3"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  float sab = a + b;
  float scd = c + d;
  asm volatile(
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %0, %1"
      : "+v"(sab), "+v"(scd));
  float v = sab + scd;
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8
  const int vi = __builtin_bit_cast(int, v);
  a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 15));
  c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 31));
  b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 47));
  d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 63));
}

// Sum of the S partials p[0], p[stride], p[2 stride], ... in a FIXED order
// with eight loads in flight: partial s goes to accumulator s % 8 (in order
// of s), the accumulators are folded ((0+1)+(2+3))+((4+5)+(6+7)).  The one
// reduction of the row-range partials of a split likelihood launch
// (lb_reduce_splits_kernel, and csrc/model_step.h where the step reads the
// partials itself): deterministic, and the same bits wherever it runs.
// (ZS_PARTS_BATCHES x 8 loads in flight where there are that many parts: the
// adds stay in the order of s per accumulator, so the batch size does not
// show in the result; a translation unit short of registers sets it to 1)
#ifndef ZS_PARTS_BATCHES
#define ZS_PARTS_BATCHES 3
#endif
template <typename T, int NB = ZS_PARTS_BATCHES>
__device__ __forceinline__ T sum_parts8(const T* __restrict__ p, int64_t stride,
                                        int S) {
  T a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = T{};
  int s = 0;
  if constexpr (NB > 1) {
    for (; s + 8 * NB <= S; s += 8 * NB) {
      T v[8 * NB];
#pragma unroll
      for (int k = 0; k < 8 * NB; ++k) v[k] = p[(int64_t)(s + k) * stride];
#pragma unroll
      for (int k = 0; k < 8 * NB; ++k) a[k & 7] += v[k];
    }
  }
  for (; s + 8 <= S; s += 8) {
    T v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[(int64_t)(s + k) * stride];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += v[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (s + k < S) a[k] += p[(int64_t)(s + k) * stride];
  return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace zshmc
