// Hand-placed instruction helpers shared by the two-GEMM likelihood kernels
// (csrc/linear_bernoulli.hip, csrc/linear_bernoulli_mid.hip): LDS-DMA rows,
// LDS operand reads with their destinations as asm outputs, waits, and the
// fp32 MFMAs as asm statements, so that a tile loop issues in the order it
// is written.  What the compiler cannot see inside an asm statement it cannot
// protect -- each helper says what it leaves to its caller.
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"

namespace zshmc {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int FB>
struct VecF {};
template <>
struct VecF<1> {
  typedef float type;
};
template <>
struct VecF<2> {
  typedef float type __attribute__((ext_vector_type(2)));
};
template <>
struct VecF<4> {
  typedef f4 type;
};

template <int FB, typename V>
__device__ __forceinline__ float vget(const V& v, int t) {
  if constexpr (FB == 1)
    return v;
  else
    return v[t];
}

// global -> LDS, BYTES (4, 8 = 2x4, 12, 16) per lane, LDS dest = dst + lane*BYTES
// (HALF: a 512-byte row is two instructions; 0 / 1 issues only the first /
// second, for callers that place them in different issue gaps)
template <int BYTES, int HALF = -1>
__device__ __forceinline__ void lds_dma_row(const float* src, uint32_t dst,
                                            uint32_t lane) {
  if constexpr (BYTES != 8 && HALF == 1) return;  // one instruction: half 0
  static_assert(BYTES == 4 || BYTES == 8 || BYTES == 12 || BYTES == 16,
                "a row is 256 B, 512 B, 768 B or 1 KB");
  if constexpr (BYTES == 12) {
    // a 768-byte row: the 16-byte form with the last quarter of the wave
    // masked off for the one instruction (the destination is M0 + 16 * lane
    // for the lanes that run; global_load_lds_dwordx3 does NOT pack its lanes
    // 12 bytes apart).  Called with all 64 lanes active.
    const uint32_t voff = lane * 16u;
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_bfm_b64 exec, 48, 0\n\t"
        "global_load_lds_dwordx4 %0, %1\n\t"
        "s_mov_b64 exec, -1"
        :
        : "v"(voff), "s"(src), "s"(dst)
        : "memory");
  } else if constexpr (BYTES == 16) {
    const uint32_t voff = lane * 16u;
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(src), "s"(dst)
        : "memory");
  } else {
    const uint32_t voff = lane * 4u;
    if constexpr (HALF != 1)
      asm volatile(
          "s_mov_b32 m0, %2\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dword %0, %1"
          :
          : "v"(voff), "s"(src), "s"(dst)
          : "memory");
    if constexpr (BYTES == 8 && HALF != 0)
      asm volatile(
          "s_mov_b32 m0, %2\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dword %0, %1 offset:256"
          :
          : "v"(voff), "s"(src), "s"(dst)
          : "memory");
  }
}

typedef float f2 __attribute__((ext_vector_type(2)));

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f,
                                                std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int OFF>
__device__ __forceinline__ void lds_read(f4& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read(f2& d, uint32_t addr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// 20 wait states: a 16-pass MFMA's result is in its VGPRs (and the compiler,
// which cannot see the MFMA inside an asm statement, reads `acc` after this)
__device__ __forceinline__ void mfma_drain(f16v& acc) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc));
}
// the same for an AGPR tile, before the compiler's own reads of it
__device__ __forceinline__ void mfma_drain_a(f16v& acc) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc));
}
// lgkmcnt(0) with the destinations of pending asm reads held until then
__device__ __forceinline__ void land_reads(f4& a) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory");
}
__device__ __forceinline__ void land_reads(f4& a, f4& b, f4& c, f4& d, f4& e) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e)
               :
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
// accumulate into an AGPR tile (G: touched by MFMAs only until the epilogue)
// (`s_nop 1`: hipcc may materialise an input with a VALU copy -- an AGPR-parked
// value, a sub-register move -- right in front of the statement, and an MFMA
// reading a VGPR needs two wait states behind a VALU write of it; inside an asm
// statement that is ours to provide.  Under the previous MFMA's 16 passes the
// two issue cycles are free.)
__device__ __forceinline__ void mfma_a(f16v& acc, float a, float b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0"
               : "+a"(acc)
               : "v"(a), "v"(b));
}

// One MFMA of a phase-1 accumulator chain in VGPRs (S = 0 + a b for the first).
// One statement per MFMA: a wave issues one instruction per ~4 clocks and an
// MFMA holds the pipe for 64 (32x32x2) or 32 (16x16x4), so ~15 / ~7 other
// instructions fit between two MFMAs of a chain for free -- the callers put a
// step's LDS read, DMA instructions and scalar arithmetic there.  (A dependent
// MFMA on the same accumulator needs no wait states, and none are lost to
// what is issued in between; hipcc adds an `s_nop 0` between two statements
// that pass a VGPR: one slot.)
template <bool FIRST>
__device__ __forceinline__ void mfma_v(f16v& S, float a, float b) {
  if constexpr (FIRST)
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, 0"
                 : "=&v"(S)
                 : "v"(a), "v"(b));
  else
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0"
                 : "+v"(S)
                 : "v"(a), "v"(b));
}

// ---- the same on v_mfma_f32_16x16x4_f32 (8 passes, 4 accumulator registers:
// csrc/linear_bernoulli_mid.hip) ------------------------------------------------
__device__ __forceinline__ void mfma16_a(f4& acc, float a, float b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0"
               : "+a"(acc)
               : "v"(a), "v"(b));
}
// one MFMA of a phase-1 chain in VGPRs (S = 0 + a b for the first)
template <bool FIRST>
__device__ __forceinline__ void mfma16_v(f4& S, float a, float b) {
  if constexpr (FIRST)
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, 0"
                 : "=&v"(S)
                 : "v"(a), "v"(b));
  else
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0"
                 : "+v"(S)
                 : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_drain(f4& acc) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc));
}
__device__ __forceinline__ void mfma_drain_a(f4& acc) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc));
}
__device__ __forceinline__ void land_reads(f4& a, f4& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory");
}

// global -> LDS, 16 bytes per lane for the first LANES lanes of the wave
// (LANES * 16 bytes of a row), LDS dest = dst + 16 * lane.  Called with all
// 64 lanes active; EXEC is narrowed for the one instruction.
template <int LANES>
__device__ __forceinline__ void lds_dma_x4(const float* src, uint32_t dst,
                                           uint32_t lane) {
  static_assert(LANES >= 1 && LANES <= 64, "lanes of one wave");
  const uint32_t voff = lane * 16u;
  if constexpr (LANES == 64)
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(src), "s"(dst)
        : "memory");
  else
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_bfm_b64 exec, %3, 0\n\t"
        "global_load_lds_dwordx4 %0, %1\n\t"
        "s_mov_b64 exec, -1"
        :
        : "v"(voff), "s"(src), "s"(dst), "n"(LANES)
        : "memory");
}

}  // namespace zshmc
