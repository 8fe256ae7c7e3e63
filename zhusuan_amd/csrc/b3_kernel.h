// The dense-logit likelihoods of csrc/linear_bernoulli.hip on the BF16 matrix
// cores with float32-level results: every float32 operand is split into three
// bfloat16 planes, x = x_hi + x_mid + x_lo (8 + 8 + 8 mantissa bits: the split
// is exact), and a product a*b is the six terms
//     hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid
// on v_mfma_f32_32x32x16_bf16 with float32 accumulation -- what is dropped
// (mid*lo, lo*mid, lo*lo) is <= 2^-24 relative, the size of float32's own
// rounding (tools/bf16x3_accuracy.py).  Six bf16 MFMAs of 32 clocks do the
// work of eight 64-clock v_mfma_f32_32x32x2_f32: 2.67x the fp32-MFMA peak
// (157.3 TFLOP/s) is the ceiling, 6/16 of the dense bf16 peak.
//
//   logits[c, n] = sum_d W[c, d] * X[n, d]
//   OP 0  Bernoulli (univariate.py:398-403), y[n]:   r = y - sigmoid(l)
//   OP 1  UnnormalizedMultinomial over a mixture (multivariate.py:435-443),
//         counts x[n] of the workgroup's document:   r = x / S
//   OP 2  Categorical (univariate.py:496-548), rows of W = (chain, class)
//         pairs, label y[n]:                         r = [k == y] - softmax_k(l)
//   gW[c, :] = sum_n r[n, c] * X[n, :]               (hmc.py:430-432)
//
// X (constant over a run) is split ONCE into a tile image
// (zshmc_bf16x3_split): 32-row tiles, three planes, 2 KB blocks of 32 rows x
// 32 features whose 16-byte chunks (row m, 8 features) are placed so that
// BOTH operand reads are conflict-free and contiguous per instruction:
//   GEMM 1, A = X rows (k = features): ds_read_b128, lane (row, k half);
//   GEMM 2, B = X^T    (k = rows):     ds_read_b64_tr_b16 -- the gfx950
//           transposing LDS read -- each 16-lane group a [4 rows][16
//           features] block.
// GEMM 1's accumulator layout (lane = chain, registers = rows in groups of
// four) is NOT the k-packed A layout of the bf16 MFMA in row order -- but k
// is a dummy index: GEMM 2 contracts over the 16 rows of a k-step in the
// order the accumulator registers hold them, and the B operand (ours to
// read in any order) follows: slot i of lane half h is row 8 (i/4) + 4 h +
// i%4 on both sides.  The residual never crosses lanes.
//
// A workgroup = 4 waves x 32 chains on the SAME 32-row tile (W planes 3 D/8
// registers, gradient accumulators D/2: one wave per SIMD at 192 / 256
// columns, two -- two workgroups per CU -- at <= 128).  The tile loop is
// software-pipelined over THREE LDS buffers with one barrier per tile:
//   iteration t:  GEMM 1 (t), the first pairs of the element-wise stage of
//                 tile t-1 in its issue gaps
//                 GEMM 2 (t-1), k-step-major: the rest of the stage (and, at
//                 <= 128 columns, the DMA of tile t+1) in its first half; the
//                 barrier, the buffer rotation and the first operand reads
//                 of iteration t+1 in front of / under its last six MFMAs
// A bf16 MFMA leaves ~5 issue slots per 32 clocks; a wave's VALU work is only
// PARTLY hidden under its own MFMAs (the stage costs ~600 clocks per tile
// wherever it rides, profiles/r05g_b3_phase_split.txt) -- a second resident
// workgroup hides the rest.
// Roofline: bf16 MFMA; algorithmic flops 4*N*D*C per call, issued 6x.
#pragma once
#include <stdlib.h>

#include "common.h"
#include "lb_asm.h"
#include "lb_ops.h"

namespace zshmc {

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

constexpr int kB3Rows = 32;     // data rows per tile
constexpr int kB3Chains = 128;  // chains per workgroup

#ifndef ZS_B3_SWIZZLE
#define ZS_B3_SWIZZLE 1
#endif
// Timing experiments only (wrong results): bit 0 no element-wise stage, bit 1
// no tile DMA in the loop, bit 2 no label DMA / reads, bit 3 no GEMM 1,
// bit 4 no GEMM 2, bit 5 no `s_nop 1` in front of the MFMAs
// (tools/build_b3_variants.sh; profiles/r05g_b3_phase_split.txt).
#ifndef ZS_B3_SKIP
#define ZS_B3_SKIP 0
#endif
// the DMA of the next tile in GEMM 1's gaps (1) or GEMM 2's first half (0);
// default: GEMM 1 for the wide tiles (D >= 192), GEMM 2 for the narrow ones
#ifndef ZS_B3_DMA_IN_GEMM1
#define ZS_B3_DMA_IN_GEMM1 (D >= 192)
#endif
#if ZS_B3_SKIP & 32
#define ZS_B3_NOP ""
#else
#define ZS_B3_NOP "s_nop 1\n\t"
#endif

// 16-byte chunk (row m of the tile, feature half h, 16-feature sub-block par)
// of a 2 KB block of 32 rows x 32 features.  Swizzled: every 16-lane service
// group of the b128 read (lanes {0-3,12-15,20-27} / {4-11,16-19,28-31} of a
// half) and every 32-lane half of the transposing read (4 rows x 2 sub-blocks
// x 2 halves) touch 16 different chunk columns (mod 16) -- all 64 banks once.
__host__ __device__ constexpr int b3_chunk(int m, int h, int par) {
#if ZS_B3_SWIZZLE
  return (m >> 2) * 16 + 4 * ((2 * h + par + (m >> 3)) & 3) + (m & 3);
#else
  return par * 64 + m * 2 + h;
#endif
}

// float32 pair -> one register of two bfloat16 (round to nearest even):
// v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 p = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float bf_lo(unsigned p) {
  return __builtin_bit_cast(float, p << 16);
}
__device__ __forceinline__ float bf_hi(unsigned p) {
  return __builtin_bit_cast(float, p & 0xffff0000u);
}
// (a, b) -> packed hi / mid / lo planes
struct Split3 {
  unsigned hi, mid, lo;
};
__device__ __forceinline__ Split3 split3(float a, float b) {
  Split3 r;
  r.hi = pk_bf16(a, b);
  const float a1 = a - bf_lo(r.hi), b1 = b - bf_hi(r.hi);
  r.mid = pk_bf16(a1, b1);
  r.lo = pk_bf16(a1 - bf_lo(r.mid), b1 - bf_hi(r.mid));
  return r;
}

// ---- LDS operand reads ------------------------------------------------------
// Compiler-tracked (it places the `s_waitcnt lgkmcnt(n)` in front of the asm
// MFMA that consumes the registers, and no copy of a value still in flight
// can happen behind its back); pinned to their issue gaps by the
// sched_barrier fences of the tile loop.  The immediate offsets fold into
// the instructions (one base register per lane pattern).
#define ZS_LDS __attribute__((address_space(3)))
typedef short s4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u4 lds_u4(uint32_t addr, int off) {
  return *reinterpret_cast<const ZS_LDS u4*>((uintptr_t)(addr + (uint32_t)off));
}
__device__ __forceinline__ f4 lds_f4(uint32_t addr, int off) {
  return *reinterpret_cast<const ZS_LDS f4*>((uintptr_t)(addr + (uint32_t)off));
}
// the transposing read (ds_read_b64_tr_b16): within each 16-lane group, lane
// i receives element i % 4 of the 8 bytes lane 4 j + i / 4 points at, j = 0..3
// -- column i of the [4 rows][16 features] block the group's lanes address.
__device__ __forceinline__ u2 lds_tr(uint32_t addr, int off) {
  return __builtin_bit_cast(
      u2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              reinterpret_cast<ZS_LDS s4v*>((uintptr_t)(addr + (uint32_t)off))));
}
// one 32x32x16 bf16 MFMA; the B operand in VGPRs or AGPRs
__device__ __forceinline__ void mfma_bv(f16v& acc, const u4& a, const u4& b) {
  asm volatile(ZS_B3_NOP "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
               : "+v"(acc)
               : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_bv0(f16v& acc, const u4& a, const u4& b) {
  asm volatile(ZS_B3_NOP "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0"
               : "=&v"(acc)
               : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_ba(f16v& acc, const u4& a, const u4& b) {
  asm volatile(ZS_B3_NOP "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
               : "+v"(acc)
               : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_ba0(f16v& acc, const u4& a, const u4& b) {
  asm volatile(ZS_B3_NOP "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0"
               : "=&v"(acc)
               : "v"(a), "a"(b));
}
// accumulate into an AGPR tile
__device__ __forceinline__ void mfma_g(f16v& acc, const u4& a, const u4& b) {
  asm volatile(ZS_B3_NOP "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
               : "+a"(acc)
               : "v"(a), "v"(b));
}
// A wave-uniform pointer the compiler computed on the vector unit (64-bit
// divisions have no scalar form) -> SGPRs, for the "s" operands below
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const uintptr_t v = reinterpret_cast<uintptr_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<T*>(((uintptr_t)hi << 32) | lo);
}
// 1 KB of a tile: global -> LDS, 16 bytes per lane
template <int OFF>
__device__ __forceinline__ void b3_dma(const unsigned char* src, uint32_t dst,
                                       uint32_t voff) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1 offset:%3"
      :
      : "v"(voff), "s"(src), "s"(dst), "n"(OFF)
      : "memory");
}
// the register homes of one k-step's W planes (and: the prologue's loads
// have landed, in the compiler's books too -- csrc/linear_bernoulli.hip)
template <bool LO_AGPR>
__device__ __forceinline__ void home_w(u4& h, u4& m, u4& l) {
  if constexpr (LO_AGPR)
    asm volatile("" : "+v"(h), "+a"(m), "+a"(l));
  else
    asm volatile("" : "+v"(h), "+a"(m), "+v"(l));
}
__device__ __forceinline__ void pin(float& a, float& b) {
  asm volatile("" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void pin(float& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pin(unsigned& a) { asm volatile("" : "+v"(a)); }
template <int I>
using template_int = std::integral_constant<int, I>;
// 10 wait states behind an 8-pass MFMA's write before a VALU read
__device__ __forceinline__ void mfma_drain8(f16v& acc) {
  asm volatile("s_nop 11" : "+v"(acc));
}

// The six terms of a product in issue order (plane 0 = hi, 1 = mid, 2 = lo),
// grouped by the plane of the operand that streams through LDS -- X: the A
// operand of GEMM 1, the B operand of GEMM 2 -- so that ONE register set
// holds it: plane p's last use in a step is term kLastUse[p], and the next
// step's plane p is read right behind it (three to six MFMAs ahead of its
// first use).  (A second, ping-pong set costs 12 registers: two waves per
// SIMD at D = 128 do not have them.)
constexpr int kTermX[6] = {0, 0, 0, 1, 1, 2};   // the streamed operand's plane
constexpr int kTermR[6] = {0, 1, 2, 0, 1, 0};   // the register operand's plane
constexpr int kLastUse[3] = {2, 4, 5};

// Waves per SIMD the register budget is held to: two where the tile's three
// buffers fit the LDS twice (D <= 128: 2 x 72 KB) -- one workgroup's
// element-wise stage then runs under the other's MFMAs.
#ifndef ZS_B3_WAVES
#define ZS_B3_WAVES(D) ((D) <= 128 ? 2 : 1)
#endif

// ---------------------------------------------------------------------------
#ifndef ZS_B3_SP_XCD
#define ZS_B3_SP_XCD 1    // SP: document-major order per XCD (A/B knob)
#endif
#ifndef ZS_B3_PK_AHEAD
#define ZS_B3_PK_AHEAD 2   // PK, D <= 128: tiles the counts are fetched ahead
#endif
// GL: log2 of the class stride (OP 2)
// PK (OP 1): packed rows -- row r of theta belongs to document r % yc_rows (a
// few chains x many documents, lntm_mcem.py's own layout: n_chains = 1), so
// every chain of a workgroup has its OWN counts row: the "labels" of a tile
// are [32 vocabulary rows][128 chains] floats (16 KB per buffer instead of
// 128 B), brought by four 16-byte-per-lane DMAs per wave -- DMA g the
// 128-byte lines of chains 8 g .. 8 g + 7, eight consecutive lanes a line,
// each chain's eight 16-byte pieces rotated by the chain on their way into
// the LDS (the DMA's global side is per lane, its LDS side linear by lane: the
// DMA itself is the permutation) so that the element-wise stage's lanes read
// their pairs without more than the two-way conflict 8-byte reads of 16-byte
// slots have anyway.  At D <= 128 the counts are fetched TWO tiles ahead (they
// come from HBM, every workgroup its own; the tile image is L2-resident).
// Needs counts rows padded to a multiple of 32 floats (no clamping at the last
// tile) and a counts matrix below 4 GB (32-bit lane offsets); three tile
// buffers + 48 KB (64 KB at D <= 128) fit the LDS up to D = 192, one workgroup
// per CU.
// SP (OP 1, one document per workgroup): the document's OWN vocabulary.  A
// bag of words is sparse -- ~1 000 tokens over 12 419 words in lntm_mcem.py's
// corpus -- and a word the document does not contain contributes exactly
// nothing (x = 0: r = x / S = 0, 0 * log S = 0; multivariate.py:435-443).  The
// tile loop therefore runs over the document's nonzero words only: `y` is the
// compacted counts, `sp_rows` the words' rows of phi^T, both at sp_off[doc]
// .. sp_off[doc + 1] (padded to whole tiles with count 0 / row 0).  A tile is
// GATHERED: the image is 16-byte chunks and global_load_lds takes a per-lane
// global offset, so lane l of a 1 KB piece fetches the chunk that belongs at
// its place -- row sp_rows[32 t + m'] of the image, m' the tile row of the
// LDS chunk, the swizzle of b3_chunk re-applied for the source row's position
// in ITS tile.  The 32 row indices of tile t + 2 travel into a small LDS ring
// with the labels of tile t + 1; two lane offsets per tile (a piece is half a
// 2 KB block: 16 of its 32 rows) are all the state.  Same MFMA schedule.
template <int D, int OP, bool LL, int NACC, int GL = 0, bool PK = false,
          bool SP = false>
__global__ __launch_bounds__(256, PK ? 1 : ZS_B3_WAVES(D)) void linear_b3_kernel(
    const float* __restrict__ W, const unsigned char* __restrict__ Ximg,
    const float* __restrict__ y, int64_t yc_rows, int64_t ldy, int64_t C,
    int64_t N_arg, int64_t ldw, float* __restrict__ ll, float* __restrict__ gW,
    int doc_major, int n_classes, const int32_t* __restrict__ sp_rows,
    const int64_t* __restrict__ sp_off) {
  static_assert(!SP || (OP == 1 && !PK), "own vocabulary: one document per "
                                         "workgroup, multinomial");
  static_assert(D % 32 == 0 && D >= 32 && D <= 256, "32 .. 256 features");
  static_assert(NACC == 1 || NACC == 2, "accumulator chains of GEMM 1");
  static_assert(!PK || (OP == 1 && D <= 192), "packed rows: multinomial, LDS");
  constexpr int kYBuf = PK ? 16384 : 128;   // bytes of one label buffer
  // PK: the counts of a tile come from HBM (every workgroup its own rows; the
  // tile image is shared and L2-resident): where the LDS has room (D <= 128:
  // 72 + 64 KB) they are fetched TWO tiles ahead into a ring of four buffers,
  // and the iteration boundary waits for everything but those four DMAs
  // (vmcnt(4): they are the last issued) -- an HBM round trip is as long as
  // a tile (rocprofv3, r06p: MFMA busy 0.52 with the counts one tile ahead)
  constexpr int kYAhead = (PK && D <= 128) ? ZS_B3_PK_AHEAD : 1;
  constexpr int kYN = kYAhead + 2;          // label buffers
  constexpr int KS = D / 16;             // k-steps of GEMM 1 (16 features)
  constexpr int NB = D / 32;             // gradient accumulators (32 features)
  constexpr int kPlane = NB * 2048;      // bytes of one plane of a tile
  constexpr int kTile = 3 * kPlane;      // bytes of a tile (3 D / 16 KB)
  constexpr int kDma = kTile / 4 / 1024; // 1 KB pieces per wave and tile
  static_assert(kTile % 4096 == 0, "a wave's share is whole KBs");
  // Register homes: the gradient accumulators (D/2) and the mid plane of W
  // (D/4) live in AGPRs, the hi plane (D/4) in VGPRs; the lo plane goes where
  // room is -- at D = 256 the AGPR file would be exactly full with it, and
  // hipcc then shuffles every operand through one spare quad.
  auto wl_agpr = [](int ks) constexpr { return D <= 192 || ks < KS / 2; };
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [3][kTile] tiles, [3][32] labels (PK: [3][4 waves][4 groups][64 lanes][4])
  const uint32_t sx_addr = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  const uint32_t sy_addr = sx_addr + 3 * kTile;
  const uint32_t sr_addr = sy_addr + kYN * kYBuf;   // SP: [4][32] row indices
  auto y_slot = [&](int t) -> uint32_t {    // PK: tile t (local index)
    return (uint32_t)((t % kYN) * kYBuf);
  };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;

  // the 128 rows of W this workgroup owns (csrc/linear_bernoulli.hip: OP 1,
  // doc_major: 128 chains of ONE document, rows chain * yc_rows + doc)
  int64_t row_base = (int64_t)blockIdx.x * kB3Chains, row_stride = 1;
  int64_t n_valid = C - row_base;
  int64_t doc = 0;
  if (OP == 1 && !PK && (doc_major || SP)) {
    int64_t grp = blockIdx.x / yc_rows;
    doc = blockIdx.x % yc_rows;
    if (SP && ZS_B3_SP_XCD && gridDim.y == 1) {
      // The workgroups of ONE document gather the same ~30 tiles (730 KB of
      // the image at K = 128), and the image (9.5 MB) does not fit an XCD's
      // 4 MB L2: with documents interleaved over the grid every gathered tile
      // came from the Infinity Cache (rocprofv3, r06s: 74 GB of fabric reads
      // per launch).  Workgroups are dealt round-robin to the 8 XCDs, so XCD x
      // takes a CONTIGUOUS range of a document-major order: a document's
      // workgroups run one after the other on one XCD, out of its L2.
      const int64_t total = gridDim.x, nG = total / yc_rows;
      const int64_t x = blockIdx.x % 8, k = blockIdx.x / 8;
      const int64_t q = total / 8, r = total % 8;
      const int64_t j = x * q + (x < r ? x : r) + k;
      doc = j / nG;
      grp = j % nG;
    }
    row_base = grp * kB3Chains * yc_rows + doc;
    row_stride = yc_rows;
    n_valid = C / yc_rows - grp * kB3Chains;
  }
  n_valid = n_valid < kB3Chains ? n_valid : kB3Chains;
  auto row_at = [&](int i) -> int64_t {
    return row_base + (int64_t)(i < n_valid ? i : (int)n_valid - 1) * row_stride;
  };
  // labels (OP 0) / the document's counts (OP 1) of the data rows
  // SP: this document's slice of the compacted counts / row indices
  const int64_t sp0 = SP ? sp_off[doc] : 0;
  const int64_t N = SP ? sp_off[doc + 1] - sp0 : N_arg;   // whole tiles
  const int32_t* rsrc = SP ? uniform_ptr(sp_rows + sp0) : nullptr;
  const float* ysrc = uniform_ptr(
      SP ? y + sp0 : (OP == 1 && !PK ? y + doc * ldy : y));
  // PK: byte offsets of this lane's share of the four counts DMAs of a tile.
  // DMA g brings chains 8 g .. 8 g + 7 of the wave, EIGHT CONSECUTIVE LANES a
  // chain's whole 128-byte line (32 rows) -- one request to the texture path
  // per line instead of one per lane; lane l is chain c = 8 g + l / 8 and lands
  // in slot l % 8 of the chain's row of the buffer, which holds piece (rows
  // 4 P .. 4 P + 3) P = (l % 8 - c / 2) mod 8: rotated by the chain so that
  // the stage's reads -- 32 chains, the SAME piece -- spread over all banks
  // (row-major placement: 16-way conflicts)
  uint32_t pk_voff[4] = {0u, 0u, 0u, 0u};
  if constexpr (PK) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 8 * g + (lane >> 3);
      const int P = ((lane & 7) - (c >> 1)) & 7;
      pk_voff[g] = (uint32_t)(((row_at(wave * 32 + c) % yc_rows) * ldy + 4 * P) * 4);
    }
  }
  // ... and where the stage's lane (chain lo, half hi) finds piece 2 G + hi:
  // slot (2 G + hi + lo / 2) mod 8 of its chain's row
  const int pk_rot = hi + (lo >> 1);

  // ---- this wave's chain block of W -> three bf16 planes (B operand of
  // GEMM 1: lane = chain, k-slot i of half hi = feature 16 ks + 8 hi + i) ----
  u4 wh[KS], wm[KS], wl[KS];
  {
    const float* __restrict__ wrow = W + row_at(wave * 32 + lo) * ldw + hi * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f4 v0 = *reinterpret_cast<const f4*>(wrow + ks * 16);
      const f4 v1 = *reinterpret_cast<const f4*>(wrow + ks * 16 + 4);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const Split3 a = split3(v0[2 * q], v0[2 * q + 1]);
        const Split3 b = split3(v1[2 * q], v1[2 * q + 1]);
        wh[ks][q] = a.hi, wm[ks][q] = a.mid, wl[ks][q] = a.lo;
        wh[ks][2 + q] = b.hi, wm[ks][2 + q] = b.mid, wl[ks][2 + q] = b.lo;
      }
    }
    // landed in the compiler's books before the loop (csrc/linear_bernoulli.hip)
    static_for<KS>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      home_w<wl_agpr(ks)>(wh[ks], wm[ks], wl[ks]);
    });
  }

  // ---- tiles of this workgroup's row range (gridDim.y slices) --------------
  const int64_t n_tiles_all = (N + kB3Rows - 1) / kB3Rows;
  const int64_t tiles_per_split = (n_tiles_all + gridDim.y - 1) / gridDim.y;
  const int64_t tile_begin = (int64_t)blockIdx.y * tiles_per_split;
  const int64_t tile_end = tile_begin + tiles_per_split < n_tiles_all
                               ? tile_begin + tiles_per_split
                               : n_tiles_all;
  if (gridDim.y > 1) {
    if (LL) ll += (int64_t)blockIdx.y * C;
    gW += (int64_t)blockIdx.y * C * ldw;
  }
  const int T = tile_end > tile_begin ? (int)(tile_end - tile_begin) : 0;
  const int64_t t_first = tile_begin < n_tiles_all ? tile_begin : n_tiles_all - 1;

  // ---- DMA: a tile is 4 x kDma linear KBs, wave w moves its quarter --------
  const uint32_t voff = (uint32_t)lane * 16u;
  const unsigned char* xsrc =   // this wave's quarter of the first tile
      SP ? uniform_ptr(Ximg + wave * (kTile / 4) - 1024)  // (see sp_voff)
         : uniform_ptr(Ximg + t_first * (int64_t)kTile + wave * (kTile / 4));
  const uint32_t dst_wave = sx_addr + (uint32_t)(wave * (kTile / 4));
  // SP: the lane offsets of the tile being fetched.  Piece i of this wave is
  // half b = (wave * kDma + i) & 1 of a 2 KB block: LDS chunk c' = 64 b + lane
  // = tile row m' = 4 (c' / 16) + c' % 4, slot column q' = (c' / 4) % 4, i.e.
  // feature group hp = 2 h + par = (q' - m' / 8) mod 4 (b3_chunk).  Its data
  // is chunk (m_s / 4) * 16 + 4 ((hp + m_s / 8) mod 4) + m_s % 4 of the same
  // block of image tile r / 32, m_s = r % 32, r the word's row.  The offset
  // carries + 1024 - 1024 b (the base is 1 KB low) so that it stays >= 0.
  uint32_t sp_v[2] = {0u, 0u};            // [b ^ first_half]: pieces i even / odd
  const int sp_first = (wave * kDma) & 1;
  auto sp_voff = [&](int r, int b) -> uint32_t {
    const int c = 64 * b + lane;
    const int mp = 4 * (c >> 4) + (c & 3);
    const int hp = (((c >> 2) & 3) - (mp >> 3)) & 3;
    const int ms = r & 31;
    const int chunk = (ms >> 2) * 16 + 4 * ((hp + (ms >> 3)) & 3) + (ms & 3);
    return (uint32_t)((r >> 5) * kTile + chunk * 16 + 1024 - 1024 * b);
  };
  // tile row of this lane's chunk in half b (what sp_voff calls m')
  auto sp_row = [&](int b) -> int {
    const int c = 64 * b + lane;
    return 4 * (c >> 4) + (c & 3);
  };
  auto dma_piece = [&](auto ic, const unsigned char* src, uint32_t dst) {
    constexpr int i = decltype(ic)::value;
    // (the immediate offset -- added to the global address AND to the LDS
    // address -- reaches 4 KB: a fresh pair of bases every fourth piece)
    b3_dma<(i & 3) * 1024>(src + (i >> 2) * 4096,
                           dst + (uint32_t)((i >> 2) * 4096),
                           SP ? sp_v[i & 1] : voff);
  };
  // SP: the 32 row indices of a tile -> ring slot (8-lane DMA like the labels)
  auto dma_rows = [&](int64_t tile, int slot) {
    const uint32_t off = (uint32_t)(wave * 8 + (lane & 7)) * 4u;
    const int32_t* src = uniform_ptr(rsrc + tile * kB3Rows);
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_bfm_b64 exec, 8, 0\n\t"
        "global_load_lds_dword %0, %1\n\t"
        "s_mov_b64 exec, -1"
        :
        : "v"(off), "s"(src),
          "s"(sr_addr + (uint32_t)((slot & 3) * 128 + wave * 32))
        : "memory");
  };
  // the tile's 32 labels: wave w brings rows 8 w .. 8 w + 7 (8-lane DMA),
  // clamped to the last row of X (masked / met by zero rows of the image)
  const int lane_y = wave * 8 + (lane & 7);
  auto dma_labels = [&](int64_t tile, uint32_t dst) {
    if constexpr (PK) {
      const float* src = uniform_ptr(ysrc + tile * kB3Rows);
      const uint32_t d0 = dst + (uint32_t)(wave * 4096);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        b3_dma<0>(reinterpret_cast<const unsigned char*>(src),
                  d0 + (uint32_t)(g * 1024), pk_voff[g]);
      return;
    }
    const int64_t left = N - 1 - tile * kB3Rows;      // >= 0: the tile exists
    const int last = left < kB3Rows - 1 ? (int)left : kB3Rows - 1;
    const uint32_t off = (uint32_t)(lane_y < last ? lane_y : last) * 4u;
    const float* src = uniform_ptr(ysrc + tile * kB3Rows);
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_bfm_b64 exec, 8, 0\n\t"
        "global_load_lds_dword %0, %1\n\t"
        "s_mov_b64 exec, -1"
        :
        : "v"(off), "s"(src), "s"(dst + (uint32_t)(wave * 32))
        : "memory");
  };

  // ---- per-lane LDS offsets inside a tile ----------------------------------
  // GEMM 1, A: chunk (row lo, half hi, par) of block ks / 2, par = ks % 2
  const uint32_t a_lane[2] = {(uint32_t)(b3_chunk(lo, hi, 0) * 16),
                              (uint32_t)(b3_chunk(lo, hi, 1) * 16)};
  // GEMM 2, B: the transposing read r of k-step s; this lane SUPPLIES the
  // 8 bytes (row 16 s + 8 r + 4 (lane / 32) + (lane % 16) / 4, features
  // 4 (lane % 4) .. + 3 of sub-block (lane / 16) % 2)
  uint32_t t_lane[2][2];
  {
    const int hq = lane >> 5, g1 = (lane >> 4) & 1, jq = (lane & 15) >> 2,
              cq = lane & 3;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int m = 16 * s + 8 * r + 4 * hq + jq;
        t_lane[s][r] =
            (uint32_t)(b3_chunk(m, cq >> 1, g1) * 16 + (cq & 1) * 8);
      }
  }
  // labels of this lane's rows: group g = rows 8 g + 4 hi .. + 3
  const uint32_t y_lane =
      PK ? (uint32_t)(wave * 4096 + lo * 128) : (uint32_t)(hi * 16);

  // (zeroed BY an MFMA, 0 * 0 + 0, straight in their AGPRs: zeros written
  // by the compiler arrive through a second set of D/2 registers that then
  // stays allocated)
  f16v G[NB];
  {
    const u4 z = u4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int t = 0; t < NB; ++t)
      asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, 0"
                   : "=&a"(G[t])
                   : "v"(z));
  }
  double ll_lane = 0.0;   // tile sums in float64 (csrc/linear_bernoulli.hip)

  // buffer rotation: cur = GEMM 1's tile, prev = GEMM 2's, next = DMA target
  uint32_t b_cur = 0, b_prev = 2 * kTile, b_next = kTile;
  uint32_t y_cur = 0, y_prev = 2 * kYBuf, y_next = kYBuf;

  // prologue: tile 0 -> buffer 0
  // (SP: the row indices of the first two tiles by plain loads -- the ring
  // is fed from iteration 0 on -- tile 0's offsets now, tile 1's behind its DMA)
  auto sp_load = [&](int64_t tile) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int b = (sp_first + k) & 1;
      sp_v[k] = sp_voff(rsrc[tile * kB3Rows + sp_row(b)], b);
    }
  };
  if constexpr (SP) sp_load(t_first);
  static_for<kDma>([&](auto ic) { dma_piece(ic, xsrc, dst_wave + b_cur); });
  dma_labels(t_first, sy_addr + y_cur);
  if constexpr (SP) sp_load(t_first + (T > 1 ? 1 : 0));
  if constexpr (PK && kYAhead > 1) {  // (past the last tile: clamped, harmless)
#pragma unroll
    for (int a = 1; a < kYAhead; ++a)
      dma_labels(t_first + (a < T ? a : (T > 0 ? T - 1 : 0)),
                 sy_addr + y_slot(a));
  }

  f16v Sa, Sb;              // GEMM 1's accumulator chains
  float Sp[16];             // the logits of the tile before
#pragma unroll
  for (int r = 0; r < 16; ++r) Sp[r] = 0.f;
  u4 opnd[3];               // the streamed operand's planes: A of GEMM 1 / B of GEMM 2
  u4 Rp[2][3];              // residual planes: [k-step][plane], A of GEMM 2
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int p = 0; p < 3; ++p) Rp[s][p] = u4{0u, 0u, 0u, 0u};
  typedef float f2v __attribute__((ext_vector_type(2)));
  f2v ypair = f2v{0.f, 0.f};   // the labels of the pair the stage works on
  uint32_t y_addr = 0;         // the labels of tile it-1 for this lane
  // OP 2: the wave's 32 rows of W are (chain, class) pairs, a chain's classes
  // in 2^GL consecutive lanes (csrc/lb_ops.h)
  const CatLane cat = cat_lane(lo, n_classes, GL);

  int rows_prev = kB3Rows;   // valid rows of the tile the element-wise stage works on
  float ll_tile = 0.f;
  float e0 = 0.f, e1 = 0.f, r0 = 0.f, r1 = 0.f;   // live between the pieces of one pair
  uint32_t a_addr0 = 0, a_addr1 = 0;
  const unsigned char* src_next = xsrc;
  uint32_t dst_next = dst_wave;
  int64_t lab_next = t_first;
  int64_t row_next = t_first;   // SP: the tile whose row indices go out next
  int row_slot = 0;

  // ---- the boundary between two iterations (placed in front of the last
  // MFMAs of the iteration that ends): this wave's DMA of tile it+1 has
  // landed, barrier -- everyone's has, and everyone has issued AND landed its
  // last reads of tile it-1 -- buffers rotate, and the first reads of
  // iteration it+1 (A operand of k-step 0, the labels of tile `it`) go out
  // under those last MFMAs ------------------------------------------------------
  auto boundary = [&](int it) {
    if constexpr (PK && kYAhead == 2)
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (PK && kYAhead == 3)
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {  // prev <- cur <- next <- prev
      const uint32_t t = b_prev;
      b_prev = b_cur;
      b_cur = b_next;
      b_next = t;
      const uint32_t u = y_prev;
      y_prev = y_cur;
      y_cur = y_next;
      y_next = u;
    }
    const int64_t left = N - (t_first + it) * kB3Rows;   // rows of tile `it`
    rows_prev = left < kB3Rows ? (int)left : kB3Rows;
    a_addr0 = sx_addr + b_cur + a_lane[0];
    a_addr1 = sx_addr + b_cur + a_lane[1];
    // (the labels of a pair -- two consecutive rows, 8 bytes -- are read by
    // the pair's first piece: four registers of labels held per tile would
    // not fit two waves per SIMD at D = 128)
    y_addr = sy_addr + (PK ? y_slot(it) : y_prev) + y_lane;
    __builtin_amdgcn_sched_barrier(0);
  };
  // where iteration `it` sends tile it+1 (clamped to the last one: a
  // harmless re-load) and its labels
  auto plan_dma = [&](int it) {
    const int nxt = it + 1 < T ? it + 1 : T - 1;
    src_next = SP ? xsrc : xsrc + (int64_t)nxt * kTile;
    dst_next = dst_wave + b_next;
    lab_next = t_first + nxt;
    if constexpr (SP) {
      // the rows of tile `nxt` came into ring slot it + 1 during iteration
      // it - 1 (iteration 0: by the prologue's plain loads)
      if (it > 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int b = (sp_first + k) & 1;
          const int r = *reinterpret_cast<const ZS_LDS int*>((uintptr_t)(
              sr_addr + (uint32_t)(((it + 1) & 3) * 128 + sp_row(b) * 4)));
          sp_v[k] = sp_voff(r, b);
        }
      }
      row_next = t_first + (it + 2 < T ? it + 2 : T - 1);
      row_slot = it + 2;
    }
    if constexpr (PK) {       // the counts of tile it + kYAhead, into its slot
      const int far = it + kYAhead < T ? it + kYAhead : T - 1;
      lab_next = t_first + far;
      y_next = y_slot(it + kYAhead);
    }
  };

  // ---- element-wise stage of tile it-1, in pieces ------------------------------
  // lane = chain wave*32 + lo; Sp[j] = logit of row 8 (j/4) + 4 hi + j%4.
  // Pair pr = registers 2 pr, 2 pr + 1 -> register pr % 4 of k-step pr / 4.
  // A piece is held in its gap from both sides: its inputs pass through an
  // empty asm statement in front of it and its results through one behind it
  // (asm volatile statements keep their order -- the MFMAs are such -- while
  // plain arithmetic is the compiler's to move: left alone it sinks the whole
  // stage in front of GEMM 2, its only consumer).
  // (a transcendental takes 16 clocks of the vector ALU, half an MFMA gap:
  // the hot form -- Bernoulli, gradient only -- has one per piece)
  constexpr int kPre = (OP == 0 && !LL) ? 4 : 2;   // pieces in front of the split
  constexpr int kSubs = kPre + 3;
  auto ew_piece = [&](auto cc) {
    constexpr int c = decltype(cc)::value;
    constexpr int pr = c / kSubs, sub = c % kSubs;
    constexpr int j0 = 2 * pr, j1 = 2 * pr + 1;
    constexpr int s = pr / 4, q = pr % 4;
    if constexpr (sub < kPre) {
      if constexpr (sub == 0) {
        // rows 8 (j0/4) + 4 hi + j0 % 4 and the next one
        if (!(ZS_B3_SKIP & 4))
          ypair = *reinterpret_cast<const ZS_LDS f2v*>((uintptr_t)(
              y_addr +
              (uint32_t)(PK ? ((pk_rot + 2 * (j0 >> 2)) & 7) * 16 + (j0 & 3) * 4
                            : (8 * (j0 >> 2) + (j0 & 3)) * 4)));
      }
      float y0 = ypair[0], y1 = ypair[1];
      const int n0 = 8 * (j0 >> 2) + 4 * hi + (j0 & 3);
      const bool v0 = n0 < rows_prev, v1 = n0 + 1 < rows_prev;
      if constexpr (OP == 0 && !LL) {
        // y - sigmoid(l), sigmoid(l) = 1 / (1 + 2^(-l log2 e))
        if constexpr (sub == 0) {
          pin(Sp[j0]);
          e0 = __builtin_amdgcn_exp2f(-1.4426950408889634f * Sp[j0]);
          pin(e0);
        } else if constexpr (sub == 1) {
          pin(Sp[j1]);
          e1 = __builtin_amdgcn_exp2f(-1.4426950408889634f * Sp[j1]);
          pin(e1);
        } else if constexpr (sub == 2) {
          pin(e0);
          e0 = __builtin_amdgcn_rcpf(1.0f + e0);
          pin(e0);
        } else {
          pin(e0, e1);
          r0 = y0 - e0;
          r1 = y1 - __builtin_amdgcn_rcpf(1.0f + e1);
          pin(r0, r1);
        }
      } else if constexpr (OP == 1 && !LL) {
        if constexpr (sub == 0) {
          pin(Sp[j0], Sp[j1]);
          e0 = __builtin_amdgcn_rcpf(Sp[j0]);
          e1 = __builtin_amdgcn_rcpf(Sp[j1]);
          pin(e0, e1);
        } else {
          pin(e0, e1);
          r0 = (v0 && y0 != 0.f) ? y0 * e0 : 0.f;
          r1 = (v1 && y1 != 0.f) ? y1 * e1 : 0.f;
          pin(r0, r1);
        }
      } else if constexpr (sub == 0) {
        pin(Sp[j0]);
        if constexpr (OP == 2)
          r0 = categorical_residual_c<GL, LL>(Sp[j0], y0, cat, v0, ll_tile);
        else
          r0 = lb_residual<OP, LL>(Sp[j0], y0, cat, v0, ll_tile);
        pin(r0);
        pin(ll_tile);
      } else {
        pin(Sp[j1]);
        if constexpr (OP == 2)
          r1 = categorical_residual_c<GL, LL>(Sp[j1], y1, cat, v1, ll_tile);
        else
          r1 = lb_residual<OP, LL>(Sp[j1], y1, cat, v1, ll_tile);
        pin(r1);
        pin(ll_tile);
      }
    } else if constexpr (sub == kPre) {
      pin(r0, r1);
      unsigned ph = pk_bf16(r0, r1);
      pin(ph);
      r0 -= bf_lo(ph);
      r1 -= bf_hi(ph);
      pin(r0, r1);
      Rp[s][0][q] = ph;
    } else if constexpr (sub == kPre + 1) {
      pin(r0, r1);
      unsigned pm = pk_bf16(r0, r1);
      pin(pm);
      r0 -= bf_lo(pm);
      r1 -= bf_hi(pm);
      pin(r0, r1);
      Rp[s][1][q] = pm;
    } else {
      pin(r0, r1);
      unsigned pl = pk_bf16(r0, r1);
      pin(pl);
      Rp[s][2][q] = pl;
    }
  };
  constexpr int kPieces = 8 * kSubs;
  // pieces [P0, P1) spread over the gaps [G0, G1) of a phase: gap g takes
  // those whose share falls on it
  auto ew_gaps = [&](auto gc, auto p0c, auto p1c, auto g0c, auto g1c) {
    constexpr int g = decltype(gc)::value, P0 = decltype(p0c)::value,
                  P1 = decltype(p1c)::value, G0 = decltype(g0c)::value,
                  G1 = decltype(g1c)::value;
    if constexpr (g >= G0 && g < G1 && !(ZS_B3_SKIP & 1)) {
      constexpr int first = P0 + (g - G0) * (P1 - P0) / (G1 - G0);
      constexpr int last = P0 + (g - G0 + 1) * (P1 - P0) / (G1 - G0);
      if constexpr (last > first) {
        static_for<last - first>([&](auto dc) {
          ew_piece(std::integral_constant<int, first + decltype(dc)::value>{});
        });
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // pairs 0 .. kPairs1-1 ride in GEMM 1 (0 .. 3 are k-step 0's rows: they must),
  // the rest in the first half of GEMM 2
  // (shader clocks per 32-row tile, profiles/r05g_b3_phase_split.txt: the
  // stage costs ~600 clocks wherever it rides -- a wave's VALU work is only
  // partly hidden under its own bf16 MFMAs -- and the split moves 1-2 %)
#ifdef ZS_B3_PAIRS1
  constexpr int kPairs1 = ZS_B3_PAIRS1;
#else
  constexpr int kPairs1 = D >= 192 ? 5 : 6;
#endif
  static_assert(kPairs1 >= 4 && kPairs1 <= 8, "k-step 0's rows before GEMM 2");
  template_int<0> c0;
  template_int<kPairs1 * kSubs> cHalf;
  template_int<kPieces> cAll;

  // tile it+1 and its labels: piece i (kDma of them, then the labels)
  auto dma_step = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < kDma) {
      if (!(ZS_B3_SKIP & 2)) dma_piece(ic, src_next, dst_next);
    } else if constexpr (i == kDma) {
      if (!(ZS_B3_SKIP & 4)) dma_labels(lab_next, sy_addr + y_next);
      if constexpr (SP) dma_rows(row_next, row_slot);
    }
  };

  // B operand of GEMM 2's step (s, nb), plane p: two transposing reads
  uint32_t t00 = 0, t01 = 0, t10 = 0, t11 = 0;
  auto plan_b = [&]() {
    const uint32_t tb = sx_addr + b_prev;
    t00 = tb + t_lane[0][0], t01 = tb + t_lane[0][1];
    t10 = tb + t_lane[1][0], t11 = tb + t_lane[1][1];
  };
  auto read_b = [&](auto stc, auto pc) -> u4 {
    constexpr int st = decltype(stc)::value, p = decltype(pc)::value;
    constexpr int s = st / NB, nb = st % NB;     // s-major: see gemm2
    constexpr int off = p * kPlane + nb * 2048;
    const u2 q0 = lds_tr(s == 0 ? t00 : t10, off);
    const u2 q1 = lds_tr(s == 0 ? t01 : t11, off);
    return u4{q0[0], q0[1], q1[0], q1[1]};
  };

  // ---- GEMM 1 of tile `it`: S[n, c] over the wave's 32 chains; in its gaps
  // the first half of the element-wise stage of tile it-1 (pairs 0 .. 3: the
  // rows of GEMM 2's k-step 0) and, in the last k-step's, the first operand
  // reads of GEMM 2 -------------------------------------------------------------
  auto gemm1 = [&](auto with_ew) {
    constexpr bool kEw = decltype(with_ew)::value;
    if constexpr (kEw) plan_b();
    static_for<KS>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      static_for<6>([&](auto tc) {
        constexpr int term = decltype(tc)::value;
        constexpr int pa = kTermX[term], pb = kTermR[term];
        constexpr int g = ks * 6 + term;
        constexpr bool first = ks == 0 && term < NACC;
        f16v& acc = (NACC == 2 && (term & 1)) ? Sb : Sa;
        if (!(ZS_B3_SKIP & 8)) {
        if constexpr (pb == 0) {
          if constexpr (first) mfma_bv0(acc, opnd[pa], wh[ks]);
          else mfma_bv(acc, opnd[pa], wh[ks]);
        } else if constexpr (pb == 1) {
          if constexpr (first) mfma_ba0(acc, opnd[pa], wm[ks]);
          else mfma_ba(acc, opnd[pa], wm[ks]);
        } else if constexpr (wl_agpr(ks)) {
          mfma_ba(acc, opnd[pa], wl[ks]);
        } else {
          mfma_bv(acc, opnd[pa], wl[ks]);
        }
        }
        // behind a plane's last use: the next k-step's plane -- behind the
        // last k-step GEMM 2's first operand
        if constexpr (term == kLastUse[pa]) {
          if constexpr (ks + 1 < KS) {
            constexpr int P = (ks + 1) >> 1;
            opnd[pa] = lds_u4(((ks + 1) & 1) ? a_addr1 : a_addr0,
                              pa * kPlane + P * 2048);
          } else if constexpr (kEw) {
            opnd[pa] = read_b(c0, template_int<pa>{});
          }
        }
        if constexpr (ZS_B3_DMA_IN_GEMM1 && kEw && term == 3 && ks <= kDma)
          dma_step(template_int<ks>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kEw)
          ew_gaps(template_int<g>{}, c0, cHalf, c0, template_int<6 * KS - 3>{});
      });
    });
  };

  // logits of the tile GEMM 1 just finished -> Sp (the accumulators are free
  // for the next GEMM 1); placed where its last MFMA is long done
  auto take_logits = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NACC == 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) Sp[r] = Sa[r] + Sb[r];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) Sp[r] = Sa[r];
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- GEMM 2 of tile it-1: G[c, f] += R^T X, k-step 0 (rows 0 .. 15) over
  // all feature blocks, then k-step 1: the second half of the element-wise
  // stage (pairs 4 .. 7 = k-step 1's rows) and the DMA of tile it+1 ride in
  // the first half's gaps, the logits of tile `it` and the iteration boundary
  // in the second's ---------------------------------------------------------------
  auto gemm2 = [&](int it) {
    constexpr int kSteps = 2 * NB;
    constexpr int kHalfGaps = 6 * NB;
    // DMA pieces: one every kEvery gaps of the first half, from gap 1
    constexpr int kEvery = (kHalfGaps - 1) / (kDma + 1);
    static_assert(kEvery >= 1, "the DMA pieces fit GEMM 2's first half");
    static_for<kSteps>([&](auto stc) {
      constexpr int st = decltype(stc)::value;
      constexpr int s = st / NB, nb = st % NB;
      if constexpr (st == kSteps - 1) boundary(it);
      static_for<6>([&](auto tc) {
        constexpr int term = decltype(tc)::value;
        constexpr int pb = kTermX[term], pa = kTermR[term];
        constexpr int g = st * 6 + term;
        if (!(ZS_B3_SKIP & 16)) mfma_g(G[nb], Rp[s][pa], opnd[pb]);
        // behind a plane's last use: the next step's plane -- behind the last
        // step's the A operand of the next iteration's GEMM 1 (the boundary
        // in front of this step has rotated the buffers)
        if constexpr (term == kLastUse[pb]) {
          if constexpr (st + 1 < kSteps)
            opnd[pb] = read_b(template_int<st + 1>{}, template_int<pb>{});
          else
            opnd[pb] = lds_u4(a_addr0, pb * kPlane);
        }
        if constexpr (!(ZS_B3_DMA_IN_GEMM1) && g < kHalfGaps && g >= 1 &&
                      (g - 1) % kEvery == 0 && (g - 1) / kEvery <= kDma)
          dma_step(template_int<(g - 1) / kEvery>{});
        __builtin_amdgcn_sched_barrier(0);
        ew_gaps(template_int<g>{}, cHalf, cAll, c0,
                template_int<kHalfGaps - 6>{});
      });
      if constexpr (st == NB + 1) take_logits();
    });
  };
  static_assert(NB >= 2, "GEMM 2's last step is not its first of k-step 1");

  if (T > 0) {
    // iteration 0: GEMM 1 of the first tile alone, the DMA of the second
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    a_addr0 = sx_addr + b_cur + a_lane[0];
    a_addr1 = sx_addr + b_cur + a_lane[1];
    opnd[0] = lds_u4(a_addr0, 0);
    opnd[1] = lds_u4(a_addr0, kPlane);
    opnd[2] = lds_u4(a_addr0, 2 * kPlane);
    plan_dma(0);
    static_for<kDma + 1>([&](auto ic) { dma_step(ic); });
    __builtin_amdgcn_sched_barrier(0);
    gemm1(std::false_type{});
    mfma_drain8(Sa);
    if constexpr (NACC == 2) mfma_drain8(Sb);
    take_logits();
    boundary(0);
    opnd[0] = lds_u4(a_addr0, 0);
    opnd[1] = lds_u4(a_addr0, kPlane);
    opnd[2] = lds_u4(a_addr0, 2 * kPlane);
    __builtin_amdgcn_sched_barrier(0);
    // iterations 1 .. T: GEMM 1 of tile `it` (of the last tile once more, unused,
    // in iteration T) around the element-wise stage of tile it-1, then GEMM 2
    // of tile it-1.  Nothing in the loop is conditional: the accumulators do
    // not pass through a phi (hipcc keeps a second set for one and copies).
    int it = 1;
#ifdef ZS_B3_TIMING  // debug: shader clocks per phase, every wave of block 0
    long long tacc[2] = {0, 0};
    long long tmark = __builtin_readcyclecounter();
#define ZS_B3_MARK(i)                                  \
  {                                                     \
    __builtin_amdgcn_sched_barrier(0);                  \
    const long long _t = __builtin_readcyclecounter();  \
    tacc[i] += _t - tmark;                              \
    tmark = _t;                                         \
    __builtin_amdgcn_sched_barrier(0);                  \
  }
#else
#define ZS_B3_MARK(i)
#endif
    do {
      ll_tile = 0.f;
      plan_dma(it);
      gemm1(std::true_type{});
      ZS_B3_MARK(0)
      gemm2(it);
      ZS_B3_MARK(1)
      if (LL) ll_lane += (double)ll_tile;
    } while (++it <= T);
#ifdef ZS_B3_TIMING
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
      gW[wave * 4 + 0] = (float)tacc[0];
      gW[wave * 4 + 1] = (float)tacc[1];
      gW[wave * 4 + 2] = (float)T;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) return;
#endif
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // ---- epilogue -------------------------------------------------------------
  // G[nb][j]: chain wave*32 + 8 (j/4) + 4 hi + j%4, feature 32 nb + lo
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    asm volatile("s_nop 11" : "+a"(G[nb]));
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int pos = wave * 32 + 8 * (j >> 2) + 4 * hi + (j & 3);
      if (pos < n_valid)
        gW[(row_base + pos * row_stride) * ldw + nb * 32 + lo] = G[nb][j];
    }
  }
  if (LL) {
    const double tot = ll_lane + __shfl_xor(ll_lane, 32, 64);
    const int pos = wave * 32 + lo;
    if (hi == 0 && pos < n_valid)
      ll[row_base + pos * row_stride] = (float)tot;
  }
}

// csrc/linear_bernoulli.hip
int lb_reduce_splits(const float* ws, int64_t C, int64_t ldw, int S, float* ll,
                     float* gW, hipStream_t s);

template <int D, int OP, int GL = 0, bool PK = false, bool SP = false>
static int launch_b3(const float* W, const unsigned char* Ximg, const float* y,
                     int64_t yc_rows, int64_t ldy, int64_t C, int64_t N,
                     float* ll, float* gW, hipStream_t s, int n_splits,
                     float* workspace, int doc_major, int n_classes = 0,
                     const int32_t* sp_rows = nullptr,
                     const int64_t* sp_off = nullptr) {
  constexpr int kTile = 3 * (D / 32) * 2048;
  // (PK: four buffers of counts where they are fetched two tiles ahead)
  const size_t lds =
      (size_t)3 * kTile +
      (PK ? (D <= 128 ? ZS_B3_PK_AHEAD + 2 : 3) * 16384 : 3 * 128) +
      (SP ? 4 * 128 : 0);
  // accumulator chains of GEMM 1: two cost 16 registers and buy nothing
  // measurable (dependent 32x32x16 MFMAs issue back to back); one where two
  // waves per SIMD need the registers
#ifndef ZS_B3_NACC
#define ZS_B3_NACC(D) ((D) <= 128 ? 1 : 2)
#endif
  auto kll = linear_b3_kernel<D, OP, true, ZS_B3_NACC(D), GL, PK, SP>;
  auto kg = linear_b3_kernel<D, OP, false, ZS_B3_NACC(D), GL, PK, SP>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(kll),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kg),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(LDS)");
    attr = true;
  }
  const int gx = doc_major
                     ? (int)(((C / yc_rows + kB3Chains - 1) / kB3Chains) * yc_rows)
                     : (int)((C + kB3Chains - 1) / kB3Chains);
  const int S = (n_splits > 1 && workspace) ? n_splits : 1;
  float* ll_out = S > 1 ? workspace : ll;
  float* g_out = S > 1 ? workspace + (int64_t)S * C : gW;
  const dim3 grid(gx, S);
  if (ll)
    hipLaunchKernelGGL(kll, grid, dim3(256), lds, s, W, Ximg, y, yc_rows, ldy,
                       C, N, (int64_t)D, ll_out, g_out, doc_major, n_classes,
                       sp_rows, sp_off);
  else
    hipLaunchKernelGGL(kg, grid, dim3(256), lds, s, W, Ximg, y, yc_rows, ldy, C,
                       N, (int64_t)D, ll_out, g_out, doc_major, n_classes, sp_rows,
                       sp_off);
  ZS_LAUNCH_CHECK("linear_b3_kernel launch");
  if (S > 1) return lb_reduce_splits(workspace, C, (int64_t)D, S, ll, gW, s);
  return ZSHMC_OK;
}

static inline bool b3_width(int64_t n) { return n == 64 || n == 128 || n == 192 || n == 256; }

}  // namespace zshmc
