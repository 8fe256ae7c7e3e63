// Fused dense-logit Bernoulli log-likelihood + gradient on the fp32 matrix
// cores of gfx950 (BASELINE config 3: Bayesian logistic regression).
//
//   logits[c, n] = sum_d W[c, d] * X[n, d]                 (user model: w @ X^T)
//   ll[c]   = sum_n  y_n*l - max(l,0) - log1p(exp(-|l|))    Bernoulli._log_prob,
//             reference zhusuan/distributions/univariate.py:398-403
//             (= -sigmoid_cross_entropy_with_logits) summed by group_ndims=1,
//             distributions/base.py:302-304
//   gW[c,:] = sum_n (y_n - sigmoid(l)) * X[n, :]            what tf.gradients
//             (hmc.py:430-432) yields through the matmul
//
// The reference materialises logits [C, N] (131 GB at config 3) and runs two
// GEMMs plus ~10 element-wise passes per gradient evaluation.  Here the two
// GEMMs are fused flash-attention style: a workgroup owns 64 chains, streams
// X in 32-row tiles through LDS (double buffered), computes the 32x64 logits
// tile on v_mfma_f32_32x32x2_f32 (exact fp32, so the 1 % acceptance parity is
// not at risk), applies the sigmoid residual in the accumulator registers,
// and feeds those registers straight back as the A operand of the second
// MFMA chain (S is computed transposed, so the C/D layout of the first GEMM
// IS the A layout of the second -- no shuffle, no LDS round trip).  Logits
// never leave registers.  Roofline: MFMA (fp32 157 TFLOP/s);
// 4*N*D*C flop per call.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "lb_ops.h"

namespace zshmc {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kMC = 64;  // chains per workgroup

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

// global -> LDS, BYTES (4, 8 = 2x4, 16) per lane, LDS dest = dst + lane*BYTES
template <int BYTES>
__device__ __forceinline__ void lds_dma_row(const float* src, uint32_t dst,
                                            uint32_t lane) {
  if constexpr (BYTES == 16) {
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
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %0, %1"
        :
        : "v"(voff), "s"(src), "s"(dst)
        : "memory");
    if constexpr (BYTES == 8)
      asm volatile(
          "s_mov_b32 m0, %2\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dword %0, %1 offset:256"
          :
          : "v"(voff), "s"(src), "s"(dst)
          : "memory");
  }
}

// ---------------------------------------------------------------------------
// 64-row tiles, W in registers, residual exchange between sibling waves.
// (The first form of this kernel -- 32-row tiles, K split over two waves,
// partial logits through LDS, 38 % of peak -- is gone; profiles/r01e_* has
// its numbers.)
//
// The 4 waves of a workgroup are (a, b): chain block a (32 chains) x half b.
//   phase 1  S'[n, i] = sum_d X[n,d] W[i,d] for the wave's 32 ROWS (b = row
//            block), full K = D in one accumulator chain.  The wave's chain
//            block of W stays in registers for the whole kernel (D/2 VGPRs,
//            the B operand): no LDS traffic for W, no exchange of partial
//            logits, D/2 MFMAs.
//   residual R' = y - sigmoid(S') and the log-likelihood terms on the
//            accumulator registers (3 hardware transcendentals per element),
//            issued under the MFMAs of phase 3a; R' is also parked in LDS.
//   phase 3  G[i, f] += sum_n R'[n, i] X[n, f] over ALL 64 rows but only the
//            wave's HALF of the features (b = feature half): 3a takes the
//            wave's own 32 rows straight from the residual registers (the
//            C/D layout of phase 1 IS the A layout of phase 3), 3b the sibling
//            wave's rows from LDS after the mid-tile barrier.  D/4 + D/4 MFMAs
//            on D/4 accumulators; every wave stores its own slice of gW.
// LDS: the double-buffered X tile (2 x 64 x (D+4) floats), streamed by
// LDS-DMA one padded row per instruction, issued between the MFMAs of phase 1;
// 16 KB for the residual exchange.  Two barriers per tile.
//
// OP selects the element-wise stage between the two GEMMs:
//   OP = 0  Bernoulli with dense logits (config 3): term = l*y - max(l,0) -
//           log1p(exp(-|l|)), residual = y - sigmoid(l), y[n] per data row.
//   OP = 1  UnnormalizedMultinomial over a mixture (config 5, the LNTM E-step,
//           lntm_mcem.py:33-48): W = theta [rows, K], X = phi^T [V, K], the
//           "logits" are log(theta.phi) (multivariate.py:435-443 with
//           normalize_logits = False): term = x*log(S), residual = x / S with
//           the counts x[c, n] streamed from `yc` [C, N] (chain-major).
//   OP = 2  Categorical with dense logits (softmax regression,
//           univariate.py:496-548): the rows of W are (chain, class) pairs,
//           `n_classes` classes in groups of 2^cls_log2 consecutive rows; term =
//           l[y] - logsumexp(l), residual = [k == y] - softmax, the softmax over
//           the group's lanes of the accumulator (csrc/lb_ops.h); y[n] = the
//           label of data row n as a float.
#ifndef ZS_LB_BUF
#define ZS_LB_BUF(D) ((D) <= 128 ? 1 : 2)
#endif
// D = 256 with gradient: the next tile's DMA rows under phase 3b (independent
// accumulators) instead of in front of the first 16 steps of phase 1 (one
// dependent chain) -- what pays in csrc/linear_bernoulli_wide.hip was measured
// here and LOSES: 128.0 against 129.3 TFLOP/s (Bernoulli), 125.5 against 130.0
// (multinomial, K = 256), profiles/r03cc_dma_phase3_ab.txt.  Off.
#ifndef ZS_LB_DMA_PHASE3
#define ZS_LB_DMA_PHASE3 0
#endif
#ifndef ZS_LB_MINW  // min waves per SIMD: D = 64 fits three workgroups per CU
#define ZS_LB_MINW(D) ((D) == 64 ? 3 : 1)
#endif
// LL = false (GRAD only): the log-likelihood terms are not formed -- the L - 1
// interior evaluations of a leapfrog trajectory need the gradient alone
// (hmc.py:348-372; the log-joint is read at its two ends, :46-61), and the
// log (one of three transcendentals) + 5 VALU per element are ~40 % of the
// element-wise stage a lone wave per SIMD cannot hide under its own MFMAs.
template <int D, bool GRAD, int OP, bool LL = true>
__global__ __launch_bounds__(256, ZS_LB_MINW(D)) void linear_bernoulli_kernel_v2(
    const float* __restrict__ W, const float* __restrict__ X,
    const float* __restrict__ y, const float* __restrict__ yc,
    int64_t yc_rows, int64_t ldy, int64_t C, int64_t N, int64_t ldw,
    int64_t ldx, float* __restrict__ ll, float* __restrict__ gW,
    int doc_major, int n_classes, int cls_log2) {
  constexpr int LD = D + 4;          // padded LDS row: conflict-free b128 reads
  constexpr int kRows = 64;          // data rows per tile
  constexpr int KK = D / 8;          // phase-1 steps of 4 MFMAs (8 features)
  constexpr int HALF = D / 2;        // features per wave in phase 3
  constexpr int FB = HALF / 32;      // 32-wide feature blocks per half (1,2,4)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // X tile buffers: two (the DMA of tile t+1 runs under the compute of tile
  // t) where only one workgroup fits a CU anyway; ONE for D <= 128, where the
  // smaller footprint (50 KB / 34 KB) lets a second (third) workgroup share
  // the CU and its MFMAs fill this one's bubbles -- DMA latency included
  constexpr int kBuf = ZS_LB_BUF(D);
  float* __restrict__ sX = reinterpret_cast<float*>(smem);  // [kBuf][kRows][LD]
  float* __restrict__ sY = sX + kBuf * kRows * LD;          // [2][kRows]
  float* __restrict__ sR = sY + 2 * kRows;                  // [4][4][64][4]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int a = wave >> 1, b = wave & 1;
  const int lo = lane & 31, hi = lane >> 5;
  // The 64 rows of W this workgroup owns: consecutive (row_stride 1), or --
  // OP 1, doc_major: rows r = chain * yc_rows + doc of the topic model's
  // [n_chains, n_docs] chain axes -- 64 CHAINS OF ONE DOCUMENT (row_stride =
  // n_docs).  With consecutive rows every lane gathers its own document's
  // counts (32 different rows of the [n_docs, V] matrix per instruction, 2 TB
  // of gathered bytes per launch at BASELINE configs[4],
  // profiles/r03e_native_full_shape_rocprofv3_summary.txt); with one document
  // per workgroup the same four loads per tile are broadcasts of one row.
  int64_t row_base = (int64_t)blockIdx.x * kMC, row_stride = 1;
  int64_t n_valid = C - row_base;
  if (OP == 1 && doc_major) {
    const int64_t grp = blockIdx.x / yc_rows, doc = blockIdx.x % yc_rows;
    row_base = grp * kMC * yc_rows + doc;
    row_stride = yc_rows;
    n_valid = C / yc_rows - grp * kMC;
  }
  n_valid = n_valid < kMC ? n_valid : kMC;
  // row of position i (0..63) of the tile; positions past the end re-read the
  // last valid row (their results are never stored)
  auto row_at = [&](int i) -> int64_t {
    return row_base + (int64_t)(i < n_valid ? i : (int)n_valid - 1) * row_stride;
  };
  // OP 1: counts rows are 16-B aligned and zero-padded to 4-float groups
  const bool yc_vec = OP == 1 && (ldy & 3) == 0 && ldy >= ((N + 3) & ~3ll) &&
                      (reinterpret_cast<uintptr_t>(yc) & 15) == 0;

  // ---- this wave's W block -> registers (B operand: k-slot = lane half) ----
  float wreg[KK * 4];
  {
    const int64_t cr = row_at(a * 32 + lo);
    const float* __restrict__ wrow = W + cr * ldw + hi * 4;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const f4 v = *reinterpret_cast<const f4*>(wrow + kk * 8);
#pragma unroll
      for (int m = 0; m < 4; ++m) wreg[kk * 4 + m] = v[m];
    }
  }

  // ---- X tile: global -> LDS by DMA, one padded row per instruction --------
  // (global_load_lds writes lane-linear: a row of D floats is D/64 dwords per
  // lane, and the 4-float pad sits between rows, i.e. between instructions).
  // Wave w moves rows 16w .. 16w+15; rows past N re-read row N-1 (masked in
  // the residual).  hipcc does not count these loads: the `s_waitcnt
  // vmcnt(0)` in front of the tile barrier lands them.
  constexpr int kDmaB = D / 16;  // bytes per lane per row: 16 (D=256), 8, 4
  // 16 rows per wave and tile, spread over the KK phase-1 steps
  constexpr int kDmaPer = KK >= 16 ? 1 : 16 / KK;
  const uint32_t sx_addr = (uint32_t)reinterpret_cast<uintptr_t>(sX);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // Row j of this wave's 16: all address arithmetic is scalar and cheap -- the
  // tile's first row pointer and the last valid row offset are formed once per
  // tile (tile_src); a row then costs one s_min, one 32-bit s_mul and a 64-bit
  // add (ldx <= 256 and row < 64, so the element offset fits 32 bits).
  const uint32_t dst_wave = __builtin_amdgcn_readfirstlane(
      sx_addr + (uint32_t)(wave_u * 16 * LD * 4));
  const int ldx32 = (int)ldx;
  struct TileSrc {
    const float* base;  // &X[n0, 0]
    int last;           // min(N - 1 - n0, kRows - 1): rows past N re-read row N-1
    uint32_t dst;       // LDS address of this wave's row 0 in the target buffer
  };
  auto tile_src = [&](int64_t n0, int buf) {
    const int64_t left = N - 1 - n0;
    return TileSrc{X + n0 * ldx, (int)(left < kRows - 1 ? left : kRows - 1),
                   dst_wave + (uint32_t)(buf * kRows * LD * 4)};
  };
  auto dma_row = [&](const TileSrc& t, int j) {
    const int row = wave_u * 16 + j;
    const int r = row < t.last ? row : t.last;
    lds_dma_row<kDmaB>(t.base + r * ldx32, t.dst + (uint32_t)(j * LD * 4),
                       (uint32_t)lane);
  };
  float yr = 0.f;

  f16v G[FB];
#pragma unroll
  for (int t = 0; t < FB; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) G[t][r] = 0.f;
  // log-likelihood of this lane's rows: summed per tile in float32 (16
  // terms), tile sums in float64.  One float32 accumulator over all tiles
  // reaches ~1e5 at N = 10^6 (ulp 0.016) and random-walks to an error of
  // O(1) in the log-density -- acceptance at BASELINE configs[2]'s full size
  // fell from 0.90 to 0.41 on it (gpurun_out/r03b); the reference's
  // tf.reduce_sum is a tree reduction and has no such growth.
  double ll_lane = 0.0;
  float ll_tile = 0.f;

  // gridDim.y > 1: the data rows are split into gridDim.y contiguous ranges of
  // whole tiles and this workgroup writes PARTIAL sums (reduced afterwards by
  // lb_reduce_splits_kernel) -- for shapes with fewer chain blocks than CUs
  const int64_t n_tiles_all = (N + kRows - 1) / kRows;
  const int64_t tiles_per_split = (n_tiles_all + gridDim.y - 1) / gridDim.y;
  const int64_t tile_begin = (int64_t)blockIdx.y * tiles_per_split;
  const int64_t n_tiles = tile_begin + tiles_per_split < n_tiles_all
                              ? tile_begin + tiles_per_split
                              : n_tiles_all;
  if (gridDim.y > 1) {
    if (LL) ll += (int64_t)blockIdx.y * C;
    if (GRAD) gW += (int64_t)blockIdx.y * C * ldw;
  }
  {
    const TileSrc t0 = tile_src(tile_begin * kRows, 0);
#pragma unroll
    for (int j = 0; j < 16; ++j) dma_row(t0, j);
  }
  if (OP != 1 && tid < kRows) {
    const int64_t nr = tile_begin * kRows + tid;
    sY[tid] = nr < N ? y[nr] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // OP 2: the class this lane's column (a*32 + lo of a 64-row block whose
  // base is a multiple of the class stride) stands for
  const CatLane cat = cat_lane(lo, n_classes, OP == 2 ? cls_log2 : 0);

  // residual exchange slots: [wave][r>>2][lane][r&3]  (b128, conflict-free)
  float* __restrict__ sr_mine = sR + (wave * 4 * 64 + lane) * 4;
  const float* __restrict__ sr_sib = sR + ((wave ^ 1) * 4 * 64 + lane) * 4;

#ifdef ZS_LB_TIMING  // debug: per-phase shader clocks of wave 0 of block 0
  long long tacc[6] = {0, 0, 0, 0, 0, 0};
  long long tmark = clock64();
  // fenced: nothing is scheduled across a mark, and the accumulators are
  // forced complete (an MFMA is asynchronous) so that a phase owns its MFMAs
#define ZS_LB_MARK(i)                                        \
  {                                                          \
    __builtin_amdgcn_sched_barrier(0);                       \
    asm volatile("" ::"v"(S[0]), "v"(G[0][0]), "v"(G[FB - 1][15])); \
    const long long _t = clock64();                          \
    tacc[i] += _t - tmark;                                   \
    tmark = _t;                                              \
    __builtin_amdgcn_sched_barrier(0);                       \
  }
#else
#define ZS_LB_MARK(i)
#endif
  // OP 1: the counts x[c, n] of one tile for this lane (a gather: every lane
  // reads its own chain's row of the counts matrix, 32 rows per instruction,
  // so the instruction count is what costs).  Rows padded with zeros to a
  // multiple of 4 floats and 16-B aligned (the caller's count_stride): 4 x
  // 16 B per lane instead of 16 x 4 B.
  float xcnt[16], xnext[16];
  auto load_counts = [&](int64_t t, float* dst) {
    const int64_t cr = row_at(a * 32 + lo);
    // counts rows repeat with period yc_rows (x[n_docs, V] shared by chains)
    const float* __restrict__ xrow0 =
        yc + (cr % yc_rows) * ldy + t * kRows + b * 32 + 4 * hi;
    const int64_t left = N - (t * kRows + b * 32 + 4 * hi);  // may be <= 0
    if (yc_vec) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f4 v = f4{0.f, 0.f, 0.f, 0.f};
        if (8 * j < left) v = *reinterpret_cast<const f4*>(xrow0 + 8 * j);
#pragma unroll
        for (int m = 0; m < 4; ++m) dst[j * 4 + m] = v[m];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          dst[j * 4 + m] = (8 * j + m < left) ? xrow0[8 * j + m] : 0.f;
    }
  };
  if (OP == 1) load_counts(tile_begin, xcnt);
  // One tile.  FULL: all 64 rows of the tile exist (every tile but possibly
  // the last of the row range): the row-validity compares and selects of the
  // element-wise stage are compiled out.
  auto tile_body = [&](auto full_tag, int64_t tile) {
    constexpr bool FULL = decltype(full_tag)::value;
    const int buf = (int)((tile - tile_begin) & 1);       // sY slot
    const int xbuf = kBuf == 2 ? buf : 0;                 // sX slot
    const float* __restrict__ xb = sX + xbuf * kRows * LD;
    const bool more = tile + 1 < n_tiles;
    // rows of tile+1 (the last tile re-streams itself: clamped rows, unused)
    const int64_t n_next = (more ? tile + 1 : tile) * kRows;
    const TileSrc tnext = tile_src(n_next, kBuf == 2 ? (xbuf ^ 1) : 0);
    if (OP != 1 && tid < kRows) {
      const int64_t nr = n_next + tid;
      yr = nr < N ? y[nr] : 0.f;
    }
    // OP 1: this lane's 16 counts (chain a*32+lo, rows b*32 + 8j + 4hi .. +3)
    // of the NEXT tile go out now and are consumed one tile later: hipcc waits
    // for its own loads with `s_waitcnt vmcnt(n)` counted WITHOUT the
    // hand-issued DMA rows behind them in the same in-order queue, so a load
    // used in this tile's residual would drag the whole next X tile's DMA
    // into the wait; the end-of-tile vmcnt(0) lands these for free.
    if (OP == 1) load_counts(more ? tile + 1 : tile, xnext);

    // ---- phase 1 (own 32 rows, full K) --------------------------------------
    // Hand-pipelined: the LDS read of step kk+1 and one DMA row of tile+1 go
    // out in front of the 4 MFMAs of step kk (a lone wave per SIMD has nobody
    // else to hide its latencies).
    f16v S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
    {
      const float* __restrict__ arow = xb + (b * 32 + lo) * LD + hi * 4;
      f4 av = *reinterpret_cast<const f4*>(arow);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        f4 an = av;
        if (kk + 1 < KK) an = *reinterpret_cast<const f4*>(arow + (kk + 1) * 8);
        // one DMA row of tile t+1 per step over the first 16 steps (KK >= 16)
        // or kDmaPer rows per step (KK = 8): issued as early as the buffer is
        // free, in front of the step's MFMAs
        // (ZS_LB_DMA_PHASE3, an A/B switch: the rows under phase 3b instead)
        if (kBuf == 2 && KK >= 16 && kk < 16 && !(GRAD && ZS_LB_DMA_PHASE3))
          dma_row(tnext, kk);
        if (kBuf == 2 && KK < 16) {
#pragma unroll
          for (int j = 0; j < kDmaPer; ++j) dma_row(tnext, kk * kDmaPer + j);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
          S = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], wreg[kk * 4 + m], S, 0,
                                                   0, 0);
        av = an;
      }
    }

    ZS_LB_MARK(0)  // head + phase 1
    // ---- residual on the accumulator layout ----------------------------------
    // lane holds chain i = a*32 + lo, rows n = b*32 + (r&3) + 8*(r>>2) + 4*hi.
    // Bernoulli._log_prob (univariate.py:398-403):
    //   l*y - max(l,0) - log1p(exp(-|l|));   d/dl = y - sigmoid(l).
    // log1p(e) = ln2*log2(1+e) with e in (0,1] is good to ~1e-7 absolute.
    const int rows_left = (int)((N - tile * kRows) < kRows ? (N - tile * kRows)
                                                           : kRows);
    auto residual = [&](int r) {
      const int nl = b * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const bool valid = FULL || nl < rows_left;
      const float sv = S[r];
      if (OP == 0) {
        const float yv = sY[buf * kRows + nl];
        if (LL) {
          const float e =
              __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(sv));
          const float t1 = 1.0f + e;
          const float inv = __builtin_amdgcn_rcpf(t1);  // sigmoid(|l|) >= 1/2
          // sigmoid(l) = 1/2 + copysign(inv - 1/2, l): one v_bfi instead of
          // a compare + select
          const float sig = 0.5f + __builtin_copysignf(inv - 0.5f, sv);
          S[r] = valid ? yv - sig : 0.f;
          const float lp = sv * yv - fmaxf(sv, 0.f) -
                           0.6931471805599453f * __builtin_amdgcn_logf(t1);
          ll_tile += valid ? lp : 0.f;
        } else {
          // gradient only: sigmoid(l) = 1 / (1 + 2^(-l log2 e)) as it stands
          // (l -> -inf: 1 / inf = 0; l -> +inf: 1 / 1): mul, exp, add, rcp, sub
          const float sig = __builtin_amdgcn_rcpf(
              1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * sv));
          S[r] = valid ? yv - sig : 0.f;
        }
      } else if (OP == 2) {
        S[r] = categorical_residual<LL>(sv, sY[buf * kRows + nl], cat, valid,
                                        ll_tile);
      } else {
        // sum_v x_v log((theta.phi)_v) and d/d(theta.phi) = x / (theta.phi);
        // x = 0 contributes nothing (also where the product underflows) --
        // and the counts of rows past N are loaded as zeros
        const float xv = xcnt[r];
        const bool on = xv != 0.f;
        S[r] = on ? xv * __builtin_amdgcn_rcpf(sv) : 0.f;
        if (LL) {
          const float lp =
              xv * (0.6931471805599453f * __builtin_amdgcn_logf(sv));
          ll_tile += on ? lp : 0.f;
        }
      }
    };
    // B operand of phase 3: X[row][b*HALF + lo*FB .. +FB-1]
    typedef typename VecF<FB>::type V;
    auto xrow = [&](int blk, int r) -> V {
      const int nl = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      return *reinterpret_cast<const V*>(xb + nl * LD + b * HALF + lo * FB);
    };

    if (GRAD) {
      // ---- phase 3a: own rows, A = the residual registers; the residual of
      // group g+1 (VALU) is issued under the MFMAs of group g ----------------
      // The B operands (four X rows per group) can be read from LDS ONE GROUP
      // AHEAD: read at the top of their own group they expose an LDS round
      // trip per group (phase 3a 5 514 and 3b 4 604 clocks against 4 096 of
      // MFMAs each at D = 256, profiles/r03z_lb_phase_timing_v2.txt).
      // Measured per width and phase (profiles/r03z_prefetch3_ab.txt, four
      // builds side by side): D = 128 (the topic model's K) gains 1.9 % --
      // 123.5 -> 125.9 TFLOP/s -- and all of it from phase 3a; at D = 256
      // phase 3b reading ahead changes nothing (129.4 / 129.3) and phase 3a
      // reading ahead LOSES 2.6 % (126.0: what that phase is short of is
      // issue slots for the residual's VALU, and the extra live registers do
      // not help); at D = 64 the registers would spill under the 168-VGPR
      // bound of three workgroups per CU.
      // Second pass (profiles/r03dd_prefetch_fence_ab.txt): the ISA showed why
      // D = 256 gained nothing -- hipcc sinks every read back to just in front
      // of the MFMAs that use it (`ds_read, s_waitcnt, 4 MFMAs` per row).  A
      // scheduling fence behind the next group's reads holds them in place:
      // with it phase 3b reading ahead gains 2.1 % at D = 256 (129.5 -> 132.3
      // TFLOP/s, 0.823 -> 0.841; with phase 3a too: 131.0), while D = 128 --
      // two workgroups per CU, whose reads the other workgroup's MFMAs cover
      // -- loses 1 % to the fence (126.3 -> 125.0).  So: D = 128 both phases,
      // unfenced; D = 256 phase 3b, fenced.
#ifndef ZS_LB_PREFETCH_FENCE
#define ZS_LB_PREFETCH_FENCE(D) ((D) == 256)
#endif
#ifndef ZS_LB_PREFETCH3A
#define ZS_LB_PREFETCH3A(D) ((D) == 128)
#endif
#ifndef ZS_LB_PREFETCH3A_INSIDE
#define ZS_LB_PREFETCH3A_INSIDE 0
#endif
#ifndef ZS_LB_PREFETCH3B
#define ZS_LB_PREFETCH3B(D) ((D) >= 128)
#endif
      constexpr bool kPreA = ZS_LB_PREFETCH3A(D), kPreB = ZS_LB_PREFETCH3B(D);
      static_assert(!kPreA || kPreB, "phase 3a hands phase 3b its first group");
      V xv[4], xn[4];
      if (kPreA) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[q] = xrow(b, q);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) residual(r);
      ZS_LB_MARK(1)  // first residual group
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (kPreA) {
          // group g+1's rows; past the last group: the first of phase 3b
#pragma unroll
          for (int q = 0; q < 4; ++q)
            xn[q] = g + 1 < 4 ? xrow(b, (g + 1) * 4 + q) : xrow(b ^ 1, q);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) xv[q] = xrow(b, g * 4 + q);
        }
        *reinterpret_cast<f4*>(sr_mine + g * 256) =
            f4{S[g * 4], S[g * 4 + 1], S[g * 4 + 2], S[g * 4 + 3]};
        // (ZS_LB_PREFETCH_FENCE: hipcc sinks each of the reads above to just in
        // front of the MFMAs that use it -- `ds_read, s_waitcnt, 4 MFMAs` per
        // row in the ISA, an LDS round trip exposed per row; the fence keeps
        // the next group's reads in front of this group's MFMAs)
        if (kPreA && ZS_LB_PREFETCH_FENCE(D)) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t = 0; t < FB; ++t)
            G[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                S[g * 4 + q], vget<FB>(xv[q], t), G[t], 0, 0, 0);
        if (g + 1 < 4) {
#pragma unroll
          for (int r = 0; r < 4; ++r) residual((g + 1) * 4 + r);
#ifndef ZS_LB_NO_SGB
          // A wave issues in order: left alone, hipcc emits the 4*FB MFMAs of
          // this group back to back and the ~60 VALU / transcendental ops of the
          // next group's residual after them, where only the last MFMA is
          // left to hide them.  Ask for the LDS traffic first (the next
          // group's rows), then one MFMA, then a slice of the VALU.
          // (ZS_LB_PREFETCH3A_INSIDE: the next group's four reads as slots
          // INSIDE the pipeline, one behind each of the first four MFMAs,
          // instead of in front of it -- measured at D = 256 without the
          // fence: 126.3 against 131.6 TFLOP/s, off)
          if (kPreA && !ZS_LB_PREFETCH_FENCE(D) && !ZS_LB_PREFETCH3A_INSIDE) {
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // DS read
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
          }
#pragma unroll
          for (int i = 0; i < 4 * FB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // MFMA
            if (kPreA && ZS_LB_PREFETCH3A_INSIDE && i < 4)
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // DS read
            __builtin_amdgcn_sched_group_barrier(0x002, 16 / FB, 0);       // VALU
            __builtin_amdgcn_sched_group_barrier(0x400, (2 + FB) / FB, 0);  // trans
          }
#endif
        }
        if (kPreA) {
#pragma unroll
          for (int q = 0; q < 4; ++q) xv[q] = xn[q];
        }
      }
      ZS_LB_MARK(2)  // phase 3a
      __syncthreads();  // the sibling's residuals are in LDS
      ZS_LB_MARK(3)  // barrier 1
      // ---- phase 3b: the sibling's rows, A from LDS ----------------------------
      f4 rs = *reinterpret_cast<const f4*>(sr_sib), rs_next = rs;
      if (kPreB && !kPreA) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[q] = xrow(b ^ 1, q);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (kPreB) {
          if (g + 1 < 4) {
            rs_next = *reinterpret_cast<const f4*>(sr_sib + (g + 1) * 256);
#pragma unroll
            for (int q = 0; q < 4; ++q) xn[q] = xrow(b ^ 1, (g + 1) * 4 + q);
          }
        } else {
          rs = *reinterpret_cast<const f4*>(sr_sib + g * 256);
#pragma unroll
          for (int q = 0; q < 4; ++q) xv[q] = xrow(b ^ 1, g * 4 + q);
        }
        if (kPreB && ZS_LB_PREFETCH_FENCE(D)) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // one DMA row of tile t+1 in front of each q's MFMAs (independent
          // accumulators: the issue hides under the previous MFMA)
          if (kBuf == 2 && KK >= 16 && ZS_LB_DMA_PHASE3) dma_row(tnext, g * 4 + q);
#pragma unroll
          for (int t = 0; t < FB; ++t)
            G[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                rs[q], vget<FB>(xv[q], t), G[t], 0, 0, 0);
        }
        if (kPreB) {
          rs = rs_next;
#pragma unroll
          for (int q = 0; q < 4; ++q) xv[q] = xn[q];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) residual(r);
    }
    ZS_LB_MARK(4)  // phase 3b
    if (OP != 1 && tid < kRows) sY[(buf ^ 1) * kRows + tid] = yr;
    if (kBuf == 1) {
      __syncthreads();  // every wave is done reading the only X buffer
      if (more) {
#pragma unroll
        for (int j = 0; j < 16; ++j) dma_row(tnext, j);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA rows landed
    if (OP == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xcnt[r] = xnext[r];
    }
    __syncthreads();  // tile+1 published; this buffer free for tile+2
    if (LL) {
      ll_lane += (double)ll_tile;
      ll_tile = 0.f;
    }
    ZS_LB_MARK(5)  // DMA wait + barrier 2
  };
  // (two copies of the tile only in the gradient-only instantiations -- the
  // ones a trajectory spends its time in: with the log-likelihood terms alive
  // as well the second copy costs D = 128 its second workgroup per CU and
  // D = 64 eleven spilled registers)
  for (int64_t tile = tile_begin; tile < n_tiles; ++tile) {
    if constexpr (!LL) {
      if ((tile + 1) * kRows <= N) {
        tile_body(std::true_type{}, tile);
        continue;
      }
    }
    tile_body(std::false_type{}, tile);
  }
#ifdef ZS_LB_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && GRAD) {
    for (int i = 0; i < 6; ++i) gW[wave * 8 + i] = (float)tacc[i];
    gW[wave * 8 + 6] = (float)(n_tiles - tile_begin);
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && GRAD) return;
#endif

  // ---- epilogue -----------------------------------------------------------
  // G[t][r]: chain = c0 + a*32 + (r&3) + 8*(r>>2) + 4*hi,
  //          feature = b*HALF + lo*FB + t
  if (GRAD) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pos = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (pos < n_valid) {
        const int64_t chain = row_base + pos * row_stride;
#pragma unroll
        for (int t = 0; t < FB; ++t)
          gW[chain * ldw + b * HALF + lo * FB + t] = G[t][r];
      }
    }
  }
  // ll of chain a*32+lo: this lane's 16 rows per tile + lane^32's + the
  // sibling wave's 32 rows (through the exchange slots, now idle)
  if (LL) {
    const double ll_half = ll_lane + __shfl_xor(ll_lane, 32, 64);
    double* __restrict__ sRd = reinterpret_cast<double*>(sR);  // 64 doubles
    if (hi == 0) sRd[wave * 32 + lo] = ll_half;
    __syncthreads();
    if (b == 0 && hi == 0) {
      const int pos = a * 32 + lo;
      if (pos < n_valid)
        ll[row_base + pos * row_stride] =
            (float)(ll_half + sRd[(wave ^ 1) * 32 + lo]);
    }
  }
}

// out[c(, f)] = sum over the S row-range partials written by a split launch
__global__ __launch_bounds__(256) void lb_reduce_splits_kernel(
    const float* __restrict__ ws, int64_t C, int64_t ldw, int S,
    float* __restrict__ ll, float* __restrict__ gW) {
  const int64_t n_ll = C, n_g = gW ? C * ldw : 0;
  const float* __restrict__ gpart = ws + (int64_t)S * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ll + n_g;
       i += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    if (i < n_ll) {
      if (!ll) continue;
      for (int s = 0; s < S; ++s) acc += ws[(int64_t)s * C + i];
      ll[i] = acc;
    } else {
      const int64_t j = i - n_ll;
      for (int s = 0; s < S; ++s) acc += gpart[(int64_t)s * C * ldw + j];
      gW[j] = acc;
    }
  }
}

template <int D, int OP>
static int launch_v2(const float* W, const float* X, const float* y,
                     const float* yc, int64_t yc_rows, int64_t ldy, int64_t C,
                     int64_t N, int64_t ldw, int64_t ldx, float* ll, float* gW,
                     hipStream_t s, int n_splits = 1,
                     float* workspace = nullptr, int doc_major = 0,
                     int n_classes = 0, int cls_log2 = 0) {
  constexpr int LD = D + 4;
  const size_t lds =
      (size_t)(ZS_LB_BUF(D) * 64 * LD + 2 * 64 + 4 * 16 * 64) * sizeof(float);
  static bool attr2 = false;
  if (!attr2) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(linear_bernoulli_kernel_v2<D, true, OP>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(
          reinterpret_cast<const void*>(linear_bernoulli_kernel_v2<D, false, OP>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(
          reinterpret_cast<const void*>(
              linear_bernoulli_kernel_v2<D, true, OP, false>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(LDS)");
    attr2 = true;
  }
  // doc_major: one workgroup per (group of 64 chains, document)
  const int gx = doc_major
                     ? (int)(((C / yc_rows + kMC - 1) / kMC) * yc_rows)
                     : (int)((C + kMC - 1) / kMC);
  const int S = (n_splits > 1 && workspace) ? n_splits : 1;
  float* ll_out = S > 1 ? workspace : ll;
  float* g_out = S > 1 ? (gW ? workspace + (int64_t)S * C : nullptr) : gW;
  const dim3 grid(gx, S);
  if (gW && !ll)
    hipLaunchKernelGGL((linear_bernoulli_kernel_v2<D, true, OP, false>), grid,
                       dim3(256), lds, s, W, X, y, yc, yc_rows, ldy, C, N, ldw, ldx,
                       ll_out, g_out, doc_major, n_classes, cls_log2);
  else if (gW)
    hipLaunchKernelGGL((linear_bernoulli_kernel_v2<D, true, OP>), grid,
                       dim3(256), lds, s, W, X, y, yc, yc_rows, ldy, C, N, ldw, ldx,
                       ll_out, g_out, doc_major, n_classes, cls_log2);
  else
    hipLaunchKernelGGL((linear_bernoulli_kernel_v2<D, false, OP>), grid,
                       dim3(256), lds, s, W, X, y, yc, yc_rows, ldy, C, N, ldw, ldx,
                       ll_out, g_out, doc_major, n_classes, cls_log2);
  ZS_LAUNCH_CHECK("linear_bernoulli_kernel_v2 launch");
  if (S > 1) {
    const int64_t n = C + (gW ? C * ldw : 0);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(lb_reduce_splits_kernel, dim3((int)blocks), dim3(256), 0, s,
                       workspace, C, ldw, S, ll, gW);
    ZS_LAUNCH_CHECK("lb_reduce_splits_kernel launch");
  }
  return ZSHMC_OK;
}

// csrc/linear_bernoulli_wide.hip: 256 < n_features / n_topics <= 1024
int linear_multinomial_wide(const float* theta, const float* phi_t,
                            const float* counts, int64_t count_rows,
                            int64_t count_stride, int64_t n_rows,
                            int64_t n_vocab, int64_t n_topics, float* ll,
                            float* g_theta, int n_splits, float* workspace,
                            int doc_major, hipStream_t s);
int linear_bernoulli_wide(const float* W, const float* X, const float* y,
                          int64_t n_chains, int64_t n_rows, int64_t n_features,
                          float* ll, float* gW, int n_splits, float* workspace,
                          hipStream_t s);
int linear_categorical_wide(const float* W, const float* X, const float* labels,
                            int64_t n_cols, int64_t n_rows, int64_t n_features,
                            int n_classes, int cls_log2, float* ll, float* gW,
                            int n_splits, float* workspace, hipStream_t s);

}  // namespace zshmc

using namespace zshmc;

extern "C" int zshmc_linear_bernoulli_log_lik(const float* W, const float* X,
                                              const float* y, int64_t n_chains,
                                              int64_t n_rows, int64_t n_features,
                                              float* log_lik, float* grad_w,
                                              int n_splits, float* workspace,
                                              void* stream) {
  if (n_chains == 0) return ZSHMC_OK;
  ZS_REQUIRE(W && X && y && (log_lik || grad_w),
             "zshmc_linear_bernoulli_log_lik: null pointer");
  ZS_REQUIRE(n_chains > 0 && n_rows > 0,
             "zshmc_linear_bernoulli_log_lik: bad shape");
  ZS_REQUIRE(n_features == 64 || n_features == 128 || n_features == 256 ||
                 n_features == 512 || n_features == 1024,
             "zshmc_linear_bernoulli_log_lik: n_features must be 64, 128, 256, "
             "512 or 1024 (zero-pad W and X), got %lld", (long long)n_features);
  ZS_REQUIRE((reinterpret_cast<uintptr_t>(W) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
                 (!grad_w || (reinterpret_cast<uintptr_t>(grad_w) & 3) == 0),
             "zshmc_linear_bernoulli_log_lik: W and X must be 16-byte aligned");
  ZS_REQUIRE(n_splits >= 1 && n_splits <= 64 && (n_splits == 1 || workspace),
             "zshmc_linear_bernoulli_log_lik: 1 <= n_splits <= 64 and a "
             "workspace of n_splits*n_chains*(n_features+1) floats when > 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (n_features > 256) {
    ZS_REQUIRE(!grad_w || (reinterpret_cast<uintptr_t>(grad_w) & 15) == 0,
               "zshmc_linear_bernoulli_log_lik: grad_w must be 16-byte aligned");
    return linear_bernoulli_wide(W, X, y, n_chains, n_rows, n_features, log_lik,
                                 grad_w, n_splits, workspace, s);
  }
  switch (n_features) {
    case 64:
      return launch_v2<64, 0>(W, X, y, nullptr, 1, n_rows, n_chains, n_rows, 64,
                              64, log_lik, grad_w, s, n_splits, workspace);
    case 128:
      return launch_v2<128, 0>(W, X, y, nullptr, 1, n_rows, n_chains, n_rows,
                               128, 128, log_lik, grad_w, s, n_splits,
                               workspace);
    default:
      return launch_v2<256, 0>(W, X, y, nullptr, 1, n_rows, n_chains, n_rows,
                               256, 256, log_lik, grad_w, s, n_splits,
                               workspace);
  }
}

// Dense-logit Categorical (softmax regression): see include/zshmc.h.
extern "C" int zshmc_linear_categorical_log_lik(
    const float* W, const float* X, const float* labels, int64_t n_cols,
    int64_t n_rows, int64_t n_features, int n_classes, int class_stride,
    float* log_lik, float* grad_w, int n_splits, float* workspace,
    void* stream) {
  if (n_cols == 0) return ZSHMC_OK;
  ZS_REQUIRE(W && X && labels && (log_lik || grad_w),
             "zshmc_linear_categorical_log_lik: null pointer");
  int cls_log2 = 0;
  while ((1 << cls_log2) < class_stride) ++cls_log2;
  ZS_REQUIRE(class_stride >= 1 && class_stride <= 32 &&
                 (1 << cls_log2) == class_stride && n_classes >= 1 &&
                 n_classes <= class_stride,
             "zshmc_linear_categorical_log_lik: class_stride must be a power "
             "of two <= 32 and 1 <= n_classes <= class_stride, got %d / %d",
             n_classes, class_stride);
  ZS_REQUIRE(n_cols > 0 && n_rows > 0 && n_cols % class_stride == 0,
             "zshmc_linear_categorical_log_lik: bad shape");
  ZS_REQUIRE(n_features == 64 || n_features == 128 || n_features == 256 ||
                 n_features == 512 || n_features == 1024,
             "zshmc_linear_categorical_log_lik: n_features must be 64, 128, "
             "256, 512 or 1024 (zero-pad W and X), got %lld",
             (long long)n_features);
  ZS_REQUIRE((reinterpret_cast<uintptr_t>(W) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
                 (!grad_w || (reinterpret_cast<uintptr_t>(grad_w) & 15) == 0),
             "zshmc_linear_categorical_log_lik: W, X and grad_w must be "
             "16-byte aligned");
  ZS_REQUIRE(n_splits >= 1 && n_splits <= 64 && (n_splits == 1 || workspace),
             "zshmc_linear_categorical_log_lik: 1 <= n_splits <= 64 and a "
             "workspace of n_splits*n_cols*(n_features+1) floats when > 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (n_features > 256)
    return linear_categorical_wide(W, X, labels, n_cols, n_rows, n_features,
                                   n_classes, cls_log2, log_lik, grad_w,
                                   n_splits, workspace, s);
  switch (n_features) {
    case 64:
      return launch_v2<64, 2>(W, X, labels, nullptr, 1, n_rows, n_cols, n_rows,
                              64, 64, log_lik, grad_w, s, n_splits, workspace,
                              0, n_classes, cls_log2);
    case 128:
      return launch_v2<128, 2>(W, X, labels, nullptr, 1, n_rows, n_cols,
                               n_rows, 128, 128, log_lik, grad_w, s, n_splits,
                               workspace, 0, n_classes, cls_log2);
    default:
      return launch_v2<256, 2>(W, X, labels, nullptr, 1, n_rows, n_cols,
                               n_rows, 256, 256, log_lik, grad_w, s, n_splits,
                               workspace, 0, n_classes, cls_log2);
  }
}

extern "C" int zshmc_linear_multinomial_log_lik(const float* theta,
                                                const float* phi_t,
                                                const float* counts,
                                                int64_t count_rows,
                                                int64_t count_stride,
                                                int64_t n_rows, int64_t n_vocab,
                                                int64_t n_topics, float* log_lik,
                                                float* grad_theta, int n_splits,
                                                float* workspace, void* stream) {
  if (n_rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(theta && phi_t && counts && (log_lik || grad_theta),
             "zshmc_linear_multinomial_log_lik: null pointer");
  ZS_REQUIRE(n_rows > 0 && n_vocab > 0 && count_rows > 0 &&
                 n_rows % count_rows == 0 && count_stride >= n_vocab,
             "zshmc_linear_multinomial_log_lik: bad shape");
  ZS_REQUIRE(n_topics == 64 || n_topics == 128 || n_topics == 256 ||
                 n_topics == 512 || n_topics == 1024,
             "zshmc_linear_multinomial_log_lik: n_topics must be 64, 128, 256, "
             "512 or 1024 (zero-pad theta and phi^T), got %lld",
             (long long)n_topics);
  ZS_REQUIRE((reinterpret_cast<uintptr_t>(theta) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(phi_t) & 15) == 0,
             "zshmc_linear_multinomial_log_lik: theta and phi^T must be 16-byte aligned");
  ZS_REQUIRE(n_splits >= 1 && n_splits <= 64 && (n_splits == 1 || workspace),
             "zshmc_linear_multinomial_log_lik: 1 <= n_splits <= 64 and a "
             "workspace of n_splits*n_rows*(n_topics+1) floats when > 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // Rows r = chain * count_rows + doc (the topic model's [n_chains, n_docs]
  // chain axes): tiles of 64 chains of ONE document when the chain axis fills
  // them well (a multiple of 64, or at least 512 chains: >= 89 % of the
  // slots used), so that a workgroup's counts are one row of the matrix.
  // Same arithmetic per row either way: bit-identical results
  // (tests/test_gpu_mixture_multinomial.py compares with the same rows given
  // as one "document" each, which keeps consecutive rows).
  const bool allow_doc_major = true;
  const int64_t n_chains = n_rows / count_rows;
  if (n_topics > 256) {
    // 32-row blocks (csrc/linear_bernoulli_wide.hip): same rule, half the size
    ZS_REQUIRE(!grad_theta || (reinterpret_cast<uintptr_t>(grad_theta) & 15) == 0,
               "zshmc_linear_multinomial_log_lik: grad_theta must be 16-byte "
               "aligned");
    const int dm = allow_doc_major && count_rows > 1 &&
                   (n_chains % 32 == 0 || n_chains >= 256);
    return linear_multinomial_wide(theta, phi_t, counts, count_rows,
                                   count_stride, n_rows, n_vocab, n_topics,
                                   log_lik, grad_theta, n_splits, workspace, dm,
                                   s);
  }
  const int doc_major =
      allow_doc_major && count_rows > 1 &&
      (n_chains % kMC == 0 || n_chains >= 512);
  switch (n_topics) {
    case 64:
      return launch_v2<64, 1>(theta, phi_t, nullptr, counts, count_rows,
                              count_stride, n_rows, n_vocab, 64, 64, log_lik,
                              grad_theta, s, n_splits, workspace, doc_major);
    case 128:
      return launch_v2<128, 1>(theta, phi_t, nullptr, counts, count_rows,
                               count_stride, n_rows, n_vocab, 128, 128,
                               log_lik, grad_theta, s, n_splits, workspace,
                               doc_major);
    default:
      return launch_v2<256, 1>(theta, phi_t, nullptr, counts, count_rows,
                               count_stride, n_rows, n_vocab, 256, 256,
                               log_lik, grad_theta, s, n_splits, workspace,
                               doc_major);
  }
}
