// The dense-logit Bernoulli log-likelihood + gradient of
// csrc/linear_bernoulli.hip -- and its mixture-multinomial mode, the topic
// model's likelihood -- for WIDE rows: 256 < D <= 1024 features or topics
// (padded to 512 or 1024).  Same mathematics (reference
// zhusuan/distributions/univariate.py:398-403 summed by group_ndims = 1, the
// gradient tf.gradients yields through the matmul, hmc.py:430-432;
// multivariate.py:435-443 over examples/topic_models/lntm_mcem.py:39-46), same
// fp32 matrix cores, different decomposition -- at D = 1024 a 64-chain block of W
// is 256 KB and its gradient accumulators another 256 KB: neither fits what a
// workgroup of the 256-wide kernel keeps in registers.
//
// A workgroup owns 32 chains and streams X in 32-row tiles; its 4 waves split
// the FEATURES, wave f owning the quarter [f*D/4, (f+1)*D/4) for both GEMMs:
//   W[32 chains, quarter]    in registers for the whole kernel (D/8 VGPRs, the
//                            B operand of phase 1)
//   X[32 rows, quarter]      the wave's PRIVATE slice of the LDS tile, moved by
//                            its own LDS-DMA instructions: no barrier guards X
//   G[32 chains, quarter]    the gradient accumulators (D/8 registers)
//   phase 1   S_f[n, i] = sum_{d in quarter} X[n,d] W[i,d]: partial logits,
//             D/8 MFMAs (v_mfma_f32_32x32x2_f32)
//   exchange  the four partials meet in LDS (16 KB); wave f sums register
//             group f (rows 8f .. 8f+7 of the tile, fixed order) and applies
//             the Bernoulli residual y - sigmoid(l) and the log-likelihood
//             term to its 4 elements per lane; the residuals go back to LDS
//             (4 KB)
//   phase 3   G[i, d] += sum_n R[n, i] X[n, d] over the 32 rows and the wave's
//             quarter: D/8 MFMAs, A = the residuals from LDS, B = the wave's
//             X slice.  The rows of the NEXT tile are DMA'd into the slice
//             as the steps of this phase free them, one instruction in front
//             of each half of a step's MFMAs (the slice is single-buffered:
//             4 x 32 x (D/4 + 4) floats = 130 KB at D = 1024 leaves no room
//             for a second one).
// Two barriers per tile, both for the 20 KB exchange.  Roofline: MFMA,
// 4*N*D*C flop per call as in the narrow kernel.
#include "common.h"
#include "lb_ops.h"

namespace zshmc {

typedef float w4 __attribute__((ext_vector_type(4)));
typedef float w16 __attribute__((ext_vector_type(16)));

constexpr int kWC = 32;  // chains per workgroup
constexpr int kWR = 32;  // data rows per tile

// Sets of X slices in LDS.  Two fit at D = 512 (2 x 66 KB: the DMA of tile
// t+1 is then free of the "row consumed" order) -- measured twice, no gain:
// all rows at the top of the tile 110.4 against 110.0 TFLOP/s
// (profiles/archive/r03x_lb_wide_buffers.txt, before the DMA issue was spread); two
// rows in front of each phase-1 step 106.9 against 120.6 with one set and the
// rows spread under phase 3 (profiles/archive/r03bb_lb_wide_buf2_ab.txt): phase 1 is
// ONE dependent MFMA chain, a DMA instruction's ~50 clocks of issue between
// its links are not hidden, while the independent accumulators of phase 3
// absorb them.
#ifndef ZS_LBW_BUF512
#define ZS_LBW_BUF512 1
#endif
constexpr int wide_buffers(int d) { return d == 512 ? ZS_LBW_BUF512 : 1; }

// global -> LDS, 16 or 8 (= 2 x 4) bytes per lane, LDS dest = dst + lane*BYTES
// (HALF: a 512-byte row as the 16-byte form with the upper half of the wave
// masked off -- one instruction, for the rows issued one at a time under
// MFMAs -- instead of two dword forms, which do better issued back to back:
// ll-only at D = 512 95.5 against 87.9 TFLOP/s)
template <int BYTES, bool HALF = false>
__device__ __forceinline__ void wide_dma_row(const float* src, uint32_t dst,
                                             uint32_t lane) {
  static_assert(BYTES == 16 || BYTES == 8 || BYTES == 4,
                "a slice row is 1 KB, 512 B or 256 B");
  if constexpr (BYTES == 16) {
    const uint32_t voff = lane * 16u;
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(src), "s"(dst)
        : "memory");
  } else if constexpr (BYTES == 8 && !HALF) {
    const uint32_t voff = lane * 4u;
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %0, %1\n\t"
        "global_load_lds_dword %0, %1 offset:256"
        :
        : "v"(voff), "s"(src), "s"(dst)
        : "memory");
  } else if constexpr (BYTES == 8) {
    const uint32_t voff = lane * 16u;
    if (lane < 32)
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
  }
}

// (D = 256 also instantiates -- 54 KB of LDS and ~100 registers, two workgroups
// per CU -- and was measured against the 64-chain-block kernel of
// csrc/linear_bernoulli.hip: 0.77 against 0.82 of peak,
// profiles/archive/r03v_split256_ab.txt; the library does not dispatch it)
// OP as in csrc/linear_bernoulli.hip: 0 = Bernoulli over dense logits (y[n]
// per data row); 1 = UnnormalizedMultinomial over a mixture (W = theta, X =
// phi^T, the counts x[c, n] from `yc` [yc_rows, ldy] with row period yc_rows
// over the chain rows; doc_major: a workgroup's 32 rows are 32 CHAINS OF ONE
// DOCUMENT of the topic model's [n_chains, n_docs] chain axes, so that its
// counts are one row of the matrix).
// LL = false (GRAD only): no log-likelihood terms, as in
// csrc/linear_bernoulli.hip -- the interior evaluations of a trajectory.
template <int D, bool GRAD, int OP, bool LL = true>
__global__ __launch_bounds__(256, D == 256 ? 2 : 1) void linear_bernoulli_wide_kernel(
    const float* __restrict__ W, const float* __restrict__ X,
    const float* __restrict__ y, const float* __restrict__ yc, int64_t yc_rows,
    int64_t ldy, int64_t C, int64_t N, int64_t ldw, int64_t ldx,
    float* __restrict__ ll, float* __restrict__ gW, int doc_major,
    int n_classes, int cls_log2) {
  static_assert(D == 256 || D == 512 || D == 1024,
                "padded widths of the wide kernel");
  constexpr int FQ = D / 4;    // features per wave
  constexpr int LDQ = FQ + 4;  // padded LDS row: conflict-free b128 reads
  constexpr int KK = FQ / 8;   // phase-1 steps of 4 MFMAs (8 features)
  constexpr int NT = FQ / 32;  // 32-wide feature blocks = accumulators (2, 4, 8)
  typedef float w2 __attribute__((ext_vector_type(2)));
  // X slices: one set (wide_buffers above; -DZS_LBW_BUF512=2: two at D = 512)
  constexpr int kBuf = wide_buffers(D);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* __restrict__ sX = reinterpret_cast<float*>(smem);  // [kBuf][4][kWR][LDQ]
  float* __restrict__ sY = sX + kBuf * 4 * kWR * LDQ;       // [2][kWR]
  float* __restrict__ sP = sY + 2 * kWR;                    // [4][4][64][4]
  float* __restrict__ sR = sP + 4 * 4 * 64 * 4;             // [4][64][4]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int f = __builtin_amdgcn_readfirstlane(tid >> 6);  // feature quarter
  const int lo = lane & 31, hi = lane >> 5;
  int64_t row_base = (int64_t)blockIdx.x * kWC, row_stride = 1;
  int64_t n_valid64 = C - row_base;
  if (OP == 1 && doc_major) {
    const int64_t grp = blockIdx.x / yc_rows, doc = blockIdx.x % yc_rows;
    row_base = grp * kWC * yc_rows + doc;
    row_stride = yc_rows;
    n_valid64 = C / yc_rows - grp * kWC;
  }
  const int n_valid = (int)(n_valid64 < kWC ? n_valid64 : kWC);
  // row of position i (0..31) of the block; positions past the end re-read
  // the last valid row (their results are never stored)
  auto row_at = [&](int i) -> int64_t {
    return row_base + (int64_t)(i < n_valid ? i : n_valid - 1) * row_stride;
  };

  // ---- this wave's W slice -> registers (B operand: k-slot = lane half) ----
  float wreg[KK * 4];
  {
    const int64_t cr = row_at(lo);
    const float* __restrict__ wrow = W + cr * ldw + f * FQ + hi * 4;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const w4 v = *reinterpret_cast<const w4*>(wrow + kk * 8);
#pragma unroll
      for (int m = 0; m < 4; ++m) wreg[kk * 4 + m] = v[m];
    }
  }

  // ---- this wave's X slice: global -> LDS by DMA, one row per instruction --
  // (hipcc does not count these loads: the explicit `s_waitcnt vmcnt(0)` at
  // the end of a tile lands them)
  float* __restrict__ sXw0 = sX + f * kWR * LDQ;
  const uint32_t dst_wave = __builtin_amdgcn_readfirstlane(
      (uint32_t)reinterpret_cast<uintptr_t>(sXw0));
  constexpr int kSetFloats = 4 * kWR * LDQ;  // one set of four slices
  const int ldx32 = (int)ldx;
  struct TileSrc {
    const float* base;  // &X[n0, f*FQ]
    int last;           // rows past N re-read row N-1 (masked in the residual)
  };
  auto tile_src = [&](int64_t n0) {
    const int64_t left = N - 1 - n0;
    return TileSrc{X + n0 * ldx + f * FQ,
                   (int)(left < kWR - 1 ? left : kWR - 1)};
  };
  auto dma_row = [&](const TileSrc& t, int row, int set = 0) {
    const int r = row < t.last ? row : t.last;
    wide_dma_row<FQ / 16, GRAD>(
        t.base + r * ldx32,
        dst_wave + (uint32_t)((set * kSetFloats + row * LDQ) * 4),
        (uint32_t)lane);
  };

  w16 G[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) G[t][r] = 0.f;
  // per tile in float32 (4 terms), tile sums in float64 (csrc/linear_bernoulli.hip)
  double ll_lane = 0.0;

  // gridDim.y > 1: contiguous ranges of whole tiles, PARTIAL sums out
  const int64_t n_tiles_all = (N + kWR - 1) / kWR;
  const int64_t tiles_per_split = (n_tiles_all + gridDim.y - 1) / gridDim.y;
  const int64_t tile_begin = (int64_t)blockIdx.y * tiles_per_split;
  const int64_t n_tiles = tile_begin + tiles_per_split < n_tiles_all
                              ? tile_begin + tiles_per_split
                              : n_tiles_all;
  if (gridDim.y > 1) {
    if (LL) ll += (int64_t)blockIdx.y * C;
    if (GRAD) gW += (int64_t)blockIdx.y * C * ldw;
  }
  if (tile_begin < n_tiles) {
    const TileSrc t0 = tile_src(tile_begin * kWR);
#pragma unroll
    for (int j = 0; j < kWR; ++j) dma_row(t0, j);
  }
  if (OP != 1 && tid < kWR) {
    const int64_t nr = tile_begin * kWR + tid;
    sY[tid] = nr < N ? y[nr] : 0.f;
  }
  // OP 2 (csrc/lb_ops.h): the class of this lane's column (lo of a 32-row
  // block whose base is a multiple of the class stride)
  const CatLane cat = cat_lane(lo, n_classes, OP == 2 ? cls_log2 : 0);
  // OP 1: this lane's 4 counts of a tile -- chain lo, rows 8f + 4hi .. +3, the
  // elements whose residual this wave computes -- 16 contiguous bytes of the
  // chain's counts row (rows zero-padded to 4-float groups and 16-B aligned:
  // the caller's count_stride), else element by element
  const bool yc_vec = OP == 1 && (ldy & 3) == 0 && ldy >= ((N + 3) & ~3ll) &&
                      (reinterpret_cast<uintptr_t>(yc) & 15) == 0;
  auto load_counts = [&](int64_t t) -> w4 {
    const int64_t n0 = t * kWR + 8 * f + 4 * hi;
    const float* __restrict__ xrow = yc + (row_at(lo) % yc_rows) * ldy + n0;
    const int64_t left = N - n0;  // may be <= 0
    w4 v = w4{0.f, 0.f, 0.f, 0.f};
    if (yc_vec) {
      if (left > 0) v = *reinterpret_cast<const w4*>(xrow);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < left) v[j] = xrow[j];
    }
    return v;
  };
  w4 xcnt = w4{0.f, 0.f, 0.f, 0.f}, xnext = xcnt;
  if (OP == 1 && tile_begin < n_tiles) xcnt = load_counts(tile_begin);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#ifdef ZS_LBW_TIMING  // debug: per-phase shader clocks of every wave of block 0
  long long tacc[6] = {0, 0, 0, 0, 0, 0};
  long long tmark = clock64();
#define ZS_LBW_MARK(i)                                        \
  {                                                           \
    __builtin_amdgcn_sched_barrier(0);                        \
    asm volatile("" ::"v"(G[0][0]), "v"(G[NT - 1][15]));      \
    const long long _t = clock64();                           \
    tacc[i] += _t - tmark;                                    \
    tmark = _t;                                               \
    __builtin_amdgcn_sched_barrier(0);                        \
  }
#else
#define ZS_LBW_MARK(i)
#endif
  float yr = 0.f;
  for (int64_t tile = tile_begin; tile < n_tiles; ++tile) {
    const int buf = (int)((tile - tile_begin) & 1);  // sY slot
    const int xset = kBuf == 2 ? buf : 0;            // X slice set
    const float* __restrict__ sXw = sXw0 + xset * kSetFloats;
    const bool more = tile + 1 < n_tiles;
    const int64_t n_next = (more ? tile + 1 : tile) * kWR;
    const TileSrc tnext = tile_src(n_next);
    if (OP != 1 && tid < kWR) {
      const int64_t nr = n_next + tid;
      yr = nr < N ? y[nr] : 0.f;
    }
    // (consumed one tile later: a load used in THIS tile's residual would
    // make hipcc's own vmcnt wait for the DMA rows queued behind it)
    if (OP == 1) xnext = load_counts(more ? tile + 1 : tile);
    // (two slice sets: the other one was last read in the previous tile's
    // phase 3; the next tile comes in under phase 1, kWR / KK rows per step)
    constexpr int kDmaStep = (kWR + KK - 1) / KK;

    // ---- phase 1: partial logits over the wave's feature quarter -----------
    w16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
    {
      const float* __restrict__ arow = sXw + lo * LDQ + hi * 4;
      w4 av = *reinterpret_cast<const w4*>(arow);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        w4 an = av;
        if (kk + 1 < KK) an = *reinterpret_cast<const w4*>(arow + (kk + 1) * 8);
        if (kBuf == 2 && more) {
#pragma unroll
          for (int j = 0; j < kDmaStep; ++j)
            if (kk * kDmaStep + j < kWR)
              dma_row(tnext, kk * kDmaStep + j, xset ^ 1);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
          S = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], wreg[kk * 4 + m], S, 0,
                                                   0, 0);
        av = an;
      }
    }
    // lane holds chain lo, rows (r&3) + 8*(r>>2) + 4*hi: register group g =
    // r>>2 is rows 8g .. 8g+7
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<w4*>(sP + ((f * 4 + g) * 64 + lane) * 4) =
          w4{S[g * 4], S[g * 4 + 1], S[g * 4 + 2], S[g * 4 + 3]};
    if (kBuf == 1 && !GRAD && more) {
      // nothing reads the slice again: the next tile may come in now
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < kWR; ++j) dma_row(tnext, j);
    }
    ZS_LBW_MARK(0)  // head + phase 1 + partials out
    __syncthreads();  // the four partials are in LDS
    ZS_LBW_MARK(1)  // barrier 1

    // ---- residual of register group f (fixed summation order) --------------
    {
      w4 s = *reinterpret_cast<const w4*>(sP + ((0 * 4 + f) * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w)
        s += *reinterpret_cast<const w4*>(sP + ((w * 4 + f) * 64 + lane) * 4);
      const int rows_left =
          (int)((N - tile * kWR) < kWR ? (N - tile * kWR) : kWR);
      float ll_tile = 0.f;
      w4 res;
      // the element-wise stage of csrc/lb_ops.h on this lane's 4 elements
      if constexpr (OP == 2) {
        float v[4], lab[4];
        bool ok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nl = j + 8 * f + 4 * hi;
          v[j] = s[j];
          lab[j] = sY[buf * kWR + nl];
          ok[j] = nl < rows_left;
        }
        categorical_residual_n<LL, 4>(v, lab, cat, ok, ll_tile);
#pragma unroll
        for (int j = 0; j < 4; ++j) res[j] = v[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nl = j + 8 * f + 4 * hi;
          const bool valid = nl < rows_left;
          const float aux = OP == 1 ? xcnt[j] : sY[buf * kWR + nl];
          res[j] = lb_residual<OP, LL>(s[j], aux, cat, valid, ll_tile);
        }
      }
      if (LL) ll_lane += (double)ll_tile;
      if (GRAD) *reinterpret_cast<w4*>(sR + (f * 64 + lane) * 4) = res;
    }
    ZS_LBW_MARK(2)  // sum + residual
    __syncthreads();  // residuals published; sP free for the next tile
    ZS_LBW_MARK(3)  // barrier 2

    if (GRAD) {
      // ---- phase 3: G[i, d] += sum_n R[n, i] X[n, d], d in the quarter -----
      // accumulator t: features (t>>2)*128 + lo*4 + (t&3) of the quarter (a
      // lane reads 16 contiguous bytes, a half-wave 512 contiguous bytes).
      // 16 steps (g, q) of NT MFMAs: rows q + 8g + 4hi.  Hand-pipelined like
      // phase 1 -- the operands of step+1 are read from LDS in front of the
      // MFMAs of step; left to hipcc every step started with its own reads
      // and exposed their latency (~115 clocks x 16 of a 20 300-clock tile at
      // D = 1024, gpurun_out/r03z).
      auto load_x = [&](int step, float* dst) {
        const int g = step >> 2, q = step & 3;
        const float* __restrict__ xrow = sXw + (q + 8 * g + 4 * hi) * LDQ;
        if constexpr (NT == 2) {  // accumulator t: feature lo*2 + t
          const w2 v = *reinterpret_cast<const w2*>(xrow + lo * 2);
          dst[0] = v[0];
          dst[1] = v[1];
        } else {
#pragma unroll
          for (int t2 = 0; t2 < NT / 4; ++t2) {
            const w4 v = *reinterpret_cast<const w4*>(xrow + lo * 4 + t2 * 128);
#pragma unroll
            for (int m = 0; m < 4; ++m) dst[t2 * 4 + m] = v[m];
          }
        }
      };
      float xc[NT], xn[NT];
      w4 rs = *reinterpret_cast<const w4*>(sR + lane * 4), rs_next = rs;
      load_x(0, xc);
      // The two rows of a step are free once its operands sit in registers
      // (read one step earlier): their successors of the NEXT tile come in
      // under this step's MFMAs, one DMA instruction in front of each half.
      // The texture path moves 64 B per clock and CU -- the 128 KB of a
      // D = 1024 tile are 2 048 clocks of it -- and a DMA instruction costs
      // its wave ~50 clocks of scalar set-up and issue: 8 rows back to back
      // after each row group (the first form of this loop) held the MFMAs
      // behind them for ~1 650 clocks per tile, two rows at the end of each
      // step for ~1 150 (gpurun_out/r03z).
#pragma unroll
      for (int step = 0; step < 16; ++step) {
        const int g = step >> 2, q = step & 3;
        const bool dma = kBuf == 1 && more;
        if (dma) {
          // (outstanding LDS reads: this step's operands, issued NT MFMAs ago)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          dma_row(tnext, q + 8 * g);
        }
        if (step + 1 < 16) {
          load_x(step + 1, xn);
          if (q == 3)
            rs_next = *reinterpret_cast<const w4*>(sR + ((g + 1) * 64 + lane) * 4);
        }
#pragma unroll
        for (int t = 0; t < NT / 2; ++t)
          G[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(rs[q], xc[t], G[t], 0, 0,
                                                      0);
        if (dma) dma_row(tnext, q + 8 * g + 4);
#pragma unroll
        for (int t = NT / 2; t < NT; ++t)
          G[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(rs[q], xc[t], G[t], 0, 0,
                                                      0);
#pragma unroll
        for (int t = 0; t < NT; ++t) xc[t] = xn[t];
        if (q == 3) rs = rs_next;
      }
    }
    ZS_LBW_MARK(4)  // phase 3 + DMA issue
    if (OP != 1 && tid < kWR) sY[(buf ^ 1) * kWR + tid] = yr;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the slice has landed
    if (OP == 1) xcnt = xnext;
    ZS_LBW_MARK(5)  // DMA wait
  }
#ifdef ZS_LBW_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && GRAD) {
    for (int i = 0; i < 6; ++i) gW[f * 8 + i] = (float)tacc[i];
    gW[f * 8 + 6] = (float)(n_tiles - tile_begin);
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && GRAD) return;
#endif

  // ---- epilogue -------------------------------------------------------------
  // G[t][r]: row position (r&3) + 8*(r>>2) + 4*hi of the block,
  //          feature = f*FQ + (t>>2)*128 + lo*4 + (t&3)
  if (GRAD) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pos = (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (pos < n_valid) {
        float* __restrict__ grow =
            gW + (row_base + pos * row_stride) * ldw + f * FQ;
        if constexpr (NT == 2) {
          *reinterpret_cast<w2*>(grow + lo * 2) = w2{G[0][r], G[1][r]};
        } else {
#pragma unroll
          for (int t2 = 0; t2 < NT / 4; ++t2)
            *reinterpret_cast<w4*>(grow + lo * 4 + t2 * 128) =
                w4{G[t2 * 4][r], G[t2 * 4 + 1][r], G[t2 * 4 + 2][r],
                   G[t2 * 4 + 3][r]};
        }
      }
    }
  }
  // ll of chain lo: this lane's 4 rows per tile + lane^32's + the other three
  // waves' (through the exchange area, now idle)
  if (LL) {
    const double ll_half = ll_lane + __shfl_xor(ll_lane, 32, 64);
    __syncthreads();
    double* __restrict__ sLd = reinterpret_cast<double*>(sP);  // [4][32]
    if (hi == 0) sLd[f * 32 + lo] = ll_half;
    __syncthreads();
    if (f == 0 && hi == 0 && lo < n_valid)
      ll[row_base + lo * row_stride] =
          (float)(((sLd[lo] + sLd[32 + lo]) + sLd[64 + lo]) + sLd[96 + lo]);
  }
}

// csrc/linear_bernoulli.hip: out[c(, f)] = the row-range partials of a split
// launch added in a fixed order (sum_parts8, csrc/common.h)
int lb_reduce_splits(const float* ws, int64_t C, int64_t ldw, int S, float* ll,
                     float* gW, hipStream_t s);

template <int D, int OP>
static int launch_wide(const float* W, const float* X, const float* y,
                       const float* yc, int64_t yc_rows, int64_t ldy,
                       int64_t C, int64_t N, float* ll, float* gW,
                       hipStream_t s, int n_splits, float* workspace,
                       int doc_major, int n_classes = 0, int cls_log2 = 0) {
  constexpr int LDQ = D / 4 + 4;
  const size_t lds = (size_t)(wide_buffers(D) * 4 * kWR * LDQ + 2 * kWR +
                              4 * 4 * 64 * 4 + 4 * 64 * 4) *
                     sizeof(float);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(linear_bernoulli_wide_kernel<D, true, OP>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(
          reinterpret_cast<const void*>(
              linear_bernoulli_wide_kernel<D, false, OP>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(
          reinterpret_cast<const void*>(
              linear_bernoulli_wide_kernel<D, true, OP, false>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(LDS)");
    attr = true;
  }
  const int S = (n_splits > 1 && workspace) ? n_splits : 1;
  float* ll_out = S > 1 ? workspace : ll;
  float* g_out = S > 1 ? (gW ? workspace + (int64_t)S * C : nullptr) : gW;
  // doc_major: one workgroup per (group of 32 chains, document)
  const int64_t gx = doc_major ? ((C / yc_rows + kWC - 1) / kWC) * yc_rows
                               : (C + kWC - 1) / kWC;
  const dim3 grid((unsigned)gx, S);
  if (gW && !ll)
    hipLaunchKernelGGL((linear_bernoulli_wide_kernel<D, true, OP, false>), grid,
                       dim3(256), lds, s, W, X, y, yc, yc_rows, ldy, C, N,
                       (int64_t)D, (int64_t)D, ll_out, g_out, doc_major, n_classes,
                       cls_log2);
  else if (gW)
    hipLaunchKernelGGL((linear_bernoulli_wide_kernel<D, true, OP>), grid,
                       dim3(256), lds, s, W, X, y, yc, yc_rows, ldy, C, N,
                       (int64_t)D, (int64_t)D, ll_out, g_out, doc_major, n_classes,
                       cls_log2);
  else
    hipLaunchKernelGGL((linear_bernoulli_wide_kernel<D, false, OP>), grid,
                       dim3(256), lds, s, W, X, y, yc, yc_rows, ldy, C, N,
                       (int64_t)D, (int64_t)D, ll_out, g_out, doc_major, n_classes,
                       cls_log2);
  ZS_LAUNCH_CHECK("linear_bernoulli_wide_kernel launch");
  if (S > 1) {
    return lb_reduce_splits(workspace, C, (int64_t)D, S, ll, gW, s);
  }
  return ZSHMC_OK;
}

// n_features 512 or 1024; called by zshmc_linear_bernoulli_log_lik
int linear_bernoulli_wide(const float* W, const float* X, const float* y,
                          int64_t n_chains, int64_t n_rows, int64_t n_features,
                          float* ll, float* gW, int n_splits, float* workspace,
                          hipStream_t s) {
  if (n_features == 512)
    return launch_wide<512, 0>(W, X, y, nullptr, 1, n_rows, n_chains, n_rows,
                               ll, gW, s, n_splits, workspace, 0);
  return launch_wide<1024, 0>(W, X, y, nullptr, 1, n_rows, n_chains, n_rows, ll,
                              gW, s, n_splits, workspace, 0);
}

// n_features 512 or 1024; called by zshmc_linear_categorical_log_lik: the
// rows of W are (chain, class) pairs in groups of 2^cls_log2
int linear_categorical_wide(const float* W, const float* X, const float* labels,
                            int64_t n_cols, int64_t n_rows, int64_t n_features,
                            int n_classes, int cls_log2, float* ll, float* gW,
                            int n_splits, float* workspace, hipStream_t s) {
  if (n_features == 512)
    return launch_wide<512, 2>(W, X, labels, nullptr, 1, n_rows, n_cols,
                               n_rows, ll, gW, s, n_splits, workspace, 0,
                               n_classes, cls_log2);
  return launch_wide<1024, 2>(W, X, labels, nullptr, 1, n_rows, n_cols, n_rows,
                              ll, gW, s, n_splits, workspace, 0, n_classes,
                              cls_log2);
}

// n_topics 512 or 1024; called by zshmc_linear_multinomial_log_lik
int linear_multinomial_wide(const float* theta, const float* phi_t,
                            const float* counts, int64_t count_rows,
                            int64_t count_stride, int64_t n_rows,
                            int64_t n_vocab, int64_t n_topics, float* ll,
                            float* g_theta, int n_splits, float* workspace,
                            int doc_major, hipStream_t s) {
  if (n_topics == 512)
    return launch_wide<512, 1>(theta, phi_t, nullptr, counts, count_rows,
                               count_stride, n_rows, n_vocab, ll, g_theta, s,
                               n_splits, workspace, doc_major);
  return launch_wide<1024, 1>(theta, phi_t, nullptr, counts, count_rows,
                              count_stride, n_rows, n_vocab, ll, g_theta, s,
                              n_splits, workspace, doc_major);
}

}  // namespace zshmc
