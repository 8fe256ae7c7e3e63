// The two-GEMM likelihood of csrc/linear_bernoulli.hip for rows of 257 .. 896
// features or topics (padded widths 320 .. 896 in steps of 64): same
// mathematics (Bernoulli._log_prob univariate.py:398-403, Categorical :496-548,
// UnnormalizedMultinomial over a mixture multivariate.py:435-443, each summed
// by group_ndims = 1, and the gradient tf.gradients yields through the matmul,
// hmc.py:430-432), same "rows stay with their wave" decomposition, on the
// 16 x 16 x 4 fp32 MFMA: a 32-chain block of W at these widths (D/2 registers)
// plus its gradient accumulators (D/2 more) no longer fits a wave; a 16-chain
// block does (D/4 + D/4: 448 of the 512 registers at D = 896).
//
// A workgroup owns 64 chains: wave a (0..3) the chains 16a .. 16a+15 -- all
// four waves work on the SAME 16 data rows of a tile, each for its own chains:
//   phase 1  S[n, i] = sum_d X[n,d] W[i,d], 16 rows x 16 chains, full K = D:
//            D/4 MFMAs (v_mfma_f32_16x16x4_f32, 8 passes) on one accumulator,
//            the wave's W block in registers (D/4 VGPRs, the B operand).
//   element-wise stage on the 4 accumulator registers (csrc/lb_ops.h).
//   phase 3  G[i, f] += sum_n R[n, i] X[n, f] over the 16 rows and ALL D
//            features: D/4 MFMAs on D/16 accumulator tiles (D/4 AGPRs); the A
//            operand is the residual register itself (the C/D layout of phase
//            1 IS the A layout of phase 3: rows 4q + r of lane quarter q).
// No exchange between waves at all: no partial logits (the wide kernel's two
// barriers per tile), no partial gradients.  One barrier per 16-row tile, for
// the double-buffered X tile (2 x 16 x (D+4) floats, one LDS-DMA instruction
// per KB of a row).  The tile loop is written out in issue order as asm
// statements (csrc/lb_asm.h), as in csrc/linear_bernoulli.hip.
// Roofline: MFMA (fp32 157 TFLOP/s); 4*N*D*C flop per call.
#include "common.h"
#include "lb_asm.h"
#include "lb_ops.h"

namespace zshmc {

constexpr int kMidC = 64;  // chains per workgroup
constexpr int kMidR = 16;  // data rows per tile

template <int D, bool GRAD, int OP, bool LL = true>
__global__ __launch_bounds__(256, 1) void linear_bernoulli_mid_kernel(
    const float* __restrict__ W, const float* __restrict__ X,
    const float* __restrict__ y, const float* __restrict__ yc,
    int64_t yc_rows, int64_t ldy, int64_t C, int64_t N, int64_t ldw,
    int64_t ldx, float* __restrict__ ll, float* __restrict__ gW,
    int doc_major, int n_classes, int cls_log2) {
  static_assert(D % 64 == 0 && D > 256 && D <= 896, "widths of this kernel");
  constexpr int LD = D + 4;       // padded LDS row: conflict-free b128 reads
  constexpr int kRows = kMidR;
  constexpr int KS = D / 16;      // phase-1 steps of 4 MFMAs (16 features)
  constexpr int NT = D / 16;      // accumulator tiles of phase 3 (16 features)
  constexpr int NTT = D / 64;     // phase-3 operand reads per residual register
  constexpr int NS = 4 * NTT;     // phase-3 steps of 4 MFMAs
  constexpr uint32_t kBufBytes = kRows * LD * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* __restrict__ sX = reinterpret_cast<float*>(smem);  // [2][kRows][LD]
  float* __restrict__ sY = sX + 2 * kRows * LD;             // [2][kRows]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int a = __builtin_amdgcn_readfirstlane(tid >> 6);  // chain block
  const int l16 = lane & 15, q4 = lane >> 4;
  // the 64 rows of W this workgroup owns (csrc/linear_bernoulli.hip:
  // consecutive, or 64 chains of ONE document of the topic model's
  // [n_chains, n_docs] chain axes)
  int64_t row_base = (int64_t)blockIdx.x * kMidC, row_stride = 1;
  int64_t n_valid = C - row_base;
  if (OP == 1 && doc_major) {
    const int64_t grp = blockIdx.x / yc_rows, doc = blockIdx.x % yc_rows;
    row_base = grp * kMidC * yc_rows + doc;
    row_stride = yc_rows;
    n_valid = C / yc_rows - grp * kMidC;
  }
  n_valid = n_valid < kMidC ? n_valid : kMidC;
  auto row_at = [&](int i) -> int64_t {
    return row_base + (int64_t)(i < n_valid ? i : (int)n_valid - 1) * row_stride;
  };
  const bool yc_vec = OP == 1 && (ldy & 3) == 0 && ldy >= ((N + 3) & ~3ll) &&
                      (reinterpret_cast<uintptr_t>(yc) & 15) == 0;

  // ---- this wave's W block -> registers: B operand of step s, MFMA m =
  // W[chain 16a + l16][16 s + 4 q4 + m] (the k-slot of a lane is its quarter)
  float wreg[KS * 4];
  {
    const int64_t cr = row_at(a * 16 + l16);
    const float* __restrict__ wrow = W + cr * ldw + q4 * 4;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const f4 v = *reinterpret_cast<const f4*>(wrow + s * 16);
#pragma unroll
      for (int m = 0; m < 4; ++m) wreg[s * 4 + m] = v[m];
    }
    // landed here, in the compiler's books too (csrc/linear_bernoulli.hip)
#pragma unroll
    for (int s = 0; s < KS; ++s)
      asm volatile("" : "+v"(wreg[s * 4]), "+v"(wreg[s * 4 + 1]),
                        "+v"(wreg[s * 4 + 2]), "+v"(wreg[s * 4 + 3]));
  }

  // ---- X tile: global -> LDS by DMA.  A row of D floats is kPieces
  // instructions (1 KB each, the last one D*4 % 1024 bytes under a lane mask);
  // wave w moves rows 4w .. 4w+3 of tile t+1 while tile t is computed; rows
  // past N re-read row N-1.
  constexpr int kRowB = D * 4;
  constexpr int kPieces = (kRowB + 1023) / 1024;
  constexpr int kLastLanes = (kRowB - (kPieces - 1) * 1024) / 16;
  constexpr int kDma = 4 * kPieces;   // DMA instructions per wave and tile
  static_assert(kDma < KS, "one DMA instruction per phase-1 step, then the labels");
  const uint32_t sx_addr = (uint32_t)reinterpret_cast<uintptr_t>(sX);
  const uint32_t sy_addr = (uint32_t)reinterpret_cast<uintptr_t>(sY);
  const uint32_t dst_wave = sx_addr + (uint32_t)(a * 4 * LD * 4);
  const int ldx32 = (int)ldx;
  struct TileSrc {
    const float* base;  // &X[n0, 0]
    int last;           // min(N - 1 - n0, kRows - 1)
    uint32_t dst;       // LDS address of this wave's row 0 in the target buffer
  };
  auto tile_src = [&](int64_t n0, int buf) {
    const int64_t left = N - 1 - n0;
    return TileSrc{X + n0 * ldx, (int)(left < kRows - 1 ? left : kRows - 1),
                   dst_wave + (uint32_t)buf * kBufBytes};
  };
  // instruction i (0 .. kDma-1) of a tile: row 4a + i / kPieces, piece i % kPieces
  auto dma_piece = [&](const TileSrc& t, auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int j = i / kPieces, p = i % kPieces;
    const int row = a * 4 + j;
    const int r = row < t.last ? row : t.last;
    const float* src = t.base + r * ldx32 + p * 256;
    const uint32_t dst = t.dst + (uint32_t)(j * LD * 4 + p * 1024);
    if constexpr (p + 1 < kPieces)
      lds_dma_x4<64>(src, dst, (uint32_t)lane);
    else
      lds_dma_x4<kLastLanes>(src, dst, (uint32_t)lane);
  };
  // the 16 labels of a tile: wave a brings labels 4a .. 4a+3 with one 4-lane
  // DMA (no branch in the step), clamped like the rows
  const uint32_t lane_b = (uint32_t)(a * 4) + ((uint32_t)lane & 3u);
  auto dma_labels = [&](const TileSrc& t, const float* src, int buf) {
    const uint32_t l = lane_b < (uint32_t)t.last ? lane_b : (uint32_t)t.last;
    const uint32_t voff = l * 4u;
    const uint32_t dst =
        sy_addr + (uint32_t)(buf * kRows * 4) + (uint32_t)(a * 16);
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_bfm_b64 exec, 4, 0\n\t"
        "global_load_lds_dword %0, %1\n\t"
        "s_mov_b64 exec, -1"
        :
        : "v"(voff), "s"(src), "s"(dst)
        : "memory");
  };

  f4 G[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) G[t] = f4{0.f, 0.f, 0.f, 0.f};
  // per tile in float32 (4 terms), tile sums in float64 (csrc/linear_bernoulli.hip)
  double ll_lane = 0.0;
  float ll_tile = 0.f;

  // gridDim.y > 1: contiguous ranges of whole tiles, PARTIAL sums out
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
  // (a row range with no tiles streams the last tile and never uses it)
  const int64_t t_first =
      tile_begin < n_tiles_all ? tile_begin : n_tiles_all - 1;
  {
    const TileSrc t0 = tile_src(t_first * kRows, 0);
    static_for<kDma>([&](auto ic) { dma_piece(t0, ic); });
  }
  // Tile state, scalar and advanced by additions inside phase 3
  // (csrc/linear_bernoulli.hip): rebuilt from the tile index at the top of a
  // tile it is scalar code between two tiles' MFMAs.
  const int last_rows = (int)(N - (n_tiles_all - 1) * kRows);   // 1 .. 16
  const bool ends_x = n_tiles == n_tiles_all;
  int tiles_left = (int)(n_tiles - tile_begin);   // <= 0: no tiles
  int buf = 0;
  int cur_rows = (ends_x && tiles_left <= 1) ? last_rows : kRows;
  const float* xcur = X + t_first * kRows * ldx;
  const float* ycur = OP != 1 ? y + t_first * kRows : nullptr;
  TileSrc nx;               // the tile behind the current one (or it again)
  int nx_rows;
  const float* ynx = nullptr;
  if (OP != 1 && tid < kRows) {
    const int64_t nr = tile_begin * kRows + tid;
    sY[tid] = nr < N ? y[nr] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // OP 2: the class of this lane's column (16a + l16 of a 64-row block whose
  // base is a multiple of the class stride, which is <= 16 here)
  const CatLane cat = cat_lane(l16, n_classes, OP == 2 ? cls_log2 : 0);

  // OP 1: this lane's 4 counts of a tile (chain 16a + l16, rows 4 q4 .. +3),
  // the NEXT tile's loaded at the top of a tile (csrc/linear_bernoulli.hip)
  const float* cnt_cur = nullptr;   // this lane's first count of the tile
  const float* cnt_nx = nullptr;
  if (OP == 1)
    cnt_cur = yc + (row_at(a * 16 + l16) % yc_rows) * ldy + t_first * kRows +
              4 * q4;
  auto load_counts = [&](const float* __restrict__ xrow, int rows_in_tile) -> f4 {
    const int left = rows_in_tile - 4 * q4;  // may be <= 0
    f4 v = f4{0.f, 0.f, 0.f, 0.f};
    if (yc_vec) {
      if (left > 0) v = *reinterpret_cast<const f4*>(xrow);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < left) v[j] = xrow[j];
    }
    return v;
  };
  f4 xcnt = f4{0.f, 0.f, 0.f, 0.f}, xnext = xcnt;
  if (OP == 1 && tiles_left > 0) xcnt = load_counts(cnt_cur, cur_rows);
  auto plan_next = [&](int part) {
    const bool more = tiles_left > 1;
    if (part == 0) {
      nx_rows = more ? ((ends_x && tiles_left == 2) ? last_rows : kRows) : cur_rows;
      nx.last = nx_rows - 1;
      nx.dst = dst_wave + (uint32_t)(buf ^ 1) * kBufBytes;
    } else if (part == 1) {
      nx.base = more ? xcur + (int64_t)kRows * ldx : xcur;
    } else {
      if (OP != 1) ynx = more ? ycur + kRows : ycur;
      if (OP == 1) cnt_nx = more ? cnt_cur + kRows : cnt_cur;
    }
  };
  auto advance = [&](int part) {
    if (part == 0) {
      cur_rows = nx_rows;
      xcur = nx.base;
      if (OP != 1) ycur = ynx;
      if (OP == 1) cnt_cur = cnt_nx;
      tiles_left -= 1;
      buf ^= 1;
    }
    plan_next(part);
  };
  for (int part = 0; part < 3; ++part) plan_next(part);

  // LDS byte addresses of this lane's operands in buffer 0:
  //   phase 1, A: X[l16][16 s + 4 q4 .. +3]                  (+ 64 s bytes)
  //   phase 3, B: X[4 q4 + r][64 T + 4 l16 .. +3]  (+ (r LD + 64 T) 4 bytes)
  //   labels    : sY[4 q4 .. +3]
  const uint32_t a_off = sx_addr + (uint32_t)((l16 * LD + q4 * 4) * 4);
  const uint32_t x_off = sx_addr + (uint32_t)((4 * q4 * LD + l16 * 4) * 4);
  const uint32_t y_off = sy_addr + (uint32_t)(4 * q4 * 4);
  f4 av[2];   // phase-1 operand ping-pong
  f4 xv[3];   // phase-3 operand ring (read two steps ahead)
  f4 yv;      // the tile's labels of this lane's 4 rows (OP 0 / 2)
  auto head = [&](int buf) {
    lds_read<0>(av[0], a_off + (uint32_t)buf * kBufBytes);
    if (OP != 1) lds_read<0>(yv, y_off + (uint32_t)(buf * kRows * 4));
  };
  auto land_head = [&]() {
    if constexpr (OP != 1)
      land_reads(av[0], yv);
    else
      land_reads(av[0]);
  };
  head(0);

  constexpr bool MASK = LL;   // csrc/linear_bernoulli.hip: rows past N
  auto tile_body = [&]() {
    if (!MASK && cur_rows < kRows) {
      land_head();
      const int first = cur_rows;   // 1 .. 15
      float* __restrict__ xt = sX + buf * kRows * LD;
      for (int i = first * LD + tid; i < kRows * LD; i += 256) xt[i] = 0.f;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();
      head(buf);
    }
    const uint32_t a_addr = a_off + (uint32_t)buf * kBufBytes;
    const uint32_t x_addr = x_off + (uint32_t)buf * kBufBytes;
    const TileSrc tnext = nx;
    const float* const ynext = ynx;
    const int buf_next = buf ^ 1;
    if (OP == 1) xnext = load_counts(cnt_nx, nx_rows);
    __builtin_amdgcn_sched_barrier(0);

    // step i of phase 3: residual register r = i / NTT, feature block T = i % NTT
    auto read_x = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      lds_read<((i / NTT) * LD + (i % NTT) * 64) * 4>(xv[i % 3], x_addr);
    };

    // ---- phase 1: 16 rows x 16 chains, full K, one accumulator chain -------
    f4 S;
    static_for<KS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s + 1 < KS) {
        lds_read<(s + 1) * 64>(av[(s + 1) & 1], a_addr);
        wait_lgkm<1>();
      } else {
        wait_lgkm<0>();
        if constexpr (GRAD) {  // phase 3's first two operand reads
          read_x(std::integral_constant<int, 0>{});
          read_x(std::integral_constant<int, 1>{});
        }
      }
      // a step = 4 MFMAs of 32 clocks: ~7 other instructions fit in a gap
      // (csrc/linear_bernoulli.hip).  The step's DMA instruction and the
      // labels go between the MFMAs, not behind the fourth.
      mfma16_v<s == 0>(S, av[s & 1][0], wreg[s * 4]);
      mfma16_v<false>(S, av[s & 1][1], wreg[s * 4 + 1]);
      if constexpr (s < kDma) dma_piece(tnext, sc);
      mfma16_v<false>(S, av[s & 1][2], wreg[s * 4 + 2]);
      if constexpr (s == kDma && OP != 1) dma_labels(tnext, ynext, buf_next);
      mfma16_v<false>(S, av[s & 1][3], wreg[s * 4 + 3]);
    });
    mfma_drain(S);
    __builtin_amdgcn_sched_barrier(0);

    // ---- element-wise stage: lane holds chain 16a + l16, rows 4 q4 + r ------
    const int rows_left = cur_rows;
    if constexpr (OP == 2) {
      // the four rows together: one branch on the class stride (csrc/lb_ops.h)
      float v[4], lab[4];
      bool ok[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = S[r];
        lab[r] = yv[r];
        ok[r] = !MASK || 4 * q4 + r < rows_left;
      }
      categorical_residual_n<LL, 4>(v, lab, cat, ok, ll_tile);
#pragma unroll
      for (int r = 0; r < 4; ++r) S[r] = v[r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool valid = !MASK || 4 * q4 + r < rows_left;
        const float aux = OP == 1 ? xcnt[r] : yv[r];
        S[r] = lb_residual<OP, LL>(S[r], aux, cat, valid, ll_tile);
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    auto end_of_tile = [&]() {
      if (OP == 1) xcnt = xnext;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
      head(buf_next);
      __builtin_amdgcn_sched_barrier(0);
    };

    if constexpr (GRAD) {
      // ---- phase 3: A = the residual register, B two steps ahead from LDS --
      static_for<NS>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int r = i / NTT, T = i % NTT;
        if constexpr (i + 2 < NS) {
          read_x(std::integral_constant<int, i + 2>{});
          wait_lgkm<2>();
        } else if constexpr (i + 1 < NS) {
          wait_lgkm<1>();
        } else {
          wait_lgkm<0>();
        }
        mfma16_a(G[4 * T], S[r], xv[i % 3][0]);
        mfma16_a(G[4 * T + 1], S[r], xv[i % 3][1]);
        // tile t+1's state, a part per step: everything below works from this
        // tile's copies (tnext, ynext, buf_next, rows_left, the LDS addresses)
        if constexpr (i < 3) advance(i);
        if constexpr (i + 1 == NS) {
          // every read of this buffer has returned; the rest is registers
          __builtin_amdgcn_sched_barrier(0);
          end_of_tile();
        }
        mfma16_a(G[4 * T + 2], S[r], xv[i % 3][2]);
        mfma16_a(G[4 * T + 3], S[r], xv[i % 3][3]);
      });
    } else {
      end_of_tile();
      for (int part = 0; part < 3; ++part) advance(part);
    }
    if (LL) {
      ll_lane += (double)ll_tile;
      ll_tile = 0.f;
    }
  };
  while (tiles_left > 0) tile_body();
  land_head();   // the reads behind the last barrier (a tile that does not exist)

  // ---- epilogue -----------------------------------------------------------
  // G[4T + m][r]: chain = 16a + 4 q4 + r, feature = 64 T + 4 l16 + m
  if (GRAD) {
#pragma unroll
    for (int t = 0; t < NT; ++t) mfma_drain_a(G[t]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pos = a * 16 + 4 * q4 + r;
      if (pos < n_valid) {
        float* __restrict__ grow =
            gW + (row_base + pos * row_stride) * ldw + 4 * l16;
#pragma unroll
        for (int T = 0; T < NTT; ++T)
          *reinterpret_cast<f4*>(grow + 64 * T) =
              f4{G[4 * T][r], G[4 * T + 1][r], G[4 * T + 2][r], G[4 * T + 3][r]};
      }
    }
  }
  // ll of chain 16a + l16: this lane's 4 rows per tile + the other three
  // lane quarters'
  if (LL) {
    double v = ll_lane + __shfl_xor(ll_lane, 16, 64);
    v += __shfl_xor(v, 32, 64);
    const int pos = a * 16 + l16;
    if (q4 == 0 && pos < n_valid) ll[row_base + pos * row_stride] = (float)v;
  }
}

// csrc/linear_bernoulli.hip: out[c(, f)] = the row-range partials of a split
// launch added in a fixed order (sum_parts8, csrc/common.h)
int lb_reduce_splits(const float* ws, int64_t C, int64_t ldw, int S, float* ll,
                     float* gW, hipStream_t s);

template <int D, int OP>
static int launch_mid(const float* W, const float* X, const float* y,
                      const float* yc, int64_t yc_rows, int64_t ldy, int64_t C,
                      int64_t N, float* ll, float* gW, hipStream_t s,
                      int n_splits, float* workspace, int doc_major,
                      int n_classes, int cls_log2) {
  constexpr int LD = D + 4;
  const size_t lds = (size_t)(2 * kMidR * LD + 2 * kMidR) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(linear_bernoulli_mid_kernel<D, true, OP>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(
          reinterpret_cast<const void*>(
              linear_bernoulli_mid_kernel<D, false, OP>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(
          reinterpret_cast<const void*>(
              linear_bernoulli_mid_kernel<D, true, OP, false>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(LDS)");
    attr = true;
  }
  const int S = (n_splits > 1 && workspace) ? n_splits : 1;
  float* ll_out = S > 1 ? workspace : ll;
  float* g_out = S > 1 ? (gW ? workspace + (int64_t)S * C : nullptr) : gW;
  const int64_t gx = doc_major ? ((C / yc_rows + kMidC - 1) / kMidC) * yc_rows
                               : (C + kMidC - 1) / kMidC;
  const dim3 grid((unsigned)gx, S);
  if (gW && !ll)
    hipLaunchKernelGGL((linear_bernoulli_mid_kernel<D, true, OP, false>), grid,
                       dim3(256), lds, s, W, X, y, yc, yc_rows, ldy, C, N,
                       (int64_t)D, (int64_t)D, ll_out, g_out, doc_major,
                       n_classes, cls_log2);
  else if (gW)
    hipLaunchKernelGGL((linear_bernoulli_mid_kernel<D, true, OP>), grid,
                       dim3(256), lds, s, W, X, y, yc, yc_rows, ldy, C, N,
                       (int64_t)D, (int64_t)D, ll_out, g_out, doc_major,
                       n_classes, cls_log2);
  else
    hipLaunchKernelGGL((linear_bernoulli_mid_kernel<D, false, OP>), grid,
                       dim3(256), lds, s, W, X, y, yc, yc_rows, ldy, C, N,
                       (int64_t)D, (int64_t)D, ll_out, g_out, doc_major,
                       n_classes, cls_log2);
  ZS_LAUNCH_CHECK("linear_bernoulli_mid_kernel launch");
  if (S > 1) {
    return lb_reduce_splits(workspace, C, (int64_t)D, S, ll, gW, s);
  }
  return ZSHMC_OK;
}

// widths 320 .. 896; OP as in csrc/linear_bernoulli.hip.  Called by the three
// zshmc_linear_*_log_lik entry points (operand checks done there).
int linear_likelihood_mid(int op, const float* W, const float* X,
                          const float* y, const float* yc, int64_t yc_rows,
                          int64_t ldy, int64_t C, int64_t N, int64_t D,
                          float* ll, float* gW, int n_splits, float* workspace,
                          int doc_major, int n_classes, int cls_log2,
                          hipStream_t s) {
#define ZS_MID_CASE(DD)                                                       \
  case DD:                                                                    \
    if (op == 0)                                                              \
      return launch_mid<DD, 0>(W, X, y, yc, yc_rows, ldy, C, N, ll, gW, s,    \
                               n_splits, workspace, doc_major, n_classes,     \
                               cls_log2);                                     \
    if (op == 1)                                                              \
      return launch_mid<DD, 1>(W, X, y, yc, yc_rows, ldy, C, N, ll, gW, s,    \
                               n_splits, workspace, doc_major, n_classes,     \
                               cls_log2);                                     \
    return launch_mid<DD, 2>(W, X, y, yc, yc_rows, ldy, C, N, ll, gW, s,      \
                             n_splits, workspace, doc_major, n_classes,       \
                             cls_log2);
  switch (D) {
    ZS_MID_CASE(320)
    ZS_MID_CASE(384)
    ZS_MID_CASE(448)
    ZS_MID_CASE(512)
    ZS_MID_CASE(576)
    ZS_MID_CASE(640)
    ZS_MID_CASE(704)
    ZS_MID_CASE(768)
    ZS_MID_CASE(832)
    ZS_MID_CASE(896)
  }
#undef ZS_MID_CASE
  set_error("linear_likelihood_mid: width %lld is not instantiated", (long long)D);
  return ZSHMC_ERR_BAD_ARG;
}

}  // namespace zshmc
