// The tile loop of csrc/linear_bernoulli.hip (the two-GEMM likelihood kernel
// for <= 256 columns) as a device function: see that file's header comment.
#pragma once
#include "common.h"
#include "lb_asm.h"
#include "lb_ops.h"

namespace zshmc {

constexpr int kMC = 64;  // chains per workgroup

#ifndef ZS_LB_MINW  // waves per SIMD the register budget is held to
#define ZS_LB_MINW(D) ((D) >= 192 ? 1 : (D) == 128 ? 2 : 3)
#endif

// The kernel body as a device function of the workgroup's coordinates --
// (bx, by) of a (chain blocks x gy row-range slices) grid -- so that the
// one-launch-per-evaluation kernel below and the persistent trajectory kernel
// (csrc/hmc_model_traj.hip: all trips of a small problem from one cooperative
// launch) run the SAME code, bit for bit.
template <int D, bool GRAD, int OP, bool LL>
__device__ __forceinline__ void lb_body(
    const float* __restrict__ W, const float* __restrict__ X,
    const float* __restrict__ y, const float* __restrict__ yc,
    int64_t yc_rows, int64_t ldy, int64_t C, int64_t N, int64_t ldw,
    int64_t ldx, float* __restrict__ ll, float* __restrict__ gW,
    int doc_major, int n_classes, int cls_log2, const int bx, const int by,
    const int gy) {
  constexpr int LD = D + 4;          // padded LDS row: conflict-free b128 reads
  constexpr int kRows = 64;          // data rows per tile
  constexpr int KK = D / 8;          // phase-1 steps of 4 MFMAs (8 features)
  constexpr int NT = D / 32;         // phase-3 accumulators (32 features each)
  constexpr int VW = NT % 4 == 0 ? 4 : 2;  // floats per phase-3 operand read
  constexpr int NH = NT / VW;        // operand reads per data row (1, 1, 3, 2)
  typedef typename VecF<VW>::type XV;
  constexpr uint32_t kBufBytes = kRows * LD * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* __restrict__ sX = reinterpret_cast<float*>(smem);  // [2][kRows][LD]
  float* __restrict__ sY = sX + 2 * kRows * LD;             // [2][kRows]
  double* __restrict__ sE = reinterpret_cast<double*>(sY + 2 * kRows);  // [128]

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
  // profiles/archive/r03e_native_full_shape_rocprofv3_summary.txt); with one document
  // per workgroup the same four loads per tile are broadcasts of one row.
  int64_t row_base = (int64_t)bx * kMC, row_stride = 1;
  int64_t n_valid = C - row_base;
  if (OP == 1 && doc_major) {
    const int64_t grp = bx / yc_rows, doc = bx % yc_rows;
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
    // The loads are landed HERE, in the compiler's books too: left pending
    // into the tile loop, hipcc waits for load kk in front of step kk of
    // EVERY tile (`s_waitcnt vmcnt(31 - kk)`), and its count does not include
    // the hand-issued DMA rows -- the last steps' vmcnt(1), vmcnt(0) then
    // drain the next tile's DMA in the middle of this one.
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
      asm volatile("" : "+v"(wreg[kk * 4]), "+v"(wreg[kk * 4 + 1]),
                        "+v"(wreg[kk * 4 + 2]), "+v"(wreg[kk * 4 + 3]));
  }

  // ---- X tile: global -> LDS by DMA, one padded row per instruction --------
  // (global_load_lds writes lane-linear: a row of D floats is D/64 dwords per
  // lane, and the 4-float pad sits between rows, i.e. between instructions).
  // Wave w moves rows 16w .. 16w+15 of tile t+1 while tile t is computed (two
  // buffers); rows past N re-read row N-1.  hipcc does not count these
  // loads: the `s_waitcnt vmcnt(0)` in front of the tile barrier lands them.
  // All address arithmetic is scalar: the tile's first row pointer and the last
  // valid row offset are formed once per tile (tile_src); a row then costs one
  // s_min, one 32-bit s_mul and a 64-bit add.
  constexpr int kDmaB = D / 16;  // bytes per lane per row: 16 (D=256), 12, 8, 4
  // 16 rows per wave and tile over the first phase-1 steps
  constexpr int kDmaPer = KK >= 16 ? 1 : 16 / KK;
  const uint32_t sx_addr = (uint32_t)reinterpret_cast<uintptr_t>(sX);
  const uint32_t sy_addr = (uint32_t)reinterpret_cast<uintptr_t>(sY);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
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
                   dst_wave + (uint32_t)buf * kBufBytes};
  };
  auto dma_row = [&](const TileSrc& t, int j, auto half) {
    const int row = wave_u * 16 + j;
    const int r = row < t.last ? row : t.last;
    lds_dma_row<kDmaB, decltype(half)::value>(
        t.base + r * ldx32, t.dst + (uint32_t)(j * LD * 4), (uint32_t)lane);
  };
  constexpr std::integral_constant<int, -1> kWhole{};
  // the 64 labels of a tile: wave w brings labels 16w .. 16w+15 with one
  // 16-lane DMA (no branch in the step), lane n <- y[n0 + 16w + n], clamped
  // like the X rows (what a row past N carries never matters: masked with the
  // log-likelihood, a zero operand row without)
  const uint32_t lane_b = (uint32_t)(wave_u * 16) + ((uint32_t)lane & 15u);
  auto dma_labels = [&](const TileSrc& t, const float* src, int buf) {
    const uint32_t l = lane_b < (uint32_t)t.last ? lane_b : (uint32_t)t.last;
    const uint32_t voff = l * 4u;
    const uint32_t dst =
        sy_addr + (uint32_t)(buf * kRows * 4) + (uint32_t)(wave_u * 64);
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_bfm_b64 exec, 16, 0\n\t"
        "global_load_lds_dword %0, %1\n\t"
        "s_mov_b64 exec, -1"
        :
        : "v"(voff), "s"(src), "s"(dst)
        : "memory");
  };

  f16v G[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
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

  // gy > 1: the data rows are split into gy contiguous ranges of
  // whole tiles and this workgroup writes PARTIAL sums (reduced afterwards by
  // lb_reduce_splits_kernel) -- for shapes with fewer chain blocks than CUs

  const int64_t n_tiles_all = (N + kRows - 1) / kRows;
  const int64_t tiles_per_split = (n_tiles_all + gy - 1) / gy;
  const int64_t tile_begin = (int64_t)by * tiles_per_split;
  const int64_t n_tiles = tile_begin + tiles_per_split < n_tiles_all
                              ? tile_begin + tiles_per_split
                              : n_tiles_all;
  if (gy > 1) {
    if (LL) ll += (int64_t)by * C;
    if (GRAD) gW += (int64_t)by * C * ldw;
  }
  // (a row range with no tiles -- more splits than tiles -- streams the last
  // tile and never uses it: its partial sums are the zeros of the epilogue)
  const int64_t t_first =
      tile_begin < n_tiles_all ? tile_begin : n_tiles_all - 1;
  {
    const TileSrc t0 = tile_src(t_first * kRows, 0);
#pragma unroll
    for (int j = 0; j < 16; ++j) dma_row(t0, j, kWhole);
  }
  // Tile state: scalar, advanced by additions.  What a tile needs from its
  // index -- buffer, rows (64, or fewer in X's last tile), the DMA source of
  // the tile behind it -- is formed for tile t+1 in front of the last MFMAs of
  // tile t (advance(), called from end_of_tile): recomputed from the index at
  // the top of a tile it was ~55 scalar instructions (a 64-bit multiply, 64-bit
  // compares on the vector unit) between the last MFMA of one tile and the
  // first of the next, ~250 clocks of an idle matrix pipe per tile.
  const int last_rows = (int)(N - (n_tiles_all - 1) * kRows);   // 1 .. 64
  const bool ends_x = n_tiles == n_tiles_all;  // the range ends with X's last tile
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
  const CatLane cat = cat_lane(lo, n_classes, OP == 2 ? cls_log2 : 0);

  // OP 1: the counts x[c, n] of one tile for this lane (chain a*32+lo, rows
  // b*32 + 8j + 4hi .. +3), 4 x 16 B when rows are padded and aligned (the
  // caller's count_stride).  The NEXT tile's go out at the top of a tile and
  // are consumed one tile later: hipcc waits for its own loads with `s_waitcnt
  // vmcnt(n)` counted WITHOUT the hand-issued DMA rows behind them in the same
  // in-order queue, so a load used in this tile's residual would drag the whole
  // next X tile's DMA into the wait; the end-of-tile vmcnt(0) lands these.
  float xcnt[16], xnext[16];
  // counts rows repeat with period yc_rows (x[n_docs, V] shared by chains);
  // this lane's first count of the current tile / of the tile behind it
  const float* cnt_cur = nullptr;
  const float* cnt_nx = nullptr;
  if (OP == 1) {
    const int64_t cr = row_at(a * 32 + lo);
    cnt_cur = yc + (cr % yc_rows) * ldy + t_first * kRows + b * 32 + 4 * hi;
  }
  auto load_counts = [&](const float* __restrict__ xrow0, int rows_in_tile,
                         float* dst) {
    const int left = rows_in_tile - (b * 32 + 4 * hi);  // may be <= 0
    if (yc_vec && rows_in_tile == kRows) {
      // a full tile (all but X's last): every lane's four groups are inside
      // it -- four plain loads, no per-lane tests
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f4 v = *reinterpret_cast<const f4*>(xrow0 + 8 * j);
#pragma unroll
        for (int m = 0; m < 4; ++m) dst[j * 4 + m] = v[m];
      }
    } else if (yc_vec) {
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
  if (OP == 1) load_counts(cnt_cur, cur_rows, xcnt);
  // (in three parts, for three different gaps between MFMAs)
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
  //   phase 1, A: X[b*32 + lo][8 kk + 4 hi .. +3]      (+ 32 kk bytes)
  //   phase 3, B: X[b*32 + 4 hi + rowoff(r)][(t/VW)*32*VW + lo*VW + t%VW]
  //   labels    : sY[b*32 + 4 hi + 8 g .. +3]
  const uint32_t a_off = sx_addr + (uint32_t)(((b * 32 + lo) * LD + hi * 4) * 4);
  const uint32_t x_off =
      sx_addr + (uint32_t)(((b * 32 + 4 * hi) * LD + lo * VW) * 4);
  const uint32_t y_off = sy_addr + (uint32_t)((b * 32 + 4 * hi) * 4);
  f4 av[2];        // phase-1 operand ping-pong
  XV xv[2][NH];    // phase-3 operand ping-pong
  f4 yv[4];        // the tile's labels of this lane's 16 rows (OP 0 / 2)
  // first reads of a tile (its A operand of step 0, its labels)
  auto head = [&](int buf) {
    lds_read<0>(av[0], a_off + (uint32_t)buf * kBufBytes);
    if (OP != 1) {
      const uint32_t ya = y_off + (uint32_t)(buf * kRows * 4);
      lds_read<0>(yv[0], ya);
      lds_read<32>(yv[1], ya);
      lds_read<64>(yv[2], ya);
      lds_read<96>(yv[3], ya);
    }
  };
  // An asm read's destination is the compiler's to reuse from the last use
  // it can see -- while the data may still be in flight.  Where a head's
  // reads are NOT consumed (behind the last tile; in front of the re-issue
  // below) they are landed here, with the registers held until they have.
  auto land_head = [&]() {
    if constexpr (OP != 1)
      land_reads(av[0], yv[0], yv[1], yv[2], yv[3]);
    else
      land_reads(av[0]);
  };
  head(0);

  // Rows of a tile past N (only the last tile of the row range can have them)
  // hold a copy of row N-1 (the DMA clamps).  With the log-likelihood the
  // element-wise stage masks them (valid); the gradient-only instantiations
  // -- where a trajectory spends its time -- carry no masks at all: those rows
  // of the LDS tile are ZEROED before the tile is used, so whatever residual
  // they get meets a zero operand row in phase 3.
  constexpr bool MASK = LL;
#ifdef ZS_LB_TIMING  // debug: shader clocks per phase, wave 0 of block 0
  long long tacc[4] = {0, 0, 0, 0};
  long long tmark = clock64();
#define ZS_LB_MARK(i)                   \
  {                                      \
    __builtin_amdgcn_sched_barrier(0);   \
    const long long _t = clock64();      \
    tacc[i] += _t - tmark;               \
    tmark = _t;                          \
    __builtin_amdgcn_sched_barrier(0);   \
  }
#else
#define ZS_LB_MARK(i)
#endif
  auto tile_body = [&]() {
    if (!MASK && cur_rows < kRows) {
      land_head();
      const int first = cur_rows;   // 1 .. 63, workgroup-uniform
      float* __restrict__ xt = sX + buf * kRows * LD;
      for (int i = first * LD + tid; i < kRows * LD; i += 256) xt[i] = 0.f;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();
      head(buf);   // the reads issued behind the previous barrier saw old rows
    }
    const uint32_t a_addr = a_off + (uint32_t)buf * kBufBytes;
    const uint32_t x_addr = x_off + (uint32_t)buf * kBufBytes;
    const TileSrc tnext = nx;
    const float* const ynext = ynx;
    const int buf_next = buf ^ 1;
    // (OP 1: the counts of tile t+1 go out in phase 1's third step)
    const float* const cnt_next = cnt_nx;
    const int rows_next = nx_rows;
    __builtin_amdgcn_sched_barrier(0);

    // ---- phase 1: own 32 rows, full K, one accumulator chain ---------------
    f16v S;
    // A step = 4 MFMAs (64 clocks each) of the one accumulator chain.  The
    // wave issues one instruction per ~4 clocks, so a gap between two MFMAs
    // takes ~15 other instructions for free: the step's LDS read, its DMA
    // row(s) and the scalar address arithmetic are spread over the three gaps
    // (all behind the fourth MFMA they were ~26 issue slots in a 16-slot gap:
    // +41 clocks per step at D = 128, profiles/archive/r04y_lb_phase_d128.txt).
    static_for<KK>([&](auto kc) {
      constexpr int kk = decltype(kc)::value;
      if constexpr (kk == 0) wait_lgkm<0>();   // the head's reads
      mfma_v<kk == 0>(S, av[kk & 1][0], wreg[kk * 4]);
      // gap 1: the next step's operand / phase 3's first operand row
      if constexpr (kk + 1 < KK) {
        lds_read<(kk + 1) * 32>(av[(kk + 1) & 1], a_addr);
        // (and, once, the wave's 16 labels of tile t+1 / this lane's counts)
        if constexpr (OP != 1 && kk == 1) dma_labels(tnext, ynext, buf_next);
        if constexpr (OP == 1 && kk == 2) load_counts(cnt_next, rows_next, xnext);
      } else if constexpr (GRAD) {
        static_for<NH>([&](auto hc) {
          constexpr int h = decltype(hc)::value;
          lds_read<h * 32 * VW * 4>(xv[0][h], x_addr);
        });
      }
      mfma_v<false>(S, av[kk & 1][1], wreg[kk * 4 + 1]);
      // gaps 2, 3: the 16 DMA rows of tile t+1 over the first steps
      // (a 512-byte row is two instructions: one per gap)
      constexpr bool kHalves = kDmaB == 8 && kDmaPer == 1;
      if constexpr (kk * kDmaPer < 16)
        dma_row(tnext, kk * kDmaPer,
                std::integral_constant<int, kHalves ? 0 : -1>{});
      mfma_v<false>(S, av[kk & 1][2], wreg[kk * 4 + 2]);
      if constexpr (kHalves && kk < 16)
        dma_row(tnext, kk, std::integral_constant<int, 1>{});
      if constexpr (kDmaPer > 1 && kk * kDmaPer + 1 < 16) {
#pragma unroll
        for (int j = 1; j < kDmaPer; ++j)
          dma_row(tnext, kk * kDmaPer + j, kWhole);
      }
      mfma_v<false>(S, av[kk & 1][3], wreg[kk * 4 + 3]);
      wait_lgkm<0>();   // the read of gap 1, three MFMAs old
    });
    // S is complete 16 passes + write-back after the last MFMA issued
    ZS_LB_MARK(0)  // head + phase 1 (issue)
    mfma_drain(S);
    __builtin_amdgcn_sched_barrier(0);

    // ---- element-wise stage on the accumulator layout (csrc/lb_ops.h) --------
    // lane holds chain i = a*32 + lo, rows n = b*32 + (r&3) + 8*(r>>2) + 4*hi
    const int rows_left = cur_rows;
    auto residual = [&](int r) {
      const int nl = b * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const bool valid = !MASK || nl < rows_left;
      const float aux = OP == 1 ? xcnt[r] : yv[r >> 2][r & 3];
      S[r] = lb_residual<OP, LL>(S[r], aux, cat, valid, ll_tile);
    };
    // the four rows r0 .. r0+3 of one slot (r0 a multiple of 4: one register
    // group, one b128 of labels); the Categorical takes them together -- one
    // branch on the class stride for the slot (csrc/lb_ops.h)
    auto residual4 = [&](int r0) {
      if constexpr (OP == 2) {
        float v[4], lab[4];
        bool ok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] = S[r0 + q];
          lab[q] = yv[r0 >> 2][q];
          ok[q] = !MASK || b * 32 + q + 8 * (r0 >> 2) + 4 * hi < rows_left;
        }
        categorical_residual_n<LL, 4>(v, lab, cat, ok, ll_tile);
#pragma unroll
        for (int q = 0; q < 4; ++q) S[r0 + q] = v[q];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) residual(r0 + q);
      }
    };
    // end of tile: labels of tile t+1 published, its X rows landed, barrier,
    // and the first reads of tile t+1 behind it
    auto end_of_tile = [&]() {
      if (OP == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xcnt[r] = xnext[r];
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
      head(buf_next);
      __builtin_amdgcn_sched_barrier(0);
    };

    if constexpr (GRAD) {
      // The element-wise stage runs kRG rows at a time: one element's
      // mul-exp-add-rcp-sub is a dependent chain (~45 clocks exposed against
      // 22 of issue -- the wave's own VALU does not run under its MFMAs);
      // kRG independent chains in one slot fill each other's latencies.
      constexpr int kRG = 4;
      residual4(0);
      __builtin_amdgcn_sched_barrier(0);
      ZS_LB_MARK(1)  // drain + first residual
      // ---- phase 3: own rows, A = the residual register ---------------------
      static_for<16>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if constexpr (r + 1 < 16) {
          constexpr int ro = ((r + 1) & 3) + 8 * ((r + 1) >> 2);
          static_for<NH>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            lds_read<ro * LD * 4 + h * 32 * VW * 4>(xv[(r + 1) & 1][h], x_addr);
          });
          wait_lgkm<NH>();
        } else {
          wait_lgkm<0>();
        }
        constexpr int kSplit = NT / 2;
        static_for<kSplit>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          mfma_a(G[t], S[r], vget<VW>(xv[r & 1][t / VW], t % VW));
        });
        if constexpr (r < 3) {
          // tile t+1's state: everything below works from this tile's copies
          // (tnext, ynext, buf_next, rows_left, the LDS addresses)
          advance(r);
        } else if constexpr (r + 1 < 16 && (r + 1) % kRG == 0) {
          __builtin_amdgcn_sched_barrier(0);
          residual4(r + 1);
          __builtin_amdgcn_sched_barrier(0);
        } else if (r + 1 == 16) {
          // every read of this buffer has returned (lgkmcnt(0) above); the
          // rest of the row works from registers
          __builtin_amdgcn_sched_barrier(0);
          end_of_tile();
        }
        static_for<NT - kSplit>([&](auto tc) {
          constexpr int t = kSplit + decltype(tc)::value;
          mfma_a(G[t], S[r], vget<VW>(xv[r & 1][t / VW], t % VW));
        });
      });
      ZS_LB_MARK(2)  // phase 3 (the tile barrier inside it)
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) residual4(4 * g);
      __builtin_amdgcn_sched_barrier(0);
      end_of_tile();
      for (int part = 0; part < 3; ++part) advance(part);
    }
    if (LL) {
      ll_lane += (double)ll_tile;
      ll_tile = 0.f;
    }
    ZS_LB_MARK(3)  // end of tile
  };
  while (tiles_left > 0) tile_body();
#ifdef ZS_LB_TIMING
  if (bx == 0 && by == 0 && lane == 0 && GRAD) {
    for (int i = 0; i < 4; ++i) gW[wave * 8 + i] = (float)tacc[i];
    gW[wave * 8 + 6] = (float)(n_tiles - tile_begin);
  }
  if (bx == 0 && by == 0 && GRAD) return;
#endif
  land_head();   // the reads behind the last barrier (a tile that does not exist)

  // ---- epilogue -----------------------------------------------------------
  // G[t][r]: chain = a*32 + (r&3) + 8*(r>>2) + 4*hi,
  //          feature = (t/VW)*32*VW + lo*VW + t%VW; partial over the wave's rows.
  // The sibling wave (a, b^1) holds the other rows' partial: each wave parks
  // the accumulators of the half of t it does not store in the (now idle) X
  // buffers, and adds the sibling's to the half it does.
  if (GRAD) {
    // the last MFMAs have written their AGPRs
#pragma unroll
    for (int t = 0; t < NT; ++t) mfma_drain_a(G[t]);
    __syncthreads();
    constexpr int NTH = NT / 2;   // >= 1
    float* __restrict__ ex = sX;  // [wave][NTH][4][64][4]
#pragma unroll
    for (int tt = 0; tt < NTH; ++tt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f4 o;
#pragma unroll
        for (int m = 0; m < 4; ++m)
          o[m] = b == 0 ? G[NTH + tt][g * 4 + m] : G[tt][g * 4 + m];
        *reinterpret_cast<f4*>(
            ex + ((((wave * NTH + tt) * 4 + g) * 64 + lane) * 4)) = o;
      }
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < NTH; ++tt) {
      const int t = b * NTH + tt;
      const int feat = (t / VW) * 32 * VW + lo * VW + t % VW;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f4 o = *reinterpret_cast<const f4*>(
            ex + (((((wave ^ 1) * NTH + tt) * 4 + g) * 64 + lane) * 4));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int r = g * 4 + m;
          const int pos = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          // fixed order: row block 0's partial + row block 1's
          const float mine = b == 0 ? G[tt][r] : G[NTH + tt][r];
          const float sum = b == 0 ? mine + o[m] : o[m] + mine;
          if (pos < n_valid)
            gW[(row_base + pos * row_stride) * ldw + feat] = sum;
        }
      }
    }
  }
  if (LL) {
    const double ll_half = ll_lane + __shfl_xor(ll_lane, 32, 64);
    if (hi == 0) sE[wave * 32 + lo] = ll_half;
    __syncthreads();
    if (b == 0 && hi == 0) {
      const int pos = a * 32 + lo;
      if (pos < n_valid)
        ll[row_base + pos * row_stride] =
            (float)(ll_half + sE[(wave ^ 1) * 32 + lo]);
    }
  }
}


// dynamic LDS of lb_body<D, ...>: two X tile buffers, two label buffers, 128
// doubles for the epilogue
constexpr size_t lb_lds_bytes(int D) {
  return (size_t)(2 * 64 * (D + 4) + 2 * 64) * sizeof(float) + 128 * 8;
}

}  // namespace zshmc
