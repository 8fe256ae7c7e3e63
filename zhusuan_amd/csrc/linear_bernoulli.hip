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
// X in 64-row tiles through LDS (double buffered), computes the logits
// tile on v_mfma_f32_32x32x2_f32 (exact fp32, so the 1 % acceptance parity is
// not at risk), applies the sigmoid residual in the accumulator registers,
// and feeds those registers straight back as the A operand of the second
// MFMA chain (S is computed transposed, so the C/D layout of the first GEMM
// IS the A layout of the second -- no shuffle, no LDS round trip).  Logits
// never leave registers.  Roofline: MFMA (fp32 157 TFLOP/s);
// 4*N*D*C flop per call.
#include <stdlib.h>

#include "common.h"
#include "lb_asm.h"
#include "lb_body.h"
#include "lb_ops.h"

namespace zshmc {

// ---------------------------------------------------------------------------
// 64-row tiles, W in registers, rows stay with their wave.
// (Earlier forms, numbers in profiles/ and docs/LABNOTES.md: 32-row tiles with K
// split over two waves and partial logits through LDS, 38 % of peak; phase 3
// as all 64 rows x half the features with the residual exchanged between
// sibling waves through LDS and a mid-tile barrier, compiler-scheduled, 0.86.)
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
// LL = false (log_lik = NULL): gradient only -- the L - 1 interior evaluations
// of a leapfrog trajectory need the gradient alone (hmc.py:348-372; the
// log-joint is read at its two ends, :46-61).
//
// The 4 waves of a workgroup are (a, b): chain block a (32 chains) x ROW block
// b (32 of the tile's 64 data rows) -- for BOTH GEMMs:
//   phase 1  S'[n, i] = sum_d X[n,d] W[i,d] for the wave's 32 rows, full K = D
//            (the wave's W block in registers, as before).
//   residual on the accumulator registers.
//   phase 3  G[i, f] += sum_n R'[n, i] X[n, f] over the wave's OWN 32 rows and
//            ALL D features (D/32 accumulators in AGPRs): the A operand is the
//            residual register itself for every MFMA of the phase.
// The two waves of a chain block hold partial gradients over disjoint rows;
// they meet once, in the epilogue (LDS, a + b: commutative, so bit-stable).
// No residual exchange, no mid-tile barrier, every phase-3 A operand a register
// -- at the price of D/32 accumulators per wave, which the register file has
// (D = 256: 128 W + 128 G + ~65).  One barrier per tile, two X buffers.
//
// The issue order is written out: every MFMA, LDS read and wait of a tile is
// an `asm volatile` statement, in the order the wave should issue them
// (hipcc's own schedule of the same loop written with builtins exposed an LDS
// round trip per operand row -- `ds_read, s_waitcnt, 4 MFMAs` -- merged the
// read-ahead registers of phase 1 into one, and moved 153 values per tile
// between AGPRs and VGPRs).
// Operands are read one step (phase 1) / one row (phase 3) ahead into
// ping-pong registers; `s_waitcnt lgkmcnt(n)` with n = the reads issued behind
// the one needed (LDS returns in order).  The element-wise stage stays C++
// (three OPs x LL), fenced into its slot by sched_barrier.  What the compiler
// cannot see it cannot protect: the wait states between an MFMA's write of S
// and the first VALU read (s_nop) and the landing of LDS data before a
// consumer are placed by hand below.
template <int D, bool GRAD, int OP, bool LL = true>
__global__ __launch_bounds__(256, ZS_LB_MINW(D)) void linear_bernoulli_kernel(
    const float* __restrict__ W, const float* __restrict__ X,
    const float* __restrict__ y, const float* __restrict__ yc,
    int64_t yc_rows, int64_t ldy, int64_t C, int64_t N, int64_t ldw,
    int64_t ldx, float* __restrict__ ll, float* __restrict__ gW,
    int doc_major, int n_classes, int cls_log2) {
  // (csrc/lb_body.h)
  lb_body<D, GRAD, OP, LL>(W, X, y, yc, yc_rows, ldy, C, N, ldw, ldx, ll, gW,
                           doc_major, n_classes, cls_log2, (int)blockIdx.x,
                           (int)blockIdx.y, (int)gridDim.y);
}

// out[c(, f)] = sum over the S row-range partials written by a split launch
__global__ __launch_bounds__(256) void lb_reduce_splits_kernel(
    const float* __restrict__ ws, int64_t C, int64_t ldw, int S,
    float* __restrict__ ll, float* __restrict__ gW) {
  const int64_t n_ll = C, n_g = gW ? C * ldw : 0;
  const float* __restrict__ gpart = ws + (int64_t)S * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ll + n_g;
       i += (int64_t)gridDim.x * blockDim.x) {
    // (sum_parts8: a fixed order with eight loads in flight -- one thread's
    // serial `acc += part s` was S dependent trips to a remote L2)
    if (i < n_ll) {
      if (!ll) continue;
      ll[i] = sum_parts8(ws + i, C, S);
    } else {
      const int64_t j = i - n_ll;
      gW[j] = sum_parts8(gpart + j, C * ldw, S);
    }
  }
}

// (for csrc/b3_kernel.h: the same fixed-order reduction of its partials)
int lb_reduce_splits(const float* ws, int64_t C, int64_t ldw, int S, float* ll,
                     float* gW, hipStream_t s) {
  // (the caller's next kernel adds the partials itself: csrc/common.h)
  if (KeepSplitParts::active()) return ZSHMC_OK;
  const int64_t n = C + (gW ? C * ldw : 0);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(lb_reduce_splits_kernel, dim3((int)blocks), dim3(256), 0, s,
                     ws, C, ldw, S, ll, gW);
  ZS_LAUNCH_CHECK("lb_reduce_splits_kernel launch");
  return ZSHMC_OK;
}

template <int D, int OP>
static int launch_lb(const float* W, const float* X, const float* y,
                     const float* yc, int64_t yc_rows, int64_t ldy, int64_t C,
                     int64_t N, int64_t ldw, int64_t ldx, float* ll, float* gW,
                     hipStream_t s, int n_splits = 1,
                     float* workspace = nullptr, int doc_major = 0,
                     int n_classes = 0, int cls_log2 = 0) {
  constexpr int LD = D + 4;
  // two X tile buffers, two label buffers, 128 doubles for the epilogue
#ifdef ZS_LB_LDS_PAD  // (A/B) fewer workgroups per CU
  const size_t lds = (size_t)(2 * 64 * LD + 2 * 64) * sizeof(float) + 128 * 8 +
                     (D <= 128 ? ZS_LB_LDS_PAD : 0);
#else
  const size_t lds = (size_t)(2 * 64 * LD + 2 * 64) * sizeof(float) + 128 * 8;
#endif
  static bool attr2 = false;
  if (!attr2) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(linear_bernoulli_kernel<D, true, OP>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(
          reinterpret_cast<const void*>(linear_bernoulli_kernel<D, false, OP>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(
          reinterpret_cast<const void*>(
              linear_bernoulli_kernel<D, true, OP, false>),
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
    hipLaunchKernelGGL((linear_bernoulli_kernel<D, true, OP, false>), grid,
                       dim3(256), lds, s, W, X, y, yc, yc_rows, ldy, C, N, ldw, ldx,
                       ll_out, g_out, doc_major, n_classes, cls_log2);
  else if (gW)
    hipLaunchKernelGGL((linear_bernoulli_kernel<D, true, OP>), grid,
                       dim3(256), lds, s, W, X, y, yc, yc_rows, ldy, C, N, ldw, ldx,
                       ll_out, g_out, doc_major, n_classes, cls_log2);
  else
    hipLaunchKernelGGL((linear_bernoulli_kernel<D, false, OP>), grid,
                       dim3(256), lds, s, W, X, y, yc, yc_rows, ldy, C, N, ldw, ldx,
                       ll_out, g_out, doc_major, n_classes, cls_log2);
  ZS_LAUNCH_CHECK("linear_bernoulli_kernel launch");
  if (S > 1) return lb_reduce_splits(workspace, C, ldw, S, ll, gW, s);
  return ZSHMC_OK;
}

// csrc/linear_bernoulli_mid.hip: widths 320 .. 896 (16-chain blocks per wave)
int linear_likelihood_mid(int op, const float* W, const float* X,
                          const float* y, const float* yc, int64_t yc_rows,
                          int64_t ldy, int64_t C, int64_t N, int64_t D,
                          float* ll, float* gW, int n_splits, float* workspace,
                          int doc_major, int n_classes, int cls_log2,
                          hipStream_t s);
// csrc/linear_bernoulli_wide.hip: 512 (32-class Categorical only) and 1024
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

// Which kernel takes rows of n columns (include/zshmc.h): the padded width
// and the chains per workgroup.  class_stride: 0 / 1 for the Bernoulli and
// multinomial families; the Categorical family's classes of one chain must
// sit inside one wave's chain block (32 rows; 16 in the 320..896 kernel).
extern "C" int zshmc_likelihood_plan(int64_t n, int class_stride,
                                     int64_t* width, int* chain_block) {
  ZS_REQUIRE(n >= 1 && n <= 1024,
             "zshmc_likelihood_plan: 1 <= n_columns <= 1024, got %lld",
             (long long)n);
  ZS_REQUIRE(class_stride >= 0 && class_stride <= 32,
             "zshmc_likelihood_plan: class_stride <= 32, got %d", class_stride);
  int64_t w;
  int block;
  if (n <= 256) {
    w = (n + 63) / 64 * 64;
    block = kMC;
  } else if (n <= 896 && class_stride <= 16) {
    w = n <= 320 ? 320 : (n + 63) / 64 * 64;
    block = 64;
  } else {
    w = n <= 512 ? 512 : 1024;
    block = 32;
  }
  if (width) *width = w;
  if (chain_block) *chain_block = block;
  return ZSHMC_OK;
}

static bool is_plan_width(int64_t n, int class_stride) {
  int64_t w = 0;
  return n >= 1 && n <= 1024 &&
         zshmc_likelihood_plan(n, class_stride, &w, nullptr) == ZSHMC_OK &&
         w == n;
}

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
  ZS_REQUIRE(is_plan_width(n_features, 0),
             "zshmc_linear_bernoulli_log_lik: n_features must be a kernel width "
             "(zshmc_likelihood_plan: 64 .. 256 and 320 .. 896 in steps of 64, "
             "1024; zero-pad W and X), got %lld", (long long)n_features);
  ZS_REQUIRE((reinterpret_cast<uintptr_t>(W) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
                 (!grad_w || (reinterpret_cast<uintptr_t>(grad_w) & 3) == 0),
             "zshmc_linear_bernoulli_log_lik: W and X must be 16-byte aligned");
  ZS_REQUIRE(n_splits >= 1 && n_splits <= 256 && (n_splits == 1 || workspace),
             "zshmc_linear_bernoulli_log_lik: 1 <= n_splits <= 256 and a "
             "workspace of n_splits*n_chains*(n_features+1) floats when > 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (n_features > 256) {
    ZS_REQUIRE(!grad_w || (reinterpret_cast<uintptr_t>(grad_w) & 15) == 0,
               "zshmc_linear_bernoulli_log_lik: grad_w must be 16-byte aligned");
    if (n_features <= 896)
      return linear_likelihood_mid(0, W, X, y, nullptr, 1, n_rows, n_chains,
                                   n_rows, n_features, log_lik, grad_w,
                                   n_splits, workspace, 0, 0, 0, s);
    return linear_bernoulli_wide(W, X, y, n_chains, n_rows, n_features, log_lik,
                                 grad_w, n_splits, workspace, s);
  }
  switch (n_features) {
    case 64:
      return launch_lb<64, 0>(W, X, y, nullptr, 1, n_rows, n_chains, n_rows, 64,
                              64, log_lik, grad_w, s, n_splits, workspace);
    case 128:
      return launch_lb<128, 0>(W, X, y, nullptr, 1, n_rows, n_chains, n_rows,
                               128, 128, log_lik, grad_w, s, n_splits,
                               workspace);
    case 192:
      return launch_lb<192, 0>(W, X, y, nullptr, 1, n_rows, n_chains, n_rows,
                               192, 192, log_lik, grad_w, s, n_splits,
                               workspace);
    default:
      return launch_lb<256, 0>(W, X, y, nullptr, 1, n_rows, n_chains, n_rows,
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
  ZS_REQUIRE(is_plan_width(n_features, class_stride),
             "zshmc_linear_categorical_log_lik: n_features must be the kernel "
             "width of zshmc_likelihood_plan for this class stride (zero-pad W "
             "and X), got %lld", (long long)n_features);
  ZS_REQUIRE((reinterpret_cast<uintptr_t>(W) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
                 (!grad_w || (reinterpret_cast<uintptr_t>(grad_w) & 15) == 0),
             "zshmc_linear_categorical_log_lik: W, X and grad_w must be "
             "16-byte aligned");
  ZS_REQUIRE(n_splits >= 1 && n_splits <= 256 && (n_splits == 1 || workspace),
             "zshmc_linear_categorical_log_lik: 1 <= n_splits <= 256 and a "
             "workspace of n_splits*n_cols*(n_features+1) floats when > 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (n_features > 256 && n_features <= 896 && class_stride <= 16)
    return linear_likelihood_mid(2, W, X, labels, nullptr, 1, n_rows, n_cols,
                                 n_rows, n_features, log_lik, grad_w, n_splits,
                                 workspace, 0, n_classes, cls_log2, s);
  if (n_features > 256)
    return linear_categorical_wide(W, X, labels, n_cols, n_rows, n_features,
                                   n_classes, cls_log2, log_lik, grad_w,
                                   n_splits, workspace, s);
  switch (n_features) {
    case 64:
      return launch_lb<64, 2>(W, X, labels, nullptr, 1, n_rows, n_cols, n_rows,
                              64, 64, log_lik, grad_w, s, n_splits, workspace,
                              0, n_classes, cls_log2);
    case 128:
      return launch_lb<128, 2>(W, X, labels, nullptr, 1, n_rows, n_cols,
                               n_rows, 128, 128, log_lik, grad_w, s, n_splits,
                               workspace, 0, n_classes, cls_log2);
    case 192:
      return launch_lb<192, 2>(W, X, labels, nullptr, 1, n_rows, n_cols,
                               n_rows, 192, 192, log_lik, grad_w, s, n_splits,
                               workspace, 0, n_classes, cls_log2);
    default:
      return launch_lb<256, 2>(W, X, labels, nullptr, 1, n_rows, n_cols,
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
  ZS_REQUIRE(is_plan_width(n_topics, 0),
             "zshmc_linear_multinomial_log_lik: n_topics must be a kernel width "
             "(zshmc_likelihood_plan; zero-pad theta and phi^T), got %lld",
             (long long)n_topics);
  ZS_REQUIRE((reinterpret_cast<uintptr_t>(theta) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(phi_t) & 15) == 0,
             "zshmc_linear_multinomial_log_lik: theta and phi^T must be 16-byte aligned");
  ZS_REQUIRE(n_splits >= 1 && n_splits <= 256 && (n_splits == 1 || workspace),
             "zshmc_linear_multinomial_log_lik: 1 <= n_splits <= 256 and a "
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
    ZS_REQUIRE(!grad_theta || (reinterpret_cast<uintptr_t>(grad_theta) & 15) == 0,
               "zshmc_linear_multinomial_log_lik: grad_theta must be 16-byte "
               "aligned");
    if (n_topics <= 896) {
      const int dm64 = allow_doc_major && count_rows > 1 &&
                       (n_chains % 64 == 0 || n_chains >= 512);
      return linear_likelihood_mid(1, theta, phi_t, nullptr, counts, count_rows,
                                   count_stride, n_rows, n_vocab, n_topics,
                                   log_lik, grad_theta, n_splits, workspace,
                                   dm64, 0, 0, s);
    }
    // 32-row blocks (csrc/linear_bernoulli_wide.hip): same rule, half the size
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
      return launch_lb<64, 1>(theta, phi_t, nullptr, counts, count_rows,
                              count_stride, n_rows, n_vocab, 64, 64, log_lik,
                              grad_theta, s, n_splits, workspace, doc_major);
    case 128:
      return launch_lb<128, 1>(theta, phi_t, nullptr, counts, count_rows,
                               count_stride, n_rows, n_vocab, 128, 128,
                               log_lik, grad_theta, s, n_splits, workspace,
                               doc_major);
    case 192:
      return launch_lb<192, 1>(theta, phi_t, nullptr, counts, count_rows,
                               count_stride, n_rows, n_vocab, 192, 192,
                               log_lik, grad_theta, s, n_splits, workspace,
                               doc_major);
    default:
      return launch_lb<256, 1>(theta, phi_t, nullptr, counts, count_rows,
                               count_stride, n_rows, n_vocab, 256, 256,
                               log_lik, grad_theta, s, n_splits, workspace,
                               doc_major);
  }
}
