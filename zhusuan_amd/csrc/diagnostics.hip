// Batched effective sample size on the device: the estimator of the
// reference's zhusuan/diagnostics.py:17-64 for every (chain, dimension)
// series of a block of recorded draws, so that "ESS/s" (BASELINE.json's
// second metric) is a statistic of the WHOLE chain population instead of a
// host-side subsample.  The reference runs an O(M^2) Python loop per series.
//
//   mu = mean(x);  var = np.var(x)*n/(n-1);  var_plus = var*(n-1)/n
//   for t = 0, 1, ...:  acov_t = mean over the n-t products (x_i-mu)(x_{i+t}-mu)
//                       rho_t = 1 - (var - acov_t)/var_plus;  stop at rho_t < 0
//   ess = n / (1 + 2*sum rho_t)          (the sum starts at lag 0: quirk kept)
//
// Layout: draws are [n_draws, n_series] (draw-major, as recorded: one [C, D]
// snapshot of the state per draw), one thread per series, adjacent threads on
// adjacent series, so every load is coalesced.  Sums in float64.  Series up
// to kLdsDraws draws are staged (centred, float32) in LDS and read back from
// there for every lag; longer ones re-read the (L2-resident) global column.
#include "common.h"

namespace zshmc {

constexpr int kEssThreads = 64;
constexpr int kLdsDraws = 512;  // 512 draws x 64 series x 4 B = 128 KiB

template <bool USE_LDS>
__global__ __launch_bounds__(kEssThreads) void ess_series_kernel(
    const float* __restrict__ x, int64_t n, int64_t n_series,
    float* __restrict__ ess) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* __restrict__ sc = reinterpret_cast<float*>(smem);  // [n][64]
  const int tid = threadIdx.x;
  const int64_t s = (int64_t)blockIdx.x * kEssThreads + tid;
  const bool live = s < n_series;
  const float* __restrict__ col = x + (live ? s : 0);

  double sum = 0.0;
  for (int64_t i = 0; i < n; ++i) sum += (double)col[i * n_series];
  const double mu = sum / (double)n;
  double ss = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const double c = (double)col[i * n_series] - mu;
    ss += c * c;
    if (USE_LDS) sc[i * kEssThreads + tid] = (float)c;
  }
  // np.var = ss/n ; var = np.var*n/(n-1) ; var_plus = var*(n-1)/n
  const double var = (ss / (double)n) * (double)n / (double)(n - 1);
  const double var_plus = var * (double)(n - 1) / (double)n;

  double sum_rho = 0.0;
  bool going = live;
  for (int64_t t = 0; t < n; ++t) {
    if (!__any(going)) break;
    if (going) {
      double acc = 0.0;
      if (USE_LDS) {
        for (int64_t i = 0; i + t < n; ++i)
          acc += (double)sc[i * kEssThreads + tid] *
                 (double)sc[(i + t) * kEssThreads + tid];
      } else {
        for (int64_t i = 0; i + t < n; ++i)
          acc += ((double)col[i * n_series] - mu) *
                 ((double)col[(i + t) * n_series] - mu);
      }
      const double acov = acc / (double)(n - t);
      const double rho = 1.0 - (var - acov) / var_plus;
      if (rho < 0.0)
        going = false;  // NaN (constant series) keeps going and poisons the sum
      else
        sum_rho += rho;
    }
  }
  if (live) ess[s] = (float)((double)n / (1.0 + 2.0 * sum_rho));
}

// out[r] = min over the row's strictly positive entries (+inf if none): the
// "minimum positive ESS over dimensions" of diagnostics.py:55-64.  One wave per
// row.
__global__ __launch_bounds__(256) void min_positive_rows_kernel(
    const float* __restrict__ v, int64_t rows, int64_t cols,
    float* __restrict__ out) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  if (row >= rows) return;
  float m = __builtin_inff();
  for (int64_t c = lane; c < cols; c += kWave) {
    const float e = v[row * cols + c];
    if (e > 0.f) m = fminf(m, e);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
  if (lane == 0) out[row] = m;
}

}  // namespace zshmc

using namespace zshmc;

extern "C" int zshmc_ess_series(const float* draws, int64_t n_draws,
                                int64_t n_series, float* ess, void* stream) {
  if (n_series == 0) return ZSHMC_OK;
  ZS_REQUIRE(draws && ess, "zshmc_ess_series: null pointer");
  ZS_REQUIRE(n_draws >= 2 && n_series > 0,
             "zshmc_ess_series: need at least 2 draws, got [%lld, %lld]",
             (long long)n_draws, (long long)n_series);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t blocks = (n_series + kEssThreads - 1) / kEssThreads;
  ZS_REQUIRE(blocks <= 0x7fffffffll, "zshmc_ess_series: too many series");
  if (n_draws <= kLdsDraws) {
    const size_t lds = (size_t)n_draws * kEssThreads * sizeof(float);
    static bool ready = false;
    if (!ready) {
      hipError_t e = hipFuncSetAttribute(
          reinterpret_cast<const void*>(ess_series_kernel<true>),
          hipFuncAttributeMaxDynamicSharedMemorySize,
          (int)(kLdsDraws * kEssThreads * sizeof(float)));
      if (e != hipSuccess) return check_hip(e, "ess kernel: LDS size attribute");
      ready = true;
    }
    hipLaunchKernelGGL(ess_series_kernel<true>, dim3((unsigned)blocks),
                       dim3(kEssThreads), lds, s, draws, n_draws, n_series, ess);
  } else {
    hipLaunchKernelGGL(ess_series_kernel<false>, dim3((unsigned)blocks),
                       dim3(kEssThreads), 0, s, draws, n_draws, n_series, ess);
  }
  ZS_LAUNCH_CHECK("ess_series_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_min_positive_rows(const float* v, int64_t rows,
                                       int64_t cols, float* out, void* stream) {
  if (rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(v && out && rows > 0 && cols > 0,
             "zshmc_min_positive_rows: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t blocks = (rows + 3) / 4;
  hipLaunchKernelGGL(min_positive_rows_kernel, dim3((unsigned)blocks), dim3(256),
                     0, s, v, rows, cols, out);
  ZS_LAUNCH_CHECK("min_positive_rows_kernel launch");
  return ZSHMC_OK;
}
