// MultivariateNormalCholesky on the HMC path:
// log_prob, d log_prob / d value and sampling for a lower-triangular scale
// factor L (L L^T = Sigma), for gfx950.  Reference closed forms
// (zhusuan/distributions/multivariate.py):
//   _log_prob :166-188   log_z = -n/2 log(2 pi) - sum_i log L_ii
//                        z     = L^{-1} (given - mean)   (matrix_triangular_solve)
//                        out   = log_z - 1/2 |z|^2
//   _sample   :141-164   mean + L . N(0, I)
// and what tf.gradients computes through them w.r.t. `given`:
//   d out / d given = -L^{-T} z.
//
// Work decomposition: one lane per row (a row = one value vector of n_dim
// floats), 64 rows per single-wave workgroup.  The 64 x n_dim residual block is
// loaded with coalesced flat reads and kept TRANSPOSED in LDS (zs[col][lane],
// row pitch 65 floats: conflict-free both for the transposing store and for
// the per-lane column walks), the substitutions run entirely out of LDS, and
// results leave through the same transposed staging.  A scale factor shared by
// all rows (tril_count == 1, the HMC case: one covariance, many chains) is
// read through the scalar cache (uniform addresses); per-row factors
// (row r uses tril[r % tril_count]) are read per lane.
// Cost: n_dim^2 / 2 FMAs + LDS reads per row and pass -- latency-bound small
// work next to the n_dim * 8 B of compulsory HBM traffic per row.
#include "common.h"
#include "philox.h"

namespace zshmc {

constexpr int kMvnRows = 64;
constexpr int kMvnPitch = 65;

__device__ __forceinline__ int mvn_stage_rows(int64_t n_rows, int64_t row0) {
  const int64_t left = n_rows - row0;
  return left < kMvnRows ? (int)left : kMvnRows;
}

// zs[c][rr] <- f(flat element) for the block's nr x D values; rows >= nr get 0
template <typename F>
__device__ __forceinline__ void mvn_stage_in(float* zs, int nr, int D, F&& f) {
  const int lane = threadIdx.x;
  const int total = nr * D;
  for (int k = lane; k < total; k += kMvnRows) {
    const int rr = k / D, c = k - rr * D;
    zs[c * kMvnPitch + rr] = f(rr, c, k);
  }
  if (lane >= nr)
    for (int c = 0; c < D; ++c) zs[c * kMvnPitch + lane] = 0.f;
  __syncthreads();
}

template <typename F>
__device__ __forceinline__ void mvn_stage_out(const float* zs, int nr, int D,
                                              F&& f) {
  __syncthreads();
  const int lane = threadIdx.x;
  const int total = nr * D;
  for (int k = lane; k < total; k += kMvnRows) {
    const int rr = k / D, c = k - rr * D;
    f(rr, c, k, zs[c * kMvnPitch + rr]);
  }
}

template <bool SHARED>
__global__ __launch_bounds__(kMvnRows) void mvn_tril_log_prob_kernel(
    const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ tril, int64_t n_rows, int D, int64_t mean_rows,
    int64_t tril_count, float* __restrict__ log_prob,
    float* __restrict__ grad_x, float* __restrict__ z_out) {
  extern __shared__ float zs[];
  const int lane = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * kMvnRows;
  const int nr = mvn_stage_rows(n_rows, row0);
  const int64_t base = row0 * D;
  mvn_stage_in(zs, nr, D, [&](int rr, int c, int k) {
    return x[base + k] - mean[((row0 + rr) % mean_rows) * D + c];
  });
  const int64_t r = row0 + lane;
  const bool live = lane < nr;
  const float* __restrict__ L =
      SHARED ? tril : tril + (live ? r % tril_count : 0) * (int64_t)D * D;
  float* col = zs + lane;

  // forward substitution L z = d, in place
  float log_diag = 0.f, ss = 0.f;
  for (int i = 0; i < D; ++i) {
    const float* __restrict__ Li = L + (int64_t)i * D;
    float a0 = col[i * kMvnPitch], a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = 0;
    for (; j + 4 <= i; j += 4) {
      a0 = fmaf(-Li[j], col[j * kMvnPitch], a0);
      a1 = fmaf(-Li[j + 1], col[(j + 1) * kMvnPitch], a1);
      a2 = fmaf(-Li[j + 2], col[(j + 2) * kMvnPitch], a2);
      a3 = fmaf(-Li[j + 3], col[(j + 3) * kMvnPitch], a3);
    }
    for (; j < i; ++j) a0 = fmaf(-Li[j], col[j * kMvnPitch], a0);
    const float lii = Li[i];
    const float z = ((a0 + a1) + (a2 + a3)) / lii;
    col[i * kMvnPitch] = z;
    ss = fmaf(z, z, ss);
    log_diag += logf(lii);
  }
  if (live) {
    const float log_z = -0.5f * (float)D * 1.8378770664093453f - log_diag;
    log_prob[r] = log_z + (-0.5f * ss);
  }
  if (z_out)
    mvn_stage_out(zs, nr, D,
                  [&](int, int, int k, float v) { z_out[base + k] = v; });
  if (!grad_x) return;
  __syncthreads();

  // back substitution L^T w = z (right-looking, rows of L), in place
  for (int j = D - 1; j >= 0; --j) {
    const float* __restrict__ Lj = L + (int64_t)j * D;
    const float w = col[j * kMvnPitch] / Lj[j];
    col[j * kMvnPitch] = w;
    for (int i = 0; i < j; ++i)
      col[i * kMvnPitch] = fmaf(-Lj[i], w, col[i * kMvnPitch]);
  }
  mvn_stage_out(zs, nr, D,
                [&](int, int, int k, float v) { grad_x[base + k] = -v; });
}

template <bool SHARED>
__global__ __launch_bounds__(kMvnRows) void mvn_tril_sample_kernel(
    float* __restrict__ out, const float* __restrict__ mean,
    const float* __restrict__ tril, int64_t n_rows, int D, int64_t mean_rows,
    int64_t tril_count, uint32_t k0, uint32_t k1, uint32_t offset) {
  extern __shared__ float zs[];
  const int lane = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * kMvnRows;
  const int nr = mvn_stage_rows(n_rows, row0);
  const int64_t base = row0 * D;  // multiple of 64: Philox groups stay aligned
  // noise element i (flat over [n_rows, D]) = word i % 4 of Philox group i / 4
  const int total = nr * D;
  for (int g = lane; g < (total + 3) / 4; g += kMvnRows) {
    const uint64_t grp = (uint64_t)(base / 4) + (uint64_t)g;
    float z[4];
    normal4((uint32_t)grp, (uint32_t)(grp >> 32), offset, kStreamDist, k0, k1,
            z[0], z[1], z[2], z[3]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = g * 4 + q;
      if (k < total) {
        const int rr = k / D, c = k - rr * D;
        zs[c * kMvnPitch + rr] = z[q];
      }
    }
  }
  if (lane >= nr)
    for (int c = 0; c < D; ++c) zs[c * kMvnPitch + lane] = 0.f;
  __syncthreads();
  const int64_t r = row0 + lane;
  const bool live = lane < nr;
  const float* __restrict__ L =
      SHARED ? tril : tril + (live ? r % tril_count : 0) * (int64_t)D * D;
  float* col = zs + lane;
  // y_i = sum_{j <= i} L_ij n_j, i descending so that y may overwrite n
  for (int i = D - 1; i >= 0; --i) {
    const float* __restrict__ Li = L + (int64_t)i * D;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = 0;
    for (; j + 4 <= i + 1; j += 4) {
      a0 = fmaf(Li[j], col[j * kMvnPitch], a0);
      a1 = fmaf(Li[j + 1], col[(j + 1) * kMvnPitch], a1);
      a2 = fmaf(Li[j + 2], col[(j + 2) * kMvnPitch], a2);
      a3 = fmaf(Li[j + 3], col[(j + 3) * kMvnPitch], a3);
    }
    for (; j <= i; ++j) a0 = fmaf(Li[j], col[j * kMvnPitch], a0);
    col[i * kMvnPitch] = (a0 + a1) + (a2 + a3);
  }
  mvn_stage_out(zs, nr, D, [&](int rr, int c, int k, float v) {
    out[base + k] = v + mean[((row0 + rr) % mean_rows) * D + c];
  });
}

static int mvn_check(const char* who, const void* a, const void* b, const void* c,
                     const void* d, int64_t n_rows, int64_t n_dim,
                     int64_t mean_rows, int64_t tril_count) {
  ZS_REQUIRE(a && b && c && d, "%s: null pointer", who);
  ZS_REQUIRE(n_rows > 0 && n_dim >= 1 && n_dim <= 512,
             "%s: n_rows > 0 and 1 <= n_dim <= 512 expected, got %lld, %lld", who,
             (long long)n_rows, (long long)n_dim);
  ZS_REQUIRE(mean_rows >= 1 && tril_count >= 1, "%s: bad parameter period", who);
  return ZSHMC_OK;
}

template <typename K>
static int mvn_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return ZSHMC_OK;
  return check_hip(
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes),
      "hipFuncSetAttribute(LDS)");
}

}  // namespace zshmc

using namespace zshmc;

extern "C" int zshmc_mvn_tril_log_prob(const float* x, const float* mean,
                                       const float* tril, int64_t n_rows,
                                       int64_t n_dim, int64_t mean_rows,
                                       int64_t tril_count, float* log_prob,
                                       float* grad_x, float* z_out,
                                       void* stream) {
  if (n_rows == 0) return ZSHMC_OK;
  int rc = mvn_check("zshmc_mvn_tril_log_prob", x, mean, tril, log_prob, n_rows,
                     n_dim, mean_rows, tril_count);
  if (rc != ZSHMC_OK) return rc;
  const size_t lds = (size_t)n_dim * kMvnPitch * sizeof(float);
  const dim3 grid((unsigned)((n_rows + kMvnRows - 1) / kMvnRows));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (tril_count == 1) {
    rc = mvn_lds(mvn_tril_log_prob_kernel<true>, lds);
    if (rc != ZSHMC_OK) return rc;
    hipLaunchKernelGGL(mvn_tril_log_prob_kernel<true>, grid, dim3(kMvnRows), lds,
                       s, x, mean, tril, n_rows, (int)n_dim, mean_rows,
                       tril_count, log_prob, grad_x, z_out);
  } else {
    rc = mvn_lds(mvn_tril_log_prob_kernel<false>, lds);
    if (rc != ZSHMC_OK) return rc;
    hipLaunchKernelGGL(mvn_tril_log_prob_kernel<false>, grid, dim3(kMvnRows), lds,
                       s, x, mean, tril, n_rows, (int)n_dim, mean_rows,
                       tril_count, log_prob, grad_x, z_out);
  }
  ZS_LAUNCH_CHECK("mvn_tril_log_prob_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_mvn_tril_sample(float* out, const float* mean,
                                     const float* tril, int64_t n_rows,
                                     int64_t n_dim, int64_t mean_rows,
                                     int64_t tril_count, uint64_t seed,
                                     uint32_t offset, void* stream) {
  if (n_rows == 0) return ZSHMC_OK;
  int rc = mvn_check("zshmc_mvn_tril_sample", out, mean, tril, out, n_rows, n_dim,
                     mean_rows, tril_count);
  if (rc != ZSHMC_OK) return rc;
  const size_t lds = (size_t)n_dim * kMvnPitch * sizeof(float);
  const dim3 grid((unsigned)((n_rows + kMvnRows - 1) / kMvnRows));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const uint32_t k0 = (uint32_t)(seed & 0xFFFFFFFFull), k1 = (uint32_t)(seed >> 32);
  if (tril_count == 1) {
    rc = mvn_lds(mvn_tril_sample_kernel<true>, lds);
    if (rc != ZSHMC_OK) return rc;
    hipLaunchKernelGGL(mvn_tril_sample_kernel<true>, grid, dim3(kMvnRows), lds, s,
                       out, mean, tril, n_rows, (int)n_dim, mean_rows, tril_count,
                       k0, k1, offset);
  } else {
    rc = mvn_lds(mvn_tril_sample_kernel<false>, lds);
    if (rc != ZSHMC_OK) return rc;
    hipLaunchKernelGGL(mvn_tril_sample_kernel<false>, grid, dim3(kMvnRows), lds, s,
                       out, mean, tril, n_rows, (int)n_dim, mean_rows, tril_count,
                       k0, k1, offset);
  }
  ZS_LAUNCH_CHECK("mvn_tril_sample_kernel launch");
  return ZSHMC_OK;
}
