// On-device adaptation state for the HMC sampler (gfx950).
//
//   zshmc_stepsize_update : StepsizeTuner.tune, reference zhusuan/hmc.py:89-112
//                           + HMC._adapt_step_size :375-380
//   zshmc_mass_colstats / zshmc_mass_update :
//                           ExponentialWeightedMovingVariance :115-159
//                           + HMC._adapt_mass :284-305
//
// All state stays in device memory (the float32 block described in
// include/zshmc.h) so consecutive transitions enqueue back-to-back with no
// host round trip; the only value that ever crosses GPUs is the tiny
// acc_sum / colsum buffer.
#include "common.h"
#include "fused_args.h"

namespace zshmc {

__global__ void stepsize_update_kernel(float* __restrict__ state,
                                       double* __restrict__ acc_sum,
                                       double inv_chains, int adapt,
                                       float fresh, TunerCfg cfg) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  // the same arithmetic as the update carried by the fused kernels
  AdaptLink k;
  k.state = state;
  k.stats = acc_sum;
  k.inv_chains = inv_chains;
  k.fresh = fresh;
  k.used_step_size = __builtin_nanf("");
  k.tuner = cfg;
  tuner_persist(k, adapt ? ZSHMC_PEND_ADAPT : ZSHMC_PEND_HOLD, *acc_sum);
  *acc_sum = 0.0;  // consumed; ready for the next transition
}

__global__ void state_set_kernel(float* state, int index, float value) {
  if (threadIdx.x == 0 && blockIdx.x == 0) state[index] = value;
}

// Column sums over chains of (q - m) and (q - m)^2, double accumulation.
// Block (64 x 4): 64 columns wide (one wave = one 256 B coalesced row
// segment), 4 row lanes; each block strides over rows, then LDS-reduces its
// 4 row lanes and issues one double atomic per column.
constexpr int kColsPerBlock = 64;
constexpr int kRowLanes = 4;

__global__ __launch_bounds__(256) void mass_colstats_kernel(
    const float* __restrict__ q, const float* __restrict__ ewmv_mean,
    int64_t n_chains, int64_t n_data, double* __restrict__ colsum,
    int row_blocks) {
  const int tx = threadIdx.x % kColsPerBlock;
  const int ty = threadIdx.x / kColsPerBlock;
  const int64_t col = (int64_t)blockIdx.x * kColsPerBlock + tx;
  const bool valid = col < n_data;
  const float m = valid ? ewmv_mean[col] : 0.f;
  double s1 = 0.0, s2 = 0.0;
  if (valid) {
    for (int64_t row = (int64_t)blockIdx.y * kRowLanes + ty; row < n_chains;
         row += (int64_t)row_blocks * kRowLanes) {
      const float d = q[row * n_data + col] - m;
      s1 += (double)d;
      s2 += (double)d * (double)d;
    }
  }
  __shared__ double sh1[kRowLanes][kColsPerBlock];
  __shared__ double sh2[kRowLanes][kColsPerBlock];
  sh1[ty][tx] = s1;
  sh2[ty][tx] = s2;
  __syncthreads();
  if (ty == 0 && valid) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int i = 0; i < kRowLanes; ++i) {
      a += sh1[i][tx];
      b += sh2[i][tx];
    }
    atomicAdd(&colsum[col], a);
    atomicAdd(&colsum[n_data + col], b);
  }
}

__global__ void mass_update_kernel(float* __restrict__ state,
                                   float* __restrict__ ewmv_mean,
                                   float* __restrict__ ewmv_var,
                                   double* __restrict__ colsum,
                                   double inv_chains, int64_t n_data,
                                   float decay, int update, int use_ones,
                                   float* __restrict__ mass_out) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // every thread reads tau BEFORE thread 0 of block 0 bumps it: the bump is
  // done by a second tiny launch (mass_tau_bump_kernel) to avoid the race.
  const float tau_new = state[ZSHMC_ST_EWMV_T] + 1.0f;
  if (d >= n_data) return;
  float var = ewmv_var[d];
  if (update) {
    // hmc.py:130-148 with S1 = mean_c(q-m), S2 = mean_c((q-m)^2):
    //   incr = w (q-m); mean' = m + w S1;
    //   var' = (1-w) var + mean_c(incr (q-mean')) = (1-w) var + w S2 - (w S1)^2
    const float w = (1.0f - decay) / (1.0f - powf(decay, tau_new));
    const double s1 = colsum[d] * inv_chains;
    const double s2 = colsum[n_data + d] * inv_chains;
    colsum[d] = 0.0;  // consumed; ready for the next colstats pass
    colsum[n_data + d] = 0.0;
    const double delta = (double)w * s1;
    ewmv_mean[d] = (float)((double)ewmv_mean[d] + delta);
    var = (float)((1.0 - (double)w) * (double)var + (double)w * s2 -
                  delta * delta);
    ewmv_var[d] = var;
  }
  // hmc.py:299-302 (ones while int(t) < mass_collect_iters), :151-152 (1/var,
  // no floor -- inf when var == 0, Appendix B #5)
  mass_out[d] = use_ones ? 1.0f : 1.0f / var;
}

// Rows of column sums -> one row, fixed order.  Block (16 x 64): 16 columns
// (one 128-byte segment per row), 64 row lanes striding over the rows -- the
// 4 MB of partials of a 256-workgroup launch at D = 1 024 are read by 128 / 64
// blocks with every load in flight at once (latency-, not bandwidth-bound);
// LDS-reduced in lane order.
constexpr int kPartCols = 16;
constexpr int kPartLanes = 64;

__device__ __forceinline__ double parts_column_sum(
    const double* __restrict__ parts, int64_t n_parts, int64_t stride,
    int64_t col, bool valid, double (*sh)[kPartCols]) {
  const int tx = threadIdx.x % kPartCols;
  const int ty = threadIdx.x / kPartCols;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (valid) {
    int64_t r = ty;
    for (; r + 3 * kPartLanes < n_parts; r += 4 * kPartLanes) {
      s0 += parts[r * stride + col];
      s1 += parts[(r + kPartLanes) * stride + col];
      s2 += parts[(r + 2 * kPartLanes) * stride + col];
      s3 += parts[(r + 3 * kPartLanes) * stride + col];
    }
    for (; r < n_parts; r += kPartLanes) s0 += parts[r * stride + col];
  }
  __syncthreads();  // (sh may still be read from a previous call)
  sh[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  // 64 -> 4 by four threads per column, then lane order
  double t = 0.0;
  if (ty < 4) {
#pragma unroll
    for (int i = 0; i < kPartLanes / 4; ++i)
      t += sh[ty * (kPartLanes / 4) + i][tx];
  }
  __syncthreads();
  if (ty < 4) sh[ty][tx] = t;
  __syncthreads();
  return (sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]);
}

__global__ __launch_bounds__(1024) void mass_colstats_reduce_kernel(
    const double* __restrict__ parts, int64_t n_parts, int64_t n_data,
    double* __restrict__ colsum) {
  __shared__ double sh[kPartLanes][kPartCols];
  const int tx = threadIdx.x % kPartCols;
  const int ty = threadIdx.x / kPartCols;
  // blockIdx.x walks the 2*n_data columns of a row
  const int64_t col = (int64_t)blockIdx.x * kPartCols + tx;
  const bool valid = col < 2 * n_data;
  const double tot =
      parts_column_sum(parts, n_parts, 2 * n_data, col, valid, sh);
  if (ty == 0 && valid) colsum[col] = tot;
}

// zshmc_mass_update(update = 1) fed by rows of column sums, tau advanced by
// the block that finishes last (every block has read tau by then: the
// increment of the retirement counter comes after the block's barrier).
__global__ __launch_bounds__(1024) void mass_update_fused_kernel(
    float* __restrict__ state, float* __restrict__ ewmv_mean,
    float* __restrict__ ewmv_var, const double* __restrict__ parts,
    int64_t n_parts, double inv_chains, int64_t n_data, float decay,
    int use_ones, float* __restrict__ mass_out,
    unsigned int* __restrict__ retired) {
  __shared__ double sh[kPartLanes][kPartCols];
  const int tx = threadIdx.x % kPartCols;
  const int ty = threadIdx.x / kPartCols;
  const int64_t d = (int64_t)blockIdx.x * kPartCols + tx;
  const bool valid = d < n_data;
  const float tau_new = state[ZSHMC_ST_EWMV_T] + 1.0f;
  const double c1 = parts_column_sum(parts, n_parts, 2 * n_data, d, valid, sh);
  const double c2 =
      parts_column_sum(parts, n_parts, 2 * n_data, n_data + d, valid, sh);
  if (ty == 0 && valid) {
    // the arithmetic of mass_update_kernel (hmc.py:130-152)
    const float w = (1.0f - decay) / (1.0f - powf(decay, tau_new));
    const double s1 = c1 * inv_chains, s2 = c2 * inv_chains;
    const double delta = (double)w * s1;
    ewmv_mean[d] = (float)((double)ewmv_mean[d] + delta);
    const float var = (float)((1.0 - (double)w) * (double)ewmv_var[d] +
                              (double)w * s2 - delta * delta);
    ewmv_var[d] = var;
    mass_out[d] = use_ones ? 1.0f : 1.0f / var;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int before = __hip_atomic_fetch_add(
        retired, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before == gridDim.x - 1) {
      state[ZSHMC_ST_EWMV_T] = tau_new;
      __hip_atomic_store(retired, 0u, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ void mass_tau_bump_kernel(float* state) {
  if (threadIdx.x == 0 && blockIdx.x == 0) state[ZSHMC_ST_EWMV_T] += 1.0f;
}

}  // namespace zshmc

using namespace zshmc;

extern "C" int zshmc_stepsize_update(float* state, double* acc_sum,
                                     int64_t n_chains_global, int adapt,
                                     int fresh_start, float delta, float gamma,
                                     float t0, float kappa, float mu,
                                     void* stream) {
  ZS_REQUIRE(state && acc_sum, "zshmc_stepsize_update: null state/acc_sum");
  ZS_REQUIRE(n_chains_global > 0, "zshmc_stepsize_update: n_chains_global <= 0");
  hipLaunchKernelGGL(stepsize_update_kernel, dim3(1), dim3(64), 0,
                     reinterpret_cast<hipStream_t>(stream), state, acc_sum,
                     1.0 / (double)n_chains_global, adapt,
                     fresh_start ? 1.0f : 0.0f,
                     TunerCfg{delta, gamma, t0, kappa, mu});
  ZS_LAUNCH_CHECK("stepsize_update_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_state_set(float* state, int index, float value,
                               void* stream) {
  ZS_REQUIRE(state, "zshmc_state_set: null state");
  ZS_REQUIRE(index >= 0 && index < ZSHMC_STATE_WORDS,
             "zshmc_state_set: index %d out of range", index);
  hipLaunchKernelGGL(state_set_kernel, dim3(1), dim3(64), 0,
                     reinterpret_cast<hipStream_t>(stream), state, index,
                     value);
  ZS_LAUNCH_CHECK("state_set_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_mass_colstats(const float* q, const float* ewmv_mean,
                                   int64_t n_chains, int64_t n_data,
                                   double* colsum, void* stream) {
  ZS_REQUIRE(q && ewmv_mean && colsum, "zshmc_mass_colstats: null pointer");
  ZS_REQUIRE(n_chains >= 0 && n_data >= 1, "zshmc_mass_colstats: bad shape");
  if (n_chains == 0) return ZSHMC_OK;
  const int col_blocks = (int)((n_data + kColsPerBlock - 1) / kColsPerBlock);
  // enough row blocks to fill the chip (~8 blocks per CU), at least 1
  int64_t want = ((int64_t)device_cu_count() * 8 + col_blocks - 1) / col_blocks;
  const int64_t max_rb = (n_chains + kRowLanes - 1) / kRowLanes;
  if (want > max_rb) want = max_rb;
  if (want < 1) want = 1;
  if (want > 65535) want = 65535;
  hipLaunchKernelGGL(mass_colstats_kernel, dim3(col_blocks, (int)want),
                     dim3(256), 0, reinterpret_cast<hipStream_t>(stream), q,
                     ewmv_mean, n_chains, n_data, colsum, (int)want);
  ZS_LAUNCH_CHECK("mass_colstats_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_mass_update(float* state, float* ewmv_mean,
                                 float* ewmv_var, double* colsum,
                                 int64_t n_chains_global, int64_t n_data,
                                 float decay, int update, int use_ones,
                                 float* mass_out, void* stream) {
  ZS_REQUIRE(state && ewmv_mean && ewmv_var && mass_out,
             "zshmc_mass_update: null pointer");
  ZS_REQUIRE(!update || colsum, "zshmc_mass_update: update needs colsum");
  ZS_REQUIRE(n_chains_global > 0 && n_data >= 1, "zshmc_mass_update: bad shape");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int threads = 256;
  const int blocks = (int)((n_data + threads - 1) / threads);
  hipLaunchKernelGGL(mass_update_kernel, dim3(blocks), dim3(threads), 0, s,
                     state, ewmv_mean, ewmv_var, colsum,
                     1.0 / (double)n_chains_global, n_data, decay, update,
                     use_ones, mass_out);
  ZS_LAUNCH_CHECK("mass_update_kernel launch");
  if (update == 1) {  // update == 2: more latents share this EWMV.t tick
    hipLaunchKernelGGL(mass_tau_bump_kernel, dim3(1), dim3(64), 0, s, state);
    ZS_LAUNCH_CHECK("mass_tau_bump_kernel launch");
  }
  return ZSHMC_OK;
}

extern "C" int zshmc_mass_colstats_reduce(const double* parts, int64_t n_parts,
                                          int64_t n_data, double* colsum,
                                          void* stream) {
  ZS_REQUIRE(parts && colsum, "zshmc_mass_colstats_reduce: null pointer");
  ZS_REQUIRE(n_parts >= 1 && n_data >= 1,
             "zshmc_mass_colstats_reduce: bad shape");
  const int blocks = (int)((2 * n_data + kPartCols - 1) / kPartCols);
  hipLaunchKernelGGL(mass_colstats_reduce_kernel, dim3(blocks),
                     dim3(kPartCols * kPartLanes), 0,
                     reinterpret_cast<hipStream_t>(stream), parts, n_parts,
                     n_data, colsum);
  ZS_LAUNCH_CHECK("mass_colstats_reduce_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_mass_update_fused(
    float* state, float* ewmv_mean, float* ewmv_var, const double* parts,
    int64_t n_parts, int64_t n_chains_global, int64_t n_data, float decay,
    int use_ones, float* mass_out, void* workspace, void* stream) {
  ZS_REQUIRE(state && ewmv_mean && ewmv_var && parts && mass_out && workspace,
             "zshmc_mass_update_fused: null pointer");
  ZS_REQUIRE(n_parts >= 1 && n_chains_global > 0 && n_data >= 1,
             "zshmc_mass_update_fused: bad shape");
  const int blocks = (int)((n_data + kPartCols - 1) / kPartCols);
  hipLaunchKernelGGL(mass_update_fused_kernel, dim3(blocks),
                     dim3(kPartCols * kPartLanes), 0,
                     reinterpret_cast<hipStream_t>(stream), state, ewmv_mean,
                     ewmv_var, parts, n_parts, 1.0 / (double)n_chains_global,
                     n_data, decay, use_ones, mass_out,
                     reinterpret_cast<unsigned int*>(workspace));
  ZS_LAUNCH_CHECK("mass_update_fused_kernel launch");
  return ZSHMC_OK;
}
