// The leapfrog step of the NATIVE plans for the dense-likelihood families
// (BASELINE configs 3 and 5): everything of one trip of HMC._leapfrog
// (reference zhusuan/hmc.py:348-372, leapfrog_integrator :38-43) that is not
// the likelihood's two GEMMs, in ONE element-wise launch -- no autograd graph,
// no ATen kernel on the path:
//
//   model        log p(q) = log N(q; mean, exp(logstd))            (prior node,
//                           univariate.py:174-181, group_ndims = 1)
//                         + log_lik(f(q))                          (fused MFMA
//                           kernel of csrc/linear_bernoulli.hip)
//   f = identity  Bayesian logistic regression, logits = q . X^T
//                 (univariate.py:398-403 through tf.matmul)
//   f = softmax   logistic-normal topic model, theta = softmax(eta)
//                 (examples/topic_models/lntm_mcem.py:39-46)
//
// Given d log_lik / d f(q) from the MFMA kernel evaluated at the current q,
// one launch does, per chain (row):
//   prior_lp   = sum_d log N(q_d)              and  d prior / dq = -prec (q - mean)
//   grad       = J_f(q)^T g_lik + d prior/dq   (softmax: theta * (g - <g, theta>),
//                                               what tf.gradients gives, hmc.py:430-432)
//   p <- p + kick_scale * eps * grad           (hmc.py:42)
//   q <- q + drift_scale * eps * p / mass      (hmc.py:39, :26-27)
//   lp_out     = lik_scale * ll_in + prior_lp  (log-joint AT the evaluation
//                                               point: old / new log-prob)
//   kinetic   += 1/2 sum p^2 / mass            (hmc.py:32-34, on request)
//   operand    = f(q_new), zero-padded to the MFMA kernel's feature width --
//                the next likelihood evaluation reads it directly.
// One wave per row -- two / four rows per wave when a row is at most 128 / 64
// floats wide, so that no lane idles (config 5: K = 128 topics) -- the row in
// registers (row stride ld <= 1024 floats, a multiple of 4, 16-B aligned; the
// n_data <= ld leading columns are the latent -- several latents packed side by
// side by the caller -- the rest is zero padding that takes no part in the
// prior, the softmax or the sums), 16 B per lane and access, row sums by
// shuffles inside the row's lane group.
// HBM-bound: 4-6 row passes of 4*n_data bytes per call, noise next to the
// 2*N*D*C flop likelihood it sits between.
#include "common.h"
#include "model_step.h"

namespace zshmc {

template <int NV, bool SOFTMAX, int LANES, bool PARTS>
__global__ __launch_bounds__(256) void model_kick_drift_kernel(ModelStepArgs a) {
  // (csrc/model_step.h)
  model_step_rows<NV, SOFTMAX, LANES, PARTS>(
      a, (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64,
      (int64_t)gridDim.x * (blockDim.x / 64));
}

// Few rows behind a split likelihood launch (a.n_parts > 1): what the step
// waits for is the n_parts loads per element, a handful of dependent batches
// from as many workgroups' L2s.  One workgroup per wave-load of rows, its
// FOUR waves adding a quarter of the partials each -- wave w the accumulators
// 2w and 2w + 1 of sum_parts8 (csrc/common.h), folded through LDS in
// sum_parts8's order: the same bits, one batch of loads instead of five --
// then wave 0 takes the step.
template <int NV, bool SOFTMAX, int LANES>
__global__ __launch_bounds__(256) void model_kick_drift_parts_kernel(
    ModelStepArgs a) {
  constexpr int kRows = 64 / LANES;
  constexpr int kCh = 16;  // loads in flight per accumulator
  __shared__ m4 sh[4][kRows * NV * LANES];
  const int w = threadIdx.x / 64;
  const int lane = threadIdx.x & (LANES - 1);
  const int sub = (threadIdx.x & 63) / LANES;
  const int64_t row = (int64_t)blockIdx.x * kRows + sub;
  const bool row_on = row < a.n_chains;
  const int64_t c = row_on ? row : a.n_chains - 1;
  const int S = a.n_parts;
  const int64_t stride = a.part_stride / 4;
  const m4 zero = m4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int64_t d = (int64_t)(k * LANES + lane) * 4;
    m4 a0 = zero, a1 = zero;
    if (d < a.n_data) {
      const m4* __restrict__ p =
          reinterpret_cast<const m4*>(a.grad_lik + c * a.grad_stride + d);
      for (int s0 = 2 * w; s0 < S; s0 += 8 * kCh) {
        m4 v0[kCh], v1[kCh];
#pragma unroll
        for (int j = 0; j < kCh; ++j) {
          const int s = s0 + 8 * j;
          v0[j] = s < S ? p[(int64_t)s * stride] : zero;
          v1[j] = s + 1 < S ? p[(int64_t)(s + 1) * stride] : zero;
        }
#pragma unroll
        for (int j = 0; j < kCh; ++j) {
          const int s = s0 + 8 * j;
          if (s < S) a0 += v0[j];
          if (s + 1 < S) a1 += v1[j];
        }
      }
    }
    sh[w][(sub * NV + k) * LANES + lane] = a0 + a1;
  }
  __syncthreads();
  if (w != 0) return;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = (sub * NV + k) * LANES + lane;
    const m4 gl = (sh[0][i] + sh[1][i]) + (sh[2][i] + sh[3][i]);
    sh[0][i] = gl;  // (read back by the lane that wrote it)
    const int64_t d = (int64_t)(k * LANES + lane) * 4;
    if (a.grad_sum && row_on && d < a.n_data)
      *reinterpret_cast<m4*>(a.grad_sum + c * a.grad_stride + d) = gl;
  }
  ModelStepArgs b = a;
  b.grad_ready = reinterpret_cast<const float*>(&sh[0][0]);
  b.grad_sum = nullptr;
  model_step_rows<NV, SOFTMAX, LANES>(b, (int64_t)blockIdx.x,
                                      (int64_t)gridDim.x);
}

template <int NV, int LANES>
static int launch_model_step(const ModelStepArgs& a, bool softmax,
                             hipStream_t s) {
  constexpr int kRows = 64 / LANES;
  const int64_t groups = (a.n_chains + kRows - 1) / kRows;
  if (a.n_parts > 1 && a.grad_lik && groups <= 2 * (int64_t)device_cu_count()) {
    const dim3 grid((unsigned)groups), block(256);
    if (softmax)
      hipLaunchKernelGGL((model_kick_drift_parts_kernel<NV, true, LANES>), grid,
                         block, 0, s, a);
    else
      hipLaunchKernelGGL((model_kick_drift_parts_kernel<NV, false, LANES>),
                         grid, block, 0, s, a);
    ZS_LAUNCH_CHECK("model_kick_drift_parts_kernel launch");
    return ZSHMC_OK;
  }
  const int64_t need = (a.n_chains + 4 * kRows - 1) / (4 * kRows);
  const int64_t cap = (int64_t)device_cu_count() * 8;
  const dim3 grid((unsigned)(need < cap ? need : cap)), block(256);
  // (the instantiation with the partial-sum path only where there are
  // partials: it costs the kernel its occupancy, csrc/model_step.h)
  const bool parts = a.n_parts > 1;
  if (softmax && parts)
    hipLaunchKernelGGL((model_kick_drift_kernel<NV, true, LANES, true>), grid,
                       block, 0, s, a);
  else if (softmax)
    hipLaunchKernelGGL((model_kick_drift_kernel<NV, true, LANES, false>), grid,
                       block, 0, s, a);
  else if (parts)
    hipLaunchKernelGGL((model_kick_drift_kernel<NV, false, LANES, true>), grid,
                       block, 0, s, a);
  else
    hipLaunchKernelGGL((model_kick_drift_kernel<NV, false, LANES, false>), grid,
                       block, 0, s, a);
  ZS_LAUNCH_CHECK("model_kick_drift_kernel launch");
  return ZSHMC_OK;
}

}  // namespace zshmc

using namespace zshmc;

// the step with its arguments as the kernel takes them (also for
// csrc/hmc_model_run.hip, whose steps read a split likelihood launch's
// partials: a.n_parts > 1)
int zshmc::model_kick_drift_launch(const ModelStepArgs& a, int softmax,
                                   void* stream) {
  const int64_t ld = a.ld;
  ZS_REQUIRE(a.q && a.p && a.prior_mean && a.prior_logstd,
             "zshmc_model_kick_drift: null q/p/prior");
  ZS_REQUIRE(a.n_chains >= 0 && a.n_data >= 1 && ld >= a.n_data && ld <= 1024 &&
                 ld % 4 == 0,
             "zshmc_model_kick_drift: 1 <= n_data %lld <= row_stride %lld <= "
             "1024, row_stride a multiple of 4", (long long)a.n_data,
             (long long)ld);
  ZS_REQUIRE(a.mean_rows >= 1 && a.logstd_rows >= 1,
             "zshmc_model_kick_drift: prior row periods must be >= 1");
  ZS_REQUIRE(!a.grad_lik || (a.grad_stride >= ld && a.grad_stride % 4 == 0),
             "zshmc_model_kick_drift: bad grad_stride");
  ZS_REQUIRE(!a.operand || (a.operand_stride >= ld && a.operand_stride % 4 == 0 &&
                            a.operand_stride <= 1024),
             "zshmc_model_kick_drift: bad operand_stride");
  ZS_REQUIRE(!softmax || a.operand, "zshmc_model_kick_drift: softmax needs operand");
  ZS_REQUIRE(a.n_parts <= 1 || (a.grad_lik && a.part_stride % 4 == 0),
             "zshmc_model_kick_drift: bad partials");
  const uintptr_t align =
      reinterpret_cast<uintptr_t>(a.q) | reinterpret_cast<uintptr_t>(a.p) |
      reinterpret_cast<uintptr_t>(a.grad_lik) |
      reinterpret_cast<uintptr_t>(a.operand) |
      reinterpret_cast<uintptr_t>(a.prior_mean) |
      reinterpret_cast<uintptr_t>(a.prior_logstd) |
      reinterpret_cast<uintptr_t>(a.mass) | reinterpret_cast<uintptr_t>(a.grad_sum);
  ZS_REQUIRE((align & 15) == 0, "zshmc_model_kick_drift: buffers must be 16-B aligned");
  if (a.n_chains == 0) return ZSHMC_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t width =
      a.operand && a.operand_stride > ld ? a.operand_stride : ld;
  const int nv = (int)((width + 255) / 256);
  // a row of at most 64 / 128 floats leaves lanes of a 64-lane group idle:
  // pack four / two rows into a wave
  if (width <= 64) return launch_model_step<1, 16>(a, softmax != 0, s);
  if (width <= 128) return launch_model_step<1, 32>(a, softmax != 0, s);
  switch (nv) {
    case 1: return launch_model_step<1, 64>(a, softmax != 0, s);
    case 2: return launch_model_step<2, 64>(a, softmax != 0, s);
    case 3: return launch_model_step<3, 64>(a, softmax != 0, s);
    default: return launch_model_step<4, 64>(a, softmax != 0, s);
  }
}

extern "C" int zshmc_model_kick_drift(
    float* q, float* p, const float* grad_lik, int64_t grad_stride,
    float* operand, int64_t operand_stride, int softmax,
    const float* prior_mean, int64_t mean_rows, const float* prior_logstd,
    int64_t logstd_rows, const float* mass, const float* step_size_dev,
    float step_size_host, float kick_scale, float drift_scale,
    float lik_scale, int64_t n_chains, int64_t n_data, int64_t row_stride,
    const float* ll_in, float* lp_out, float* kinetic, void* stream) {
  ModelStepArgs a{q, p, grad_lik, grad_stride, operand, operand_stride,
                  prior_mean, mean_rows, prior_logstd, logstd_rows, mass,
                  step_size_dev, step_size_host, kick_scale, drift_scale,
                  lik_scale, n_chains, n_data, row_stride, ll_in, lp_out,
                  kinetic, 1, 0, nullptr, nullptr, nullptr};
  return model_kick_drift_launch(a, softmax, stream);
}
