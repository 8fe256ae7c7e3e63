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

namespace zshmc {

typedef float m4 __attribute__((ext_vector_type(4)));
constexpr float kNegHalfLog2Pi = -0.91893853320467274178f;

struct ModelStepArgs {
  float* q;
  float* p;
  const float* grad_lik;  // [C, grad_stride] or NULL (= 0)
  int64_t grad_stride;
  float* operand;  // [C, operand_stride]: softmax: theta in / theta' out;
  int64_t operand_stride;  // identity: padded q' out (or NULL)
  const float* prior_mean;  // [mean_rows, D], row r uses r % mean_rows
  int64_t mean_rows;
  const float* prior_logstd;  // [logstd_rows, D]
  int64_t logstd_rows;
  const float* mass;  // [D] or NULL
  const float* step_size_dev;
  float step_size_host;
  float kick_scale, drift_scale;
  float lik_scale;  // multiplies log_lik and its gradient (AIS temperature)
  int64_t n_chains, n_data;  // n_data: valid leading columns of a row
  int64_t ld;                // row stride of q, p, the prior rows; mass length
  const float* ll_in;  // [C] or NULL
  float* lp_out;       // [C] or NULL
  float* kinetic;      // [C] or NULL
};

template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int off = WIDTH / 2; off > 0; off >>= 1)
    v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// LANES lanes per row (64, 32 or 16): a wave holds 64 / LANES rows at once.
template <int NV, bool SOFTMAX, int LANES>
__global__ __launch_bounds__(256) void model_kick_drift_kernel(ModelStepArgs a) {
  constexpr int kRows = 64 / LANES;  // rows per wave
  static_assert(NV == 1 || LANES == 64, "narrow rows are one chunk per lane");
  const int lane = threadIdx.x & (LANES - 1);     // lane inside the row group
  const int sub = (threadIdx.x & 63) / LANES;     // which row of the wave
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / 64);
  const int64_t D = a.n_data, LD = a.ld;
  const float eps = a.step_size_dev ? *a.step_size_dev : a.step_size_host;
  const float s2 = a.kick_scale * eps;
  const float s1 = a.drift_scale * eps;
  const m4 zero = m4{0.f, 0.f, 0.f, 0.f};

  // (wave-uniform trip count: the rows past the end are clamped to the last
  // row for the loads and masked for the stores, so every lane takes part in
  // the shuffles)
  for (int64_t base = wave * kRows; base < a.n_chains;
       base += n_waves * kRows) {
    const bool row_on = base + sub < a.n_chains;
    const int64_t c = row_on ? base + sub : a.n_chains - 1;
    float* __restrict__ qrow = a.q + c * LD;
    float* __restrict__ prow = a.p + c * LD;
    const float* __restrict__ mrow = a.prior_mean + (c % a.mean_rows) * LD;
    const float* __restrict__ lrow = a.prior_logstd + (c % a.logstd_rows) * LD;
    m4 q[NV], p[NV], g[NV], im[NV];
    bool in[NV];
    float prior = 0.f, dot = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int64_t d = (int64_t)(k * LANES + lane) * 4;
      in[k] = d < D;
      q[k] = p[k] = g[k] = zero;
      im[k] = m4{1.f, 1.f, 1.f, 1.f};
      if (!in[k]) continue;
      q[k] = *reinterpret_cast<const m4*>(qrow + d);
      p[k] = *reinterpret_cast<const m4*>(prow + d);
      const m4 mu = *reinterpret_cast<const m4*>(mrow + d);
      const m4 ls = *reinterpret_cast<const m4*>(lrow + d);
      if (a.mass) {
        const m4 m = *reinterpret_cast<const m4*>(a.mass + d);
#pragma unroll
        for (int j = 0; j < 4; ++j) im[k][j] = 1.0f / m[j];
      }
      m4 gl = zero;
      if (a.grad_lik)
        gl = a.lik_scale *
             *reinterpret_cast<const m4*>(a.grad_lik + c * a.grad_stride + d);
      m4 th = zero;
      if (SOFTMAX && a.grad_lik)
        th = *reinterpret_cast<const m4*>(a.operand + c * a.operand_stride + d);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // Normal._log_prob and d/dx, univariate.py:174-181
        const bool on = d + j < D;  // (the last group of a padded row)
        const float prec = expf(-2.0f * ls[j]);
        const float r = q[k][j] - mu[j];
        prior += on ? kNegHalfLog2Pi - ls[j] - 0.5f * prec * r * r : 0.f;
        g[k][j] = on ? -prec * r : 0.f;  // prior part; the likelihood part
                                         // joins below
        if (SOFTMAX) dot += gl[j] * th[j];
      }
      // likelihood part: J_f^T g_lik.  softmax: theta * (g - <g, theta>);
      // theta * g joins now, the <g, theta> term after the row sum (theta is
      // re-read from the operand row then: one L2-resident 16-B load instead
      // of 4*NV live registers)
      if (SOFTMAX)
        g[k] += th * gl;
      else
        g[k] += gl;
    }
    prior = group_sum<LANES>(prior);
    if (SOFTMAX && a.grad_lik) {
      dot = group_sum<LANES>(dot);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        if (!in[k]) continue;
        const int64_t d = (int64_t)(k * LANES + lane) * 4;
        const m4 th =
            *reinterpret_cast<const m4*>(a.operand + c * a.operand_stride + d);
        g[k] -= dot * th;
      }
    }
    // kick, drift (hmc.py:38-43), kinetic energy of the new momentum
    float kin = 0.f, qmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (!in[k]) continue;
      const int64_t d = (int64_t)(k * LANES + lane) * 4;
      p[k] = p[k] + s2 * g[k];
      const m4 vel = p[k] * im[k];
      if (a.kick_scale != 0.f && row_on)
        *reinterpret_cast<m4*>(prow + d) = p[k];
      if (a.drift_scale != 0.f) {
        q[k] = q[k] + s1 * vel;
        if (row_on) *reinterpret_cast<m4*>(qrow + d) = q[k];
      }
      const m4 e = p[k] * vel;
      kin += (e[0] + e[1]) + (e[2] + e[3]);
      if (SOFTMAX) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (d + j < D) qmax = fmaxf(q[k][j], qmax);
      }
    }
    if (a.kinetic) {
      kin = group_sum<LANES>(kin);
      if (lane == 0 && row_on) a.kinetic[c] += 0.5f * kin;
    }
    if (a.lp_out && lane == 0 && row_on)
      a.lp_out[c] = (a.ll_in ? a.lik_scale * a.ll_in[c] : 0.f) + prior;
    // operand of the next likelihood evaluation: f(q_new), zero padding
    if (a.operand) {
      float inv_sum = 1.f;
      if (SOFTMAX) {
        qmax = group_max<LANES>(qmax);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          if (!in[k]) continue;
          const int64_t d = (int64_t)(k * LANES + lane) * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            q[k][j] = d + j < D ? expf(q[k][j] - qmax) : 0.f;
            sum += q[k][j];
          }
        }
        inv_sum = 1.0f / group_sum<LANES>(sum);
      }
      float* __restrict__ orow = a.operand + c * a.operand_stride;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int64_t d = (int64_t)(k * LANES + lane) * 4;
        if (d >= a.operand_stride || !row_on) continue;
        *reinterpret_cast<m4*>(orow + d) = in[k] ? q[k] * inv_sum : zero;
      }
    }
  }
}

template <int NV, int LANES>
static int launch_model_step(const ModelStepArgs& a, bool softmax,
                             hipStream_t s) {
  constexpr int kRows = 64 / LANES;
  const int64_t need = (a.n_chains + 4 * kRows - 1) / (4 * kRows);
  const int64_t cap = (int64_t)device_cu_count() * 8;
  const dim3 grid((unsigned)(need < cap ? need : cap)), block(256);
  if (softmax)
    hipLaunchKernelGGL((model_kick_drift_kernel<NV, true, LANES>), grid, block,
                       0, s, a);
  else
    hipLaunchKernelGGL((model_kick_drift_kernel<NV, false, LANES>), grid, block,
                       0, s, a);
  ZS_LAUNCH_CHECK("model_kick_drift_kernel launch");
  return ZSHMC_OK;
}

}  // namespace zshmc

using namespace zshmc;

extern "C" int zshmc_model_kick_drift(
    float* q, float* p, const float* grad_lik, int64_t grad_stride,
    float* operand, int64_t operand_stride, int softmax,
    const float* prior_mean, int64_t mean_rows, const float* prior_logstd,
    int64_t logstd_rows, const float* mass, const float* step_size_dev,
    float step_size_host, float kick_scale, float drift_scale,
    float lik_scale, int64_t n_chains, int64_t n_data, int64_t row_stride,
    const float* ll_in, float* lp_out, float* kinetic, void* stream) {
  const int64_t ld = row_stride;
  ZS_REQUIRE(q && p && prior_mean && prior_logstd,
             "zshmc_model_kick_drift: null q/p/prior");
  ZS_REQUIRE(n_chains >= 0 && n_data >= 1 && ld >= n_data && ld <= 1024 &&
                 ld % 4 == 0,
             "zshmc_model_kick_drift: 1 <= n_data %lld <= row_stride %lld <= "
             "1024, row_stride a multiple of 4", (long long)n_data,
             (long long)ld);
  ZS_REQUIRE(mean_rows >= 1 && logstd_rows >= 1,
             "zshmc_model_kick_drift: prior row periods must be >= 1");
  ZS_REQUIRE(!grad_lik || (grad_stride >= ld && grad_stride % 4 == 0),
             "zshmc_model_kick_drift: bad grad_stride");
  ZS_REQUIRE(!operand || (operand_stride >= ld && operand_stride % 4 == 0 &&
                          operand_stride <= 1024),
             "zshmc_model_kick_drift: bad operand_stride");
  ZS_REQUIRE(!softmax || operand, "zshmc_model_kick_drift: softmax needs operand");
  const uintptr_t align =
      reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(p) |
      reinterpret_cast<uintptr_t>(grad_lik) | reinterpret_cast<uintptr_t>(operand) |
      reinterpret_cast<uintptr_t>(prior_mean) |
      reinterpret_cast<uintptr_t>(prior_logstd) | reinterpret_cast<uintptr_t>(mass);
  ZS_REQUIRE((align & 15) == 0, "zshmc_model_kick_drift: buffers must be 16-B aligned");
  if (n_chains == 0) return ZSHMC_OK;
  ModelStepArgs a{q, p, grad_lik, grad_stride, operand, operand_stride,
                  prior_mean, mean_rows, prior_logstd, logstd_rows, mass,
                  step_size_dev, step_size_host, kick_scale, drift_scale,
                  lik_scale, n_chains, n_data, ld, ll_in, lp_out, kinetic};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t width = operand && operand_stride > ld ? operand_stride : ld;
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
