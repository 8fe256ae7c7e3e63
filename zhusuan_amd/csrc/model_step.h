// The element-wise leapfrog step of the native model plans as a device function
// (see csrc/hmc_model.hip for what it computes and the reference lines).
#pragma once
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
  // The likelihood evaluation as the n_parts row-range partials a split
  // launch left (csrc/lb_body.h: grad part s at grad_lik + s * part_stride,
  // log-likelihood part s at ll_in + s * n_chains): added here in the order
  // and arithmetic of lb_reduce_splits_kernel (sum_parts8, csrc/common.h), and -- where asked -- stored reduced (grad_sum [C, grad_stride],
  // ll_sum [C]).  n_parts <= 1: grad_lik / ll_in are the evaluation itself.
  // (The persistent trajectory kernel, csrc/hmc_model_traj.hip, and the steps
  // of csrc/hmc_model_run.hip behind a split launch.)
  int n_parts;
  int64_t part_stride;
  float* grad_sum;
  float* ll_sum;
  // set by model_kick_drift_parts_kernel (csrc/hmc_model.hip), never by a
  // caller: the partials' sum of this wave's rows, [64 / LANES][NV][LANES]
  // float4 in LDS, formed by the workgroup's four waves together
  const float* grad_ready;
};

// host side (csrc/hmc_model.hip): validate and launch
int model_kick_drift_launch(const ModelStepArgs& a, int softmax, void* stream);

__device__ __forceinline__ float step_ll(const ModelStepArgs& a, int64_t c) {
  if (a.n_parts <= 1) return a.ll_in[c];
  return sum_parts8(a.ll_in + c, a.n_chains, a.n_parts);
}

template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int off = WIDTH / 2; off > 0; off >>= 1)
    v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// LANES lanes per row (64, 32 or 16): a wave holds 64 / LANES rows at once.
// `wave` of `n_waves`: this wave's place among those that share the rows (the
// launch's grid -- or the persistent trajectory kernel's, csrc/hmc_model_traj.hip).
// PARTS = false: an instantiation without the partial-sum path (a.n_parts <=
// 1 is the caller's promise).  sum_parts8 keeps 24 16-byte loads in flight;
// compiled in, it is 216 VGPRs = two waves per SIMD for the whole kernel, and
// the plain step -- six HBM streams, nothing to add up -- ran at 3.7 TB/s
// where it had run at 5.5 (round 6: configs[4]'s own-vocabulary likelihood
// kernel made the step a third of a transition).
template <int NV, bool SOFTMAX, int LANES, bool PARTS = true>
__device__ __forceinline__ void model_step_rows(const ModelStepArgs& a,
                                                const int64_t wave,
                                                const int64_t n_waves) {
  constexpr int kRows = 64 / LANES;  // rows per wave
  static_assert(NV == 1 || LANES == 64, "narrow rows are one chunk per lane");
  const int lane = threadIdx.x & (LANES - 1);     // lane inside the row group
  const int sub = (threadIdx.x & 63) / LANES;     // which row of the wave
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
      if (a.grad_lik) {
        if (a.grad_ready) {
          gl = reinterpret_cast<const m4*>(
              a.grad_ready)[(sub * NV + k) * LANES + lane];
        } else if (!PARTS || a.n_parts <= 1) {
          gl = *reinterpret_cast<const m4*>(a.grad_lik + c * a.grad_stride + d);
        } else if constexpr (PARTS) {
          // (part_stride is a multiple of 4 floats: strides in m4 units)
          gl = sum_parts8(reinterpret_cast<const m4*>(
                              a.grad_lik + c * a.grad_stride + d),
                          a.part_stride / 4, a.n_parts);
          if (a.grad_sum && row_on)
            *reinterpret_cast<m4*>(a.grad_sum + c * a.grad_stride + d) = gl;
        }
        gl = a.lik_scale * gl;
      }
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
    if ((a.lp_out || (a.ll_sum && a.n_parts > 1)) && lane == 0 && row_on) {
      const float llc = a.ll_in ? step_ll(a, c) : 0.f;
      if (a.ll_sum && a.n_parts > 1 && a.ll_in) a.ll_sum[c] = llc;
      if (a.lp_out) a.lp_out[c] = (a.ll_in ? a.lik_scale * llc : 0.f) + prior;
    }
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

}  // namespace zshmc
