// K transitions of a NATIVE model plan from one call -- the launch loop of
// zhusuan_amd/plans/dense.py::_DenseLikelihoodPlan.transition + plans/base.py::_PlanBase.finish on
// THIS side of the C-ABI (the counterpart of zshmc_hmc_diag_normal_run for
// the dense-likelihood, dense-logit Categorical and gathered-dot plans).
//
// One transition (reference zhusuan/hmc.py:418-520) is
//   [mass update from the column sums of the state it starts in]   (:284-305)
//   momentum                                                        (:458)
//   likelihood + gradient at q (or: carried over from the transition before,
//     m.grad_start), then (L+1) x [element-wise step:
//     prior + Jacobian + kick (+ drift + next operand), likelihood] (:348-372)
//   MH accept, select                                               (:479-498)
//   [column sums of the end state] [all-reduce] step-size update    (:501-505)
// -- 2 (L + 1) + 5 .. 9 launches, every one of them an entry point of this
// library.  Issued from a host language each costs a foreign-function call
// (ctypes: ~4 us); at the sizes the reference's own loops run -- the E-steps
// of lntm_mcem.py:157-182 on a minibatch of 100 documents, the 1000
// temperatures of AIS.run (evaluation.py:119-165) -- that is what a
// transition takes.  Here the loop is C: one call per run of transitions.
// Nothing in it reads device memory; the sequence is exactly the one the
// front-end issues, so results are bit-identical to a loop of single runs.
#include "common.h"
#include "model_step.h"

using namespace zshmc;

namespace zshmc {
// csrc/hmc_model_traj.hip: the trips of a transition from one cooperative launch
int trajectory_grid(const zshmc_model_plan& m, int* gx, int* doc_major);
int trajectory_launch(const zshmc_model_plan& m, float lik_scale,
                      bool start_valid, int gx, int doc_major, void* stream);
}  // namespace zshmc

namespace {

__global__ void ais_accumulate_kernel(float* __restrict__ log_w,
                                      const float* __restrict__ orig_lp,
                                      const float* __restrict__ lp,
                                      int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    // evaluation.py:150-163: log w += log p_k(x_{k-1}) [- log p_k(x_k)], the
    // two float32 updates in the order the reference applies them
    float w = log_w[i] + orig_lp[i];
    if (lp) w -= lp[i];
    log_w[i] = w;
  }
}

#define ZS_TRY(call)                 \
  do {                               \
    const int _rc = (call);          \
    if (_rc != ZSHMC_OK) return _rc; \
  } while (0)

// want_ll = false: the gradient alone (the interior evaluations of a
// trajectory; the MFMA kernels then skip the log-likelihood terms)
// (grad_out / ll_out: m.grad / m.ll, or the start buffers)
int likelihood(const zshmc_model_plan& m, const float* q, bool want_ll,
               float* grad_out, float* ll_out, void* s) {
  const float* w = m.operand ? m.operand : q;
  float* ws = m.n_splits > 1 ? m.split_ws : nullptr;
  float* ll = want_ll ? ll_out : nullptr;
  // the bf16x3 kernels (float32-level results on the bf16 matrix cores) where
  // the plan carries the operand's tile image and a gradient is wanted
  if (m.inner_image && grad_out) {
    if (m.kind == ZSHMC_PLAN_LINEAR_BERNOULLI)
      return zshmc_linear_bernoulli_log_lik_bf16x3(
          w, m.inner_image, m.obs, m.n_chains, m.n_inner, m.width, ll, grad_out,
          m.n_splits, ws, s);
    if (m.kind == ZSHMC_PLAN_MIXTURE_MULTINOMIAL && m.obs_sp_rows)
      return zshmc_linear_multinomial_log_lik_bf16x3_sparse(
          w, m.inner_image, m.obs_sp_counts, m.obs_sp_rows, m.obs_sp_off,
          m.obs_rows, m.n_chains, m.n_inner, m.width, ll, grad_out, m.n_splits,
          ws, s);
    if (m.kind == ZSHMC_PLAN_MIXTURE_MULTINOMIAL)
      return zshmc_linear_multinomial_log_lik_bf16x3(
          w, m.inner_image, m.obs, m.obs_rows, m.obs_stride, m.n_chains,
          m.n_inner, m.width, ll, grad_out, m.n_splits, ws, s);
    if (m.kind == ZSHMC_PLAN_LINEAR_CATEGORICAL)
      return zshmc_linear_categorical_log_lik_bf16x3(
          w, m.inner_image, m.obs, m.lik_rows, m.n_inner, m.width, m.n_classes,
          (int)m.groups, ll, grad_out, m.n_splits, ws, s);
  }
  // a small topic model on the exact-fp32 path: row by row over each row's
  // own words (csrc/sparse_multinomial.hip)
  if (m.kind == ZSHMC_PLAN_MIXTURE_MULTINOMIAL && !m.inner_image &&
      m.obs_sp_rows && grad_out)
    return zshmc_sparse_multinomial_log_lik(
        w, m.inner, m.obs_sp_counts, m.obs_sp_rows, m.obs_sp_off, m.obs_rows,
        m.n_chains, m.n_inner, m.width, ll, grad_out, m.n_splits, ws, s);
  switch (m.kind) {
    case ZSHMC_PLAN_LINEAR_BERNOULLI:
      return zshmc_linear_bernoulli_log_lik(w, m.inner, m.obs, m.n_chains,
                                            m.n_inner, m.width, ll, grad_out,
                                            m.n_splits, ws, s);
    case ZSHMC_PLAN_MIXTURE_MULTINOMIAL:
      return zshmc_linear_multinomial_log_lik(
          w, m.inner, m.obs, m.obs_rows, m.obs_stride, m.n_chains, m.n_inner,
          m.width, ll, grad_out, m.n_splits, ws, s);
    case ZSHMC_PLAN_LINEAR_CATEGORICAL:
      return zshmc_linear_categorical_log_lik(
          w, m.inner, m.obs, m.lik_rows, m.n_inner, m.width, m.n_classes,
          (int)m.groups, ll, grad_out, m.n_splits, ws, s);
    case ZSHMC_PLAN_GATHERED_DOT: {
      const bool lat_u = m.gd_latent_is_u != 0;
      if (m.gd_seg_ptr)   // likelihood + gradient in one pass over the pairs
        return zshmc_gather_dot_normal_lik_grad(
            q, m.inner, m.gd_seg_ptr, m.gd_seg_row, m.gd_seg_first,
            m.gd_long_rows, m.gd_n_long, m.gd_idx_other_csr, m.gd_obs_csr,
            m.obs_rows, m.gd_logstd, m.gd_lp_const, m.n_chains, m.gd_n_latent,
            m.n_inner, m.gd_n_pairs, m.gd_n_seg, m.gd_n_dim, grad_out, ll_out,
            m.split_ws, s);
      ZS_TRY(zshmc_gather_dot_normal_lik(
          lat_u ? q : m.inner, lat_u ? m.inner : q,
          lat_u ? m.gd_idx_latent : m.gd_idx_other,
          lat_u ? m.gd_idx_other : m.gd_idx_latent, m.obs, m.obs_rows,
          m.gd_logstd, m.gd_lp_const, m.n_chains, lat_u ? m.gd_n_latent : m.n_inner,
          lat_u ? m.n_inner : m.gd_n_latent, m.gd_n_pairs, m.gd_n_dim,
          m.gd_g_pairs, ll_out, m.split_ws, s));
      if (m.gd_n_pairs)
        return zshmc_gather_dot_grad(m.inner, m.gd_g_pairs, m.gd_seg,
                                     m.gd_order, m.gd_idx_other, m.n_chains,
                                     m.gd_n_latent, m.n_inner, m.gd_n_pairs,
                                     m.gd_n_dim, grad_out, s);
      return zshmc_zero(grad_out, 4 * m.n_chains * m.ld, s);
    }
  }
  set_error("zshmc_hmc_model_run: unknown plan kind %d", m.kind);
  return ZSHMC_ERR_BAD_ARG;
}

// The steps behind a split likelihood launch can add its row-range partials
// themselves (csrc/model_step.h: the arithmetic and order of the reduction
// kernel): one launch less per leapfrog trip, which is what a small problem's
// trip is made of.  The dense kernels' partial layout
// (lb_reduce_splits), the plain step kernel, 16-byte aligned parts.
bool steps_take_parts(const zshmc_model_plan& m) {
  return !m.segmented && m.n_splits > 1 && m.split_ws &&
         (m.kind == ZSHMC_PLAN_LINEAR_BERNOULLI ||
          m.kind == ZSHMC_PLAN_MIXTURE_MULTINOMIAL) &&
         ((int64_t)m.n_splits * m.lik_rows) % 4 == 0 &&
         m.lik_rows == m.n_chains;
}

// the step of a plan that reads the partials a split launch left in
// m.split_ws; grad_sum / ll_sum: where the sums are also stored (or NULL)
int step_parts(const zshmc_model_plan& m, bool with_ll, float* grad_sum,
               float* ll_sum, float kick, float drift, float lik_scale,
               float* lp_out, float* kinetic, void* s) {
  const int64_t C = m.lik_rows, S = m.n_splits;
  ModelStepArgs a{m.q_new, m.p, m.split_ws + S * C, m.width, m.operand,
                  m.width, m.prior_mean, m.mean_rows, m.prior_logstd,
                  m.logstd_rows, m.use_mass ? m.mass : nullptr, m.state, 0.f,
                  kick, drift, lik_scale, m.n_chains, m.n_total, m.ld,
                  with_ll ? m.split_ws : nullptr, lp_out, kinetic, (int)S,
                  C * m.width, grad_sum, with_ll ? ll_sum : nullptr, nullptr};
  return model_kick_drift_launch(a, m.softmax, s);
}

// (grad / ll: the likelihood evaluation the step reads, or NULL: none)
int step(const zshmc_model_plan& m, const float* grad, const float* ll,
         float kick, float drift, float lik_scale, float* lp_out,
         float* kinetic, void* s) {
  const float* mass = m.use_mass ? m.mass : nullptr;
  if (m.segmented)
    return zshmc_model_kick_drift_seg(
        m.q_new, m.p, grad, m.width, m.seg_len, m.groups, m.operand, m.width,
        m.prior_mean, m.mean_rows, m.prior_logstd, m.logstd_rows, mass,
        m.state, 0.f, kick, drift, lik_scale, m.n_chains, m.n_total, m.ld, ll,
        lp_out, kinetic, m.seg_ws, s);
  return zshmc_model_kick_drift(
      m.q_new, m.p, grad, m.width, m.operand, m.width, m.softmax, m.prior_mean,
      m.mean_rows, m.prior_logstd, m.logstd_rows, mass, m.state, 0.f, kick,
      drift, lik_scale, m.n_chains, m.n_total, m.ld, ll, lp_out, kinetic, s);
}

// start_valid: m.grad_start / m.ll_start hold the evaluation at the latents
int transition(const zshmc_model_plan& m, uint32_t t, float lik_scale,
               bool start_valid, void* s) {
  const int L = m.n_leapfrogs;
  const bool carry = m.grad_start && m.ll_start;
  float* g0 = carry ? m.grad_start : m.grad;
  float* l0 = carry ? m.ll_start : m.ll;
  // the latents -> the packed working state
  for (int k = 0; k < m.n_latents; ++k)
    ZS_TRY(zshmc_copy_rows(m.q_new + m.latent_offset[k], m.ld, m.latent[k],
                           m.latent_size[k], nullptr, m.n_chains,
                           m.latent_size[k], s));
  ZS_TRY(zshmc_zero(m.kin_old, 4 * m.n_chains, s));
  for (int k = 0; k < m.n_latents; ++k)
    ZS_TRY(zshmc_momentum_rows(m.p + m.latent_offset[k], m.ld,
                               m.use_mass ? m.latent_mass[k] : nullptr,
                               m.n_chains, m.latent_size[k], m.chain_offset,
                               m.seed, t, (uint32_t)k, m.kin_old, s));
  // small problems: the L + 1 trips from ONE cooperative launch (the same
  // device code in the same order, csrc/hmc_model_traj.hip)
  int traj_gx = 0, traj_dm = 0;
  const bool one_launch = m.traj_sync && trajectory_grid(m, &traj_gx, &traj_dm);
  if (one_launch) {
    ZS_TRY(zshmc_zero(m.kin_new, 4 * m.n_chains, s));
    ZS_TRY(trajectory_launch(m, lik_scale, carry && start_valid, traj_gx,
                             traj_dm, s));
  } else {
  // operand(q), then likelihood + gradient at q -- unless the start buffers
  // hold them already (the previous transition's, selected by its MH test)
  if (!(carry && start_valid)) {
    if (m.operand)
      ZS_TRY(step(m, nullptr, nullptr, 0.f, 0.f, lik_scale, nullptr, nullptr, s));
    ZS_TRY(likelihood(m, m.q_new, true, g0, l0, s));
  } else if (m.softmax && m.operand) {
    // the step's Jacobian reads theta = softmax(q) from the operand buffer,
    // which holds the last PROPOSAL's
    ZS_TRY(step(m, nullptr, nullptr, 0.f, 0.f, lik_scale, nullptr, nullptr, s));
  }
  ZS_TRY(zshmc_zero(m.kin_new, 4 * m.n_chains, s));
  // trip 0: zero-length drift, half kick (hmc.py:352-364); the drift of trip
  // i+1 rides behind the kick of trip i
  ZS_TRY(step(m, g0, l0, 0.5f, L >= 1 ? 1.f : 0.f, lik_scale, m.lp_old,
              L == 0 ? m.kin_new : nullptr, s));
  if (L == 0)
    ZS_TRY(check_hip(hipMemcpyAsync(m.lp_new, m.lp_old, 4 * m.n_chains,
                                    hipMemcpyDeviceToDevice,
                                    reinterpret_cast<hipStream_t>(s)),
                     "hipMemcpyAsync"));
  const bool parts = steps_take_parts(m);
  for (int i = 1; i <= L; ++i) {
    const bool last = i == L;
    if (parts) {
      // the evaluation stays in its row-range partials; the last one of the
      // trajectory is also stored (what an accepting chain carries along)
      {
        KeepSplitParts keep;
        ZS_TRY(likelihood(m, m.q_new, last, m.grad, m.ll, s));
      }
      ZS_TRY(step_parts(m, last, last ? m.grad : nullptr, m.ll,
                        last ? 0.5f : 1.f, last ? 0.f : 1.f, lik_scale,
                        last ? m.lp_new : nullptr, last ? m.kin_new : nullptr,
                        s));
      continue;
    }
    ZS_TRY(likelihood(m, m.q_new, last, m.grad, m.ll, s));
    ZS_TRY(step(m, m.grad, m.ll, last ? 0.5f : 1.f, last ? 0.f : 1.f, lik_scale,
                last ? m.lp_new : nullptr, last ? m.kin_new : nullptr, s));
  }
  }  // launch per trip
  ZS_TRY(zshmc_mh_accept(m.lp_old, m.lp_new, m.kin_old, m.kin_new, m.n_chains,
                         m.chain_offset, m.seed, t, m.acceptance_rate,
                         m.orig_hamiltonian, m.hamiltonian, m.log_prob,
                         m.accept, m.acc_sum, m.flags, s));
  // where(accept, q', q) for every latent (hmc.py:488-497)
  for (int k = 0; k < m.n_latents; ++k)
    ZS_TRY(zshmc_copy_rows(m.latent[k], m.latent_size[k],
                           m.q_new + m.latent_offset[k], m.ld, m.accept,
                           m.n_chains, m.latent_size[k], s));
  // ... and the evaluation that goes with the state: the last one of the
  // trajectory where the chain accepted (a chain's rows of the likelihood
  // matrices are contiguous: lik_rows / n_chains rows of `width`)
  if (carry && L >= 1) {
    const int64_t per_chain = m.lik_rows / m.n_chains;
    ZS_TRY(zshmc_copy_rows(m.grad_start, per_chain * m.width, m.grad,
                           per_chain * m.width, m.accept, m.n_chains,
                           per_chain * m.width, s));
    ZS_TRY(zshmc_copy_rows(m.ll_start, per_chain, m.ll, per_chain, m.accept,
                           m.n_chains, per_chain, s));
  }
  return ZSHMC_OK;
}

int colstats(const zshmc_model_plan& m, void* s) {
  ZS_TRY(zshmc_zero(m.comm_buf + ZSHMC_STATS_WORDS,
                    8 * (m.comm_words - ZSHMC_STATS_WORDS), s));
  for (int k = 0; k < m.n_latents; ++k)
    ZS_TRY(zshmc_mass_colstats(m.latent[k], m.ewmv_mean[k], m.n_chains,
                               m.latent_size[k], m.colsum[k], s));
  return ZSHMC_OK;
}

int mass_update(const zshmc_model_plan& m, void* s) {
  if (m.n_latents == 1)
    return zshmc_mass_update_fused(m.state, m.ewmv_mean[0], m.ewmv_var[0],
                                   m.colsum[0], 1, m.n_chains_global,
                                   m.latent_size[0], m.mass_decay, 0,
                                   m.latent_mass[0], m.mass_ws, s);
  for (int k = 0; k < m.n_latents; ++k)
    // EWMV.t is shared by the latents (hmc.py:118,131): bumped once, after
    // the last one
    ZS_TRY(zshmc_mass_update(m.state, m.ewmv_mean[k], m.ewmv_var[k],
                             m.colsum[k], m.n_chains_global, m.latent_size[k],
                             m.mass_decay, k == m.n_latents - 1 ? 1 : 2, 0,
                             m.latent_mass[k], s));
  return ZSHMC_OK;
}

}  // namespace

extern "C" int zshmc_hmc_model_run(const zshmc_model_plan* plan,
                                   uint32_t iteration_first, int n_transitions,
                                   int update_kind, int adapt_mass,
                                   const float* lik_scale_host,
                                   float* ais_log_weights,
                                   int ais_ends_here, void* comm,
                                   void* stream) {
  ZS_REQUIRE(plan && n_transitions >= 0, "zshmc_hmc_model_run: bad argument");
  const zshmc_model_plan& m = *plan;
  ZS_REQUIRE(m.n_latents >= 1 && m.n_latents <= ZSHMC_MAX_LATENTS &&
                 m.n_chains > 0 && m.n_leapfrogs >= 0,
             "zshmc_hmc_model_run: bad plan");
  ZS_REQUIRE(update_kind == ZSHMC_PEND_NONE || update_kind == ZSHMC_PEND_ADAPT ||
                 update_kind == ZSHMC_PEND_HOLD,
             "zshmc_hmc_model_run: bad update_kind");
  ZS_REQUIRE(!m.grad_start == !m.ll_start && m.lik_rows % m.n_chains == 0,
             "zshmc_hmc_model_run: grad_start and ll_start go together; "
             "lik_rows is a multiple of n_chains");
  ZS_REQUIRE(!adapt_mass || (m.comm_buf && m.use_mass),
             "zshmc_hmc_model_run: mass adaptation needs the column-sum "
             "buffer and the mass vectors");
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
  for (int i = 0; i < n_transitions; ++i) {
    const uint32_t t = iteration_first + (uint32_t)i;
    // (the column sums of the state this transition starts in were taken at
    // the end of the previous one -- by the caller before the first)
    if (adapt_mass) ZS_TRY(mass_update(m, stream));
    const float ls = lik_scale_host ? lik_scale_host[i] : 1.0f;
    ZS_TRY(transition(m, t, ls, i > 0 || m.start_valid != 0, stream));
    if (adapt_mass) ZS_TRY(colstats(m, stream));
    if (comm) {
      if (adapt_mass)
        ZS_TRY(zshmc_comm_all_reduce_sum(comm, m.comm_buf, m.comm_words,
                                         stream));
      else if (update_kind != ZSHMC_PEND_NONE)
        ZS_TRY(zshmc_comm_all_reduce_sum(comm, m.comm_buf, ZSHMC_STATS_WORDS,
                                         stream));
    }
    if (update_kind != ZSHMC_PEND_NONE)
      ZS_TRY(zshmc_stepsize_update(m.state, m.acc_sum, m.n_chains_global,
                                   update_kind == ZSHMC_PEND_ADAPT, 0, m.delta,
                                   m.gamma, m.t0, m.kappa, m.mu, stream));
    if (ais_log_weights) {
      const bool final = ais_ends_here && i == n_transitions - 1;
      hipLaunchKernelGGL(ais_accumulate_kernel,
                         dim3((unsigned)((m.n_chains + 255) / 256)), dim3(256),
                         0, hs, ais_log_weights, m.lp_old,
                         final ? nullptr : m.log_prob, m.n_chains);
      ZS_LAUNCH_CHECK("ais_accumulate_kernel launch");
    }
  }
  return ZSHMC_OK;
}

extern "C" int zshmc_hmc_model_transition(const zshmc_model_plan* plan,
                                          uint32_t iteration, float lik_scale,
                                          void* stream) {
  ZS_REQUIRE(plan, "zshmc_hmc_model_transition: null plan");
  const zshmc_model_plan& m = *plan;
  ZS_REQUIRE(m.n_latents >= 1 && m.n_latents <= ZSHMC_MAX_LATENTS &&
                 m.n_chains > 0 && m.n_leapfrogs >= 0 &&
                 !m.grad_start == !m.ll_start && m.lik_rows % m.n_chains == 0,
             "zshmc_hmc_model_transition: bad plan");
  return transition(m, iteration, lik_scale, m.start_valid != 0, stream);
}

extern "C" int zshmc_trajectory_capacity(int64_t width, int kind,
                                         int* n_workgroups) {
  ZS_REQUIRE(n_workgroups, "zshmc_trajectory_capacity: null pointer");
  zshmc_model_plan m = {};
  m.kind = kind;
  m.width = width;
  m.n_chains = 1;
  m.obs_rows = 1;
  m.n_leapfrogs = 1;
  m.n_splits = 1;
  m.one_launch = 1;
  int gx = 0, dm = 0;
  *n_workgroups = 0;
  // (the largest grid trajectory_grid accepts: probe by doubling)
  int lo = 0;
  for (int64_t c = 1; c <= 4096; c *= 2) {
    m.n_chains = c * 64;
    if (!trajectory_grid(m, &gx, &dm)) break;
    lo = (int)c;
  }
  if (lo) {
    int hi = lo * 2;   // first power of two that failed (or 8192)
    while (hi - lo > 1) {
      const int mid = (lo + hi) / 2;
      m.n_chains = (int64_t)mid * 64;
      if (trajectory_grid(m, &gx, &dm)) lo = mid; else hi = mid;
    }
  }
  *n_workgroups = lo;
  return ZSHMC_OK;
}
