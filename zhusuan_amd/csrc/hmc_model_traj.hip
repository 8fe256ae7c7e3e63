// One launch per TRAJECTORY for native model plans whose likelihood fits one
// resident wave of workgroups -- the sizes the reference's own loops run at:
// the E-steps of examples/topic_models/lntm_mcem.py:62-70,157-182 (one chain,
// 100 documents per minibatch, K = 100, L = 20), the 1 000 temperatures of
// AIS.run (zhusuan/evaluation.py:119-165).  There one likelihood evaluation is
// ~5 us of matrix-core work and a transition of the launch-per-trip path
// (csrc/hmc_model_run.hip) is ~2 (L + 1) + 6 dependent launches of ~8 us each:
// the device-side launch latency is what a transition costs.
//
// Here the L + 1 trips of HMC._leapfrog (zhusuan/hmc.py:348-372) run inside
// ONE cooperative launch: the workgroups of the likelihood kernel's own grid
// (chain blocks x row-range slices) stay resident and alternate between
//   LIK   the two-GEMM likelihood + gradient of their slice (csrc/lb_body.h --
//         the SAME device code as linear_bernoulli_kernel), partial sums to the
//         split workspace, and
//   STEP  the element-wise leapfrog step (csrc/model_step.h -- the SAME code as
//         model_kick_drift_kernel) on the grid's waves, which adds the partials
//         in lb_reduce_splits_kernel's order while it reads them,
// with a grid-wide barrier (one agent-scope atomic counter, release / acquire)
// where the launch-per-trip path has a kernel boundary.  Same arithmetic in
// the same order: results are bit-identical to the launch-per-trip path
// (tests/test_gpu_model_run.py).
// (the persistent kernel holds a tile loop AND a step: eight partial loads in
// flight is what its register budgets take without scratch)
#define ZS_PARTS_BATCHES 1
#include "common.h"
#include "lb_body.h"
#include "model_step.h"

namespace zshmc {

struct TrajArgs {
  // likelihood (zshmc_linear_bernoulli_log_lik / _multinomial_ arguments)
  const float* W;      // operand [C, width] (or q_new itself)
  const float* X;      // [N, width]
  const float* y;      // labels (OP 0)
  const float* yc;     // counts (OP 1)
  int64_t yc_rows, ldy, C, N;
  int doc_major, gx, S;
  float* ws;           // split workspace (S > 1) [S * C * (width + 1)]
  float* grad;         // [C, width] the trajectory's evaluation
  float* ll;           // [C]
  float* grad0;        // the start evaluation's buffers (grad_start or grad)
  float* ll0;
  int first_eval;      // 1: evaluate at the start point; 0: grad0 / ll0 hold it
  int first_operand;   // 1: operand(q) has to be formed first
  int L;
  float lik_scale;
  float* lp_old;
  float* lp_new;
  float* kin_new;
  ModelStepArgs step;  // q, p, operand, prior, mass, step size: per-trip fields
                       // (grad_lik, ll_in, scales, outputs) are set in the kernel
  unsigned* barrier;   // [2]: arrivals, generation (zeroed once at creation)
  unsigned* fault;     // set when a barrier wait ran out of patience
};

// Grid-wide barrier between two phases.  One thread per workgroup arrives with
// an agent-scope RELEASE (the workgroup's writes -- partial sums, q, p, the
// operand -- are written back past the XCD's L2, which is not coherent with
// the other XCDs') and leaves with an ACQUIRE (its L2 / L1 lines are
// invalidated before the next phase's plain loads); __syncthreads on both
// sides extends that to the workgroup.  The spin is bounded: a grid that is
// not co-resident (it is launched cooperatively, so it should be) sets
// *fault and falls through instead of hanging the GPU.
__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned n_wg,
                                             unsigned* fault) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned gen =
        __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned arrived = __hip_atomic_fetch_add(
        &bar[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived == n_wg - 1) {
      __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&bar[1], gen + 1, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_AGENT);
    } else {
      long long spins = 0;
      while (__hip_atomic_load(&bar[1], __ATOMIC_ACQUIRE,
                               __HIP_MEMORY_SCOPE_AGENT) == gen) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1ll << 24)) {   // ~seconds: not co-resident
          __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  __syncthreads();
}

// the step kernel's instantiation for rows of `width` floats
// (zshmc_model_kick_drift's dispatch)
template <int D, bool SOFTMAX>
__device__ __forceinline__ void traj_step(const ModelStepArgs& a, int64_t wave,
                                          int64_t n_waves) {
  if constexpr (D <= 64)
    model_step_rows<1, SOFTMAX, 16>(a, wave, n_waves);
  else if constexpr (D <= 128)
    model_step_rows<1, SOFTMAX, 32>(a, wave, n_waves);
  else
    model_step_rows<1, SOFTMAX, 64>(a, wave, n_waves);
}

// (Register budget: one wave per SIMD from 128 columns on, two below -- the
// tile loop plus both call forms plus the step spill a few registers at the
// likelihood kernel's own, tighter budgets, and a grid that has to be
// resident at once has no use for more waves per SIMD.)
template <int D, int OP>
__global__ __launch_bounds__(256, D >= 128 ? 1 : 2) void model_trajectory_kernel(
    TrajArgs t) {
  constexpr bool SOFTMAX = OP == 1;
  const int n_wg = t.gx * t.S;
  const int bx = (int)blockIdx.x % t.gx, by = (int)blockIdx.x / t.gx;
  const int64_t wave = (int64_t)blockIdx.x * 4 + threadIdx.x / 64;
  const int64_t n_waves = (int64_t)n_wg * 4;
  const int64_t ld_w = D;
  float* ll_part = t.S > 1 ? t.ws : nullptr;
  float* g_part = t.S > 1 ? t.ws + (int64_t)t.S * t.C : nullptr;
  // likelihood + gradient (want_ll) of this workgroup's slice
  auto lik = [&](bool want_ll, float* g_out, float* ll_out) {
    float* gq = t.S > 1 ? g_part : g_out;
    float* lq = t.S > 1 ? ll_part : ll_out;
    if (want_ll)
      lb_body<D, true, OP, true>(t.W, t.X, t.y, t.yc, t.yc_rows, t.ldy, t.C,
                                 t.N, ld_w, ld_w, lq, gq, t.doc_major, 0, 0, bx,
                                 by, t.S);
    else
      lb_body<D, true, OP, false>(t.W, t.X, t.y, t.yc, t.yc_rows, t.ldy, t.C,
                                  t.N, ld_w, ld_w, lq, gq, t.doc_major, 0, 0,
                                  bx, by, t.S);
  };
  // one element-wise step reading the evaluation in (g, l) -- or its partials
  auto step = [&](const float* g, const float* l, bool parts, float* g_sum,
                  float* l_sum, float kick, float drift, float* lp_out,
                  float* kinetic) {
    ModelStepArgs a = t.step;
    a.kick_scale = kick;
    a.drift_scale = drift;
    a.lik_scale = t.lik_scale;
    a.lp_out = lp_out;
    a.kinetic = kinetic;
    if (parts && t.S > 1) {
      a.grad_lik = g_part;
      a.ll_in = l ? ll_part : nullptr;
      a.n_parts = t.S;
      a.part_stride = t.C * ld_w;
      a.grad_sum = g_sum;
      a.ll_sum = l_sum;
    } else {
      a.grad_lik = g;
      a.ll_in = l;
      a.n_parts = 1;
      a.part_stride = 0;
      a.grad_sum = nullptr;
      a.ll_sum = nullptr;
    }
    traj_step<D, SOFTMAX>(a, wave, n_waves);
  };
  auto barrier = [&]() { grid_barrier(t.barrier, (unsigned)n_wg, t.fault); };

  // the sequence of csrc/hmc_model_run.hip::transition between the momentum
  // and the MH test, a barrier where that has a kernel boundary
  const int L = t.L;
  bool start_parts = false;
  if (t.first_eval) {
    if (t.first_operand) {
      step(nullptr, nullptr, false, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr);
      barrier();
    }
    lik(true, t.grad0, t.ll0);
    barrier();
    start_parts = true;
  } else if (t.first_operand) {
    step(nullptr, nullptr, false, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr);
    barrier();
  }
  // trip 0: zero-length drift, half kick (hmc.py:352-364)
  step(t.grad0, t.ll0, start_parts, t.grad0, t.ll0, 0.5f, L >= 1 ? 1.f : 0.f,
       t.lp_old, L == 0 ? t.kin_new : nullptr);
  for (int i = 1; i <= L; ++i) {
    const bool last = i == L;
    barrier();
    lik(last, t.grad, t.ll);
    barrier();
    step(t.grad, last ? t.ll : nullptr, true, t.grad, last ? t.ll : nullptr,
         last ? 0.5f : 1.f, last ? 0.f : 1.f, last ? t.lp_new : nullptr,
         last ? t.kin_new : nullptr);
  }
}

// workgroups per CU that can be resident (cached per instantiation)
template <int D, int OP>
static int traj_capacity() {
  static int cap = -1;
  if (cap < 0) {
    int per_cu = 0;
    const size_t lds = lb_lds_bytes(D);
    if (hipFuncSetAttribute(
            reinterpret_cast<const void*>(model_trajectory_kernel<D, OP>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &per_cu, model_trajectory_kernel<D, OP>, 256, lds) != hipSuccess)
      per_cu = 0;
    (void)hipGetLastError();
    cap = per_cu * device_cu_count();
  }
  return cap;
}

template <int D, int OP>
static int launch_traj(TrajArgs& t, hipStream_t s) {
  void* params[] = {&t};
  return check_hip(
      hipLaunchCooperativeKernel(
          reinterpret_cast<const void*>(model_trajectory_kernel<D, OP>),
          dim3((unsigned)(t.gx * t.S)), dim3(256), params,
          (unsigned)lb_lds_bytes(D), s),
      "hipLaunchCooperativeKernel(model_trajectory_kernel)");
}

// Can the trips of this plan run from one launch?  (Bernoulli / mixture
// multinomial, <= 256 columns, the exact-fp32 kernels, and the likelihood's
// grid fits the device at once with room to spare.)
int trajectory_grid(const zshmc_model_plan& m, int* gx_out, int* doc_major_out) {
  if (m.segmented || m.width > 256 || m.inner_image || !m.one_launch ||
      (m.kind != ZSHMC_PLAN_LINEAR_BERNOULLI &&
       m.kind != ZSHMC_PLAN_MIXTURE_MULTINOMIAL) ||
      m.n_leapfrogs < 1 || m.n_splits < 1)
    return 0;
  const int64_t C = m.n_chains;
  int doc_major = 0;
  int64_t gx = (C + kMC - 1) / kMC;
  if (m.kind == ZSHMC_PLAN_MIXTURE_MULTINOMIAL) {
    // zshmc_linear_multinomial_log_lik's rule
    const int64_t n_chains = C / m.obs_rows;
    doc_major = m.obs_rows > 1 && (n_chains % kMC == 0 || n_chains >= 512);
    if (doc_major) gx = ((n_chains + kMC - 1) / kMC) * m.obs_rows;
  }
  const int S = (m.n_splits > 1 && m.split_ws) ? m.n_splits : 1;
  int cap = 0;
  const bool mult = m.kind == ZSHMC_PLAN_MIXTURE_MULTINOMIAL;
  switch (m.width) {
    case 64: cap = mult ? traj_capacity<64, 1>() : traj_capacity<64, 0>(); break;
    case 128: cap = mult ? traj_capacity<128, 1>() : traj_capacity<128, 0>(); break;
    case 192: cap = mult ? traj_capacity<192, 1>() : traj_capacity<192, 0>(); break;
    case 256: cap = mult ? traj_capacity<256, 1>() : traj_capacity<256, 0>(); break;
    default: return 0;
  }
  if (gx * S > cap || gx * S > 4096) return 0;
  *gx_out = (int)gx;
  *doc_major_out = doc_major;
  return 1;
}

// The L + 1 trips of one transition (csrc/hmc_model_run.hip::transition from
// the first evaluation to the last step) from one cooperative launch.
int trajectory_launch(const zshmc_model_plan& m, float lik_scale,
                      bool start_valid, int gx, int doc_major, void* stream) {
  const bool carry = m.grad_start && m.ll_start;
  const bool mult = m.kind == ZSHMC_PLAN_MIXTURE_MULTINOMIAL;
  TrajArgs t;
  t.W = m.operand ? m.operand : m.q_new;
  t.X = m.inner;
  t.y = mult ? nullptr : m.obs;
  t.yc = mult ? m.obs : nullptr;
  t.yc_rows = mult ? m.obs_rows : 1;
  t.ldy = mult ? m.obs_stride : m.n_inner;
  t.C = m.n_chains;
  t.N = m.n_inner;
  t.doc_major = doc_major;
  t.gx = gx;
  t.S = (m.n_splits > 1 && m.split_ws) ? m.n_splits : 1;
  t.ws = m.split_ws;
  t.grad = m.grad;
  t.ll = m.ll;
  t.grad0 = carry ? m.grad_start : m.grad;
  t.ll0 = carry ? m.ll_start : m.ll;
  t.first_eval = !(carry && start_valid);
  t.first_operand = m.operand && (t.first_eval || m.softmax);
  t.L = m.n_leapfrogs;
  t.lik_scale = lik_scale;
  t.lp_old = m.lp_old;
  t.lp_new = m.lp_new;
  t.kin_new = m.kin_new;
  t.step = ModelStepArgs{
      m.q_new, m.p, nullptr, m.width, m.operand, m.width, m.prior_mean,
      m.mean_rows, m.prior_logstd, m.logstd_rows, m.use_mass ? m.mass : nullptr,
      m.state, 0.f, 0.f, 0.f, lik_scale, m.n_chains, m.n_total, m.ld, nullptr,
      nullptr, nullptr, 1, 0, nullptr, nullptr};
  t.barrier = reinterpret_cast<unsigned*>(m.traj_sync);
  t.fault = t.barrier + 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (m.width) {
    case 64: return mult ? launch_traj<64, 1>(t, s) : launch_traj<64, 0>(t, s);
    case 128: return mult ? launch_traj<128, 1>(t, s) : launch_traj<128, 0>(t, s);
    case 192: return mult ? launch_traj<192, 1>(t, s) : launch_traj<192, 0>(t, s);
    default: return mult ? launch_traj<256, 1>(t, s) : launch_traj<256, 0>(t, s);
  }
}

}  // namespace zshmc
