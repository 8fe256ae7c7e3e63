// Argument block shared by the fused diag-Normal HMC kernels
// (hmc_fused_normal.hip: register-prefetch kernel for small / ragged rows;
//  hmc_fused_ring.hip: LDS-DMA ring kernel for 16-B aligned rows > 512 B)
// and the dual-averaging step-size update they carry.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/zshmc.h"

namespace zshmc {

constexpr float kHalfLog2PiNeg = -0.91893853320467274178f;  // -0.5*log(2*pi)

// ---- dual averaging (StepsizeTuner.tune, hmc.py:89-112) --------------------
struct TunerCfg {
  float delta, gamma, t0, kappa, mu;  // mu = 10 * initial step size (sic, :79)
};
struct TunerState {
  float step_size, step, log_eps_bar, h_bar;
};

__device__ __forceinline__ TunerState tuner_load(const float* state) {
  return TunerState{state[ZSHMC_ST_STEP_SIZE], state[ZSHMC_ST_TUNER_STEP],
                    state[ZSHMC_ST_LOG_EPS_BAR], state[ZSHMC_ST_H_BAR]};
}

// One update from the mean acceptance `acc` of a finished transition.
// kind: ZSHMC_PEND_ADAPT (the run's adapt_step_size flag was true, :92-106) or
// ZSHMC_PEND_HOLD (false, :108-110: epsilon <- exp(log_epsilon_bar)).
// The update runs in three places -- the prologue of every workgroup of the
// NEXT transition kernel, the workgroup that retires that kernel last, and the
// stand-alone flush kernel -- which must agree to the bit: contraction is off
// so that hipcc cannot fuse differently in different inlining contexts.
__device__ __forceinline__ TunerState tuner_apply(TunerState s, float acc,
                                                  int kind, float fresh,
                                                  const TunerCfg& c) {
#pragma clang fp contract(off)
  if (kind == ZSHMC_PEND_ADAPT) {
    const float keep = 1.0f - fresh;
    const float step = keep * s.step + 1.0f;
    const float rate1 = 1.0f / (step + c.t0);
    const float h_bar = keep * (1.0f - rate1) * s.h_bar + rate1 * (c.delta - acc);
    const float log_eps = c.mu - sqrtf(step) / c.gamma * h_bar;
    const float rate = powf(step, -c.kappa);
    s.log_eps_bar = rate * log_eps + keep * (1.0f - rate) * s.log_eps_bar;
    s.step = step;
    s.h_bar = h_bar;
    s.step_size = expf(log_eps);
  } else {
    s.step_size = expf(s.log_eps_bar);
  }
  return s;
}

// The link between consecutive transitions (include/zshmc.h,
// zshmc_adapt_link): where the acceptance sum goes and which dual-averaging
// update rides on this launch.
struct AdaptLink {
  float* state;      // ZSHMC_ST_* block; NULL: no on-device step size / tuner
  double* stats;     // [0] sum acc (in: previous, all-reduced, if `pending`;
                     // out: this launch's, order-fixed), [1] non-finite flag
  double* partials;  // workspace: per-workgroup acceptance sums
  uint32_t* done;    // workspace: retired-workgroup counter (0 between launches)
  double inv_chains; // 1 / n_chains_global
  int pending;       // ZSHMC_PEND_*: update of the PREVIOUS transition, applied
                     // in this launch's prologue from stats[0] (sharded chains:
                     // an all-reduce sat in between)
  int retire;        // ZSHMC_PEND_*: update of THIS transition, applied by the
                     // workgroup that retires last from its own total (all
                     // chains on this GPU)
  float fresh;       // 1: fresh start (hmc.py:466-467, :92-102)
  float used_step_size;  // epsilon the updated-for transition used if it came
                         // from the search (NaN: it used state[STEP_SIZE])
  TunerCfg tuner;
};

// state <- update(state, acc_sum) and the two diagnostic words; one thread.
__device__ __forceinline__ void tuner_persist(const AdaptLink& k, int kind,
                                              double acc_sum) {
  const TunerState s0 = tuner_load(k.state);
  const float acc = (float)(acc_sum * k.inv_chains);  // hmc.py:377
  const TunerState s = tuner_apply(s0, acc, kind, k.fresh, k.tuner);
  k.state[ZSHMC_ST_MEAN_ACCEPT] = acc;
  k.state[ZSHMC_ST_USED_STEP_SIZE] =
      k.used_step_size == k.used_step_size ? k.used_step_size : s0.step_size;
  k.state[ZSHMC_ST_STEP_SIZE] = s.step_size;
  k.state[ZSHMC_ST_TUNER_STEP] = s.step;
  k.state[ZSHMC_ST_LOG_EPS_BAR] = s.log_eps_bar;
  k.state[ZSHMC_ST_H_BAR] = s.h_bar;
}

// The step size this launch integrates with: the host value, or the device
// state with the pending update applied (every workgroup computes the same
// scalars from the same inputs; nobody writes them until all have read).
__device__ __forceinline__ float link_step_size(const AdaptLink& k,
                                                float step_size_host) {
  if (!k.state) return step_size_host;
  TunerState s = tuner_load(k.state);
  if (k.pending != ZSHMC_PEND_NONE)
    s = tuner_apply(s, (float)(k.stats[0] * k.inv_chains), k.pending, k.fresh,
                    k.tuner);
  return s.step_size;
}

// End of a transition kernel.  link_publish (thread 0 of every workgroup):
// hand this workgroup's acceptance sum over; true for the workgroup that
// retires last.  link_finish (ONE FULL WAVE of that workgroup): add the
// partials in a fixed order -- lane l takes partials l, l+64, ..., then the
// xor butterfly: run-to-run identical, unlike floating-point atomics --
// persist the dual-averaging update that rides on this launch and publish the
// total.  All hand-over goes through agent-scope atomics (performed at the
// device-coherent level), so no cache write-back / invalidate is needed (a
// release fence here flushed the L2 under the workgroups still running).  The
// exchange RETURNS, so the partial has landed before the counter moves.  The
// partial loads bypass the L2 (~100 ns each): spread over the lanes they cost
// ~0.5 us, read one after another by one thread they cost 25 us.
__device__ __forceinline__ bool link_publish(const AdaptLink& k,
                                             double wg_sum) {
  if (!k.partials) return false;
  const double prev = __hip_atomic_exchange(
      &k.partials[blockIdx.x], wg_sum, __ATOMIC_RELAXED,
      __HIP_MEMORY_SCOPE_AGENT);
  // the counter's operand is made to depend on the returned value (an opaque
  // asm that consumes it), so hipcc must wait for the exchange to come back
  unsigned one = 1u;
  asm volatile("" : "+v"(one) : "v"(prev));
  const unsigned ticket = __hip_atomic_fetch_add(
      k.done, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return ticket == gridDim.x - 1;
}

__device__ __forceinline__ void link_finish(const AdaptLink& k,
                                            const uint32_t* flags, int lane) {
  const unsigned nblk = gridDim.x;
  double part = 0.0;
  for (unsigned i = lane; i < nblk; i += 64)
    part += __hip_atomic_load(&k.partials[i], __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  if (lane != 0) return;
  const double total = part;
  if (k.state && k.pending != ZSHMC_PEND_NONE)
    tuner_persist(k, k.pending, k.stats[0]);
  if (k.state && k.retire != ZSHMC_PEND_NONE) tuner_persist(k, k.retire, total);
  k.stats[0] = total;
  uint32_t f = 0;
  if (flags)
    f = __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  k.stats[1] = (f & ZSHMC_FLAG_OLD_LOGPROB_NONFINITE) ? 1.0 : 0.0;
  __hip_atomic_store(k.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct FusedArgs {
  float* q;
  const float* mean;  // NULL: all zeros (no subtraction / re-addition)
  const float* logstd;
  const float* mass;
  float step_size_host;  // used when link.state == NULL
  int64_t n_chains;
  int64_t n_data;
  int64_t chain_offset;
  int n_leapfrogs;
  uint32_t k0, k1;
  uint32_t iteration;
  int commit;
  float* acceptance_rate;
  float* orig_hamiltonian;
  float* hamiltonian;
  float* orig_log_prob;
  float* log_prob;
  uint32_t* flags;
  AdaptLink link;
  // ring kernel only (set by its launcher): HMCInfo scalars staged in LDS
  // (info_cap chains per workgroup) or stored straight from the trip loop
  int info_cap;          // 0 = no staging
  uint32_t commit_direct;  // commit && !staging: enables the in-loop stores
#ifdef ZS_TIMING
  unsigned long long* timing;  // [n_waves][4]: start, end, xcc, chains
#endif
};

typedef float f4 __attribute__((ext_vector_type(4)));

// hmc_fused_ring.hip.  Returns ZSHMC_ERR_UNSUPPORTED (without setting the
// error string) when the shape is not one it covers.
int launch_fused_ring(const FusedArgs& a, hipStream_t stream);
// "NCH,K,mass" of the ring instantiation for this shape; false if the shape
// is not covered (alignment aside).
bool fused_ring_config(int64_t n_data, bool has_mass, bool zero_mean, int* nch,
                       int* k);
bool fused_ring_enabled();  // ZSHMC_FUSED_RING != 0
// largest grid either fused kernel launches (sizes the link workspace)
constexpr int kFusedMaxGrid = 4096;

}  // namespace zshmc
