// Argument block shared by the fused diag-Normal HMC kernels
// (hmc_fused_normal.hip: register-prefetch kernel for small / ragged rows;
//  hmc_fused_ring.hip: LDS-DMA ring kernel for 16-B aligned rows > 512 B)
// and the dual-averaging step-size update they carry.
#pragma once
#ifdef ZS_HOST_ONLY
// tests/host_link: the link / dual-averaging logic below compiled for the
// HOST (one thread standing for one workgroup), so that the CPU tests of the
// front-end's orchestration run the product's update code, not a restatement
#include <math.h>
#define __device__
#define __forceinline__ inline
#define __HIP_MEMORY_SCOPE_AGENT 0
typedef void* hipStream_t;
namespace zshmc {
struct HostDim {
  unsigned x;
};
static HostDim gridDim = {1};
template <class T>
inline T __hip_atomic_fetch_add(T* p, T v, int, int) {
  const T old = *p;
  *p = old + v;
  return old;
}
template <class T>
inline T __hip_atomic_load(const T* p, int, int) {
  return *p;
}
template <class T>
inline void __hip_atomic_store(T* p, T v, int, int) {
  *p = v;
}
}  // namespace zshmc
#else
#include <hip/hip_runtime.h>
#endif
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
// Split in two so that everything that does not depend on the acceptance --
// the reciprocal, the square root and the power of the step counter, or the
// whole HOLD update -- can be evaluated while the transition runs and only
// tuner_finish (a handful of flops and one expf) sits behind the last
// workgroup's retirement.
struct TunerPrep {
  float keep, step, rate1, sqrt_over_gamma, rate, hold_step_size;
};

__device__ __forceinline__ TunerPrep tuner_prepare(const TunerState& s,
                                                   int kind, float fresh,
                                                   const TunerCfg& c) {
#pragma clang fp contract(off)
  TunerPrep p{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (kind == ZSHMC_PEND_ADAPT) {
    p.keep = 1.0f - fresh;
    p.step = p.keep * s.step + 1.0f;
    p.rate1 = 1.0f / (p.step + c.t0);
    p.sqrt_over_gamma = sqrtf(p.step) / c.gamma;
    p.rate = powf(p.step, -c.kappa);
  } else {
    p.hold_step_size = expf(s.log_eps_bar);
  }
  return p;
}

__device__ __forceinline__ TunerState tuner_finish(TunerState s,
                                                   const TunerPrep& p,
                                                   float acc, int kind,
                                                   const TunerCfg& c) {
#pragma clang fp contract(off)
  if (kind == ZSHMC_PEND_ADAPT) {
    const float h_bar =
        p.keep * (1.0f - p.rate1) * s.h_bar + p.rate1 * (c.delta - acc);
    const float log_eps = c.mu - p.sqrt_over_gamma * h_bar;
    s.log_eps_bar = p.rate * log_eps + p.keep * (1.0f - p.rate) * s.log_eps_bar;
    s.step = p.step;
    s.h_bar = h_bar;
    s.step_size = expf(log_eps);
  } else {
    s.step_size = p.hold_step_size;
  }
  return s;
}

__device__ __forceinline__ TunerState tuner_apply(TunerState s, float acc,
                                                  int kind, float fresh,
                                                  const TunerCfg& c) {
  return tuner_finish(s, tuner_prepare(s, kind, fresh, c), acc, kind, c);
}

// The link between consecutive transitions (include/zshmc.h,
// zshmc_adapt_link): where the acceptance sum goes and which dual-averaging
// update rides on this launch.
struct AdaptLink {
  float* state;      // ZSHMC_ST_* block; NULL: no on-device step size / tuner
  double* stats;     // [0] sum acc (in: previous, all-reduced, if `pending`;
                     // out: this launch's, order-fixed), [1] non-finite flag
  unsigned long long* accum;  // workspace: {retired count | fixed-point sum},
                              // 0 between launches
  double fx_scale, fx_inv_scale;  // 2^shift, 2^-shift of the fixed point
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
  // column statistics of the END state for the next mass update
  // (zshmc_adapt_link.colstats_*): EWMV mean [n_data]; one row of 2*n_data
  // doubles per workgroup.  NULL: not asked for.
  const float* cs_mean;
  double* cs_parts;
  // launches replayed from a hipGraph (zshmc_hmc_diag_normal_run): the
  // iteration of the Philox counters is args.iteration + *iter_dev, and the
  // workgroup that retires last advances *iter_dev.  NULL: args.iteration.
  uint32_t* iter_dev;
};

// state <- update(state, acc_sum) and the two diagnostic words; one thread.
__device__ __forceinline__ void tuner_persist(const AdaptLink& k, int kind,
                                              double acc_sum,
                                              const TunerPrep* prep = nullptr) {
  const TunerState s0 = tuner_load(k.state);
  const float acc = (float)(acc_sum * k.inv_chains);  // hmc.py:377
  const TunerState s =
      prep ? tuner_finish(s0, *prep, acc, kind, k.tuner)
           : tuner_apply(s0, acc, kind, k.fresh, k.tuner);
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
// link_prepare: thread 0's head start on the update this launch will retire
// with (the state block is not written before the last workgroup retires).
__device__ __forceinline__ bool link_prepare(const AdaptLink& k,
                                             TunerPrep* out) {
  if (!k.state || k.retire == ZSHMC_PEND_NONE) return false;
  *out = tuner_prepare(tuner_load(k.state), k.retire, k.fresh, k.tuner);
  return true;
}

__device__ __forceinline__ float link_step_size(const AdaptLink& k,
                                                float step_size_host) {
  if (!k.state) return step_size_host;
  TunerState s = tuner_load(k.state);
  if (k.pending != ZSHMC_PEND_NONE)
    s = tuner_apply(s, (float)(k.stats[0] * k.inv_chains), k.pending, k.fresh,
                    k.tuner);
  return s.step_size;
}

// End of a transition kernel, thread 0 of every workgroup: ONE 64-bit atomic
// add carries both this workgroup's acceptance sum (fixed point in the low
// kSumBits bits) and its retirement (count in the bits above).  Integer
// addition is associative, so the total is identical from run to run whatever
// the order (unlike floating-point atomics), and the workgroup whose add
// returns count == gridDim.x - 1 holds the complete total in the returned
// value: no second atomic, no partial array to read back, no fence (a
// release fence here flushed the L2 under the workgroups still running).
// That workgroup persists the dual-averaging update riding on this launch,
// publishes the total and clears the accumulator.
// Fixed point: 2^-shift resolution per workgroup, shift chosen by the
// launcher so that n_chains * 2^shift < 2^kSumBits (65 536 chains: 2^-34);
// the quantisation is far below the float32 mean the tuner consumes.
constexpr int kSumBits = 50;  // count: 64 - 50 = 14 bits >= log2(kFusedMaxGrid)+1

// `prep`: the acceptance-independent half of THIS transition's update
// (retire), evaluated by the caller while the transition ran (or NULL).
__device__ __forceinline__ void link_retire(const AdaptLink& k, double wg_sum,
                                            const uint32_t* flags,
                                            const TunerPrep* prep = nullptr) {
  if (!k.accum) return;
  const unsigned long long sum_fx =
      (unsigned long long)(wg_sum * k.fx_scale + 0.5);
  const unsigned long long mine = (1ull << kSumBits) + sum_fx;
  const unsigned long long before = __hip_atomic_fetch_add(
      k.accum, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((before >> kSumBits) != gridDim.x - 1) return;
  const unsigned long long all = before + mine;
  const double total =
      (double)(all & ((1ull << kSumBits) - 1)) * k.fx_inv_scale;
  if (k.state && k.pending != ZSHMC_PEND_NONE)
    tuner_persist(k, k.pending, k.stats[0]);
  if (k.state && k.retire != ZSHMC_PEND_NONE)
    tuner_persist(k, k.retire, total, prep);
  if (k.stats) {
    k.stats[0] = total;
    uint32_t f = 0;
    if (flags)
      f = __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    k.stats[1] = (f & ZSHMC_FLAG_OLD_LOGPROB_NONFINITE) ? 1.0 : 0.0;
  }
  // (every other workgroup has retired, hence read the counter long ago)
  if (k.iter_dev) *k.iter_dev += 1u;
  __hip_atomic_store(k.accum, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the iteration word of this launch's Philox counters (wave-uniform)
__device__ __forceinline__ uint32_t link_iteration(const AdaptLink& k,
                                                   uint32_t iteration) {
  if (!k.iter_dev) return iteration;
  return iteration + __hip_atomic_load(k.iter_dev, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
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
// grid of the ring launch for this many chains (= rows of colstats partials)
int64_t fused_ring_grid(int64_t n_chains);
// whether the ring instantiation of this shape can produce the column
// statistics of the end state
bool fused_ring_colstats(int64_t n_data, bool has_mass, bool zero_mean);
// largest grid either fused kernel launches (sizes the link workspace)
constexpr int kFusedMaxGrid = 4096;

}  // namespace zshmc
