/*
 * zshmc.h -- C-ABI of libzshmc.so: the MI355X (gfx950) native HMC hot path
 * behind the zhusuan.HMC API.
 *
 * The reference (thu-ml/zhusuan) has NO FFI/plugin boundary: the path is a
 * Python API over a TensorFlow graph (zhusuan/hmc.py).  This header is the
 * boundary a maintainer would bind with ctypes; every
 * entry point cites the reference code it replaces (paths relative to the
 * reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - all tensors are dense, row-major ("chain-major"), float32
 *     (hmc.py:22,72-87,258-264 hard-code float32);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     every call only ENQUEUES work, nothing synchronises;
 *   - return value: ZSHMC_OK or an error code; zshmc_last_error() gives the
 *     thread-local message of the last failing call;
 *   - the caller owns every buffer.  Nothing is allocated by the library.
 *
 * Random numbers: Philox4x32-7, key = seed, counter =
 *   (d/4, global chain index, iteration, stream id | latent_id<<8); see
 *   DESIGN.md "RNG".  Results are invariant to how chains are sharded.
 */
#ifndef ZSHMC_H_
#define ZSHMC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZSHMC_VERSION 600 /* 0.6.0: + the documents' own vocabularies in the bf16x3 multinomial kernel (zshmc_linear_multinomial_log_lik_bf16x3_sparse, zshmc_model_plan.obs_sp_*); 0.5.1: packed rows, zshmc_bf16x3_multinomial_rows_packed; 0.5.0: the bf16x3 likelihood kernels, zshmc_model_plan.inner_image */

/* status codes */
#define ZSHMC_OK 0
#define ZSHMC_ERR_BAD_ARG 1
#define ZSHMC_ERR_HIP 2
#define ZSHMC_ERR_UNSUPPORTED 3
#define ZSHMC_ERR_COMM 4 /* RCCL missing or an RCCL call failed */

/* bits of the device-side `flags` word */
#define ZSHMC_FLAG_OLD_LOGPROB_NONFINITE 1u /* hmc.py:51-53 check_numerics */

/* layout of the float32 sampler state block (device, ZSHMC_STATE_WORDS) */
#define ZSHMC_STATE_WORDS 8
#define ZSHMC_ST_STEP_SIZE 0       /* HMC.step_size            hmc.py:258 */
#define ZSHMC_ST_TUNER_STEP 1      /* StepsizeTuner.step       hmc.py:82  */
#define ZSHMC_ST_LOG_EPS_BAR 2     /* .log_epsilon_bar         hmc.py:84  */
#define ZSHMC_ST_H_BAR 3           /* .h_bar                   hmc.py:86  */
#define ZSHMC_ST_EWMV_T 4          /* EWMV.t                   hmc.py:118 */
#define ZSHMC_ST_USED_STEP_SIZE 5  /* epsilon used by the last transition */
#define ZSHMC_ST_MEAN_ACCEPT 6     /* mean acceptance fed to the tuner    */
#define ZSHMC_ST_RESERVED 7

/* broadcast modes of a distribution parameter against x[rows, cols] */
#define ZSHMC_BCAST_FULL 0   /* [rows, cols]            */
#define ZSHMC_BCAST_ROW 1    /* [cols], same for each row */
#define ZSHMC_BCAST_SCALAR 2 /* [1]                      */

const char* zshmc_last_error(void);
int zshmc_version(void);
/* Rounds of the Philox4x32 generator this library was built with: 7 (the
 * default: Random123's minimum Crush-resistant count, SC'11 table 2) or 10
 * (-DZS_PHILOX_ROUNDS=10: Random123's and TensorFlow's default;
 * zhusuan_amd/lib/libzshmc_philox10.so).  The counter mapping is the same;
 * the streams differ, so a run is reproducible only on one of them. */
int zshmc_philox_rounds(void);
/* Clear n_bytes of device memory on `stream` (the += accumulators below:
 * kinetic energies, column sums). */
int zshmc_zero(void* ptr, int64_t n_bytes, void* stream);

/* Largest n_data the fused diag-Normal kernel accepts. */
int64_t zshmc_fused_max_n_data(void);

/* Name of the kernel zshmc_hmc_diag_normal_step dispatches for 16-B aligned
 * buffers of this shape ("hmc_diag_normal_ring_kernel<NCH,K,mass,*,zero_mean>"
 * or "hmc_diag_normal_kernel<G,NCH,vec,mass>"), so that benchmarks and profiler
 * summaries can name the kernel they measured.  zero_mean: mean == NULL.
 * Static storage, thread-local. */
const char* zshmc_fused_kernel_name(int64_t n_data, int has_mass,
                                    int zero_mean);

/* ------------------------------------------------------------------------
 * Fused HMC transition for a diagonal-Normal log-joint
 *      log p(q_c) = sum_d  -0.5*log(2*pi) - logstd_d
 *                          - 0.5*exp(-2*logstd_d)*(q_cd - mean_d)^2
 * ONE launch = momentum resample + (L+1) kicks / L drifts + both
 * Hamiltonians + MH accept + in-place update of q.
 *
 * Replaces, for this model family, one execution of `sample_op`:
 *   hmc.py:21-23   random_momentum           (Philox N(0,1) * sqrt(mass))
 *   hmc.py:348-372 HMC._leapfrog + :38-43 leapfrog_integrator
 *   hmc.py:30-35   hamiltonian, :46-61 get_acceptance_rate
 *   hmc.py:479-498 MH test, where(accept, q', q), Variable.assign
 *   distributions/univariate.py:174-181 Normal._log_prob and its gradient,
 *   distributions/base.py:302-304 group_ndims reduce_sum,
 *   framework/bn.py:454-465 log_joint of the single stochastic node.
 *
 *   q               [n_chains, n_data]  in/out (written only where accepted)
 *   mean, logstd    [n_data]; mean may be NULL (= zeros: the subtraction and
 *                   re-addition are compiled out)
 *   mass            [n_data] or NULL (= ones; hmc.py:456)
 *   step_size_host  used when link->state == NULL
 *   chain_offset    global index of local chain 0 (RNG counter)
 *   commit          1: full transition.  0: dry run for _init_step_size
 *                   (hmc.py:308-345): nothing but the statistics is written
 *   acceptance_rate, orig_hamiltonian, hamiltonian, orig_log_prob, log_prob
 *                   [n_chains] each, any may be NULL  (HMCInfo, hmc.py:162-201)
 *   flags           device word, OR-ed with ZSHMC_FLAG_* ; may be NULL
 *   link            HOST pointer (read during the call), see below
 *
 * zshmc_adapt_link ties consecutive transitions together so that an ADAPTIVE
 * transition (hmc.py:501-505: StepsizeTuner.tune on the mean acceptance over
 * ALL chains, :377) is still ONE launch and, when chains are sharded over
 * GPUs, ONE collective:
 *   - the launch publishes  stats[0] = sum_c acceptance_rate_c  of its own
 *     chains and stats[1] = 1 if some chain started from a non-finite
 *     log-prob (else 0).  The sum is order-independent: every workgroup adds
 *     its partial as a fixed-point integer, together with its retirement
 *     count, in ONE 64-bit atomic on `workspace` (integer addition is
 *     associative: bit-identical from run to run; no floating-point atomics);
 *   - the caller may all-reduce stats[0..1] (+ the 2*D mass statistics it
 *     keeps behind them) in ONE message (zshmc_comm_all_reduce_sum);
 *   - all chains on one GPU: `retire_update` = ZSHMC_PEND_ADAPT / _HOLD (this
 *     run's adapt_step_size flag is true / false, hmc.py:92-110) makes the
 *     workgroup that retires last apply the dual-averaging update to state[]
 *     from the total it has just formed;
 *   - chains sharded over GPUs: the NEXT launch, told `pending` = ..._ADAPT /
 *     _HOLD, applies the update in its prologue from the all-reduced
 *     stats[0] -- every workgroup computes the same scalars -- integrates
 *     with the updated step size, and its last-retiring workgroup writes the
 *     updated state[] back.  zshmc_stepsize_flush applies a pending update
 *     without a transition (before the host reads state[], e.g. for
 *     HMCInfo.updated_step_size).
 *   state == NULL: no on-device step size (step_size_host is used; pending
 *   must be ZSHMC_PEND_NONE).  stats == NULL: no statistics are collected
 *   (the steady non-adaptive phase).  workspace: ZSHMC_LINK_WORKSPACE_BYTES
 *   of device memory, zeroed once by the caller, private to one sampler.
 */
#define ZSHMC_PEND_NONE 0
#define ZSHMC_PEND_ADAPT 1 /* hmc.py:92-106 */
#define ZSHMC_PEND_HOLD 2  /* hmc.py:108-110: step_size <- exp(log_epsilon_bar) */
#define ZSHMC_STATS_WORDS 2 /* doubles: sum of acceptance, non-finite flag */
#define ZSHMC_LINK_WORKSPACE_BYTES 64
typedef struct zshmc_adapt_link {
  float* state;            /* device, ZSHMC_STATE_WORDS floats, or NULL */
  double* stats;           /* device, ZSHMC_STATS_WORDS doubles, or NULL */
  void* workspace;         /* device, ZSHMC_LINK_WORKSPACE_BYTES */
  int64_t n_chains_global; /* chains over all ranks (the mean of hmc.py:377) */
  int32_t pending;         /* ZSHMC_PEND_* : update owed by the PREVIOUS run,
                              applied in this launch's prologue from stats[0] */
  int32_t retire_update;   /* ZSHMC_PEND_* : update of THIS run, applied by the
                              workgroup that retires last from its own total
                              (use when all chains live on this GPU: no
                              all-reduce has to sit in between); exclusive
                              with `pending` */
  int32_t fresh_start;     /* the updated-for run had if_initialize_step_size
                              (hmc.py:466-467) */
  float used_step_size;    /* step size that run used if it came from the
                              search (hmc.py:308-345); NaN otherwise */
  float delta, gamma, t0, kappa; /* StepsizeTuner parameters, hmc.py:67-78 */
  float mu;                /* 10 * initial step size (hmc.py:79, sic) */
  /* Column statistics of the state the transition ENDS in, for the NEXT
   * iteration's mass update (hmc.py:130-148 takes them of the state an
   * iteration starts from, :288): with colstats_parts != NULL a committing
   * launch writes zshmc_fused_colstats_rows(n_chains, ...) rows of
   * 2*n_data doubles, row b = workgroup b's
   *   [ sum_c (q'_cd - colstats_mean_d) | sum_c (q'_cd - colstats_mean_d)^2 ]
   * over its chains (q' = the proposal where accepted, else the start row),
   * so mass adaptation costs no read pass of its own.  Reduce the rows with
   * zshmc_mass_colstats_reduce / zshmc_mass_update_fused.  Only shapes for
   * which zshmc_fused_colstats_rows returns > 0; NULL otherwise. */
  const float* colstats_mean; /* device [n_data]: the EWMV mean */
  double* colstats_parts;     /* device [rows, 2*n_data] */
} zshmc_adapt_link;

/* Rows of colstats_parts a launch of zshmc_hmc_diag_normal_step on this shape
 * fills (one per workgroup), or 0 if the kernel of this shape cannot produce
 * the column statistics (rows of <= 128 or > 1 536 latents, not a multiple
 * of 4): use zshmc_mass_colstats then.  All device pointers of the call must
 * be 16-byte aligned. */
int64_t zshmc_fused_colstats_rows(int64_t n_chains, int64_t n_data,
                                  int has_mass, int zero_mean);

int zshmc_hmc_diag_normal_step(
    float* q, const float* mean, const float* logstd, const float* mass,
    float step_size_host,
    int64_t n_chains, int64_t n_data, int64_t chain_offset,
    int n_leapfrogs, uint64_t seed, uint32_t iteration, int commit,
    float* acceptance_rate, float* orig_hamiltonian, float* hamiltonian,
    float* orig_log_prob, float* log_prob,
    uint32_t* flags, const zshmc_adapt_link* link, void* stream);

/* n_transitions consecutive transitions (iterations iteration_first,
 * iteration_first + 1, ...) from ONE call: the launch loop runs on this side
 * of the boundary, so a host language pays its per-call cost once per run.
 * Same arguments as zshmc_hmc_diag_normal_step (commit = 1); the link applies
 * to every transition of the run:
 *   - comm == NULL (all chains on this GPU): link->pending, fresh_start and
 *     used_step_size belong to the FIRST transition; link->retire_update
 *     (ZSHMC_PEND_ADAPT / _HOLD / _NONE) is what every transition of the run
 *     owes and retires itself;
 *   - comm != NULL (a zshmc_comm_create communicator: chains sharded over
 *     GPUs): after every launch stats[0..1] are all-reduced on the stream and
 *     the update of kind link->retire_update is applied by the NEXT launch's
 *     prologue; on return the LAST transition's update is still pending (pass
 *     it as link->pending of the next call, or zshmc_stepsize_flush it).
 * The mass vector is the same for the whole run and no column statistics are
 * taken (link->colstats_parts must be NULL): iterations that adapt the mass or
 * search the step size (hmc.py:284-345) go through zshmc_hmc_diag_normal_step.
 * HMCInfo arrays hold the LAST transition's values. */
int zshmc_hmc_diag_normal_run(
    float* q, const float* mean, const float* logstd, const float* mass,
    float step_size_host,
    int64_t n_chains, int64_t n_data, int64_t chain_offset,
    int n_leapfrogs, uint64_t seed, uint32_t iteration_first,
    int n_transitions,
    float* acceptance_rate, float* orig_hamiltonian, float* hamiltonian,
    float* orig_log_prob, float* log_prob,
    uint32_t* flags, const zshmc_adapt_link* link, void* comm, void* stream);

/* Apply link->pending to link->state from link->stats[0] (one tiny launch);
 * the caller then resets its pending marker to ZSHMC_PEND_NONE. */
int zshmc_stepsize_flush(const zshmc_adapt_link* link, void* stream);

/* ------------------------------------------------------------------------
 * Chain sharding over the GPUs of one node: one communicator per
 * process (one process per GPU), RCCL over xGMI, collectives enqueued on the
 * caller's stream.  librccl.so is opened at run time (the copy already in the
 * process if there is one), so single-GPU use has no RCCL dependency.
 *   zshmc_comm_unique_id  rank 0 fills 128 bytes; the caller distributes them
 *                         (any side channel) to the other ranks
 *   zshmc_comm_create     ncclCommInitRank on the current device
 *   zshmc_comm_all_reduce_sum
 *                         in-place sum of `count` device doubles: the ONE
 *                         message per transition -- stats[0..1] and, when mass
 *                         adaptation is collecting, the 2*D column sums of
 *                         hmc.py:138,143 laid out behind them
 */
#define ZSHMC_COMM_ID_BYTES 128
int zshmc_comm_unique_id(void* id_host);
int zshmc_comm_create(const void* id_host, int rank, int world_size,
                      void** comm_out);
int zshmc_comm_all_reduce_sum(void* comm, double* buf, int64_t count,
                              void* stream);
int zshmc_comm_world_size(void* comm);
int zshmc_comm_destroy(void* comm);

/* ------------------------------------------------------------------------
 * Dual-averaging step-size update as its own launch (the generic plan, whose
 * acceptance sum comes from zshmc_mh_accept), state stays on device.
 * Replaces StepsizeTuner.tune (hmc.py:89-112) + HMC._adapt_step_size
 * (hmc.py:375-380).  mean acceptance = *acc_sum / n_chains_global.
 *   adapt        this run's value of the adapt_step_size flag
 *   fresh_start  1 when if_initialize_step_size (hmc.py:466-467)
 *   mu           10 * initial step size (hmc.py:79, sic)
 * Before updating, state[ZSHMC_ST_USED_STEP_SIZE] = state[ZSHMC_ST_STEP_SIZE]
 * (the epsilon the transition just used when no search ran).  *acc_sum is
 * consumed and reset to 0 so the next transition can accumulate into it.
 */
int zshmc_stepsize_update(
    float* state, double* acc_sum, int64_t n_chains_global,
    int adapt, int fresh_start, float delta, float gamma, float t0,
    float kappa, float mu, void* stream);

/* Store `value` into state[index] (used by the host-driven step-size
 * search hmc.py:308-345 and by checkpoint restore). */
int zshmc_state_set(float* state, int index, float value, void* stream);

/* ------------------------------------------------------------------------
 * Diagonal mass adaptation (hmc.py:115-159 EWMV, :284-305 _adapt_mass).
 * Two launches so that a cross-GPU sum of `colsum` can sit between them:
 *   colstats : colsum[0:D]   += sum_c (q_cd - ewmv_mean_d)
 *              colsum[D:2D]  += sum_c (q_cd - ewmv_mean_d)^2   (double)
 *   update   : tau += 1; w = (1-decay)/(1-decay^tau);
 *              delta = w*S1/C; mean += delta;
 *              var = (1-w)*var + w*S2/C - delta^2   (== hmc.py:135-145)
 *              mass_out = ones if use_ones else 1/var (hmc.py:151-152,299-302)
 * With update == 0 only mass_out is produced (hmc.py:158-159).  An update
 * consumes colsum and resets it to 0 for the next colstats call.
 * update == 1 also advances tau (state[ZSHMC_ST_EWMV_T]) afterwards;
 * update == 2 leaves tau alone, for all but the last latent of a
 * multi-latent model (EWMV.t is shared by the latents, hmc.py:118,131).
 */
int zshmc_mass_colstats(const float* q, const float* ewmv_mean,
                        int64_t n_chains, int64_t n_data, double* colsum,
                        void* stream);
/* colsum[0:2*n_data] = sum over the n_parts rows of `parts` (the per-workgroup
 * partials a fused launch left in link->colstats_parts), added in row order:
 * deterministic.  Overwrites colsum -- what travels in the one all-reduce of
 * a sharded run. */
int zshmc_mass_colstats_reduce(const double* parts, int64_t n_parts,
                               int64_t n_data, double* colsum, void* stream);
/* zshmc_mass_update(update = 1) in ONE launch, fed by n_parts rows of column
 * sums (the partials of a fused launch; or 1 row: an already reduced /
 * all-reduced colsum): row reduction in fixed order, EWMV update, mass, and
 * the advance of tau by the workgroup that finishes last (counter in
 * `workspace`: 4 bytes of device memory, zero between calls).  `parts` is not
 * modified. */
int zshmc_mass_update_fused(float* state, float* ewmv_mean, float* ewmv_var,
                            const double* parts, int64_t n_parts,
                            int64_t n_chains_global, int64_t n_data,
                            float decay, int use_ones, float* mass_out,
                            void* workspace, void* stream);
int zshmc_mass_update(float* state, float* ewmv_mean, float* ewmv_var,
                      double* colsum, int64_t n_chains_global,
                      int64_t n_data, float decay, int update, int use_ones,
                      float* mass_out, void* stream);

/* ------------------------------------------------------------------------
 * Building blocks of the generic transition (arbitrary log-joint whose
 * gradient is supplied by the caller, e.g. autograd over the log_prob ops
 * below).  Same counter mapping as the fused kernel, so both paths draw the
 * same momentum / uniforms.
 */
/* p = N(0,1)*sqrt(mass)  (hmc.py:21-23).  kinetic[c] += 0.5*sum_d p^2/mass
 * (hmc.py:32-34) when kinetic != NULL. */
int zshmc_momentum(float* p, const float* mass, int64_t n_chains,
                   int64_t n_data, int64_t chain_offset, uint64_t seed,
                   uint32_t iteration, uint32_t latent_id, float* kinetic,
                   void* stream);
/* The same draw into the leading n_data columns of rows that are row_stride
 * floats apart (a latent's columns of a plan's packed momentum; counters are
 * those of the contiguous form, so both lay down the same numbers). */
int zshmc_momentum_rows(float* p, int64_t row_stride, const float* mass,
                        int64_t n_chains, int64_t n_data, int64_t chain_offset,
                        uint64_t seed, uint32_t iteration, uint32_t latent_id,
                        float* kinetic, void* stream);
/* p += kick_scale*eps*grad ; then q += drift_scale*eps*p/mass
 * (hmc.py:38-43 with the schedule of :352-364; drift_scale==0 skips q).
 * kinetic[c] += 0.5*sum_d p_new^2/mass when kinetic != NULL. */
int zshmc_kick_drift(float* q, float* p, const float* grad, const float* mass,
                     const float* step_size_dev, float step_size_host,
                     float kick_scale, float drift_scale, int64_t n_chains,
                     int64_t n_data, float* kinetic, void* stream);
/* MH test (hmc.py:46-61, 479-498) from per-chain log-probs and kinetic
 * energies.  accept[c] (uint8) = u < acceptance_rate; log_prob_out =
 * accept ? log_prob_new : log_prob_old. Any output may be NULL. */
int zshmc_mh_accept(const float* log_prob_old, const float* log_prob_new,
                    const float* kinetic_old, const float* kinetic_new,
                    int64_t n_chains, int64_t chain_offset, uint64_t seed,
                    uint32_t iteration, float* acceptance_rate,
                    float* orig_hamiltonian, float* hamiltonian,
                    float* log_prob_out, uint8_t* accept, double* acc_sum,
                    uint32_t* flags, void* stream);
/* q[c,:] = accept[c] ? q_new[c,:] : q[c,:]   (hmc.py:488-497) */
int zshmc_select_rows(float* q, const float* q_new, const uint8_t* accept,
                      int64_t n_chains, int64_t n_data, void* stream);
/* dst[r, 0:n_cols] = src[r, 0:n_cols] for the rows with accept[r] != 0 (all
 * rows when accept == NULL); rows dst_stride / src_stride floats apart: a
 * latent's own tensor on one side, its columns of a packed state on the
 * other (several latents, or a row padded to a multiple of 4). */
int zshmc_copy_rows(float* dst, int64_t dst_stride, const float* src,
                    int64_t src_stride, const uint8_t* accept, int64_t n_rows,
                    int64_t n_cols, void* stream);

/* ------------------------------------------------------------------------
 * One trip of HMC._leapfrog (hmc.py:348-372, :38-43) for the NATIVE plans of
 * the dense-likelihood families (BASELINE configs 3 and 5): a Normal prior on
 * the latent (univariate.py:174-181, group_ndims = 1) plus a likelihood whose
 * value and gradient come from zshmc_linear_bernoulli_log_lik (f = identity)
 * or zshmc_linear_multinomial_log_lik (f = softmax, lntm_mcem.py:39-46).
 * Everything of the trip that is not the likelihood's GEMMs, in one launch:
 *   grad  = J_f(q)^T grad_lik - exp(-2 logstd)(q - mean)   (tf.gradients of
 *           the joint, hmc.py:430-432; softmax: theta*(g - <g,theta>))
 *   p += kick_scale*eps*grad ;  q += drift_scale*eps*p/mass
 *   lp_out[c]   = lik_scale * (ll_in ? ll_in[c] : 0) + log N(q_c)
 *                                                     AT the evaluation point
 *   kinetic[c] += 1/2 sum p'^2/mass                      (if not NULL)
 *   operand[c, 0:operand_stride] = f(q') zero-padded     (if not NULL): the
 *           next likelihood evaluation's W / theta operand.  softmax != 0:
 *           operand is also READ (theta of the current q, for the Jacobian).
 * lik_scale multiplies the likelihood's log-density and gradient: 1 for the
 * joint; the temperature T of annealed importance sampling, whose target is
 * (1 - T) log prior + T (log prior + log lik) = log prior + T log lik
 * (evaluation.py:101-103).
 * grad_lik [n_chains, grad_stride] or NULL (= 0); prior_mean / prior_logstd
 * are [rows, row_stride] used with row period (r % rows) -- 1 row: shared by
 * all chains; n_docs rows: lntm's per-document eta_mean.  Rows of q, p and the
 * prior are row_stride floats apart (a multiple of 4, <= 1024; mass has
 * row_stride entries); the n_data leading columns are the latent -- several
 * Normal-prior latents packed side by side, or one whose size is not a
 * multiple of 4 -- and the columns behind them are zero padding (kept zero by
 * the caller) that enters neither the prior nor the softmax.  All buffers
 * 16-byte aligned.
 */
int zshmc_model_kick_drift(
    float* q, float* p, const float* grad_lik, int64_t grad_stride,
    float* operand, int64_t operand_stride, int softmax,
    const float* prior_mean, int64_t mean_rows, const float* prior_logstd,
    int64_t logstd_rows, const float* mass, const float* step_size_dev,
    float step_size_host, float kick_scale, float drift_scale,
    float lik_scale, int64_t n_chains, int64_t n_data, int64_t row_stride,
    const float* ll_in, float* lp_out, float* kinetic, void* stream);

/* ------------------------------------------------------------------------
 * Stand-alone distribution ops (forward, analytic backward, sampling).
 * x is viewed as [rows, cols]; reduce_cols != 0 sums each row (the
 * group_ndims reduction, distributions/base.py:302-304) and out is [rows],
 * otherwise out is [rows, cols].
 */
/* Normal._log_prob, distributions/univariate.py:174-181 */
int zshmc_normal_log_prob(const float* x, const float* mean,
                          const float* logstd, float* out, int64_t rows,
                          int64_t cols, int mean_bcast, int logstd_bcast,
                          int reduce_cols, void* stream);
/* Backward of the above given upstream gout ([rows] if reduce_cols else
 * [rows, cols]).  gx/gmean/glogstd are full [rows, cols] element-wise
 * gradients (any may be NULL); the caller sums over broadcast axes. */
int zshmc_normal_log_prob_grad(const float* x, const float* mean,
                               const float* logstd, const float* gout,
                               float* gx, float* gmean, float* glogstd,
                               int64_t rows, int64_t cols, int mean_bcast,
                               int logstd_bcast, int reduce_cols,
                               void* stream);
/* Bernoulli._log_prob, univariate.py:398-403
 * (= -sigmoid_cross_entropy_with_logits(labels=given, logits)). */
int zshmc_bernoulli_log_prob(const float* logits, const float* given,
                             float* out, int64_t rows, int64_t cols,
                             int logits_bcast, int given_bcast,
                             int reduce_cols, void* stream);
/* d/dlogits = gout * (given - sigmoid(logits)), full [rows, cols]. */
int zshmc_bernoulli_log_prob_grad(const float* logits, const float* given,
                                  const float* gout, float* glogits,
                                  int64_t rows, int64_t cols,
                                  int logits_bcast, int given_bcast,
                                  int reduce_cols, void* stream);
/* Categorical._log_prob, univariate.py:496-548
 * (= logits[k] - logsumexp(logits)); logits [rows, n_cat], labels [rows]. */
int zshmc_categorical_log_prob(const float* logits, const int64_t* labels,
                               float* out, int64_t rows, int64_t n_cat,
                               void* stream);
/* d/dlogits = gout * (onehot(k) - softmax(logits)), [rows, n_cat]. */
int zshmc_categorical_log_prob_grad(const float* logits,
                                    const int64_t* labels, const float* gout,
                                    float* glogits, int64_t rows,
                                    int64_t n_cat, void* stream);
/* UnnormalizedMultinomial._log_prob, distributions/multivariate.py:435-443:
 * sum_v given_v * (logits_v - [normalize] logsumexp(logits)). */
int zshmc_unnormalized_multinomial_log_prob(const float* logits,
                                            const float* given, float* out,
                                            int64_t rows, int64_t n_cat,
                                            int normalize, void* stream);
int zshmc_unnormalized_multinomial_log_prob_grad(
    const float* logits, const float* given, const float* gout,
    float* glogits, int64_t rows, int64_t n_cat, int normalize, void* stream);

/* ------------------------------------------------------------------------
 * K transitions of a NATIVE model plan from one call
 * (csrc/hmc_model_run.hip): the launch sequence of one transition of the
 * dense-likelihood / dense-logit Categorical / gathered-dot plans --
 *   [mass update]  momentum  likelihood  (L+1) x [step, likelihood]
 *   MH accept  select  [column sums]  [all-reduce]  step-size update
 * -- issued by the library itself, n_transitions times, so that a host
 * language pays its foreign-function overhead once per run of transitions
 * (the E-steps of lntm_mcem.py:157-182, the 1000 temperatures of AIS.run,
 * evaluation.py:119-165).  Every launch is one of the entry points above with
 * the plan's buffers -- except that behind a SPLIT likelihood launch
 * (n_splits > 1, Bernoulli / mixture-multinomial plans) the step adds the
 * row-range partials itself, in the order and arithmetic of the reduction
 * that launch would have ended with (no reduction launch inside a
 * trajectory); results are bit-identical to issuing them one by one.
 * Outside a run (one transition at a time): the step-size search
 * (hmc.py:308-345), iterations with the mass still at ones, anything that
 * changes the model's tensors.
 *
 * The plan: plain pointers and sizes, filled by the caller (device pointers
 * unless noted; the struct itself is read on the host during the call). */
#define ZSHMC_MAX_LATENTS 8
#define ZSHMC_PLAN_LINEAR_BERNOULLI 0   /* zshmc_linear_bernoulli_log_lik   */
#define ZSHMC_PLAN_MIXTURE_MULTINOMIAL 1 /* zshmc_linear_multinomial_log_lik */
#define ZSHMC_PLAN_LINEAR_CATEGORICAL 2 /* zshmc_linear_categorical_log_lik */
#define ZSHMC_PLAN_GATHERED_DOT 3       /* zshmc_gather_dot_normal_lik      */
typedef struct zshmc_model_plan {
  int32_t kind, n_latents, n_leapfrogs, softmax; /* softmax: f = softmax (1) */
  int32_t segmented, use_mass, n_splits, n_classes;
  /* the latents (updated in place where accepted) and their columns of the
   * packed working state */
  float* latent[ZSHMC_MAX_LATENTS];
  int64_t latent_size[ZSHMC_MAX_LATENTS], latent_offset[ZSHMC_MAX_LATENTS];
  float* latent_mass[ZSHMC_MAX_LATENTS]; /* per latent, [size]: written by the
                                          * mass update when adapt_mass */
  float* q_new; /* [n_chains, ld] */
  float* p;     /* [n_chains, ld] */
  int64_t n_chains, n_total, ld;
  /* likelihood: operand (or NULL: q_new itself), gradient, per-row values */
  float* operand; /* [lik_rows, width] */
  float* grad;    /* [lik_rows, width] */
  float* ll;      /* [lik_rows] */
  int64_t lik_rows, width;
  /* The likelihood evaluation AT THE STATE THE LATENTS HOLD (same shapes as
   * grad / ll), or NULL.  With them a transition's first evaluation is the
   * previous transition's last one where the chain accepted, its own first
   * one where it did not: after the MH test the accepted chains' rows of
   * grad / ll are copied over (L likelihood launches per transition instead
   * of L + 1; results bit-identical).  start_valid: they hold the evaluation
   * at the current latents on entry (else the first transition of the call
   * evaluates into them); always valid on return when n_transitions >= 1. */
  float* grad_start;
  float* ll_start;
  int32_t start_valid;
  /* ABI 0.5.0: 1 = run the L + 1 trips of a transition from ONE cooperative
   * launch where the likelihood's grid (chain blocks x n_splits) fits the
   * device at once (csrc/hmc_model_traj.hip: Bernoulli / mixture multinomial,
   * <= 256 columns, fp32 kernels; zshmc_trajectory_capacity) -- same code,
   * same order, bit-identical results; needs traj_sync.  0: one launch per
   * likelihood evaluation / element-wise step. */
  int32_t one_launch;
  const float* inner; /* X / phi^T / the other factor table */
  int64_t n_inner;    /* data rows / vocabulary / rows of the other table */
  /* ABI 0.5.0: the bf16x3 tile image of `inner` (zshmc_bf16x3_split) or NULL.
   * With it the Bernoulli, Categorical and mixture-multinomial likelihood
   * evaluations that want a gradient run on the bf16 matrix cores
   * (zshmc_linear_*_log_lik_bf16x3); NULL: the exact-fp32 kernels. */
  const void* inner_image;
  const float* obs;   /* labels / counts / ratings */
  int64_t obs_rows, obs_stride;
  float* split_ws; /* row-range partials; gathered dot: per-block sums */
  /* segmented step (zshmc_model_kick_drift_seg) */
  int64_t seg_len, groups;
  float* seg_ws;
  /* gathered dot */
  int32_t gd_latent_is_u, gd_pad;
  const int32_t *gd_idx_latent, *gd_idx_other, *gd_seg, *gd_order;
  int64_t gd_n_latent, gd_n_pairs, gd_n_dim;
  float gd_logstd, gd_pad2;
  const float* gd_lp_const;
  float* gd_g_pairs;
  /* prior (row periods as zshmc_model_kick_drift) and packed mass */
  const float* prior_mean;
  int64_t mean_rows;
  const float* prior_logstd;
  int64_t logstd_rows;
  const float* mass; /* [ld] */
  /* MH and HMCInfo */
  float *lp_old, *lp_new, *kin_old, *kin_new;
  uint8_t* accept;
  float *acceptance_rate, *orig_hamiltonian, *hamiltonian, *log_prob;
  double* acc_sum; /* = comm_buf: [sum acc, flag, column sums ...] */
  uint32_t* flags;
  float* state; /* ZSHMC_STATE_WORDS */
  int64_t chain_offset, n_chains_global;
  uint64_t seed;
  float delta, gamma, t0, kappa, mu, mass_decay;
  /* mass adaptation (hmc.py:115-159, :284-305) */
  float* ewmv_mean[ZSHMC_MAX_LATENTS];
  float* ewmv_var[ZSHMC_MAX_LATENTS];
  double* colsum[ZSHMC_MAX_LATENTS];
  double* comm_buf;
  int64_t comm_words;
  void* mass_ws;
  /* 16 bytes, zeroed once by the caller: the grid barrier's arrival counter
   * and generation, and a fault word the kernel sets (and the caller checks
   * at its next synchronisation) should a barrier wait ever run out */
  void* traj_sync;
  /* ABI 0.5.0, gathered dot: the pair list as SEGMENTS of <= 256 consecutive
   * CSR slots of one latent row (zshmc_gather_dot_normal_lik_grad: likelihood
   * and gradient in one pass); gd_seg_ptr NULL: the two-kernel form */
  const int32_t *gd_seg_ptr, *gd_seg_row, *gd_seg_first, *gd_long_rows;
  int64_t gd_n_seg, gd_n_long;
  const int32_t* gd_idx_other_csr;
  const float* gd_obs_csr;
  /* ABI 0.6.0, mixture multinomial with inner_image: the documents' OWN
   * vocabularies (zshmc_linear_multinomial_log_lik_bf16x3_sparse) -- compacted
   * counts, the words' rows of phi^T, [obs_rows + 1] offsets; obs_sp_rows
   * NULL: the dense counts in `obs`.  WITHOUT inner_image the same fields
   * route the evaluations to zshmc_sparse_multinomial_log_lik (row by row,
   * exact float32: small problems) */
  const float* obs_sp_counts;
  const int32_t* obs_sp_rows;
  const int64_t* obs_sp_off;
} zshmc_model_plan;

/*   iteration_first   Philox iteration word of the first transition
 *   update_kind       ZSHMC_PEND_NONE / _ADAPT / _HOLD: the dual-averaging
 *                     update every transition of the run owes (hmc.py:501-505)
 *   adapt_mass        1: every transition starts with the EWMV / mass update
 *                     from the column sums of its start state and ends with
 *                     the column sums of its end state (the caller provides
 *                     the first ones in colsum[]); the mass is 1 / var
 *   lik_scale_host    HOST array [n_transitions] multiplying the likelihood
 *                     term of each transition (AIS temperatures,
 *                     evaluation.py:101-103), or NULL (= 1)
 *   ais_log_weights   [n_chains] or NULL: after each transition
 *                     log_w += orig_log_prob - log_prob  (evaluation.py:150-163);
 *                     ais_ends_here: the last transition of this call is the
 *                     last temperature (its log_prob is not subtracted)
 *   comm              RCCL communicator (chains sharded) or NULL */
int zshmc_hmc_model_run(const zshmc_model_plan* plan, uint32_t iteration_first,
                        int n_transitions, int update_kind, int adapt_mass,
                        const float* lik_scale_host, float* ais_log_weights,
                        int ais_ends_here, void* comm, void* stream);
/* ABI 0.5.0: ONE transition of a native model plan and nothing around it
 * (latents -> packed state, momentum, the L + 1 trips, MH test, select, the
 * carried start evaluation): what zshmc_hmc_model_run does per transition
 * between its mass update and its statistics -- for a host loop that keeps
 * the adaptation on its side (sample_op.run) and still wants the one-launch
 * trajectory.  The step size is the state block's. */
int zshmc_hmc_model_transition(const zshmc_model_plan* plan, uint32_t iteration,
                               float lik_scale, void* stream);
/* Workgroups of the likelihood kernel of this width (64 / 128 / 192 / 256)
 * and family (ZSHMC_PLAN_LINEAR_BERNOULLI / _MIXTURE_MULTINOMIAL) that can be
 * resident at once: chain blocks x n_splits must not exceed it for the
 * one-launch trajectory (the caller sizes n_splits with it); 0: unsupported. */
int zshmc_trajectory_capacity(int64_t width, int kind, int* n_workgroups);

/* ------------------------------------------------------------------------
 * zshmc_model_kick_drift for latents that are LONG per chain and whose
 * likelihood gradient arrives in segments (csrc/hmc_model_seg.hip): the
 * dense-logit Categorical -- a chain's latent w[c, 0:K, 0:F] is K class rows
 * of seg_len = F features, rows c * groups + k of the likelihood kernel's
 * [n_chains * groups, stride] operand / gradient matrices (groups = the class
 * stride of zshmc_linear_categorical_log_lik) -- and the gathered-dot rating
 * model (pmf_hmc.py:19-31), whose gradient is a plain [n_chains, n_data]
 * matrix (groups = 1, seg_len = n_data).  f = identity, Normal prior
 * (univariate.py:174-181); per element e = k * seg_len + j of chain c:
 *   grad = lik_scale * grad_lik[(c*groups + k), j] - exp(-2 logstd)(q - mean)
 *   p += kick_scale*eps*grad ;  q += drift_scale*eps*p/mass      (hmc.py:38-43)
 *   operand[(c*groups + k), j] = q'   (if not NULL; rows k >= ceil(n_data /
 *                                      seg_len) and columns j >= seg_len are
 *                                      never touched: zero them once)
 * and per chain
 *   lp_out[c]   = lik_scale * sum_{k<groups} ll_in[c*groups + k] + log N(q_c)
 *                 AT the evaluation point;  kinetic[c] += 1/2 sum p'^2/mass.
 * q, p, prior rows: row stride `row_stride` (a multiple of 4, >= n_data);
 * mass [row_stride] or NULL; all 16-byte aligned.  seg_len need not be a
 * multiple of 4 (element-wise gradient / operand addressing then).
 * `workspace`: zshmc_model_seg_workspace(n_chains, n_data) floats -- the
 * per-chunk sums, added per chain in chunk order by a second small launch:
 * deterministic, no atomics. */
int64_t zshmc_model_seg_workspace(int64_t n_chains, int64_t n_data);
int zshmc_model_kick_drift_seg(
    float* q, float* p, const float* grad_lik, int64_t grad_stride,
    int64_t seg_len, int64_t groups, float* operand, int64_t operand_stride,
    const float* prior_mean, int64_t mean_rows, const float* prior_logstd,
    int64_t logstd_rows, const float* mass, const float* step_size_dev,
    float step_size_host, float kick_scale, float drift_scale,
    float lik_scale, int64_t n_chains, int64_t n_data, int64_t row_stride,
    const float* ll_in, float* lp_out, float* kinetic, float* workspace,
    void* stream);

/* ------------------------------------------------------------------------
 * Fused dense-logit Bernoulli likelihood on the fp32 matrix cores
 * (BASELINE config 3, Bayesian logistic regression):
 *     logits[c, n] = sum_d W[c, d] * X[n, d]
 *     log_lik[c]   = sum_n Bernoulli(logits[c, n]).log_prob(y[n])
 *     grad_w[c, :] = d log_lik[c] / d W[c, :] = sum_n (y_n - sigmoid(l)) X[n, :]
 * Replaces, for this model family, Bernoulli._log_prob (univariate.py:398-403)
 * with group_ndims = 1 (base.py:302-304) on logits = matmul(w, X^T), and what
 * tf.gradients (hmc.py:430-432) computes through them; the [C, N] logits are
 * never materialised.  W [n_chains, n_features], X [n_rows, n_features]
 * row-major, 16-byte aligned; y [n_rows] float (0/1); n_features a
 * kernel width -- zshmc_likelihood_plan(n, 0, &width, &chain_block) names the
 * one to zero-pad n columns to, and the chains a workgroup takes (64 .. 256 in
 * steps of 64: 64-chain blocks, a 32-chain block of W in each wave's registers,
 * csrc/linear_bernoulli.hip; 320 .. 896 in steps of 64: the same with 16-chain
 * blocks on the 16x16x4 MFMA, csrc/linear_bernoulli_mid.hip; 1024 -- and 512
 * for a Categorical with 32 classes -- 32-chain blocks whose four waves split
 * the features, csrc/linear_bernoulli_wide.hip); grad_w may be NULL (16-byte
 * aligned), or
 * log_lik may be (ABI 0.4.0: gradient only -- what the interior evaluations
 * of a leapfrog trajectory need, hmc.py:348-372; the element-wise stage then
 * skips the log and the kernel is 2-4 % faster), not both.
 * n_splits > 1 cuts the n_rows range into that many slices per chain block
 * (for chain counts that would otherwise leave compute units idle); the
 * partial sums (n_splits <= 256) go to `workspace` (n_splits * n_chains * (n_features + 1)
 * floats) and are added in a fixed order, so the result is deterministic.
 */
int zshmc_likelihood_plan(int64_t n_columns, int class_stride, int64_t* width,
                          int* chain_block); /* class_stride: 0 unless Categorical */
int zshmc_linear_bernoulli_log_lik(const float* W, const float* X,
                                   const float* y, int64_t n_chains,
                                   int64_t n_rows, int64_t n_features,
                                   float* log_lik, float* grad_w, int n_splits,
                                   float* workspace, void* stream);

/* ------------------------------------------------------------------------
 * ABI 0.5.0 -- the same Bernoulli / mixture-multinomial / Categorical
 * likelihoods on the BF16 matrix cores with float32-level results (csrc/b3_kernel.h):
 * every float32 operand is split into three bfloat16 planes (hi + mid + lo =
 * the value exactly) and a product is the six terms hi*hi, hi*mid, mid*hi,
 * hi*lo, lo*hi, mid*mid on v_mfma_f32_32x32x16_bf16 with float32
 * accumulation; what is dropped is <= 2^-24 relative.  Same formulas, same
 * reference lines (univariate.py:398-403, univariate.py:496-548,
 * multivariate.py:435-443, hmc.py:430-432) and same argument meaning as the
 * fp32 entry points, except:
 *   X_image / phi_image  the constant operand as a tile image, made once per
 *          tensor version by zshmc_bf16x3_split from X [n_rows, ldx] (the
 *          first `width` columns; width 64 / 128 / 192 / 256 = the fp32
 *          kernels' width for <= 256 columns); zshmc_bf16x3_image_bytes names
 *          its size (6 bytes per element, rows padded to 32);
 *   grad_w / grad_theta  required (the log-likelihood alone stays with the
 *          fp32 entry point); log_lik may be NULL;
 *   a workgroup takes 128 chains.  Mixture multinomial, count_rows > 1 (the
 *          rows are chain * count_rows + doc): 128 chains of ONE document
 *          where the chain axis fills such workgroups (chains per document a
 *          multiple of 128, or >= 1 024); otherwise -- a few chains x many
 *          documents, lntm_mcem.py:62-70's own layout --
 *          zshmc_bf16x3_multinomial_rows_packed() is 1 and a workgroup takes
 *          128 CONSECUTIVE rows, each with its own counts row (ABI 0.5.1),
 *          provided n_topics <= 192, the counts rows are 16-byte aligned and
 *          zero-padded to a multiple of 32 floats (count_stride >=
 *          round_up(n_vocab, 32)) and the counts matrix is below 4 GB; where
 *          those do not hold, one document per workgroup whatever the fill.
 * Not bit-identical to the fp32 kernels (other summation order), and held to
 * the same parity tests at the same tolerances. */
int zshmc_bf16x3_image_bytes(int64_t n_rows, int64_t width, int64_t* bytes);
int zshmc_bf16x3_split(const float* X, int64_t n_rows, int64_t width,
                       int64_t ldx, void* image, void* stream);
int zshmc_linear_bernoulli_log_lik_bf16x3(
    const float* W, const void* X_image, const float* y, int64_t n_chains,
    int64_t n_rows, int64_t n_features, float* log_lik, float* grad_w,
    int n_splits, float* workspace, void* stream);
int zshmc_bf16x3_multinomial_rows_packed(int64_t count_rows,
                                        int64_t chains_per_doc);
int zshmc_linear_multinomial_log_lik_bf16x3(
    const float* theta, const void* phi_image, const float* counts,
    int64_t count_rows, int64_t count_stride, int64_t n_rows, int64_t n_vocab,
    int64_t n_topics, float* log_lik, float* grad_theta, int n_splits,
    float* workspace, void* stream);
/* ABI 0.6.0 -- the same likelihood over the documents' OWN vocabularies.  A
 * bag of words is sparse (lntm_mcem.py's corpus: ~1 000 tokens over 12 419
 * words) and a word a document does not contain contributes exactly nothing
 * to multivariate.py:435-443 (x = 0: 0 * log S, d/dS = 0): a workgroup -- 128
 * chains of ONE document, rows chain * count_rows + doc -- runs its tile loop
 * over that document's nonzero words only, gathering their rows of the
 * phi^T image.  Document d's words are slots doc_offsets[d] ..
 * doc_offsets[d + 1] (a multiple of 32, >= 32: padded with count 0 / row 0)
 * of counts_csr (the counts) and row_index (rows of phi^T, < n_vocab).
 * Same results as the dense form up to summation order; n_splits cuts every
 * document's word list. */
int zshmc_linear_multinomial_log_lik_bf16x3_sparse(
    const float* theta, const void* phi_image, const float* counts_csr,
    const int32_t* row_index, const int64_t* doc_offsets, int64_t count_rows,
    int64_t n_rows, int64_t n_vocab, int64_t n_topics, float* log_lik,
    float* grad_theta, int n_splits, float* workspace, void* stream);
/* ABI 0.6.0 -- the same likelihood and gradient ROW BY ROW over each row's
 * own words, in exact float32 on the vector ALU (csrc/sparse_multinomial.hip):
 * the small-problem form -- lntm_mcem.py:62-70,157-182 runs ONE chain x a
 * minibatch of 100 documents, 100 rows with a word list each, which the
 * matrix-core kernels cannot use (a workgroup's chains would have to share a
 * document) and on which their launch costs 17 us whatever the arithmetic.
 * Arguments as above with phi_t [n_vocab, n_topics] float32 (n_topics = 64 /
 * 128 / 192 / 256 padded columns = the row strides of theta, phi_t and the
 * gradient) instead of the image; a workgroup per (row, slice of its word
 * list).  Deterministic. */
int zshmc_sparse_multinomial_log_lik(
    const float* theta, const float* phi_t, const float* counts_csr,
    const int32_t* row_index, const int64_t* doc_offsets, int64_t count_rows,
    int64_t n_rows, int64_t n_vocab, int64_t n_topics, float* log_lik,
    float* grad_theta, int n_splits, float* workspace, void* stream);
/* (declared here, described with zshmc_linear_categorical_log_lik below:
 * W rows are (chain, class) pairs at `class_stride` rows per chain) */
int zshmc_linear_categorical_log_lik_bf16x3(
    const float* W, const void* X_image, const float* labels, int64_t n_cols,
    int64_t n_rows, int64_t n_features, int n_classes, int class_stride,
    float* log_lik, float* grad_w, int n_splits, float* workspace,
    void* stream);

/* ------------------------------------------------------------------------
 * Fused dense-logit Categorical likelihood on the fp32 matrix cores (softmax
 * regression: w[c, k, :] ~ Normal, y_n ~ Categorical(logits = X w[c]^T)):
 *     logits[c, n, k] = sum_f X[n, f] * w[c, k, f]
 *     log_lik[c]      = sum_n logits[c, n, y_n] - logsumexp_k logits[c, n, :]
 *     grad_w[c, k, :] = sum_n ([k == y_n] - softmax_k(logits[c, n, :])) X[n, :]
 * Replaces Categorical._log_prob (univariate.py:496-548:
 * -sparse_softmax_cross_entropy_with_logits) with group_ndims = 1
 * (base.py:302-304) on logits = matmul(X, transpose(w)), and what tf.gradients
 * (hmc.py:430-432) computes through them; the [C, N, K] logits are never
 * materialised.  The same two-GEMM kernels as zshmc_linear_bernoulli_log_lik
 * with the (chain, class) pairs as their "chain rows" and the softmax taken
 * over the lanes of the accumulator that hold one chain's classes:
 *   W      [n_cols, n_features], n_cols = n_chains * class_stride: row
 *          c * class_stride + k is w[c, k, :]; class_stride = a power of two
 *          <= 32, >= n_classes; the rows k >= n_classes are padding (zero: they
 *          take no part in the softmax and get a zero gradient)
 *   X      [n_rows, n_features]; labels [n_rows] float32 holding 0..n_classes-1
 *   log_lik[n_cols]: the terms of class k's lane (those rows n with y_n = k) --
 *          log_lik[c] of the formula is the sum over the chain's class_stride
 *          entries (zshmc_model_kick_drift_seg adds them)
 *   grad_w [n_cols, n_features] or NULL.
 * n_features, alignment, n_splits / workspace as zshmc_linear_bernoulli_log_lik
 * (workspace: n_splits * n_cols * (n_features + 1) floats). */
int zshmc_linear_categorical_log_lik(
    const float* W, const float* X, const float* labels, int64_t n_cols,
    int64_t n_rows, int64_t n_features, int n_classes, int class_stride,
    float* log_lik, float* grad_w, int n_splits, float* workspace,
    void* stream);

/* Sampling (Normal._sample univariate.py:161-172, Bernoulli._sample
 * :386-396, Categorical._sample :478-494) on Philox stream STREAM_DIST with
 * counter (i/4, offset).  n = total number of output elements. */
int zshmc_normal_sample(float* out, const float* mean, const float* std,
                        int64_t n, int64_t inner, int mean_bcast,
                        int std_bcast, uint64_t seed, uint32_t offset,
                        void* stream);
int zshmc_bernoulli_sample(int32_t* out, const float* logits, int64_t n,
                           int64_t inner, uint64_t seed, uint32_t offset,
                           void* stream);
int zshmc_categorical_sample(int32_t* out, const float* logits,
                             int64_t n_samples, int64_t rows, int64_t n_cat,
                             uint64_t seed, uint32_t offset, void* stream);

/* ------------------------------------------------------------------------
 * Gathered row dot products (the rating logits of
 * examples/probabilistic_matrix_factorization/pmf_hmc.py:26-28):
 *   out[k, e] = sum_d u[k, select_u[e], d] * v[k, select_v[e], d]
 * u [n_chains, n_u, n_dim], v [n_chains, n_v, n_dim] row-major, indices int32
 * in range (the caller validates them), out [n_chains, n_pairs].  Replaces
 * tf.gather(u, select_u, axis=1) * tf.gather(v, select_v, axis=1) summed over
 * axis 2 without materialising the [K, E, D] gathers.
 * zshmc_gather_dot_grad is what tf.gradients gives for ONE side:
 *   grad[k, i, :] = sum_{e : own_index[e] = i} gout[k, e] * other[k, other_index[e], :]
 * over a CSR view of the pair list: `order` lists the pair ids grouped by
 * own_index, `seg_ptr[i] .. seg_ptr[i+1]` is row i's slice of it
 * (n_rows + 1 offsets).  No atomics: deterministic. */
/* ABI 0.5.0 -- the rating likelihood of pmf_hmc.py:26-31 AND its gradient
 * w.r.t. the latent table in ONE pass (csrc/gather_dot.hip: gd_fused_kernel).
 * The pair list arrives as a CSR view by latent row, cut into segments:
 *   seg_ptr [n_segments + 1]  CSR slots of segment s: seg_ptr[s] .. seg_ptr[s+1]
 *                             (<= 256 of them, all of latent row seg_row[s];
 *                             consecutive segments are consecutive slots)
 *   seg_row [n_segments], seg_first [n_latent] (a row's first segment; every
 *   row has >= 1, rows without pairs an empty one), long_rows [n_long] (the
 *   rows with more than one segment)
 *   other_index_csr [n_pairs] the other table's row, obs_csr [obs_rows,
 *   n_pairs] the rating, of CSR slot q
 *   log_lik[k] = sum over pairs of log N(obs; sigmoid(u_i . v_j), exp(logstd))
 *                + lp_const[k];  grad [n_chains, n_latent, n_dim]
 *   n_dim <= 128, a multiple of 4; tables 16-byte aligned;
 *   workspace: 16-byte aligned, n_chains * n_segments * n_dim +
 *              round_up(n_chains * n_segments, 4) floats.
 * Deterministic (fixed summation order, no atomics). */
int zshmc_gather_dot_normal_lik_grad(
    const float* latent, const float* other, const int32_t* seg_ptr,
    const int32_t* seg_row, const int32_t* seg_first, const int32_t* long_rows,
    int64_t n_long, const int32_t* other_index_csr, const float* obs_csr,
    int64_t obs_rows, float logstd, const float* lp_const, int64_t n_chains,
    int64_t n_latent, int64_t n_other, int64_t n_pairs, int64_t n_segments,
    int64_t n_dim, float* grad, float* log_lik, float* workspace, void* stream);
int zshmc_gather_dot(const float* u, const float* v, const int32_t* select_u,
                     const int32_t* select_v, int64_t n_chains, int64_t n_u,
                     int64_t n_v, int64_t n_pairs, int64_t n_dim, float* out,
                     void* stream);
/* The rating likelihood of pmf_hmc.py:26-31 in one pass over the pair list
 * (the per-trip evaluation of the gathered-dot model's NATIVE plan):
 *   d[k, e]    = sum_j u[k, select_u[e], j] * v[k, select_v[e], j]
 *   log_lik[k] = sum_e log N(obs[e]; sigmoid(d[k, e]), exp(logstd))
 *                + (lp_const ? lp_const[k] : 0)
 *                (bn.normal("r", tf.sigmoid(r_logits), std=alpha_pred) summed
 *                over the pairs, univariate.py:174-181; lp_const: the
 *                log-densities of the observed nodes that do not depend on
 *                the latent, e.g. log p(v) while u is sampled)
 *   g_out[k, e] = d log_lik[k] / d d[k, e]   (or NULL) -- what
 *                zshmc_gather_dot_grad scatters into the latent's gradient.
 * obs [obs_rows, n_pairs] with obs_rows 1 (shared by the chains) or n_chains.
 * Per-block partial sums (n_splits <= 256) go to `workspace`
 * (zshmc_gather_dot_normal_workspace(n_chains, n_pairs) floats) and are added
 * per chain in block order: deterministic, no atomics. */
int64_t zshmc_gather_dot_normal_workspace(int64_t n_chains, int64_t n_pairs);
int zshmc_gather_dot_normal_lik(
    const float* u, const float* v, const int32_t* select_u,
    const int32_t* select_v, const float* obs, int64_t obs_rows, float logstd,
    const float* lp_const, int64_t n_chains, int64_t n_u, int64_t n_v,
    int64_t n_pairs, int64_t n_dim, float* g_out, float* log_lik,
    float* workspace, void* stream);
int zshmc_gather_dot_grad(const float* other, const float* gout,
                          const int32_t* seg_ptr, const int32_t* order,
                          const int32_t* other_index, int64_t n_chains,
                          int64_t n_rows, int64_t n_other, int64_t n_pairs,
                          int64_t n_dim, float* grad, void* stream);

/* ------------------------------------------------------------------------
 * MultivariateNormalCholesky (zhusuan/distributions/
 * multivariate.py:41-193).  x [n_rows, n_dim]; row r uses
 * mean[r % mean_rows, :] and the lower-triangular factor
 * tril[r % tril_count, :, :] (row-major [n_dim, n_dim]; the strict upper
 * triangle is never read).  1 <= n_dim <= 512.
 *   log_prob[r] = -n_dim/2 log(2 pi) - sum_i log L_ii - 1/2 |z|^2,
 *                 z = L^{-1} (x_r - mean)                    (_log_prob :166-188)
 *   grad_x[r,:] = -L^{-T} z     (what tf.gradients gives w.r.t. `given`; NULL
 *                                to skip the back substitution)
 *   z_out[r,:]  = z             (optional; the parameter gradients are
 *                                d/dmean = -grad_x, d/dL = tril(-grad_x z^T) - diag(1/L_ii))
 * zshmc_mvn_tril_sample: out[r,:] = mean + L . n, n ~ N(0, I) (_sample
 * :141-164) on Philox stream STREAM_DIST, noise element i = r * n_dim + c
 * taken from counter (i / 4, offset) word i % 4, as zshmc_normal_sample. */
int zshmc_mvn_tril_log_prob(const float* x, const float* mean,
                            const float* tril, int64_t n_rows, int64_t n_dim,
                            int64_t mean_rows, int64_t tril_count,
                            float* log_prob, float* grad_x, float* z_out,
                            void* stream);
int zshmc_mvn_tril_sample(float* out, const float* mean, const float* tril,
                          int64_t n_rows, int64_t n_dim, int64_t mean_rows,
                          int64_t tril_count, uint64_t seed, uint32_t offset,
                          void* stream);

/* ------------------------------------------------------------------------
 * Fused mixture-multinomial likelihood + gradient (fp32 MFMA), the E-step
 * likelihood of the logistic-normal topic model
 * (examples/topic_models/lntm_mcem.py:33-48): for every row r
 *   ll[r]   = sum_v counts[r % count_rows, v] * log( sum_k theta[r,k] phi_t[v,k] )
 *           = UnnormalizedMultinomial(log(theta.phi), normalize_logits=False)
 *             .log_prob(counts)                     multivariate.py:435-443
 *   grad_theta[r,k] = sum_v counts[.,v] * phi_t[v,k] / (theta.phi)[r,v]
 * theta [n_rows, n_topics], phi_t = phi^T [n_vocab, n_topics] (n_topics 64,
 * 128, 256, 512 or 1024: zero-pad; above 256 the feature-split kernel of
 * csrc/linear_bernoulli_wide.hip, grad_theta 16-byte aligned), counts
 * [count_rows, n_vocab] with `count_stride`
 * floats between rows, shared by the rows with period count_rows.  A
 * count_stride that is a multiple of 4 (>= n_vocab rounded up to 4, the pad
 * zero-filled) on a 16-byte aligned base lets every lane fetch its counts --
 * a gather, one matrix row per chain -- 16 bytes at a time.  The [n_rows, n_vocab] product never reaches memory.
 * grad_theta may be NULL.  n_splits > 1 splits the vocabulary into that many
 * row ranges handled by separate workgroups (for n_rows / 64 < #CUs) whose
 * partial sums land in `workspace` (n_splits * n_rows * (n_topics + 1) floats,
 * caller-owned) and are reduced in a fixed order; n_splits = 1: workspace NULL.
 */
int zshmc_linear_multinomial_log_lik(const float* theta, const float* phi_t,
                                     const float* counts, int64_t count_rows,
                                     int64_t count_stride,
                                     int64_t n_rows, int64_t n_vocab,
                                     int64_t n_topics, float* log_lik,
                                     float* grad_theta, int n_splits,
                                     float* workspace, void* stream);

/* ------------------------------------------------------------------------
 * Batched effective sample size (zhusuan/diagnostics.py:17-64), the estimator
 * behind BASELINE.json's "ESS/s".
 *   draws : [n_draws, n_series] float32, draw-major (one recorded snapshot of
 *           the flattened [chains, dims] state per draw, burn-in already
 *           dropped); ess : [n_series].
 * Per series exactly effective_sample_size_1d (diagnostics.py:17-40): var =
 * np.var*n/(n-1), lag-t autocovariance = mean over the n-t products, the rho
 * sum starts at lag 0 and stops at the first negative rho; float64 sums.
 * zshmc_min_positive_rows then gives diagnostics.py:55-64's "minimum positive
 * ESS over dimensions" per chain: out[r] = min{v[r, c] : v[r, c] > 0}, +inf
 * if the row has none.
 */
int zshmc_ess_series(const float* draws, int64_t n_draws, int64_t n_series,
                     float* ess, void* stream);
int zshmc_min_positive_rows(const float* v, int64_t rows, int64_t cols,
                            float* out, void* stream);

/* ------------------------------------------------------------------------
 * Stochastic-gradient MCMC updates (zhusuan/sgmcmc.py), element-wise over a
 * flat latent of n float32 elements; the caller supplies grad = d log p / dq
 * of the (mini-batch) log joint (tf.gradients, sgmcmc.py:95-99).  Gaussian
 * terms are generated in the kernel: Philox counter (i/4 lo, i/4 hi,
 * iteration, 3 | sub<<4 | latent_id<<8), sub 0 = step noise, 1 = momentum.
 *
 * zshmc_sgld_update   SGLD  q += lr/2 g + N(0, lr)              sgmcmc.py:199-204
 *                     PSGLD (aux != NULL): aux = decay aux + (1-decay) g^2,
 *                     G = 1/(epsilon + sqrt(aux)), q += lr/2 G g + N(0, lr G)
 *                                                                sgmcmc.py:221-253
 * zshmc_sg_momentum   v = N(0, std^2)  (initial value and resampling,
 *                     std = sqrt(lr))                            sgmcmc.py:310-318
 * zshmc_sg_half_drift q += v/2 (2nd-order integrators, :341, :463); optionally
 *                     v2_sum += sum v^2 (scalar-friction SGNHT, :464)
 * zshmc_sghmc_update  SGHMC._update (:331-349); v2_sum += sum v'^2 (mean_k)
 * zshmc_sgnht_update  SGNHT._update (:452-481) with a friction per element
 *                     (alpha_vec) or the scalar alpha_scalar[1]
 * zshmc_sgnht_scalar  scalar-friction bookkeeping; alpha_scalar = {alpha,
 *                     alpha of this step}, sums = {sum v_old^2, sum v'^2};
 *                     phase 0 before, phase 1 after the update kernel
 */
int zshmc_sgld_update(float* q, const float* grad, float* aux,
                      float learning_rate, float decay, float epsilon,
                      int64_t n, uint64_t seed, uint32_t iteration,
                      uint32_t latent_id, void* stream);
int zshmc_sg_momentum(float* v, float std, int64_t n, uint64_t seed,
                      uint32_t iteration, uint32_t latent_id, void* stream);
int zshmc_sg_half_drift(float* q, const float* v, int64_t n, double* v2_sum,
                        void* stream);
int zshmc_sghmc_update(float* q, float* v, const float* grad, int64_t n,
                       float learning_rate, float friction, float noise_std,
                       int second_order, uint64_t seed, uint32_t iteration,
                       uint32_t latent_id, double* v2_sum, void* stream);
int zshmc_sgnht_update(float* q, float* v, const float* grad, float* alpha_vec,
                       const float* alpha_scalar, float* mean_k_vec, int64_t n,
                       float learning_rate, float tune_rate, float noise_std,
                       int second_order, uint64_t seed, uint32_t iteration,
                       uint32_t latent_id, double* v2_sum, void* stream);
int zshmc_sgnht_scalar(float* alpha_scalar, double* sums, int64_t n,
                       float learning_rate, float tune_rate, int second_order,
                       int phase, float* mean_k_out, void* stream);

/* ------------------------------------------------------------------------
 * Two-parameter continuous families (zhusuan/distributions/univariate.py):
 *   kind 0 Laplace(loc, scale) :1164-1277      kind 1 Gamma(alpha, beta) :662-751
 *   kind 2 InverseGamma(alpha, beta) :1070-1161  kind 3 Beta(alpha, beta) :753-855
 * Same conventions as zshmc_normal_log_prob[_grad]: x is [rows, cols], a / b are
 * broadcast per their mode, reduce_cols sums each row (group_ndims,
 * base.py:302-304); the gradient call fills any of gx / ga / gb that is not
 * NULL with gout * d log_prob / d(value, a, b) at full [rows, cols] shape.
 * zshmc_uni2_sample draws n = n_samples * inner values (parameters indexed by
 * i % inner, FULL over one sample or SCALAR); Gamma-type draws restate
 * tf.random_gamma (:725-727) as Marsaglia-Tsang rejection on the Philox stream.
 */
#define ZSHMC_UNI2_LAPLACE 0
#define ZSHMC_UNI2_GAMMA 1
#define ZSHMC_UNI2_INVERSE_GAMMA 2
#define ZSHMC_UNI2_BETA 3
int zshmc_uni2_log_prob(int kind, const float* x, const float* a, const float* b,
                        float* out, int64_t rows, int64_t cols, int a_bcast,
                        int b_bcast, int reduce_cols, void* stream);
int zshmc_uni2_log_prob_grad(int kind, const float* x, const float* a,
                             const float* b, const float* gout, float* gx,
                             float* ga, float* gb, int64_t rows, int64_t cols,
                             int a_bcast, int b_bcast, int reduce_cols,
                             void* stream);
int zshmc_uni2_sample(int kind, float* out, const float* a, const float* b,
                      int64_t n, int64_t inner, int a_bcast, int b_bcast,
                      uint64_t seed, uint32_t offset, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ZSHMC_H_ */
