/* A host program in plain C that drives the fused HMC transition through the
 * C-ABI of include/zshmc.h -- no Python, no torch: the library takes device
 * pointers and a stream, whoever owns them.
 *
 * Workload: the shape of the reference's examples/toy_examples/gaussian.py
 * (a D-dimensional Normal with a different stdev per dimension -- here
 * exp(-j / 8), whose log is exact in float32 -- many chains from q = 0),
 * L = 5, dual-averaging step-size adaptation towards 0.9 acceptance for the
 * first half of the run, frozen afterwards (hmc.py:108-110).  One kernel
 * launch per transition, the step-size update included
 * (zshmc_adapt_link.retire_update); the step-size search of hmc.py:308-345 at
 * t == 1 is the host-driven loop of single-leapfrog dry runs it is in the
 * Python front-end (zhusuan_amd/hmc.py::_search_step_size), statement for
 * statement, so the two hosts produce the same numbers bit for bit
 * (tests/test_gpu_examples.py).
 *
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude \
 *       examples/c_host/diag_gaussian_hmc.c -Lzhusuan_amd/lib -lzshmc \
 *       -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/zhusuan_amd/lib -Wl,-rpath,/opt/rocm/lib -o /tmp/hmc_c
 *   /tmp/hmc_c [n_chains] [n_data] [n_iters]
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zshmc.h"

#define CHECK_HIP(x)                                                      \
  do {                                                                    \
    hipError_t e_ = (x);                                                  \
    if (e_ != hipSuccess) {                                               \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
      return 2;                                                           \
    }                                                                     \
  } while (0)
#define CHECK_ZS(x)                                                       \
  do {                                                                    \
    if ((x) != ZSHMC_OK) {                                                \
      fprintf(stderr, "%s: %s\n", #x, zshmc_last_error());                \
      return 3;                                                           \
    }                                                                     \
  } while (0)

int main(int argc, char** argv) {
  const int64_t C = argc > 1 ? atoll(argv[1]) : 1000;
  const int64_t D = argc > 2 ? atoll(argv[2]) : 12;
  const int n_iters = argc > 3 ? atoi(argv[3]) : 400;
  const int n_leapfrogs = 5, burn_in = n_iters / 2;
  const float step_size0 = 0.05f, delta = 0.9f;
  const uint64_t seed = 1234u;

  float *q, *logstd, *state, *acc, *h0, *h1, *lp0, *lp1;
  double* stats;
  void* workspace;
  uint32_t* flags;
  CHECK_HIP(hipMalloc((void**)&q, sizeof(float) * C * D));
  CHECK_HIP(hipMalloc((void**)&logstd, sizeof(float) * D));
  CHECK_HIP(hipMalloc((void**)&state, sizeof(float) * ZSHMC_STATE_WORDS));
  CHECK_HIP(hipMalloc((void**)&stats, sizeof(double) * ZSHMC_STATS_WORDS));
  CHECK_HIP(hipMalloc(&workspace, ZSHMC_LINK_WORKSPACE_BYTES));
  CHECK_HIP(hipMalloc((void**)&flags, sizeof(uint32_t)));
  CHECK_HIP(hipMalloc((void**)&acc, sizeof(float) * C));
  CHECK_HIP(hipMalloc((void**)&h0, sizeof(float) * C));
  CHECK_HIP(hipMalloc((void**)&h1, sizeof(float) * C));
  CHECK_HIP(hipMalloc((void**)&lp0, sizeof(float) * C));
  CHECK_HIP(hipMalloc((void**)&lp1, sizeof(float) * C));
  CHECK_HIP(hipMemset(q, 0, sizeof(float) * C * D));
  CHECK_HIP(hipMemset(state, 0, sizeof(float) * ZSHMC_STATE_WORDS));
  CHECK_HIP(hipMemset(stats, 0, sizeof(double) * ZSHMC_STATS_WORDS));
  CHECK_HIP(hipMemset(workspace, 0, ZSHMC_LINK_WORKSPACE_BYTES));
  CHECK_HIP(hipMemset(flags, 0, sizeof(uint32_t)));

  float* h_logstd = (float*)malloc(sizeof(float) * D);
  for (int64_t j = 0; j < D; ++j) h_logstd[j] = -0.125f * (float)j;
  CHECK_HIP(hipMemcpy(logstd, h_logstd, sizeof(float) * D, hipMemcpyHostToDevice));
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  CHECK_ZS(zshmc_state_set(state, ZSHMC_ST_STEP_SIZE, step_size0, stream));

  zshmc_adapt_link link;
  memset(&link, 0, sizeof(link));
  link.state = state;
  link.stats = stats;
  link.workspace = workspace;
  link.n_chains_global = C;
  link.pending = ZSHMC_PEND_NONE;
  link.used_step_size = NAN;
  link.delta = delta;
  link.gamma = 0.05f;
  link.t0 = 100.f;
  link.kappa = 0.75f;
  link.mu = 10.f * step_size0; /* hmc.py:79 (sic) */

  float* h_acc = (float*)malloc(sizeof(float) * C);
  float* h_q = (float*)malloc(sizeof(float) * C * D);
  double* sum = (double*)calloc(D, sizeof(double));
  double* sq = (double*)calloc(D, sizeof(double));
  double acc_total = 0.0;
  int kept = 0;
  for (int t = 1; t <= n_iters; ++t) {
    /* the update of THIS transition, applied by the workgroup that retires
       last: adapt while burning in, then hold exp(log_epsilon_bar) */
    link.retire_update = t <= burn_in ? ZSHMC_PEND_ADAPT : ZSHMC_PEND_HOLD;
    link.fresh_start = t == 1;
    link.used_step_size = NAN;
    if (t == 1) {
      /* HMC._init_step_size (hmc.py:308-345): one full leapfrog step from the
         same (q, p0) per trip, nothing committed; grow / shrink by 1.5 until
         the mean acceptance changes sides of the target */
      float step = step_size0, last = 1.0f;
      int go = 1;
      zshmc_adapt_link dry = link;
      dry.state = NULL;          /* integrate with the host step size below */
      dry.retire_update = ZSHMC_PEND_NONE;
      dry.fresh_start = 0;
      while (go) {
        double h_stats[ZSHMC_STATS_WORDS];
        CHECK_ZS(zshmc_hmc_diag_normal_step(
            q, NULL, logstd, NULL, step, C, D, 0, 1 /* one leapfrog */, seed,
            (uint32_t)t, 0 /* dry run */, NULL, NULL, NULL, NULL, NULL, flags,
            &dry, stream));
        CHECK_HIP(hipMemcpyAsync(h_stats, stats, sizeof(h_stats),
                                 hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        const float a = (float)(h_stats[0] / (double)C);
        const float next = a < delta
            ? (float)((double)step * (double)(float)(1.0 / 1.5))
            : (float)((double)step * 1.5);
        go = !((last < delta) ^ (a < delta));
        step = next;
        last = a;
      }
      /* the searched step size travels through the state block: the kernel
         integrates with it and updates from it */
      CHECK_ZS(zshmc_state_set(state, ZSHMC_ST_STEP_SIZE, step, stream));
      link.used_step_size = step;
    }
    CHECK_ZS(zshmc_hmc_diag_normal_step(
        q, NULL /* mean = 0 */, logstd, NULL /* unit mass */, 0.0f, C, D,
        0 /* chain_offset */, n_leapfrogs, seed, (uint32_t)t,
        1 /* commit */, acc, h0, h1, lp0, lp1, flags, &link, stream));
    if (t > burn_in && t % 10 == 0) {
      CHECK_HIP(hipMemcpyAsync(h_q, q, sizeof(float) * C * D,
                               hipMemcpyDeviceToHost, stream));
      CHECK_HIP(hipMemcpyAsync(h_acc, acc, sizeof(float) * C,
                               hipMemcpyDeviceToHost, stream));
      CHECK_HIP(hipStreamSynchronize(stream));
      for (int64_t c = 0; c < C; ++c) {
        acc_total += h_acc[c];
        for (int64_t j = 0; j < D; ++j) {
          sum[j] += h_q[c * D + j];
          sq[j] += (double)h_q[c * D + j] * h_q[c * D + j];
        }
      }
      ++kept;
    }
  }
  CHECK_HIP(hipStreamSynchronize(stream));
  uint32_t h_flags = 0;
  float h_state[ZSHMC_STATE_WORDS];
  CHECK_HIP(hipMemcpy(&h_flags, flags, sizeof(h_flags), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_state, state, sizeof(h_state), hipMemcpyDeviceToHost));
  if (h_flags & ZSHMC_FLAG_OLD_LOGPROB_NONFINITE) {
    fprintf(stderr, "HMC: old_log_prob has numeric errors!\n");
    return 4;
  }
  double worst = 0.0;
  const double n = (double)kept * (double)C;
  for (int64_t j = 0; j < D; ++j) {
    const double m = sum[j] / n, sd = sqrt(sq[j] / n - m * m);
    const double rel = fabs(sd * exp(0.125 * (double)j) - 1.0);
    if (rel > worst) worst = rel;
  }
  CHECK_HIP(hipMemcpy(h_q, q, sizeof(float) * C * D, hipMemcpyDeviceToHost));
  printf("zshmc %d: %lld chains x %lld-D, %d transitions, final step size %.5f, "
         "mean acceptance %.3f, worst relative error of stdev %.4f\n",
         zshmc_version(), (long long)C, (long long)D, n_iters,
         h_state[ZSHMC_ST_STEP_SIZE], acc_total / n, worst);
  /* exact values, for the comparison with the Python front-end */
  printf("bits: step_size %a q[0][0] %a q[0][1] %a q[last][last] %a\n",
         h_state[ZSHMC_ST_STEP_SIZE], h_q[0], h_q[1], h_q[C * D - 1]);
  return 0;
}
