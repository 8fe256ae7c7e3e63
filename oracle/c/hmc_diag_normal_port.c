/* CPU restatement, in plain C + OpenMP, of ONE HMC transition for a
 * diagonal-Normal joint -- TEST / BASELINE INFRASTRUCTURE (see
 * oracle/__init__.py): bench.py times it as the all-cores CPU baseline and
 * tests/test_oracle_c_port.py holds it to the NumPy oracle (oracle/hmc_ref.py),
 * which is itself pinned to traces of the reference's hmc.py.
 *
 * Follows /root/reference/zhusuan/hmc.py:
 *   :21-23   random_momentum          p = N(0,1) * sqrt(mass)   (mass = 1)
 *   :38-43   leapfrog_integrator      q += s1 p / m ; p += s2 grad log p(q)
 *   :348-372 HMC._leapfrog            L + 1 trips, steps (0, e/2), (e, e)..., (e, e/2)
 *   :30-35   hamiltonian              H = -log p + 1/2 sum p^2 / m
 *   :46-61   get_acceptance_rate      exp(min(H0 - H1, 0)); non-finite -> 0
 *   :479-498 MH                       u < acc (strict), in-place select
 * and zhusuan/distributions/univariate.py:174-181 (Normal._log_prob summed
 * over the data axis, base.py:302-304).  Random numbers: Philox4x32-7 with
 * the counter mapping of oracle/philox.py (momentum: (d/4, chain, iteration,
 * 0); MH uniform: (0, chain, iteration, 1)), Box-Muller in double rounded
 * once to float, exactly as oracle/philox.py:box_muller.
 *
 * Unlike the NumPy oracle (one full-array pass per TF op), a chain's whole
 * trajectory stays in cache here: this is the strongest CPU formulation of
 * the path, not a model of the TF-CPU executor.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                          uint32_t k0, uint32_t k1, uint32_t out[4]) {
  for (int r = 0; r < 7; ++r) { /* oracle/philox.py: PHILOX_ROUNDS */
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void box_muller(uint32_t xa, uint32_t xb, float* z0, float* z1) {
  /* oracle/philox.py: bm_radius_uniform / bm_angle_fraction */
  const double u1 = (double)(float)(((double)(float)xa + 1.0) * 0x1p-32);
  const double u2 = (double)(xb >> 9) * 0x1p-23;
  const double r = sqrt(-2.0 * log(u1));
  const double ang = 2.0 * 3.14159265358979323846 * u2;
  *z0 = (float)(r * cos(ang));
  *z1 = (float)(r * sin(ang));
}

int zs_oracle_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* One transition for chains [0, n_chains) of q [n_chains, n_data] (updated in
 * place where accepted).  info: 5 arrays of n_chains floats (acceptance_rate,
 * orig_hamiltonian, hamiltonian, orig_log_prob, log_prob) or NULL.
 * Returns 0, or 1 if some chain started from a non-finite log-prob
 * (the reference's check_numerics, hmc.py:51-53). */
int zs_oracle_hmc_diag_normal_step(float* q, const float* mean,
                                   const float* logstd, int64_t n_chains,
                                   int64_t n_data, int64_t chain_offset,
                                   int n_leapfrogs, float step_size,
                                   uint64_t seed, uint32_t iteration,
                                   float* acceptance_rate, float* orig_hamiltonian,
                                   float* hamiltonian, float* orig_log_prob,
                                   float* log_prob, int n_threads) {
  const uint32_t k0 = (uint32_t)(seed & 0xFFFFFFFFu), k1 = (uint32_t)(seed >> 32);
  const int64_t D = n_data;
  float* prec = (float*)malloc(sizeof(float) * (size_t)D);
  float logz = 0.f;
  for (int64_t d = 0; d < D; ++d) {
    prec[d] = expf(-2.0f * logstd[d]);              /* univariate.py:178 */
    logz += -0.9189385332046727f - logstd[d];
  }
  int bad = 0;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel
  {
    float* r = (float*)malloc(sizeof(float) * (size_t)D * 2);
    float* p = r + D;
#pragma omp for schedule(static) reduction(| : bad)
    for (int64_t c = 0; c < n_chains; ++c) {
      float* qc = q + c * D;
      const uint32_t gchain = (uint32_t)(c + chain_offset);
      /* momentum (hmc.py:458) */
      for (int64_t g = 0; g < (D + 3) / 4; ++g) {
        uint32_t x[4];
        float z[4];
        philox4x32((uint32_t)g, gchain, iteration, 0u, k0, k1, x);
        box_muller(x[0], x[1], &z[0], &z[1]);
        box_muller(x[2], x[3], &z[2], &z[3]);
        for (int j = 0; j < 4 && g * 4 + j < D; ++j) p[g * 4 + j] = z[j];
      }
      float u_old = 0.f, k_old = 0.f;
      for (int64_t d = 0; d < D; ++d) {
        r[d] = qc[d] - mean[d];
        u_old += prec[d] * r[d] * r[d];
        k_old += p[d] * p[d];
      }
      const float lp_old = logz - 0.5f * u_old;
      const float h_old = -lp_old + 0.5f * k_old;
      /* leapfrog (hmc.py:348-372); grad log p = -prec * r */
      const float e = step_size;
      for (int64_t d = 0; d < D; ++d) p[d] -= 0.5f * e * prec[d] * r[d];
      for (int i = 1; i <= n_leapfrogs; ++i) {
        const float s2 = i < n_leapfrogs ? e : 0.5f * e;
        for (int64_t d = 0; d < D; ++d) {
          r[d] += e * p[d];
          p[d] -= s2 * prec[d] * r[d];
        }
      }
      float u_new = 0.f, k_new = 0.f;
      for (int64_t d = 0; d < D; ++d) {
        u_new += prec[d] * r[d] * r[d];
        k_new += p[d] * p[d];
      }
      const float lp_new = logz - 0.5f * u_new;
      const float h_new = -lp_new + 0.5f * k_new;
      const float dh = h_old - h_new;
      float acc = expf(dh < 0.f ? dh : 0.f);
      if (!(dh == dh) || !isfinite(acc) || !isfinite(lp_new)) acc = 0.f;
      if (!isfinite(lp_old)) bad |= 1;
      uint32_t x[4];
      philox4x32(0u, gchain, iteration, 1u, k0, k1, x);
      const float u = (float)(x[0] >> 8) * (1.0f / 16777216.0f);
      const int accept = u < acc;                   /* strict, hmc.py:486 */
      if (accept)
        for (int64_t d = 0; d < D; ++d) qc[d] = r[d] + mean[d];
      if (acceptance_rate) acceptance_rate[c] = acc;
      if (orig_hamiltonian) orig_hamiltonian[c] = h_old;
      if (hamiltonian) hamiltonian[c] = h_new;
      if (orig_log_prob) orig_log_prob[c] = lp_old;
      if (log_prob) log_prob[c] = accept ? lp_new : lp_old;
    }
    free(r);
  }
  free(prec);
  return bad;
}
