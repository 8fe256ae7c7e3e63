// Stochastic-gradient MCMC update kernels on gfx950:
// the element-wise updates of reference zhusuan/sgmcmc.py with the Gaussian
// noise generated in the kernel (Philox4x32-7 + Box-Muller, counter
// (i/4 lo, i/4 hi, iteration, STREAM_SG | sub<<4 | latent<<8) over the flat
// element index i), so one launch per latent replaces the reference's
// random_normal op + 5-10 element-wise TF ops + assigns.  The gradient of the
// (mini-batch) log joint comes from the caller, as tf.gradients does in
// sgmcmc.py:95-99.  HBM-bound: 12-20 B per element per iteration.
//
//   sgld_kernel        SGLD._update_single   sgmcmc.py:199-204
//                      PSGLD (RMSprop)       sgmcmc.py:221-253
//   sg_momentum_kernel v ~ N(0, lr)          sgmcmc.py:310-314, :317-318
//   sg_half_drift      q1 = q + 0.5*v        sgmcmc.py:341, :463
//   sghmc_kernel       SGHMC._update         sgmcmc.py:331-349
//   sgnht_kernel       SGNHT._update         sgmcmc.py:452-481
//   sgnht_scalar_finalize  scalar friction   sgmcmc.py:459-461, :466-468, :477-480
#include "common.h"
#include "philox.h"

namespace zshmc {

constexpr uint32_t kStreamSG = 3;
constexpr uint32_t kSubNoise = 0;     // the per-step Gaussian term
constexpr uint32_t kSubMomentum = 1;  // momentum (re)sampling

__device__ __forceinline__ uint32_t sg_word(uint32_t sub, uint32_t latent_id) {
  return kStreamSG | (sub << 4) | (latent_id << 8);
}

static inline int sg_grid(int64_t n_groups) {
  const int64_t need = (n_groups + 255) / 256;
  const int64_t cap = (int64_t)device_cu_count() * 16;
  const int64_t g = need < cap ? need : cap;
  return (int)(g > 0 ? g : 1);
}

typedef float sg4 __attribute__((ext_vector_type(4)));

// 4 consecutive elements starting at 4g: one 16-B access when the group is
// whole and the array 16-B aligned (VEC), element-wise otherwise
template <bool VEC>
__device__ __forceinline__ sg4 sg_load(const float* __restrict__ p, int64_t i0,
                                       int64_t n) {
  if (VEC && i0 + 3 < n) return *reinterpret_cast<const sg4*>(p + i0);
  sg4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (i0 + j < n) v[j] = p[i0 + j];
  return v;
}
template <bool VEC>
__device__ __forceinline__ void sg_store(float* __restrict__ p, int64_t i0,
                                         int64_t n, sg4 v) {
  if (VEC && i0 + 3 < n) {
    *reinterpret_cast<sg4*>(p + i0) = v;
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (i0 + j < n) p[i0 + j] = v[j];
}

// grid-stride over 4-element groups: `z` holds the group's four N(0,1), `i0`
// its first flat index
#define ZS_SG_FOREACH4(n, word, BODY)                                         \
  const int64_t n_groups_ = ((n) + 3) / 4;                                    \
  for (int64_t g_ = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;           \
       g_ < n_groups_; g_ += (int64_t)gridDim.x * blockDim.x) {               \
    sg4 z;                                                                    \
    {                                                                         \
      float z0_, z1_, z2_, z3_;                                               \
      normal4((uint32_t)g_, (uint32_t)((uint64_t)g_ >> 32), iteration,        \
              (word), k0, k1, z0_, z1_, z2_, z3_);                            \
      z = sg4{z0_, z1_, z2_, z3_};                                            \
    }                                                                         \
    const int64_t i0 = g_ * 4;                                                \
    BODY                                                                      \
  }

template <bool VEC>
__global__ __launch_bounds__(256) void sgld_kernel(
    float* __restrict__ q, const float* __restrict__ grad,
    float* __restrict__ aux, float lr, float decay, float epsilon, int64_t n,
    uint32_t k0, uint32_t k1, uint32_t iteration, uint32_t latent_id) {
  const float sqrt_lr = sqrtf(lr);
  ZS_SG_FOREACH4(n, sg_word(kSubNoise, latent_id), {
    const sg4 g = sg_load<VEC>(grad, i0, n);
    sg4 qv = sg_load<VEC>(q, i0, n);
    if (aux) {  // PSGLD, RMSprop preconditioner (sgmcmc.py:233-236, :247-250)
      sg4 a = sg_load<VEC>(aux, i0, n);
      a = decay * a + (1.0f - decay) * (g * g);
      sg_store<VEC>(aux, i0, n, a);
_Pragma("unroll")
      for (int j = 0; j < 4; ++j) {
        const float pre = 1.0f / (epsilon + sqrtf(a[j]));
        qv[j] = qv[j] + 0.5f * lr * pre * g[j] + z[j] * sqrtf(lr * pre);
      }
    } else {    // SGLD (sgmcmc.py:200-201)
      qv = qv + (0.5f * lr) * g + z * sqrt_lr;
    }
    sg_store<VEC>(q, i0, n, qv);
  })
}

template <bool VEC>
__global__ __launch_bounds__(256) void sg_momentum_kernel(
    float* __restrict__ v, float std, int64_t n, uint32_t k0, uint32_t k1,
    uint32_t iteration, uint32_t latent_id) {
  ZS_SG_FOREACH4(n, sg_word(kSubMomentum, latent_id),
                 { sg_store<VEC>(v, i0, n, z * std); })
}

// q <- q + 0.5*v ; optionally sum(v^2) for the scalar-friction thermostat
template <bool VEC>
__global__ __launch_bounds__(256) void sg_half_drift_kernel(
    float* __restrict__ q, const float* __restrict__ v, int64_t n,
    double* __restrict__ v2_sum) {
  double acc = 0.0;
  const int64_t n_groups = (n + 3) / 4;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups;
       g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = g * 4;
    const sg4 vv = sg_load<VEC>(v, i0, n);
    sg_store<VEC>(q, i0, n, sg_load<VEC>(q, i0, n) + 0.5f * vv);
    const sg4 sq = vv * vv;  // elements past n were loaded as 0
    acc += (double)((sq[0] + sq[1]) + (sq[2] + sq[3]));
  }
  if (v2_sum) {
    const double w = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0 && w != 0.0) atomicAdd(v2_sum, w);
  }
}

// SGHMC (sgmcmc.py:331-349).  First order: v' = (1-alpha) v + lr g + noise,
// q' = q + v'.  Second order (q already holds q1 = q + v/2): v' = d (d v +
// lr g + noise), d = exp(-alpha/2), q' = q1 + v'/2.  sum(v'^2) -> v2_sum.
template <bool VEC>
__global__ __launch_bounds__(256) void sghmc_kernel(
    float* __restrict__ q, float* __restrict__ v,
    const float* __restrict__ grad, int64_t n, float lr, float alpha,
    float noise_std, int second_order, uint32_t k0, uint32_t k1,
    uint32_t iteration, uint32_t latent_id, double* __restrict__ v2_sum) {
  const float dh = expf(-0.5f * alpha);
  double acc = 0.0;
  ZS_SG_FOREACH4(n, sg_word(kSubNoise, latent_id), {
    const sg4 noise = z * noise_std;
    const sg4 ov = sg_load<VEC>(v, i0, n);
    const sg4 g = sg_load<VEC>(grad, i0, n);
    const sg4 qv = sg_load<VEC>(q, i0, n);
    sg4 nv;
    if (second_order) {
      nv = dh * (dh * ov + lr * g + noise);
      sg_store<VEC>(q, i0, n, qv + 0.5f * nv);
    } else {
      nv = (1.0f - alpha) * ov + lr * g + noise;
      sg_store<VEC>(q, i0, n, qv + nv);
    }
    sg_store<VEC>(v, i0, n, nv);
_Pragma("unroll")
    for (int j = 0; j < 4; ++j)
      if (i0 + j < n) acc += (double)nv[j] * (double)nv[j];
  })
  if (v2_sum) {
    const double w = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0 && w != 0.0) atomicAdd(v2_sum, w);
  }
}

// SGNHT (sgmcmc.py:452-481).  alpha_vec != NULL: one friction per element,
// everything element-wise.  alpha_vec == NULL: scalar friction alpha1 read
// from alpha_scalar[1] (prepared by sgnht_scalar_kernel) and sum(v'^2)
// accumulated for its second phase.
template <bool VEC>
__global__ __launch_bounds__(256) void sgnht_kernel(
    float* __restrict__ q, float* __restrict__ v,
    const float* __restrict__ grad, float* __restrict__ alpha_vec,
    const float* __restrict__ alpha_scalar, float* __restrict__ mean_k_vec,
    int64_t n, float lr, float tune_rate, float noise_std, int second_order,
    uint32_t k0, uint32_t k1, uint32_t iteration, uint32_t latent_id,
    double* __restrict__ v2_sum) {
  // scalar mode: alpha_scalar[1] = the friction this step integrates with
  // (alpha for first order, alpha1 for second order)
  const float a_s = alpha_scalar ? alpha_scalar[1] : 0.f;
  const float dh_s = expf(-0.5f * a_s);
  double acc = 0.0;
  ZS_SG_FOREACH4(n, sg_word(kSubNoise, latent_id), {
    const sg4 noise = z * noise_std;
    const sg4 ov = sg_load<VEC>(v, i0, n);
    const sg4 g = sg_load<VEC>(grad, i0, n);
    const sg4 qv = sg_load<VEC>(q, i0, n);
    sg4 nv;
    if (alpha_vec) {
      const sg4 al = sg_load<VEC>(alpha_vec, i0, n);
      sg4 na;
      if (second_order) {
        const sg4 a1 = al + (0.5f * tune_rate) * (ov * ov - lr);
        sg4 dh;
_Pragma("unroll")
        for (int j = 0; j < 4; ++j) dh[j] = expf(-0.5f * a1[j]);
        nv = dh * (dh * ov + lr * g + noise);
        sg_store<VEC>(q, i0, n, qv + 0.5f * nv);
        na = a1 + (0.5f * tune_rate) * (nv * nv - lr);
      } else {
        nv = (1.0f - al) * ov + lr * g + noise;
        sg_store<VEC>(q, i0, n, qv + nv);
        na = al + tune_rate * (nv * nv - lr);
      }
      sg_store<VEC>(alpha_vec, i0, n, na);
      if (mean_k_vec) sg_store<VEC>(mean_k_vec, i0, n, nv * nv);
    } else {
      if (second_order) {
        nv = dh_s * (dh_s * ov + lr * g + noise);
        sg_store<VEC>(q, i0, n, qv + 0.5f * nv);
      } else {
        nv = (1.0f - a_s) * ov + lr * g + noise;
        sg_store<VEC>(q, i0, n, qv + nv);
      }
    }
    sg_store<VEC>(v, i0, n, nv);
_Pragma("unroll")
    for (int j = 0; j < 4; ++j)
      if (i0 + j < n) acc += (double)nv[j] * (double)nv[j];
  })
  if (v2_sum) {
    const double w = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0 && w != 0.0) atomicAdd(v2_sum, w);
  }
}

// Scalar friction bookkeeping, one thread.  alpha_scalar = {alpha, alpha_step}.
//   phase 0 (before the update): second order: alpha_step = alpha + 0.5*tune*
//           (mean(v_old^2) - lr) from sums[0]; first order: alpha_step = alpha.
//   phase 1 (after): mean_k = sums[1]/n; alpha = alpha_step + (0.5|1)*tune*
//           (mean_k - lr); mean_k_out[0] = mean_k.  Consumed sums are zeroed.
__global__ void sgnht_scalar_kernel(float* __restrict__ alpha_scalar,
                                    double* __restrict__ sums, double n,
                                    float lr, float tune_rate, int second_order,
                                    int phase, float* __restrict__ mean_k_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (phase == 0) {
    if (second_order) {
      const float mk1 = (float)(sums[0] / n);
      alpha_scalar[1] = alpha_scalar[0] + 0.5f * tune_rate * (mk1 - lr);
    } else {
      alpha_scalar[1] = alpha_scalar[0];
    }
    sums[0] = 0.0;
  } else {
    const float mk = (float)(sums[1] / n);
    alpha_scalar[0] =
        alpha_scalar[1] + (second_order ? 0.5f : 1.0f) * tune_rate * (mk - lr);
    if (mean_k_out) mean_k_out[0] = mk;
    sums[1] = 0.0;
  }
}

}  // namespace zshmc

using namespace zshmc;

#define ZS_SG_KEYS (uint32_t)(seed & 0xFFFFFFFFull), (uint32_t)(seed >> 32)

static inline bool al16(const void* p) {
  return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}
// launch KERNEL<true> when every array is 16-B aligned, else KERNEL<false>
#define ZS_SG_LAUNCH(KERNEL, vec, grid, strm, ...)                             \
  do {                                                                         \
    if (vec)                                                                   \
      hipLaunchKernelGGL(KERNEL<true>, dim3(grid), dim3(256), 0, strm,         \
                         __VA_ARGS__);                                         \
    else                                                                       \
      hipLaunchKernelGGL(KERNEL<false>, dim3(grid), dim3(256), 0, strm,        \
                         __VA_ARGS__);                                         \
  } while (0)

extern "C" int zshmc_sgld_update(float* q, const float* grad, float* aux,
                                 float learning_rate, float decay, float epsilon,
                                 int64_t n, uint64_t seed, uint32_t iteration,
                                 uint32_t latent_id, void* stream) {
  if (n == 0) return ZSHMC_OK;
  ZS_REQUIRE(q && grad && n > 0, "zshmc_sgld_update: bad arguments");
  ZS_REQUIRE(learning_rate >= 0.f, "zshmc_sgld_update: learning_rate < 0");
  ZS_SG_LAUNCH(sgld_kernel, al16(q) && al16(grad) && al16(aux),
               sg_grid((n + 3) / 4), reinterpret_cast<hipStream_t>(stream), q,
               grad, aux, learning_rate, decay, epsilon, n, ZS_SG_KEYS,
               iteration, latent_id);
  ZS_LAUNCH_CHECK("sgld_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_sg_momentum(float* v, float std, int64_t n, uint64_t seed,
                                 uint32_t iteration, uint32_t latent_id,
                                 void* stream) {
  if (n == 0) return ZSHMC_OK;
  ZS_REQUIRE(v && n > 0, "zshmc_sg_momentum: bad arguments");
  ZS_SG_LAUNCH(sg_momentum_kernel, al16(v), sg_grid((n + 3) / 4),
               reinterpret_cast<hipStream_t>(stream), v, std, n, ZS_SG_KEYS,
               iteration, latent_id);
  ZS_LAUNCH_CHECK("sg_momentum_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_sg_half_drift(float* q, const float* v, int64_t n,
                                   double* v2_sum, void* stream) {
  if (n == 0) return ZSHMC_OK;
  ZS_REQUIRE(q && v && n > 0, "zshmc_sg_half_drift: bad arguments");
  ZS_SG_LAUNCH(sg_half_drift_kernel, al16(q) && al16(v), sg_grid((n + 3) / 4),
               reinterpret_cast<hipStream_t>(stream), q, v, n, v2_sum);
  ZS_LAUNCH_CHECK("sg_half_drift_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_sghmc_update(float* q, float* v, const float* grad,
                                  int64_t n, float learning_rate, float friction,
                                  float noise_std, int second_order,
                                  uint64_t seed, uint32_t iteration,
                                  uint32_t latent_id, double* v2_sum,
                                  void* stream) {
  if (n == 0) return ZSHMC_OK;
  ZS_REQUIRE(q && v && grad && n > 0, "zshmc_sghmc_update: bad arguments");
  ZS_SG_LAUNCH(sghmc_kernel, al16(q) && al16(v) && al16(grad),
               sg_grid((n + 3) / 4), reinterpret_cast<hipStream_t>(stream), q, v,
               grad, n, learning_rate, friction, noise_std, second_order,
               ZS_SG_KEYS, iteration, latent_id, v2_sum);
  ZS_LAUNCH_CHECK("sghmc_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_sgnht_update(float* q, float* v, const float* grad,
                                  float* alpha_vec, const float* alpha_scalar,
                                  float* mean_k_vec, int64_t n,
                                  float learning_rate, float tune_rate,
                                  float noise_std, int second_order,
                                  uint64_t seed, uint32_t iteration,
                                  uint32_t latent_id, double* v2_sum,
                                  void* stream) {
  if (n == 0) return ZSHMC_OK;
  ZS_REQUIRE(q && v && grad && n > 0, "zshmc_sgnht_update: bad arguments");
  ZS_REQUIRE((alpha_vec != nullptr) != (alpha_scalar != nullptr),
             "zshmc_sgnht_update: exactly one of alpha_vec / alpha_scalar");
  ZS_REQUIRE(alpha_vec || v2_sum,
             "zshmc_sgnht_update: scalar friction needs v2_sum");
  ZS_SG_LAUNCH(sgnht_kernel,
               al16(q) && al16(v) && al16(grad) && al16(alpha_vec) &&
                   al16(mean_k_vec),
               sg_grid((n + 3) / 4), reinterpret_cast<hipStream_t>(stream), q, v,
               grad, alpha_vec, alpha_scalar, mean_k_vec, n, learning_rate,
               tune_rate, noise_std, second_order, ZS_SG_KEYS, iteration,
               latent_id, v2_sum);
  ZS_LAUNCH_CHECK("sgnht_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_sgnht_scalar(float* alpha_scalar, double* sums, int64_t n,
                                  float learning_rate, float tune_rate,
                                  int second_order, int phase, float* mean_k_out,
                                  void* stream) {
  ZS_REQUIRE(alpha_scalar && sums && n > 0 && (phase == 0 || phase == 1),
             "zshmc_sgnht_scalar: bad arguments");
  hipLaunchKernelGGL(sgnht_scalar_kernel, dim3(1), dim3(64), 0,
                     reinterpret_cast<hipStream_t>(stream), alpha_scalar, sums,
                     (double)n, learning_rate, tune_rate, second_order, phase,
                     mean_k_out);
  ZS_LAUNCH_CHECK("sgnht_scalar_kernel launch");
  return ZSHMC_OK;
}
