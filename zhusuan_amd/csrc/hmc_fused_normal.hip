// Fused HMC transition for a diagonal-Normal log-joint on gfx950 (MI355X).
//
// One launch performs, for every chain, what one execution of the
// reference's `sample_op` does through ~15-20 un-fused TensorFlow passes per
// leapfrog step (zhusuan/hmc.py:382-522): momentum resample (:21-23), the
// (L+1)-trip leapfrog loop (:348-372, :38-43) with the analytic gradient of
// Normal._log_prob (distributions/univariate.py:174-181) reduced over the
// data axes (distributions/base.py:302-304), both Hamiltonians (:30-35), the
// acceptance rate (:46-61) and the MH select + in-place assign (:479-498).
//
// Data layout: q is chain-major [n_chains, n_data] (the API layout,
// hmc.py:209-216).  G lanes cooperate on one chain ("lanes across
// latents"): lane l of the group owns the 4-element chunks
// (k*G + l)*4 .. +3, k = 0..NCH-1, so every global access is a fully
// coalesced 16 B/lane row segment, the per-latent parameters live in
// registers for the whole (persistent) life of the wave, the trajectory never
// leaves registers, and the per-chain energy sums are log2(G) shuffle steps.
// G = 64 (one wave per chain) for n_data > 128; smaller power-of-two groups
// pack 64/G chains into a wave for small n_data.
//
// HBM traffic per chain: read q (4*D B) + write q (4*D B, only if accepted)
// + 5 floats of HMCInfo: 8 B per element per transition.
#include "common.h"
#include "philox.h"

namespace zshmc {

constexpr float kHalfLog2PiNeg = -0.91893853320467274178f;  // -0.5*log(2*pi)

struct FusedArgs {
  float* q;
  const float* mean;
  const float* logstd;
  const float* mass;
  const float* step_size_dev;
  float step_size_host;
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
  double* acc_sum;
  uint32_t* flags;
};

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool VEC>
__device__ __forceinline__ f4 load4(const float* __restrict__ base, int64_t d0,
                                    int64_t n_data, float fill) {
  if (VEC) {
    if (d0 < n_data) return *reinterpret_cast<const f4*>(base + d0);
    return f4{fill, fill, fill, fill};
  } else {
    f4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (d0 + j < n_data) ? base[d0 + j] : fill;
    return v;
  }
}

template <bool VEC>
__device__ __forceinline__ void store4(float* __restrict__ base, int64_t d0,
                                       int64_t n_data, f4 v) {
  if (VEC) {
    if (d0 < n_data) *reinterpret_cast<f4*>(base + d0) = v;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (d0 + j < n_data) base[d0 + j] = v[j];
  }
}

// G lanes per chain, NCH 4-element chunks per lane.
template <int G, int NCH, bool VEC, bool HAS_MASS>
__global__ __launch_bounds__(256) void hmc_diag_normal_kernel(FusedArgs a) {
  constexpr int kChainsPerWave = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int l = lane % G;    // lane within the chain group
  const int sub = lane / G;  // which chain of this wave
  const int64_t wave_id =
      (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / kWave);
  const int64_t D = a.n_data;

  // ---- per-latent parameters: registers, loaded once per wave -----------
  f4 mean[NCH], prec[NCH], inv_m[NCH], sqrt_m[NCH];
  float logz_part = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int64_t d0 = (int64_t)(k * G + l) * 4;
    mean[k] = load4<VEC>(a.mean, d0, D, 0.f);
    const f4 ls = load4<VEC>(a.logstd, d0, D, 0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool valid = d0 + j < D;
      // precision = exp(-2*logstd) (univariate.py:178); 0 for padding so
      // padded elements contribute nothing to any sum.
      prec[k][j] = valid ? __expf(-2.0f * ls[j]) : 0.f;
      logz_part += valid ? (kHalfLog2PiNeg - ls[j]) : 0.f;
    }
    if (HAS_MASS) {
      const f4 m = load4<VEC>(a.mass, d0, D, 1.f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool valid = d0 + j < D;
        inv_m[k][j] = valid ? 1.0f / m[j] : 0.f;
        sqrt_m[k][j] = valid ? sqrtf(m[j]) : 0.f;
      }
    }
  }
  const float logz = group_sum<G>(logz_part);
  const float eps =
      a.step_size_dev ? *a.step_size_dev : a.step_size_host;  // wave-uniform
  const int L = a.n_leapfrogs;

  double acc_local = 0.0;
  bool bad_old = false;

  for (int64_t base = wave_id * kChainsPerWave; base < a.n_chains;
       base += n_waves * kChainsPerWave) {
    const int64_t chain = base + sub;
    const bool active = chain < a.n_chains;
    const int64_t chain_c = active ? chain : a.n_chains - 1;  // clamp: loads
    float* __restrict__ qrow = a.q + chain_c * D;
    const uint32_t gchain = (uint32_t)(chain_c + a.chain_offset);

    // ---- load q, resample momentum (hmc.py:21-23, :458) ----------------
    f4 r[NCH], p[NCH];  // r = q - mean
    float k_old = 0.f, u_old = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int64_t d0 = (int64_t)(k * G + l) * 4;
      const f4 qv = load4<VEC>(qrow, d0, D, 0.f);
      r[k] = qv - mean[k];
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const uint32_t group = (uint32_t)(k * G + l);
      float z0, z1, z2, z3;
      normal4(group, gchain, a.iteration, kStreamMomentum, a.k0, a.k1, z0, z1,
              z2, z3);
      p[k] = f4{z0, z1, z2, z3};
      if (HAS_MASS) {
        p[k] = p[k] * sqrt_m[k];
        const f4 pp = p[k] * p[k] * inv_m[k];
        k_old += (pp[0] + pp[1]) + (pp[2] + pp[3]);
      } else {
        // padded elements must not contribute: prec==0 marks padding only
        // when VEC is false or D is not a multiple of 4*G*NCH
        const int64_t d0 = (int64_t)group * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (d0 + j >= D) p[k][j] = 0.f;
        const f4 pp = p[k] * p[k];
        k_old += (pp[0] + pp[1]) + (pp[2] + pp[3]);
      }
      const f4 uu = prec[k] * r[k] * r[k];
      u_old += (uu[0] + uu[1]) + (uu[2] + uu[3]);
    }

    // ---- leapfrog (hmc.py:348-372): L+1 kicks, L drifts ------------------
    // grad log p = -prec * r  (d/dx of univariate.py:181).
    // kick:  p += s2 * (-prec * r) ;  drift: r += eps * p / m
    const float half = 0.5f * eps;
#pragma unroll
    for (int k = 0; k < NCH; ++k) p[k] -= (half * prec[k]) * r[k];
    for (int i = 1; i <= L; ++i) {
      const float s2 = (i < L) ? eps : half;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (HAS_MASS)
          r[k] += (eps * inv_m[k]) * p[k];
        else
          r[k] += eps * p[k];
        p[k] -= (s2 * prec[k]) * r[k];
      }
    }

    // ---- Hamiltonians (hmc.py:30-35) and acceptance (hmc.py:46-61) -------
    float k_new = 0.f, u_new = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      f4 pp = p[k] * p[k];
      if (HAS_MASS) pp = pp * inv_m[k];
      k_new += (pp[0] + pp[1]) + (pp[2] + pp[3]);
      const f4 uu = prec[k] * r[k] * r[k];
      u_new += (uu[0] + uu[1]) + (uu[2] + uu[3]);
    }
    k_old = group_sum<G>(k_old);
    u_old = group_sum<G>(u_old);
    k_new = group_sum<G>(k_new);
    u_new = group_sum<G>(u_new);
    const float lp_old = logz - 0.5f * u_old;
    const float lp_new = logz - 0.5f * u_new;
    const float h_old = -lp_old + 0.5f * k_old;
    const float h_new = -lp_new + 0.5f * k_new;
    float acc = __expf(fminf(h_old - h_new, 0.0f));
    // fminf drops a NaN operand; test the operands explicitly (hmc.py:56-59)
    const bool finite = isfinite(h_old - h_new) || (h_old - h_new) == INFINITY;
    if (!(finite && isfinite(acc) && isfinite(lp_new))) acc = 0.f;
    if (active && !isfinite(lp_old)) bad_old = true;

    const float u = uniform_chain(gchain, a.iteration, a.k0, a.k1);
    const bool accept = u < acc;  // strict, hmc.py:486

    if (active && l == 0) acc_local += (double)acc;
    if (a.commit && active) {
      if (accept) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          const int64_t d0 = (int64_t)(k * G + l) * 4;
          store4<VEC>(qrow, d0, D, r[k] + mean[k]);
        }
      }
      if (l == 0) {
        if (a.acceptance_rate) a.acceptance_rate[chain] = acc;
        if (a.orig_hamiltonian) a.orig_hamiltonian[chain] = h_old;
        if (a.hamiltonian) a.hamiltonian[chain] = h_new;
        if (a.orig_log_prob) a.orig_log_prob[chain] = lp_old;
        if (a.log_prob) a.log_prob[chain] = accept ? lp_new : lp_old;
      }
    }
  }

  // ---- sum of acceptance rates: wave shuffle -> LDS -> one atomic/block --
  __shared__ double s_acc[4];
  __shared__ int s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  const double w = wave_sum_f64(acc_local);
  __syncthreads();
  if (lane == 0) s_acc[threadIdx.x / kWave] = w;
  if (bad_old) s_bad = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < (int)(blockDim.x / kWave); ++i) tot += s_acc[i];
    if (a.acc_sum) atomicAdd(a.acc_sum, tot);
    if (s_bad && a.flags) atomicOr(a.flags, ZSHMC_FLAG_OLD_LOGPROB_NONFINITE);
  }
}

template <int G, int NCH>
static int launch_cfg(const FusedArgs& a, hipStream_t stream) {
  const bool vec = (a.n_data % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(a.q) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(a.mean) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(a.logstd) & 15) == 0) &&
                   (!a.mass || (reinterpret_cast<uintptr_t>(a.mass) & 15) == 0);
  const bool has_mass = a.mass != nullptr;
  constexpr int kChainsPerWave = kWave / G;
  constexpr int kWavesPerBlock = 4;
  const int64_t chains_per_block = (int64_t)kChainsPerWave * kWavesPerBlock;
  const int64_t need = (a.n_chains + chains_per_block - 1) / chains_per_block;
  // persistent grid: 8 blocks of 4 waves per CU fill the 32 wave slots
  const int64_t cap = (int64_t)device_cu_count() * 8;
  const int grid = (int)(need < cap ? need : cap);
  dim3 g(grid > 0 ? grid : 1), b(kWave * kWavesPerBlock);
  if (vec && has_mass)
    hipLaunchKernelGGL((hmc_diag_normal_kernel<G, NCH, true, true>), g, b, 0,
                       stream, a);
  else if (vec)
    hipLaunchKernelGGL((hmc_diag_normal_kernel<G, NCH, true, false>), g, b, 0,
                       stream, a);
  else if (has_mass)
    hipLaunchKernelGGL((hmc_diag_normal_kernel<G, NCH, false, true>), g, b, 0,
                       stream, a);
  else
    hipLaunchKernelGGL((hmc_diag_normal_kernel<G, NCH, false, false>), g, b, 0,
                       stream, a);
  ZS_LAUNCH_CHECK("hmc_diag_normal_kernel launch");
  return ZSHMC_OK;
}

constexpr int64_t kFusedMaxData = 64 * 8 * 4;  // G=64, NCH=8

}  // namespace zshmc

using namespace zshmc;

extern "C" int64_t zshmc_fused_max_n_data(void) { return kFusedMaxData; }

extern "C" int zshmc_hmc_diag_normal_step(
    float* q, const float* mean, const float* logstd, const float* mass,
    const float* step_size_dev, float step_size_host, int64_t n_chains,
    int64_t n_data, int64_t chain_offset, int n_leapfrogs, uint64_t seed,
    uint32_t iteration, int commit, float* acceptance_rate,
    float* orig_hamiltonian, float* hamiltonian, float* orig_log_prob,
    float* log_prob, double* acc_sum, uint32_t* flags, void* stream) {
  ZS_REQUIRE(q && mean && logstd, "zshmc_hmc_diag_normal_step: null q/mean/logstd");
  ZS_REQUIRE(n_chains >= 0 && n_data >= 1,
             "zshmc_hmc_diag_normal_step: bad shape [%lld, %lld]",
             (long long)n_chains, (long long)n_data);
  ZS_REQUIRE(n_data <= kFusedMaxData,
             "zshmc_hmc_diag_normal_step: n_data %lld exceeds the fused "
             "kernel's limit %lld; use the generic path",
             (long long)n_data, (long long)kFusedMaxData);
  ZS_REQUIRE(n_leapfrogs >= 0, "zshmc_hmc_diag_normal_step: n_leapfrogs < 0");
  ZS_REQUIRE(n_chains + chain_offset <= 0xFFFFFFFFll,
             "zshmc_hmc_diag_normal_step: global chain index exceeds 2^32");
  if (n_chains == 0) return ZSHMC_OK;
  FusedArgs a;
  a.q = q;
  a.mean = mean;
  a.logstd = logstd;
  a.mass = mass;
  a.step_size_dev = step_size_dev;
  a.step_size_host = step_size_host;
  a.n_chains = n_chains;
  a.n_data = n_data;
  a.chain_offset = chain_offset;
  a.n_leapfrogs = n_leapfrogs;
  a.k0 = (uint32_t)(seed & 0xFFFFFFFFull);
  a.k1 = (uint32_t)(seed >> 32);
  a.iteration = iteration;
  a.commit = commit;
  a.acceptance_rate = acceptance_rate;
  a.orig_hamiltonian = orig_hamiltonian;
  a.hamiltonian = hamiltonian;
  a.orig_log_prob = orig_log_prob;
  a.log_prob = log_prob;
  a.acc_sum = acc_sum;
  a.flags = flags;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t ng = (n_data + 3) / 4;  // 4-element chunks per chain
  if (ng <= 1) return launch_cfg<1, 1>(a, s);
  if (ng <= 2) return launch_cfg<2, 1>(a, s);
  if (ng <= 4) return launch_cfg<4, 1>(a, s);
  if (ng <= 8) return launch_cfg<8, 1>(a, s);
  if (ng <= 16) return launch_cfg<16, 1>(a, s);
  if (ng <= 32) return launch_cfg<32, 1>(a, s);
  if (ng <= 64) return launch_cfg<64, 1>(a, s);
  if (ng <= 128) return launch_cfg<64, 2>(a, s);
  if (ng <= 192) return launch_cfg<64, 3>(a, s);
  if (ng <= 256) return launch_cfg<64, 4>(a, s);
  if (ng <= 384) return launch_cfg<64, 6>(a, s);
  return launch_cfg<64, 8>(a, s);
}
