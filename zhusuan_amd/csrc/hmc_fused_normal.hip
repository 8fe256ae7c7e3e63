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
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "philox.h"
#include "fused_args.h"

namespace zshmc {


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
//
// Register budget (the kernel is latency-bound below ~4 waves/SIMD): per lane
// r, p, the prefetched next row and ONE parameter array (nep = -eps*prec; with
// mass also eim = eps/mass) = 4..5 x 4*NCH floats.  mean and sqrt(mass) are
// touched only when a row is loaded / stored / its momentum scaled, so they
// are staged in LDS once per workgroup ("LDS-staged parameter tile") and read
// back with conflict-free ds_read_b128.  prec itself is never kept:
//   sum prec*r^2 = (-1/eps) * sum nep*r^2   (one scalar multiply per chain).
#ifndef ZS_WAVES_PER_EU
#define ZS_WAVES_PER_EU 1  // A/B knob: minimum waves per SIMD asked of hipcc
#endif
template <int G, int NCH, bool VEC, bool HAS_MASS>
__global__ __launch_bounds__(256, ZS_WAVES_PER_EU) void hmc_diag_normal_kernel(
    FusedArgs a) {
  constexpr int kChainsPerWave = kWave / G;
  constexpr int kPad = G * NCH * 4;  // padded row length held by one group
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* __restrict__ s_mean = reinterpret_cast<float*>(smem);
  float* __restrict__ s_sqrtm = s_mean + kPad;  // only if HAS_MASS
  double* __restrict__ s_acc =
      reinterpret_cast<double*>(s_mean + (HAS_MASS ? 2 : 1) * kPad);
  int* __restrict__ s_bad = reinterpret_cast<int*>(s_acc + 4);

#ifdef ZS_TIMING
  const unsigned long long t_start = wall_clock64();
  unsigned long long n_done = 0;
#endif
  const int lane = threadIdx.x & (kWave - 1);
  const int l = lane % G;    // lane within the chain group
  const int sub = lane / G;  // which chain of this wave
  const int64_t wave_id =
      (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / kWave);
  const int64_t D = a.n_data;
  const uint32_t iteration = __builtin_amdgcn_readfirstlane(
      link_iteration(a.link, a.iteration));

  // ---- stage mean / sqrt(mass) in LDS (zero padding beyond n_data) --------
  for (int d = threadIdx.x; d < kPad; d += blockDim.x) {
    s_mean[d] = (a.mean && d < D) ? a.mean[d] : 0.f;
    if (HAS_MASS) s_sqrtm[d] = d < D ? sqrtf(a.mass[d]) : 0.f;
  }
  if (threadIdx.x == 0) *s_bad = 0;

  // wave-uniform step size; se is the scale folded into nep / eim
  // (the pending dual-averaging update of the previous transition is
  // applied here, see fused_args.h)
  const float eps = link_step_size(a.link, a.step_size_host);
  const bool moving = eps != 0.f;
  const float se = moving ? eps : 1.f;
  const float inv_se = 1.0f / se;

  // ---- per-latent parameters kept in registers ---------------------------
  f4 nep[NCH];   // -se * exp(-2*logstd)   (precision: univariate.py:178)
  f4 eim[NCH];   //  se / mass             (only if HAS_MASS)
  float logz_part = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int64_t d0 = (int64_t)(k * G + l) * 4;
    const f4 ls = load4<VEC>(a.logstd, d0, D, 0.f);
    f4 m = f4{1.f, 1.f, 1.f, 1.f};
    if (HAS_MASS) m = load4<VEC>(a.mass, d0, D, 1.f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool valid = d0 + j < D;
      // padding gets precision 0 and inverse mass 0: no contribution anywhere
      nep[k][j] = valid ? -se * expf(-2.0f * ls[j]) : 0.f;
      logz_part += valid ? (kHalfLog2PiNeg - ls[j]) : 0.f;
      if (HAS_MASS) eim[k][j] = valid ? se / m[j] : 0.f;
    }
  }
  const float logz = group_sum<G>(logz_part);
  // eps == 0 (degenerate but legal): nothing moves, acceptance is 1
  const int Lr = moving ? a.n_leapfrogs : 0;
  const float hk = moving ? 0.5f : 0.f;          // first half kick
  const float hk2 = Lr >= 1 ? 0.5f : 0.f;        // taken back from the last
  __syncthreads();  // LDS tile ready

  double acc_local = 0.0;
  bool bad_old = false;

  // ---- software pipeline: the next row is in flight while this one runs ---
  // Row indices are clamped instead of branched on, so the prefetch is
  // unconditional (no phi copies of the 4*NCH row registers); the one
  // redundant load per wave at the end of its range re-reads a row this wave
  // has just touched (an L2 hit, not HBM traffic).
  const int64_t last_row = a.n_chains - 1;
  auto row_of = [&](int64_t b) -> int64_t {
    int64_t c = b + sub;
    c = c < last_row ? c : last_row;
    if (G == kWave) {  // one chain per wave: keep the row index scalar
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)c);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(c >> 32));
      c = (int64_t)(((uint64_t)hi << 32) | lo);
    }
    return c;
  };
  const int loff = l * 4;  // this lane's element offset inside a G*4 chunk
  f4 qn[NCH];
  {
    const float* __restrict__ row0 =
        a.q + row_of(wave_id * kChainsPerWave) * D + loff;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
      qn[k] = load4<VEC>(row0, k * G * 4, D - loff, 0.f);
  }

  for (int64_t base = wave_id * kChainsPerWave; base < a.n_chains;
       base += n_waves * kChainsPerWave) {
    const int64_t chain = base + sub;
    const bool active = chain < a.n_chains;
    const int64_t chain_c = row_of(base);
    float* __restrict__ qrow = a.q + chain_c * D + loff;
    const uint32_t gchain = (uint32_t)(chain_c + a.chain_offset);

    // ---- r = q - mean; issue the prefetch of the next row ----------------
    f4 r[NCH], p[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const f4 mu = *reinterpret_cast<const f4*>(s_mean + (k * G + l) * 4);
      r[k] = qn[k] - mu;
    }
    {
      const float* __restrict__ nrow =
          a.q + row_of(base + n_waves * kChainsPerWave) * D + loff;
#pragma unroll
      for (int k = 0; k < NCH; ++k)
        qn[k] = load4<VEC>(nrow, k * G * 4, D - loff, 0.f);
    }
    // keep the LDS reads of `mean` here and at the store from being merged
    // (which would pin 4*NCH more registers across the whole trajectory)
    asm volatile("" ::: "memory");

    // ---- momentum resample (hmc.py:21-23, :458), initial energies and the
    // first half kick (trip i = 0 of hmc.py:352-364: zero drift, eps/2 kick),
    // fused chunk by chunk.  grad log p = -prec * r (d/dx of
    // univariate.py:181); with nep = -eps*prec a kick of s2 is
    // p += (s2/eps) * (nep * r), a drift is r += eps * p / m.
    // The Philox key is made opaque per chain so that the 20 round keys are
    // recomputed on the scalar unit instead of being pinned in (spilled)
    // SGPRs for the whole kernel.
    uint32_t key0 = a.k0, key1 = a.k1;
    asm volatile("" : "+s"(key0), "+s"(key1));
    f4 ko = f4{0.f, 0.f, 0.f, 0.f}, uo = ko;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const uint32_t group = (uint32_t)(k * G + l);
      float z0, z1, z2, z3;
      normal4(group, gchain, iteration, kStreamMomentum, key0, key1, z0, z1,
              z2, z3);
      p[k] = f4{z0, z1, z2, z3};
      if (HAS_MASS) {
        p[k] = p[k] * *reinterpret_cast<const f4*>(s_sqrtm + (k * G + l) * 4);
        ko += (p[k] * p[k]) * eim[k];
      } else {
        if (!VEC || kPad != D) {  // zero the padding lanes' momentum
          const int64_t d0 = (int64_t)group * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (d0 + j >= D) p[k][j] = 0.f;
        }
        ko += p[k] * p[k];
      }
      const f4 t = nep[k] * r[k];
      uo += t * r[k];
      p[k] += hk * t;
    }

    // ---- leapfrog (hmc.py:348-372) ----------------------------------------
    // Trips 1..L each do a full drift and a FULL kick; the last trip's kick
    // must be eps/2, so half of it is taken back below (hk2), sharing the
    // product nep*r with the potential energy of the proposal.
    if (Lr > 0) {
      int i = Lr;
      do {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          if (HAS_MASS)
            r[k] += eim[k] * p[k];
          else
            r[k] += eps * p[k];
          p[k] += nep[k] * r[k];
        }
      } while (--i > 0);
    }

    // ---- Hamiltonians (hmc.py:30-35) and acceptance (hmc.py:46-61) -------
    f4 kn = f4{0.f, 0.f, 0.f, 0.f}, un = kn;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const f4 t = nep[k] * r[k];
      un += t * r[k];
      p[k] -= hk2 * t;
      if (HAS_MASS)
        kn += (p[k] * p[k]) * eim[k];
      else
        kn += p[k] * p[k];
    }
    float k_old = (ko[0] + ko[1]) + (ko[2] + ko[3]);
    float u_old = (uo[0] + uo[1]) + (uo[2] + uo[3]);
    float k_new = (kn[0] + kn[1]) + (kn[2] + kn[3]);
    float u_new = (un[0] + un[1]) + (un[2] + un[3]);
    k_old = group_sum<G>(k_old);
    u_old = group_sum<G>(u_old);
    k_new = group_sum<G>(k_new);
    u_new = group_sum<G>(u_new);
    if (HAS_MASS) {  // sum p^2/m = (1/se) sum p^2 * (se/m)
      k_old *= inv_se;
      k_new *= inv_se;
    }
    // sum prec r^2 = -(1/se) sum nep r^2 ;  log p = logz - 0.5 * that
    const float lp_old = logz + 0.5f * inv_se * u_old;
    const float lp_new = logz + 0.5f * inv_se * u_new;
    const float h_old = -lp_old + 0.5f * k_old;
    const float h_new = -lp_new + 0.5f * k_new;
    const float dh = h_old - h_new;
    float acc = expf(fminf(dh, 0.0f));
    // fminf drops a NaN operand: test explicitly (hmc.py:56-59)
    if (!(dh == dh) || !isfinite(acc) || !isfinite(lp_new)) acc = 0.f;
    if (active && !isfinite(lp_old)) bad_old = true;

    const float u = uniform_chain(gchain, iteration, key0, key1);
    const bool accept = u < acc;  // strict, hmc.py:486

#ifdef ZS_TIMING
    ++n_done;
#endif
    if (active && l == 0) acc_local += (double)acc;
    if (a.commit && active) {
      if (accept) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          const f4 mu = *reinterpret_cast<const f4*>(s_mean + (k * G + l) * 4);
          store4<VEC>(qrow, k * G * 4, D - loff, r[k] + mu);
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

#ifdef ZS_TIMING
  if (a.timing && lane == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* t = a.timing + wave_id * 4;
    t[0] = t_start;
    t[1] = wall_clock64();
    t[2] = xcc & 0xf;
    t[3] = n_done;
  }
#endif
  // ---- sum of acceptance rates: wave shuffle -> LDS -> per-workgroup
  // partial; the order-fixed total over workgroups is link_retire's (the
  // chain -> wave map is static here, so every level of the sum is too)
  const double w = wave_sum_f64(acc_local);
  if (lane == 0) s_acc[threadIdx.x / kWave] = w;
  if (bad_old) *s_bad = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < (int)(blockDim.x / kWave); ++i) tot += s_acc[i];
    if (*s_bad && a.flags) atomicOr(a.flags, ZSHMC_FLAG_OLD_LOGPROB_NONFINITE);
    link_retire(a.link, tot, a.flags);
  }
}

template <int G, int NCH>
static int launch_cfg(const FusedArgs& a, hipStream_t stream) {
  const bool vec = (a.n_data % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(a.q) & 15) == 0) &&
                   (!a.mean || (reinterpret_cast<uintptr_t>(a.mean) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(a.logstd) & 15) == 0) &&
                   (!a.mass || (reinterpret_cast<uintptr_t>(a.mass) & 15) == 0);
  const bool has_mass = a.mass != nullptr;
  constexpr int kChainsPerWave = kWave / G;
  constexpr int kWavesPerBlock = 4;
  const int64_t chains_per_block = (int64_t)kChainsPerWave * kWavesPerBlock;
  const int64_t need = (a.n_chains + chains_per_block - 1) / chains_per_block;
  // dynamic LDS: mean tile (+ sqrt(mass) tile) + 4 doubles + 1 int, all
  // carved from the 16-B aligned dynamic region (no static __shared__)
  const size_t lds = (size_t)(has_mass ? 2 : 1) * G * NCH * 4 * sizeof(float) +
                     4 * sizeof(double) + 16;
  // persistent grid = exactly the resident blocks (register-limited
  // occupancy x CUs), so every wave amortises its parameter prologue over
  // as many chains as possible
  static int blocks_per_cu[4] = {0, 0, 0, 0};
  const int variant = (vec ? 2 : 0) + (has_mass ? 1 : 0);
  if (blocks_per_cu[variant] == 0) {
    int nb = 0;
    hipError_t e = hipSuccess;
    if (vec && has_mass)
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
          &nb, hmc_diag_normal_kernel<G, NCH, true, true>, 256, lds);
    else if (vec)
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
          &nb, hmc_diag_normal_kernel<G, NCH, true, false>, 256, lds);
    else if (has_mass)
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
          &nb, hmc_diag_normal_kernel<G, NCH, false, true>, 256, lds);
    else
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
          &nb, hmc_diag_normal_kernel<G, NCH, false, false>, 256, lds);
    if (e != hipSuccess || nb < 1) nb = 2;
    if (nb > 8) nb = 8;
    blocks_per_cu[variant] = nb;
  }
  const int64_t cap = (int64_t)device_cu_count() * blocks_per_cu[variant];
  int grid = (int)(need < cap ? need : cap);
  if (grid > kFusedMaxGrid) grid = kFusedMaxGrid;  // link workspace (persistent
                                                   // waves stride over chains)
  dim3 g(grid > 0 ? grid : 1), b(kWave * kWavesPerBlock);
  if (vec && has_mass)
    hipLaunchKernelGGL((hmc_diag_normal_kernel<G, NCH, true, true>), g, b, lds,
                       stream, a);
  else if (vec)
    hipLaunchKernelGGL((hmc_diag_normal_kernel<G, NCH, true, false>), g, b,
                       lds, stream, a);
  else if (has_mass)
    hipLaunchKernelGGL((hmc_diag_normal_kernel<G, NCH, false, true>), g, b,
                       lds, stream, a);
  else
    hipLaunchKernelGGL((hmc_diag_normal_kernel<G, NCH, false, false>), g, b,
                       lds, stream, a);
  ZS_LAUNCH_CHECK("hmc_diag_normal_kernel launch");
  return ZSHMC_OK;
}

constexpr int64_t kFusedMaxData = 64 * 8 * 4;  // G=64, NCH=8

}  // namespace zshmc

using namespace zshmc;

extern "C" int64_t zshmc_fused_max_n_data(void) { return kFusedMaxData; }

extern "C" int64_t zshmc_fused_colstats_rows(int64_t n_chains, int64_t n_data,
                                             int has_mass, int zero_mean) {
  if (n_chains <= 0 ||
      !fused_ring_colstats(n_data, has_mass != 0, zero_mean != 0))
    return 0;
  return fused_ring_grid(n_chains);
}

extern "C" const char* zshmc_fused_kernel_name(int64_t n_data, int has_mass,
                                               int zero_mean) {
  static thread_local char buf[96];
  int nch = 0, k = 0;
  if (fused_ring_enabled() &&
      fused_ring_config(n_data, has_mass != 0, zero_mean != 0, &nch, &k)) {
    snprintf(buf, sizeof(buf), "hmc_diag_normal_ring_kernel<%d,%d,%s,*,%s>",
             nch, k, has_mass ? "true" : "false",
             zero_mean ? "true" : "false");
    return buf;
  }
  const int64_t ng = (n_data + 3) / 4;
  int g = 64;
  nch = 1;
  if (ng <= 32) {
    g = 1;
    while (g < ng) g *= 2;
  } else if (ng > 64) {
    const int steps[] = {2, 3, 4, 6, 8};
    nch = 8;
    for (int st : steps)
      if (ng <= 64 * st) {
        nch = st;
        break;
      }
  }
  snprintf(buf, sizeof(buf), "hmc_diag_normal_kernel<%d,%d,%s,%s>", g, nch,
           n_data % 4 == 0 ? "true" : "false", has_mass ? "true" : "false");
  return buf;
}

namespace zshmc {

// host-side zshmc_adapt_link -> the device-side view
static int make_link(const zshmc_adapt_link* link, AdaptLink* out,
                     const char* who) {
  AdaptLink k;
  memset(&k, 0, sizeof(k));
  k.used_step_size = __builtin_nanf("");
  if (link) {
    ZS_REQUIRE(link->pending == ZSHMC_PEND_NONE ||
                   link->pending == ZSHMC_PEND_ADAPT ||
                   link->pending == ZSHMC_PEND_HOLD,
               "%s: link->pending %d is not a ZSHMC_PEND_* value", who,
               (int)link->pending);
    ZS_REQUIRE(link->retire_update == ZSHMC_PEND_NONE ||
                   link->retire_update == ZSHMC_PEND_ADAPT ||
                   link->retire_update == ZSHMC_PEND_HOLD,
               "%s: link->retire_update %d is not a ZSHMC_PEND_* value", who,
               (int)link->retire_update);
    ZS_REQUIRE(link->pending == ZSHMC_PEND_NONE ||
                   link->retire_update == ZSHMC_PEND_NONE,
               "%s: pending and retire_update are exclusive", who);
    ZS_REQUIRE((link->pending == ZSHMC_PEND_NONE &&
                link->retire_update == ZSHMC_PEND_NONE) ||
                   (link->state && link->stats),
               "%s: a step-size update needs link->state and link->stats", who);
    ZS_REQUIRE(!link->stats || link->workspace,
               "%s: link->stats needs link->workspace", who);
    ZS_REQUIRE(!(link->state || link->stats) || link->n_chains_global > 0,
               "%s: link->n_chains_global <= 0", who);
    k.state = link->state;
    k.stats = link->stats;
    if (link->stats)
      k.accum = reinterpret_cast<unsigned long long*>(link->workspace);
    k.inv_chains =
        link->n_chains_global > 0 ? 1.0 / (double)link->n_chains_global : 0.0;
    k.pending = link->pending;
    k.retire = link->retire_update;
    k.fresh = link->fresh_start ? 1.0f : 0.0f;
    k.used_step_size = link->used_step_size;
    k.tuner = TunerCfg{link->delta, link->gamma, link->t0, link->kappa,
                       link->mu};
    ZS_REQUIRE(!link->colstats_parts || link->colstats_mean,
               "%s: link->colstats_parts needs link->colstats_mean", who);
    k.cs_mean = link->colstats_mean;
    k.cs_parts = link->colstats_parts;
  }
  *out = k;
  return ZSHMC_OK;
}

// a pending update without a transition (also: a rank that owns no chains
// still has to retire the update and publish an empty sum)
__global__ void stepsize_flush_kernel(AdaptLink k, int zero_stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (k.state && k.pending != ZSHMC_PEND_NONE)
    tuner_persist(k, k.pending, k.stats[0]);
  // (a rank without chains: this transition's own sum is 0)
  if (zero_stats && k.state && k.retire != ZSHMC_PEND_NONE)
    tuner_persist(k, k.retire, 0.0);
  if (zero_stats && k.stats) {
    k.stats[0] = 0.0;
    k.stats[1] = 0.0;
  }
}

}  // namespace zshmc

extern "C" int zshmc_stepsize_flush(const zshmc_adapt_link* link,
                                    void* stream) {
  ZS_REQUIRE(link, "zshmc_stepsize_flush: null link");
  AdaptLink k;
  const int rc = make_link(link, &k, "zshmc_stepsize_flush");
  if (rc != ZSHMC_OK) return rc;
  if (k.pending == ZSHMC_PEND_NONE) return ZSHMC_OK;
  hipLaunchKernelGGL(stepsize_flush_kernel, dim3(1), dim3(64), 0,
                     reinterpret_cast<hipStream_t>(stream), k, 0);
  ZS_LAUNCH_CHECK("stepsize_flush_kernel launch");
  return ZSHMC_OK;
}

// iter_from_device: the launch is (being captured as) a node of a hipGraph --
// its iteration is `iteration` + the device counter at workspace + 16, which
// the workgroup that retires last advances (so the retirement atomic runs
// even when no statistics are collected)
static int step_impl(
    float* q, const float* mean, const float* logstd, const float* mass,
    float step_size_host, int64_t n_chains, int64_t n_data,
    int64_t chain_offset, int n_leapfrogs, uint64_t seed, uint32_t iteration,
    int commit, float* acceptance_rate, float* orig_hamiltonian,
    float* hamiltonian, float* orig_log_prob, float* log_prob, uint32_t* flags,
    const zshmc_adapt_link* link, void* stream, int iter_from_device) {
  ZS_REQUIRE(q && logstd, "zshmc_hmc_diag_normal_step: null q/logstd");
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
  FusedArgs a;
  {
    const int rc = make_link(link, &a.link, "zshmc_hmc_diag_normal_step");
    if (rc != ZSHMC_OK) return rc;
  }
  if (iter_from_device) {
    ZS_REQUIRE(link && link->workspace && n_chains > 0,
               "zshmc_hmc_diag_normal_run: graph replay needs link->workspace");
    char* ws = reinterpret_cast<char*>(link->workspace);
    a.link.accum = reinterpret_cast<unsigned long long*>(ws);
    a.link.iter_dev = reinterpret_cast<uint32_t*>(ws + 16);
  }
  if (n_chains == 0) {
    if (a.link.stats || a.link.pending != ZSHMC_PEND_NONE ||
        a.link.retire != ZSHMC_PEND_NONE) {
      hipLaunchKernelGGL(stepsize_flush_kernel, dim3(1), dim3(64), 0,
                         reinterpret_cast<hipStream_t>(stream), a.link, 1);
      ZS_LAUNCH_CHECK("stepsize_flush_kernel launch");
    }
    return ZSHMC_OK;
  }
  a.q = q;
  a.mean = mean;
  a.logstd = logstd;
  a.mass = mass;
  a.step_size_host = step_size_host;
  a.n_chains = n_chains;
  a.n_data = n_data;
  a.chain_offset = chain_offset;
  a.n_leapfrogs = n_leapfrogs;
  a.k0 = (uint32_t)(seed & 0xFFFFFFFFull);
  a.k1 = (uint32_t)(seed >> 32);
  a.iteration = iteration;
  a.commit = commit ? 1 : 0;
  a.acceptance_rate = acceptance_rate;
  a.orig_hamiltonian = orig_hamiltonian;
  a.hamiltonian = hamiltonian;
  a.orig_log_prob = orig_log_prob;
  a.log_prob = log_prob;
  a.flags = flags;
  a.info_cap = 0;
  a.commit_direct = 0;
  {
    // fixed-point scale of the acceptance sum: n_chains * 2^shift < 2^kSumBits
    int shift = 40;
    while (shift > 0 && (double)(n_chains + 1) * (double)(1ull << shift) >=
                            (double)(1ull << kSumBits))
      --shift;
    a.link.fx_scale = (double)(1ull << shift);
    a.link.fx_inv_scale = 1.0 / a.link.fx_scale;
  }
#ifdef ZS_TIMING
  a.timing = reinterpret_cast<unsigned long long*>(orig_hamiltonian);  // debug
  a.orig_hamiltonian = nullptr;
#endif
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // a dry run (the step-size search) does not move the state: no statistics
  if (!a.commit) a.link.cs_parts = nullptr;
  // rows of more than 128 latents, 16-B aligned: the LDS-DMA ring kernel
  // (ZSHMC_FUSED_RING=0 keeps the register-prefetch kernel, for A/B runs)
  if (fused_ring_enabled()) {
    const int rc = launch_fused_ring(a, s);
    if (rc != ZSHMC_ERR_UNSUPPORTED) return rc;
  }
  ZS_REQUIRE(!a.link.cs_parts,
             "zshmc_hmc_diag_normal_step: link->colstats_parts given, but the "
             "kernel of this shape / alignment does not produce column "
             "statistics (zshmc_fused_colstats_rows)");
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

extern "C" int zshmc_hmc_diag_normal_step(
    float* q, const float* mean, const float* logstd, const float* mass,
    float step_size_host, int64_t n_chains, int64_t n_data,
    int64_t chain_offset, int n_leapfrogs, uint64_t seed, uint32_t iteration,
    int commit, float* acceptance_rate, float* orig_hamiltonian,
    float* hamiltonian, float* orig_log_prob, float* log_prob, uint32_t* flags,
    const zshmc_adapt_link* link, void* stream) {
  return step_impl(q, mean, logstd, mass, step_size_host, n_chains, n_data,
                   chain_offset, n_leapfrogs, seed, iteration, commit,
                   acceptance_rate, orig_hamiltonian, hamiltonian,
                   orig_log_prob, log_prob, flags, link, stream, 0);
}

// K transitions from one call: the launch loop runs on THIS side of the C-ABI
// (a host language pays its per-call overhead once per run, not once per
// transition; at BASELINE configs[0]'s size a transition is a few
// microseconds of device time).  (Replaying stretches of the loop from a
// hipGraph was built and measured in round 3 and is gone: on ROCm 7.2 a
// replayed kernel node costs MORE than a plain launch into a busy queue --
// 5.85 against 4.39 us per transition at 1 000 x 10, 99.2 against 95.3 at
// 65 536 x 1 024, profiles/archive/r03l_run_graph.txt.)
extern "C" int zshmc_hmc_diag_normal_run(
    float* q, const float* mean, const float* logstd, const float* mass,
    float step_size_host, int64_t n_chains, int64_t n_data,
    int64_t chain_offset, int n_leapfrogs, uint64_t seed,
    uint32_t iteration_first, int n_transitions, float* acceptance_rate,
    float* orig_hamiltonian, float* hamiltonian, float* orig_log_prob,
    float* log_prob, uint32_t* flags, const zshmc_adapt_link* link, void* comm,
    void* stream) {
  ZS_REQUIRE(n_transitions >= 0, "zshmc_hmc_diag_normal_run: n_transitions < 0");
  ZS_REQUIRE(link, "zshmc_hmc_diag_normal_run: null link");
  ZS_REQUIRE(!link->colstats_parts,
             "zshmc_hmc_diag_normal_run: column statistics belong to the "
             "mass-adapting iterations, which run one zshmc_hmc_diag_normal_"
             "step at a time");
  ZS_REQUIRE(!comm || link->retire_update == ZSHMC_PEND_NONE || link->stats,
             "zshmc_hmc_diag_normal_run: an update across ranks needs stats");
  const int kind = link->retire_update;
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
  int done = 0;
  // one transition as its own launch (i = index within the run)
  auto single = [&](int i) -> int {
    zshmc_adapt_link l = *link;
    if (i > 0) {
      l.fresh_start = 0;
      l.used_step_size = __builtin_nanf("");
    }
    if (comm) {
      // sharded chains: this transition's update is applied by the NEXT
      // launch's prologue, from the all-reduced sum
      l.retire_update = ZSHMC_PEND_NONE;
      l.pending = i == 0 ? link->pending : kind;
    } else if (i > 0) {
      l.pending = ZSHMC_PEND_NONE;
    }
    int rc = step_impl(q, mean, logstd, mass, step_size_host, n_chains, n_data,
                       chain_offset, n_leapfrogs, seed,
                       iteration_first + (uint32_t)i, 1, acceptance_rate,
                       orig_hamiltonian, hamiltonian, orig_log_prob, log_prob,
                       flags, &l, stream, 0);
    if (rc != ZSHMC_OK) return rc;
    if (comm && l.stats)
      rc = zshmc_comm_all_reduce_sum(comm, l.stats, ZSHMC_STATS_WORDS, stream);
    return rc;
  };

  for (int i = done; i < n_transitions; ++i) {
    const int rc = single(i);
    if (rc != ZSHMC_OK) return rc;
  }
  return ZSHMC_OK;
}
