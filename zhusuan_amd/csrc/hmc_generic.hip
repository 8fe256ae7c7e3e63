// Building blocks of the generic HMC transition on gfx950: the log-joint and
// its gradient come from the caller (autograd over the log_prob ops of
// distributions.hip, or any user callable), these kernels do everything else
// of reference zhususan/hmc.py's sample_op with q, p resident in HBM:
//
//   momentum_kernel     random_momentum            hmc.py:21-23
//   kick_drift_kernel   leapfrog_integrator        hmc.py:38-43 (schedule :352-364)
//                       + kinetic part of hamiltonian :32-34
//   mh_accept_kernel    get_acceptance_rate        hmc.py:46-61, MH test :479-487
//   select_rows_kernel  where(accept, q', q)       hmc.py:488-497
//
// Row layout [n_chains, n_data]; one wave per row, lanes stride over the row
// in 4-element chunks (coalesced), per-row sums by wave shuffles.  The Philox
// counter mapping is the fused kernel's, so both paths draw identical
// momenta and uniforms.
#include "common.h"
#include "philox.h"

namespace zshmc {

__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }

typedef float g4 __attribute__((ext_vector_type(4)));

// VEC: rows are 16-B aligned multiples of 4 floats -> one 16-B access per lane
template <bool VEC>
__global__ __launch_bounds__(256) void momentum_kernel(
    float* __restrict__ p, int64_t ld, const float* __restrict__ mass,
    int64_t n_chains, int64_t n_data, int64_t chain_offset, uint32_t k0,
    uint32_t k1, uint32_t iteration, uint32_t stream_word,
    float* __restrict__ kinetic) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / 64);
  const int64_t n_groups = (n_data + 3) / 4;
  for (int64_t c = wave; c < n_chains; c += n_waves) {
    const uint32_t gchain = (uint32_t)(c + chain_offset);
    float* __restrict__ row = p + c * ld;
    float kin = 0.f;
    if (VEC) {
      // two 4-latent groups per trip, their Philox calls advanced together
      // (philox.h: normal4x2): twice the independent work between dependent
      // instructions of the generator
      for (int64_t base = 0; base + 128 <= n_groups; base += 128) {
        const int64_t g = base + lane;
        float za[4], zb[4];
        normal4x2((uint32_t)g, (uint32_t)(g + 64), gchain, iteration,
                  stream_word, k0, k1, za, zb);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float* z = h ? zb : za;
          const int64_t gg = g + 64 * h;
          g4 v = g4{z[0], z[1], z[2], z[3]};
          if (mass) {
            const g4 m = *reinterpret_cast<const g4*>(mass + gg * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] *= sqrtf(m[j]);
              kin += v[j] * v[j] / m[j];
            }
          } else {
            const g4 sq = v * v;
            kin += (sq[0] + sq[1]) + (sq[2] + sq[3]);
          }
          *reinterpret_cast<g4*>(row + gg * 4) = v;
        }
      }
    }
    // (the paired loop covers whole 128-group spans; what is left -- and the
    // ragged path -- one group per trip)
    const int64_t g_start = VEC ? (n_groups / 128) * 128 + lane : lane;
    for (int64_t g = g_start; g < n_groups; g += 64) {
      float z[4];
      normal4((uint32_t)g, gchain, iteration, stream_word, k0, k1, z[0], z[1],
              z[2], z[3]);
      if (VEC) {
        g4 v = g4{z[0], z[1], z[2], z[3]};
        if (mass) {
          const g4 m = *reinterpret_cast<const g4*>(mass + g * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] *= sqrtf(m[j]);
            kin += v[j] * v[j] / m[j];
          }
        } else {
          const g4 sq = v * v;
          kin += (sq[0] + sq[1]) + (sq[2] + sq[3]);
        }
        *reinterpret_cast<g4*>(row + g * 4) = v;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t d = g * 4 + j;
          if (d < n_data) {
            const float m = mass ? mass[d] : 1.0f;
            const float v = z[j] * sqrtf(m);
            row[d] = v;
            kin += v * v / m;
          }
        }
      }
    }
    if (kinetic) {
      kin = wave_sum(kin);
      if (lane == 0) kinetic[c] += 0.5f * kin;
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void kick_drift_kernel(
    float* __restrict__ q, float* __restrict__ p,
    const float* __restrict__ grad, const float* __restrict__ mass,
    const float* __restrict__ step_size_dev, float step_size_host,
    float kick_scale, float drift_scale, int64_t n_chains, int64_t n_data,
    float* __restrict__ kinetic) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / 64);
  const float eps = step_size_dev ? *step_size_dev : step_size_host;
  const float s2 = kick_scale * eps;
  const float s1 = drift_scale * eps;
  for (int64_t c = wave; c < n_chains; c += n_waves) {
    const int64_t off = c * n_data;
    float kin = 0.f;
    if (VEC) {
      for (int64_t d = (int64_t)lane * 4; d < n_data; d += 256) {
        g4 m = g4{1.f, 1.f, 1.f, 1.f};
        if (mass) m = *reinterpret_cast<const g4*>(mass + d);
        // p = p + step_size2 * grad           (hmc.py:42)
        const g4 pv = *reinterpret_cast<const g4*>(p + off + d) +
                      s2 * *reinterpret_cast<const g4*>(grad + off + d);
        *reinterpret_cast<g4*>(p + off + d) = pv;
        // q = q + step_size1 * (p / mass)     (hmc.py:39, :26-27)
        g4 vel = pv;
        if (mass) {
#pragma unroll
          for (int j = 0; j < 4; ++j) vel[j] = pv[j] / m[j];
        }
        if (drift_scale != 0.f)
          *reinterpret_cast<g4*>(q + off + d) =
              *reinterpret_cast<const g4*>(q + off + d) + s1 * vel;
        const g4 e = pv * vel;
        kin += (e[0] + e[1]) + (e[2] + e[3]);
      }
    } else {
      for (int64_t d = lane; d < n_data; d += 64) {
        const float m = mass ? mass[d] : 1.0f;
        const float pv = p[off + d] + s2 * grad[off + d];
        p[off + d] = pv;
        if (drift_scale != 0.f) q[off + d] = q[off + d] + s1 * (pv / m);
        kin += pv * pv / m;
      }
    }
    if (kinetic) {
      kin = wave_sum(kin);
      if (lane == 0) kinetic[c] += 0.5f * kin;
    }
  }
}

__global__ __launch_bounds__(256) void mh_accept_kernel(
    const float* __restrict__ lp_old, const float* __restrict__ lp_new,
    const float* __restrict__ kin_old, const float* __restrict__ kin_new,
    int64_t n_chains, int64_t chain_offset, uint32_t k0, uint32_t k1,
    uint32_t iteration, float* __restrict__ acceptance_rate,
    float* __restrict__ orig_hamiltonian, float* __restrict__ hamiltonian,
    float* __restrict__ log_prob_out, uint8_t* __restrict__ accept,
    double* __restrict__ acc_sum, uint32_t* __restrict__ flags) {
  double acc_local = 0.0;
  bool bad = false;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       c < n_chains; c += (int64_t)gridDim.x * blockDim.x) {
    const float lo = lp_old[c], ln = lp_new[c];
    const float h_old = -lo + kin_old[c];  // hmc.py:31-35
    const float h_new = -ln + kin_new[c];
    const float diff = h_old - h_new;
    float acc = expf(fminf(diff, 0.0f));
    if (!(diff == diff) || !isfinite(acc) || !isfinite(ln)) acc = 0.f;
    if (!isfinite(lo)) bad = true;  // hmc.py:51-53
    const float u = uniform_chain((uint32_t)(c + chain_offset), iteration, k0, k1);
    const bool ok = u < acc;  // hmc.py:486
    if (acceptance_rate) acceptance_rate[c] = acc;
    if (orig_hamiltonian) orig_hamiltonian[c] = h_old;
    if (hamiltonian) hamiltonian[c] = h_new;
    if (log_prob_out) log_prob_out[c] = ok ? ln : lo;
    if (accept) accept[c] = ok ? 1 : 0;
    acc_local += (double)acc;
  }
  // Each wave's partial is rounded to a multiple of 2^-20 before it is added:
  // doubles that are exact multiples of 2^-20 add without rounding up to
  // 2^33, so the total does not depend on the order of the atomics -- the
  // mean acceptance, hence the adapted step size, is bit-identical from run
  // to run (the quantisation, < 5e-7 per 64 chains, is far below the float32
  // mean the tuner consumes).
  const double w =
      __builtin_rint(wave_sum_f64(acc_local) * 1048576.0) * (1.0 / 1048576.0);
  const unsigned long long any_bad = __ballot(bad);
  if ((threadIdx.x & 63) == 0) {
    if (acc_sum && w != 0.0) atomicAdd(acc_sum, w);
    if (any_bad && flags) atomicOr(flags, ZSHMC_FLAG_OLD_LOGPROB_NONFINITE);
  }
}

// Accepted rows only are copied (a rejected row costs one byte of `accept`;
// accept == NULL: every row): the row copy is EXEC-uniform per wave.  Rows of
// dst / src are ld_dst / ld_src floats apart (a latent's own [n_rows, n_cols]
// tensor on one side, its columns of a plan's packed state on the other).
// VEC: 16 B per lane, up to four 1-KiB chunks of the row loaded before the
// first store so that a wave has 4 KiB in flight (the scalar form had 256 B).
template <bool VEC>
__global__ __launch_bounds__(256) void select_rows_kernel(
    float* __restrict__ q, int64_t ld_dst, const float* __restrict__ q_new,
    int64_t ld_src, const uint8_t* __restrict__ accept, int64_t n_chains,
    int64_t n_data) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / 64);
  for (int64_t c = wave; c < n_chains; c += n_waves) {
    if (accept && !accept[c]) continue;  // wave-uniform
    float* __restrict__ drow = q + c * ld_dst;
    const float* __restrict__ srow = q_new + c * ld_src;
    if (VEC) {
      for (int64_t d0 = (int64_t)lane * 4; d0 < n_data; d0 += 4 * 256) {
        g4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (d0 + k * 256 < n_data)
            v[k] = *reinterpret_cast<const g4*>(srow + d0 + k * 256);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (d0 + k * 256 < n_data)
            *reinterpret_cast<g4*>(drow + d0 + k * 256) = v[k];
      }
    } else {
      for (int64_t d = lane; d < n_data; d += 64) drow[d] = srow[d];
    }
  }
}

static inline bool aligned16(const void* p) {
  return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}

static inline int row_grid(int64_t n_rows) {
  const int64_t need = (n_rows + 3) / 4;  // 4 waves per block
  const int64_t cap = (int64_t)device_cu_count() * 8;
  int64_t g = need < cap ? need : cap;
  return (int)(g > 0 ? g : 1);
}

}  // namespace zshmc

using namespace zshmc;

static int momentum_rows(float* p, int64_t ld, const float* mass,
                         int64_t n_chains, int64_t n_data, int64_t chain_offset,
                         uint64_t seed, uint32_t iteration, uint32_t latent_id,
                         float* kinetic, void* stream, const char* who) {
  ZS_REQUIRE(p, "%s: null p", who);
  ZS_REQUIRE(n_chains >= 0 && n_data >= 1 && ld >= n_data, "%s: bad shape", who);
  ZS_REQUIRE(latent_id < (1u << 24), "%s: latent_id too large", who);
  ZS_REQUIRE(n_chains + chain_offset <= 0xFFFFFFFFll,
             "%s: global chain index exceeds 2^32", who);
  if (n_chains == 0) return ZSHMC_OK;
  const bool vec =
      n_data % 4 == 0 && ld % 4 == 0 && aligned16(p) && aligned16(mass);
  if (vec)
    hipLaunchKernelGGL(momentum_kernel<true>, dim3(row_grid(n_chains)),
                       dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p,
                       ld, mass, n_chains, n_data, chain_offset,
                       (uint32_t)(seed & 0xFFFFFFFFull), (uint32_t)(seed >> 32),
                       iteration, kStreamMomentum | (latent_id << 8), kinetic);
  else
    hipLaunchKernelGGL(momentum_kernel<false>, dim3(row_grid(n_chains)),
                       dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p,
                       ld, mass, n_chains, n_data, chain_offset,
                       (uint32_t)(seed & 0xFFFFFFFFull), (uint32_t)(seed >> 32),
                       iteration, kStreamMomentum | (latent_id << 8), kinetic);
  ZS_LAUNCH_CHECK("momentum_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_momentum(float* p, const float* mass, int64_t n_chains,
                              int64_t n_data, int64_t chain_offset,
                              uint64_t seed, uint32_t iteration,
                              uint32_t latent_id, float* kinetic,
                              void* stream) {
  return momentum_rows(p, n_data, mass, n_chains, n_data, chain_offset, seed,
                       iteration, latent_id, kinetic, stream, "zshmc_momentum");
}

extern "C" int zshmc_momentum_rows(float* p, int64_t row_stride,
                                   const float* mass, int64_t n_chains,
                                   int64_t n_data, int64_t chain_offset,
                                   uint64_t seed, uint32_t iteration,
                                   uint32_t latent_id, float* kinetic,
                                   void* stream) {
  return momentum_rows(p, row_stride, mass, n_chains, n_data, chain_offset,
                       seed, iteration, latent_id, kinetic, stream,
                       "zshmc_momentum_rows");
}

extern "C" int zshmc_kick_drift(float* q, float* p, const float* grad,
                                const float* mass, const float* step_size_dev,
                                float step_size_host, float kick_scale,
                                float drift_scale, int64_t n_chains,
                                int64_t n_data, float* kinetic, void* stream) {
  ZS_REQUIRE(q && p && grad, "zshmc_kick_drift: null q/p/grad");
  ZS_REQUIRE(n_chains >= 0 && n_data >= 1, "zshmc_kick_drift: bad shape");
  if (n_chains == 0) return ZSHMC_OK;
  const bool vec = n_data % 4 == 0 && aligned16(q) && aligned16(p) &&
                   aligned16(grad) && aligned16(mass);
  if (vec)
    hipLaunchKernelGGL(kick_drift_kernel<true>, dim3(row_grid(n_chains)),
                       dim3(256), 0, reinterpret_cast<hipStream_t>(stream), q, p,
                       grad, mass, step_size_dev, step_size_host, kick_scale,
                       drift_scale, n_chains, n_data, kinetic);
  else
    hipLaunchKernelGGL(kick_drift_kernel<false>, dim3(row_grid(n_chains)),
                       dim3(256), 0, reinterpret_cast<hipStream_t>(stream), q, p,
                       grad, mass, step_size_dev, step_size_host, kick_scale,
                       drift_scale, n_chains, n_data, kinetic);
  ZS_LAUNCH_CHECK("kick_drift_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_mh_accept(const float* log_prob_old,
                               const float* log_prob_new,
                               const float* kinetic_old,
                               const float* kinetic_new, int64_t n_chains,
                               int64_t chain_offset, uint64_t seed,
                               uint32_t iteration, float* acceptance_rate,
                               float* orig_hamiltonian, float* hamiltonian,
                               float* log_prob_out, uint8_t* accept,
                               double* acc_sum, uint32_t* flags, void* stream) {
  ZS_REQUIRE(log_prob_old && log_prob_new && kinetic_old && kinetic_new,
             "zshmc_mh_accept: null input");
  ZS_REQUIRE(n_chains >= 0, "zshmc_mh_accept: bad n_chains");
  ZS_REQUIRE(n_chains + chain_offset <= 0xFFFFFFFFll,
             "zshmc_mh_accept: global chain index exceeds 2^32");
  if (n_chains == 0) return ZSHMC_OK;
  int64_t blocks = (n_chains + 255) / 256;
  const int64_t cap = (int64_t)device_cu_count() * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(mh_accept_kernel, dim3((int)blocks), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), log_prob_old,
                     log_prob_new, kinetic_old, kinetic_new, n_chains,
                     chain_offset, (uint32_t)(seed & 0xFFFFFFFFull),
                     (uint32_t)(seed >> 32), iteration, acceptance_rate,
                     orig_hamiltonian, hamiltonian, log_prob_out, accept,
                     acc_sum, flags);
  ZS_LAUNCH_CHECK("mh_accept_kernel launch");
  return ZSHMC_OK;
}

static int copy_rows(float* dst, int64_t ld_dst, const float* src,
                     int64_t ld_src, const uint8_t* accept, int64_t n_rows,
                     int64_t n_cols, void* stream) {
  if (n_cols % 4 == 0 && ld_dst % 4 == 0 && ld_src % 4 == 0 && aligned16(dst) &&
      aligned16(src))
    hipLaunchKernelGGL(select_rows_kernel<true>, dim3(row_grid(n_rows)),
                       dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dst,
                       ld_dst, src, ld_src, accept, n_rows, n_cols);
  else
    hipLaunchKernelGGL(select_rows_kernel<false>, dim3(row_grid(n_rows)),
                       dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dst,
                       ld_dst, src, ld_src, accept, n_rows, n_cols);
  ZS_LAUNCH_CHECK("select_rows_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_select_rows(float* q, const float* q_new,
                                 const uint8_t* accept, int64_t n_chains,
                                 int64_t n_data, void* stream) {
  ZS_REQUIRE(q && q_new && accept, "zshmc_select_rows: null pointer");
  ZS_REQUIRE(n_chains >= 0 && n_data >= 1, "zshmc_select_rows: bad shape");
  if (n_chains == 0) return ZSHMC_OK;
  return copy_rows(q, n_data, q_new, n_data, accept, n_chains, n_data, stream);
}

extern "C" int zshmc_copy_rows(float* dst, int64_t dst_stride, const float* src,
                               int64_t src_stride, const uint8_t* accept,
                               int64_t n_rows, int64_t n_cols, void* stream) {
  ZS_REQUIRE(dst && src, "zshmc_copy_rows: null pointer");
  ZS_REQUIRE(n_rows >= 0 && n_cols >= 1 && dst_stride >= n_cols &&
                 src_stride >= n_cols,
             "zshmc_copy_rows: bad shape");
  if (n_rows == 0) return ZSHMC_OK;
  return copy_rows(dst, dst_stride, src, src_stride, accept, n_rows, n_cols,
                   stream);
}
