// Gathered row dot products and their scatter gradient -- the rating
// likelihood of the reference's probabilistic matrix factorisation example
// (examples/probabilistic_matrix_factorization/
// pmf_hmc.py:26-28):
//     r_logits[k, e] = sum_d u[k, select_u[e], d] * v[k, select_v[e], d]
// which the reference writes as two tf.gather(axis=1) that materialise
// [K, E, D] each, a product and a reduce_sum, and whose tf.gradients is an
// unsorted-segment scatter-add.  Here the [K, E, D] intermediates never exist:
//   forward   16 lanes per (chain, pair) read the two rows where they lie
//             (rows are D*4 contiguous bytes; repeats hit L2) and reduce the
//             products inside their DPP row;
//   backward  d/du[k, i, :] = sum_{e : select_u[e] = i} g[k, e] * v[k, select_v[e], :]
//             as a segmented sum over a CSR view of the pair list (pairs
//             ordered by the row they contribute to): one 16-lane group per
//             (chain, row), no atomics, so the result is bit-reproducible.
// HBM/L2-bound irregular access; algorithmic bytes per pair and chain:
// 2*D*4 (rows) + 4 (logit) forward, D*4 + 4 + 8 backward per side.
#include "common.h"

namespace zshmc {

constexpr int kGdLanes = 16;  // lanes per (chain, pair) / (chain, row)

__global__ __launch_bounds__(256) void gather_dot_kernel(
    const float* __restrict__ u, const float* __restrict__ v,
    const int32_t* __restrict__ su, const int32_t* __restrict__ sv,
    int64_t n_chains, int64_t n_u, int64_t n_v, int64_t n_pairs, int D,
    float* __restrict__ out) {
  const int sub = threadIdx.x % kGdLanes;
  const int64_t groups_per_block = blockDim.x / kGdLanes;
  const int64_t total = n_chains * n_pairs;
  for (int64_t t = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / kGdLanes;
       t < total; t += (int64_t)gridDim.x * groups_per_block) {
    const int64_t k = t / n_pairs, e = t - k * n_pairs;
    const float* __restrict__ ur = u + (k * n_u + su[e]) * D;
    const float* __restrict__ vr = v + (k * n_v + sv[e]) * D;
    float acc = 0.f;
    for (int d = sub; d < D; d += kGdLanes) acc = fmaf(ur[d], vr[d], acc);
    acc = group_sum<kGdLanes>(acc);
    if (sub == 0) out[t] = acc;
  }
}

// grad[k, i, :] = sum_{p in order[seg[i] .. seg[i+1])} g[k, p] * other[k, oidx[p], :]
__global__ __launch_bounds__(256) void gather_dot_grad_kernel(
    const float* __restrict__ other, const float* __restrict__ gout,
    const int32_t* __restrict__ seg, const int32_t* __restrict__ order,
    const int32_t* __restrict__ oidx, int64_t n_chains, int64_t n_rows,
    int64_t n_other, int64_t n_pairs, int D, float* __restrict__ grad) {
  const int sub = threadIdx.x % kGdLanes;
  const int64_t groups_per_block = blockDim.x / kGdLanes;
  const int64_t total = n_chains * n_rows;
  for (int64_t t = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / kGdLanes;
       t < total; t += (int64_t)gridDim.x * groups_per_block) {
    const int64_t k = t / n_rows, i = t - k * n_rows;
    const int32_t b = seg[i], e = seg[i + 1];
    const float* __restrict__ g = gout + k * n_pairs;
    const float* __restrict__ ob = other + k * n_other * D;
    float* __restrict__ gr = grad + t * D;
    for (int d = sub; d < D; d += kGdLanes) {
      float a0 = 0.f, a1 = 0.f;
      int32_t q = b;
      for (; q + 2 <= e; q += 2) {
        const int32_t p0 = order[q], p1 = order[q + 1];
        a0 = fmaf(g[p0], ob[(int64_t)oidx[p0] * D + d], a0);
        a1 = fmaf(g[p1], ob[(int64_t)oidx[p1] * D + d], a1);
      }
      if (q < e) {
        const int32_t p0 = order[q];
        a0 = fmaf(g[p0], ob[(int64_t)oidx[p0] * D + d], a0);
      }
      gr[d] = a0 + a1;
    }
  }
}

// ---------------------------------------------------------------------------
// The rating likelihood of pmf_hmc.py:26-31 in one pass over the pair list --
// what the NATIVE plan of the gathered-dot model evaluates per leapfrog trip:
//   d[k, e]   = sum_j u[k, su[e], j] * v[k, sv[e], j]
//   pred      = sigmoid(d)                       (bn.normal("r", tf.sigmoid(..)))
//   term      = log N(obs[e]; pred, exp(logstd)) (univariate.py:174-181)
//   g[k, e]   = d term / d d = (obs - pred) exp(-2 logstd) pred (1 - pred)
//               (what tf.gradients, hmc.py:430-432, yields through the sigmoid)
// A workgroup owns kGdBlockPairs consecutive pairs of ONE chain: its terms are
// added in a fixed order into partial[k, block]; gd_lik_finish_kernel adds a
// chain's partials in block order (+ the constant of the observed nodes):
// deterministic, no atomics.
constexpr int kGdBlockPairs = 128;  // 16 groups x 8 pairs

__global__ __launch_bounds__(256) void gather_dot_normal_lik_kernel(
    const float* __restrict__ u, const float* __restrict__ v,
    const int32_t* __restrict__ su, const int32_t* __restrict__ sv,
    const float* __restrict__ obs, int64_t obs_rows, float logstd,
    int64_t n_chains, int64_t n_u, int64_t n_v, int64_t n_pairs, int D,
    float* __restrict__ g_out, float* __restrict__ partial, int64_t n_blocks) {
  __shared__ float red[256 / kGdLanes];
  const int sub = threadIdx.x % kGdLanes, grp = threadIdx.x / kGdLanes;
  const float prec = expf(-2.0f * logstd);
  const float c0 = -0.91893853320467274178f - logstd;
  const int64_t n_work = n_chains * n_blocks;
  for (int64_t wk = blockIdx.x; wk < n_work; wk += gridDim.x) {
    const int64_t k = wk / n_blocks, blk = wk - k * n_blocks;
    const float* __restrict__ ob = obs + (k % obs_rows) * n_pairs;
    float acc_ll = 0.f;
    for (int it = 0; it < kGdBlockPairs / (256 / kGdLanes); ++it) {
      const int64_t e = blk * kGdBlockPairs + it * (256 / kGdLanes) + grp;
      if (e < n_pairs) {      // (uniform inside a 16-lane group)
        const float* __restrict__ ur = u + (k * n_u + su[e]) * D;
        const float* __restrict__ vr = v + (k * n_v + sv[e]) * D;
        float acc = 0.f;
        for (int d = sub; d < D; d += kGdLanes) acc = fmaf(ur[d], vr[d], acc);
        acc = group_sum<kGdLanes>(acc);
        // sigmoid as tf.sigmoid: 1 / (1 + exp(-d))
        const float pred = 1.0f / (1.0f + expf(-acc));
        const float diff = ob[e] - pred;
        acc_ll += c0 - 0.5f * prec * diff * diff;
        if (g_out && sub == 0)
          g_out[k * n_pairs + e] = diff * prec * pred * (1.0f - pred);
      }
    }
    if (sub == 0) red[grp] = acc_ll;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 256 / kGdLanes; ++i) t += red[i];
      partial[wk] = t;
    }
    __syncthreads();
  }
}

// one workgroup per chain: thread i adds the partials i, i + 256, ... in that
// order, then a fixed tree over the 256 threads -- the same bits every run
__global__ __launch_bounds__(256) void gd_lik_finish_kernel(
    const float* __restrict__ partial, int64_t n_blocks,
    const float* __restrict__ lp_const, int64_t n_chains,
    float* __restrict__ log_lik) {
  __shared__ float red[256];
  const int64_t k = blockIdx.x;
  float t = 0.f;
  for (int64_t b = threadIdx.x; b < n_blocks; b += 256)
    t += partial[k * n_blocks + b];
  red[threadIdx.x] = t;
  __syncthreads();
#pragma unroll
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) log_lik[k] = red[0] + (lp_const ? lp_const[k] : 0.f);
}

static int gd_grid(int64_t groups) {
  int64_t blocks = (groups + (256 / kGdLanes) - 1) / (256 / kGdLanes);
  const int64_t cap = (int64_t)device_cu_count() * 32;
  if (blocks > cap) blocks = cap;
  return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace zshmc

using namespace zshmc;

extern "C" int zshmc_gather_dot(const float* u, const float* v,
                                const int32_t* select_u, const int32_t* select_v,
                                int64_t n_chains, int64_t n_u, int64_t n_v,
                                int64_t n_pairs, int64_t n_dim, float* out,
                                void* stream) {
  if (n_chains * n_pairs == 0) return ZSHMC_OK;
  ZS_REQUIRE(u && v && select_u && select_v && out,
             "zshmc_gather_dot: null pointer");
  ZS_REQUIRE(n_chains > 0 && n_u > 0 && n_v > 0 && n_pairs > 0 && n_dim > 0 &&
                 n_dim <= (1 << 20) && n_pairs < (1ll << 31),
             "zshmc_gather_dot: bad shape");
  hipLaunchKernelGGL(gather_dot_kernel, dim3(gd_grid(n_chains * n_pairs)),
                     dim3(256), 0, reinterpret_cast<hipStream_t>(stream), u, v,
                     select_u, select_v, n_chains, n_u, n_v, n_pairs, (int)n_dim,
                     out);
  ZS_LAUNCH_CHECK("gather_dot_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_gather_dot_grad(const float* other, const float* gout,
                                     const int32_t* seg_ptr, const int32_t* order,
                                     const int32_t* other_index, int64_t n_chains,
                                     int64_t n_rows, int64_t n_other,
                                     int64_t n_pairs, int64_t n_dim, float* grad,
                                     void* stream) {
  if (n_chains * n_rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(other && gout && seg_ptr && order && other_index && grad,
             "zshmc_gather_dot_grad: null pointer");
  ZS_REQUIRE(n_chains > 0 && n_rows > 0 && n_other > 0 && n_pairs >= 0 &&
                 n_dim > 0 && n_dim <= (1 << 20) && n_pairs < (1ll << 31),
             "zshmc_gather_dot_grad: bad shape");
  hipLaunchKernelGGL(gather_dot_grad_kernel, dim3(gd_grid(n_chains * n_rows)),
                     dim3(256), 0, reinterpret_cast<hipStream_t>(stream), other,
                     gout, seg_ptr, order, other_index, n_chains, n_rows, n_other,
                     n_pairs, (int)n_dim, grad);
  ZS_LAUNCH_CHECK("gather_dot_grad_kernel launch");
  return ZSHMC_OK;
}

extern "C" int64_t zshmc_gather_dot_normal_workspace(int64_t n_chains,
                                                     int64_t n_pairs) {
  return n_chains * ((n_pairs + kGdBlockPairs - 1) / kGdBlockPairs);
}

extern "C" int zshmc_gather_dot_normal_lik(
    const float* u, const float* v, const int32_t* select_u,
    const int32_t* select_v, const float* obs, int64_t obs_rows, float logstd,
    const float* lp_const, int64_t n_chains, int64_t n_u, int64_t n_v,
    int64_t n_pairs, int64_t n_dim, float* g_out, float* log_lik,
    float* workspace, void* stream) {
  if (n_chains == 0) return ZSHMC_OK;
  ZS_REQUIRE(u && v && log_lik && (n_pairs == 0 || (select_u && select_v &&
                                                    obs && workspace)),
             "zshmc_gather_dot_normal_lik: null pointer");
  ZS_REQUIRE(n_chains > 0 && n_u > 0 && n_v > 0 && n_pairs >= 0 && n_dim > 0 &&
                 n_dim <= (1 << 20) && n_pairs < (1ll << 31) &&
                 (obs_rows == 1 || obs_rows == n_chains),
             "zshmc_gather_dot_normal_lik: bad shape");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t n_blocks = (n_pairs + kGdBlockPairs - 1) / kGdBlockPairs;
  if (n_blocks > 0) {
    const int64_t n_work = n_chains * n_blocks;
    const int64_t cap = (int64_t)device_cu_count() * 32;
    hipLaunchKernelGGL(gather_dot_normal_lik_kernel,
                       dim3((unsigned)(n_work < cap ? n_work : cap)), dim3(256),
                       0, s, u, v, select_u, select_v, obs, obs_rows, logstd,
                       n_chains, n_u, n_v, n_pairs, (int)n_dim, g_out, workspace,
                       n_blocks);
    ZS_LAUNCH_CHECK("gather_dot_normal_lik_kernel launch");
  }
  hipLaunchKernelGGL(gd_lik_finish_kernel, dim3((unsigned)n_chains), dim3(256),
                     0, s, workspace, n_blocks, lp_const, n_chains, log_lik);
  ZS_LAUNCH_CHECK("gd_lik_finish_kernel launch");
  return ZSHMC_OK;
}
