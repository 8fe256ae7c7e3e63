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

// ---------------------------------------------------------------------------
// The same likelihood AND its gradient w.r.t. the latent table in ONE pass
// over a CSR view of the pair list (round 5; what the native plan runs per
// leapfrog trip when rows are <= 128 floats, a multiple of 4):
//   work item = (chain k, segment s): <= kGdSegPairs consecutive CSR slots of
//   ONE latent row i = seg_row[s].  An 8-lane group holds the latent row
//   u[k, i, :] in registers (16 B per lane and 32 floats), streams the slots'
//   other-table indices and ratings -- CSR order, so sequential -- gathers
//   each other row ONCE (8 lanes x 16 B = the 128 contiguous bytes of 32
//   floats; four slots in flight), and uses it twice: for the dot product /
//   sigmoid / Normal term, and for grad += g * v.  The 8 groups of a wave are
//   8 chains of the SAME segment (chain fastest), so the index and rating
//   loads are wave-wide broadcasts and the trip counts agree.
// Against the two-kernel form above (dot per pair with 16 lanes and scalar
// loads, g[k, e] to memory, then a second gather of the other rows for the
// scatter): one gather instead of three, no [K, E] round trip.  Rows with
// more slots than a segment are finished by gd_fused_combine_kernel (partial
// gradients of a row added in segment order): deterministic, no atomics.
// Bound: the L2 gather rate (the tables are L2 / Infinity-Cache resident) --
// algorithmic gathered bytes per (chain, pair) = one other row, 4 D.
constexpr int kGdSegPairs = 256;
typedef float g4 __attribute__((ext_vector_type(4)));

template <int NCH>   // row chunks of 32 floats: D <= 32 NCH
__global__ __launch_bounds__(256) void gd_fused_kernel(
    const float* __restrict__ lat, const float* __restrict__ other,
    const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_row,
    const int32_t* __restrict__ seg_first,   // [n_lat]: first segment of a row
    const int32_t* __restrict__ oidx_csr, const float* __restrict__ obs_csr,
    int64_t obs_rows, float logstd, int64_t n_chains, int64_t n_lat,
    int64_t n_other, int64_t n_pairs, int64_t n_seg, int D,
    float* __restrict__ grad, float* __restrict__ grad_extra,
    float* __restrict__ partial) {
  const int sub = threadIdx.x & 7;
  const int64_t grp0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
  const int64_t n_groups = (int64_t)gridDim.x * 32;
  const float prec = expf(-2.0f * logstd);
  const float c0 = -0.91893853320467274178f - logstd;
  const int64_t total = n_seg * n_chains;
  for (int64_t t = grp0; t < total; t += n_groups) {
    const int64_t s = t / n_chains, k = t - s * n_chains;   // chain fastest
    const int32_t i = seg_row[s];
    const int32_t b = seg_ptr[s], e = seg_ptr[s + 1];
    const float* __restrict__ lrow = lat + (k * n_lat + i) * D;
    const float* __restrict__ ob = other + k * n_other * D;
    const float* __restrict__ rk = obs_csr + (k % obs_rows) * n_pairs;
    g4 uq[NCH], ga[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int d = c * 32 + sub * 4;
      uq[c] = d < D ? *reinterpret_cast<const g4*>(lrow + d)
                    : g4{0.f, 0.f, 0.f, 0.f};
      ga[c] = g4{0.f, 0.f, 0.f, 0.f};
    }
    float ll = 0.f;
    for (int32_t q = b; q < e; q += 4) {
      int32_t j[4];
      float r[4];
      g4 v[4][NCH];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int32_t qq = q + m < e ? q + m : e - 1;   // (clamped: masked below)
        j[m] = oidx_csr[qq];
        r[m] = rk[qq];
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int d = c * 32 + sub * 4;
          v[m][c] = d < D ? *reinterpret_cast<const g4*>(
                                ob + (int64_t)j[m] * D + d)
                          : g4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const g4 pr = uq[c] * v[m][c];
          acc += (pr[0] + pr[1]) + (pr[2] + pr[3]);
        }
        acc = group_sum<8>(acc);
        // sigmoid as tf.sigmoid: 1 / (1 + exp(-d))
        const float pred = 1.0f / (1.0f + expf(-acc));
        const float diff = r[m] - pred;
        const bool on = q + m < e;
        ll += on ? c0 - 0.5f * prec * diff * diff : 0.f;
        const float g = on ? diff * prec * pred * (1.0f - pred) : 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) ga[c] += g * v[m][c];
      }
    }
    // a row's only segment writes the row; the segments of a longer row
    // leave partials for gd_fused_combine_kernel
    const bool whole = seg_first[i] == (int32_t)s &&
                       (s + 1 == n_seg || seg_row[s + 1] != i);
    float* __restrict__ dst = whole ? grad + (k * n_lat + i) * D
                                    : grad_extra + (k * n_seg + s) * D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int d = c * 32 + sub * 4;
      if (d < D) *reinterpret_cast<g4*>(dst + d) = ga[c];
    }
    if (sub == 0) partial[k * n_seg + s] = ll;
  }
}

// rows cut into several segments: grad[k, i, :] = their partials in segment order
__global__ __launch_bounds__(256) void gd_fused_combine_kernel(
    const int32_t* __restrict__ long_rows, int64_t n_long,
    const int32_t* __restrict__ seg_first, const int32_t* __restrict__ seg_row,
    int64_t n_chains, int64_t n_lat, int64_t n_seg, int D,
    const float* __restrict__ grad_extra, float* __restrict__ grad) {
  const int64_t total = n_long * n_chains * D;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(t % D);
    const int64_t k = (t / D) % n_chains;
    const int32_t i = long_rows[t / D / n_chains];
    float acc = 0.f;
    for (int64_t s = seg_first[i]; s < n_seg && seg_row[s] == i; ++s)
      acc += grad_extra[(k * n_seg + s) * D + d];
    grad[(k * n_lat + i) * D + d] = acc;
  }
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

extern "C" int zshmc_gather_dot_normal_lik_grad(
    const float* latent, const float* other, const int32_t* seg_ptr,
    const int32_t* seg_row, const int32_t* seg_first, const int32_t* long_rows,
    int64_t n_long, const int32_t* other_index_csr, const float* obs_csr,
    int64_t obs_rows, float logstd, const float* lp_const, int64_t n_chains,
    int64_t n_latent, int64_t n_other, int64_t n_pairs, int64_t n_segments,
    int64_t n_dim, float* grad, float* log_lik, float* workspace, void* stream) {
  if (n_chains == 0) return ZSHMC_OK;
  ZS_REQUIRE(latent && other && grad && log_lik && workspace && seg_ptr &&
                 seg_row && seg_first && (n_pairs == 0 || (other_index_csr &&
                                                           obs_csr)) &&
                 (n_long == 0 || long_rows),
             "zshmc_gather_dot_normal_lik_grad: null pointer");
  ZS_REQUIRE(n_chains > 0 && n_latent > 0 && n_other > 0 && n_pairs >= 0 &&
                 n_segments >= 0 && n_dim > 0 && n_dim <= 128 && n_dim % 4 == 0 &&
                 n_pairs < (1ll << 31) && n_long >= 0 &&
                 (obs_rows == 1 || obs_rows == n_chains),
             "zshmc_gather_dot_normal_lik_grad: bad shape (rows of <= 128 "
             "floats, a multiple of 4)");
  ZS_REQUIRE(((reinterpret_cast<uintptr_t>(latent) |
               reinterpret_cast<uintptr_t>(other) |
               reinterpret_cast<uintptr_t>(grad)) & 15) == 0,
             "zshmc_gather_dot_normal_lik_grad: tables must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // (every latent row has at least one segment -- an empty one writes its
  // zero gradient -- so nothing is cleared beforehand)
  ZS_REQUIRE(n_segments >= n_latent,
             "zshmc_gather_dot_normal_lik_grad: every latent row needs a "
             "segment (rows without pairs an empty one)");
  float* partial = workspace;                       // [n_chains, n_segments]
  // (the partial sums are padded to a multiple of 4 floats: `extra` is written
  // with 16-byte stores)
  float* extra = workspace + ((n_chains * n_segments + 3) & ~int64_t(3));
                                                    // [n_chains, n_segments, D]
  ZS_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
             "zshmc_gather_dot_normal_lik_grad: workspace must be 16-byte "
             "aligned");
  if (n_segments > 0) {
    const int64_t groups = n_chains * n_segments;
    int64_t blocks = (groups + 31) / 32;
    const int64_t cap = (int64_t)device_cu_count() * 32;
    if (blocks > cap) blocks = cap;
    const int nch = (int)((n_dim + 31) / 32);
#define ZS_GD_FUSED(N)                                                          \
  hipLaunchKernelGGL(gd_fused_kernel<N>, dim3((unsigned)blocks), dim3(256), 0,  \
                     s, latent, other, seg_ptr, seg_row, seg_first,             \
                     other_index_csr, obs_csr, obs_rows, logstd, n_chains,      \
                     n_latent, n_other, n_pairs, n_segments, (int)n_dim, grad,  \
                     extra, partial)
    switch (nch) {
      case 1: ZS_GD_FUSED(1); break;
      case 2: ZS_GD_FUSED(2); break;
      case 3: ZS_GD_FUSED(3); break;
      default: ZS_GD_FUSED(4); break;
    }
#undef ZS_GD_FUSED
    ZS_LAUNCH_CHECK("gd_fused_kernel launch");
    if (n_long > 0) {
      const int64_t n = n_long * n_chains * n_dim;
      int64_t cb = (n + 255) / 256;
      if (cb > cap) cb = cap;
      hipLaunchKernelGGL(gd_fused_combine_kernel, dim3((unsigned)cb), dim3(256),
                         0, s, long_rows, n_long, seg_first, seg_row, n_chains,
                         n_latent, n_segments, (int)n_dim, extra, grad);
      ZS_LAUNCH_CHECK("gd_fused_combine_kernel launch");
    }
  }
  hipLaunchKernelGGL(gd_lik_finish_kernel, dim3((unsigned)n_chains), dim3(256),
                     0, s, partial, n_segments, lp_const, n_chains, log_lik);
  ZS_LAUNCH_CHECK("gd_lik_finish_kernel launch");
  return ZSHMC_OK;
}
