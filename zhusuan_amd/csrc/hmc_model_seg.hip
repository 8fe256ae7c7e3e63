// The leapfrog step of the NATIVE plans whose latent is LONG per chain and
// whose likelihood gradient arrives in SEGMENTS -- the counterpart of
// csrc/hmc_model.hip (one wave per row, row <= 1024 floats, in registers) for
//
//   * the dense-logit Categorical (softmax regression): one chain's latent
//     w[c, 0:K, 0:F] is K class rows of F features; the fused likelihood
//     (zshmc_linear_categorical_log_lik) takes and returns them as rows
//     c * G + k of a [C * G, width] matrix (G = class stride, a power of two);
//   * the gathered-dot rating model (pmf_hmc.py:19-31): a chain is a whole
//     [n_users, n_factors] table -- 10^5 floats per chain, a handful of chains;
//     the gradient arrives as a plain [C, n_data] matrix (G = 1, F = n_data).
//
// Per element e = k * F + f of chain c (reference zhusuan/hmc.py:38-43 with
// the Normal prior of univariate.py:174-181):
//   grad   = lik_scale * grad_lik[(c*G + k), f] - exp(-2 logstd_e)(q_e - mean_e)
//   p_e   += kick_scale * eps * grad ;  q_e += drift_scale * eps * p_e / mass_e
//   operand[(c*G + k), f] = q_e'      (the next likelihood evaluation's W rows;
//                                      padding rows / columns are never touched
//                                      and stay zero)
// and per chain
//   lp_out[c]   = lik_scale * sum_{k<G} ll_in[c*G + k] + sum_e log N(q_e)   AT
//                 the evaluation point;   kinetic[c] += 1/2 sum_e p_e'^2 / mass_e
// A workgroup owns a 1024-element chunk of one chain (so a few long chains
// still fill the chip); the chunk sums go to a workspace and a second small
// launch adds them per chain IN CHUNK ORDER: deterministic, no atomics.
// HBM-bound: 5-6 passes of 4 bytes per element.
#include "common.h"

namespace zshmc {

typedef float s4 __attribute__((ext_vector_type(4)));
constexpr int kSegChunk = 1024;  // elements per workgroup
constexpr float kSegNegHalfLog2Pi = -0.91893853320467274178f;

struct SegArgs {
  float* q;
  float* p;
  const float* grad;
  int64_t grad_stride;
  float* operand;
  int64_t operand_stride;
  int64_t seg_len, groups;
  const float* prior_mean;
  int64_t mean_rows;
  const float* prior_logstd;
  int64_t logstd_rows;
  const float* mass;
  const float* step_size_dev;
  float step_size_host, kick_scale, drift_scale, lik_scale;
  int64_t n_chains, n_data, ld;
  float* partials;  // [n_chains, n_chunks, 2]
  int64_t n_chunks;
};

__global__ __launch_bounds__(256) void model_seg_step_kernel(SegArgs a) {
  __shared__ float red[2][4];
  const float eps = a.step_size_dev ? *a.step_size_dev : a.step_size_host;
  const float s2 = a.kick_scale * eps, s1 = a.drift_scale * eps;
  const int64_t F = a.seg_len, G = a.groups;
  const bool vec_seg = (F & 3) == 0;  // 4 consecutive elements share a segment
  const int64_t n_work = a.n_chains * a.n_chunks;
  for (int64_t wk = blockIdx.x; wk < n_work; wk += gridDim.x) {
    const int64_t c = wk / a.n_chunks, chunk = wk % a.n_chunks;
    const int64_t e0 = chunk * kSegChunk + (int64_t)threadIdx.x * 4;
    float prior = 0.f, kin = 0.f;
    if (e0 < a.n_data) {
      float* __restrict__ qp = a.q + c * a.ld + e0;
      float* __restrict__ pp = a.p + c * a.ld + e0;
      s4 q = *reinterpret_cast<const s4*>(qp);
      s4 p = *reinterpret_cast<const s4*>(pp);
      const s4 mu = *reinterpret_cast<const s4*>(
          a.prior_mean + (c % a.mean_rows) * a.ld + e0);
      const s4 ls = *reinterpret_cast<const s4*>(
          a.prior_logstd + (c % a.logstd_rows) * a.ld + e0);
      s4 im = s4{1.f, 1.f, 1.f, 1.f};
      if (a.mass) {
        const s4 m = *reinterpret_cast<const s4*>(a.mass + e0);
#pragma unroll
        for (int j = 0; j < 4; ++j) im[j] = 1.0f / m[j];
      }
      // likelihood gradient of the four elements (their segment rows)
      s4 gl = s4{0.f, 0.f, 0.f, 0.f};
      int64_t row[4], col[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t e = e0 + j;
        const int64_t k = e / F;
        row[j] = c * G + k;
        col[j] = e - k * F;
      }
      if (a.grad) {
        if (vec_seg) {
          gl = *reinterpret_cast<const s4*>(a.grad + row[0] * a.grad_stride +
                                            col[0]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (e0 + j < a.n_data)
              gl[j] = a.grad[row[j] * a.grad_stride + col[j]];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool on = e0 + j < a.n_data;  // (the last group of a padded row)
        const float prec = expf(-2.0f * ls[j]);
        const float r = q[j] - mu[j];
        prior += on ? kSegNegHalfLog2Pi - ls[j] - 0.5f * prec * r * r : 0.f;
        const float g = on ? a.lik_scale * gl[j] - prec * r : 0.f;
        p[j] = p[j] + s2 * g;
        const float vel = p[j] * im[j];
        q[j] = q[j] + s1 * vel;
        kin += on ? p[j] * vel : 0.f;
      }
      if (a.kick_scale != 0.f) *reinterpret_cast<s4*>(pp) = p;
      if (a.drift_scale != 0.f) *reinterpret_cast<s4*>(qp) = q;
      if (a.operand) {
        if (vec_seg) {
          *reinterpret_cast<s4*>(a.operand + row[0] * a.operand_stride +
                                 col[0]) = q;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (e0 + j < a.n_data)
              a.operand[row[j] * a.operand_stride + col[j]] = q[j];
        }
      }
    }
    // chunk sums: wave butterflies, then the four waves in order
    prior = group_sum<64>(prior);
    kin = group_sum<64>(kin);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
      red[0][wave] = prior;
      red[1][wave] = kin;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float* __restrict__ out = a.partials + wk * 2;
      out[0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
      out[1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    }
    __syncthreads();
  }
}

// A wave per chain: lane l adds the chunk sums l, l + 64, ... in that order,
// then a fixed butterfly over the 64 lanes -- the same bits every run.
__global__ __launch_bounds__(256) void model_seg_finish_kernel(
    const float* __restrict__ partials, int64_t n_chunks,
    const float* __restrict__ ll_in, int64_t groups, float lik_scale,
    int64_t n_chains, float* __restrict__ lp_out, float* __restrict__ kinetic) {
  const int lane = threadIdx.x & 63;
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n_chains) return;        // (wave-uniform)
  float prior = 0.f, kin = 0.f, ll = 0.f;
  const float* __restrict__ pc = partials + c * n_chunks * 2;
  for (int64_t j = lane; j < n_chunks; j += 64) {
    prior += pc[2 * j];
    kin += pc[2 * j + 1];
  }
  if (lp_out && ll_in)
    for (int64_t k = lane; k < groups; k += 64) ll += ll_in[c * groups + k];
  prior = group_sum<64>(prior);
  kin = group_sum<64>(kin);
  ll = group_sum<64>(ll);
  if (lane == 0) {
    if (lp_out) lp_out[c] = lik_scale * ll + prior;
    if (kinetic) kinetic[c] += 0.5f * kin;
  }
}

}  // namespace zshmc

using namespace zshmc;

extern "C" int64_t zshmc_model_seg_workspace(int64_t n_chains, int64_t n_data) {
  return 2 * n_chains * ((n_data + kSegChunk - 1) / kSegChunk);
}

extern "C" int zshmc_model_kick_drift_seg(
    float* q, float* p, const float* grad_lik, int64_t grad_stride,
    int64_t seg_len, int64_t groups, float* operand, int64_t operand_stride,
    const float* prior_mean, int64_t mean_rows, const float* prior_logstd,
    int64_t logstd_rows, const float* mass, const float* step_size_dev,
    float step_size_host, float kick_scale, float drift_scale,
    float lik_scale, int64_t n_chains, int64_t n_data, int64_t row_stride,
    const float* ll_in, float* lp_out, float* kinetic, float* workspace,
    void* stream) {
  const int64_t ld = row_stride;
  ZS_REQUIRE(q && p && prior_mean && prior_logstd && workspace,
             "zshmc_model_kick_drift_seg: null q/p/prior/workspace");
  ZS_REQUIRE(n_chains >= 0 && n_data >= 1 && ld >= n_data && ld % 4 == 0,
             "zshmc_model_kick_drift_seg: 1 <= n_data %lld <= row_stride %lld, "
             "row_stride a multiple of 4", (long long)n_data, (long long)ld);
  ZS_REQUIRE(seg_len >= 1 && groups >= 1 &&
                 (n_data + seg_len - 1) / seg_len <= groups,
             "zshmc_model_kick_drift_seg: n_data %lld does not fit %lld "
             "segments of %lld", (long long)n_data, (long long)groups,
             (long long)seg_len);
  ZS_REQUIRE(mean_rows >= 1 && logstd_rows >= 1,
             "zshmc_model_kick_drift_seg: prior row periods must be >= 1");
  // (a segment length that is a multiple of 4 is read / written 16 bytes at a
  // time: the strides then have to keep that alignment)
  const bool vec = seg_len % 4 == 0;
  ZS_REQUIRE(!grad_lik || (grad_stride >= seg_len &&
                           (!vec || grad_stride % 4 == 0)),
             "zshmc_model_kick_drift_seg: bad grad_stride");
  ZS_REQUIRE(!operand || (operand_stride >= seg_len &&
                          (!vec || operand_stride % 4 == 0)),
             "zshmc_model_kick_drift_seg: bad operand_stride");
  const uintptr_t align =
      reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(p) |
      reinterpret_cast<uintptr_t>(grad_lik) | reinterpret_cast<uintptr_t>(operand) |
      reinterpret_cast<uintptr_t>(prior_mean) |
      reinterpret_cast<uintptr_t>(prior_logstd) | reinterpret_cast<uintptr_t>(mass);
  ZS_REQUIRE((align & 15) == 0,
             "zshmc_model_kick_drift_seg: buffers must be 16-B aligned");
  if (n_chains == 0) return ZSHMC_OK;
  const int64_t n_chunks = (n_data + kSegChunk - 1) / kSegChunk;
  SegArgs a{q, p, grad_lik, grad_stride, operand, operand_stride, seg_len,
            groups, prior_mean, mean_rows, prior_logstd, logstd_rows, mass,
            step_size_dev, step_size_host, kick_scale, drift_scale, lik_scale,
            n_chains, n_data, ld, workspace, n_chunks};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t n_work = n_chains * n_chunks;
  const int64_t cap = (int64_t)device_cu_count() * 16;
  hipLaunchKernelGGL(model_seg_step_kernel,
                     dim3((unsigned)(n_work < cap ? n_work : cap)), dim3(256), 0,
                     s, a);
  ZS_LAUNCH_CHECK("model_seg_step_kernel launch");
  if (lp_out || kinetic) {
    hipLaunchKernelGGL(model_seg_finish_kernel,
                       dim3((unsigned)((n_chains + 3) / 4)), dim3(256), 0, s,
                       workspace, n_chunks, ll_in, groups, lik_scale, n_chains,
                       lp_out, kinetic);
    ZS_LAUNCH_CHECK("model_seg_finish_kernel launch");
  }
  return ZSHMC_OK;
}
