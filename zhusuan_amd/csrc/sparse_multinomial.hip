// The mixture-multinomial likelihood of the topic model and its gradient, ROW
// BY ROW over each row's own words -- the small-problem form.
//
//   ll[r]      = sum_v x[doc(r), v] log S[r, v],   S[r, v] = theta[r, :] . phi[:, v]
//   g_theta[r] = sum_v (x[doc(r), v] / S[r, v]) phi[:, v]
//   (UnnormalizedMultinomial._log_prob, multivariate.py:435-443, over
//    log(softmax(eta) @ phi), lntm_mcem.py:33-48; what tf.gradients gives,
//    hmc.py:430-432)
//
// A word the document does not contain contributes exactly nothing (x = 0),
// and a bag of words is sparse: ~960 of 12 419 words per document in
// lntm_mcem.py's corpus.  The MFMA kernels (csrc/lb_body.h, csrc/b3_kernel.h)
// can only use that where a workgroup's chains share a document; the
// reference's own loop (lntm_mcem.py:62-70,157-182) is ONE chain x a minibatch
// of 100 documents -- 100 rows, every one with its own word list, a likelihood
// launch of 17 us on the matrix cores whatever the arithmetic (DESIGN 3.6:
// the serial part every workgroup pays, not the flops).  Here a row costs its
// ~960 words x K multiply-adds on the vector ALU: 8 % of the dense flops, no
// tile pipeline to fill, phi^T (6 MB) read from L2.
//
// Grid (rows, S): workgroup (r, y) takes slice y of document doc(r) = r %
// count_rows's word list (the padded CSR of zshmc_linear_multinomial_log_lik_
// bf16x3_sparse: counts, rows of phi^T, [count_rows + 1] offsets, slices of
// whole 32-word tiles).  A word is a group of W / 4 lanes (16 bytes of its
// phi^T row each; W = 64 / 128: four / two words per wave step), theta's row
// in registers, the dot product by shuffles inside the group, the gradient
// accumulated per lane, four word-steps of loads in flight.  Partials of a
// split launch in the dense kernels' workspace layout (csrc/linear_bernoulli.
// hip lb_reduce_splits: [S][C] log-likelihoods, [S][C][ldw] gradients), so
// that the leapfrog step adds them itself (csrc/model_step.h).
// Exact float32 arithmetic (FMA), fixed summation order: deterministic.
// Bound: L2 gather rate / launch latency; not an MFMA kernel.
#include "common.h"

namespace zshmc {

typedef float s4 __attribute__((ext_vector_type(4)));

template <int RW>
__device__ __forceinline__ float sp_group_sum(float v) {
#pragma unroll
  for (int off = RW / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// W: padded topic count (row stride of phi^T and of the gradient)
template <int W>
__global__ __launch_bounds__(256) void sparse_multinomial_kernel(
    const float* __restrict__ theta, int64_t ldw,
    const float* __restrict__ phi_t, const float* __restrict__ vals,
    const int32_t* __restrict__ rows, const int64_t* __restrict__ off,
    int64_t count_rows, int64_t C, float* __restrict__ ll,
    float* __restrict__ gW) {
  constexpr int LW = W / 4;                         // lanes that hold data
  constexpr int RW = LW <= 16 ? 16 : (LW <= 32 ? 32 : 64);   // lanes per word
  constexpr int G = 64 / RW;                        // words per wave step
  constexpr int kBatch = 4;                         // word-steps in flight
  __shared__ s4 sh_g[4][64];
  __shared__ float sh_l[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane / RW, l = lane % RW;
  const bool active = l < LW;
  const int64_t r = blockIdx.x;
  const int64_t doc = r % count_rows;
  const int64_t o0 = off[doc];
  const int len = (int)(off[doc + 1] - o0);         // a multiple of 32
  const int S = gridDim.y;
  const int chunk = ((len / 32 + S - 1) / S) * 32;
  const int w0 = blockIdx.y * chunk;
  const int w1 = w0 + chunk < len ? w0 + chunk : len;
  const s4 zero = s4{0.f, 0.f, 0.f, 0.f};
  const s4 th = active ? *reinterpret_cast<const s4*>(theta + r * ldw + 4 * l)
                       : zero;
  const float* __restrict__ xs = vals + o0;
  const int32_t* __restrict__ is = rows + o0;
  s4 acc = zero;
  float lsum = 0.f;
  for (int w = w0 + wave * G + grp; w < w1; w += 4 * G * kBatch) {
    float x[kBatch];
    s4 ph[kBatch];
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      const int wb = w + b * 4 * G;
      const bool on = wb < w1;
      x[b] = on ? xs[wb] : 0.f;
      const int idx = on ? is[wb] : 0;
      ph[b] = active ? *reinterpret_cast<const s4*>(phi_t + (int64_t)idx * W +
                                                    4 * l)
                     : zero;
    }
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      float d = (th[0] * ph[b][0] + th[1] * ph[b][1]) +
                (th[2] * ph[b][2] + th[3] * ph[b][3]);
      d = sp_group_sum<RW>(d);
      // the residual and the term of the MFMA kernels (csrc/lb_ops.h, OP 1)
      const bool on = x[b] != 0.f;
      const float rr = on ? x[b] * __builtin_amdgcn_rcpf(d) : 0.f;
      acc += rr * ph[b];
      if (ll && on)
        lsum += x[b] * (0.6931471805599453f * __builtin_amdgcn_logf(d));
    }
  }
  // the wave's word groups, then the four waves (fixed order)
#pragma unroll
  for (int o = RW; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
    lsum += __shfl_xor(lsum, o, 64);
  }
  sh_g[wave][lane] = acc;
  if (lane == 0) sh_l[wave][0] = lsum;
  __syncthreads();
  if (wave != 0) return;
  float* __restrict__ g_out = gW + ((int64_t)blockIdx.y * C + r) * ldw;
  if (lane < LW) {
    const s4 g = (sh_g[0][lane] + sh_g[1][lane]) + (sh_g[2][lane] + sh_g[3][lane]);
    *reinterpret_cast<s4*>(g_out + 4 * lane) = g;
  }
  if (ll && lane == 0)
    ll[(int64_t)blockIdx.y * C + r] =
        (sh_l[0][0] + sh_l[1][0]) + (sh_l[2][0] + sh_l[3][0]);
}

// csrc/linear_bernoulli.hip
int lb_reduce_splits(const float* ws, int64_t C, int64_t ldw, int S, float* ll,
                     float* gW, hipStream_t s);

}  // namespace zshmc

using namespace zshmc;

extern "C" int zshmc_sparse_multinomial_log_lik(
    const float* theta, const float* phi_t, const float* counts_csr,
    const int32_t* row_index, const int64_t* doc_offsets, int64_t count_rows,
    int64_t n_rows, int64_t n_vocab, int64_t n_topics, float* log_lik,
    float* grad_theta, int n_splits, float* workspace, void* stream) {
  if (n_rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(theta && phi_t && counts_csr && row_index && doc_offsets &&
                 grad_theta,
             "zshmc_sparse_multinomial_log_lik: null pointer");
  ZS_REQUIRE(n_rows > 0 && n_vocab > 0 && count_rows > 0 &&
                 n_rows % count_rows == 0 && n_rows < (1ll << 31) &&
                 (n_topics == 64 || n_topics == 128 || n_topics == 192 ||
                  n_topics == 256),
             "zshmc_sparse_multinomial_log_lik: bad shape (n_topics 64 / 128 / "
             "192 / 256 padded columns)");
  ZS_REQUIRE(((reinterpret_cast<uintptr_t>(theta) |
               reinterpret_cast<uintptr_t>(phi_t) |
               reinterpret_cast<uintptr_t>(grad_theta)) & 15) == 0,
             "zshmc_sparse_multinomial_log_lik: theta, phi^T and the gradient "
             "must be 16-byte aligned");
  ZS_REQUIRE(n_splits >= 1 && n_splits <= 256 && (n_splits == 1 || workspace),
             "zshmc_sparse_multinomial_log_lik: 1 <= n_splits <= 256 and a "
             "workspace of n_splits*n_rows*(n_topics+1) floats when > 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int S = (n_splits > 1 && workspace) ? n_splits : 1;
  float* ll_out = S > 1 ? (log_lik ? workspace : nullptr) : log_lik;
  float* g_out = S > 1 ? workspace + (int64_t)S * n_rows : grad_theta;
  const dim3 grid((unsigned)n_rows, (unsigned)S);
#define ZS_SPM(W)                                                            \
  hipLaunchKernelGGL(sparse_multinomial_kernel<W>, grid, dim3(256), 0, s,    \
                     theta, n_topics, phi_t, counts_csr, row_index,          \
                     doc_offsets, count_rows, n_rows, ll_out, g_out)
  switch (n_topics) {
    case 64: ZS_SPM(64); break;
    case 128: ZS_SPM(128); break;
    case 192: ZS_SPM(192); break;
    default: ZS_SPM(256); break;
  }
#undef ZS_SPM
  ZS_LAUNCH_CHECK("sparse_multinomial_kernel launch");
  if (S > 1)
    return lb_reduce_splits(workspace, n_rows, n_topics, S, log_lik, grad_theta,
                            s);
  return ZSHMC_OK;
}
