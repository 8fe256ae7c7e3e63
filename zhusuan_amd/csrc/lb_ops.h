// Element-wise stage shared by the two fused two-GEMM likelihood kernels
// (csrc/linear_bernoulli.hip, csrc/linear_bernoulli_wide.hip) for OP = 2, the
// dense-logit Categorical (softmax regression):
//
//   logits[c, n, k] = sum_f X[n, f] * w[c, k, f]
//   ll[c]  = sum_n  logits[c, n, y_n] - logsumexp_k logits[c, n, :]
//            Categorical._log_prob = -sparse_softmax_cross_entropy_with_logits,
//            reference zhusuan/distributions/univariate.py:496-548, summed over
//            the data rows by group_ndims = 1 (distributions/base.py:302-304)
//   d ll / d logits[c, n, k] = [k == y_n] - softmax_k(logits[c, n, :])
//            (what tf.gradients, hmc.py:430-432, yields through the op)
//
// The kernels' "chain rows" are the (chain, class) pairs: row c * G + k of the
// W operand, G = the class count rounded up to a power of two <= 32, so that
// the classes of one chain sit in G CONSECUTIVE LANES of a 32-lane half of the
// logits accumulator (lane -> column of the MFMA tile).  The softmax over the
// classes is then a butterfly over those lanes on the DPP data path
// (quad_perm / row_half_mirror / row_mirror; the 16 <-> 16 step of G = 32
// through ds_swizzle), no LDS round trip and no barrier.  Padding classes
// (k >= n_classes) are kept out of the max and the sum and get a zero
// residual; their W rows are zero and stay zero.
#pragma once
#include "common.h"

namespace zshmc {

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL,
                                         0xf, 0xf, true));
}

// lane ^ 16 inside each 32-lane half: ds_swizzle bit mode, and 0x1f, xor 0x10
__device__ __forceinline__ float swizzle_xor16(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401f));
}

// all-reduce over the aligned group of 2^gl consecutive lanes (gl = 0..5,
// wave-uniform) a lane belongs to
template <bool MAX>
__device__ __forceinline__ float lane_group_allreduce(float v, int gl) {
#define ZS_COMBINE(o) v = MAX ? fmaxf(v, (o)) : v + (o)
  if (gl >= 1) ZS_COMBINE(dpp_move<0xb1>(v));   // quad_perm [1,0,3,2]
  if (gl >= 2) ZS_COMBINE(dpp_move<0x4e>(v));   // quad_perm [2,3,0,1]
  if (gl >= 3) ZS_COMBINE(dpp_move<0x141>(v));  // row_half_mirror
  if (gl >= 4) ZS_COMBINE(dpp_move<0x140>(v));  // row_mirror
  if (gl >= 5) ZS_COMBINE(swizzle_xor16(v));
#undef ZS_COMBINE
  return v;
}

struct CatLane {
  int gl;          // log2 of the class stride G
  float kcls;      // this lane's class (column % G) as a float
  bool cls_on;     // kcls < n_classes
};

__device__ __forceinline__ CatLane cat_lane(int column, int n_classes,
                                            int gl) {
  const int k = column & ((1 << gl) - 1);
  return CatLane{gl, (float)k, k < n_classes};
}

// logit `sv` of (chain, class) column for data row n with label `label`
// (a float holding 0 .. n_classes-1): returns the residual
// [k == label] - softmax_k and adds the row's log-likelihood term to `lp` on
// the label's lane.  Every lane of the wave must call it (cross-lane ops).
template <bool LL = true>
__device__ __forceinline__ float categorical_residual(float sv, float label,
                                                      const CatLane& c,
                                                      bool valid, float& lp) {
  const float sm = c.cls_on ? sv : -INFINITY;
  const float m = lane_group_allreduce<true>(sm, c.gl);
  const float d = sv - m;
  const float e =
      c.cls_on ? __builtin_amdgcn_exp2f(1.4426950408889634f * d) : 0.f;
  const float z = lane_group_allreduce<false>(e, c.gl);   // in [1, G]
  const float p = e * __builtin_amdgcn_rcpf(z);
  const bool hit = valid && label == c.kcls;
  if (LL) lp += hit ? d - 0.6931471805599453f * __builtin_amdgcn_logf(z) : 0.f;
  return (valid && c.cls_on) ? (hit ? 1.0f : 0.f) - p : 0.f;
}


// The element-wise stage between the two GEMMs of the fused likelihood
// kernels, on one logit `sv` of (chain, data row): returns the residual
// d log_lik / d logit and adds the row's log-likelihood term to `ll_tile`.
//   OP 0  Bernoulli._log_prob (univariate.py:398-403): l*y - max(l,0) -
//         log1p(exp(-|l|)), d/dl = y - sigmoid(l); `aux` = y.
//         log1p(e) = ln2*log2(1+e) with e in (0,1] is good to ~1e-7 absolute.
//   OP 1  UnnormalizedMultinomial over a mixture (multivariate.py:435-443,
//         normalize_logits = False): x*log(S), d/dS = x / S; `aux` = the
//         count x (0 contributes nothing, also where the product underflows).
//   OP 2  Categorical (csrc/lb_ops.h above); `aux` = the label.
// LL = false: the residual alone (sigmoid as 1 / (1 + 2^(-l log2 e)): l ->
// -inf gives 1 / inf = 0, l -> +inf 1 / 1).  `valid` = false: a row past N.
template <int OP, bool LL>
__device__ __forceinline__ float lb_residual(float sv, float aux,
                                             const CatLane& cat, bool valid,
                                             float& ll_tile) {
  if constexpr (OP == 0) {
    if constexpr (LL) {
      const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(sv));
      const float t1 = 1.0f + e;
      const float inv = __builtin_amdgcn_rcpf(t1);  // sigmoid(|l|) >= 1/2
      // sigmoid(l) = 1/2 + copysign(inv - 1/2, l): one v_bfi instead of a
      // compare + select
      const float sig = 0.5f + __builtin_copysignf(inv - 0.5f, sv);
      const float lp = sv * aux - fmaxf(sv, 0.f) -
                       0.6931471805599453f * __builtin_amdgcn_logf(t1);
      ll_tile += valid ? lp : 0.f;
      return valid ? aux - sig : 0.f;
    } else {
      const float sig = __builtin_amdgcn_rcpf(
          1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * sv));
      return valid ? aux - sig : 0.f;
    }
  } else if constexpr (OP == 2) {
    return categorical_residual<LL>(sv, aux, cat, valid, ll_tile);
  } else {
    const bool on = valid && aux != 0.f;
    if constexpr (LL) {
      const float lp = aux * (0.6931471805599453f * __builtin_amdgcn_logf(sv));
      ll_tile += on ? lp : 0.f;
    }
    return on ? aux * __builtin_amdgcn_rcpf(sv) : 0.f;
  }
}

}  // namespace zshmc
