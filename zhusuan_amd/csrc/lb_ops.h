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

// One butterfly step as ONE instruction: v <- op(v, dpp(v)) with the DPP
// modifier on the operand (hipcc's fmaxf on a moved value is v_mov_b32_dpp + a
// canonicalising v_max + the v_max: three).  A DPP read of a VGPR needs two
// wait states behind a VALU write of it and hipcc does not look inside asm:
// `s_nop 1` in front (free next to the MFMAs around it).
#define ZS_DPP_STEP(NAME, OPC, CTRL)                                          \
  __device__ __forceinline__ float NAME(float v) {                            \
    asm("s_nop 1\n\t" OPC " %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf"    \
        : "+v"(v));                                                           \
    return v;                                                                 \
  }
ZS_DPP_STEP(dpp_max_q1, "v_max_f32_dpp", "quad_perm:[1,0,3,2]")
ZS_DPP_STEP(dpp_max_q2, "v_max_f32_dpp", "quad_perm:[2,3,0,1]")
ZS_DPP_STEP(dpp_max_hm, "v_max_f32_dpp", "row_half_mirror")
ZS_DPP_STEP(dpp_max_rm, "v_max_f32_dpp", "row_mirror")
ZS_DPP_STEP(dpp_add_q1, "v_add_f32_dpp", "quad_perm:[1,0,3,2]")
ZS_DPP_STEP(dpp_add_q2, "v_add_f32_dpp", "quad_perm:[2,3,0,1]")
ZS_DPP_STEP(dpp_add_hm, "v_add_f32_dpp", "row_half_mirror")
ZS_DPP_STEP(dpp_add_rm, "v_add_f32_dpp", "row_mirror")
#undef ZS_DPP_STEP

// all-reduce over the aligned group of 2^GL consecutive lanes a lane belongs
// to, GL = 0..5 a compile-time constant (straight-line code)
template <int GL, bool MAX>
__device__ __forceinline__ float lane_group_allreduce_c(float v) {
  if constexpr (GL >= 1) v = MAX ? dpp_max_q1(v) : dpp_add_q1(v);
  if constexpr (GL >= 2) v = MAX ? dpp_max_q2(v) : dpp_add_q2(v);
  if constexpr (GL >= 3) v = MAX ? dpp_max_hm(v) : dpp_add_hm(v);
  if constexpr (GL >= 4) v = MAX ? dpp_max_rm(v) : dpp_add_rm(v);
  if constexpr (GL >= 5) {
    const float o = swizzle_xor16(v);
    v = MAX ? fmaxf(v, o) : v + o;
  }
  return v;
}

// logit `sv` of (chain, class) column for data row n with label `label`
// (a float holding 0 .. n_classes-1): returns the residual
// [k == label] - softmax_k and adds the row's log-likelihood term to `lp` on
// the label's lane.  Every lane of the wave must call it (cross-lane ops).
// GL = log2 of the class stride, compile-time (categorical_residual_n picks it).
template <int GL, bool LL>
__device__ __forceinline__ float categorical_residual_c(float sv, float label,
                                                        const CatLane& c,
                                                        bool valid, float& lp) {
  const float sm = c.cls_on ? sv : -INFINITY;
  const float m = lane_group_allreduce_c<GL, true>(sm);
  const float d = sv - m;
  const float e =
      c.cls_on ? __builtin_amdgcn_exp2f(1.4426950408889634f * d) : 0.f;
  const float z = lane_group_allreduce_c<GL, false>(e);   // in [1, G]
  const float p = e * __builtin_amdgcn_rcpf(z);
  const bool hit = valid && label == c.kcls;
  if (LL) lp += hit ? d - 0.6931471805599453f * __builtin_amdgcn_logf(z) : 0.f;
  return (valid && c.cls_on) ? (hit ? 1.0f : 0.f) - p : 0.f;
}

// N logits at once (the rows of one element-wise slot of the kernels): ONE
// wave-uniform branch on the class stride for all of them, then straight-line
// code in which the N independent butterflies fill each other's DPP wait
// states.  (With the stride a run-time value inside the per-element code hipcc
// emitted a scalar branch per butterfly step: 160 branches per 64-row tile.)
template <bool LL, int N>
__device__ __forceinline__ void categorical_residual_n(float (&sv)[N],
                                                       const float (&label)[N],
                                                       const CatLane& c,
                                                       const bool (&valid)[N],
                                                       float& lp) {
#define ZS_CAT_CASE(G)                                                        \
  case G:                                                                     \
    _Pragma("unroll") for (int i = 0; i < N; ++i) sv[i] =                     \
        categorical_residual_c<G, LL>(sv[i], label[i], c, valid[i], lp);      \
    break;
  switch (c.gl) {
    ZS_CAT_CASE(0)
    ZS_CAT_CASE(1)
    ZS_CAT_CASE(2)
    ZS_CAT_CASE(3)
    ZS_CAT_CASE(4)
    default:
      _Pragma("unroll") for (int i = 0; i < N; ++i) sv[i] =
          categorical_residual_c<5, LL>(sv[i], label[i], c, valid[i], lp);
  }
#undef ZS_CAT_CASE
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
//   OP 2  Categorical (above; the kernels take categorical_residual_n for a
//         whole slot of rows instead); `aux` = the label.
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
    // (one element: the kernels call categorical_residual_n on a slot's rows)
    float v[1] = {sv};
    const float l[1] = {aux};
    const bool ok[1] = {valid};
    categorical_residual_n<LL, 1>(v, l, cat, ok, ll_tile);
    return v[0];
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
