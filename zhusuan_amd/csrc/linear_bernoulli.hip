// Fused dense-logit Bernoulli log-likelihood + gradient on the fp32 matrix
// cores of gfx950 (BASELINE config 3: Bayesian logistic regression).
//
//   logits[c, n] = sum_d W[c, d] * X[n, d]                 (user model: w @ X^T)
//   ll[c]   = sum_n  y_n*l - max(l,0) - log1p(exp(-|l|))    Bernoulli._log_prob,
//             reference zhusuan/distributions/univariate.py:398-403
//             (= -sigmoid_cross_entropy_with_logits) summed by group_ndims=1,
//             distributions/base.py:302-304
//   gW[c,:] = sum_n (y_n - sigmoid(l)) * X[n, :]            what tf.gradients
//             (hmc.py:430-432) yields through the matmul
//
// The reference materialises logits [C, N] (131 GB at config 3) and runs two
// GEMMs plus ~10 element-wise passes per gradient evaluation.  Here the two
// GEMMs are fused flash-attention style: a workgroup owns 64 chains, streams
// X in 32-row tiles through LDS (double buffered), computes the 32x64 logits
// tile on v_mfma_f32_32x32x2_f32 (exact fp32, so the 1 % acceptance parity is
// not at risk), applies the sigmoid residual in the accumulator registers,
// and feeds those registers straight back as the A operand of the second
// MFMA chain (S is computed transposed, so the C/D layout of the first GEMM
// IS the A layout of the second -- no shuffle, no LDS round trip).  Logits
// never leave registers.  Roofline: MFMA (fp32 157 TFLOP/s);
// 4*N*D*C flop per call.
#include "common.h"

namespace zshmc {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kMC = 64;  // chains per workgroup
constexpr int kNT = 32;  // data rows per tile

template <int FB>
struct VecF {};
template <>
struct VecF<1> {
  typedef float type;
};
template <>
struct VecF<2> {
  typedef float type __attribute__((ext_vector_type(2)));
};
template <>
struct VecF<4> {
  typedef f4 type;
};

template <int FB, typename V>
__device__ __forceinline__ float vget(const V& v, int t) {
  if constexpr (FB == 1)
    return v;
  else
    return v[t];
}

// D: padded feature count (64, 128 or 256).  4 waves: wave = (a, h), a =
// chain block (32 chains), h = feature half.
template <int D>
__global__ __launch_bounds__(256, 1) void linear_bernoulli_kernel(
    const float* __restrict__ W, const float* __restrict__ X,
    const float* __restrict__ y, int64_t C, int64_t N, int64_t ldw,
    int64_t ldx, float* __restrict__ ll, float* __restrict__ gW) {
  constexpr int LD = D + 4;        // padded LDS row: conflict-free b128 reads
  constexpr int HALF = D / 2;      // features per wave in each phase
  constexpr int FB = HALF / 32;    // 32-wide feature blocks per half (1,2,4)
  constexpr int X4 = kNT * D / 4 / 256;  // float4 per thread per X tile
  constexpr int W4 = kMC * D / 4 / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* __restrict__ sW = reinterpret_cast<float*>(smem);  // [kMC][LD]
  float* __restrict__ sX = sW + kMC * LD;                   // [2][kNT][LD]
  float* __restrict__ sY = sX + 2 * kNT * LD;               // [2][kNT]
  float* __restrict__ sEx = sY + 2 * kNT;                   // [4][16][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int a = wave >> 1, h = wave & 1;
  const int lo = lane & 31, hi = lane >> 5;
  const int64_t c0 = (int64_t)blockIdx.x * kMC;

  // ---- W tile -> LDS (rows clamped; stores masked at the end) -------------
#pragma unroll
  for (int i = 0; i < W4; ++i) {
    const int idx = i * 256 + tid;  // float4 index in [kMC][D/4]
    const int row = idx / (D / 4), c4 = idx % (D / 4);
    int64_t cr = c0 + row;
    cr = cr < C ? cr : C - 1;
    const f4 v = *reinterpret_cast<const f4*>(W + cr * ldw + c4 * 4);
    *reinterpret_cast<f4*>(sW + row * LD + c4 * 4) = v;
  }

  // ---- X tile prefetch registers ------------------------------------------
  f4 xr[X4];
  float yr = 0.f;
  auto load_tile = [&](int64_t n0) {
#pragma unroll
    for (int i = 0; i < X4; ++i) {
      const int idx = i * 256 + tid;
      const int row = idx / (D / 4), c4 = idx % (D / 4);
      int64_t nr = n0 + row;
      nr = nr < N ? nr : N - 1;
      xr[i] = *reinterpret_cast<const f4*>(X + nr * ldx + c4 * 4);
    }
    if (tid < kNT) {
      const int64_t nr = n0 + tid;
      yr = nr < N ? y[nr] : 0.f;
    }
  };
  auto store_tile = [&](int buf) {
    float* __restrict__ dst = sX + buf * kNT * LD;
#pragma unroll
    for (int i = 0; i < X4; ++i) {
      const int idx = i * 256 + tid;
      const int row = idx / (D / 4), c4 = idx % (D / 4);
      *reinterpret_cast<f4*>(dst + row * LD + c4 * 4) = xr[i];
    }
    if (tid < kNT) sY[buf * kNT + tid] = yr;
  };

  f16v G[FB];
#pragma unroll
  for (int t = 0; t < FB; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) G[t][r] = 0.f;
  float ll_lane = 0.f;

  const int64_t n_tiles = (N + kNT - 1) / kNT;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int64_t tile = 0; tile < n_tiles; ++tile) {
    const int buf = (int)(tile & 1);
    const float* __restrict__ xb = sX + buf * kNT * LD;
    if (tile + 1 < n_tiles) load_tile((tile + 1) * kNT);  // in flight

    // ---- phase 1: S'[n, i] = sum_d X[n,d] W[i,d] over this wave's K half --
    f16v S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll 4
    for (int kk = 0; kk < HALF / 8; ++kk) {
      const int d = h * HALF + kk * 8 + hi * 4;
      const f4 av = *reinterpret_cast<const f4*>(xb + lo * LD + d);
      const f4 bv = *reinterpret_cast<const f4*>(sW + (a * 32 + lo) * LD + d);
#pragma unroll
      for (int m = 0; m < 4; ++m)
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[m], S, 0, 0, 0);
    }
    // ---- exchange the K-half partials with the sibling wave ----------------
#pragma unroll
    for (int r = 0; r < 16; ++r) sEx[(wave * 16 + r) * 64 + lane] = S[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] += sEx[((wave ^ 1) * 16 + r) * 64 + lane];

    // ---- sigmoid residual in the accumulator layout -------------------------
    // lane holds chain i = a*32 + lo, rows n = (r&3) + 8*(r>>2) + 4*hi
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int nl = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const bool valid = tile * kNT + nl < N;
      const float s = S[r];
      const float yv = sY[buf * kNT + nl];
      const float e = __expf(-fabsf(s));
      const float inv = __builtin_amdgcn_rcpf(1.0f + e);
      const float sig = s >= 0.f ? inv : e * inv;
      const float lp = s * yv - fmaxf(s, 0.f) - log1pf(e);
      S[r] = valid ? yv - sig : 0.f;
      ll_lane += valid ? lp : 0.f;
    }

    // ---- phase 3: G[i, f] += sum_n R'[n, i] X[n, f] over this feature half -
    // A operand = the residual registers themselves (k-slot = lane half)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int nl = (r & 3) + 8 * (r >> 2) + 4 * hi;
      typedef typename VecF<FB>::type V;
      const V xv = *reinterpret_cast<const V*>(xb + nl * LD + h * HALF + lo * FB);
#pragma unroll
      for (int t = 0; t < FB; ++t)
        G[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(S[r], vget<FB>(xv, t), G[t],
                                                    0, 0, 0);
    }

    // ---- publish the prefetched tile into the other buffer -----------------
    if (tile + 1 < n_tiles) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------
  // G[t][r]: chain = c0 + a*32 + (r&3) + 8*(r>>2) + 4*hi, feature =
  // h*HALF + lo*FB + t
  if (gW) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t chain = c0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (chain < C) {
#pragma unroll
        for (int t = 0; t < FB; ++t)
          gW[chain * ldw + h * HALF + lo * FB + t] = G[t][r];
      }
    }
  }
  // every lane's ll covers its 16 rows per tile; the other 16 sit in lane^32
  const float ll_tot = ll_lane + __shfl_xor(ll_lane, 32, 64);
  if (h == 0 && hi == 0) {
    const int64_t chain = c0 + a * 32 + lo;
    if (chain < C) ll[chain] = ll_tot;
  }
}

template <int D>
static int launch_lb(const float* W, const float* X, const float* y, int64_t C,
                     int64_t N, int64_t ldw, int64_t ldx, float* ll, float* gW,
                     hipStream_t s) {
  constexpr int LD = D + 4;
  const size_t lds = (size_t)(kMC * LD + 2 * kNT * LD + 2 * kNT + 4 * 16 * 64) *
                     sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(linear_bernoulli_kernel<D>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(LDS)");
    attr_set = true;
  }
  const int grid = (int)((C + kMC - 1) / kMC);
  hipLaunchKernelGGL(linear_bernoulli_kernel<D>, dim3(grid), dim3(256), lds, s,
                     W, X, y, C, N, ldw, ldx, ll, gW);
  ZS_LAUNCH_CHECK("linear_bernoulli_kernel launch");
  return ZSHMC_OK;
}

}  // namespace zshmc

using namespace zshmc;

extern "C" int zshmc_linear_bernoulli_log_lik(const float* W, const float* X,
                                              const float* y, int64_t n_chains,
                                              int64_t n_rows, int64_t n_features,
                                              float* log_lik, float* grad_w,
                                              void* stream) {
  if (n_chains == 0) return ZSHMC_OK;
  ZS_REQUIRE(W && X && y && log_lik, "zshmc_linear_bernoulli_log_lik: null pointer");
  ZS_REQUIRE(n_chains > 0 && n_rows > 0,
             "zshmc_linear_bernoulli_log_lik: bad shape");
  ZS_REQUIRE(n_features == 64 || n_features == 128 || n_features == 256,
             "zshmc_linear_bernoulli_log_lik: n_features must be 64, 128 or 256 "
             "(zero-pad W and X), got %lld", (long long)n_features);
  ZS_REQUIRE((reinterpret_cast<uintptr_t>(W) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
                 (!grad_w || (reinterpret_cast<uintptr_t>(grad_w) & 3) == 0),
             "zshmc_linear_bernoulli_log_lik: W and X must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (n_features) {
    case 64:
      return launch_lb<64>(W, X, y, n_chains, n_rows, 64, 64, log_lik, grad_w, s);
    case 128:
      return launch_lb<128>(W, X, y, n_chains, n_rows, 128, 128, log_lik, grad_w, s);
    default:
      return launch_lb<256>(W, X, y, n_chains, n_rows, 256, 256, log_lik, grad_w, s);
  }
}
