// Two-parameter continuous families beside Normal on the HMC path (SURVEY.md
// section 8f-4): Laplace, Gamma, InverseGamma, Beta -- log_prob with the
// group_ndims row sum fused, analytic gradients w.r.t. the value and both
// parameters, and sampling, for gfx950.  Reference closed forms
// (zhusuan/distributions/univariate.py):
//   Laplace(loc, scale)      -log 2 - log s - |x - l| / s              :1268-1275
//   Gamma(alpha, beta)       a log b - lgamma a + (a-1) log x - b x    :735-748
//   InverseGamma(alpha,beta) a log b - lgamma a - (a+1) log x - b / x  :1145-1157
//   Beta(alpha, beta)        (a-1) log x + (b-1) log(1-x)
//                            - (lgamma a + lgamma b - lgamma(a+b))     :834-853
// with base.py:302-304's reduce_sum over the last group_ndims axes.
// Sampling (:725-727, :826-832, :1140-1143, :1246-1265): tf.random_gamma is
// restated as Marsaglia-Tsang (2000) squeeze rejection on the Philox stream
// (alpha < 1 boosted by U^(1/alpha)); Laplace as the reference's inverse CDF
// loc - scale * sign(u) * log1p(-|u|), u in (-1, 1).
// HBM-bound element-wise passes, same broadcast modes as distributions.hip.
#include "common.h"
#include "philox.h"

namespace zshmc {

enum { kLaplace = 0, kGamma = 1, kInvGamma = 2, kBeta = 3 };

__device__ __forceinline__ float fetch2(const float* __restrict__ p, int mode,
                                        int64_t i, int64_t col) {
  return mode == ZSHMC_BCAST_FULL ? p[i] : (mode == ZSHMC_BCAST_ROW ? p[col] : p[0]);
}

// digamma: recurrence up to x >= 6, then the asymptotic series
__device__ __forceinline__ float digammaf_(float x) {
  float acc = 0.f;
  while (x < 6.f) {
    acc -= 1.0f / x;
    x += 1.0f;
  }
  const float r = 1.0f / x, r2 = r * r;
  return acc + logf(x) - 0.5f * r -
         r2 * (1.0f / 12.0f - r2 * (1.0f / 120.0f - r2 * (1.0f / 252.0f)));
}

template <int KIND>
__device__ __forceinline__ float uni2_lp(float x, float a, float b) {
  if (KIND == kLaplace) return -0.6931471805599453f - logf(b) - fabsf(x - a) / b;
  if (KIND == kGamma)
    return a * logf(b) - lgammaf(a) + (a - 1.0f) * logf(x) - b * x;
  if (KIND == kInvGamma)
    return a * logf(b) - lgammaf(a) - (a + 1.0f) * logf(x) - b / x;
  return (a - 1.0f) * logf(x) + (b - 1.0f) * logf(1.0f - x) -
         (lgammaf(a) + lgammaf(b) - lgammaf(a + b));
}

template <int KIND>
__device__ __forceinline__ void uni2_grad(float x, float a, float b, float& dx,
                                          float& da, float& db) {
  if (KIND == kLaplace) {
    const float d = x - a;
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    dx = -sgn / b;
    da = sgn / b;
    db = -1.0f / b + fabsf(d) / (b * b);
  } else if (KIND == kGamma) {
    dx = (a - 1.0f) / x - b;
    da = logf(b) - digammaf_(a) + logf(x);
    db = a / b - x;
  } else if (KIND == kInvGamma) {
    dx = -(a + 1.0f) / x + b / (x * x);
    da = logf(b) - digammaf_(a) - logf(x);
    db = a / b - 1.0f / x;
  } else {
    const float dab = digammaf_(a + b);
    dx = (a - 1.0f) / x - (b - 1.0f) / (1.0f - x);
    da = logf(x) - digammaf_(a) + dab;
    db = logf(1.0f - x) - digammaf_(b) + dab;
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void uni2_elementwise_kernel(
    const float* __restrict__ x, const float* __restrict__ a,
    const float* __restrict__ b, float* __restrict__ out, int64_t n,
    int64_t cols, int mode_a, int mode_b) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t col = i % cols;
    out[i] = uni2_lp<KIND>(x[i], fetch2(a, mode_a, i, col), fetch2(b, mode_b, i, col));
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void uni2_rowsum_kernel(
    const float* __restrict__ x, const float* __restrict__ a,
    const float* __restrict__ b, float* __restrict__ out, int64_t rows,
    int64_t cols, int mode_a, int mode_b) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / 64);
  for (int64_t r = wave; r < rows; r += n_waves) {
    float s = 0.f;
    for (int64_t col = lane; col < cols; col += 64) {
      const int64_t i = r * cols + col;
      s += uni2_lp<KIND>(x[i], fetch2(a, mode_a, i, col), fetch2(b, mode_b, i, col));
    }
    s = group_sum<64>(s);
    if (lane == 0) out[r] = s;
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void uni2_grad_kernel(
    const float* __restrict__ x, const float* __restrict__ a,
    const float* __restrict__ b, const float* __restrict__ gout,
    float* __restrict__ gx, float* __restrict__ ga, float* __restrict__ gb,
    int64_t n, int64_t cols, int mode_a, int mode_b, int reduce_cols) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t col = i % cols;
    const float g = reduce_cols ? gout[i / cols] : gout[i];
    float dx, da, db;
    uni2_grad<KIND>(x[i], fetch2(a, mode_a, i, col), fetch2(b, mode_b, i, col),
                    dx, da, db);
    if (gx) gx[i] = g * dx;
    if (ga) ga[i] = g * da;
    if (gb) gb[i] = g * db;
  }
}

// ---- sampling ---------------------------------------------------------------
// Standard Gamma(alpha, 1), Marsaglia & Tsang: d = alpha' - 1/3, c = 1/sqrt(9d),
// v = (1 + c z)^3, accept if log u < z^2/2 + d - d v + d log v.  Attempt t of
// element i draws (z, u) from Philox counter (i lo, i hi, offset, STREAM_DIST
// | (sub << 4) | (t << 8)); sub separates the two Gammas of a Beta draw.
__device__ __forceinline__ float std_gamma(float alpha, uint64_t i,
                                           uint32_t offset, uint32_t sub,
                                           uint32_t k0, uint32_t k1) {
  const float a1 = alpha < 1.0f ? alpha + 1.0f : alpha;
  const float d = a1 - 1.0f / 3.0f;
  const float c = 1.0f / sqrtf(9.0f * d);
  float out = d;  // fallback after 64 rejections (probability < 1e-80)
  float u_boost = 1.0f;
  for (uint32_t t = 0; t < 64; ++t) {
    const U4 r = philox4x32((uint32_t)i, (uint32_t)(i >> 32), offset,
                               kStreamDist | (sub << 4) | (t << 8), k0, k1);
    float z, z_unused;
    box_muller(r.x, r.y, z, z_unused);
    const float u = u01_open_low(r.z);
    if (t == 0) u_boost = u01_open_low(r.w);
    const float v1 = 1.0f + c * z;
    if (v1 <= 0.f) continue;
    const float v = v1 * v1 * v1;
    if (logf(u) < 0.5f * z * z + d - d * v + d * logf(v)) {
      out = d * v;
      break;
    }
  }
  if (alpha < 1.0f) out *= powf(u_boost, 1.0f / alpha);
  return out;
}

template <int KIND>
__global__ __launch_bounds__(256) void uni2_sample_kernel(
    float* __restrict__ out, const float* __restrict__ a,
    const float* __restrict__ b, int64_t n, int64_t inner, int mode_a,
    int mode_b, uint32_t k0, uint32_t k1, uint32_t offset) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = i % inner;  // index inside one sample of batch_shape
    const float pa = mode_a == ZSHMC_BCAST_SCALAR ? a[0] : a[j];
    const float pb = mode_b == ZSHMC_BCAST_SCALAR ? b[0] : b[j];
    if (KIND == kLaplace) {
      const U4 r = philox4x32((uint32_t)i, (uint32_t)((uint64_t)i >> 32),
                                 offset, kStreamDist, k0, k1);
      // u in (-1, 1): sign from one word, magnitude in [0, 1) from another
      const float mag = u01(r.x);
      const float sgn = (r.y & 1u) ? 1.0f : -1.0f;
      out[i] = pa - pb * sgn * log1pf(-mag);
    } else if (KIND == kGamma) {
      out[i] = std_gamma(pa, (uint64_t)i, offset, 0, k0, k1) / pb;
    } else if (KIND == kInvGamma) {
      out[i] = pb / std_gamma(pa, (uint64_t)i, offset, 0, k0, k1);
    } else {
      const float gx = std_gamma(pa, (uint64_t)i, offset, 0, k0, k1);
      const float gy = std_gamma(pb, (uint64_t)i, offset, 1, k0, k1);
      out[i] = gx / (gx + gy);
    }
  }
}

static inline int flat_grid2(int64_t n) {
  const int64_t need = (n + 255) / 256;
  const int64_t cap = (int64_t)device_cu_count() * 16;
  const int64_t g = need < cap ? need : cap;
  return (int)(g > 0 ? g : 1);
}
static inline int row_grid2(int64_t rows) {
  const int64_t need = (rows + 3) / 4;
  const int64_t cap = (int64_t)device_cu_count() * 8;
  const int64_t g = need < cap ? need : cap;
  return (int)(g > 0 ? g : 1);
}
static inline bool mode_ok2(int m) {
  return m == ZSHMC_BCAST_FULL || m == ZSHMC_BCAST_ROW || m == ZSHMC_BCAST_SCALAR;
}

template <int KIND>
static void launch_fwd(const float* x, const float* a, const float* b, float* out,
                       int64_t rows, int64_t cols, int ma, int mb, int reduce_cols,
                       hipStream_t s) {
  if (reduce_cols)
    hipLaunchKernelGGL(uni2_rowsum_kernel<KIND>, dim3(row_grid2(rows)), dim3(256),
                       0, s, x, a, b, out, rows, cols, ma, mb);
  else
    hipLaunchKernelGGL(uni2_elementwise_kernel<KIND>, dim3(flat_grid2(rows * cols)),
                       dim3(256), 0, s, x, a, b, out, rows * cols, cols, ma, mb);
}

template <int KIND>
static void launch_bwd(const float* x, const float* a, const float* b,
                       const float* gout, float* gx, float* ga, float* gb,
                       int64_t rows, int64_t cols, int ma, int mb, int reduce_cols,
                       hipStream_t s) {
  hipLaunchKernelGGL(uni2_grad_kernel<KIND>, dim3(flat_grid2(rows * cols)),
                     dim3(256), 0, s, x, a, b, gout, gx, ga, gb, rows * cols, cols,
                     ma, mb, reduce_cols);
}

template <int KIND>
static void launch_sample(float* out, const float* a, const float* b, int64_t n,
                          int64_t inner, int ma, int mb, uint64_t seed,
                          uint32_t offset, hipStream_t s) {
  hipLaunchKernelGGL(uni2_sample_kernel<KIND>, dim3(flat_grid2(n)), dim3(256), 0, s,
                     out, a, b, n, inner, ma, mb, (uint32_t)(seed & 0xFFFFFFFFull),
                     (uint32_t)(seed >> 32), offset);
}

}  // namespace zshmc

using namespace zshmc;

#define ZS_KIND_SWITCH(kind, CALL)  \
  switch (kind) {                   \
    case kLaplace: CALL(kLaplace); break;   \
    case kGamma: CALL(kGamma); break;       \
    case kInvGamma: CALL(kInvGamma); break; \
    default: CALL(kBeta); break;            \
  }

extern "C" int zshmc_uni2_log_prob(int kind, const float* x, const float* a,
                                   const float* b, float* out, int64_t rows,
                                   int64_t cols, int a_bcast, int b_bcast,
                                   int reduce_cols, void* stream) {
  if (rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(kind >= 0 && kind <= 3, "zshmc_uni2_log_prob: unknown family %d", kind);
  ZS_REQUIRE(x && a && b && out, "zshmc_uni2_log_prob: null pointer");
  ZS_REQUIRE(rows >= 0 && cols >= 1, "zshmc_uni2_log_prob: bad shape");
  ZS_REQUIRE(mode_ok2(a_bcast) && mode_ok2(b_bcast),
             "zshmc_uni2_log_prob: bad broadcast mode");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define CALL(K) launch_fwd<K>(x, a, b, out, rows, cols, a_bcast, b_bcast, reduce_cols, s)
  ZS_KIND_SWITCH(kind, CALL)
#undef CALL
  ZS_LAUNCH_CHECK("uni2 log_prob launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_uni2_log_prob_grad(int kind, const float* x, const float* a,
                                        const float* b, const float* gout,
                                        float* gx, float* ga, float* gb,
                                        int64_t rows, int64_t cols, int a_bcast,
                                        int b_bcast, int reduce_cols,
                                        void* stream) {
  if (rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(kind >= 0 && kind <= 3, "zshmc_uni2_log_prob_grad: unknown family %d", kind);
  ZS_REQUIRE(x && a && b && gout, "zshmc_uni2_log_prob_grad: null pointer");
  ZS_REQUIRE(rows >= 0 && cols >= 1, "zshmc_uni2_log_prob_grad: bad shape");
  ZS_REQUIRE(mode_ok2(a_bcast) && mode_ok2(b_bcast),
             "zshmc_uni2_log_prob_grad: bad broadcast mode");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define CALL(K) launch_bwd<K>(x, a, b, gout, gx, ga, gb, rows, cols, a_bcast, b_bcast, reduce_cols, s)
  ZS_KIND_SWITCH(kind, CALL)
#undef CALL
  ZS_LAUNCH_CHECK("uni2 grad launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_uni2_sample(int kind, float* out, const float* a,
                                 const float* b, int64_t n, int64_t inner,
                                 int a_bcast, int b_bcast, uint64_t seed,
                                 uint32_t offset, void* stream) {
  if (n == 0) return ZSHMC_OK;
  ZS_REQUIRE(kind >= 0 && kind <= 3, "zshmc_uni2_sample: unknown family %d", kind);
  ZS_REQUIRE(out && a && b && n > 0 && inner >= 1, "zshmc_uni2_sample: bad arguments");
  ZS_REQUIRE((a_bcast == ZSHMC_BCAST_FULL || a_bcast == ZSHMC_BCAST_SCALAR) &&
                 (b_bcast == ZSHMC_BCAST_FULL || b_bcast == ZSHMC_BCAST_SCALAR),
             "zshmc_uni2_sample: parameters are FULL (over one sample) or SCALAR");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define CALL(K) launch_sample<K>(out, a, b, n, inner, a_bcast, b_bcast, seed, offset, s)
  ZS_KIND_SWITCH(kind, CALL)
#undef CALL
  ZS_LAUNCH_CHECK("uni2 sample launch");
  return ZSHMC_OK;
}
