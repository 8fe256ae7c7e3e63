// Stand-alone log_prob / analytic-gradient / sampling kernels for the
// distributions on the HMC hot path (gfx950).  Reference formulas:
//   Normal       zhusuan/distributions/univariate.py:161-181
//   Bernoulli    zhusuan/distributions/univariate.py:386-403
//   Categorical  zhusuan/distributions/univariate.py:478-548
//   UnnormalizedMultinomial  zhusuan/distributions/multivariate.py:435-443
//   group_ndims reduction    zhusuan/distributions/base.py:302-304
// The TensorFlow closed forms they delegate to are restated:
//   sigmoid_cross_entropy_with_logits(z, l) = max(l,0) - l z + log1p(exp(-|l|))
//   sparse_softmax_cross_entropy_with_logits(k, l) = logsumexp(l) - l[k]
//
// All kernels are HBM-bound element-wise / row-reduction passes: flat
// grid-stride loops with coalesced accesses; row sums use one wave per row
// and shuffle reductions.
#include "common.h"
#include "philox.h"

namespace zshmc {

constexpr float kNegHalfLog2Pi = -0.91893853320467274178f;

__device__ __forceinline__ float fetch(const float* __restrict__ p, int mode,
                                       int64_t i, int64_t col) {
  return mode == ZSHMC_BCAST_FULL ? p[i] : (mode == ZSHMC_BCAST_ROW ? p[col] : p[0]);
}

__device__ __forceinline__ float normal_lp(float x, float mean, float logstd) {
  const float prec = expf(-2.0f * logstd);
  const float d = x - mean;
  return kNegHalfLog2Pi - logstd - 0.5f * prec * d * d;
}

__device__ __forceinline__ float bernoulli_lp(float l, float z) {
  return -(fmaxf(l, 0.f) - l * z + log1pf(expf(-fabsf(l))));
}

__device__ __forceinline__ float sigmoidf(float l) {
  return 1.0f / (1.0f + expf(-l));
}

typedef float d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d4 fetch4(const float* __restrict__ p, int mode,
                                     int64_t i, int64_t col) {
  if (mode == ZSHMC_BCAST_FULL) return *reinterpret_cast<const d4*>(p + i);
  if (mode == ZSHMC_BCAST_ROW) return *reinterpret_cast<const d4*>(p + col);
  const float v = p[0];
  return d4{v, v, v, v};
}

// 16-B vectorised variants (cols % 4 == 0, every array 16-B aligned): four
// consecutive columns per lane
template <int KIND>
__global__ __launch_bounds__(256) void lp_rowsum_vec_kernel(
    const float* __restrict__ a, const float* __restrict__ b,
    const float* __restrict__ c, float* __restrict__ out, int64_t rows,
    int64_t cols, int mode_b, int mode_c) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / 64);
  for (int64_t r = wave; r < rows; r += n_waves) {
    float s = 0.f;
    for (int64_t col = (int64_t)lane * 4; col < cols; col += 256) {
      const int64_t i = r * cols + col;
      if (KIND == 0) {
        const d4 x = *reinterpret_cast<const d4*>(a + i);
        const d4 m = fetch4(b, mode_b, i, col), ls = fetch4(c, mode_c, i, col);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += normal_lp(x[j], m[j], ls[j]);
      } else {
        const d4 l = fetch4(a, mode_b, i, col), z = fetch4(b, mode_c, i, col);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += bernoulli_lp(l[j], z[j]);
      }
    }
    s = group_sum<64>(s);
    if (lane == 0) out[r] = s;
  }
}

__global__ __launch_bounds__(256) void normal_grad_vec_kernel(
    const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ logstd, const float* __restrict__ gout,
    float* __restrict__ gx, float* __restrict__ gmean,
    float* __restrict__ glogstd, int64_t n, int64_t cols, int mode_m,
    int mode_s, int reduce_cols) {
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n;
       i += (int64_t)gridDim.x * blockDim.x * 4) {
    const int64_t col = i % cols;  // cols % 4 == 0: the group stays in one row
    d4 g;
    if (reduce_cols) {
      const float gr = gout[i / cols];
      g = d4{gr, gr, gr, gr};
    } else {
      g = *reinterpret_cast<const d4*>(gout + i);
    }
    const d4 ls = fetch4(logstd, mode_s, i, col);
    const d4 m = fetch4(mean, mode_m, i, col);
    const d4 xv = *reinterpret_cast<const d4*>(x + i);
    d4 ox, om, os;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float prec = expf(-2.0f * ls[j]);
      const float d = xv[j] - m[j];
      const float pd = prec * d;
      ox[j] = -g[j] * pd;
      om[j] = g[j] * pd;
      os[j] = g[j] * (pd * d - 1.0f);
    }
    if (gx) *reinterpret_cast<d4*>(gx + i) = ox;
    if (gmean) *reinterpret_cast<d4*>(gmean + i) = om;
    if (glogstd) *reinterpret_cast<d4*>(glogstd + i) = os;
  }
}

// ---- element-wise forward (reduce_cols == 0) -----------------------------
template <int KIND>  // 0 normal, 1 bernoulli
__global__ __launch_bounds__(256) void lp_elementwise_kernel(
    const float* __restrict__ a, const float* __restrict__ b,
    const float* __restrict__ c, float* __restrict__ out, int64_t n,
    int64_t cols, int mode_b, int mode_c) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t col = i % cols;
    if (KIND == 0) {
      out[i] = normal_lp(a[i], fetch(b, mode_b, i, col), fetch(c, mode_c, i, col));
    } else {
      out[i] = bernoulli_lp(fetch(a, mode_b, i, col), fetch(b, mode_c, i, col));
    }
  }
}

// ---- row-reduced forward (reduce_cols != 0): one wave per row -------------
template <int KIND>
__global__ __launch_bounds__(256) void lp_rowsum_kernel(
    const float* __restrict__ a, const float* __restrict__ b,
    const float* __restrict__ c, float* __restrict__ out, int64_t rows,
    int64_t cols, int mode_b, int mode_c) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / 64);
  for (int64_t r = wave; r < rows; r += n_waves) {
    float s = 0.f;
    for (int64_t col = lane; col < cols; col += 64) {
      const int64_t i = r * cols + col;
      if (KIND == 0)
        s += normal_lp(a[i], fetch(b, mode_b, i, col), fetch(c, mode_c, i, col));
      else
        s += bernoulli_lp(fetch(a, mode_b, i, col), fetch(b, mode_c, i, col));
    }
    s = group_sum<64>(s);
    if (lane == 0) out[r] = s;
  }
}

__global__ __launch_bounds__(256) void normal_grad_kernel(
    const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ logstd, const float* __restrict__ gout,
    float* __restrict__ gx, float* __restrict__ gmean,
    float* __restrict__ glogstd, int64_t n, int64_t cols, int mode_m,
    int mode_s, int reduce_cols) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t col = i % cols;
    const float g = reduce_cols ? gout[i / cols] : gout[i];
    const float ls = fetch(logstd, mode_s, i, col);
    const float prec = expf(-2.0f * ls);
    const float d = x[i] - fetch(mean, mode_m, i, col);
    const float pd = prec * d;
    if (gx) gx[i] = -g * pd;
    if (gmean) gmean[i] = g * pd;
    if (glogstd) glogstd[i] = g * (pd * d - 1.0f);
  }
}

__global__ __launch_bounds__(256) void bernoulli_grad_kernel(
    const float* __restrict__ logits, const float* __restrict__ given,
    const float* __restrict__ gout, float* __restrict__ glogits, int64_t n,
    int64_t cols, int mode_l, int mode_z, int reduce_cols) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t col = i % cols;
    const float g = reduce_cols ? gout[i / cols] : gout[i];
    glogits[i] = g * (fetch(given, mode_z, i, col) -
                      sigmoidf(fetch(logits, mode_l, i, col)));
  }
}

// ---- Categorical / UnnormalizedMultinomial: one wave per row ----------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

__device__ __forceinline__ float row_logsumexp(const float* __restrict__ row,
                                               int64_t n_cat, int lane) {
  float m = -INFINITY;
  for (int64_t j = lane; j < n_cat; j += 64) m = fmaxf(m, row[j]);
  m = wave_max(m);
  float s = 0.f;
  for (int64_t j = lane; j < n_cat; j += 64) s += expf(row[j] - m);
  s = group_sum<64>(s);
  return m + logf(s);
}

// MODE 0: categorical forward; 1: categorical grad; 2: multinomial forward;
// 3: multinomial grad
template <int MODE>
__global__ __launch_bounds__(256) void softmax_family_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ labels,
    const float* __restrict__ given, const float* __restrict__ gout,
    float* __restrict__ out, int64_t rows, int64_t n_cat, int normalize) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / 64);
  for (int64_t r = wave; r < rows; r += n_waves) {
    const float* __restrict__ row = logits + r * n_cat;
    const float lse = (MODE < 2 || normalize) ? row_logsumexp(row, n_cat, lane) : 0.f;
    if (MODE == 0) {
      if (lane == 0) {
        const int64_t k = labels[r];
        out[r] = (k >= 0 && k < n_cat) ? row[k] - lse : NAN;
      }
    } else if (MODE == 1) {
      const int64_t k = labels[r];
      const float g = gout[r];
      for (int64_t j = lane; j < n_cat; j += 64)
        out[r * n_cat + j] = g * ((j == k ? 1.0f : 0.0f) - expf(row[j] - lse));
    } else if (MODE == 2) {
      float s = 0.f;
      for (int64_t j = lane; j < n_cat; j += 64)
        s += given[r * n_cat + j] * (row[j] - lse);
      s = group_sum<64>(s);
      if (lane == 0) out[r] = s;
    } else {
      const float g = gout[r];
      float tot = 0.f;
      if (normalize) {
        for (int64_t j = lane; j < n_cat; j += 64) tot += given[r * n_cat + j];
        tot = group_sum<64>(tot);
      }
      for (int64_t j = lane; j < n_cat; j += 64) {
        float v = given[r * n_cat + j];
        if (normalize) v -= tot * expf(row[j] - lse);
        out[r * n_cat + j] = g * v;
      }
    }
  }
}

// Register-resident variant for rows of up to NPL*256 categories that are
// 16-B aligned multiples of 4: the logits row is read ONCE into NPL float4 per
// lane (chunk layout (k*64 + lane)*4), max / logsumexp / outputs come from
// registers.  MODE as softmax_family_kernel.  A wave works on U adjacent rows
// at once: U = 1.  U = 2 (both rows' loads in flight before the first of the
// two wave reductions per row, interleaved shuffle chains) was measured for
// the slowest of these passes, the categorical gradient at [65 536, 1 024]:
// 4.34-4.40 TB/s against 4.68 with one row (profiles/archive/r03y_softmax_rows.txt)
// -- the pass is not short of bytes in flight; -DZS_SMX_ROWS=2 keeps the A/B.
template <int MODE, int NPL, int U>
__global__ __launch_bounds__(256) void softmax_family_reg_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ labels,
    const float* __restrict__ given, const float* __restrict__ gout,
    float* __restrict__ out, int64_t rows, int64_t n_cat, int normalize) {
  const int lane = threadIdx.x & 63;
  // (wave-uniform by construction; said so to the compiler: row pointers,
  // labels[r] and gout[r] become scalar loads issued at the top of a row)
  const int64_t wave =
      (int64_t)blockIdx.x * (blockDim.x / 64) +
      __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64));
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / 64);
  constexpr bool kKeepExp = MODE == 1 || MODE == 3;
  for (int64_t r0 = wave * U; r0 < rows; r0 += n_waves * U) {
    // rows r0 .. r0 + U - 1 of this trip (adjacent: U * n_cat contiguous
    // floats per wave; wave-uniform, `on` masks the tail)
    int64_t rr[U];
    bool on[U];
    d4 x[U][NPL];
    float mx[U];
    int64_t lab[U];
    float gr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rr[u] = r0 + u;
      on[u] = rr[u] < rows;
      if (!on[u]) rr[u] = r0;  // (re-reads row r0; nothing of it is stored)
      lab[u] = MODE < 2 ? labels[rr[u]] : 0;
      gr[u] = (MODE == 1 || MODE == 3) ? gout[rr[u]] : 0.f;
      const float* __restrict__ row = logits + rr[u] * n_cat;
      mx[u] = -INFINITY;
#pragma unroll
      for (int k = 0; k < NPL; ++k) {
        const int j0 = (k * 64 + lane) * 4;
        if (j0 < n_cat) {
          x[u][k] = *reinterpret_cast<const d4*>(row + j0);
          mx[u] = fmaxf(mx[u], fmaxf(fmaxf(x[u][k][0], x[u][k][1]),
                                     fmaxf(x[u][k][2], x[u][k][3])));
        } else {
          x[u][k] = d4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
      }
    }
    d4 gv[(MODE >= 2) ? U : 1][(MODE >= 2) ? NPL : 1];
    if (MODE >= 2) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
          const int j0 = (k * 64 + lane) * 4;
          gv[u][k] = d4{0.f, 0.f, 0.f, 0.f};
          if (j0 < n_cat)
            gv[u][k] =
                *reinterpret_cast<const d4*>(given + rr[u] * n_cat + j0);
        }
    }
    float lse[U], inv_se[U];
#pragma unroll
    for (int u = 0; u < U; ++u) lse[u] = inv_se[u] = 0.f;
    if (MODE < 2 || normalize) {
#pragma unroll
      for (int u = 0; u < U; ++u) mx[u] = wave_max(mx[u]);
      float se[U], shift[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        shift[u] = isfinite(mx[u]) ? mx[u] : 0.f;
        se[u] = 0.f;
        // the gradients need softmax = exp(x - shift) / sum: keep the
        // exponentials in place of x (one expf per element instead of two)
#pragma unroll
        for (int k = 0; k < NPL; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float e = expf(x[u][k][j] - shift[u]);
            se[u] += e;
            if (kKeepExp) x[u][k][j] = e;
          }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) se[u] = group_sum<64>(se[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        lse[u] = shift[u] + logf(se[u]);
        inv_se[u] = 1.0f / se[u];
      }
    }
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (lane == 0 && on[u]) {
          const int64_t kk = lab[u];
          out[rr[u]] = (kk >= 0 && kk < n_cat)
                           ? logits[rr[u] * n_cat + kk] - lse[u]
                           : NAN;
        }
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!on[u]) continue;
        const int64_t k64 = lab[u];
        const int kk = (k64 >= 0 && k64 < n_cat) ? (int)k64 : -1;
        const float g = gr[u];
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
          const int j0 = (k * 64 + lane) * 4;
          if (j0 < n_cat) {
            d4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              o[j] = g * ((j0 + j == kk ? 1.0f : 0.0f) -
                          x[u][k][j] * inv_se[u]);
            *reinterpret_cast<d4*>(out + rr[u] * n_cat + j0) = o;
          }
        }
      }
    } else {
      float tot[U], sdot[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        tot[u] = sdot[u] = 0.f;
#pragma unroll
        for (int k = 0; k < NPL; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // (chunks past the row hold given = 0 and x = -inf: skipped)
            const int j0 = (k * 64 + lane) * 4;
            if (j0 < n_cat) {
              tot[u] += gv[u][k][j];
              if (MODE == 2) sdot[u] += gv[u][k][j] * (x[u][k][j] - lse[u]);
            }
          }
      }
      if (MODE == 2) {
#pragma unroll
        for (int u = 0; u < U; ++u) sdot[u] = group_sum<64>(sdot[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (lane == 0 && on[u]) out[rr[u]] = sdot[u];
      } else {
        if (normalize) {
#pragma unroll
          for (int u = 0; u < U; ++u) tot[u] = group_sum<64>(tot[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!on[u]) continue;
          const float g = gr[u];
#pragma unroll
          for (int k = 0; k < NPL; ++k) {
            const int j0 = (k * 64 + lane) * 4;
            if (j0 < n_cat) {
              d4 o;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float v = gv[u][k][j];
                if (normalize) v -= tot[u] * (x[u][k][j] * inv_se[u]);
                o[j] = g * v;
              }
              *reinterpret_cast<d4*>(out + rr[u] * n_cat + j0) = o;
            }
          }
        }
      }
    }
  }
}

static inline int wave_row_grid(int64_t rows);

#ifndef ZS_SMX_ROWS  // rows a wave of the register-resident kernel works on at once
#define ZS_SMX_ROWS 1
#endif

template <int MODE>
static bool launch_softmax_reg(const float* logits, const int64_t* labels,
                               const float* given, const float* gout, float* out,
                               int64_t rows, int64_t n_cat, int normalize,
                               hipStream_t s, int grid) {
  const bool ok = n_cat % 4 == 0 && n_cat <= 2048 &&
                  (reinterpret_cast<uintptr_t>(logits) & 15) == 0 &&
                  (!given || (reinterpret_cast<uintptr_t>(given) & 15) == 0) &&
                  (MODE == 0 || MODE == 2 ||
                   (reinterpret_cast<uintptr_t>(out) & 15) == 0);
  if (!ok) return false;
#define ZS_SMX(NPL, U)                                                         \
  hipLaunchKernelGGL((softmax_family_reg_kernel<MODE, NPL, U>), dim3(grid),    \
                     dim3(256), 0, s, logits, labels, given, gout, out, rows,  \
                     n_cat, normalize)
  if (n_cat <= 256) ZS_SMX(1, ZS_SMX_ROWS);
  else if (n_cat <= 512) ZS_SMX(2, ZS_SMX_ROWS);
  else if (n_cat <= 1024) ZS_SMX(4, ZS_SMX_ROWS);
  else ZS_SMX(8, 1);
#undef ZS_SMX
  return true;
}

// ---- sampling ---------------------------------------------------------------
__global__ __launch_bounds__(256) void normal_sample_kernel(
    float* __restrict__ out, const float* __restrict__ mean,
    const float* __restrict__ std, int64_t n, int64_t inner, int mode_m,
    int mode_s, uint32_t k0, uint32_t k1, uint32_t offset) {
  const int64_t n_groups = (n + 3) / 4;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups;
       g += (int64_t)gridDim.x * blockDim.x) {
    float z[4];
    normal4((uint32_t)g, (uint32_t)((uint64_t)g >> 32), offset, kStreamDist, k0,
            k1, z[0], z[1], z[2], z[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = g * 4 + j;
      if (i < n) {
        const int64_t b = i % inner;
        const float m = mode_m == ZSHMC_BCAST_SCALAR ? mean[0] : mean[b];
        const float s = mode_s == ZSHMC_BCAST_SCALAR ? std[0] : std[b];
        out[i] = z[j] * s + m;  // univariate.py:167
      }
    }
  }
}

__global__ __launch_bounds__(256) void bernoulli_sample_kernel(
    int32_t* __restrict__ out, const float* __restrict__ logits, int64_t n,
    int64_t inner, uint32_t k0, uint32_t k1, uint32_t offset) {
  const int64_t n_groups = (n + 3) / 4;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups;
       g += (int64_t)gridDim.x * blockDim.x) {
    const U4 r = philox4x32((uint32_t)g, (uint32_t)((uint64_t)g >> 32),
                               offset, kStreamDist, k0, k1);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = g * 4 + j;
      if (i < n) out[i] = u01(w[j]) < sigmoidf(logits[i % inner]) ? 1 : 0;
    }
  }
}

__global__ __launch_bounds__(256) void categorical_sample_kernel(
    int32_t* __restrict__ out, const float* __restrict__ logits,
    int64_t n_samples, int64_t rows, int64_t n_cat, uint32_t k0, uint32_t k1,
    uint32_t offset) {
  const int64_t n = n_samples * rows;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = i >> 2;
    const U4 r = philox4x32((uint32_t)g, (uint32_t)((uint64_t)g >> 32),
                               offset, kStreamDist, k0, k1);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    const float u = u01(w[i & 3]);
    const float* __restrict__ row = logits + (i % rows) * n_cat;
    float m = -INFINITY;
    for (int64_t j = 0; j < n_cat; ++j) m = fmaxf(m, row[j]);
    float s = 0.f;
    for (int64_t j = 0; j < n_cat; ++j) s += expf(row[j] - m);
    const float lse = m + logf(s);
    float cdf = 0.f;
    int32_t k = 0;
    for (int64_t j = 0; j < n_cat; ++j) {
      cdf += expf(row[j] - lse);
      if (cdf <= u) ++k;
    }
    out[i] = k < (int32_t)n_cat ? k : (int32_t)n_cat - 1;
  }
}

static inline int flat_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  const int64_t cap = (int64_t)device_cu_count() * 16;
  if (b > cap) b = cap;
  return (int)(b > 0 ? b : 1);
}
static inline int wave_row_grid(int64_t rows) {
  int64_t b = (rows + 3) / 4;
  const int64_t cap = (int64_t)device_cu_count() * 8;
  if (b > cap) b = cap;
  return (int)(b > 0 ? b : 1);
}
static inline bool al16(const void* p) {
  return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}
static inline bool mode_ok(int m) {
  return m == ZSHMC_BCAST_FULL || m == ZSHMC_BCAST_ROW || m == ZSHMC_BCAST_SCALAR;
}

}  // namespace zshmc

using namespace zshmc;
#define ZS_STREAM reinterpret_cast<hipStream_t>(stream)

extern "C" int zshmc_normal_log_prob(const float* x, const float* mean,
                                     const float* logstd, float* out,
                                     int64_t rows, int64_t cols,
                                     int mean_bcast, int logstd_bcast,
                                     int reduce_cols, void* stream) {
  if (rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(x && mean && logstd && out, "zshmc_normal_log_prob: null pointer");
  ZS_REQUIRE(rows >= 0 && cols >= 1, "zshmc_normal_log_prob: bad shape");
  ZS_REQUIRE(mode_ok(mean_bcast) && mode_ok(logstd_bcast),
             "zshmc_normal_log_prob: bad broadcast mode");
  const bool vec = cols % 4 == 0 && al16(x) && al16(mean) && al16(logstd);
  if (reduce_cols && vec)
    hipLaunchKernelGGL(lp_rowsum_vec_kernel<0>, dim3(wave_row_grid(rows)),
                       dim3(256), 0, ZS_STREAM, x, mean, logstd, out, rows, cols,
                       mean_bcast, logstd_bcast);
  else if (reduce_cols)
    hipLaunchKernelGGL(lp_rowsum_kernel<0>, dim3(wave_row_grid(rows)), dim3(256),
                       0, ZS_STREAM, x, mean, logstd, out, rows, cols,
                       mean_bcast, logstd_bcast);
  else
    hipLaunchKernelGGL(lp_elementwise_kernel<0>, dim3(flat_grid(rows * cols)),
                       dim3(256), 0, ZS_STREAM, x, mean, logstd, out,
                       rows * cols, cols, mean_bcast, logstd_bcast);
  ZS_LAUNCH_CHECK("normal_log_prob launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_normal_log_prob_grad(
    const float* x, const float* mean, const float* logstd, const float* gout,
    float* gx, float* gmean, float* glogstd, int64_t rows, int64_t cols,
    int mean_bcast, int logstd_bcast, int reduce_cols, void* stream) {
  if (rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(x && mean && logstd && gout, "zshmc_normal_log_prob_grad: null pointer");
  ZS_REQUIRE(rows >= 0 && cols >= 1, "zshmc_normal_log_prob_grad: bad shape");
  ZS_REQUIRE(mode_ok(mean_bcast) && mode_ok(logstd_bcast),
             "zshmc_normal_log_prob_grad: bad broadcast mode");
  const bool vec = cols % 4 == 0 && al16(x) && al16(mean) && al16(logstd) &&
                   al16(gout) && al16(gx) && al16(gmean) && al16(glogstd);
  if (vec)
    hipLaunchKernelGGL(normal_grad_vec_kernel, dim3(flat_grid(rows * cols / 4)),
                       dim3(256), 0, ZS_STREAM, x, mean, logstd, gout, gx, gmean,
                       glogstd, rows * cols, cols, mean_bcast, logstd_bcast,
                       reduce_cols);
  else
    hipLaunchKernelGGL(normal_grad_kernel, dim3(flat_grid(rows * cols)),
                       dim3(256), 0, ZS_STREAM, x, mean, logstd, gout, gx, gmean,
                       glogstd, rows * cols, cols, mean_bcast, logstd_bcast,
                       reduce_cols);
  ZS_LAUNCH_CHECK("normal_grad_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_bernoulli_log_prob(const float* logits, const float* given,
                                        float* out, int64_t rows, int64_t cols,
                                        int logits_bcast, int given_bcast,
                                        int reduce_cols, void* stream) {
  if (rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(logits && given && out, "zshmc_bernoulli_log_prob: null pointer");
  ZS_REQUIRE(rows >= 0 && cols >= 1, "zshmc_bernoulli_log_prob: bad shape");
  ZS_REQUIRE(mode_ok(logits_bcast) && mode_ok(given_bcast),
             "zshmc_bernoulli_log_prob: bad broadcast mode");
  if (reduce_cols && cols % 4 == 0 && al16(logits) && al16(given))
    hipLaunchKernelGGL(lp_rowsum_vec_kernel<1>, dim3(wave_row_grid(rows)),
                       dim3(256), 0, ZS_STREAM, logits, given, nullptr, out,
                       rows, cols, logits_bcast, given_bcast);
  else if (reduce_cols)
    hipLaunchKernelGGL(lp_rowsum_kernel<1>, dim3(wave_row_grid(rows)), dim3(256),
                       0, ZS_STREAM, logits, given, nullptr, out, rows, cols,
                       logits_bcast, given_bcast);
  else
    hipLaunchKernelGGL(lp_elementwise_kernel<1>, dim3(flat_grid(rows * cols)),
                       dim3(256), 0, ZS_STREAM, logits, given, nullptr, out,
                       rows * cols, cols, logits_bcast, given_bcast);
  ZS_LAUNCH_CHECK("bernoulli_log_prob launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_bernoulli_log_prob_grad(
    const float* logits, const float* given, const float* gout, float* glogits,
    int64_t rows, int64_t cols, int logits_bcast, int given_bcast,
    int reduce_cols, void* stream) {
  if (rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(logits && given && gout && glogits,
             "zshmc_bernoulli_log_prob_grad: null pointer");
  ZS_REQUIRE(rows >= 0 && cols >= 1, "zshmc_bernoulli_log_prob_grad: bad shape");
  ZS_REQUIRE(mode_ok(logits_bcast) && mode_ok(given_bcast),
             "zshmc_bernoulli_log_prob_grad: bad broadcast mode");
  hipLaunchKernelGGL(bernoulli_grad_kernel, dim3(flat_grid(rows * cols)),
                     dim3(256), 0, ZS_STREAM, logits, given, gout, glogits,
                     rows * cols, cols, logits_bcast, given_bcast, reduce_cols);
  ZS_LAUNCH_CHECK("bernoulli_grad_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_categorical_log_prob(const float* logits,
                                          const int64_t* labels, float* out,
                                          int64_t rows, int64_t n_cat,
                                          void* stream) {
  if (rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(logits && labels && out, "zshmc_categorical_log_prob: null pointer");
  ZS_REQUIRE(rows >= 0 && n_cat >= 1, "zshmc_categorical_log_prob: bad shape");
  if (!launch_softmax_reg<0>(logits, labels, nullptr, nullptr, out, rows, n_cat, 1, ZS_STREAM,
                              wave_row_grid(rows)))
    hipLaunchKernelGGL(softmax_family_kernel<0>, dim3(wave_row_grid(rows)),
                       dim3(256), 0, ZS_STREAM, logits, labels, nullptr, nullptr, out, rows, n_cat, 1);
  ZS_LAUNCH_CHECK("categorical_log_prob launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_categorical_log_prob_grad(const float* logits,
                                               const int64_t* labels,
                                               const float* gout,
                                               float* glogits, int64_t rows,
                                               int64_t n_cat, void* stream) {
  if (rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(logits && labels && gout && glogits,
             "zshmc_categorical_log_prob_grad: null pointer");
  ZS_REQUIRE(rows >= 0 && n_cat >= 1, "zshmc_categorical_log_prob_grad: bad shape");
  if (!launch_softmax_reg<1>(logits, labels, nullptr, gout, glogits, rows, n_cat, 1, ZS_STREAM,
                              wave_row_grid(rows)))
    hipLaunchKernelGGL(softmax_family_kernel<1>, dim3(wave_row_grid(rows)),
                       dim3(256), 0, ZS_STREAM, logits, labels, nullptr, gout, glogits, rows, n_cat, 1);
  ZS_LAUNCH_CHECK("categorical_log_prob_grad launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_unnormalized_multinomial_log_prob(
    const float* logits, const float* given, float* out, int64_t rows,
    int64_t n_cat, int normalize, void* stream) {
  if (rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(logits && given && out,
             "zshmc_unnormalized_multinomial_log_prob: null pointer");
  ZS_REQUIRE(rows >= 0 && n_cat >= 1,
             "zshmc_unnormalized_multinomial_log_prob: bad shape");
  if (!launch_softmax_reg<2>(logits, nullptr, given, nullptr, out, rows, n_cat, normalize, ZS_STREAM,
                              wave_row_grid(rows)))
    hipLaunchKernelGGL(softmax_family_kernel<2>, dim3(wave_row_grid(rows)),
                       dim3(256), 0, ZS_STREAM, logits, nullptr, given, nullptr, out, rows, n_cat, normalize);
  ZS_LAUNCH_CHECK("unnormalized_multinomial_log_prob launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_unnormalized_multinomial_log_prob_grad(
    const float* logits, const float* given, const float* gout, float* glogits,
    int64_t rows, int64_t n_cat, int normalize, void* stream) {
  if (rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(logits && given && gout && glogits,
             "zshmc_unnormalized_multinomial_log_prob_grad: null pointer");
  ZS_REQUIRE(rows >= 0 && n_cat >= 1,
             "zshmc_unnormalized_multinomial_log_prob_grad: bad shape");
  if (!launch_softmax_reg<3>(logits, nullptr, given, gout, glogits, rows, n_cat, normalize, ZS_STREAM,
                              wave_row_grid(rows)))
    hipLaunchKernelGGL(softmax_family_kernel<3>, dim3(wave_row_grid(rows)),
                       dim3(256), 0, ZS_STREAM, logits, nullptr, given, gout, glogits, rows, n_cat, normalize);
  ZS_LAUNCH_CHECK("unnormalized_multinomial_log_prob_grad launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_normal_sample(float* out, const float* mean,
                                   const float* std, int64_t n, int64_t inner,
                                   int mean_bcast, int std_bcast, uint64_t seed,
                                   uint32_t offset, void* stream) {
  if (n == 0) return ZSHMC_OK;
  ZS_REQUIRE(out && mean && std, "zshmc_normal_sample: null pointer");
  ZS_REQUIRE(n >= 0 && inner >= 1, "zshmc_normal_sample: bad shape");
  hipLaunchKernelGGL(normal_sample_kernel, dim3(flat_grid((n + 3) / 4)),
                     dim3(256), 0, ZS_STREAM, out, mean, std, n, inner,
                     mean_bcast, std_bcast, (uint32_t)(seed & 0xFFFFFFFFull),
                     (uint32_t)(seed >> 32), offset);
  ZS_LAUNCH_CHECK("normal_sample_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_bernoulli_sample(int32_t* out, const float* logits,
                                      int64_t n, int64_t inner, uint64_t seed,
                                      uint32_t offset, void* stream) {
  if (n == 0) return ZSHMC_OK;
  ZS_REQUIRE(out && logits, "zshmc_bernoulli_sample: null pointer");
  ZS_REQUIRE(n >= 0 && inner >= 1, "zshmc_bernoulli_sample: bad shape");
  hipLaunchKernelGGL(bernoulli_sample_kernel, dim3(flat_grid((n + 3) / 4)),
                     dim3(256), 0, ZS_STREAM, out, logits, n, inner,
                     (uint32_t)(seed & 0xFFFFFFFFull), (uint32_t)(seed >> 32),
                     offset);
  ZS_LAUNCH_CHECK("bernoulli_sample_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_categorical_sample(int32_t* out, const float* logits,
                                        int64_t n_samples, int64_t rows,
                                        int64_t n_cat, uint64_t seed,
                                        uint32_t offset, void* stream) {
  if (n_samples * rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(out && logits, "zshmc_categorical_sample: null pointer");
  ZS_REQUIRE(n_samples >= 0 && rows >= 0 && n_cat >= 1,
             "zshmc_categorical_sample: bad shape");
  hipLaunchKernelGGL(categorical_sample_kernel, dim3(flat_grid(n_samples * rows)),
                     dim3(256), 0, ZS_STREAM, out, logits, n_samples, rows,
                     n_cat, (uint32_t)(seed & 0xFFFFFFFFull),
                     (uint32_t)(seed >> 32), offset);
  ZS_LAUNCH_CHECK("categorical_sample_kernel launch");
  return ZSHMC_OK;
}
