// The bf16x3 likelihood kernels (csrc/b3_kernel.h: the design, the tile image
// and the kernel template): the image split and the Bernoulli / mixture-
// multinomial entry points.  (The Categorical family is its own translation
// unit, csrc/linear_bf16x3_cat.hip: one kernel per class stride.)
#include "b3_kernel.h"

namespace zshmc {
// ---------------------------------------------------------------------------
// X [n_rows, ldx] float32 (the first `width` columns) -> tile image:
//   tile t (rows 32 t ..), plane p, block P (features 32 P ..): 2 KB at
//   ((t * 3 + p) * (width / 32) + P) * 2048, chunk b3_chunk(m, h, par) of it
//   = bf16 plane p of X[32 t + m][32 P + 16 par + 8 h .. + 7]; rows >= n_rows
//   are zero.
__global__ __launch_bounds__(256) void b3_split_kernel(
    const float* __restrict__ X, int64_t n_rows, int width, int64_t ldx,
    unsigned char* __restrict__ image) {
  const int c8n = width / 8;
  const int64_t n_tiles = (n_rows + kB3Rows - 1) / kB3Rows;
  const int64_t total = n_tiles * kB3Rows * c8n;
  const int np = width / 32;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % c8n);
    const int64_t n = i / c8n;
    f4 lo4 = f4{0.f, 0.f, 0.f, 0.f}, hi4 = lo4;
    if (n < n_rows) {
      const float* __restrict__ src = X + n * ldx + c8 * 8;
      lo4 = *reinterpret_cast<const f4*>(src);
      hi4 = *reinterpret_cast<const f4*>(src + 4);
    }
    u4 pl[3];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const Split3 a = split3(lo4[2 * q], lo4[2 * q + 1]);
      const Split3 b = split3(hi4[2 * q], hi4[2 * q + 1]);
      pl[0][q] = a.hi, pl[1][q] = a.mid, pl[2][q] = a.lo;
      pl[0][2 + q] = b.hi, pl[1][2 + q] = b.mid, pl[2][2 + q] = b.lo;
    }
    const int64_t t = n / kB3Rows;
    const int m = (int)(n % kB3Rows);
    const int P = c8 >> 2, par = (c8 >> 1) & 1, h = c8 & 1;
    const int chunk = b3_chunk(m, h, par);
#pragma unroll
    for (int p = 0; p < 3; ++p)
      *reinterpret_cast<u4*>(image + ((t * 3 + p) * np + P) * 2048 +
                             chunk * 16) = pl[p];
  }
}

}  // namespace zshmc

using namespace zshmc;

extern "C" int zshmc_bf16x3_image_bytes(int64_t n_rows, int64_t width,
                                        int64_t* bytes) {
  ZS_REQUIRE(n_rows > 0 && b3_width(width) && bytes,
             "zshmc_bf16x3_image_bytes: n_rows > 0, width 64 / 128 / 192 / 256, "
             "got %lld x %lld", (long long)n_rows, (long long)width);
  *bytes = (n_rows + kB3Rows - 1) / kB3Rows * 3 * (width / 32) * 2048;
  return ZSHMC_OK;
}

extern "C" int zshmc_bf16x3_split(const float* X, int64_t n_rows, int64_t width,
                                  int64_t ldx, void* image, void* stream) {
  ZS_REQUIRE(X && image, "zshmc_bf16x3_split: null pointer");
  ZS_REQUIRE(n_rows > 0 && b3_width(width) && ldx >= width && ldx % 4 == 0 &&
                 (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(image) & 15) == 0,
             "zshmc_bf16x3_split: width 64 / 128 / 192 / 256, rows of ldx >= "
             "width floats (a multiple of 4), 16-byte aligned pointers");
  const int64_t total = (n_rows + kB3Rows - 1) / kB3Rows * kB3Rows * (width / 8);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(b3_split_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), X, n_rows,
                     (int)width, ldx, reinterpret_cast<unsigned char*>(image));
  ZS_LAUNCH_CHECK("b3_split_kernel launch");
  return ZSHMC_OK;
}

extern "C" int zshmc_linear_bernoulli_log_lik_bf16x3(
    const float* W, const void* X_image, const float* y, int64_t n_chains,
    int64_t n_rows, int64_t n_features, float* log_lik, float* grad_w,
    int n_splits, float* workspace, void* stream) {
  if (n_chains == 0) return ZSHMC_OK;
  ZS_REQUIRE(W && X_image && y && grad_w,
             "zshmc_linear_bernoulli_log_lik_bf16x3: null pointer (grad_w is "
             "required: the log-likelihood alone is the fp32 entry point's)");
  ZS_REQUIRE(n_chains > 0 && n_rows > 0 && b3_width(n_features),
             "zshmc_linear_bernoulli_log_lik_bf16x3: n_features 64 / 128 / 192 "
             "/ 256, got %lld", (long long)n_features);
  ZS_REQUIRE((reinterpret_cast<uintptr_t>(W) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(X_image) & 15) == 0,
             "zshmc_linear_bernoulli_log_lik_bf16x3: W and the image must be "
             "16-byte aligned");
  ZS_REQUIRE(n_splits >= 1 && n_splits <= 256 && (n_splits == 1 || workspace),
             "zshmc_linear_bernoulli_log_lik_bf16x3: 1 <= n_splits <= 256 and a "
             "workspace of n_splits*n_chains*(n_features+1) floats when > 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned char* img = reinterpret_cast<const unsigned char*>(X_image);
  switch (n_features) {
    case 64:
      return launch_b3<64, 0>(W, img, y, 1, n_rows, n_chains, n_rows, log_lik,
                              grad_w, s, n_splits, workspace, 0);
    case 128:
      return launch_b3<128, 0>(W, img, y, 1, n_rows, n_chains, n_rows, log_lik,
                               grad_w, s, n_splits, workspace, 0);
    case 192:
      return launch_b3<192, 0>(W, img, y, 1, n_rows, n_chains, n_rows, log_lik,
                               grad_w, s, n_splits, workspace, 0);
    default:
      return launch_b3<256, 0>(W, img, y, 1, n_rows, n_chains, n_rows, log_lik,
                               grad_w, s, n_splits, workspace, 0);
  }
}

extern "C" int zshmc_bf16x3_multinomial_rows_packed(int64_t count_rows,
                                                    int64_t chains_per_doc) {
  return count_rows > 1 && chains_per_doc % kB3Chains != 0 &&
         chains_per_doc < 8 * kB3Chains;
}

extern "C" int zshmc_linear_multinomial_log_lik_bf16x3(
    const float* theta, const void* phi_image, const float* counts,
    int64_t count_rows, int64_t count_stride, int64_t n_rows, int64_t n_vocab,
    int64_t n_topics, float* log_lik, float* grad_theta, int n_splits,
    float* workspace, void* stream) {
  if (n_rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(theta && phi_image && counts && grad_theta,
             "zshmc_linear_multinomial_log_lik_bf16x3: null pointer");
  ZS_REQUIRE(n_rows > 0 && n_vocab > 0 && count_rows > 0 &&
                 n_rows % count_rows == 0 && count_stride >= n_vocab &&
                 b3_width(n_topics),
             "zshmc_linear_multinomial_log_lik_bf16x3: bad shape (n_topics 64 "
             "/ 128 / 192 / 256)");
  ZS_REQUIRE((reinterpret_cast<uintptr_t>(theta) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(phi_image) & 15) == 0,
             "zshmc_linear_multinomial_log_lik_bf16x3: theta and the image "
             "must be 16-byte aligned");
  ZS_REQUIRE(n_splits >= 1 && n_splits <= 256 && (n_splits == 1 || workspace),
             "zshmc_linear_multinomial_log_lik_bf16x3: 1 <= n_splits <= 256 and "
             "a workspace of n_splits*n_rows*(n_topics+1) floats when > 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned char* img = reinterpret_cast<const unsigned char*>(phi_image);
  // rows are chain * count_rows + doc.  One document per workgroup (128
  // chains of it) where the chain axis fills such workgroups; otherwise --
  // a few chains x many documents, lntm_mcem.py's own layout -- 128
  // CONSECUTIVE rows per workgroup, every row with its own counts row
  // (b3_kernel.h, PK)
  const int64_t per_doc = n_rows / count_rows;
  // (taken where its preconditions hold: <= 192 topics -- three tile buffers
  // + 48 KB of counts in the LDS --, 16-byte aligned counts rows padded with
  // zeros to a multiple of 32 floats -- no clamping at the last tile --, a
  // counts matrix below 4 GB -- 32-bit lane offsets; otherwise one document
  // per workgroup whatever the fill)
  const bool packed =
      zshmc_bf16x3_multinomial_rows_packed(count_rows, per_doc) &&
      n_topics <= 192 && count_stride % 4 == 0 &&
      count_stride >= (n_vocab + 31) / 32 * 32 &&
      count_rows * count_stride < (1ll << 30) &&
      (reinterpret_cast<uintptr_t>(counts) & 15) == 0;
  const int dm = count_rows > 1;
  if (packed) {
    switch (n_topics) {
      case 64:
        return launch_b3<64, 1, 0, true>(theta, img, counts, count_rows,
                                         count_stride, n_rows, n_vocab, log_lik,
                                         grad_theta, s, n_splits, workspace, 0);
      case 128:
        return launch_b3<128, 1, 0, true>(theta, img, counts, count_rows,
                                          count_stride, n_rows, n_vocab,
                                          log_lik, grad_theta, s, n_splits,
                                          workspace, 0);
      default:
        return launch_b3<192, 1, 0, true>(theta, img, counts, count_rows,
                                          count_stride, n_rows, n_vocab,
                                          log_lik, grad_theta, s, n_splits,
                                          workspace, 0);
    }
  }
  switch (n_topics) {
    case 64:
      return launch_b3<64, 1>(theta, img, counts, count_rows, count_stride,
                              n_rows, n_vocab, log_lik, grad_theta, s, n_splits,
                              workspace, dm);
    case 128:
      return launch_b3<128, 1>(theta, img, counts, count_rows, count_stride,
                               n_rows, n_vocab, log_lik, grad_theta, s,
                               n_splits, workspace, dm);
    case 192:
      return launch_b3<192, 1>(theta, img, counts, count_rows, count_stride,
                               n_rows, n_vocab, log_lik, grad_theta, s,
                               n_splits, workspace, dm);
    default:
      return launch_b3<256, 1>(theta, img, counts, count_rows, count_stride,
                               n_rows, n_vocab, log_lik, grad_theta, s,
                               n_splits, workspace, dm);
  }
}

extern "C" int zshmc_linear_multinomial_log_lik_bf16x3_sparse(
    const float* theta, const void* phi_image, const float* counts_csr,
    const int32_t* row_index, const int64_t* doc_offsets, int64_t count_rows,
    int64_t n_rows, int64_t n_vocab, int64_t n_topics, float* log_lik,
    float* grad_theta, int n_splits, float* workspace, void* stream) {
  if (n_rows == 0) return ZSHMC_OK;
  ZS_REQUIRE(theta && phi_image && counts_csr && row_index && doc_offsets &&
                 grad_theta,
             "zshmc_linear_multinomial_log_lik_bf16x3_sparse: null pointer");
  ZS_REQUIRE(n_rows > 0 && n_vocab > 0 && count_rows > 0 &&
                 n_rows % count_rows == 0 && b3_width(n_topics) &&
                 n_vocab * n_topics * 6 < (1ll << 31),
             "zshmc_linear_multinomial_log_lik_bf16x3_sparse: bad shape "
             "(n_topics 64 / 128 / 192 / 256; the image below 2 GB)");
  ZS_REQUIRE((reinterpret_cast<uintptr_t>(theta) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(phi_image) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(counts_csr) & 3) == 0,
             "zshmc_linear_multinomial_log_lik_bf16x3_sparse: theta and the "
             "image must be 16-byte aligned");
  ZS_REQUIRE(n_splits >= 1 && n_splits <= 256 && (n_splits == 1 || workspace),
             "zshmc_linear_multinomial_log_lik_bf16x3_sparse: 1 <= n_splits <= "
             "256 and a workspace of n_splits*n_rows*(n_topics+1) floats when "
             "> 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned char* img = reinterpret_cast<const unsigned char*>(phi_image);
#define ZS_SP(D)                                                              \
  launch_b3<D, 1, 0, false, true>(theta, img, counts_csr, count_rows, 0,      \
                                  n_rows, n_vocab, log_lik, grad_theta, s,    \
                                  n_splits, workspace, 1, 0, row_index,       \
                                  doc_offsets)
  switch (n_topics) {
    case 64: return ZS_SP(64);
    case 128: return ZS_SP(128);
    case 192: return ZS_SP(192);
    default: return ZS_SP(256);
  }
#undef ZS_SP
}
