// The bf16x3 likelihood kernel (csrc/b3_kernel.h) for the Categorical family
// (univariate.py:496-548): rows of W are (chain, class) pairs, a chain's
// classes in 2^GL consecutive lanes of a wave's 32; GL is a template argument
// of the kernel -- the softmax butterflies of the element-wise stage are
// straight-line code in the MFMA issue gaps (csrc/lb_ops.h) -- hence one
// kernel per class stride, and a translation unit of their own.
#include "b3_kernel.h"

using namespace zshmc;

extern "C" int zshmc_linear_categorical_log_lik_bf16x3(
    const float* W, const void* X_image, const float* labels, int64_t n_cols,
    int64_t n_rows, int64_t n_features, int n_classes, int class_stride,
    float* log_lik, float* grad_w, int n_splits, float* workspace,
    void* stream) {
  if (n_cols == 0) return ZSHMC_OK;
  ZS_REQUIRE(W && X_image && labels && grad_w,
             "zshmc_linear_categorical_log_lik_bf16x3: null pointer");
  int cls_log2 = 0;
  while ((1 << cls_log2) < class_stride) ++cls_log2;
  ZS_REQUIRE(class_stride >= 1 && class_stride <= 32 &&
                 (1 << cls_log2) == class_stride && n_classes >= 1 &&
                 n_classes <= class_stride,
             "zshmc_linear_categorical_log_lik_bf16x3: class_stride must be a "
             "power of two <= 32 and 1 <= n_classes <= class_stride, got %d / "
             "%d", n_classes, class_stride);
  ZS_REQUIRE(n_cols > 0 && n_rows > 0 && n_cols % class_stride == 0 &&
                 b3_width(n_features),
             "zshmc_linear_categorical_log_lik_bf16x3: bad shape (n_features "
             "64 / 128 / 192 / 256)");
  ZS_REQUIRE((reinterpret_cast<uintptr_t>(W) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(X_image) & 15) == 0,
             "zshmc_linear_categorical_log_lik_bf16x3: W and the image must "
             "be 16-byte aligned");
  ZS_REQUIRE(n_splits >= 1 && n_splits <= 256 && (n_splits == 1 || workspace),
             "zshmc_linear_categorical_log_lik_bf16x3: 1 <= n_splits <= 256 "
             "and a workspace of n_splits*n_cols*(n_features+1) floats when > 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned char* img = reinterpret_cast<const unsigned char*>(X_image);
  const int C = n_classes;
#define ZS_CAT_W(D, G)                                                        \
  return launch_b3<D, 2, G>(W, img, labels, 1, n_rows, n_cols, n_rows,        \
                            log_lik, grad_w, s, n_splits, workspace, 0, C)
#define ZS_CAT_G(G)                                                           \
  case G:                                                                     \
    switch (n_features) {                                                     \
      case 64: ZS_CAT_W(64, G);                                               \
      case 128: ZS_CAT_W(128, G);                                             \
      case 192: ZS_CAT_W(192, G);                                             \
      default: ZS_CAT_W(256, G);                                              \
    }
  switch (cls_log2) {
    ZS_CAT_G(0)
    ZS_CAT_G(1)
    ZS_CAT_G(2)
    ZS_CAT_G(3)
    ZS_CAT_G(4)
    default:
      switch (n_features) {
        case 64: ZS_CAT_W(64, 5);
        case 128: ZS_CAT_W(128, 5);
        case 192: ZS_CAT_W(192, 5);
        default: ZS_CAT_W(256, 5);
      }
  }
#undef ZS_CAT_G
#undef ZS_CAT_W
}
