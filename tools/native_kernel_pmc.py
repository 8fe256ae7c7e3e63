#!/usr/bin/env python
"""Workload for tools/profile_native_full.sh: the two modes of the fused
fp32-MFMA likelihood kernel AT THE SHAPES THE BENCH LINE QUOTES -- BASELINE
configs[2] (32 768 chains x 10^6 rows x 256) and configs[4] (8 192 x 5 000
(chain, document) rows x K = 128 x V = 12 419) -- launched straight through
the C-ABI, two launches of each call form, so that rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
can put their HBM traffic next to the algorithmic bytes.
  python tools/native_kernel_pmc.py [n_chains_config5]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhusuan_amd import _capi, _ops  # noqa: E402

dev = torch.device('cuda', 0)
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)

if len(sys.argv) > 1 and sys.argv[1] == 'packed':
    # round 6: lntm_mcem.py's own layout at scale -- ONE chain x 32 768
    # documents (every row of theta its own counts row), K = 128, V = 12 419:
    # the fp32 kernel and the packed-rows form of the bf16x3 kernel
    #   bash tools/profile_native_full.sh r06p packed
    import ctypes
    n_docs, K, V = 32768, 128, 12419
    phi = torch.softmax(torch.randn(K, V, device=dev, generator=g), -1)
    x = torch.poisson(torch.full((n_docs, V), 0.08, device=dev), generator=g)
    theta = torch.softmax(torch.randn(n_docs, K, device=dev, generator=g), -1)
    phi_t = _ops._padded_phi_t(phi, K)
    xp, stride = _ops._padded_counts(x, 32)
    assert _capi.load().zshmc_bf16x3_multinomial_rows_packed(n_docs, 1) == 1
    ll = torch.empty(n_docs, device=dev)
    gt = torch.empty(n_docs, K, device=dev)
    nb = ctypes.c_int64()
    _capi.call('zshmc_bf16x3_image_bytes', phi_t.shape[0], K,
               ctypes.addressof(nb))
    img = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    _capi.call('zshmc_bf16x3_split', phi_t.data_ptr(), phi_t.shape[0], K,
               phi_t.stride(0), img.data_ptr(), s)
    for name, inner in (('zshmc_linear_multinomial_log_lik', phi_t),
                        ('zshmc_linear_multinomial_log_lik_bf16x3', img)):
        for ll_ptr in (ll.data_ptr(), ll.data_ptr(), None, None, None):
            _capi.call(name, theta.data_ptr(), inner.data_ptr(),
                       xp.data_ptr(), xp.shape[0], stride, n_docs, V, K,
                       ll_ptr, gt.data_ptr(), 1, None, s)
    torch.cuda.synchronize()
    print('config5 shape: rows=%d K=%d V=%d; PACKED ROWS (1 chain x %d '
          'documents): counts streamed per launch %.3e B' % (
              n_docs, K, V, n_docs, 4.0 * n_docs * stride))
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == 'own':
    # round 6: configs[4] over the documents' OWN vocabularies (one document
    # per workgroup, tiles gathered from the phi^T image): 2 048 chains x 5 000
    # documents, K = 128, V = 12 419, beside the dense bf16x3 form
    #   bash tools/profile_native_full.sh r06s own
    import ctypes
    n_chains, n_docs, K, V = 2048, 5000, 128, 12419
    rows = n_chains * n_docs
    phi = torch.softmax(torch.randn(K, V, device=dev, generator=g), -1)
    x = torch.poisson(torch.full((n_docs, V), 0.08, device=dev), generator=g)
    theta = torch.softmax(torch.randn(rows, K, device=dev, generator=g), -1)
    phi_t = _ops._padded_phi_t(phi, K)
    xp, stride = _ops._padded_counts(x, 32)
    vals, rws, off, total = _ops.counts_csr(x)
    ll = torch.empty(rows, device=dev)
    gt = torch.empty(rows, K, device=dev)
    nb = ctypes.c_int64()
    _capi.call('zshmc_bf16x3_image_bytes', phi_t.shape[0], K,
               ctypes.addressof(nb))
    img = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    _capi.call('zshmc_bf16x3_split', phi_t.data_ptr(), phi_t.shape[0], K,
               phi_t.stride(0), img.data_ptr(), s)
    for ll_ptr in (ll.data_ptr(), None, None, None):
        _capi.call('zshmc_linear_multinomial_log_lik_bf16x3_sparse',
                   theta.data_ptr(), img.data_ptr(), vals.data_ptr(),
                   rws.data_ptr(), off.data_ptr(), n_docs, rows, V, K, ll_ptr,
                   gt.data_ptr(), 1, None, s)
    for ll_ptr in (None, None):
        _capi.call('zshmc_linear_multinomial_log_lik_bf16x3', theta.data_ptr(),
                   img.data_ptr(), xp.data_ptr(), n_docs, stride, rows, V, K,
                   ll_ptr, gt.data_ptr(), 1, None, s)
    torch.cuda.synchronize()
    print('config5 shape: rows=%d K=%d V=%d; OWN VOCABULARY: %d of %d words per '
          'document run (padded); the dense form beside it' % (
              rows, K, V, total // n_docs, V))
    sys.exit(0)

# ---- configs[2]: Bernoulli mode --------------------------------------------
C, N, D = 32768, 1000000, 256
X = torch.randn(N, D, device=dev, generator=g)
y = (torch.rand(N, device=dev, generator=g) < 0.5).float()
W = torch.randn(C, D, device=dev, generator=g) * 0.05
ll = torch.empty(C, device=dev)
gw = torch.empty(C, D, device=dev)
# two launches of each form a transition issues: likelihood + gradient (its two
# ends) and gradient only (the L - 1 interior evaluations: log_lik = NULL)
for ll_ptr in (ll.data_ptr(), ll.data_ptr(), None, None):
    _capi.call('zshmc_linear_bernoulli_log_lik', W.data_ptr(), X.data_ptr(),
               y.data_ptr(), C, N, D, ll_ptr, gw.data_ptr(), 1, None, s)
# ... and the same launches on the bf16x3 kernels (csrc/b3_kernel.h)
import ctypes  # noqa: E402


def image(Xp, width):
    nb = ctypes.c_int64()
    _capi.call('zshmc_bf16x3_image_bytes', Xp.shape[0], width,
               ctypes.addressof(nb))
    img = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    _capi.call('zshmc_bf16x3_split', Xp.data_ptr(), Xp.shape[0], width,
               Xp.stride(0), img.data_ptr(), s)
    return img


img = image(X, D)
for ll_ptr in (ll.data_ptr(), ll.data_ptr(), None, None):
    _capi.call('zshmc_linear_bernoulli_log_lik_bf16x3', W.data_ptr(),
               img.data_ptr(), y.data_ptr(), C, N, D, ll_ptr, gw.data_ptr(), 1,
               None, s)
del img
torch.cuda.synchronize()
print('config3 shape: C=%d N=%d D=%d; algorithmic bytes per launch: X %.3e '
      '(streamed once per 64-chain block: x%d = %.3e) + W, grad 2 x %.3e' % (
          C, N, D, 4.0 * N * D, C // 64, 4.0 * N * D * (C // 64), 4.0 * C * D))
del X, y, W, ll, gw
torch.cuda.empty_cache()

# ---- configs[4]: multinomial mode ------------------------------------------
n_chains = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n_docs, K, V = 5000, 128, 12419
rows = n_chains * n_docs
phi = torch.softmax(torch.randn(K, V, device=dev, generator=g), -1)
x = torch.poisson(torch.full((n_docs, V), 0.08, device=dev), generator=g)
theta = torch.softmax(torch.randn(rows, K, device=dev, generator=g), -1)
phi_t = _ops._padded_phi_t(phi, K)
xp, stride = _ops._padded_counts(x)
ll = torch.empty(rows, device=dev)
gt = torch.empty(rows, K, device=dev)
for ll_ptr in (ll.data_ptr(), ll.data_ptr(), None, None):
    _capi.call('zshmc_linear_multinomial_log_lik', theta.data_ptr(),
               phi_t.data_ptr(), xp.data_ptr(), xp.shape[0], stride, rows, V,
               K, ll_ptr, gt.data_ptr(), 1, None, s)
img = image(phi_t, K)
for ll_ptr in (ll.data_ptr(), ll.data_ptr(), None, None):
    _capi.call('zshmc_linear_multinomial_log_lik_bf16x3', theta.data_ptr(),
               img.data_ptr(), xp.data_ptr(), xp.shape[0], stride, rows, V, K,
               ll_ptr, gt.data_ptr(), 1, None, s)
torch.cuda.synchronize()
print('config5 shape: rows=%d K=%d V=%d; algorithmic bytes per launch: theta + '
      'grad 2 x %.3e, phi^T %.3e (x%d row blocks = %.3e), counts gathered '
      '%.3e' % (rows, K, V, 4.0 * rows * K, 4.0 * V * K, rows // 64,
                4.0 * V * K * (rows // 64), 4.0 * rows * V))
