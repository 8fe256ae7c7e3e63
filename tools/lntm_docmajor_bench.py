#!/usr/bin/env python
"""The multinomial mode of the MFMA likelihood kernel at BASELINE configs[4]'s
full shape (n_chains x 5 000 documents, K = 128, V = 12 419), document-major
tiles against consecutive-row tiles: run once with ZSHMC_LB_DOC_MAJOR=1 and
once with =0 (the switch is read once per process).
  python tools/lntm_docmajor_bench.py [n_chains] [n_topics]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhusuan_amd import _capi, _ops  # noqa: E402

dev = torch.device('cuda', 0)
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
n_chains = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
# (third argument: the number of topics -- above 256 the feature-split kernel)
n_docs, K, V = 5000, int(sys.argv[2]) if len(sys.argv) > 2 else 128, 12419
rows = n_chains * n_docs
phi = torch.softmax(torch.randn(K, V, device=dev, generator=g), -1)
mix = torch.softmax(torch.randn(n_docs, K, device=dev, generator=g), -1)
words = torch.multinomial(mix @ phi, 1000, replacement=True, generator=g)
x = torch.zeros(n_docs, V, device=dev).scatter_add_(
    1, words, torch.ones(words.shape, device=dev))
theta = torch.softmax(torch.randn(rows, K, device=dev, generator=g), -1)
phi_t = _ops._padded_phi_t(phi, K)
xp, stride = _ops._padded_counts(x)
ll = torch.empty(rows, device=dev)
gt = torch.empty(rows, K, device=dev)


def launch():
    _capi.call('zshmc_linear_multinomial_log_lik', theta.data_ptr(),
               phi_t.data_ptr(), xp.data_ptr(), xp.shape[0], stride, rows, V,
               K, ll.data_ptr(), gt.data_ptr(), 1, None, s)


launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(2):
    launch()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 2
flop = 4.0 * rows * K * V
print('ZSHMC_LB_DOC_MAJOR=%s  K=%d n_chains=%d  %.1f ms per evaluation = %.1f TFLOP/s '
      '= %.3f of the fp32-MFMA peak; checksum ll %.6e grad %.6e' % (
          os.environ.get('ZSHMC_LB_DOC_MAJOR', '(unset: on)'), K, n_chains, ms,
          flop / ms / 1e9, flop / ms / 1e9 / 157.3, float(ll.double().sum()),
          float(gt.double().abs().sum())))
