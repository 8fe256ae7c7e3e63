#!/usr/bin/env python
"""Round 6: the own-vocabulary form of the bf16x3 multinomial kernel (one
document per workgroup, tiles gathered from the phi^T image) beside the dense
form at the configs[4] family's shape.  LB_LIB = another build (A/B).
    python tools/b3_own_vocab_bench.py [n_chains] [n_docs] [K] [poisson rate]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhusuan_amd import _capi, _ops  # noqa: E402

if os.environ.get('LB_LIB'):
    _capi.LIB_PATH = os.path.abspath(os.environ['LB_LIB'])
    print('# library: %s' % _capi.LIB_PATH, flush=True)
n_chains = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n_docs = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 128
rate = float(sys.argv[4]) if len(sys.argv) > 4 else 0.08   # Poisson rate per word
V = 12419
dev = torch.device('cuda', 0)
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
R = n_chains * n_docs
phi = torch.softmax(torch.randn(K, V, device=dev, generator=g), -1)
x = torch.poisson(torch.full((n_docs, V), rate, device=dev), generator=g)
theta = torch.softmax(torch.randn(R, K, device=dev, generator=g), -1)
phi_t = _ops._padded_phi_t(phi, K)
vals, rws, off, total = _ops.counts_csr(x)
gt = torch.empty(R, K, device=dev)
nb = ctypes.c_int64()
_capi.call('zshmc_bf16x3_image_bytes', phi_t.shape[0], K, ctypes.addressof(nb))
img = torch.empty(nb.value, dtype=torch.uint8, device=dev)
_capi.call('zshmc_bf16x3_split', phi_t.data_ptr(), phi_t.shape[0], K,
           phi_t.stride(0), img.data_ptr(), s)


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(
        enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = timeit(lambda: _capi.call(
    'zshmc_linear_multinomial_log_lik_bf16x3_sparse', theta.data_ptr(),
    img.data_ptr(), vals.data_ptr(), rws.data_ptr(), off.data_ptr(), n_docs, R,
    V, K, None, gt.data_ptr(), 1, None, s))
run = total / n_docs
print('own vocabulary, grad only: %d x %d rows, K=%d: %.2f ms (%d of %d words '
      'run: %.1f TFLOP/s on those = %.3f of the fp32 peak)' % (
          n_chains, n_docs, K, ms, run, V, 4.0 * R * K * run / ms / 1e9,
          4.0 * R * K * run / ms / 1e9 / 157.3), flush=True)
