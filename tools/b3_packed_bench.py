#!/usr/bin/env python
"""Round 6: the packed-rows form of the bf16x3 multinomial kernel
(csrc/b3_kernel.h, PK) beside the fp32 kernel on lntm_mcem.py's own layout at
scale -- n_chains x n_docs rows, every row its own counts row.  TFLOP/s =
4 R K V / HIP-event time (fp32-equivalent flops).
    python tools/b3_packed_bench.py [n_docs] [n_chains] [K]
Environment: LB_LIB = another build of libzshmc.so (A/B)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhusuan_amd import _capi, _ops  # noqa: E402

if os.environ.get('LB_LIB'):
    _capi.LIB_PATH = os.path.abspath(os.environ['LB_LIB'])
    print('# library: %s' % _capi.LIB_PATH, flush=True)
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
n_chains = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K = int(sys.argv[3]) if len(sys.argv) > 3 else 128
V = 12419
dev = torch.device('cuda', 0)
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
R = n_docs * n_chains
phi = torch.softmax(torch.randn(K, V, device=dev, generator=g), -1)
x = torch.poisson(torch.full((n_docs, V), 0.08, device=dev), generator=g)
theta = torch.softmax(torch.randn(R, K, device=dev, generator=g), -1)
phi_t = _ops._padded_phi_t(phi, K)
xp, stride = _ops._padded_counts(x, 32)
ll = torch.empty(R, device=dev)
gt = torch.empty(R, K, device=dev)
nb = ctypes.c_int64()
_capi.call('zshmc_bf16x3_image_bytes', phi_t.shape[0], K, ctypes.addressof(nb))
img = torch.empty(nb.value, dtype=torch.uint8, device=dev)
_capi.call('zshmc_bf16x3_split', phi_t.data_ptr(), phi_t.shape[0], K,
           phi_t.stride(0), img.data_ptr(), s)
flop = 4.0 * R * K * V


def timeit(fn, reps=10):
    for _ in range(3):
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


# the row-by-row vector-ALU kernel (csrc/sparse_multinomial.hip, exact fp32)
vals, rws, off, total = _ops.counts_csr(x)
for form, ll_ptr in (('grad only', None), ('ll + grad', ll.data_ptr())):
    ms = timeit(lambda: _capi.call(
        'zshmc_sparse_multinomial_log_lik', theta.data_ptr(), phi_t.data_ptr(),
        vals.data_ptr(), rws.data_ptr(), off.data_ptr(), n_docs, R, V, K,
        ll_ptr, gt.data_ptr(), 1, None, s))
    print('%-44s %-10s %d x %d rows, K=%d: %.3f ms (%d of %d words per row run)'
          % ('zshmc_sparse_multinomial_log_lik', form, n_chains, n_docs, K, ms,
             total // n_docs, V), flush=True)
for name, inner in (('zshmc_linear_multinomial_log_lik', phi_t),
                    ('zshmc_linear_multinomial_log_lik_bf16x3', img)):
    for form, ll_ptr in (('grad only', None), ('ll + grad', ll.data_ptr())):
        ms = timeit(lambda: _capi.call(
            name, theta.data_ptr(), inner.data_ptr(), xp.data_ptr(),
            xp.shape[0], stride, R, V, K, ll_ptr, gt.data_ptr(), 1, None, s))
        print('%-44s %-10s %d x %d rows, K=%d: %.3f ms = %.1f TFLOP/s = %.3f '
              'of the fp32 peak' % (name, form, n_chains, n_docs, K, ms,
                                    flop / ms / 1e9, flop / ms / 1e9 / 157.3),
              flush=True)
