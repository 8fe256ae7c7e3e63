#!/usr/bin/env python
"""Micro-benchmark of the fused linear-Bernoulli kernel at (a slice of)
BASELINE config 3: C chains x N rows x D=256.  Reports TFLOP/s against the
fp32-MFMA peak (157.3)."""
import sys
import torch
sys.path.insert(0, '.')
from zhusuan_amd import _capi  # noqa

C = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
D = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device('cuda', 0)
W = torch.randn(C, D, device=dev) * 0.1
X = torch.randn(N, D, device=dev)
y = (torch.rand(N, device=dev) < 0.5).float()
ll = torch.empty(C, device=dev)
g = torch.empty(C, D, device=dev)
s = torch.cuda.current_stream().cuda_stream
SPLITS = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ws = torch.empty(SPLITS * C * (D + 1), device=dev) if SPLITS > 1 else None


import ctypes, os  # noqa
_fn = None
if os.environ.get('LB_LIB'):          # A/B: a variant library instead of the product's
    _fn = ctypes.CDLL(os.environ['LB_LIB']).zshmc_linear_bernoulli_log_lik
    _fn.restype, _fn.argtypes = _capi.PROTOTYPES['zshmc_linear_bernoulli_log_lik']


def run():
    if _fn is not None:
        rc = _fn(W.data_ptr(), X.data_ptr(), y.data_ptr(), C, N, D, ll.data_ptr(),
                 g.data_ptr(), SPLITS, _capi.ptr(ws), s)
        assert rc == 0
        return
    _capi.call('zshmc_linear_bernoulli_log_lik', W.data_ptr(), X.data_ptr(),
               y.data_ptr(), C, N, D, ll.data_ptr(), g.data_ptr(), SPLITS,
               _capi.ptr(ws), s)


run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 3
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
flop = 4.0 * N * D * C
print('C=%d N=%d D=%d: %.2f ms  %.1f TFLOP/s  (%.1f%% of 157.3 fp32 MFMA peak)' % (
    C, N, D, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100))
