#!/usr/bin/env python
"""What a SMALL launch of the two-GEMM likelihood kernel is made of: duration
per launch (back-to-back launches between two events, so a pipelined launch's
dispatch is included) for one or two chain blocks against the number of
64-row tiles, gradient only -- the fixed part (prologue, W load, epilogue
store) and the cost per tile on a workgroup's critical path.
    python tools/lb_small_latency.py [D]"""
import sys

import torch

sys.path.insert(0, '.')
from zhusuan_amd import _capi  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device('cuda', 0)
s = _capi.current_stream()


def time_launch(C, N, splits, n=300):
    W = torch.randn(C, D, device=dev) * 0.1
    X = torch.randn(N, D, device=dev)
    y = (torch.rand(N, device=dev) < 0.5).float()
    g = torch.empty(C, D, device=dev)
    ws = torch.empty(max(1, splits * C * (D + 1)), device=dev)

    def run():
        _capi.call('zshmc_linear_bernoulli_log_lik', W.data_ptr(), X.data_ptr(),
                   y.data_ptr(), C, N, D, None, g.data_ptr(), splits,
                   ws.data_ptr() if splits > 1 else None, s)
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('D = %d, gradient only, us per launch (300 back to back)' % D)
for C in (64, 128):
    for tiles in (1, 2, 4, 8, 16):
        print('  C=%4d  %2d tiles, 1 slice: %6.2f' % (C, tiles,
                                                     time_launch(C, 64 * tiles, 1)))
# the E-step's shape: 100 rows, 12 419 rows of X in 98 slices (+ the reduction)
print('  C= 100  12419 rows, 98 slices + reduce: %6.2f' % time_launch(100, 12419, 98))
print('  C= 100  12419 rows, 1 slice:            %6.2f' % time_launch(100, 12419, 1))
