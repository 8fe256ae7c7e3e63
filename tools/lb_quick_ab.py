#!/usr/bin/env python
"""Round 3: a 10-second timing of the 64-chain-block likelihood kernel at D = 256
(32 768 chains x 32 768 rows) for A/B builds copied over zhusuan_amd/lib/libzshmc.so
(tools/ab_lb_variants.sh): python tools/lb_quick_ab.py TAG"""
import sys, torch
sys.path.insert(0, '.')
from zhusuan_amd import _capi
C, n, D = 32768, 32768, 256
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(5)
X = torch.randn(n, D, device=dev, generator=g); y = (torch.rand(n, device=dev, generator=g) < 0.4).float()
W = torch.randn(C, D, device=dev, generator=g) * 0.03
ll = torch.empty(C, device=dev); gw = torch.empty(C, D, device=dev)
s = torch.cuda.current_stream().cuda_stream
run = lambda: _capi.call('zshmc_linear_bernoulli_log_lik', W.data_ptr(), X.data_ptr(), y.data_ptr(), C, n, D, ll.data_ptr(), gw.data_ptr(), 1, None, s)
run(); run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(6): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 6
print(sys.argv[1], 'D=256 %.3f ms %.1f TFLOP/s' % (ms, 4.0 * C * n * D / ms * 1e-9), float(ll.double().sum()))
