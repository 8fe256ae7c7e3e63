#!/usr/bin/env python
"""The likelihood kernels above 256 columns -- width 512 (round 3: the
feature-split kernel; since round 4 the 16-chain-block kernel of
csrc/linear_bernoulli_mid.hip) and 1024 (csrc/linear_bernoulli_wide.hip) --
timed against the 256-wide kernel at the same flop count, and checked against
a float64 reference on a sub-block.
  python tools/archive/lb_wide_bench.py [n_chains] [n_rows]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zhusuan_amd import _capi, _ops  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(5)
out = {}
for D in (256, 512, 1024):
    n = N * 256 // D                       # same flops per call for every D
    X = torch.randn(n, D, device=dev, generator=g)
    y = (torch.rand(n, device=dev, generator=g) < 0.4).float()
    W = torch.randn(C, D, device=dev, generator=g) * (0.5 / D ** 0.5)
    ll = torch.empty(C, device=dev)
    gw = torch.empty(C, D, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    splits = _ops._row_splits(C, n, dev, D)
    ws = torch.empty(splits * C * (D + 1), device=dev) if splits > 1 else None

    def run(grad=True):
        _capi.call('zshmc_linear_bernoulli_log_lik', W.data_ptr(), X.data_ptr(),
                   y.data_ptr(), C, n, D, ll.data_ptr(),
                   gw.data_ptr() if grad else None, splits, _capi.ptr(ws), s)

    run()
    torch.cuda.synchronize()
    z = W[:64].double() @ X.double().t()
    want = (y.double() * z - torch.nn.functional.softplus(z)).sum(-1)
    gwant = (y.double() - torch.sigmoid(z)) @ X.double()
    err_ll = float((ll[:64].double() - want).abs().max() / want.abs().max())
    err_g = float((gw[:64].double() - gwant).abs().max() / gwant.abs().max())
    res = {}
    for grad in (True, False):
        for _ in range(2):
            run(grad)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run(grad)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        flop = (4.0 if grad else 2.0) * C * n * D
        res['grad' if grad else 'll_only'] = {
            'ms': round(ms, 3), 'tflops': round(flop / ms * 1e-9, 1),
            'frac_of_157': round(flop / ms * 1e-9 / 157.3, 3)}
    out['D=%d' % D] = dict(res, n_rows=n, splits=splits, rel_err_ll=err_ll,
                           rel_err_grad=err_g)
    print('D=%d' % D, json.dumps(out['D=%d' % D]), flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump({'n_chains': C, 'results': out}, open('gpurun_out/lb_wide_bench.json', 'w'), indent=1)
