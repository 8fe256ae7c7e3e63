#!/usr/bin/env python
"""Round 4: the fused two-GEMM likelihood kernels in their three call modes
-- likelihood + gradient, gradient only (the interior evaluations of a
trajectory: log_lik = NULL), likelihood only (grad = NULL) -- at every padded
width, Bernoulli (OP 0) / mixture-multinomial (OP 1) / Categorical (OP 2).
TFLOP/s = 4 N D C (2 N D C without gradient) / HIP-event time, against the
fp32-MFMA peak 157.3.
    python tools/lb_modes_bench.py [flops_scale]
Environment (this tool only): LB_LIB = another build of libzshmc.so to time
(tools/build_lb_variants.sh), LB_WIDTHS = comma-separated widths to run."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhusuan_amd import _capi, _ops  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
if os.environ.get('LB_LIB'):
    _capi.LIB_PATH = os.path.abspath(os.environ['LB_LIB'])
    print('# library: %s' % _capi.LIB_PATH, flush=True)
WIDTHS = tuple(int(w) for w in os.environ.get(
    'LB_WIDTHS', '64,128,192,256,320,384,448,512,576,640,768,832,896,1024').split(','))
dev = torch.device('cuda', 0)
s = torch.cuda.current_stream().cuda_stream
PEAK = 157.3


def timeit(fn, reps=3):
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


def report(tag, D, flop_full, modes):
    out = []
    for name, fn, frac in modes:
        ms = timeit(fn)
        tf = flop_full * frac / ms / 1e9
        out.append('%s %7.2f ms %6.1f TF = %.3f' % (name, ms, tf, tf / PEAK))
    print('%-26s D=%-5d %s' % (tag, D, ' | '.join(out)), flush=True)


g = torch.Generator(device=dev).manual_seed(1)
for D in WIDTHS:
    C = 32768 if D <= 896 else 8192
    N = int(scale * (32768 * 256 // D if D <= 896 else 65536 * 256 // D))
    X = torch.randn(N, D, device=dev, generator=g)
    y = (torch.rand(N, device=dev, generator=g) < 0.4).float()
    W = torch.randn(C, D, device=dev, generator=g) * (0.5 / D ** 0.5)
    ll = torch.empty(C, device=dev)
    gw = torch.empty(C, D, device=dev)

    def call(ll_, g_):
        _capi.call('zshmc_linear_bernoulli_log_lik', W.data_ptr(),
                   X.data_ptr(), y.data_ptr(), C, N, D, _capi.ptr(ll_),
                   _capi.ptr(g_), 1, None, s)
    report('bernoulli', D, 4.0 * N * D * C, [
        ('ll+grad', lambda: call(ll, gw), 1.0),
        ('grad', lambda: call(None, gw), 1.0),
        ('ll', lambda: call(ll, None), 0.5)])
    # Categorical: 16 classes per chain (no padding classes)
    K = 16
    lab = torch.randint(0, K, (N,), device=dev, generator=g).float()

    def callc(ll_, g_):
        _capi.call('zshmc_linear_categorical_log_lik', W.data_ptr(),
                   X.data_ptr(), lab.data_ptr(), C, N, D, K, K,
                   _capi.ptr(ll_), _capi.ptr(g_), 1, None, s)
    report('categorical (16 classes)', D, 4.0 * N * D * C, [
        ('ll+grad', lambda: callc(ll, gw), 1.0),
        ('grad', lambda: callc(None, gw), 1.0),
        ('ll', lambda: callc(ll, None), 0.5)])
    del X, W, gw

# mixture multinomial at the topic model's shapes: rows = chains x documents
for K, n_chains, n_docs in ((128, 256, 512), (192, 128, 512), (256, 128, 512),
                            (384, 128, 512), (512, 64, 512),
                            (1024, 32, 512)):
    if K not in WIDTHS:
        continue
    V = int(scale * 12419)
    R = n_chains * n_docs
    theta = torch.softmax(torch.randn(R, K, device=dev, generator=g), -1)
    phi_t = torch.softmax(torch.randn(K, V, device=dev, generator=g),
                          -1).t().contiguous()
    stride = (V + 3) // 4 * 4
    x = torch.zeros(n_docs, stride, device=dev)
    x[:, :V] = torch.poisson(torch.full((n_docs, V), 0.08, device=dev),
                             generator=g)
    ll = torch.empty(R, device=dev)
    gt = torch.empty(R, K, device=dev)

    def callm(ll_, g_):
        _capi.call('zshmc_linear_multinomial_log_lik', theta.data_ptr(),
                   phi_t.data_ptr(), x.data_ptr(), n_docs, stride, R, V, K,
                   _capi.ptr(ll_), _capi.ptr(g_), 1, None, s)
    report('multinomial %dx%d rows' % (n_chains, n_docs), K,
           4.0 * R * K * V, [
               ('ll+grad', lambda: callm(ll, gt), 1.0),
               ('grad', lambda: callm(None, gt), 1.0),
               ('ll', lambda: callm(ll, None), 0.5)])
    del theta, gt
