#!/usr/bin/env python
"""Round 5: the bf16x3 likelihood kernels (csrc/b3_kernel.h) beside the
exact-fp32 ones at the BASELINE shapes' widths -- gradient only (the interior
evaluations of a trajectory) and likelihood + gradient.  TFLOP/s = 4 N D C /
HIP-event time (fp32-equivalent flops), against the fp32-MFMA peak 157.3 and
against the bf16 dense peak / 6 (six bf16 MFMAs per product) = 416.7.
    python tools/b3_bench.py [flops_scale]
Environment: LB_LIB = another build of libzshmc.so; B3_WIDTHS."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhusuan_amd import _capi  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
if os.environ.get('LB_LIB'):
    _capi.LIB_PATH = os.path.abspath(os.environ['LB_LIB'])
    print('# library: %s' % _capi.LIB_PATH, flush=True)
WIDTHS = tuple(int(w) for w in os.environ.get('B3_WIDTHS', '64,128,192,256').split(','))
dev = torch.device('cuda', 0)
s = torch.cuda.current_stream().cuda_stream
PEAK32, PEAK16 = 157.3, 2500.0 / 6
SPLITS = int(os.environ.get('B3_SPLITS', '1'))   # row-range slices (bf16x3 launches)


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


def image(X, width):
    nb = ctypes.c_int64()
    _capi.call('zshmc_bf16x3_image_bytes', X.shape[0], width,
               ctypes.addressof(nb))
    img = torch.empty(nb.value, dtype=torch.uint8, device=X.device)
    _capi.call('zshmc_bf16x3_split', X.data_ptr(), X.shape[0], width,
               X.stride(0), img.data_ptr(), s)
    return img


def report(tag, D, flop, modes):
    out = []
    for name, fn in modes:
        ms = timeit(fn)
        tf = flop / ms / 1e9
        out.append('%s %8.3f ms %6.1f TF = %.3f of fp32 peak, %.3f of bf16/6' % (
            name, ms, tf, tf / PEAK32, tf / PEAK16))
    for o in out:
        print('%-28s D=%-4d %s' % (tag, D, o), flush=True)


g = torch.Generator(device=dev).manual_seed(1)
for D in WIDTHS:
    C = 32768
    N = int(scale * (32768 * 256 // D)) // 32 * 32
    X = torch.randn(N, D, device=dev, generator=g)
    y = (torch.rand(N, device=dev, generator=g) < 0.4).float()
    W = torch.randn(C, D, device=dev, generator=g) * (0.5 / D ** 0.5)
    ll = torch.empty(C, device=dev)
    gw = torch.empty(C, D, device=dev)
    img = image(X, D)
    ws3 = torch.empty(SPLITS * C * (D + 1), device=dev) if SPLITS > 1 else None
    t_split = timeit(lambda: image(X, D))

    def call32(ll_, g_):
        _capi.call('zshmc_linear_bernoulli_log_lik', W.data_ptr(),
                   X.data_ptr(), y.data_ptr(), C, N, D, _capi.ptr(ll_),
                   _capi.ptr(g_), 1, None, s)

    def call3(ll_, g_):
        _capi.call('zshmc_linear_bernoulli_log_lik_bf16x3', W.data_ptr(),
                   img.data_ptr(), y.data_ptr(), C, N, D, _capi.ptr(ll_),
                   _capi.ptr(g_), SPLITS, _capi.ptr(ws3), s)
    print('# D=%d: C=%d N=%d, image split %.3f ms' % (D, C, N, t_split))
    report('bernoulli', D, 4.0 * N * D * C, [
        ('fp32   grad   ', lambda: call32(None, gw)),
        ('bf16x3 grad   ', lambda: call3(None, gw)),
        ('fp32   ll+grad', lambda: call32(ll, gw)),
        ('bf16x3 ll+grad', lambda: call3(ll, gw))])
    g32 = torch.empty_like(gw)
    call32(ll, g32)
    call3(ll, gw)
    torch.cuda.synchronize()
    print('#   max |g3 - g32| / max |g32| = %.2e' % (
        (gw - g32).abs().max().item() / g32.abs().max().item()))
    del X, W, gw, g32, img

for K, n_chains, n_docs in ((128, 256, 512), (256, 128, 512)):
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
    img = image(phi_t, K)

    def m32(ll_, g_):
        _capi.call('zshmc_linear_multinomial_log_lik', theta.data_ptr(),
                   phi_t.data_ptr(), x.data_ptr(), n_docs, stride, R, V, K,
                   _capi.ptr(ll_), _capi.ptr(g_), 1, None, s)

    def m3(ll_, g_):
        _capi.call('zshmc_linear_multinomial_log_lik_bf16x3', theta.data_ptr(),
                   img.data_ptr(), x.data_ptr(), n_docs, stride, R, V, K,
                   _capi.ptr(ll_), _capi.ptr(g_), 1, None, s)
    report('multinomial %dx%d rows' % (n_chains, n_docs), K, 4.0 * R * K * V, [
        ('fp32   grad   ', lambda: m32(None, gt)),
        ('bf16x3 grad   ', lambda: m3(None, gt)),
        ('fp32   ll+grad', lambda: m32(ll, gt)),
        ('bf16x3 ll+grad', lambda: m3(ll, gt))])
    g32 = torch.empty_like(gt)
    m32(ll, g32)
    m3(ll, gt)
    torch.cuda.synchronize()
    print('#   max |g3 - g32| / max |g32| = %.2e' % (
        (gt - g32).abs().max().item() / g32.abs().max().item()))
    del theta, gt, g32
