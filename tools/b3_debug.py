#!/usr/bin/env python
"""First-contact diagnostics for csrc/b3_kernel.h: small structured
cases whose error pattern names the layout assumption that is wrong
(the MFMA operand slots, the transposing LDS read, the image placement)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhusuan_amd import _capi  # noqa: E402

dev = torch.device('cuda', 0)
s = torch.cuda.current_stream().cuda_stream


def image(X, width):
    nb = ctypes.c_int64()
    _capi.call('zshmc_bf16x3_image_bytes', X.shape[0], width, ctypes.addressof(nb))
    img = torch.empty(nb.value, dtype=torch.uint8, device=X.device)
    _capi.call('zshmc_bf16x3_split', X.data_ptr(), X.shape[0], width,
               X.stride(0), img.data_ptr(), s)
    return img


def run(W, X, y, want_ll=True):
    C, D = W.shape
    N = X.shape[0]
    Wt, Xt, yt = (torch.tensor(a, device=dev) for a in (W, X, y))
    ll = torch.full((C,), float('nan'), device=dev)
    g = torch.full((C, D), float('nan'), device=dev)
    _capi.call('zshmc_linear_bernoulli_log_lik_bf16x3', Wt.data_ptr(),
               image(Xt, D).data_ptr(), yt.data_ptr(), C, N, D,
               ll.data_ptr() if want_ll else None, g.data_ptr(), 1, None, s)
    torch.cuda.synchronize()
    return ll.cpu().numpy(), g.cpu().numpy()


def ref(W, X, y):
    l = W.astype(np.float64) @ X.astype(np.float64).T
    ll = (y * l - np.maximum(l, 0) - np.log1p(np.exp(-np.abs(l)))).sum(1)
    g = (y - 1 / (1 + np.exp(-l))) @ X.astype(np.float64)
    return ll, g


def show(tag, got, want, tol):
    err = np.abs(got - want)
    bad = np.argwhere(~(err <= tol))
    print('%s: max err %.3e, %d / %d beyond %.1e' % (
        tag, np.nanmax(err) if np.isfinite(err).any() else float('nan'),
        len(bad), got.size, tol))
    for idx in bad[:6]:
        print('    at %s got % .6f want % .6f' % (tuple(idx), got[tuple(idx)],
                                                   want[tuple(idx)]))
    return len(bad) == 0


rng = np.random.RandomState(0)
for (C, N, D) in ((128, 32, 64), (128, 96, 64), (128, 64, 256), (200, 1000, 128)):
    X = rng.normal(size=(N, D)).astype(np.float32)
    W = (rng.normal(size=(C, D)) * 0.3).astype(np.float32)
    y = (rng.uniform(size=N) < 0.5).astype(np.float32)
    for want_ll in (True, False):
        ll, g = run(W, X, y, want_ll)
        llr, gr = ref(W, X, y)
        print('== C=%d N=%d D=%d want_ll=%s' % (C, N, D, want_ll))
        if want_ll:
            ok1 = show('log-lik', ll, llr, 2e-5 * N + 2e-5 * np.abs(llr).max())
        ok2 = show('gradient', g, gr, 2e-5 * (np.abs(gr).max() + 1))
        if not ok2:
            e = np.abs(g - gr) > 2e-5 * (np.abs(gr).max() + 1)
            print('    bad chains (mod 32) histogram:',
                  np.bincount(np.argwhere(e)[:, 0] % 32, minlength=32).tolist())
            print('    bad features (mod 32) histogram:',
                  np.bincount(np.argwhere(e)[:, 1] % 32, minlength=32).tolist())
            # is it a permutation of the features / chains?
            for c in (0, 1, 33):
                if c < C:
                    d = np.abs(g[c][None, :] - gr[c][:, None])
                    print('    chain %d: got feature j matches want feature:' % c,
                          d.argmin(0)[:16].tolist())
# one-hot probes of GEMM 1's slot pairing: W = e_d picks feature d of X
C, N, D = 128, 32, 64
X = rng.normal(size=(N, D)).astype(np.float32)
W = np.zeros((C, D), np.float32)
for c in range(C):
    W[c, c % D] = 1.0
y = np.zeros(N, np.float32)
ll, g = run(W, X, y)
llr, gr = ref(W, X, y)
print('== one-hot W')
show('log-lik', ll, llr, 1e-3)
show('gradient', g, gr, 1e-3)
