#!/usr/bin/env python
"""Shader clocks per phase of the bf16x3 tile loop (a -DZS_B3_TIMING build of
csrc/b3_kernel.h: block 0 writes its waves' totals into grad_w).
    LB_LIB=build/variants/libzshmc_timing.so python tools/b3_phase_timing.py
MFMA issue per phase: 96 MFMAs x 32 clocks = 3072 at D = 256 (D * 12)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhusuan_amd import _capi  # noqa: E402

_capi.LIB_PATH = os.path.abspath(os.environ['LB_LIB'])
dev = torch.device('cuda', 0)
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(1)
for D in (128, 256):
    C, N = 32768, 32768
    X = torch.randn(N, D, device=dev, generator=g)
    y = (torch.rand(N, device=dev, generator=g) < 0.4).float()
    W = torch.randn(C, D, device=dev, generator=g) * (0.5 / D ** 0.5)
    gw = torch.zeros(C, D, device=dev)
    nb = ctypes.c_int64()
    _capi.call('zshmc_bf16x3_image_bytes', N, D, ctypes.addressof(nb))
    img = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    _capi.call('zshmc_bf16x3_split', X.data_ptr(), N, D, D, img.data_ptr(), s)
    for rep in range(2):
        _capi.call('zshmc_linear_bernoulli_log_lik_bf16x3', W.data_ptr(),
                   img.data_ptr(), y.data_ptr(), C, N, D, None, gw.data_ptr(),
                   1, None, s)
        torch.cuda.synchronize()
    t = gw[0, :16].cpu().tolist()
    for w in range(4):
        T = t[w * 4 + 2]
        print('D=%d wave %d: GEMM 1 phase %.0f clocks / tile, GEMM 2 phase %.0f '
              '(MFMA issue %d each), %d tiles' % (
                  D, w, t[w * 4] / T, t[w * 4 + 1] / T, D * 12, T))
