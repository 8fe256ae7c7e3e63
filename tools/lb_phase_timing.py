#!/usr/bin/env python
"""Debug: per-phase shader-clock breakdown of the fused MFMA likelihood
kernel (library built with -DZS_LB_TIMING; wave 0 of block 0 overwrites the
first 7 gradient words with its accumulated clocks).
Usage: python tools/lb_phase_timing.py lib.so [D] [C] [N]"""
import ctypes
import sys
import torch
sys.path.insert(0, '.')
from zhusuan_amd import _capi  # noqa
lib = ctypes.CDLL(sys.argv[1])
fn = lib.zshmc_linear_bernoulli_log_lik
fn.restype, fn.argtypes = _capi.PROTOTYPES['zshmc_linear_bernoulli_log_lik']
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
C = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
N = int(sys.argv[4]) if len(sys.argv) > 4 else 50000
dev = torch.device('cuda', 0)
W = torch.randn(C, D, device=dev) * 0.1
X = torch.randn(N, D, device=dev)
y = (torch.rand(N, device=dev) < 0.5).float()
ll = torch.empty(C, device=dev)
g = torch.empty(C, D, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    rc = fn(W.data_ptr(), X.data_ptr(), y.data_ptr(), C, N, D, ll.data_ptr(),
            g.data_ptr(), 1, None, s)
    assert rc == 0
torch.cuda.synchronize()
tall = g[0, :32].cpu().numpy().reshape(4, 8)
print('per-wave clocks/tile by phase (waves (a,b) = 00 01 10 11):')
for w in range(4):
    print('  wave %d: ' % w + ' '.join('%7.0f' % (v / tall[w, 6]) for v in tall[w, :6]))
t = tall[0, :7]
names = ['head+phase1', 'residual g0', 'phase 3a', 'barrier 1', 'phase 3b',
         'dma wait+barrier 2']
tiles = t[6]
tot = t[:6].sum()
mfma = D * 64.0        # D MFMAs of 64 cycles per wave and tile
print('D=%d: %d tiles, %.0f clocks/tile (MFMA-only floor %.0f = %.1f%%)' % (
    D, tiles, tot / tiles, mfma, 100 * mfma / (tot / tiles)))
for n, v in zip(names, t[:6]):
    print('  %-20s %8.0f clocks/tile  %5.1f%%' % (n, v / tiles, 100 * v / tot))
