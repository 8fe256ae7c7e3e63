#!/usr/bin/env python
"""Debug: per-phase shader-clock breakdown of the feature-split likelihood
kernel (csrc/linear_bernoulli_wide.hip built with -DZS_LBW_TIMING: every wave
of block 0 overwrites the first gradient words with its accumulated clocks).
Usage: python tools/archive/lbw_phase_timing.py lib.so [D] [C] [N]"""
import ctypes
import sys
import torch
sys.path.insert(0, '.')
from zhusuan_amd import _capi  # noqa
lib = ctypes.CDLL(sys.argv[1])
fn = lib.zshmc_linear_bernoulli_log_lik
fn.restype, fn.argtypes = _capi.PROTOTYPES['zshmc_linear_bernoulli_log_lik']
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
C = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
N = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
dev = torch.device('cuda', 0)
W = torch.randn(C, D, device=dev) * 0.02
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
names = ['head+phase1+partials', 'barrier 1', 'sum+residual', 'barrier 2',
         'phase 3 + DMA issue', 'DMA wait']
print('D=%d C=%d N=%d: clocks/tile by phase, waves f = 0..3' % (D, C, N))
for i, n in enumerate(names):
    print('  %-22s' % n + ' '.join('%8.0f' % (tall[w, i] / tall[w, 6]) for w in range(4)))
tot = tall[:, :6].sum(1) / tall[:, 6]
mfma = D / 4 * 64.0        # D/4 MFMAs of 64 cycles per wave and tile
print('  %-22s' % 'total' + ' '.join('%8.0f' % v for v in tot))
print('  MFMA-only floor %.0f clocks/tile = %.1f%% of wave 0' % (mfma, 100 * mfma / tot[0]))
