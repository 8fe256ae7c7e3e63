#!/usr/bin/env python
"""Debug: per-phase shader-clock breakdown of the two-GEMM likelihood kernel
(csrc/linear_bernoulli.hip) from a library built with -DZS_LB_TIMING
(tools/build_lb_variants.sh timing "-DZS_LB_TIMING"; add -DZS_LB_LDS_PAD=26000
for ONE workgroup per CU at D <= 128, i.e. clocks without a partner wave on the
SIMD): the waves of block 0 overwrite the first gradient words with their
accumulated clocks.
Usage: python tools/lb_phase_timing.py lib.so [D] [C] [N] [grad_only]"""
import ctypes
import sys
import torch
sys.path.insert(0, '.')
from zhusuan_amd import _capi  # noqa
lib = ctypes.CDLL(sys.argv[1])
fn = lib.zshmc_linear_bernoulli_log_lik
fn.restype, fn.argtypes = _capi.PROTOTYPES['zshmc_linear_bernoulli_log_lik']
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
C = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
N = int(sys.argv[4]) if len(sys.argv) > 4 else 50048
grad_only = len(sys.argv) > 5 and sys.argv[5] == '1'
dev = torch.device('cuda', 0)
W = torch.randn(C, D, device=dev) * 0.1
X = torch.randn(N, D, device=dev)
y = (torch.rand(N, device=dev) < 0.5).float()
ll = torch.empty(C, device=dev)
g = torch.empty(C, D, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    rc = fn(W.data_ptr(), X.data_ptr(), y.data_ptr(), C, N, D,
            None if grad_only else ll.data_ptr(), g.data_ptr(), 1, None, s)
    assert rc == 0
torch.cuda.synchronize()
tall = g[0, :32].cpu().numpy().reshape(4, 8)
names = ['head + phase 1 (issue)', 'drain + residual 0', 'phase 3', 'end of tile']
print('D=%d %s: per-wave clocks/tile by phase (waves (a,b) = 00 01 10 11)' % (
    D, 'gradient only' if grad_only else 'likelihood + gradient'))
for w in range(4):
    print('  wave %d: ' % w + ' '.join('%7.0f' % (v / tall[w, 6]) for v in tall[w, :4]))
t = tall[0]
tiles, tot = t[6], t[:4].sum()
mfma = D * 64.0        # D MFMAs of 64 cycles per wave and tile
floor = [D / 2 * 64.0, 0, D / 2 * 64.0, 0]
print('  %d tiles, %.0f clocks/tile (MFMA-only floor %.0f = %.1f%%)' % (
    tiles, tot / tiles, mfma, 100 * mfma / (tot / tiles)))
for n, v, f in zip(names, t[:4], floor):
    print('  %-24s %8.0f clocks/tile  %5.1f%%   (MFMA issue %5.0f)' % (
        n, v / tiles, 100 * v / tot, f))
