#!/usr/bin/env python
"""Debug: per-wave start/end clocks of the fused kernel (library built with
-DZS_TIMING, which hijacks the orig_hamiltonian pointer as a timing buffer)."""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, '.')
from zhusuan_amd import _capi
lib = ctypes.CDLL(sys.argv[1])
fn = lib.zshmc_hmc_diag_normal_step
fn.restype, fn.argtypes = _capi.PROTOTYPES['zshmc_hmc_diag_normal_step']
C, D, L = 65536, 1024, 10
dev = torch.device('cuda', 0)
logstd = torch.linspace(-1, 1, D, device=dev)
mean = torch.zeros(D, device=dev)
q = torch.randn(C, D, device=dev) * torch.exp(logstd)
timing = torch.zeros(8192 * 4, dtype=torch.int64, device=dev)
acc = torch.zeros(C, device=dev)
s = torch.cuda.current_stream().cuda_stream
for it in range(4):
    timing.zero_()
    fn(q.data_ptr(), mean.data_ptr(), logstd.data_ptr(), None, 0.14, C, D, 0, L, 1, it, 1,
       acc.data_ptr(), timing.data_ptr(), None, None, None, None, None, s)
torch.cuda.synchronize()
t = timing.cpu().numpy().reshape(-1, 4)
t = t[t[:, 1] > 0]
start, end, xcc, n = t[:, 0], t[:, 1], t[:, 2] & 0xf, t[:, 3]
cyc = t[:, 2] >> 8
if cyc.max() > 0:
    print('shader clock during kernel: %.0f MHz (s_memtime ticks / 100 MHz ticks)' % (100.0 * (cyc / np.maximum(end - start, 1)).mean()))
t0 = start.min()
print('waves', len(t), 'chains/wave min/max', n.min(), n.max())
dur = (end - start).astype(np.float64)
print('clock ticks: kernel span %d; wave duration mean %.0f min %.0f max %.0f' % (
    end.max() - t0, dur.mean(), dur.min(), dur.max()))
print('start offset pct: p50 %.1f%% p90 %.1f%% max %.1f%%' % tuple(
    100.0 * np.percentile(start - t0, [50, 90, 100]) / (end.max() - t0)))
print('end   offset pct: p10 %.1f%% p50 %.1f%% p90 %.1f%% max 100' % tuple(
    100.0 * np.percentile(end - t0, [10, 50, 90]) / (end.max() - t0)))
for x in range(8):
    m = xcc == x
    if m.any():
        print('xcc %d: waves %d  mean dur %.0f  mean end %.1f%%  per-chain %.1f' % (
            x, m.sum(), dur[m].mean(), 100.0 * (end[m] - t0).mean() / (end.max() - t0),
            (dur[m] / n[m]).mean()))
