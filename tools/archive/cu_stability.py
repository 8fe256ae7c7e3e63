#!/usr/bin/env python
"""Debug: is the per-CU speed of the fused kernel stable from launch to launch?
(library built with -DZS_TIMING).  Records, for several launches, the time at
which each workgroup (= CU) finished, and prints the correlation of the
per-workgroup finish times between launches and their spread."""
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
acc = torch.zeros(C, device=dev)
s = torch.cuda.current_stream().cuda_stream
ends = []
for it in range(12):
    timing = torch.zeros(8192 * 4, dtype=torch.int64, device=dev)
    fn(q.data_ptr(), mean.data_ptr(), logstd.data_ptr(), None, 0.14, C, D, 0, L, 1, it, 1,
       acc.data_ptr(), timing.data_ptr(), None, None, None, None, None, s)
    torch.cuda.synchronize()
    t = timing.cpu().numpy().reshape(-1, 4)[:4096]
    t0 = t[:, 0][t[:, 1] > 0].min()
    e = (t[:, 1] - t0).reshape(256, 16).max(1).astype(np.float64)   # per workgroup
    xcc = (t[:, 2] & 0xf).reshape(256, 16)[:, 0]
    if it >= 4:
        ends.append(e)
E = np.stack(ends)
print('kernel span (100 MHz ticks) per launch:', E.max(1).astype(int))
print('per-WG finish / span: mean %.3f  p10 %.3f  min %.3f' % (
    (E / E.max(1, keepdims=True)).mean(), np.percentile(E / E.max(1, keepdims=True), 10),
    (E / E.max(1, keepdims=True)).min()))
c = np.corrcoef(E)
print('correlation of per-WG finish times between launches: mean off-diagonal %.3f' % (
    (c.sum() - len(c)) / (len(c) * (len(c) - 1))))
m = E.mean(0)
print('per-WG mean finish: std/mean %.4f; slowest 8 WGs %s (xcc %s)' % (
    m.std() / m.mean(), np.argsort(m)[-8:], xcc[np.argsort(m)[-8:]]))
print('if shares were set from the mean speeds: predicted span %.0f vs now %.0f' % (
    (1.0 / (1.0 / m).mean()), E.max(1).mean()))
for x in range(8):
    print('  xcc %d: mean finish %.0f' % (x, m[xcc == x].mean()))
