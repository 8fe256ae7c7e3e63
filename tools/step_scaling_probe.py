#!/usr/bin/env python
"""Round 6: achieved bandwidth of the native plans' softmax leapfrog step
(csrc/hmc_model.hip, zshmc_model_kick_drift) against the number of rows --
configs[4] runs it over 4.1e7 rows x 128 (21 GB per buffer), where the
own-vocabulary likelihood kernel has made it a third of a transition.
    python tools/step_scaling_probe.py [mass 0|1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhusuan_amd import _capi  # noqa: E402

dev = torch.device('cuda', 0)
s = torch.cuda.current_stream().cuda_stream
use_mass = bool(int(sys.argv[1])) if len(sys.argv) > 1 else False
Dm = width = 128


def time_ms(fn, reps=5):
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


for Cc in (1280000, 5120000, 10240000, 20480000, 40960000):
    qq = torch.randn(Cc, Dm, device=dev) * 0.1
    pp = torch.randn(Cc, Dm, device=dev)
    gg = torch.randn(Cc, width, device=dev)
    op = torch.softmax(qq, -1).contiguous()
    pm = torch.zeros(5000, Dm, device=dev)        # per-document prior rows
    pl = torch.zeros(1, Dm, device=dev)
    mass = torch.ones(Dm, device=dev) if use_mass else None
    llv = torch.zeros(Cc, device=dev)
    lpo = torch.zeros(Cc, device=dev)
    ms = time_ms(lambda: _capi.call(
        'zshmc_model_kick_drift', qq.data_ptr(), pp.data_ptr(), gg.data_ptr(),
        width, op.data_ptr(), width, 1, pm.data_ptr(), 5000, pl.data_ptr(), 1,
        _capi.ptr(mass), None, 1e-3, 1.0, 1.0, 1.0, Cc, Dm, Dm, llv.data_ptr(),
        lpo.data_ptr(), None, s))
    b = 8 * 4      # rw q, rw p, r grad, r theta twice, w theta
    print('model_kick_drift softmax [%9d, %d]  %8.3f ms  %6.0f GB/s at %d '
          'B/elem (7 distinct passes: %6.0f GB/s)' % (
              Cc, Dm, ms, b * Cc * Dm / ms / 1e6, b,
              28.0 * Cc * Dm / ms / 1e6), flush=True)
    del qq, pp, gg, op, llv, lpo
    torch.cuda.empty_cache()
