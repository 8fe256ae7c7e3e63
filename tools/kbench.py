#!/usr/bin/env python
"""A/B micro-benchmark of the fused kernel: times zshmc_hmc_diag_normal_step
from each shared library given on the command line (same inputs, interleaved
repetitions).  Libraries of ABI 0.1 (round 1: step_size_dev / acc_sum
arguments) and 0.2 (zshmc_adapt_link) are both understood, so that a build of
an earlier commit can sit in the same table.
Usage: python tools/kbench.py lib1.so lib2.so ... [--mass] [--mean] [--adapt]
                              [--colstats]
  --mean   non-zero mean vector (0.2: the mean-tile instantiation)
  --adapt  0.2 only: every launch carries a pending dual-averaging update
  --colstats  0.3 only: the launch also leaves the column sums of its end
           state (zshmc_adapt_link.colstats_*)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, '.')
from zhusuan_amd import _capi  # noqa: E402

_p = ctypes.c_void_p
OLD_ARGS = [_p, _p, _p, _p, _p, ctypes.c_float, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int64, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32,
            ctypes.c_int, _p, _p, _p, _p, _p, _p, _p, _p]


def load(path):
    lib = ctypes.CDLL(path)
    lib.zshmc_version.restype = ctypes.c_int
    ver = lib.zshmc_version()
    fn = lib.zshmc_hmc_diag_normal_step
    fn.restype = ctypes.c_int
    fn.argtypes = OLD_ARGS if ver < 200 else \
        _capi.PROTOTYPES['zshmc_hmc_diag_normal_step'][1]
    return ver, fn


def main():
    libs = [a for a in sys.argv[1:] if not a.startswith('--')]
    mass_on = '--mass' in sys.argv
    mean_on = '--mean' in sys.argv
    adapt_on = '--adapt' in sys.argv
    cs_on = '--colstats' in sys.argv
    C = int(os.environ.get('KB_C', '65536'))
    D = int(os.environ.get('KB_D', '1024'))
    L = int(os.environ.get('KB_L', '10'))
    dev = torch.device('cuda', 0)
    logstd = torch.linspace(-1, 1, D, device=dev)
    mean = torch.linspace(-1, 1, D, device=dev) if mean_on else \
        torch.zeros(D, device=dev)
    mass = torch.exp(-2 * logstd) if mass_on else None
    q0 = torch.randn(C, D, device=dev) * torch.exp(logstd) + mean
    info = [torch.zeros(C, device=dev) for _ in range(5)]
    acc_sum = torch.zeros(1, dtype=torch.float64, device=dev)
    stats = torch.zeros(_capi.STATS_WORDS, dtype=torch.float64, device=dev)
    ws = torch.zeros(_capi.LINK_WORKSPACE_BYTES, dtype=torch.uint8, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    fns = [(p,) + load(p) for p in libs]
    stream = torch.cuda.current_stream().cuda_stream
    eps = 0.9 if mass_on else 0.14
    state = torch.zeros(_capi.STATE_WORDS, device=dev)
    state[_capi.ST_STEP_SIZE] = eps
    state[_capi.ST_LOG_EPS_BAR] = float(torch.log(torch.tensor(eps)))

    link = _capi.AdaptLink()
    link.state, link.stats = state.data_ptr(), stats.data_ptr()
    link.workspace, link.n_chains_global = ws.data_ptr(), C
    link.pending = _capi.PEND_HOLD if adapt_on else _capi.PEND_NONE
    link.fresh_start, link.used_step_size = 0, float('nan')
    link.delta, link.gamma, link.t0, link.kappa, link.mu = \
        0.8, 0.05, 100.0, 0.75, 10 * eps

    if cs_on:
        cs_mean = torch.zeros(D, device=dev)
        cs_parts = torch.zeros(512, 2 * D, dtype=torch.float64, device=dev)
        link.colstats_mean = cs_mean.data_ptr()
        link.colstats_parts = cs_parts.data_ptr()

    def run(ver, fn, q, it):
        mptr = None if mass is None else mass.data_ptr()
        if ver < 200:
            rc = fn(q.data_ptr(), mean.data_ptr(), logstd.data_ptr(), mptr,
                    state.data_ptr(), 0.0, C, D, 0, L, 1, it, 1,
                    *[x.data_ptr() for x in info], acc_sum.data_ptr(),
                    flags.data_ptr(), stream)
        else:
            rc = fn(q.data_ptr(), mean.data_ptr() if mean_on else None,
                    logstd.data_ptr(), mptr, 0.0, C, D, 0, L, 1, it, 1,
                    *[x.data_ptr() for x in info], flags.data_ptr(),
                    ctypes.byref(link), stream)
        assert rc == 0
    res = {p: [] for p, _, _ in fns}
    last_acc = {}
    for rep in range(int(os.environ.get("KB_REPS", "3"))):
        for p, ver, fn in fns:
            q = q0.clone()
            for i in range(5):
                run(ver, fn, q, i)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            n = 50
            e0.record()
            for i in range(n):
                run(ver, fn, q, 10 + i)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            res[p].append(ms)
            last_acc[p] = float(info[0].mean().item())
        acc_sum.zero_()
    for p, v in res.items():
        ms = min(v)
        print('%-34s best %.4f ms  (%s)  %.0f GB/s algorithmic = %.3f of 8 TB/s  acc %.3f' % (
            os.path.basename(p), ms, ' '.join('%.4f' % x for x in v),
            8.0 * C * D / ms / 1e6, 8.0 * C * D / ms / 1e6 / 8000.0,
            last_acc[p]))


if __name__ == '__main__':
    main()
