#!/usr/bin/env python
"""A/B micro-benchmark of the fused kernel: times zshmc_hmc_diag_normal_step
from each shared library given on the command line (same inputs, interleaved
repetitions).  Usage: python tools/kbench.py lib1.so lib2.so ... [--mass]"""
import ctypes
import sys

import torch

sys.path.insert(0, '.')
from zhusuan_amd import _capi  # noqa: E402


def load(path):
    lib = ctypes.CDLL(path)
    fn = lib.zshmc_hmc_diag_normal_step
    fn.restype, fn.argtypes = _capi.PROTOTYPES['zshmc_hmc_diag_normal_step']
    return fn


def main():
    libs = [a for a in sys.argv[1:] if not a.startswith('--')]
    mass_on = '--mass' in sys.argv
    import os
    C, D, L = 65536, 1024, int(os.environ.get('KB_L', '10'))
    dev = torch.device('cuda', 0)
    logstd = torch.linspace(-1, 1, D, device=dev)
    mean = torch.zeros(D, device=dev)
    mass = torch.exp(-2 * logstd) if mass_on else None
    q0 = torch.randn(C, D, device=dev) * torch.exp(logstd)
    acc = torch.zeros(C, device=dev)
    acc_sum = torch.zeros(1, dtype=torch.float64, device=dev)
    fns = [(p, load(p)) for p in libs]
    stream = torch.cuda.current_stream().cuda_stream
    eps = 0.9 if mass_on else 0.14

    full = os.environ.get('KB_INFO', '1') != '0'   # all HMCInfo outputs, as the API
    extra = [torch.zeros(C, device=dev) for _ in range(4)]
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    eps_dev = torch.full((1,), eps, device=dev)

    def run(fn, q, it):
        if full:
            rc = fn(q.data_ptr(), mean.data_ptr(), logstd.data_ptr(),
                    None if mass is None else mass.data_ptr(), eps_dev.data_ptr(), 0.0, C, D, 0,
                    L, 1, it, 1, acc.data_ptr(), extra[0].data_ptr(), extra[1].data_ptr(),
                    extra[2].data_ptr(), extra[3].data_ptr(), acc_sum.data_ptr(), flags.data_ptr(),
                    stream)
        else:
            rc = fn(q.data_ptr(), mean.data_ptr(), logstd.data_ptr(),
                    None if mass is None else mass.data_ptr(), None, eps, C, D, 0,
                    L, 1, it, 1, acc.data_ptr(), None, None, None, None,
                    acc_sum.data_ptr(), None, stream)
        assert rc == 0
    res = {p: [] for p, _ in fns}
    for rep in range(int(os.environ.get("KB_REPS", "3"))):
        for p, fn in fns:
            q = q0.clone()
            for i in range(5):
                run(fn, q, i)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            n = 50
            e0.record()
            for i in range(n):
                run(fn, q, 10 + i)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            res[p].append(ms)
            last_acc = float(acc.mean().item())
        acc_sum.zero_()
    for p, v in res.items():
        ms = min(v)
        print('%-40s best %.4f ms  (%s)  %.0f GB/s algorithmic  acc %.3f' % (
            p.split('/')[-1], ms, ' '.join('%.4f' % x for x in v),
            8.0 * C * D / ms / 1e6, last_acc))


if __name__ == '__main__':
    main()
