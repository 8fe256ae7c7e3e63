#!/usr/bin/env python
"""A/B correctness: the ring kernel vs the register-prefetch kernel of the SAME
library (ZSHMC_FUSED_RING toggled per process is static, so two libraries are
loaded: argv[1] = reference build (-DZS_NO_RING or older), argv[2] = ring)."""
import ctypes, sys
import torch
sys.path.insert(0, '.')
from zhusuan_amd import _capi


def load(path):
    lib = ctypes.CDLL(path)
    fn = lib.zshmc_hmc_diag_normal_step
    fn.restype, fn.argtypes = _capi.PROTOTYPES['zshmc_hmc_diag_normal_step']
    lib.zshmc_last_error.restype = ctypes.c_char_p
    return lib, fn


def run(fn, lib, C, D, L, mass_on, it, commit=1, eps=0.14):
    dev = torch.device('cuda', 0)
    g = torch.Generator(device='cpu').manual_seed(1)
    logstd = torch.linspace(-1, 1, D).to(dev)
    mean = torch.randn(D, generator=g).to(dev)
    mass = (torch.exp(-2 * logstd) if mass_on else None)
    q = (torch.randn(C, D, generator=g) * torch.exp(logstd.cpu()) + mean.cpu()).to(dev)
    info = [torch.full((C,), -7.0, device=dev) for _ in range(5)]
    acc_sum = torch.zeros(_capi.STATS_WORDS, dtype=torch.float64, device=dev)
    ws = torch.zeros(_capi.LINK_WORKSPACE_BYTES, dtype=torch.uint8, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    link = _capi.AdaptLink(stats=acc_sum.data_ptr(), workspace=ws.data_ptr(),
                           n_chains_global=C, used_step_size=float('nan'))
    s = torch.cuda.current_stream().cuda_stream
    for t in range(it):
        rc = fn(q.data_ptr(), mean.data_ptr(), logstd.data_ptr(),
                None if mass is None else mass.data_ptr(), eps, C, D, 5, L, 1234567, t, commit,
                info[0].data_ptr(), info[1].data_ptr(), info[2].data_ptr(), info[3].data_ptr(),
                info[4].data_ptr(), flags.data_ptr(), ctypes.byref(link), s)
        if rc != 0:
            raise RuntimeError(lib.zshmc_last_error().decode())
    torch.cuda.synchronize()
    return [q.cpu()] + [x.cpu() for x in info] + [acc_sum.cpu(), flags.cpu()]


def main():
    la, fa = load(sys.argv[1])
    lb, fb = load(sys.argv[2])
    cases = [(4, 1024, 10, False, 1), (7, 1024, 3, False, 2), (1000, 1024, 10, False, 3),
             (5000, 1024, 10, True, 3), (333, 516, 5, False, 2), (333, 260, 5, True, 2),
             (100, 132, 2, False, 2), (257, 2048, 4, False, 2), (257, 1540, 4, True, 2),
             (70000, 1024, 10, False, 1), (300000, 256, 3, False, 1), (64, 1024, 10, False, 1)]
    if len(sys.argv) > 3:
        C, D, L, m, it = [int(x) for x in sys.argv[3].split(',')]
        cases = [(C, D, L, bool(m), it)]
    for C, D, L, m, it in cases:
        a = run(fa, la, C, D, L, m, it)
        b = run(fb, lb, C, D, L, m, it)
        names = ['q', 'acc', 'h_old', 'h_new', 'lp_old', 'lp', 'acc_sum', 'flags']
        if all(torch.equal(x, y) for x, y in zip(a, b)):
            print('C=%d D=%d L=%d mass=%d it=%d: BIT-EXACT' % (C, D, L, m, it), flush=True)
            continue
        # different summation tree (DPP scan vs xor butterfly): energies agree
        # to rounding; a chain whose accept decision flipped must be borderline
        rel = lambda x, y: float(((x.double() - y.double()).abs() / (1.0 + y.double().abs())).max())
        e = {n: rel(x, y) for n, x, y in zip(names[1:5], a[1:5], b[1:5])}
        rows = (a[0] != b[0]).any(dim=1)
        n_flip = int(rows.sum())
        print('C=%d D=%d L=%d mass=%d it=%d: max rel err %s; rows differing %d of %d (%.4f%%)' % (
            C, D, L, m, it, ' '.join('%s=%.1e' % kv for kv in e.items()), n_flip, C,
            100.0 * n_flip / C), flush=True)


main()
