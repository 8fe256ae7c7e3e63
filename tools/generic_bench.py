#!/usr/bin/env python
"""Achieved HBM bandwidth of the element-wise kernels behind the generic HMC
plan and the distribution ops (csrc/hmc_generic.hip, distributions.hip,
sgmcmc.hip) at [65536, 1024] float32: algorithmic bytes / HIP-event time."""
import sys
import torch
sys.path.insert(0, '.')
from zhusuan_amd import _capi  # noqa: E402

C, D = 65536, 1024
dev = torch.device('cuda', 0)
q = torch.randn(C, D, device=dev)
p = torch.randn(C, D, device=dev)
g = torch.randn(C, D, device=dev)
qn = torch.randn(C, D, device=dev)
mean = torch.zeros(D, device=dev)
logstd = torch.linspace(-1, 1, D, device=dev)
kin = torch.zeros(C, device=dev)
lp = torch.zeros(C, device=dev)
gout = torch.ones(C, device=dev)
acc = torch.randint(0, 2, (C,), device=dev, dtype=torch.uint8)
s = torch.cuda.current_stream().cuda_stream
N = C * D


def timeit(name, bytes_per_elem, fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gbs = bytes_per_elem * N / ms / 1e6
    print('%-34s %7.3f ms  %6.0f GB/s  (%4.1f %% of 8 TB/s, %d B/elem)' % (
        name, ms, gbs, gbs / 80.0, bytes_per_elem))


timeit('momentum (write p)', 4, lambda: _capi.call(
    'zshmc_momentum', p.data_ptr(), None, C, D, 0, 1, 3, 0, kin.data_ptr(), s))
timeit('kick_drift (rw p, rw q, r grad)', 20, lambda: _capi.call(
    'zshmc_kick_drift', q.data_ptr(), p.data_ptr(), g.data_ptr(), None, None,
    1e-3, 1.0, 1.0, C, D, kin.data_ptr(), s))
timeit('select_rows (r q_new, w q, ~50%)', 4, lambda: _capi.call(
    'zshmc_select_rows', q.data_ptr(), qn.data_ptr(), acc.data_ptr(), C, D, s))
timeit('normal_log_prob rowsum (r x)', 4, lambda: _capi.call(
    'zshmc_normal_log_prob', q.data_ptr(), mean.data_ptr(), logstd.data_ptr(),
    lp.data_ptr(), C, D, _capi.BCAST_ROW, _capi.BCAST_ROW, 1, s))
timeit('normal_log_prob_grad (r x, w gx)', 8, lambda: _capi.call(
    'zshmc_normal_log_prob_grad', q.data_ptr(), mean.data_ptr(), logstd.data_ptr(),
    gout.data_ptr(), g.data_ptr(), None, None, C, D, _capi.BCAST_ROW,
    _capi.BCAST_ROW, 1, s))
timeit('sgld_update (rw q, r grad)', 12, lambda: _capi.call(
    'zshmc_sgld_update', q.data_ptr(), g.data_ptr(), None, 1e-4, 0.0, 0.0, N, 1, 0, 0, s))
timeit('sghmc_update (rw q, rw v, r grad)', 20, lambda: _capi.call(
    'zshmc_sghmc_update', q.data_ptr(), p.data_ptr(), g.data_ptr(), N, 1e-4, 0.1,
    1e-3, 1, 1, 0, 0, None, s))

# adaptation / softmax-family kernels
em = torch.zeros(D, device=dev)
colsum = torch.zeros(2 * D, dtype=torch.float64, device=dev)
timeit('mass_colstats (r q)', 4, lambda: _capi.call(
    'zshmc_mass_colstats', q.data_ptr(), em.data_ptr(), C, D, colsum.data_ptr(), s))
labels = torch.randint(0, D, (C,), device=dev, dtype=torch.int64)
timeit('categorical_log_prob (r logits)', 4, lambda: _capi.call(
    'zshmc_categorical_log_prob', q.data_ptr(), labels.data_ptr(), lp.data_ptr(), C, D, s))
timeit('categorical_log_prob_grad (r,w)', 8, lambda: _capi.call(
    'zshmc_categorical_log_prob_grad', q.data_ptr(), labels.data_ptr(), gout.data_ptr(),
    g.data_ptr(), C, D, s))
timeit('unnorm_multinomial_lp (r l, r x)', 8, lambda: _capi.call(
    'zshmc_unnormalized_multinomial_log_prob', q.data_ptr(), p.data_ptr(), lp.data_ptr(),
    C, D, 1, s))
timeit('unnorm_multinomial_grad (r,r,w)', 12, lambda: _capi.call(
    'zshmc_unnormalized_multinomial_log_prob_grad', q.data_ptr(), p.data_ptr(),
    gout.data_ptr(), g.data_ptr(), C, D, 1, s))
