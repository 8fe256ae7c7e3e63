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


# ---- the section 8(f) kernels: what bounds each, achieved vs that bound -----
def time_ms(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# MultivariateNormalCholesky log_prob + grad: n^2 FMAs per vector out of LDS,
# one lane per vector: compulsory HBM traffic 12n B per vector
for n_dim in (16, 64, 128):
    R = 1 << 20 if n_dim <= 64 else 1 << 18
    x = torch.randn(R, n_dim, device=dev)
    mu = torch.zeros(1, n_dim, device=dev)
    tril = torch.tril(torch.randn(n_dim, n_dim, device=dev)) * 0.1 + \
        torch.eye(n_dim, device=dev)
    lpv = torch.empty(R, device=dev)
    gx = torch.empty(R, n_dim, device=dev)
    ms = time_ms(lambda: _capi.call(
        'zshmc_mvn_tril_log_prob', x.data_ptr(), mu.data_ptr(), tril.data_ptr(),
        R, n_dim, 1, 1, lpv.data_ptr(), gx.data_ptr(), None, s))
    fma = 2.0 * R * n_dim * n_dim          # forward + back substitution
    lds = fma * 4                          # one 4-B LDS operand per FMA
    print('mvn_tril_log_prob+grad n=%-4d %7.3f ms  %6.0f GB/s HBM (8n B/vector)  '
          '%6.2f TFMA/s  %5.1f TB/s LDS operand rate' % (
              n_dim, ms, 8.0 * R * n_dim / ms / 1e6, fma / ms / 1e9,
              lds / ms / 1e9))

# gathered row dots (pmf_hmc.py shape: K chains, E pairs, D = 32)
K, NU, NV, E, Dd = 8, 6040, 3706, 900000, 32
u = torch.randn(K, NU, Dd, device=dev)
v = torch.randn(K, NV, Dd, device=dev)
su = torch.randint(0, NU, (E,), device=dev, dtype=torch.int32)
sv = torch.randint(0, NV, (E,), device=dev, dtype=torch.int32)
out = torch.empty(K, E, device=dev)
ms = time_ms(lambda: _capi.call(
    'zshmc_gather_dot', u.data_ptr(), v.data_ptr(), su.data_ptr(), sv.data_ptr(),
    K, NU, NV, E, Dd, out.data_ptr(), s))
gathered = K * E * (2 * Dd * 4 + 4)
print('gather_dot fwd K=%d E=%d D=%d      %7.3f ms  %6.0f GB/s gathered (%.0f MB; '
      'factor tables %.1f MB: L2 / Infinity-Cache resident)' % (
          K, E, Dd, ms, gathered / ms / 1e6, gathered / 1e6,
          (u.numel() + v.numel()) * 4 / 1e6))

# batched ESS: n draws x series, float64 autocovariance sums, O(n * lags)
n_draws, n_series = 300, 1 << 20
draws = torch.randn(n_draws, n_series, device=dev).cumsum(0) * 0.1 + \
    torch.randn(n_draws, n_series, device=dev)
ess = torch.empty(n_series, device=dev)
ms = time_ms(lambda: _capi.call('zshmc_ess_series', draws.data_ptr(), n_draws,
                                n_series, ess.data_ptr(), s), reps=3)
print('ess_series n=%d x %d series       %7.3f ms  %6.0f GB/s (one read of the '
      'record), %.2f G series-draws/s' % (
          n_draws, n_series, ms, 4.0 * n_draws * n_series / ms / 1e6,
          n_draws * n_series / ms / 1e6))

# the native plans' leapfrog step (csrc/hmc_model.hip) at the config-3 / 5 shapes
for (Cc, Dm, width, soft) in ((32768, 256, 256, 0), (1280000, 128, 128, 1)):
    qq = torch.randn(Cc, Dm, device=dev) * 0.1
    pp = torch.randn(Cc, Dm, device=dev)
    gg = torch.randn(Cc, width, device=dev)
    op = torch.softmax(qq, -1).contiguous() if soft else None
    pm = torch.zeros(1, Dm, device=dev)
    pl = torch.zeros(1, Dm, device=dev)
    llv = torch.zeros(Cc, device=dev)
    lpo = torch.zeros(Cc, device=dev)
    ms = time_ms(lambda: _capi.call(
        'zshmc_model_kick_drift', qq.data_ptr(), pp.data_ptr(), gg.data_ptr(),
        width, None if op is None else op.data_ptr(), width, soft, pm.data_ptr(), 1,
        pl.data_ptr(), 1, None, None, 1e-3, 1.0, 1.0, 1.0, Cc, Dm, Dm, llv.data_ptr(),
        lpo.data_ptr(), None, s))
    b = (5 + (3 if soft else 0)) * 4      # rw q, rw p, r grad (+ r theta, r theta, w theta)
    print('model_kick_drift %s [%d, %d]   %7.3f ms  %6.0f GB/s (%d B/elem)' % (
        'softmax ' if soft else 'identity', Cc, Dm, ms, b * Cc * Dm / ms / 1e6, b))
