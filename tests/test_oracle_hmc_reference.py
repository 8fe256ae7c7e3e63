"""PIN of oracle/hmc_ref.py (leapfrog, MH, dual averaging, EWMV mass,
step-size search, Appendix-B quirks) against traces produced by the
reference's OWN zhusuan/hmc.py, executed unmodified over the eager TF-API
shim oracle/tf_shim.py on the shared Philox stream
(oracle/make_golden_hmc.py -> tests/golden/hmc_reference_traces.npz)."""
import os

import numpy as np
import pytest

from oracle import hmc_ref
from helpers_hmc_cases import cases, cases_r3


@pytest.fixture(scope='module')
def traces():
    return np.load(os.path.join(os.path.dirname(__file__), 'golden',
                                'hmc_reference_traces.npz'))


@pytest.fixture(scope='module')
def traces_r3():
    return np.load(os.path.join(os.path.dirname(__file__), 'golden',
                                'hmc_reference_traces_r3.npz'))


@pytest.mark.parametrize('case', list(cases_r3()), ids=lambda c: c['name'])
def test_oracle_reproduces_reference_hmc_traces_several_latents(traces_r3,
                                                                case):
    """Round 3 (oracle/make_golden_hmc_r3.py): the reference's own hmc.py on
    `matmul(u, X1^T) + matmul(v, X2^T) + expand_dims(b, 1)` -- two weight
    blocks and a per-chain scalar intercept, step-size and mass adaptation
    fed per run.  The device's packed native plan is compared with this same
    oracle (tests/test_gpu_native_plan_limits.py)."""
    test_oracle_reproduces_reference_hmc_traces(traces_r3, case)


@pytest.mark.parametrize('case', list(cases()), ids=lambda c: c['name'])
def test_oracle_reproduces_reference_hmc_traces(traces, case):
    name = case['name']
    qs = [traces['%s/q0_%s' % (name, k)].copy() for k in case['latent_names']]
    log_joint, grad = case['model']
    ref = hmc_ref.HMC(seed=case['seed'], **case['hmc_kwargs'])
    ref.sample(log_joint, grad, qs)
    for i in range(case['n_iters']):
        f_ss, f_m = case['flags'](i)
        info = ref.step(adapt_step_size=f_ss, adapt_mass=f_m)
        # same random numbers went in
        for k, nm in enumerate(case['latent_names']):
            np.testing.assert_allclose(info.init_momentum[k],
                                       traces['%s/p0_%s' % (name, nm)][i],
                                       rtol=1e-6, atol=1e-7)
        # float32 torch-CPU vs float32 NumPy: reductions / exp differ in the
        # last bits; energies are O(10-100)
        for f in ('orig_hamiltonian', 'hamiltonian', 'orig_log_prob',
                  'log_prob'):
            np.testing.assert_allclose(getattr(info, f),
                                       traces['%s/%s' % (name, f)][i],
                                       rtol=5e-5, atol=3e-5, err_msg='%s it %d' % (f, i))
        np.testing.assert_allclose(info.acceptance_rate,
                                   traces[name + '/acceptance_rate'][i],
                                   rtol=0, atol=4e-4)   # |H| ~ 300 at D = 260:
                                                        # 1 ulp of H is 3e-5
        np.testing.assert_allclose(info.updated_step_size,
                                   traces[name + '/updated_step_size'][i],
                                   rtol=5e-5)   # mean(acc) in float32: torch
                                                # and NumPy sum in different
                                                # orders; dual averaging
                                                # multiplies that by sqrt(t)/gamma
        for k, nm in enumerate(case['latent_names']):
            want = traces['%s/q_%s' % (name, nm)][i]
            # (1/var masses and exp(-2 logstd) differ in the last bit between
            # NumPy and torch; L drifts with eps ~ 1 and 1/m up to 13 carry
            # that to a few 1e-4 absolute on |q| ~ 8; the energies above, which
            # decide acceptance, agree to 1-2 ulp.)
            np.testing.assert_allclose(qs[k], want, rtol=2e-4, atol=1e-3)
            # teacher forcing: continue from the reference's state so that
            # float32 rounding differences do not compound over iterations
            qs[k][...] = want
    assert float(ref.t) == float(traces[name + '/t'])


def test_traces_exercise_the_quirks(traces):
    """The golden run itself shows the Appendix-B behaviour the oracle must
    keep: eps jumps to ~exp(10*eps0) after the first adapted iteration
    (mu = 10*eps0 used as a log step size, hmc.py:79), the search at t = 1 and
    again at t = mass_collect_iters, exp(log_epsilon_bar) once the flag is
    off (hmc.py:108-110)."""
    eps = traces['gauss_adapt/updated_step_size']
    assert 0.9 < eps[0] < 1.2            # B#1: ~exp(0.01 - small)
    assert eps[16] == eps[17] == eps[21]  # flag off: frozen exp(log_eps_bar)
    acc = traces['gauss_adapt/acceptance_rate']
    assert acc[1].mean() < 0.05          # eps ~ 1 on stdev 1/10: all rejected
