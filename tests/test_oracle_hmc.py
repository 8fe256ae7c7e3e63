"""The HMC restatement (oracle/hmc_ref.py) against what the reference pins:
its statistical sampler test (tests/test_mcmc.py:14-62), the gaussian.py
recovery check (examples/toy_examples/gaussian.py:67-72) and the behavioural
quirks of hmc.py listed in SURVEY.md Appendix B.  Transition numerics are
'parity unpinned' (no TensorFlow here); these are the strongest anchors the
reference itself offers."""
import numpy as np
import pytest
from scipy import stats

from oracle import hmc_ref
from oracle.hmc_ref import HMC, DiagNormalModel


def _gaussian(n_x=10, C=1000, **kw):
    stdev = (1 / (np.arange(n_x, dtype=np.float32) + 1)).astype(np.float32)
    m = DiagNormalModel(np.zeros(n_x, np.float32), std=stdev)
    x = np.zeros((C, n_x), np.float32)
    h = HMC(**kw)
    h.sample(m.log_joint, m.grad, [x])
    return h, x, stdev


def test_config1_gaussian_recovery_and_trace_shape():
    h, x, stdev = _gaussian(step_size=1e-3, n_leapfrogs=5,
                            adapt_step_size=True, adapt_mass=True,
                            target_acceptance_rate=0.9, seed=1)
    eps_used, acc, samples = [], [], []
    for i in range(200):
        info = h.step(adapt_step_size=i < 50, adapt_mass=i < 50)
        eps_used.append(float(h.used_step_size))
        acc.append(float(info.acceptance_rate.mean()))
        if i >= 100:
            samples.append(info.samples[0])
    # Appendix B #1: mu = 10*eps0 (not log) -> eps jumps to ~1 after the
    # first adapted iteration, acceptance collapses, then recovers
    assert 0.02 < eps_used[0] < 0.2 and acc[0] > 0.8
    assert abs(eps_used[1] - 1.0) < 0.05 and acc[1] < 0.05
    assert acc[2] < 0.05 and eps_used[2] < eps_used[1]
    # re-initialisation at t == mass_collect_iters (index 9)
    assert eps_used[9] > 3 * eps_used[8]
    assert abs(np.mean(acc[100:]) - 0.9) < 0.03
    s = np.vstack(samples)
    assert np.abs(s.mean(0)).max() < 0.01
    np.testing.assert_allclose(s.std(0), stdev, rtol=0.02)


def test_reference_double_well_statistical_test():
    """tests/test_mcmc.py:14-62 with the reference's own settings."""
    n_chains, n_iters, thinning = 100, 1000, 50
    burnin = n_iters * 2 // 3
    rng = np.random.RandomState(0)

    def log_joint(q):
        x = q[0]
        noise = rng.normal(scale=2.0, size=x.shape).astype(np.float32)
        return np.float32(2) * x ** 2 - x ** 4 + noise

    def grad(q):
        x = q[0]
        return [np.float32(4) * x - np.float32(4) * x ** 3]

    x = np.zeros(n_chains, np.float32)
    # (the reference's test is unseeded and its bound sits inside the spread
    # of the estimate -- 700 thinned draws: over seeds 1..8 the error is
    # 0.025-0.034 on this stream and 0.027-0.036 with 10 Philox rounds)
    h = HMC(step_size=0.01, n_leapfrogs=10, seed=1)
    h.sample(log_joint, grad, [x])
    samples = []
    for t in range(n_iters):
        h.step()
        if t >= burnin and t % thinning == 0:
            samples.append(x.copy())
    samples = np.array(samples).reshape(-1)
    A = 3
    xs = np.linspace(-A, A, 1000)
    pdfs = np.exp(2 * (xs ** 2) - xs ** 4)
    pdfs = pdfs / pdfs.mean() / A / 2
    est = stats.gaussian_kde(samples)(xs)
    assert np.abs(est - pdfs).mean() <= 0.030


def test_leapfrog_evaluates_L_plus_1_gradients_and_2_log_joints():
    """Appendix B #3 / #7: L+1 gradient evaluations, old log-prob recomputed."""
    calls = {'g': 0, 'lp': 0}
    m = DiagNormalModel(np.zeros(3, np.float32), std=np.ones(3, np.float32))

    def lj(q):
        calls['lp'] += 1
        return m.log_joint(q)

    def gr(q):
        calls['g'] += 1
        return m.grad(q)

    x = np.zeros((4, 3), np.float32)
    h = HMC(step_size=0.1, n_leapfrogs=7)
    h.sample(lj, gr, [x])
    calls['lp'] = 0
    h.step()
    assert calls['g'] == 8 and calls['lp'] == 2


def test_flag_false_overwrites_step_size_with_exp_log_eps_bar():
    """Appendix B #4: with the adapt flag False, eps = exp(log_epsilon_bar),
    i.e. 1.0 when never adapted."""
    h, x, _ = _gaussian(n_x=2, C=8, step_size=0.01, n_leapfrogs=2,
                        adapt_step_size=True, seed=0)
    info = h.step(adapt_step_size=False)
    assert float(info.updated_step_size) == 1.0


def test_init_step_size_shrinks_at_most_once_but_grows_repeatedly():
    """Appendix B #2."""
    h, x, _ = _gaussian(n_x=4, C=64, step_size=50.0, n_leapfrogs=2,
                        adapt_step_size=True, seed=0)
    h.step()
    assert h.n_init_trips == 1
    assert np.isclose(h.used_step_size, 50.0 / 1.5)
    h, x, _ = _gaussian(n_x=4, C=64, step_size=1e-4, n_leapfrogs=2,
                        adapt_step_size=True, seed=0)
    h.step()
    assert h.n_init_trips > 5 and h.used_step_size > 1e-3


def test_constructor_and_numeric_errors():
    with pytest.raises(ValueError, match='we should also adapt step size'):
        HMC(adapt_mass=True)
    m = DiagNormalModel(np.zeros(2, np.float32), std=np.ones(2, np.float32))
    x = np.full((3, 2), np.inf, np.float32)
    h = HMC(step_size=0.1, n_leapfrogs=1)
    h.sample(m.log_joint, m.grad, [x])
    with pytest.raises(hmc_ref.NumericError,
                       match='old_log_prob has numeric errors'):
        h.step()
    # non-finite proposal: acceptance 0, state kept (Appendix B #8)
    x = np.ones((3, 2), np.float32)
    h = HMC(step_size=1e20, n_leapfrogs=3)
    h.sample(m.log_joint, m.grad, [x])
    with np.errstate(all='ignore'):
        info = h.step()
    assert np.all(info.acceptance_rate == 0)
    np.testing.assert_array_equal(x, np.ones((3, 2), np.float32))
    with pytest.raises(ValueError, match='at least partially defined'):
        HMC().sample(lambda q: np.float32(0.0), lambda q: q,
                     [np.zeros(3, np.float32)])


def test_mass_is_precision_shared_by_all_chains():
    """Appendix B #5/#6: mass = 1/var over all chain axes; ones while
    int(t) < mass_collect_iters."""
    h, x, stdev = _gaussian(n_x=3, C=4000, step_size=0.05, n_leapfrogs=5,
                            adapt_step_size=True, adapt_mass=True,
                            mass_collect_iters=3, seed=2)
    x[...] = (np.random.RandomState(0).normal(size=x.shape) *
              stdev).astype(np.float32)
    h.step()
    assert np.all(h.last_mass[0] == 1.0) and h.last_mass[0].shape == (1, 3)
    h.step()
    h.step()
    np.testing.assert_allclose(h.last_mass[0].reshape(-1), 1 / stdev ** 2,
                               rtol=0.15)
