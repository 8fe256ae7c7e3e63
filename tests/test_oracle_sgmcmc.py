"""CPU checks of the SGMCMC oracle (oracle/sgmcmc_ref.py): the restated
updates reproduce sgmcmc.py's equations on hand-computed cases, and each
sampler's stationary distribution on a standard normal target is right (the
property the reference's own tests/test_mcmc.py:65-88 checks statistically)."""
import numpy as np
import pytest

from oracle import philox, sgmcmc_ref as ref

F32 = np.float32


def _grad_std_normal(qs):
    return [-q for q in qs]


def test_sgld_one_step_matches_equation():
    q = np.array([0.5, -1.0, 2.0, 0.1, 0.3], F32)
    s = ref.SGLD(learning_rate=0.04, seed=7).sample(_grad_std_normal, [q.copy()])
    z = philox.normal_flat(7, 0, 5, stream=3)
    s.step()
    want = q + F32(0.5) * F32(0.04) * (-q) + z * np.sqrt(F32(0.04))
    np.testing.assert_allclose(s.qs[0], want, rtol=1e-6)
    assert s.t == 1


def test_psgld_preconditioner_state():
    q = np.array([1.0, -2.0], F32)
    s = ref.PSGLD(learning_rate=0.01, seed=1).sample(_grad_std_normal, [q.copy()])
    s.step()
    np.testing.assert_allclose(s.vs[0], 0.1 * q * q, rtol=1e-6)   # 0.9*0 + 0.1*g^2


def test_sghmc_resamples_on_first_run_and_every_n():
    s = ref.SGHMC(learning_rate=0.01, n_iter_resample_v=3, second_order=False,
                  friction=1.0, variance_estimate=1.0, seed=3)
    s.sample(lambda qs: [np.zeros_like(q) for q in qs], [np.zeros(8, F32)])
    # alpha = 1, beta = alpha, zero gradient: v' = 0*v + 0 + 0 -> 0 always
    info = s.step()
    np.testing.assert_array_equal(s.vs[0], 0)
    assert info['mean_k'][0] == 0
    # with alpha = 0 (no decay, no noise) the momentum changes only at t % 3 == 0
    s = ref.SGHMC(learning_rate=0.01, n_iter_resample_v=3, second_order=False,
                  friction=0.0, variance_estimate=0.0, seed=3)
    s.sample(lambda qs: [np.zeros_like(q) for q in qs], [np.zeros(8, F32)])
    vs = []
    for _ in range(7):
        s.step()
        vs.append(s.vs[0].copy())
    assert np.array_equal(vs[0], vs[1]) and np.array_equal(vs[1], vs[2])
    assert not np.array_equal(vs[2], vs[3])
    assert np.array_equal(vs[3], vs[5]) and not np.array_equal(vs[5], vs[6])


@pytest.mark.parametrize('make', [
    lambda: ref.SGLD(0.05, seed=11),
    lambda: ref.PSGLD(0.05, epsilon=1.0, seed=12),
    lambda: ref.SGHMC(0.05, friction=0.3, n_iter_resample_v=50,
                      second_order=False, seed=13),
    lambda: ref.SGHMC(0.05, friction=0.3, n_iter_resample_v=50,
                      second_order=True, seed=14),
    lambda: ref.SGNHT(0.05, variance_extra=0.1, second_order=True,
                      use_vector_alpha=False, seed=15),
    lambda: ref.SGNHT(0.01, variance_extra=0.1, second_order=False,
                      use_vector_alpha=True, seed=16),
])
def test_stationary_distribution_is_standard_normal(make):
    s = make().sample(_grad_std_normal, [np.zeros(4000, F32)])
    acc = []
    for it in range(400):
        s.step()
        if it >= 200 and it % 20 == 0:
            acc.append(s.qs[0].copy())
    x = np.concatenate(acc)
    assert abs(x.mean()) < 0.05
    # PSGLD's preconditioner without the Gamma correction term is biased by
    # design (sgmcmc.py:207-253 implements Eq. 4-5 only; with the default
    # epsilon = 1e-3 an element near a zero gradient gets G ~ 1000): looser
    # bound, larger epsilon
    tol = 0.25 if isinstance(s, ref.PSGLD) else 0.12
    assert abs(x.std() - 1.0) < tol, x.std()
