"""The C + OpenMP restatement of the diag-Normal transition
(oracle/c/hmc_diag_normal_port.c, bench.py's all-cores CPU baseline) against
the NumPy oracle (oracle/hmc_ref.py), same Philox stream."""
import numpy as np
import pytest

from oracle import hmc_c, philox
from oracle.hmc_ref import HMC as RefHMC, DiagNormalModel


@pytest.mark.parametrize('C,D,L', [(37, 10, 5), (64, 1024, 10), (5, 7, 1)])
def test_c_port_matches_numpy_oracle(C, D, L):
    rng = np.random.RandomState(C + D)
    logstd = np.linspace(-1, 1, D).astype(np.float32)
    mean = rng.normal(size=D).astype(np.float32)
    q0 = (mean + rng.normal(size=(C, D)) * np.exp(logstd)).astype(np.float32)
    eps, seed = 0.11, 12345
    model = DiagNormalModel(mean, logstd=logstd)
    qr = q0.copy()
    ref = RefHMC(step_size=eps, n_leapfrogs=L, seed=seed)
    ref.sample(model.log_joint, model.grad, [qr])
    qc = q0.copy()
    for it in (1, 2, 3):
        rinfo = ref.step()
        info, bad = hmc_c.step(qc, mean, logstd, L, eps, seed, it, n_threads=3)
        assert not bad
        scale = max(1.0, float(np.abs(rinfo.orig_hamiltonian).max()))
        tol = 2e-5 * scale + 1e-4
        np.testing.assert_allclose(info['orig_log_prob'], rinfo.orig_log_prob,
                                   rtol=0, atol=tol)
        np.testing.assert_allclose(info['orig_hamiltonian'],
                                   rinfo.orig_hamiltonian, rtol=0, atol=tol)
        np.testing.assert_allclose(info['hamiltonian'], rinfo.hamiltonian,
                                   rtol=0, atol=2 * tol)
        np.testing.assert_allclose(info['acceptance_rate'],
                                   rinfo.acceptance_rate, rtol=0, atol=6 * tol)
        # chains whose accept decision is numerically borderline may flip
        u = philox.uniform_per_chain(seed, it, C)
        gap = np.abs(info['acceptance_rate'] - rinfo.acceptance_rate)
        firm = np.abs(u - rinfo.acceptance_rate) > 4 * gap + 1e-6
        np.testing.assert_allclose(qc[firm], qr[firm], rtol=2e-5, atol=2e-5)
        assert firm.mean() > 0.9
        qc[...] = qr            # keep the two on the same state


def test_c_port_thread_count_invariance_and_bad_start():
    D, C = 33, 200
    logstd = np.zeros(D, np.float32)
    mean = np.zeros(D, np.float32)
    q = np.random.RandomState(0).normal(size=(C, D)).astype(np.float32)
    a, b = q.copy(), q.copy()
    ia, _ = hmc_c.step(a, mean, logstd, 4, 0.2, 9, 1, n_threads=1)
    ib, _ = hmc_c.step(b, mean, logstd, 4, 0.2, 9, 1, n_threads=4)
    assert np.array_equal(a, b)
    assert all(np.array_equal(ia[k], ib[k]) for k in ia)
    # chain offset shifts the stream exactly
    c = q[100:].copy()
    ic, _ = hmc_c.step(c, mean, logstd, 4, 0.2, 9, 1, chain_offset=100)
    assert np.array_equal(c, a[100:])
    q[3, 0] = np.inf
    _, bad = hmc_c.step(q, mean, logstd, 4, 0.2, 9, 1)
    assert bad


def test_c_port_free_run_follows_the_numpy_oracle():
    """hmc_c.DiagNormalFreeRun (the C transition + the step-size search +
    oracle/hmc_ref.py's tuner) against hmc_ref.HMC free-running from the same
    seed: search at t == 1, 30 adaptive transitions through the mu = 10 eps0
    transient (acceptance 0 for a few iterations), 10 with adaptation held."""
    C, D = 256, 64
    logstd = np.linspace(-1, 1, D).astype(np.float32)
    mean = np.zeros(D, np.float32)
    q = np.zeros((C, D), np.float32)
    fr = hmc_c.DiagNormalFreeRun(q, mean, logstd, 0.05, 10, seed=1)
    m = DiagNormalModel(mean, logstd=logstd)
    xr = np.zeros((C, D), np.float32)
    ref = RefHMC(step_size=0.05, n_leapfrogs=10, adapt_step_size=True,
                 seed=1)
    ref.sample(m.log_joint, m.grad, [xr])
    for i in range(40):
        a = i < 30
        info, acc = fr.run(a)
        ri = ref.step(adapt_step_size=a)
        if i == 0:
            assert fr.n_init_trips == ref.n_init_trips
        np.testing.assert_allclose(acc, np.mean(ri.acceptance_rate),
                                   atol=2e-5)
        np.testing.assert_allclose(fr.step_size, ref.step_size, rtol=1e-5)
    same = np.isclose(q, xr, rtol=1e-4, atol=1e-4).all(axis=1)
    assert same.mean() > 0.97
