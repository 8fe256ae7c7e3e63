"""PIN of oracle/ais_ref.py (+ oracle/hmc_ref.py under a tempered, per-run
changing log-joint) against a run of the reference's OWN
zhusuan/evaluation.py:AIS -- with its own hmc.py and model layer below it --
over oracle/tf_shim.py (oracle/make_golden_ais.py ->
tests/golden/ais_reference.npz)."""
import os

import numpy as np

from oracle import ais_ref, hmc_ref, philox
import helpers_ais_case as case


def test_oracle_ais_reproduces_reference_run():
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden',
                                'ais_reference.npz'))
    z = np.zeros((case.N_CHAINS, case.D), np.float32)

    def draw_prior(k):
        return [philox.normal_flat(case.GLOBAL_SEED, k, z.size).reshape(z.shape)]

    hmc = hmc_ref.HMC(seed=case.HMC_SEED, **case.HMC_KW)
    ais = ais_ref.AIS(case.log_prior, case.grad_prior, case.log_joint,
                      case.grad_joint, hmc, [z], draw_prior,
                      n_temperatures=case.N_TEMPERATURES,
                      n_adapt=case.N_ADAPT)
    est = ais.run()
    acc = np.stack(ais.acceptance)
    # free-running for 48 transitions: float32 torch-CPU vs NumPy differ in the
    # last bits, so a borderline accept may flip in a chain or two
    close = np.isclose(ais.log_weights, gold['log_weights'], atol=2e-3)
    assert close.mean() >= 0.95, close.mean()
    same_acc = np.isclose(acc, gold['acceptance_rate'], atol=2e-3).all(axis=0)
    assert same_acc.mean() >= 0.95
    np.testing.assert_allclose(est, float(gold['estimate']), atol=0.05)
    np.testing.assert_allclose(float(hmc.step_size),
                               float(gold['final_step_size']), rtol=2e-2)
    # and the estimate is a sane estimate of the exact marginal likelihood
    assert abs(est - float(gold['true_log_marginal'])) < 1.0


def test_oracle_ais_reproduces_reference_lntm_run():
    """The evaluation block of lntm_mcem.py (:116-141): the reference's own
    AIS + hmc.py + `lntm` model function (imported from the unmodified
    example) with the prior of eta as the proposal
    (oracle/make_golden_ais.py -> tests/golden/ais_lntm_reference.npz),
    against oracle/ais_ref.py over the restated topic model."""
    from oracle import distributions_ref as dref
    from oracle.hmc_case_data import lntm_data
    from oracle.make_golden_ais import (
        LNTM_GLOBAL_SEED, LNTM_HMC_KW, LNTM_HMC_SEED, LNTM_N_ADAPT,
        LNTM_N_TEMPERATURES)
    from helpers_hmc_cases import lntm_model
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden',
                                'ais_lntm_reference.npz'))
    beta, x, eta_mean, eta_logstd, eta0 = lntm_data()
    log_joint, grad_joint = lntm_model(beta, x, eta_mean, eta_logstd)
    prior = dref.Normal(eta_mean, logstd=eta_logstd, group_ndims=1)
    eta = np.zeros_like(eta0)

    def draw_prior(k):
        # every reset samples eta AND beta (the model function reads
        # beta.tensor): two sampling ops per reset, eta first
        z = philox.normal_flat(LNTM_GLOBAL_SEED, 2 * k, eta.size)
        return [(z.reshape(eta.shape) * np.exp(eta_logstd) + eta_mean)
                .astype(np.float32)]

    hmc = hmc_ref.HMC(seed=LNTM_HMC_SEED, **LNTM_HMC_KW)
    ais = ais_ref.AIS(lambda q: prior.log_prob(q[0]),
                      lambda q: [prior.grad_given(q[0])],
                      log_joint, grad_joint, hmc, [eta], draw_prior,
                      n_temperatures=LNTM_N_TEMPERATURES,
                      n_adapt=LNTM_N_ADAPT)
    est = ais.run()
    acc = np.stack(ais.acceptance)
    # 14 free-running transitions on 12 (chain, document) rows (adaptation
    # couples every row through eps, and the leapfrog at eps ~ 0.8 amplifies
    # rounding ~4x per transition: 30 + 6 transitions part ways at the 21st): float32
    # torch-CPU vs NumPy differ in the last bits; allow one row to part ways
    close = np.isclose(ais.log_weights, gold['log_weights'], atol=5e-3)
    assert close.mean() >= 0.9, (close.mean(), ais.log_weights,
                                 gold['log_weights'])
    same_acc = np.isclose(acc, gold['acceptance_rate'], atol=3e-3).all(axis=0)
    assert same_acc.mean() >= 0.9
    np.testing.assert_allclose(est, float(gold['estimate']), atol=0.05)
    np.testing.assert_allclose(float(hmc.step_size),
                               float(gold['final_step_size']), rtol=2e-2)
