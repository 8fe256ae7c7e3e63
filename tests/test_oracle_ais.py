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
