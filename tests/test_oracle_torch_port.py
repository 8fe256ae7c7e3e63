"""oracle/hmc_torch_port.py (the multi-threaded torch-CPU port bench.py times
as one of its CPU baselines) against the NumPy oracle on the same Philox
numbers."""
import numpy as np
import torch

from oracle import hmc_torch_port, philox
from oracle.hmc_ref import HMC as RefHMC, DiagNormalModel


def test_torch_port_matches_numpy_oracle():
    C, D, L, eps, seed = 40, 36, 4, 0.21, 17
    rng = np.random.RandomState(0)
    mean = rng.normal(size=D).astype(np.float32)
    logstd = rng.uniform(-0.6, 0.4, size=D).astype(np.float32)
    q0 = (mean + rng.normal(size=(C, D))).astype(np.float32)
    model = DiagNormalModel(mean, logstd=logstd)
    xr = q0.copy()
    ref = RefHMC(step_size=eps, n_leapfrogs=L, seed=seed)
    ref.sample(model.log_joint, model.grad, [xr])
    rinfo = ref.step()
    z = torch.from_numpy(philox.normal_chain_major(seed, 1, C, D))
    u = torch.from_numpy(philox.uniform_per_chain(seed, 1, C))
    q = torch.from_numpy(q0.copy())
    acc, h0, h1, lp0, lp = hmc_torch_port.transition(
        q, torch.from_numpy(mean), torch.from_numpy(logstd), eps, L, z, u)
    np.testing.assert_allclose(acc.numpy(), rinfo.acceptance_rate, atol=2e-5)
    np.testing.assert_allclose(h0.numpy(), rinfo.orig_hamiltonian, rtol=2e-6,
                               atol=1e-4)
    np.testing.assert_allclose(h1.numpy(), rinfo.hamiltonian, rtol=2e-6,
                               atol=1e-4)
    np.testing.assert_allclose(lp.numpy(), rinfo.log_prob, rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(q.numpy(), xr, atol=1e-5)
