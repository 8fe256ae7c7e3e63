"""The AIS case of tests/golden/ais_reference.npz (oracle/make_golden_ais.py:
the reference's own evaluation.py over the shim), restated for the oracle and
the device: z ~ N(0, I_6) [64 chains], x ~ N(z * w, 0.7), x observed."""
import numpy as np

F32 = np.float32
N_CHAINS, D = 64, 6
N_TEMPERATURES, N_ADAPT = 40, 8
HMC_SEED, GLOBAL_SEED = 31, 5
W = np.linspace(0.5, 1.5, D).astype(F32)
X_STD = F32(0.7)
X_OBS = (np.random.RandomState(3).normal(size=D) * 1.2).astype(F32)
HMC_KW = dict(step_size=0.05, n_leapfrogs=5, adapt_step_size=True,
              target_acceptance_rate=0.7)
C0 = F32(-0.5 * np.log(2 * np.pi))


def log_prior(q):
    z = q[0]
    return np.sum(C0 - F32(0.5) * np.square(z), axis=-1, dtype=F32)


def grad_prior(q):
    return [(-q[0]).astype(F32)]


def log_joint(q):
    z = q[0]
    ls = F32(np.log(X_STD))
    prec = F32(np.exp(F32(-2) * ls))
    lik = np.sum(C0 - ls - F32(0.5) * prec * np.square(X_OBS - z * W),
                 axis=-1, dtype=F32)
    return (log_prior(q) + lik).astype(F32)


def grad_joint(q):
    z = q[0]
    prec = F32(np.exp(F32(-2) * F32(np.log(X_STD))))
    return [(-z + prec * (X_OBS - z * W) * W).astype(F32)]
