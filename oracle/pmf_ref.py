"""NumPy float32 restatement of the rating model of the reference's
examples/probabilistic_matrix_factorization/pmf_hmc.py.  TEST INFRASTRUCTURE
(see oracle/__init__.py).

Follows, in /root/reference:
  examples/probabilistic_matrix_factorization/pmf_hmc.py:19-31   the model
      u ~ N(0, alpha_u) [K, n, D], v ~ N(0, alpha_v) [K, m, D] (group_ndims=1),
      r_logits = sum_d gather(u, select_u)[.., d] * gather(v, select_v)[.., d],
      r ~ N(sigmoid(r_logits), alpha_pred)
  pmf_hmc.py:136-143   log_joint = sum_n log p(u) + sum_m log p(v) + sum_e log p(r)
  zhusuan/distributions/univariate.py:174-181   Normal._log_prob
tf.gather / tf.gradients (unsorted segment sum) are TensorFlow's; restated as
NumPy fancy indexing and np.add.at.

Pinning (round 2): the example ships no test, no golden output and its data
set (MovieLens-1M) is not available offline, but its MODEL FUNCTION runs: the
`pmf` of the unmodified pmf_hmc.py, imported by oracle/make_golden_hmc.py and
sampled by the reference's own hmc.py over oracle/tf_shim.py, produced case
`pmf` of tests/golden/hmc_reference_traces.npz, which this restatement
reproduces under oracle/hmc_ref.py (tests/test_oracle_hmc_reference.py[pmf]).
The closed form is also checked in float64 against torch autograd in
tests/test_gpu_gather_dot.py."""
import numpy as np

F32 = np.float32


def gathered_dot(u, su, v, sv):
    return (u[..., su, :] * v[..., sv, :]).sum(-1, dtype=u.dtype)


def gathered_dot_grads(u, su, v, sv, gout):
    gu, gv = np.zeros_like(u), np.zeros_like(v)
    lead = u.shape[:-2]
    for k in np.ndindex(*lead):
        np.add.at(gu[k], su, gout[k][:, None] * v[k][sv])
        np.add.at(gv[k], sv, gout[k][:, None] * u[k][su])
    return gu, gv


def _normal_lp(x, mean, std):
    logstd = np.log(F32(std))
    c = F32(-0.5 * np.log(2 * np.pi))
    prec = np.exp(-2 * logstd)
    return (c - logstd - F32(0.5) * prec * np.square(x - mean)).astype(F32)


def _sigmoid(z):
    return (1 / (1 + np.exp(-z))).astype(F32)


def log_joint(u, v, su, sv, r, alpha_u, alpha_v, alpha_pred):
    """[K]: pmf_hmc.py:136-143."""
    lpu = _normal_lp(u, F32(0), alpha_u).sum(-1, dtype=F32).sum(-1, dtype=F32)
    lpv = _normal_lp(v, F32(0), alpha_v).sum(-1, dtype=F32).sum(-1, dtype=F32)
    p = _sigmoid(gathered_dot(u, su, v, sv))
    lpr = _normal_lp(r, p, alpha_pred).sum(-1, dtype=F32)
    return lpu + lpv + lpr


def grad_log_joint(u, v, su, sv, r, alpha_u, alpha_v, alpha_pred):
    """(d/du, d/dv) of log_joint."""
    p = _sigmoid(gathered_dot(u, su, v, sv))
    g = ((r - p) / F32(alpha_pred) ** 2 * p * (1 - p)).astype(F32)
    gu, gv = gathered_dot_grads(u, su, v, sv, g)
    gu = gu - u / F32(alpha_u) ** 2
    gv = gv - v / F32(alpha_v) ** 2
    return gu.astype(F32), gv.astype(F32)
