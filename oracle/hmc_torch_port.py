"""torch-CPU restatement of ONE HMC transition for a diagonal-Normal joint,
op for op in the pass structure of the reference's TensorFlow graph
(/root/reference/zhusuan/hmc.py:21-61, :348-372, :479-498 over
distributions/univariate.py:174-181): every line below is one full-array
pass, executed by torch's intra-op thread pool -- the closest thing to
"TF-CPU with intra_op_parallelism_threads = nproc" that can run where
TensorFlow cannot be installed.  TEST / BASELINE INFRASTRUCTURE (see
oracle/__init__.py): bench.py times it on the GPU box's host cores;
tests/test_oracle_torch_port.py holds it to oracle/hmc_ref.py.
"""
import math

import torch

C0 = -0.5 * math.log(2 * math.pi)


def log_prob(x, mean, logstd):
    """Normal._log_prob (univariate.py:174-181) + group_ndims = 1 reduce_sum
    (base.py:302-304)."""
    precision = torch.exp(-2 * logstd)
    return torch.sum(C0 - logstd - 0.5 * precision * torch.square(x - mean),
                     dim=-1)


def grad_log_prob(x, mean, logstd):
    """What tf.gradients returns for the above (hmc.py:430-432)."""
    return -torch.exp(-2 * logstd) * (x - mean)


def transition(q, mean, logstd, step_size, n_leapfrogs, z, u):
    """q [C, D] updated in place; z [C, D] standard normals (random_momentum,
    :21-23, mass = 1), u [C] uniforms (:485).  Returns (acceptance_rate,
    old_hamiltonian, new_hamiltonian, old_log_prob, log_prob)."""
    p = z * 1.0                                         # :22 (sqrt(mass) = 1)
    cq, cp = q.clone(), p
    for i in range(n_leapfrogs + 1):                    # :348-372
        s1 = step_size if i > 0 else 0.0
        s2 = step_size if 0 < i < n_leapfrogs else step_size / 2
        cq = cq + s1 * (cp / 1.0)                       # :39, :26-27
        cp = cp + s2 * grad_log_prob(cq, mean, logstd)  # :40-42
    old_lp = log_prob(q, mean, logstd)                  # :30-35, :46-61
    new_lp = log_prob(cq, mean, logstd)
    old_h = -old_lp + 0.5 * torch.sum(torch.square(p) / 1.0, dim=-1)
    new_h = -new_lp + 0.5 * torch.sum(torch.square(cp) / 1.0, dim=-1)
    acc = torch.exp(torch.minimum(old_h - new_h, torch.zeros_like(old_h)))
    ok = torch.logical_and(torch.isfinite(acc), torch.isfinite(new_lp))
    acc = torch.where(ok, acc, torch.zeros_like(acc))
    accept = u < acc                                    # :486 (strict)
    q.copy_(torch.where(accept[:, None], cq, q))        # :488-497
    return acc, old_h, new_h, old_lp, torch.where(accept, new_lp, old_lp)
