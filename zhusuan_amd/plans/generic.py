"""The generic plan: any log-joint (callable or MetaBayesianNet); torch autograd
over the HIP log_prob ops plays tf.gradients (reference zhusuan/hmc.py:430-432),
momentum / kick + drift / MH / select are the kernels of csrc/hmc_generic.hip."""

import torch

from .. import _capi
from .base import _PlanBase


class _GenericPlan(_PlanBase):
    """Arbitrary log-joint: autograd supplies the gradient (tf.gradients,
    hmc.py:430-432); everything else runs in csrc/hmc_generic.hip."""
    kind = 'generic'

    def __init__(self, hmc, names, values, chain_shape, device):
        super(_GenericPlan, self).__init__(hmc, names, values, chain_shape,
                                           device)
        f32 = dict(dtype=torch.float32, device=device)
        C = self.n_chains
        self.p = [torch.empty_like(q) for q in self.q]
        self.q_new = [torch.empty_like(q) for q in self.q]
        self.kin_old = torch.zeros(C, **f32)
        self.kin_new = torch.zeros(C, **f32)
        self.accept = torch.zeros(C, dtype=torch.uint8, device=device)
        self._search_cache = None
        self._in_search = False

    def value_and_grad(self, qs):
        """log p(q) per chain and d/dq (hmc.py:426-432)."""
        leaves = [q.detach().requires_grad_(True) for q in qs]
        lp = self.hmc._eval_log_joint(self.names, leaves)
        if tuple(lp.shape) != tuple(self.chain_shape):
            raise ValueError(
                "log joint returned shape {} but the chain shape is {}"
                .format(tuple(lp.shape), tuple(self.chain_shape)))
        grads = torch.autograd.grad(lp.sum(), leaves, allow_unused=True)
        grads = [torch.zeros_like(q) if g is None else g.contiguous()
                 for g, q in zip(grads, leaves)]
        return lp.detach().reshape(-1).to(torch.float32).contiguous(), grads

    def _momentum(self, t, stream):
        self.kin_old.zero_()
        for k, p in enumerate(self.p):
            _capi.call('zshmc_momentum', p.data_ptr(), self.mass_ptr(k),
                       self.n_chains, self.n_data[k], self.chain_offset,
                       self.hmc.seed, t & 0xFFFFFFFF, k,
                       self.kin_old.data_ptr(), stream)

    def _kick_drift(self, qs, ps, grads, eps_host, kick, drift, kinetic,
                    stream):
        for k in range(len(qs)):
            _capi.call('zshmc_kick_drift', qs[k].data_ptr(), ps[k].data_ptr(),
                       grads[k].data_ptr(), self.mass_ptr(k),
                       None if eps_host is not None else self.state.data_ptr(),
                       0.0 if eps_host is None else float(eps_host),
                       float(kick), float(drift), self.n_chains,
                       self.n_data[k],
                       None if kinetic is None else kinetic.data_ptr(), stream)

    def begin_search(self, t, stream):
        self._momentum(t, stream)
        lp0, g0 = self.value_and_grad(self.q)
        self._search_cache = (lp0, g0)

    def reduce_stats(self, sharding, stream):
        """Only the step-size search asks (the transition's own sum is
        reduced and consumed by stepsize_update): acceptance sum and the
        non-finite flag of the last dry run, summed over ranks."""
        if not self._in_search:
            return
        self.stats[1] = (self.flags != 0).to(torch.float64)[0]
        if sharding is not None and sharding.active:
            sharding.all_reduce_sum(self.stats)

    def end_search_trip(self):
        self.stats.zero_()
        self._in_search = False

    def search_trip(self, t, step_size, stream):
        self._in_search = True
        lp0, g0 = self._search_cache
        q1 = [q.clone() for q in self.q]
        p1 = [p.clone() for p in self.p]
        self._kick_drift(q1, p1, g0, step_size, 0.5, 1.0, None, stream)
        lp1, g1 = self.value_and_grad(q1)
        self.kin_new.zero_()
        self._kick_drift(q1, p1, g1, step_size, 0.5, 0.0, self.kin_new, stream)
        _capi.call('zshmc_mh_accept', lp0.data_ptr(), lp1.data_ptr(),
                   self.kin_old.data_ptr(), self.kin_new.data_ptr(),
                   self.n_chains, self.chain_offset, self.hmc.seed,
                   t & 0xFFFFFFFF, None, None, None, None, None,
                   self.acc_sum.data_ptr(), self.flags.data_ptr(), stream)

    def transition(self, t, eps_host, stream, update=None,
                   want_colstats=False):
        self._transition(t, eps_host, stream)

    def _transition(self, t, eps_host, stream):
        self.last_t = t
        L = self.hmc.n_leapfrogs
        if self._search_cache is not None:
            lp_old, g = self._search_cache     # same q, same p0 (Appendix B 11)
            self._search_cache = None
        else:
            self._momentum(t, stream)
            lp_old, g = self.value_and_grad(self.q)
        for qn, q in zip(self.q_new, self.q):
            qn.copy_(q)
        p = self.p
        lp_new = lp_old
        self.kin_new.zero_()
        # i = 0: zero-length drift, half kick (hmc.py:352-364); the drift of
        # trip i+1 is fused behind the kick of trip i
        self._kick_drift(self.q_new, p, g, eps_host, 0.5,
                         1.0 if L >= 1 else 0.0,
                         self.kin_new if L == 0 else None, stream)
        for i in range(1, L + 1):
            lp_new, g = self.value_and_grad(self.q_new)
            last = i == L
            self._kick_drift(self.q_new, p, g, eps_host,
                             0.5 if last else 1.0, 0.0 if last else 1.0,
                             self.kin_new if last else None, stream)
        _capi.call('zshmc_mh_accept', lp_old.data_ptr(), lp_new.data_ptr(),
                   self.kin_old.data_ptr(), self.kin_new.data_ptr(),
                   self.n_chains, self.chain_offset, self.hmc.seed,
                   t & 0xFFFFFFFF, self.acceptance_rate.data_ptr(),
                   self.orig_hamiltonian.data_ptr(),
                   self.hamiltonian.data_ptr(), self.log_prob.data_ptr(),
                   self.accept.data_ptr(), self.acc_sum.data_ptr(),
                   self.flags.data_ptr(), stream)
        self.orig_log_prob.copy_(lp_old)
        for k in range(len(self.q)):
            _capi.call('zshmc_select_rows', self.q[k].data_ptr(),
                       self.q_new[k].data_ptr(), self.accept.data_ptr(),
                       self.n_chains, self.n_data[k], stream)
        self._own_write()
