"""The fused diagonal-Normal plan (BASELINE configs 0 / 1 / 3): one transition
= ONE launch of csrc/hmc_fused_{ring,normal}.hip (reference
zhusuan/hmc.py:382-522 for a model that is a single Normal node,
examples/toy_examples/gaussian.py:15-20)."""
import ctypes

import torch

from .. import _capi
from ..distributions import Normal
from ..framework.bn import StochasticTensor
from ..framework.meta_bn import MetaBayesianNet
from ..utils import merge_dicts
from .base import _PlanBase, _versions, _prod


class _FusedDiagNormalPlan(_PlanBase):
    """One kernel per transition (csrc/hmc_fused_ring.hip /
    hmc_fused_normal.hip), adaptive or not: the dual-averaging update of
    transition t rides in the prologue of launch t+1 (include/zshmc.h,
    zshmc_adapt_link)."""
    kind = 'fused_diag_normal'
    can_skip_acc = True      # no statistics are collected when stats is NULL
    collect_acc = True

    def __init__(self, hmc, names, values, chain_shape, device, probe):
        super(_FusedDiagNormalPlan, self).__init__(hmc, names, values,
                                                   chain_shape, device)
        self._probe = probe
        self._src = None
        self._cs_rows_cache = {}
        self.workspace = torch.zeros(_capi.LINK_WORKSPACE_BYTES,
                                     dtype=torch.uint8, device=device)
        f32 = dict(dtype=torch.float32, device=device)
        self.mean = torch.zeros(self.n_data[0], **f32)
        self.logstd = torch.zeros(self.n_data[0], **f32)
        self.zero_mean = True
        self.refresh_model()

    def refresh_model(self):
        """Re-resolve the Normal's parameters (the generic plan re-runs the
        model function on every transition; a parameter fed through a
        placeholder -- lntm_mcem.py:164-169 -- or updated in place between
        runs must reach the fused kernel too).  The model function is
        re-evaluated (host only); device copies happen only when a parameter
        tensor is a different object or version than last time."""
        mean_src, spread_src, dist = self._probe()
        src = self._src
        if (src is not None and src[0] is mean_src and src[1] is spread_src
                and src[2] == mean_src._version
                and src[3] == spread_src._version):
            return
        data_shape = tuple(self.q[0].shape[len(self.chain_shape):])
        mean_d = _to_data_shape(dist.mean, data_shape)
        logstd_d = _to_data_shape(dist.logstd, data_shape)
        if mean_d is None or logstd_d is None:
            raise ValueError(
                "HMC (fused diagonal-Normal plan): the parameters of '{}' "
                "now vary along the chain axes; build a new HMC for the "
                "changed model.".format(self.names[0]))
        self.mean.copy_(mean_d)
        self.logstd.copy_(logstd_d)
        # The zero-mean instantiation (no mean tile) is chosen when the mean
        # is verified to be all zeros -- a host read, so only at plan build.
        # A model function that hands over a NEW parameter tensor on a later
        # run (torch.zeros(...) built inside the function, a fed mean) gets
        # the general instantiation from then on: no synchronisation on the
        # per-run path.
        self.zero_mean = not bool(mean_d.any().item()) if src is None \
            else False
        self._src = (mean_src, spread_src, mean_src._version,
                     spread_src._version)

    def _colstats_rows(self):
        """Rows of per-workgroup column sums the launch of the current
        configuration leaves behind (0: this shape's kernel cannot)."""
        key = (self.use_mass, self.zero_mean)
        if key not in self._cs_rows_cache:
            ok = all(t.data_ptr() % 16 == 0 for t in
                     (self.q[0], self.mean, self.logstd, self.mass[0]))
            self._cs_rows_cache[key] = int(
                _capi.load().zshmc_fused_colstats_rows(
                    self.n_chains, self.n_data[0], int(self.use_mass),
                    int(self.zero_mean))) if ok else 0
        return self._cs_rows_cache[key]

    def _link(self, eps_host, collect, retire=None, colstats_rows=0):
        hmc = self.hmc
        k = _capi.AdaptLink()
        if colstats_rows:
            if self.cs_parts is None or \
                    self.cs_parts.shape[0] < colstats_rows:
                self.cs_parts = torch.empty(
                    colstats_rows, 2 * self.n_data[0], dtype=torch.float64,
                    device=self.device)
            k.colstats_mean = self.ewmv_mean[0].data_ptr()
            k.colstats_parts = self.cs_parts.data_ptr()
        # (an in-kernel update needs the state block even when this launch
        # integrates with the step size the search just returned)
        k.state = None if (eps_host is not None and retire is None) \
            else self.state.data_ptr()
        k.stats = self.stats.data_ptr() if collect else None
        k.workspace = self.workspace.data_ptr()
        k.n_chains_global = self.n_chains_global
        k.pending, k.retire_update, k.fresh_start = _capi.PEND_NONE, \
            _capi.PEND_NONE, 0
        k.used_step_size = float('nan')
        k.delta, k.gamma = hmc.target_acceptance_rate, hmc.gamma
        k.t0, k.kappa = hmc.t0, hmc.kappa
        k.mu = 10.0 * hmc._init_step_size_value            # hmc.py:79 (sic)
        if self.pending is not None:
            kind, fresh, used = self.pending
            k.pending, k.fresh_start = kind, int(fresh)
            if used is not None:
                k.used_step_size = float(used)
        if retire is not None:
            kind, fresh, used = retire
            k.retire_update, k.fresh_start = kind, int(fresh)
            if used is not None:
                k.used_step_size = float(used)
        return k

    def _launch(self, t, eps_host, commit, n_leapfrogs, stream, retire=None,
                colstats_rows=0, lib=None):
        # (`lib`: another build of the library, _capi.load_build -- only
        # bench.py's side-by-side timing of the two generators passes one)
        info = commit
        if self.pending is not None and eps_host is not None:
            raise RuntimeError("a pending step-size update must be flushed "
                               "before a launch with a host step size")
        # a launch that carries an update also publishes its sum
        collect = (self.collect_acc or not commit or
                   self.pending is not None or retire is not None)
        link = self._link(eps_host, collect, retire, colstats_rows)
        (_capi.call if lib is None else
         (lambda *a: _capi.call_on(lib, *a)))(
            'zshmc_hmc_diag_normal_step', self.q[0].data_ptr(),
            None if self.zero_mean else self.mean.data_ptr(),
            self.logstd.data_ptr(), self.mass_ptr(0),
            # (with an in-kernel update the kernel must still integrate with
            # the searched step size: the state block then carries it)
            0.0 if eps_host is None else float(eps_host),
            self.n_chains, self.n_data[0], self.chain_offset, n_leapfrogs,
            self.hmc.seed, t & 0xFFFFFFFF, int(commit),
            self.acceptance_rate.data_ptr() if info else None,
            self.orig_hamiltonian.data_ptr() if info else None,
            self.hamiltonian.data_ptr() if info else None,
            self.orig_log_prob.data_ptr() if info else None,
            self.log_prob.data_ptr() if info else None,
            self.flags.data_ptr(), ctypes.byref(link), stream)
        self.pending = None            # retired by this launch
        if collect:
            sh = self.hmc.sharding
            self.stats_local = sh is not None and sh.active

    can_run_block = True

    def run_block(self, t_first, n, kind, stream, sharding):
        """`n` plain transitions (mass fixed, no search) from one call:
        zshmc_hmc_diag_normal_run.  Sharded chains: the C side enqueues the
        all-reduce of [sum acc, flag] between the launches on the same
        communicator; the last transition's update stays pending."""
        sharded = sharding is not None and sharding.active
        update = None if kind == _capi.PEND_NONE else (kind, False, None)
        link = self._link(None, update is not None or self.pending is not None,
                          update)
        if self.pending is not None:
            # (fresh_start / used_step_size describe the FIRST launch's
            # pending update; the run's own updates are never fresh)
            link.fresh_start = int(self.pending[1])
            link.used_step_size = float('nan') if self.pending[2] is None \
                else float(self.pending[2])
        _capi.call(
            'zshmc_hmc_diag_normal_run', self.q[0].data_ptr(),
            None if self.zero_mean else self.mean.data_ptr(),
            self.logstd.data_ptr(), self.mass_ptr(0), 0.0, self.n_chains,
            self.n_data[0], self.chain_offset, self.hmc.n_leapfrogs,
            self.hmc.seed, t_first & 0xFFFFFFFF, n,
            self.acceptance_rate.data_ptr(), self.orig_hamiltonian.data_ptr(),
            self.hamiltonian.data_ptr(), self.orig_log_prob.data_ptr(),
            self.log_prob.data_ptr(), self.flags.data_ptr(),
            ctypes.byref(link), sharding._comm if sharded else None, stream)
        self._own_write()
        self.last_t = t_first + n - 1
        self.pending = update if sharded else None
        self.stats_local = False
        if self.colsum_state in ('fresh', 'parts'):
            self.colsum_state = 'dirty'

    def flush(self, stream, sharding):
        """Retire the pending update from the (already all-reduced)
        acceptance sum: a local one-thread launch, no communication."""
        if self.pending is None:
            return
        link = self._link(None, True)
        _capi.call('zshmc_stepsize_flush', ctypes.byref(link), stream)
        self.pending = None

    def begin_search(self, t, stream):
        pass

    def search_trip(self, t, step_size, stream):
        # one full leapfrog step (hmc.py:316-321) == the kernel with L = 1
        self._launch(t, step_size, 0, 1, stream)

    def _apply_update(self, update, eps_host, stream):
        pass        # carried by the transition kernel / the next prologue

    def transition(self, t, eps_host, stream, update=None,
                   want_colstats=False):
        self.last_t = t
        sh = self.hmc.sharding
        sharded = sh is not None and sh.active
        retire = None if sharded else update
        if retire is not None and eps_host is not None:
            # the searched step size travels through the state block so that
            # the kernel can both use it and update from it
            _capi.call('zshmc_state_set', self.state.data_ptr(),
                       _capi.ST_STEP_SIZE, float(eps_host), stream)
            eps_host = None
        # the column sums of the end state come out of the same launch where
        # the kernel of this shape can produce them
        rows = self._colstats_rows() if want_colstats else 0
        self._launch(t, eps_host, 1, self.hmc.n_leapfrogs, stream, retire, rows)
        self._own_write()
        if rows:
            self._cs_rows = rows
            self.colsum_state = 'parts'
            self._colsum_versions = _versions(self.q)
        if sharded and update is not None:
            # applied by the next launch's prologue (or flush()) once the
            # acceptance sums of all ranks have been added
            self.pending = update


def _to_data_shape(param, data_shape):
    """Flatten a parameter that is constant along the chain axes to a
    contiguous float32 [prod(data_shape)] vector, or None if it is not."""
    t = param.detach().to(torch.float32)
    extra = t.dim() - len(data_shape)
    if extra > 0:
        if any(int(s) != 1 for s in t.shape[:extra]):
            return None
        t = t.reshape(t.shape[extra:])
    try:
        t = t.expand(data_shape)
    except RuntimeError:
        return None
    return t.contiguous().reshape(-1)


def _try_fused_plan(hmc, meta_bn, names, values, chain_shape, device):
    """Recognise the diagonal-Normal family: a MetaBayesianNet with the
    default log-joint whose only stochastic node is the (single) latent, a
    Normal with group_ndims == #data axes and chain-independent parameters."""
    if not isinstance(meta_bn, MetaBayesianNet) or meta_bn.log_joint is not None:
        return None
    if len(names) != 1:
        return None
    name, q = names[0], values[0]
    n_chain_dims = len(chain_shape)
    data_shape = tuple(q.shape[n_chain_dims:])
    n_data = _prod(data_shape)
    if n_data == 0 or n_data > int(_capi.load().zshmc_fused_max_n_data()):
        return None

    def node_dist(value):
        # (as a symbol: a dense-likelihood model written with the reference's
        # literal spelling must not materialise its logits here)
        bn = meta_bn.observe(**merge_dicts(
            {name: hmc._as_symbol(value)}, hmc._resolved_observed()))
        stoch = [n for n in bn.nodes.values()
                 if isinstance(n, StochasticTensor)]
        if len(stoch) != 1 or stoch[0].name != name:
            return None
        dist = stoch[0].dist
        if type(dist) is not Normal or dist.group_ndims != len(data_shape):
            return None
        if dist.use_path_derivative:
            return None
        return dist

    dist = node_dist(q.detach().requires_grad_(True))
    if dist is None:
        return None
    if dist.mean.requires_grad or dist.given_spread[1].requires_grad:
        return None                      # parameters depend on the latent
    if _to_data_shape(dist.mean, data_shape) is None or \
            _to_data_shape(dist.given_spread[1], data_shape) is None:
        return None                      # parameters vary along chain axes

    def probe():
        d = node_dist(q)
        if d is None:
            raise ValueError(
                "HMC (fused diagonal-Normal plan): the model no longer is a "
                "single Normal node '{}'; build a new HMC for the changed "
                "model.".format(name))
        return d.mean, d.given_spread[1], d

    return _FusedDiagNormalPlan(hmc, names, values, chain_shape, device,
                                probe)
