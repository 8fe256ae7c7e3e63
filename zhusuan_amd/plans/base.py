"""What every execution plan of zhusuan_amd.hmc.HMC shares: the sampler state
on the device, the acceptance / column-sum statistics and their one
all-reduce, the mass estimator (reference zhusuan/hmc.py:64-159,284-305,
375-380), and how a plan knows that nobody else wrote its latents."""

import torch

from .. import _capi, _writes


def _versions(tensors):
    """What identifies the CONTENTS of `tensors` between two runs: torch's
    version counter (what an in-place torch op bumps) and the library's own
    write generation of the storage (what every sampler bumps when it writes
    a latent through the C-ABI, zhusuan_amd/_writes.py) -- or None when a
    tensor keeps no version counter (an inference-mode tensor): then nothing
    may be assumed about what happened to it between two runs."""
    out = []
    for t in tensors:
        try:
            out.append((t._version, _writes.generation(t)))
        except RuntimeError:
            return None
    return out


def _prod(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


class _PlanBase(object):
    def __init__(self, hmc, names, values, chain_shape, device):
        self.hmc = hmc
        self.names = list(names)
        self.q = list(values)
        self.chain_shape = chain_shape
        self.n_chains = _prod(chain_shape)
        self.n_data = [_prod(v.shape[len(chain_shape):]) for v in values]
        self.device = device
        sh = hmc.sharding
        if sh is not None:
            self.chain_offset, self.n_chains_global = sh.layout(self.n_chains,
                                                                device)
        else:
            self.chain_offset, self.n_chains_global = 0, self.n_chains
        f32 = dict(dtype=torch.float32, device=device)
        C = self.n_chains
        self.state = torch.zeros(_capi.STATE_WORDS, **f32)
        # Everything that may cross GPUs in one transition sits in ONE buffer
        # so that it is ONE all-reduce: [0] sum of acceptance
        # rates, [1] non-finite-start flag, then per latent the 2*D column
        # sums of the mass estimator (hmc.py:138,143).
        n_col = 2 * sum(self.n_data) if hmc.adapt_mass is not None else 0
        self.comm_buf = torch.zeros(_capi.STATS_WORDS + n_col,
                                    dtype=torch.float64, device=device)
        self.stats = self.comm_buf[:_capi.STATS_WORDS]
        self.acc_sum = self.comm_buf[:1]
        self.stats_local = False      # stats not yet summed over the ranks
        self.pending = None           # (kind, fresh, used step size) owed
        self.flags = torch.zeros(1, dtype=torch.int32, device=device)
        self.acceptance_rate = torch.zeros(C, **f32)
        self.orig_hamiltonian = torch.zeros(C, **f32)
        self.hamiltonian = torch.zeros(C, **f32)
        self.orig_log_prob = torch.zeros(C, **f32)
        self.log_prob = torch.zeros(C, **f32)
        self.use_mass = False
        if hmc.adapt_mass is not None:
            self.mass = [torch.ones(d, **f32) for d in self.n_data]
            self.ewmv_mean = [torch.zeros(d, **f32) for d in self.n_data]
            self.ewmv_var = [torch.zeros(d, **f32) for d in self.n_data]
            self.colsum, off = [], _capi.STATS_WORDS
            for d in self.n_data:
                self.colsum.append(self.comm_buf[off:off + 2 * d])
                off += 2 * d
        self.colsum_state = 'zero'
        self._colsum_versions = []
        self._mass_ones = None        # `use_ones` the mass buffers reflect
        self.cs_parts, self._cs_rows = None, 0
        self.mass_ws = torch.zeros(2, dtype=torch.int32, device=device)
        self.last_t = 0

    def refresh_model(self):
        """Called at the start of every run: the generic plan re-evaluates the
        model function on every gradient anyway."""

    def _own_write(self):
        """This plan has just written its latents (through the C-ABI): other
        samplers on the same tensors must see that (zhusuan_amd/_writes.py),
        while what THIS plan still knows about them -- the carried start
        evaluation, which the writing call itself brought up to date -- stays
        its own."""
        _writes.note(self.q)
        if getattr(self, '_start_valid', False):
            self._start_versions = _versions(self.q)

    # -- mass adaptation (hmc.py:284-305) ------------------------------------
    # colsum life cycle: 'zero' (cleared, what the atomics of
    # zshmc_mass_colstats need), 'fresh' (global column sums of the CURRENT
    # latents around the current EWMV mean, summed over the ranks), 'dirty'.
    def _colstats_fresh(self):
        now = _versions(self.q)
        return self.colsum_state in ('fresh', 'parts') and now is not None \
            and now == self._colsum_versions

    def compute_colstats(self, stream):
        """Local column sums of (q - m), (q - m)^2 of every latent."""
        if self.colsum_state != 'zero':
            _capi.call('zshmc_zero', self.comm_buf.data_ptr() +
                       8 * _capi.STATS_WORDS,
                       8 * (self.comm_buf.numel() - _capi.STATS_WORDS),
                       stream)
        for k, q in enumerate(self.q):
            _capi.call('zshmc_mass_colstats', q.data_ptr(),
                       self.ewmv_mean[k].data_ptr(), self.n_chains,
                       self.n_data[k], self.colsum[k].data_ptr(), stream)
        self._mark_colstats()

    def _mark_colstats(self):
        self.colsum_state = 'fresh'
        self._colsum_versions = _versions(self.q)

    def update_mass(self, update, use_ones, stream, sharding):
        """HMC._adapt_mass (hmc.py:284-305) for every latent.  The column
        sums normally are already there (taken at the end of the previous
        run, all-reduced with its acceptance sum); otherwise they are taken
        now and cross the ranks in an all-reduce of their own."""
        hmc = self.hmc
        if update:
            if not self._colstats_fresh():
                self.compute_colstats(stream)
                if sharding is not None and sharding.active:
                    sharding.all_reduce_sum(
                        self.comm_buf[_capi.STATS_WORDS:])
            self._mass_ones = None
            if len(self.q) == 1:
                # one launch: rows of column sums (the per-workgroup partials
                # a fused transition left behind, or the one reduced row) ->
                # EWMV update -> mass -> tau
                parts, rows = (self.cs_parts, self._cs_rows) \
                    if self.colsum_state == 'parts' else (self.colsum[0], 1)
                _capi.call('zshmc_mass_update_fused', self.state.data_ptr(),
                           self.ewmv_mean[0].data_ptr(),
                           self.ewmv_var[0].data_ptr(), parts.data_ptr(), rows,
                           self.n_chains_global, self.n_data[0],
                           hmc.mass_decay, int(use_ones),
                           self.mass[0].data_ptr(),
                           self.mass_ws.data_ptr(), stream)
                self.colsum_state = 'dirty'
                self._mass_ones = bool(use_ones)
                return
            self.colsum_state = 'zero'       # consumed and cleared below
        elif self._mass_ones == bool(use_ones):
            return          # mass is what it was (hmc.py:158-159, :299-302)
        self._mass_ones = None if update else bool(use_ones)
        for k in range(len(self.q)):
            # EWMV.t is shared by all latents (hmc.py:118,131): bump once,
            # after the last latent
            last = k == len(self.q) - 1
            _capi.call('zshmc_mass_update', self.state.data_ptr(),
                       self.ewmv_mean[k].data_ptr(),
                       self.ewmv_var[k].data_ptr(),
                       self.colsum[k].data_ptr(), self.n_chains_global,
                       self.n_data[k], hmc.mass_decay,
                       (1 if last else 2) if update else 0,
                       int(use_ones), self.mass[k].data_ptr(), stream)

    def reduce_stats(self, sharding, stream):
        """Sum the acceptance statistic over the ranks if that is still owed
        (the trips of the step-size search; a transition's own statistics
        travel in `finish`)."""
        if self.stats_local:
            if sharding is not None and sharding.active:
                sharding.all_reduce_sum(self.stats)
            self.stats_local = False

    def finish(self, update, eps_host, want_colstats, stream, sharding):
        """End of a run: the column sums of the end state (next run's mass
        update), ONE all-reduce of [sum acc, flag, colsum...], then the
        step-size update of this transition (hmc.py:501-505)."""
        sharded = sharding is not None and sharding.active
        if want_colstats:
            if not self._colstats_fresh():
                self.compute_colstats(stream)
        elif self.colsum_state in ('fresh', 'parts'):
            self.colsum_state = 'dirty'      # q moved on, sums did not
        if sharded:
            if want_colstats and self.colsum_state == 'parts':
                # the partials of this rank -> the row that crosses the ranks
                _capi.call('zshmc_mass_colstats_reduce',
                           self.cs_parts.data_ptr(), self._cs_rows,
                           self.n_data[0], self.colsum[0].data_ptr(), stream)
                self.colsum_state = 'fresh'
            if want_colstats:
                sharding.all_reduce_sum(self.comm_buf)
            elif update is not None:
                sharding.all_reduce_sum(self.stats)
            self.stats_local = False
        if update is not None:
            self._apply_update(update, eps_host, stream)

    def _apply_update(self, update, eps_host, stream):
        """hmc.py:501-505 as its own launch (acc_sum filled by atomics,
        already summed over the ranks)."""
        hmc = self.hmc
        kind, init, _ = update
        _capi.call('zshmc_stepsize_update', self.state.data_ptr(),
                   self.acc_sum.data_ptr(), self.n_chains_global,
                   int(kind == _capi.PEND_ADAPT), int(init),
                   hmc.target_acceptance_rate, hmc.gamma, hmc.t0, hmc.kappa,
                   10.0 * hmc._init_step_size_value, stream)
        if eps_host is not None:
            _capi.call('zshmc_state_set', self.state.data_ptr(),
                       _capi.ST_USED_STEP_SIZE, float(eps_host), stream)

    def flush(self, stream, sharding):
        pass

    def end_search_trip(self):
        pass

    def mass_ptr(self, k):
        return self.mass[k].data_ptr() if self.use_mass else None

    def regenerate_momentum(self, name):
        k = self.names.index(name)
        p = torch.empty_like(self.q[k])
        _capi.call('zshmc_momentum', p.data_ptr(), self.mass_ptr(k),
                   self.n_chains, self.n_data[k], self.chain_offset,
                   self.hmc.seed, self.last_t & 0xFFFFFFFF, k, None,
                   _capi.current_stream())
        return p


class _Unsupported(ValueError):
    """The model is outside what a native plan handles: the caller falls back
    to the generic plan."""
