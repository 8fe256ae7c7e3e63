"""The native model plans: Normal-prior latents + one observed node whose
log-likelihood and gradient come from a fused kernel (reference
zhusuan/hmc.py:348-372, :418-520).  This module is what the four families
share -- the packed state, the transition, the carried start evaluation, the
choice of arithmetic, the C-side block runs (csrc/hmc_model*.hip); the
families themselves (dense-logit Bernoulli / Categorical, the mixture
multinomial of lntm_mcem.py, the gathered-dot rating model of pmf_hmc.py)
are subclasses in zhusuan_amd/plans/families.py."""
import ctypes

import torch

from .. import _capi, _writes
from .base import _PlanBase, _versions, _Unsupported


class _DenseLikelihoodPlan(_PlanBase):
    """Native plan for the dense-likelihood families (BASELINE configs 3 / 5):
    latents with Normal priors and one observed node whose log-likelihood and
    gradient come from a fused MFMA (or gather) kernel.  The base class of
    families.py's _LinearBernoulliPlan, _MixtureMultinomialPlan,
    _LinearCategoricalPlan and _GatheredDotPlan; `_DenseLikelihoodPlan(...,
    kind)` builds the family named `kind`.

    The plan works on a PACKED state: the latents' columns side by side in
    rows of `ld` floats (the total rounded up to a multiple of 4; the columns
    behind the last latent stay zero) -- what the likelihood kernel takes as
    its W operand once the design matrices are laid out the same way.  A
    single latent whose size is a multiple of 4 is its own packed state.

    A transition is momentum + (L+1) x [likelihood kernel, one element-wise
    launch doing prior gradient / softmax Jacobian / kick / drift / next
    operand] + MH + select: no autograd graph and no ATen kernel on the path
    (csrc/hmc_model.hip).  The model function is still re-evaluated on the
    host at the start of every run, so fed placeholders (mini-batches,
    eta_mean / eta_logstd of lntm_mcem.py:164-169) and in-place parameter
    updates are seen."""
    can_skip_acc = False

    # what a family declares (zhusuan_amd/plans/families.py)
    kind = None            # the name HMC.plan_kind reports
    softmax = False        # the operand is softmax(q) (Jacobian in the step)
    segmented = False      # class rows / table rows: csrc/hmc_model_seg.hip
    takes_bf16x3 = False   # csrc/b3_kernel.h has this likelihood
    one_launch_capable = False   # csrc/hmc_model_traj.hip has it

    def __new__(cls, hmc, names, values, chain_shape, device, probe,
                kind=None):
        # `_DenseLikelihoodPlan(..., kind)` builds the family's plan
        if cls is _DenseLikelihoodPlan:
            from .families import FAMILIES
            cls = FAMILIES[kind]
        return super(_DenseLikelihoodPlan, cls).__new__(cls)

    def __init__(self, hmc, names, values, chain_shape, device, probe,
                 kind=None):
        super(_DenseLikelihoodPlan, self).__init__(hmc, names, values,
                                                   chain_shape, device)
        from .. import _ops
        self._ops = _ops
        assert kind in (None, self.kind), (kind, self.kind)
        kind = self.kind
        self._probe = probe
        f32 = dict(dtype=torch.float32, device=device)
        C = self.n_chains
        self.offsets = [sum(self.n_data[:k]) for k in range(len(self.n_data))]
        D = self.n_total = sum(self.n_data)
        self.ld = ld = (D + 3) // 4 * 4
        self.packed = len(self.q) > 1 or ld != D
        # chain axes flattened: [C, D_k] views of the latents
        self.q_rows = [q.view(C, d) for q, d in zip(self.q, self.n_data)]
        self.p = torch.zeros(C, ld, **f32)
        self.q_new = torch.zeros(C, ld, **f32)
        # kernel width, chain block, likelihood rows, family buffers
        need_operand = self._layout(C, D, ld, f32)
        self.grad = torch.empty(self.lik_rows, self.width, **f32)
        # operand of the likelihood kernel: theta = softmax(q) / zero-padded q
        # / the class rows of q (padding rows and columns stay zero)
        self.operand = torch.zeros(self.lik_rows, self.width, **f32) \
            if need_operand else None
        if hmc.adapt_mass is not None:
            # the latents' mass vectors are the columns of ONE packed vector
            # (what the step kernel reads); the padding keeps mass 1
            self.mass_pack = torch.ones(ld, **f32)
            self.mass = [self.mass_pack[o:o + d]
                         for o, d in zip(self.offsets, self.n_data)]
        self.ll = torch.empty(self.lik_rows, **f32)
        # The likelihood evaluation AT THE STATE THE LATENTS HOLD: a transition
        # starts from the previous one's last evaluation where the chain
        # accepted, from its own first one where it did not (the accepted
        # chains' rows of grad / ll are copied over behind the MH test), so a
        # transition is L likelihood launches, not L + 1 -- as long as nobody
        # else wrote the latents and the model's tensors are the same
        # (`_start_is_valid`).  The likelihood term is carried UNSCALED
        # (lik_scale is applied by the step), so annealing keeps it.
        self.grad0 = torch.empty(self.lik_rows, self.width, **f32)
        self.ll0 = torch.empty(self.lik_rows, **f32)
        self.carry_start = hmc.reuse_start_evaluation
        self._start_valid = False
        self._start_versions = []
        self.lp_old = self.orig_log_prob      # HMCInfo.orig_log_prob itself
        self.lp_new = torch.empty(C, **f32)
        self.kin_old = torch.zeros(C, **f32)
        self.kin_new = torch.zeros(C, **f32)
        self.accept = torch.zeros(C, dtype=torch.uint8, device=device)
        self._in_search = False
        self._src = None
        self._ws = None
        # the one-launch trajectory: grid barrier words [arrivals, generation,
        # fault, -]; how many workgroups of this kernel fit the device at once
        # (False: the launch loop below, from Python -- what a transition
        # behind a step-size search runs, and what the tests count calls of)
        self.c_transition = True
        self.traj_sync = torch.zeros(4, dtype=torch.int32, device=device)
        self.traj_capacity = 0
        if hmc.one_launch_trajectory and self.one_launch_capable and \
                self.width <= 256:
            cap = ctypes.c_int(0)
            _capi.call('zshmc_trajectory_capacity', self.width,
                       _capi.PLAN_KINDS[kind], ctypes.addressof(cap))
            self.traj_capacity = int(cap.value)
        # multiplies the likelihood term (log-density and gradient): 1 for the
        # joint; AIS installs its temperature (evaluation.py:101-103)
        self.lik_scale = lambda: 1.0
        self.refresh_model()

    # -- model tensors -------------------------------------------------------
    def refresh_model(self):
        priors, inner, obs = self._probe()
        # priors: [(mean, ('std' | 'logstd', tensor as given))] per latent
        t = [m for m, _ in priors] + [sp[1] for _, sp in priors] + \
            [a for a in _flat_tensors(inner)] + [obs]
        # same storage, layout and version counter as last run (the tensors
        # are held, so an address cannot have been handed to another one;
        # `X.t()` of the literal spelling is a new view object every time)
        # -- and no sampler has moved it as ITS latent in between: those
        # writes go through the C-ABI, past torch's counter (_writes)
        key = [(a.data_ptr(), tuple(a.shape), tuple(a.stride()), a.dtype,
                a._version, _writes.generation(a)) for a in t]
        if not self.carry_start:
            # reuse_start_evaluation=False: nothing about the model's tensors
            # is remembered from one run to the next (hmc.py:47-50) -- about
            # THIS model's: the other samplers' cached operands stay
            self._ops.forget(t)
            self._src = None
        if self._src is not None and key == self._src[0]:
            return
        # another likelihood (design matrix, observations): another
        # evaluation at the start.  New PRIOR tensors alone -- a model function
        # that builds `torch.zeros(d)` per call -- leave the likelihood term's
        # carried evaluation valid: the step recomputes the prior.
        n_prior = 2 * len(priors)
        if self._src is None or key[n_prior:] != self._src[0][n_prior:]:
            self._start_valid = False
        self._src = (key, t)
        C = self.n_chains
        self._pack_prior(priors)
        ops = self._ops
        n_inner = self._operands(inner, obs)
        if n_inner is None:         # (the family keeps no dense operand)
            return
        # the bf16x3 kernels ('bf16x3': wherever they exist -- <= 256 columns
        # and, one document per 128-chain workgroup, chain axes that fill
        # those workgroups -- with a warning where they do not; 'auto', the
        # default: there, when the evaluation is large enough to be bound by
        # the matrix cores rather than by its critical path)
        self.inner_image = None
        self.arithmetic_reason = None
        self.packed_rows = False
        arith = self.hmc.likelihood_arithmetic
        if arith in ('bf16x3', 'auto') and self.takes_bf16x3:
            why = None
            per_doc = self._chains_per_counts_row()
            n_docs = C // per_doc
            if self.width not in ops.BF16X3_WIDTHS:
                why = ('the bf16x3 kernels take <= %d padded columns, this '
                       'likelihood has %d' % (max(ops.BF16X3_WIDTHS),
                                              self.width))
            elif not (n_docs == 1 or not ops.BF16X3_REQUIRE_FILL or
                      per_doc % ops.BF16X3_CHAIN_BLOCK == 0 or
                      per_doc >= 8 * ops.BF16X3_CHAIN_BLOCK) and not (
                          self.width <= ops.BF16X3_PACKED_MAX_WIDTH and
                          self.obs.numel() < (1 << 30)):
                # (a few chains x many documents run on the packed-rows form
                # of the kernel -- 128 consecutive rows per workgroup, each
                # with its own counts row -- up to 192 topics)
                why = ('%d chains per document do not fill the %d-chain '
                       'workgroups of the bf16x3 multinomial kernel, and its '
                       'packed-rows form takes <= %d topics and counts below '
                       '4 GB' % (per_doc, ops.BF16X3_CHAIN_BLOCK,
                                 ops.BF16X3_PACKED_MAX_WIDTH))
            elif arith == 'auto':
                # (of ALL ranks' chains: every shard of one problem runs the
                # same arithmetic, whatever the number of ranks)
                rows = self.lik_rows * (self.n_chains_global /
                                        max(self.n_chains, 1))
                flop = 4.0 * n_inner * self.width * rows
                if flop < ops.BF16X3_AUTO_MIN_FLOP:
                    why = ('%.2g flop per evaluation: latency-bound, the '
                           'finer-grained fp32 kernels' % flop)
                else:
                    # (a family may know an exact-fp32 kernel that beats the
                    # bf16 matrix cores on this problem)
                    why = self._auto_prefers_fp32(n_inner, per_doc, n_docs)
            if why is None:
                self.inner_image = ops.bf16x3_image(self.inner)
                self.block = ops.BF16X3_CHAIN_BLOCK
                # (the library's rule: zshmc_bf16x3_multinomial_rows_packed)
                self.packed_rows = bool(
                    n_docs > 1 and
                    self.width <= ops.BF16X3_PACKED_MAX_WIDTH and
                    self.obs.numel() < (1 << 30) and
                    _capi.load().zshmc_bf16x3_multinomial_rows_packed(
                        n_docs, per_doc))
            else:
                self.arithmetic_reason = why
                if arith == 'bf16x3':
                    import warnings
                    from ..hmc import LikelihoodArithmeticWarning
                    warnings.warn(
                        "HMC(likelihood_arithmetic='bf16x3'): this model's "
                        "likelihood runs on the fp32 kernels -- " + why,
                        LikelihoodArithmeticWarning, stacklevel=2)
        if self.inner_image is None:
            self.block = self._fp32_block()
        # (a family may shorten the inner range it really runs over: the topic
        # model's documents' own vocabularies)
        n_inner = self._inner_rows_run(n_inner)
        R = self.lik_rows
        # (packed rows: three tile buffers + 48 KB of counts, one per CU)
        per_cu = 1 if self.packed_rows else self._resident_per_cu()
        self.splits = self._choose_splits(R, n_inner, per_cu)
        # (chain blocks x slices resident at once where the chain blocks
        # alone are: the trips then run from one cooperative launch)
        n_wg = (R + self.block - 1) // self.block
        if self.inner_image is None and 0 < n_wg <= self.traj_capacity:
            self.splits = max(1, min(self.splits, self.traj_capacity // n_wg))
        need = self.splits * R * (self.width + 1) if self.splits > 1 else 0
        if need and (self._ws is None or self._ws.numel() < need):
            self._ws = torch.empty(need, dtype=torch.float32,
                                   device=self.device)

    # -- what a family implements (zhusuan_amd/plans/families.py) -------------
    def _layout(self, C, D, ld, f32):
        """Set width, block, lik_rows (+ the family's buffers) for C chains
        of D packed floats in rows of ld; return whether the likelihood
        kernel needs an operand other than the packed state itself."""
        raise NotImplementedError

    def _operands(self, inner, obs):
        """The model's tensors -> self.inner / self.obs in the kernel's
        layout; returns the number of inner (data / vocabulary) rows, or None
        when the family keeps no dense operand."""
        raise NotImplementedError

    def _evaluate(self, w, q, grad, ll, ll_ptr, ws, stream):
        """One likelihood + gradient launch at operand w (state q)."""
        raise NotImplementedError

    def _describe(self, d):
        """The family's fields of zshmc_model_plan."""
        d.inner, d.n_inner = self.inner.data_ptr(), self.inner.shape[0]
        d.inner_image = _capi.ptr(self.inner_image)
        d.obs = self.obs.data_ptr()

    def _chains_per_counts_row(self):
        """Chains that share one row of the observed operand (all of them,
        except in the topic model: one counts row per document)."""
        return self.n_chains

    def _fp32_block(self):
        return self._ops.likelihood_plan(self.width)[1]

    def _inner_rows_run(self, n_inner):
        """Inner rows one evaluation runs over (called once the arithmetic is
        chosen): all of them, unless the family knows better."""
        return n_inner

    def _auto_prefers_fp32(self, n_inner, per_doc, n_docs):
        """likelihood_arithmetic='auto', a bf16x3 kernel exists and the
        problem is large enough for it: a reason to stay on the fp32 path
        anyway, or None."""
        return None

    def _choose_splits(self, R, n_inner, per_cu):
        """Slices of the inner range per likelihood launch (their partials
        are added by the step / a reduction launch)."""
        return self._ops._row_splits(R, n_inner, self.device, self.block,
                                     per_cu)

    def _resident_per_cu(self):
        return self._ops.resident_per_cu(
            self.width, 'bf16x3' if self.inner_image is not None else 'fp32')

    def _pack_prior(self, priors):
        """Prior mean / log-std as [rows, ld] matrices used with row period
        `rows` over the flattened chain axes (_to_row_period); several
        latents: their columns side by side, a common row period."""
        parts = []
        for (mean, (how, spread)), d, q in zip(priors, self.n_data, self.q):
            logstd = torch.log(spread) if how == 'std' else spread  # :96-103
            if q.dim() == len(self.chain_shape):    # per-chain scalar latent
                mean, logstd = mean.unsqueeze(-1), logstd.unsqueeze(-1)
            elif q.dim() > len(self.chain_shape) + 1:   # [K, F] class rows
                ds = tuple(q.shape[len(self.chain_shape):])
                mean = _flatten_data_axes(mean, ds)
                logstd = _flatten_data_axes(logstd, ds)
            try:
                parts.append((_to_row_period(mean, self.chain_shape, d),
                              _to_row_period(logstd, self.chain_shape, d)))
            except (RuntimeError, ValueError) as e:
                raise _Unsupported(str(e))
        if not self.packed:
            (self.prior_mean, self.mean_rows), \
                (self.prior_logstd, self.logstd_rows) = parts[0]
            return
        out = []
        for which in (0, 1):
            rows = max(p[which][1] for p in parts)
            m = torch.zeros(rows, self.ld, dtype=torch.float32,
                            device=self.device)
            for p, o, d in zip(parts, self.offsets, self.n_data):
                t, r = p[which]
                if r not in (1, rows):
                    raise _Unsupported(
                        "HMC (native %s plan): the priors' parameters vary "
                        "along different chain axes" % self.kind)
                m[:, o:o + d] = t
            out.append((m, rows))
        (self.prior_mean, self.mean_rows), \
            (self.prior_logstd, self.logstd_rows) = out

    # -- building blocks -----------------------------------------------------
    def _load_state(self, stream):
        """The latents -> the packed working state q_new."""
        if not self.packed:
            self.q_new.copy_(self.q_rows[0])
            return
        for k, qk in enumerate(self.q_rows):
            _capi.call('zshmc_copy_rows',
                       self.q_new.data_ptr() + 4 * self.offsets[k], self.ld,
                       qk.data_ptr(), self.n_data[k], None, self.n_chains,
                       self.n_data[k], stream)

    def _store_state(self, stream):
        """where(accept, q_new, q) for every latent (hmc.py:488-497)."""
        for k, qk in enumerate(self.q_rows):
            _capi.call('zshmc_copy_rows', qk.data_ptr(), self.n_data[k],
                       self.q_new.data_ptr() + 4 * self.offsets[k], self.ld,
                       self.accept.data_ptr(), self.n_chains, self.n_data[k],
                       stream)

    def _likelihood(self, q, stream, want_ll=True, start=False):
        """ll[c] and d ll / d operand at the operand derived from q.
        `want_ll=False`: the gradient alone -- the interior evaluations of a
        trajectory (hmc.py:348-372 reads the log-joint at its two ends only);
        the MFMA kernels then skip the log-likelihood terms.  `start`: into
        the start buffers (grad0 / ll0) instead of the trajectory's."""
        grad, ll = (self.grad0, self.ll0) if start else (self.grad, self.ll)
        ll_ptr = ll.data_ptr() if want_ll else None
        w = self.operand if self.operand is not None else q
        ws = self._ws if self.splits > 1 else None
        self._evaluate(w, q, grad, ll, ll_ptr, ws, stream)

    def _step(self, q, p, use_grad, eps_host, kick, drift, lp_out, kinetic,
              stream, start=False):
        """csrc/hmc_model.hip: prior + Jacobian + kick + drift + operand.
        `start`: the evaluation it reads is the start buffers'."""
        grad, ll = (self.grad0, self.ll0) if start else (self.grad, self.ll)
        if self.segmented:
            # csrc/hmc_model_seg.hip: the class rows of a chain are rows
            # c * stride + k of the gradient / operand matrices
            _capi.call(
                'zshmc_model_kick_drift_seg', q.data_ptr(), p.data_ptr(),
                grad.data_ptr() if use_grad else None, self.width,
                self.seg_len, self.stride, _capi.ptr(self.operand),
                self.width, self.prior_mean.data_ptr(), self.mean_rows,
                self.prior_logstd.data_ptr(), self.logstd_rows,
                self.mass_pack.data_ptr() if self.use_mass else None,
                None if eps_host is not None else self.state.data_ptr(),
                0.0 if eps_host is None else float(eps_host), float(kick),
                float(drift), float(self.lik_scale()), self.n_chains,
                self.n_total, self.ld,
                ll.data_ptr() if use_grad else None, _capi.ptr(lp_out),
                _capi.ptr(kinetic), self.seg_ws.data_ptr(), stream)
            return
        _capi.call(
            'zshmc_model_kick_drift', q.data_ptr(), p.data_ptr(),
            grad.data_ptr() if use_grad else None, self.width,
            _capi.ptr(self.operand), self.width, int(self.softmax),
            self.prior_mean.data_ptr(), self.mean_rows,
            self.prior_logstd.data_ptr(), self.logstd_rows,
            self.mass_pack.data_ptr() if self.use_mass else None,
            None if eps_host is not None else self.state.data_ptr(),
            0.0 if eps_host is None else float(eps_host), float(kick),
            float(drift), float(self.lik_scale()), self.n_chains,
            self.n_total, self.ld,
            ll.data_ptr() if use_grad else None, _capi.ptr(lp_out),
            _capi.ptr(kinetic), stream)

    def _momentum(self, t, stream):
        _capi.call('zshmc_zero', self.kin_old.data_ptr(),
                   4 * self.n_chains, stream)
        # per latent, with the latent's own counters (the generic plan's and
        # regenerate_momentum's: Philox stream word = latent index)
        for k, d in enumerate(self.n_data):
            _capi.call('zshmc_momentum_rows',
                       self.p.data_ptr() + 4 * self.offsets[k], self.ld,
                       self.mass_ptr(k), self.n_chains, d, self.chain_offset,
                       self.hmc.seed, t & 0xFFFFFFFF, k,
                       self.kin_old.data_ptr(), stream)

    def _start_is_valid(self):
        """grad0 / ll0 hold the likelihood evaluation at the latents as they
        are: left there by the last transition (or evaluation), the model's
        tensors unchanged since (refresh_model), nobody else having written a
        latent (our own writes go through the C-ABI and leave torch's version
        counters alone)."""
        now = _versions(self.q)
        return self.carry_start and self._start_valid and now is not None \
            and now == self._start_versions

    def _mark_start(self):
        self._start_valid = True
        self._start_versions = _versions(self.q)

    def _first_evaluation(self, q, stream):
        """operand(q), then likelihood + gradient at q (ll0, grad0) -- unless
        they are there already."""
        if self._start_is_valid():
            # (softmax: the step's Jacobian reads theta = softmax(q) from the
            # operand buffer, which holds the last PROPOSAL's)
            if self.softmax:
                self._step(q, self.p, False, 0.0, 0.0, 0.0, None, None, stream)
            return
        if self.operand is not None:
            self._step(q, self.p, False, 0.0, 0.0, 0.0, None, None, stream)
        self._likelihood(q, stream, start=True)
        self._mark_start()

    def _carry_start(self, stream):
        """Behind the MH test and the select: the accepted chains' last
        evaluation becomes the evaluation at their (new) state."""
        if self.hmc.n_leapfrogs < 1:
            return
        n = self.lik_rows // self.n_chains * self.width
        _capi.call('zshmc_copy_rows', self.grad0.data_ptr(), n,
                   self.grad.data_ptr(), n, self.accept.data_ptr(),
                   self.n_chains, n, stream)
        g = self.lik_rows // self.n_chains
        _capi.call('zshmc_copy_rows', self.ll0.data_ptr(), g,
                   self.ll.data_ptr(), g, self.accept.data_ptr(),
                   self.n_chains, g, stream)

    # -- step-size search (hmc.py:308-345) -----------------------------------
    def reduce_stats(self, sharding, stream):
        if not self._in_search:
            return
        self.stats[1] = (self.flags != 0).to(torch.float64)[0]
        if sharding is not None and sharding.active:
            sharding.all_reduce_sum(self.stats)

    def end_search_trip(self):
        self.stats.zero_()
        self._in_search = False

    def begin_search(self, t, stream):
        self._momentum(t, stream)
        self._load_state(stream)
        self._first_evaluation(self.q_new, stream)

    def _restore_start(self, t, stream):
        """(q, p0) of the start point: q from the latent, p0 regenerated from
        its Philox counters (cheaper in memory than a copy: config 5 holds
        21 GB per [rows, K] buffer); its evaluation sits in the start buffers,
        which a search trip reads and never writes."""
        self._load_state(stream)
        self._momentum(t, stream)
        if self.softmax:      # theta(q) for the step's Jacobian (see above)
            self._step(self.q_new, self.p, False, 0.0, 0.0, 0.0, None, None,
                       stream)

    def search_trip(self, t, step_size, stream):
        self._in_search = True
        self._restore_start(t, stream)
        q1, p1 = self.q_new, self.p
        self._step(q1, p1, True, step_size, 0.5, 1.0, self.lp_old, None,
                   stream, start=True)
        self._likelihood(q1, stream)
        _capi.call('zshmc_zero', self.kin_new.data_ptr(), 4 * self.n_chains,
                   stream)
        self._step(q1, p1, True, step_size, 0.5, 0.0, self.lp_new,
                   self.kin_new, stream)
        _capi.call('zshmc_mh_accept', self.lp_old.data_ptr(),
                   self.lp_new.data_ptr(), self.kin_old.data_ptr(),
                   self.kin_new.data_ptr(), self.n_chains, self.chain_offset,
                   self.hmc.seed, t & 0xFFFFFFFF, None, None, None, None, None,
                   self.acc_sum.data_ptr(), self.flags.data_ptr(), stream)

    # -- n transitions from ONE call (csrc/hmc_model_run.hip) ------------------
    can_run_block = True
    block_adapts_mass = True

    def _descriptor(self):
        """zshmc_model_plan of the current buffers (rebuilt per block: the
        model's tensors may have been re-fed since the last one)."""
        hmc, c = self.hmc, _capi
        if len(self.q) > c.MAX_LATENTS:
            return None
        d = c.ModelPlan()
        d.kind = c.PLAN_KINDS[self.kind]
        d.n_latents, d.n_leapfrogs = len(self.q), hmc.n_leapfrogs
        d.softmax, d.segmented = int(self.softmax), int(self.segmented)
        d.use_mass = int(self.use_mass)
        d.n_splits = int(self.splits)
        d.n_classes = int(getattr(self, 'n_classes', 0))
        for k, qk in enumerate(self.q_rows):
            d.latent[k] = qk.data_ptr()
            d.latent_size[k] = self.n_data[k]
            d.latent_offset[k] = self.offsets[k]
            if hmc.adapt_mass is not None:
                d.latent_mass[k] = self.mass[k].data_ptr()
                d.ewmv_mean[k] = self.ewmv_mean[k].data_ptr()
                d.ewmv_var[k] = self.ewmv_var[k].data_ptr()
                d.colsum[k] = self.colsum[k].data_ptr()
        d.q_new, d.p = self.q_new.data_ptr(), self.p.data_ptr()
        d.n_chains, d.n_total, d.ld = self.n_chains, self.n_total, self.ld
        d.operand = c.ptr(self.operand)
        d.grad, d.ll = self.grad.data_ptr(), self.ll.data_ptr()
        d.lik_rows, d.width = self.lik_rows, self.width
        if self.carry_start:
            d.grad_start, d.ll_start = self.grad0.data_ptr(), \
                self.ll0.data_ptr()
            d.start_valid = int(self._start_is_valid())
        d.one_launch = int(self.traj_capacity > 0)
        d.traj_sync = self.traj_sync.data_ptr()
        d.split_ws = c.ptr(self._ws)
        self._describe(d)
        d.prior_mean, d.mean_rows = self.prior_mean.data_ptr(), self.mean_rows
        d.prior_logstd = self.prior_logstd.data_ptr()
        d.logstd_rows = self.logstd_rows
        if hmc.adapt_mass is not None:
            d.mass = self.mass_pack.data_ptr()
            d.comm_buf = self.comm_buf.data_ptr()
            d.comm_words = self.comm_buf.numel()
            d.mass_ws = self.mass_ws.data_ptr()
        d.lp_old, d.lp_new = self.lp_old.data_ptr(), self.lp_new.data_ptr()
        d.kin_old, d.kin_new = self.kin_old.data_ptr(), self.kin_new.data_ptr()
        d.accept = self.accept.data_ptr()
        d.acceptance_rate = self.acceptance_rate.data_ptr()
        d.orig_hamiltonian = self.orig_hamiltonian.data_ptr()
        d.hamiltonian = self.hamiltonian.data_ptr()
        d.log_prob = self.log_prob.data_ptr()
        d.acc_sum, d.flags = self.acc_sum.data_ptr(), self.flags.data_ptr()
        d.state = self.state.data_ptr()
        d.chain_offset, d.n_chains_global = self.chain_offset, \
            self.n_chains_global
        d.seed = hmc.seed
        d.delta, d.gamma = hmc.target_acceptance_rate, hmc.gamma
        d.t0, d.kappa = hmc.t0, hmc.kappa
        d.mu = 10.0 * hmc._init_step_size_value            # hmc.py:79 (sic)
        d.mass_decay = hmc.mass_decay
        return d

    def run_block(self, t_first, n, kind, stream, sharding, adapt_mass=False,
                  lik_scales=None, ais=None):
        """`n` transitions with the same feeds and flags -- no step-size
        search, the mass at 1 / var -- from one call into libzshmc.so; with
        `adapt_mass` every one of them updates the mass from the column sums
        of its start state and leaves those of its end state."""
        sharded = sharding is not None and sharding.active
        if adapt_mass and not self._colstats_fresh():
            self.compute_colstats(stream)
            if sharded:
                sharding.all_reduce_sum(self.comm_buf[_capi.STATS_WORDS:])
        d = self._descriptor()
        scales = None
        if lik_scales is not None:
            scales = (ctypes.c_float * n)(*[float(v) for v in lik_scales])
        elif float(self.lik_scale()) != 1.0:
            scales = (ctypes.c_float * n)(*([float(self.lik_scale())] * n))
        log_w, ends = (None, False) if ais is None else ais
        if log_w is not None and not (
                log_w.is_contiguous() and log_w.dtype == torch.float32 and
                log_w.numel() == self.n_chains):
            raise ValueError("annealing: log_weights must be a contiguous "
                             "float32 tensor with one entry per chain")
        # (a call that fails part-way has already overwritten latents: the
        # start evaluation is trusted again only behind a successful return)
        self._start_valid = False
        if adapt_mass:
            self.colsum_state = 'dirty'
        _capi.call('zshmc_hmc_model_run', ctypes.byref(d),
                   t_first & 0xFFFFFFFF, n, kind, int(bool(adapt_mass)),
                   scales, _capi.ptr(log_w), int(bool(ends)),
                   sharding._comm if sharded else None, stream)
        self.last_t = t_first + n - 1
        self.stats_local = False
        if n >= 1:
            self._own_write()
            if self.carry_start:
                self._mark_start()
        if adapt_mass:
            self._mark_colstats()
            self._mass_ones = False
        elif self.colsum_state in ('fresh', 'parts'):
            self.colsum_state = 'dirty'

    # -- one transition --------------------------------------------------------
    def transition(self, t, eps_host, stream, update=None,
                   want_colstats=False):
        self.last_t = t
        L = self.hmc.n_leapfrogs
        q, p = self.q_new, self.p
        if eps_host is None and not self._in_search and self.c_transition \
                and len(self.q) <= _capi.MAX_LATENTS:
            # the same sequence on the other side of the C-ABI (one foreign
            # call instead of ~2 L + 8; small problems: the L + 1 trips from
            # one cooperative launch) -- bit-identical
            d = self._descriptor()
            self._start_valid = False
            _capi.call('zshmc_hmc_model_transition', ctypes.byref(d),
                       t & 0xFFFFFFFF, float(self.lik_scale()), stream)
            self._own_write()
            if self.carry_start:
                self._mark_start()
            return
        # (behind a step-size search: same q, same p0 -- Appendix B 11 -- and
        # the start evaluation is still in its buffers)
        self._load_state(stream)
        self._momentum(t, stream)
        self._first_evaluation(q, stream)
        _capi.call('zshmc_zero', self.kin_new.data_ptr(), 4 * self.n_chains,
                   stream)
        # trip 0: zero-length drift, half kick (hmc.py:352-364); the drift of
        # trip i+1 rides behind the kick of trip i
        self._step(q, p, True, eps_host, 0.5, 1.0 if L >= 1 else 0.0,
                   self.lp_old, self.kin_new if L == 0 else None, stream,
                   start=True)
        if L == 0:
            self.lp_new.copy_(self.lp_old)
        for i in range(1, L + 1):
            last = i == L
            self._likelihood(q, stream, want_ll=last)
            self._step(q, p, True, eps_host, 0.5 if last else 1.0,
                       0.0 if last else 1.0, self.lp_new if last else None,
                       self.kin_new if last else None, stream)
        _capi.call('zshmc_mh_accept', self.lp_old.data_ptr(),
                   self.lp_new.data_ptr(), self.kin_old.data_ptr(),
                   self.kin_new.data_ptr(), self.n_chains, self.chain_offset,
                   self.hmc.seed, t & 0xFFFFFFFF,
                   self.acceptance_rate.data_ptr(),
                   self.orig_hamiltonian.data_ptr(),
                   self.hamiltonian.data_ptr(), self.log_prob.data_ptr(),
                   self.accept.data_ptr(), self.acc_sum.data_ptr(),
                   self.flags.data_ptr(), stream)
        self._store_state(stream)
        if self.carry_start:
            self._carry_start(stream)
        else:
            self._start_valid = False
        self._own_write()


def _to_row_period(param, chain_shape, n_data):
    """A prior parameter as a contiguous float32 [rows, n_data] matrix used
    with row period `rows` over the flattened chain axes: the leading chain
    axes it does not vary along are dropped (1 row: shared by every chain;
    lntm's eta_mean [n_docs, K] under chain axes [n_chains, n_docs]: n_docs
    rows)."""
    t = param.detach().to(torch.float32)
    full = tuple(chain_shape) + (n_data,)
    if t.dim() > len(full):
        t = t.reshape(t.shape[t.dim() - len(full):])
    shape = (1,) * (len(full) - t.dim()) + tuple(t.shape)
    t = t.reshape(shape)
    lead = 0
    while lead < len(chain_shape) and shape[lead] == 1:
        lead += 1
    tail = full[lead:]
    t = t.reshape(shape[lead:]).expand(tail).contiguous()
    rows = 1
    for d in tail[:-1]:
        rows *= int(d)
    return _aligned16(t.reshape(rows, n_data)), rows


def _flat_tensors(x):
    """The tensors inside a nested list / tuple (None and strings skipped)."""
    if isinstance(x, torch.Tensor):
        yield x
    elif isinstance(x, (list, tuple)):
        for y in x:
            for t in _flat_tensors(y):
                yield t


def _flatten_data_axes(param, data_shape):
    """A prior parameter of a latent with several data axes ([K, F] class
    rows) broadcast over them and flattened to one, leading (chain) axes
    kept."""
    nd = len(data_shape)
    lead = tuple(param.shape[:max(param.dim() - nd, 0)])
    t = param.expand(lead + tuple(data_shape))
    return t.reshape(lead + (-1,))


def _aligned16(t):
    """`t` itself, or a copy if its storage offset breaks the 16-byte
    alignment the row kernels require (a contiguous slice `param[1:]` of a
    user tensor is a view)."""
    return t if t.data_ptr() % 16 == 0 else t.clone()
