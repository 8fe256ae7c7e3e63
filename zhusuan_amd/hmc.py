"""HMC front-end: the `zhusuan.HMC(...).sample(meta_bn, observed, latent)`
surface (reference zhusuan/hmc.py:204-522) over the HIP kernels of
libzshmc.so.

What differs from the reference, and why (no TensorFlow):
  * latents are float32 torch tensors living on an MI355X instead of
    tf.Variables; they are updated in place by `sample_op`;
  * `sample_op` is a host object: `sample_op.run(feed_dict)` (or calling it)
    performs one transition; `HMCInfo` fields are device tensors valid until
    the next run (the reference: "must be fetched together with the sampling
    operation", hmc.py:168-172);
  * `adapt_step_size` / `adapt_mass` accept None, a Python bool, or a
    `placeholder()` whose value comes from `feed_dict` per run (the
    tf.placeholder idiom of examples/toy_examples/gaussian.py:40-41,57-58);
  * random numbers come from the documented Philox4x32-7 counter mapping
    (csrc/philox.h), not TensorFlow's graph-seeded stream.

Two execution plans, chosen in `sample()`:
  fused   -- the model is one Normal node with chain-independent parameters
             (gaussian.py:15-20, BASELINE config 2): ONE kernel per
             transition (csrc/hmc_fused_normal.hip).
  generic -- any other log-joint: torch autograd over the HIP log_prob ops
             supplies grad log p; momentum / kick / drift / MH run in the
             kernels of csrc/hmc_generic.hip.
Both use the same RNG counters and the same on-device adaptation state.
"""


import torch

from . import _capi, _symbolic
from .framework.meta_bn import MetaBayesianNet
from .utils import merge_dicts, next_sampler_seed

__all__ = ['deferred', 'HMCInfo', 'HMC', 'placeholder', 'InvalidArgumentError',
           'NativePlanFallbackWarning']

OLD_LOG_PROB_MSG = ('HMC: old_log_prob has numeric errors! Try better '
                    'initialization.')


class NativePlanFallbackWarning(UserWarning):
    """A model with a dense likelihood was refused by the native plans and
    runs on the autograd-driven generic plan (every gradient evaluation an
    autograd graph over torch / rocBLAS kernels): `hmc.plan_reason` says
    which construct was refused."""


class LikelihoodArithmeticWarning(UserWarning):
    """`HMC(likelihood_arithmetic='bf16x3')` was asked for a model whose
    likelihood has no bf16x3 kernel (more than 256 padded columns, or a
    topic model whose per-document chains do not fill the kernel's
    workgroups): the exact-fp32 kernels run; `hmc.arithmetic_reason` says
    why, `hmc.likelihood_arithmetic_used` what ran."""


class InvalidArgumentError(ArithmeticError):
    """Raised where the reference's tf.check_numerics raises
    tf.errors.InvalidArgumentError (hmc.py:51-53)."""


class placeholder(object):
    """Per-run feedable value (the reference's tf.placeholder as far as the
    sampling path uses it).
      * flags: `flag = placeholder(bool)`;
        `sample_op.run(feed_dict={flag: i < burnin})` (gaussian.py:40-41,57-58);
      * tensors: `x = placeholder(torch.float32, name='x')` may stand in
        `observed` or be read as `x.value` inside a model function; the value
        fed to the latest `sample_op.run(feed_dict={x: ...})` (NumPy array,
        list or tensor; moved to the sampler's device) stays bound until fed
        again (pmf_hmc.py:84-87,186-192).  `default` is the value before the
        first feed (needed if HMC.sample has to evaluate the model to derive
        the chain shape)."""

    def __init__(self, dtype=bool, shape=None, name=None, default=None):
        self.dtype = dtype
        self.shape = shape
        self.name = name
        self.default = default
        self._value = default

    def __repr__(self):
        return 'placeholder(%s)' % (self.name or hex(id(self)))

    @property
    def value(self):
        if self._value is None:
            raise ValueError(
                "You must feed a value for placeholder %r" % (self,))
        return self._value

    def feed(self, value, device=None):
        if isinstance(self.dtype, torch.dtype):
            value = torch.as_tensor(value, dtype=self.dtype).to(
                device if device is not None else default_feed_device())
        self._value = value


class deferred(object):
    """An `observed` value computed at run time from fed placeholders -- what
    a graph expression of placeholders is in the reference
    (`(true_rating - 1.0) / 4.0`, `tf.gather(V, neighbor_v, axis=1)`,
    pmf_hmc.py:113,121-122): `deferred(lambda: (true_rating.value - 1) / 4)`."""

    def __init__(self, fn):
        self._fn = fn

    @property
    def value(self):
        return self._fn()


def default_feed_device():
    return torch.device('cuda', torch.cuda.current_device())


def bind_feed(feed_dict, device=None):
    """Bind every placeholder key of `feed_dict` to its value."""
    if feed_dict:
        for k, v in feed_dict.items():
            if isinstance(k, placeholder):
                k.feed(v, device)


def _flag_value(flag, feed_dict, what):
    if isinstance(flag, placeholder):
        if feed_dict is not None and flag in feed_dict:
            return bool(feed_dict[flag])
        if flag.default is None:
            raise ValueError(
                "You must feed a value for placeholder %r (%s)" % (flag, what))
        return bool(flag.default)
    if isinstance(flag, torch.Tensor):
        return bool(flag.item())
    return bool(flag)


class HMCInfo(object):
    """Statistics of one HMC iteration (hmc.py:162-201).  All fields are
    device tensors refreshed in place by every run of the sampling op."""

    def __init__(self, samples, acceptance_rate, updated_step_size,
                 init_momentum, orig_hamiltonian, hamiltonian, orig_log_prob,
                 log_prob, _flush=None):
        self.samples = samples
        self.acceptance_rate = acceptance_rate
        self._updated_step_size = updated_step_size
        self._flush = _flush
        self.init_momentum = init_momentum
        self.orig_hamiltonian = orig_hamiltonian
        self.hamiltonian = hamiltonian
        self.orig_log_prob = orig_log_prob
        self.log_prob = log_prob


    @property
    def updated_step_size(self):
        """Step size for the NEXT iteration (hmc.py:514).  The fused plan
        applies the dual-averaging update of a transition in the prologue of
        the following launch; reading the value retires a pending update."""
        if self._flush is not None:
            self._flush()
        return self._updated_step_size


class _LazyMomentum(dict):
    """HMCInfo.init_momentum: p0 is a pure function of (seed, iteration,
    chain, latent, mass), so it is regenerated on demand from the Philox
    counters instead of being written to HBM every transition (4 B/element
    saved unless somebody asks)."""

    def __init__(self, plan):
        super(_LazyMomentum, self).__init__()
        self._plan = plan

    def __getitem__(self, name):
        return self._plan.regenerate_momentum(name)

    def keys(self):
        return list(self._plan.names)

    def __iter__(self):
        return iter(self._plan.names)

    def __len__(self):
        return len(self._plan.names)

    def __contains__(self, name):
        return name in self._plan.names

    def items(self):
        return [(k, self[k]) for k in self._plan.names]

    def values(self):
        return [self[k] for k in self._plan.names]


class _SampleOp(object):
    """What `HMC.sample` returns in place of a tf.Operation."""

    def __init__(self, hmc):
        self._hmc = hmc

    def run(self, feed_dict=None, sync=True):
        """Execute one HMC transition.  sync=True mirrors sess.run (returns
        after the device finished; raises InvalidArgumentError if the current
        log-prob was non-finite).  sync=False only enqueues; call
        `hmc.check_numerics()` later."""
        self._hmc._run(feed_dict, sync)

    __call__ = run

    def run_many(self, n, feed_dict=None, sync=True):
        """`n` consecutive transitions with the same feeds -- what a loop of
        `n` `sess.run(sample_op, feed_dict)` does.  Stretches of the run that
        need nothing from the host between transitions (fused plan, mass not
        adapting, no step-size search) are ONE call into libzshmc.so, which
        launches them back to back (zshmc_hmc_diag_normal_run): the
        per-transition cost of the Python front-end (~25 us) is paid once.
        HMCInfo holds the last transition's values."""
        self._hmc._run_many(int(n), feed_dict, sync)

    def anneal(self, lik_scales, log_weights, ends=True, feed_dict=None):
        """len(lik_scales) transitions, the i-th with the likelihood term of
        a native plan's joint multiplied by lik_scales[i] (AIS temperatures,
        evaluation.py:101-103), accumulating the importance log-weights
        (evaluation.py:150-163) in `log_weights` -- the annealing loop of
        AIS.run from one call (zshmc_hmc_model_run)."""
        self._hmc._run_many(len(lik_scales), feed_dict, False,
                            lik_scales=[float(v) for v in lik_scales],
                            ais=(log_weights, bool(ends)))


class HMC(object):
    """Hamiltonian Monte Carlo with dual-averaging step-size adaptation and
    diagonal mass adaptation (hmc.py:204-281; same arguments and defaults).

    Extra keyword-only arguments: `seed` (Philox key; default derives from
    zhusuan_amd.set_random_seed), `sharding`
    (zhusuan_amd.distributed.ChainSharding) for chains sharded over GPUs, and
    `likelihood_arithmetic`: 'fp32' (the dense-likelihood plans' two GEMMs
    on the exact-fp32 MFMAs), 'bf16x3' (three bfloat16 planes per float32
    operand, six bf16 MFMAs per product, float32 accumulation: float32-level
    results -- every parity test of the fp32 kernels holds on it at the same
    tolerances -- at 1.6-1.8x the fp32 matrix peak; taken where the kernels
    exist: Bernoulli / mixture-multinomial / Categorical likelihoods of <=
    256 padded columns; elsewhere the fp32 kernels run and a
    LikelihoodArithmeticWarning says so) or 'auto' (the default: bf16x3
    where a kernel exists AND one evaluation is >= 1e10 flop, i.e. bound by
    the matrix cores; fp32 for the latency-bound small problems, silently).
    `hmc.likelihood_arithmetic_used` says which ran,
    `hmc.arithmetic_reason` why fp32 did.

    `one_launch_trajectory` (default False): native model plans whose
    likelihood grid fits the device at once can run the L + 1 trips of a
    transition inside ONE cooperative launch with grid-wide barriers
    (csrc/hmc_model_traj.hip: the same device code in the same order,
    bit-identical results).  Built for the sizes the reference's own loops
    run at (lntm_mcem.py's E-step, AIS.run's 1 000 temperatures) -- and
    measured SLOWER there (1.33 vs 1.06 ms per transition,
    profiles/r05k_*): on this chip a dependent kernel boundary costs ~1.5 us
    and a grid barrier 4-13 us; what a small transition costs is the length
    of its kernels' critical paths, not their launches.

    The start evaluation (`reuse_start_evaluation`, native model plans).
    The reference re-evaluates the log-joint at the state a transition starts
    from on every `sess.run` (hmc.py:47-50).  With
    `reuse_start_evaluation=True` (default) a native plan instead starts from
    what it already has: the previous transition's last likelihood
    evaluation where the chain accepted, its own first one where it did not
    -- L likelihood launches per transition instead of L + 1, bit-identical
    results AS LONG AS the model is the same function of the same values.
    What invalidates it, automatically: an in-place torch op on a latent or on
    a tensor the likelihood reads (version counters), a new tensor fed
    through a placeholder, another sampler of this library writing the same
    latent (zhusuan_amd/_writes.py), `set_state`.  What does not, and needs a
    call: writes torch cannot see -- `x.data`, DLPack, a raw pointer --
    `hmc.latents_changed()` for a latent, `hmc.observed_changed()` for an
    observed / parameter tensor; and a log-joint that is random or depends on
    state outside its tensors.  With `reuse_start_evaluation=False` every
    transition evaluates its start, and every run re-reads the observed and
    parameter tensors (no cached padded copies), exactly as the reference's
    graph does.
    """

    def __init__(self, step_size=1., n_leapfrogs=10, adapt_step_size=None,
                 target_acceptance_rate=0.8, gamma=0.05, t0=100, kappa=0.75,
                 adapt_mass=None, mass_collect_iters=10, mass_decay=0.99,
                 *, seed=None, sharding=None, native_plans=True,
                 likelihood_arithmetic='auto', reuse_start_evaluation=True,
                 one_launch_trajectory=False):
        if likelihood_arithmetic not in ('auto', 'fp32', 'bf16x3'):
            raise ValueError("likelihood_arithmetic must be 'auto', 'fp32' "
                             "or 'bf16x3', got %r" % (likelihood_arithmetic,))
        self.likelihood_arithmetic = likelihood_arithmetic
        # see the class docstring ("The start evaluation")
        self.reuse_start_evaluation = bool(reuse_start_evaluation)
        # native model plans whose likelihood grid fits the device at once run
        # the L + 1 trips of a transition from ONE cooperative launch
        # (csrc/hmc_model_traj.hip; bit-identical to a launch per trip)
        self.one_launch_trajectory = bool(one_launch_trajectory)
        self._init_step_size_value = float(step_size)
        self.n_leapfrogs = int(n_leapfrogs)
        self.target_acceptance_rate = float(target_acceptance_rate)
        self.t = 0                                     # hmc.py:264
        self._nonadaptive_streak = 0
        self.adapt_step_size = adapt_step_size
        self.gamma, self.t0, self.kappa = float(gamma), float(t0), float(kappa)
        if adapt_mass is not None:
            if adapt_step_size is None:                # hmc.py:270-272
                raise ValueError(
                    'If adapt mass is set, we should also adapt step size')
            self.adapt_mass = adapt_mass
        else:
            mass_collect_iters = 0                     # hmc.py:276
            self.adapt_mass = None
        self.mass_collect_iters = int(mass_collect_iters)
        self.mass_decay = float(mass_decay)
        self.seed = next_sampler_seed() if seed is None else \
            int(seed) & 0xFFFFFFFFFFFFFFFF
        self.sharding = sharding
        # False keeps the autograd-driven generic plan for models the native
        # dense-likelihood plans would otherwise take (A/B and parity tests)
        self.native_plans = bool(native_plans)
        self._plan = None
        self._pending_check = False
        self._symbolic_latents = True
        self._refusal = None
        self.plan_reason = None

    # -- sample(): builds the execution plan (hmc.py:382-522) -------------
    def sample(self, meta_bn, observed, latent):
        """Return `(sample_op, hmc_info)`; see hmc.py:382-411 for the
        argument contract (log-joint callable or MetaBayesianNet; `observed`
        name->tensor; `latent` name->device tensor of shape
        chain axes + data axes, updated in place)."""
        if self._plan is not None:
            raise RuntimeError(
                "HMC.sample may be invoked once per HMC instance "
                "(reference hmc.py:218-222); declare one HMC per call.")
        if callable(meta_bn) and not isinstance(meta_bn, MetaBayesianNet):
            log_joint = meta_bn
        else:
            log_joint = lambda obs: meta_bn.observe(**obs).log_joint()
        latent_k, latent_v = [list(i) for i in zip(*latent.items())]
        for k, v in zip(latent_k, latent_v):
            if not isinstance(v, torch.Tensor):
                raise TypeError(
                    "latent['{}'] is not a torch Tensor (the device buffer "
                    "that replaces a tensorflow Variable).".format(k))
            if v.dtype != torch.float32:
                raise TypeError("latent['{}'] must be float32 (HMC is "
                                "float32-only, hmc.py:22), got {}."
                                .format(k, v.dtype))
            if not v.is_cuda:
                raise RuntimeError(
                    "latent['{}'] lives on {}; the sampler runs on an MI355X "
                    "only (no CPU fallback).".format(k, v.device))
            if not v.is_contiguous():
                raise ValueError("latent['{}'] must be contiguous."
                                 .format(k))
            if v.requires_grad:
                raise ValueError("latent['{}'] must not require grad."
                                 .format(k))
        self._log_joint = log_joint
        self._observed = dict(observed)
        # chain shape = shape of the log-joint (hmc.py:434-442)
        lp = self._eval_log_joint(latent_k, latent_v)
        if not isinstance(lp, torch.Tensor) or lp.dim() == 0:
            raise ValueError(
                "HMC requires that the static shape of the value returned "
                "by log joint function should be at least partially defined. "
                "(shape: {})".format(tuple(getattr(lp, 'shape', ()))))
        chain_shape = tuple(lp.shape)
        n_chain_dims = len(chain_shape)
        for k, v in zip(latent_k, latent_v):
            if tuple(v.shape[:n_chain_dims]) != chain_shape:
                raise ValueError(
                    "latent['{}'] has shape {} whose leading axes do not "
                    "match the chain shape {} of the log joint."
                    .format(k, tuple(v.shape), chain_shape))
        device = latent_v[0].device
        plan = self._recognise_plan(meta_bn, latent_k, latent_v, chain_shape,
                                    device)
        if plan is None:
            plan = _GenericPlan(self, latent_k, latent_v, chain_shape, device)
            reason, loud = self._refusal or ('no native plan applies', False)
            if not self.native_plans:
                reason, loud = 'native_plans=False', False
            self.plan_reason = 'generic plan: ' + reason
            if loud:
                import warnings
                warnings.warn(
                    'HMC.sample: the model has a dense likelihood but runs '
                    'on the autograd-driven generic plan -- %s.' % reason,
                    NativePlanFallbackWarning, stacklevel=2)
        else:
            self.plan_reason = 'native plan: %s' % plan.kind
        self._plan = plan
        st = plan.state
        st[_capi.ST_STEP_SIZE] = self._init_step_size_value
        info = HMCInfo(
            samples=dict(zip(latent_k, latent_v)),
            acceptance_rate=plan.acceptance_rate.view(chain_shape),
            updated_step_size=st[_capi.ST_STEP_SIZE],
            _flush=self.flush,
            init_momentum=_LazyMomentum(plan),
            orig_hamiltonian=plan.orig_hamiltonian.view(chain_shape),
            hamiltonian=plan.hamiltonian.view(chain_shape),
            orig_log_prob=plan.orig_log_prob.view(chain_shape),
            log_prob=plan.log_prob.view(chain_shape))
        self.hmc_info = info
        return _SampleOp(self), info

    def _recognise_plan(self, meta_bn, names, values, chain_shape, device):
        """The fused or a native dense-likelihood plan for this model, or
        None (generic plan).  The recognisers re-run the model on latents
        that require grad: a user autograd.Function (or a torch.no_grad()
        block) that takes a symbolic latent raises SymbolicCut THERE, not in
        sample()'s first evaluation (whose latents carry no grad).  Same
        answer as in _eval_log_joint: plain tensors from now on, and the
        recognisers run once more on those."""
        for _ in range(2):
            # (what an earlier model / the pass before the SymbolicCut noted
            # does not describe this one)
            self._refusal = None
            try:
                plan = _try_fused_plan(self, meta_bn, names, values,
                                       chain_shape, device)
                if plan is None and self.native_plans:
                    plan = _try_dense_likelihood_plan(
                        self, meta_bn, names, values, chain_shape, device)
                if plan is None and self.native_plans:
                    plan = _try_gathered_dot_plan(
                        self, meta_bn, names, values, chain_shape, device)
                return plan
            except _symbolic.SymbolicCut:
                if not self._symbolic_latents:
                    raise
                self._symbolic_latents = False
        return None

    def _note_refusal(self, reason, loud=False):
        """Why a native plan was not taken (the last, most specific reason
        wins; a loud one is not overwritten by a quiet one)."""
        if self._refusal is None or loud or not self._refusal[1]:
            self._refusal = (reason, bool(loud))

    def _eval_log_joint(self, names, values):
        # the latents travel as symbols so that the reference's literal dense
        # spellings (`w @ X.T`, `log(softmax(eta) @ phi)`) reach the fused
        # likelihood kernels instead of materialising the logits
        # (zhusuan_amd/_symbolic.py); any other op sees the plain tensor
        if self._symbolic_latents:
            try:
                joint_obs = merge_dicts(
                    {k: _symbolic.wrap_latent(v)
                     for k, v in zip(names, values)},
                    self._resolved_observed())
                return _symbolic.force(self._log_joint(joint_obs))
            except _symbolic.SymbolicCut:
                # a custom autograd.Function took a symbol into its forward:
                # plain tensors from now on (nothing is lost but the
                # recognition of the literal dense spellings)
                self._symbolic_latents = False
        joint_obs = merge_dicts(dict(zip(names, values)),
                                self._resolved_observed())
        return self._log_joint(joint_obs)                # hmc.py:426-428

    def _as_symbol(self, value):
        return _symbolic.wrap_latent(value) if self._symbolic_latents \
            else value

    def _resolved_observed(self):
        return {k: (v.value if isinstance(v, (placeholder, deferred)) else v)
                for k, v in self._observed.items()}

    @property
    def plan_kind(self):
        return None if self._plan is None else self._plan.kind

    @property
    def likelihood_arithmetic_used(self):
        """'bf16x3' when the plan's likelihood evaluations run on the
        bf16 matrix cores (csrc/b3_kernel.h), 'fp32' for the exact-fp32
        MFMA kernels, None for plans without a dense likelihood kernel."""
        plan = self._plan
        if plan is None or not hasattr(plan, 'inner_image'):
            return None
        return 'bf16x3' if plan.inner_image is not None else 'fp32'

    @property
    def arithmetic_reason(self):
        """Why the fp32 kernels run where 'auto' / 'bf16x3' was asked for
        (None when bf16x3 runs, when 'fp32' was asked for, or when the plan
        has no dense likelihood kernel)."""
        return getattr(self._plan, 'arithmetic_reason', None)

    # -- one execution of sample_op ----------------------------------------
    def _run(self, feed_dict, sync):
        plan = self._plan
        bind_feed(feed_dict, plan.device)
        self.t += 1                                       # hmc.py:418
        t = self.t
        adapt_ss = None if self.adapt_step_size is None else _flag_value(
            self.adapt_step_size, feed_dict, 'adapt_step_size')
        adapt_m = None if self.adapt_mass is None else _flag_value(
            self.adapt_mass, feed_dict, 'adapt_mass')
        stream = _capi.current_stream()
        sh = self.sharding
        plan.refresh_model()          # parameters fed / updated since last run

        # mass (hmc.py:452-456, :284-305).  The column sums of the state this
        # iteration starts from were taken at the END of the previous run
        # (they travelled in that run's one all-reduce); they are recomputed
        # here only on the first adaptive run, after set_state, or when the
        # latent was written to between runs.
        use_mass = False
        if self.adapt_mass is not None:
            use_ones = t < self.mass_collect_iters        # hmc.py:299-302
            plan.update_mass(adapt_m, use_ones, stream, sh)
            use_mass = not use_ones
        plan.use_mass = use_mass

        # step size for this iteration (hmc.py:463-472)
        init = False
        eps_host = None
        if self.adapt_step_size is not None:
            init = (t == 1) or (t == self.mass_collect_iters)
            if init:
                eps_host = self._search_step_size(plan, stream, sh)
        self.last_init = init

        # With the adapt flag off, hmc.py:108-110 re-assigns
        # step_size <- exp(log_epsilon_bar) every iteration: after two such
        # updates in a row the whole sampler state is at its fixed point, so
        # neither the mean acceptance (and its all-reduce) nor the update has
        # anything left to do.  (ST_MEAN_ACCEPT, a diagnostic slot, keeps the
        # value of the last update.)
        steady = (self.adapt_step_size is not None and not adapt_ss and
                  not init and self._nonadaptive_streak >= 2 and
                  getattr(plan, 'can_skip_acc', False))
        plan.collect_acc = not steady

        # the dual-averaging update of this transition (hmc.py:501-505): the
        # plan decides where it runs -- inside the transition kernel (fused
        # plan, all chains on this GPU), in the next launch's prologue (fused
        # plan, sharded chains: the all-reduce sits in between), or as its own
        # launch (generic plan)
        update = None
        if self.adapt_step_size is not None and not steady:
            update = (_capi.PEND_ADAPT if adapt_ss else _capi.PEND_HOLD,
                      bool(init), eps_host)
            self._nonadaptive_streak = 0 if (adapt_ss or init) else \
                self._nonadaptive_streak + 1
        # column statistics of the state this transition ENDS in, for the next
        # run's mass update: wanted while the mass flag is on (speculating
        # that the next run's flag equals this one's; a miss is recomputed)
        want_colstats = self.adapt_mass is not None and bool(adapt_m)
        plan.transition(t, eps_host, stream, update, want_colstats)  # leapfrog + MH
        # everything this transition owes the other ranks -- acceptance sum,
        # non-finite flag, column sums -- in ONE all-reduce, issued here so
        # that no accessor (get_state, updated_step_size) ever has to
        # communicate; then the step-size update where it is its own launch
        plan.finish(update, eps_host, want_colstats, stream, sh)
        self._pending_check = True
        if sync:
            self.check_numerics()

    # -- many transitions, one call where nothing needs the host ----------
    def _block_length(self, n_left, feed_dict):
        """How many of the next transitions can run as one block (ONE call
        into libzshmc.so), the dual-averaging update each of them owes
        (ZSHMC_PEND_*), and whether the mass adapts inside the block."""
        plan = self._plan
        if n_left < 2 or not getattr(plan, 'can_run_block', False):
            return 0, None, False
        sh = self.sharding
        if sh is not None and sh.active and sh.backend != 'rccl':
            return 0, None, False       # the collective is not ours to enqueue
        t = self.t + 1
        mass_in_block = False
        if self.adapt_mass is not None:
            if t <= self.mass_collect_iters:
                return 0, None, False   # ones as mass / a search lies ahead
            if _flag_value(self.adapt_mass, feed_dict, 'adapt_mass'):
                if not getattr(plan, 'block_adapts_mass', False):
                    return 0, None, False   # column statistics, mass update
                mass_in_block = True
            elif plan._mass_ones is not False:
                return 0, None, False   # the mass buffer has to be (re)made
        kind = _capi.PEND_NONE
        if self.adapt_step_size is not None:
            if t == 1 or t <= self.mass_collect_iters:
                return 0, None, False   # a step-size search lies ahead
            if _flag_value(self.adapt_step_size, feed_dict, 'adapt_step_size'):
                kind = _capi.PEND_ADAPT
            elif not getattr(plan, 'can_skip_acc', False):
                kind = _capi.PEND_HOLD  # (this plan runs the HOLD update as
                #                         its own launch every time)
            elif self._nonadaptive_streak < 2:
                return 0, None, False   # HOLD updates until the fixed point
        return n_left, kind, mass_in_block

    def _run_many(self, n, feed_dict, sync, lik_scales=None, ais=None):
        """`lik_scales` (one per transition) / `ais` = (log-weight buffer,
        the run ends with the last temperature): AIS.run's annealing loop on
        a native plan (zshmc_hmc_model_run's lik_scale_host /
        ais_log_weights)."""
        plan = self._plan
        done = 0
        while done < n:
            k, kind, mass = self._block_length(n - done, feed_dict)
            if k < 2:
                if lik_scales is not None:
                    saved = plan.lik_scale
                    plan.lik_scale = (lambda v: (lambda: v))(
                        float(lik_scales[done]))
                    try:
                        self._run(feed_dict, sync=False)
                    finally:
                        plan.lik_scale = saved
                else:
                    self._run(feed_dict, sync=False)
                if ais is not None:
                    log_w, ends = ais
                    log_w += self.hmc_info.orig_log_prob.reshape(log_w.shape)
                    if not (ends and done == n - 1):
                        log_w -= self.hmc_info.log_prob.reshape(log_w.shape)
                done += 1
                continue
            bind_feed(feed_dict, plan.device)
            plan.refresh_model()
            plan.use_mass = self.adapt_mass is not None
            self.last_init = False
            if kind == _capi.PEND_ADAPT:
                self._nonadaptive_streak = 0
            elif kind == _capi.PEND_HOLD:
                self._nonadaptive_streak += k
            extra = {}
            if mass:
                extra['adapt_mass'] = True
            if lik_scales is not None:
                extra['lik_scales'] = lik_scales[done:done + k]
            if ais is not None:
                extra['ais'] = (ais[0], ais[1] and done + k == n)
            plan.run_block(self.t + 1, k, kind, _capi.current_stream(),
                           self.sharding, **extra)
            self.t += k
            self._pending_check = True
            done += k
        if sync:
            self.check_numerics()

    def latents_changed(self):
        """Tell the sampler that a latent was written behind torch's back.

        What the sampler keeps ABOUT the latents between runs -- the
        likelihood evaluation at the current state (native model plans), the
        column sums of the mass estimator -- is dropped when a latent's
        version counter moved: every in-place torch op on the tensor does
        that.  A write through `x.data`, a raw pointer or another library
        does not; call this after one."""
        plan = self._plan
        if plan is None:
            return
        if hasattr(plan, '_start_valid'):
            plan._start_valid = False
        if plan.colsum_state in ('fresh', 'parts'):
            plan.colsum_state = 'dirty'

    def observed_changed(self):
        """Tell the sampler that an observed or parameter tensor of the model
        was written behind torch's back (`X.data[...] = ...`, DLPack, a raw
        pointer): the padded / re-laid-out copies the kernels read and the
        carried start evaluation are dropped and rebuilt on the next run."""
        from . import _ops
        _ops.clear_caches()
        plan = self._plan
        if plan is None:
            return
        if hasattr(plan, '_src'):
            plan._src = None
        if hasattr(plan, '_start_valid'):
            plan._start_valid = False

    def flush(self):
        """Retire a step-size update still owed to the last transition (the
        fused plan carries it into the next launch); afterwards the device
        state block holds `updated_step_size` & co."""
        if self._plan is not None:
            self._plan.flush(_capi.current_stream(), self.sharding)

    def _search_step_size(self, plan, stream, sh):
        """HMC._init_step_size (hmc.py:308-345): host-driven loop of dry-run
        single-leapfrog launches from the same (q, p0); runs only at t == 1
        and t == mass_collect_iters, so the host sync is off the hot loop."""
        factor = 1.5
        f32 = lambda x: float(torch.tensor(x, dtype=torch.float32))
        plan.flush(stream, sh)
        step_size = float(plan.state[_capi.ST_STEP_SIZE].item())
        delta = f32(self.target_acceptance_rate)
        last = 1.0
        cond = True
        trips = 0
        plan.begin_search(self.t, stream)
        while cond:
            plan.search_trip(self.t, step_size, stream)
            plan.reduce_stats(sh, stream)
            acc_sum, bad = plan.stats[:2].tolist()
            plan.end_search_trip()
            acc = f32(acc_sum / plan.n_chains_global)
            if bad > 0:      # every rank sees the reduced flag: all raise
                plan.flags.zero_()
                raise InvalidArgumentError(OLD_LOG_PROB_MSG)
            if acc < delta:
                new_step = f32(step_size * f32(1.0 / factor))
            else:
                new_step = f32(step_size * factor)
            cond = not ((last < delta) ^ (acc < delta))
            step_size, last = new_step, acc
            trips += 1
            if trips > 200:
                raise RuntimeError("step-size search did not terminate")
        self.n_init_trips = trips
        return step_size

    def check_numerics(self, sync=True):
        """Raise InvalidArgumentError if any transition since the last check
        started from a non-finite log-prob (tf.check_numerics, hmc.py:51-53).
        With sharded chains the flag is summed over ranks first, so every rank
        raises (a rank raising alone would leave its peers in a collective):
        this -- like `sample_op.run` itself -- is a COLLECTIVE call, every
        rank makes it.  `get_state`, `HMCInfo.updated_step_size` and `flush`
        never communicate (a run ends with its statistics already summed
        over the ranks) and may be called by one rank alone."""
        if self._plan is None:
            return
        plan = self._plan
        self.flush()
        # the one-launch trajectory kernel's barrier-fault word is read only
        # where that kernel can have run (one more device-to-host copy
        # otherwise, on every synchronous run of the small plans)
        sync_words = getattr(plan, 'traj_sync', None)
        if getattr(plan, 'traj_capacity', 0) <= 0:
            sync_words = None
        if self.sharding is not None and self.sharding.active:
            # one message: every rank raises on a peer's fault too (a rank
            # raising alone would leave the others in the next collective)
            words = (plan.flags != 0).to(torch.float64).reshape(-1)[:1]
            if sync_words is not None:
                words = torch.cat(
                    [words, (sync_words[2:3] != 0).to(torch.float64)])
            words = self.sharding.all_reduce_sum(words).tolist()
        else:
            words = [int(plan.flags.item())]       # (the copy synchronises)
            if sync_words is not None:
                words.append(int(sync_words[2].item()))
        bad = words[0] != 0
        self._pending_check = False
        if sync_words is not None and words[1] != 0:
            # the arrival counter of the barrier that timed out is left
            # non-zero: clear all of it, and stay off that kernel
            sync_words.zero_()
            plan.traj_capacity = 0
            raise RuntimeError(
                "zhusuan_amd: a grid barrier of the one-launch trajectory "
                "kernel timed out (its workgroups were not resident at once);"
                " the results of that run are invalid -- this sampler has "
                "switched to one_launch_trajectory=False")
        if bad:
            plan.flags.zero_()
            raise InvalidArgumentError(OLD_LOG_PROB_MSG)

    # -- checkpoint / resume of the sampler state -----
    def get_state(self):
        """Sampler state as host values: t, step_size, tuner triple, EWMV
        t/mean/var (the tf.Variables of hmc.py:82-87,118-123,258-264)."""
        plan = self._plan
        self.flush()
        st = plan.state.cpu()
        out = {'t': self.t, 'state': st.clone(), 'seed': self.seed}
        if self.adapt_mass is not None:
            out['ewmv_mean'] = [m.cpu().clone() for m in plan.ewmv_mean]
            out['ewmv_var'] = [v.cpu().clone() for v in plan.ewmv_var]
            out['mass'] = [m.cpu().clone() for m in plan.mass]
        return out

    def set_state(self, state):
        """Restore `get_state()`'s snapshot.  With sharded chains: call it on
        every rank alike (and write to a sharded latent on every rank or on
        none) -- whether the next run takes fresh column sums, and all-reduces
        them, is decided from rank-local state; a rank that diverges here
        enters a collective the others skip."""
        plan = self._plan
        plan.pending = None
        if hasattr(plan, '_start_valid'):
            # (a snapshot goes with a state the caller is about to write, or
            # has written: the carried start evaluation is not part of it)
            plan._start_valid = False
        if plan.colsum_state in ('fresh', 'parts'):
            plan.colsum_state = 'dirty'   # taken around the EWMV mean of before
        plan._mass_ones = None
        self._nonadaptive_streak = 0
        self.t = int(state['t'])
        self.seed = int(state['seed'])
        plan.state.copy_(state['state'])
        if self.adapt_mass is not None:
            for dst, src in zip(plan.ewmv_mean, state['ewmv_mean']):
                dst.copy_(src)
            for dst, src in zip(plan.ewmv_var, state['ewmv_var']):
                dst.copy_(src)
            for dst, src in zip(plan.mass, state['mass']):
                dst.copy_(src)


# ----------------------------------------------------------------------------
# execution plans: zhusuan_amd/plans/ (base, fused, generic, dense, recognise)
# ----------------------------------------------------------------------------
from .plans import (_DenseLikelihoodPlan, _FusedDiagNormalPlan,  # noqa: E402,F401
                    _GenericPlan, _PlanBase, _Unsupported, _prod,
                    _try_dense_likelihood_plan, _try_fused_plan,
                    _try_gathered_dot_plan, _versions)
