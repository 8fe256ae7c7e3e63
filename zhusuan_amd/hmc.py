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

import ctypes

import torch

from . import _capi, _symbolic, _writes
from .distributions import Normal
from .framework.bn import StochasticTensor
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
    `likelihood_arithmetic`: 'fp32' (default: the dense-likelihood plans'
    two GEMMs on the exact-fp32 MFMAs) or 'bf16x3' (three bfloat16 planes
    per float32 operand, six bf16 MFMAs per product, float32 accumulation:
    float32-level results at 1.6-1.8x the fp32 matrix peak; taken where the
    kernels exist -- Bernoulli / mixture-multinomial likelihoods of <= 256
    columns -- `hmc.likelihood_arithmetic_used` says which ran).

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
                 likelihood_arithmetic='fp32', reuse_start_evaluation=True,
                 one_launch_trajectory=False):
        if likelihood_arithmetic not in ('fp32', 'bf16x3'):
            raise ValueError("likelihood_arithmetic must be 'fp32' or "
                             "'bf16x3', got %r" % (likelihood_arithmetic,))
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
        bf16 matrix cores (csrc/linear_bf16x3.hip), 'fp32' for the exact-fp32
        MFMA kernels, None for plans without a dense likelihood kernel."""
        plan = self._plan
        if plan is None or not hasattr(plan, 'inner_image'):
            return None
        return 'bf16x3' if plan.inner_image is not None else 'fp32'

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
        flags = plan.flags
        if self.sharding is not None and self.sharding.active:
            flags = self.sharding.all_reduce_sum(
                (plan.flags != 0).to(torch.float64))
        elif sync:
            torch.cuda.current_stream().synchronize()
        bad = int(flags.item()) != 0
        self._pending_check = False
        sync_words = getattr(plan, 'traj_sync', None)
        if sync_words is not None and int(sync_words[2].item()) != 0:
            sync_words[2] = 0
            raise RuntimeError(
                "zhusuan_amd: a grid barrier of the one-launch trajectory "
                "kernel timed out (its workgroups were not resident at once);"
                " the results of that run are invalid -- construct the "
                "sampler with one_launch_trajectory=False")
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
# execution plans
# ----------------------------------------------------------------------------
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


class _DenseLikelihoodPlan(_PlanBase):
    """Native plan for the dense-likelihood families (BASELINE configs 3 / 5):
    latents with Normal priors and one observed node whose log-likelihood and
    gradient come from the fused fp32-MFMA kernels --

      'linear_bernoulli'    y ~ Bernoulli(w @ X^T [+ w2 @ X2^T ...] [+ b],
                                          group_ndims=1)
                            one latent per term, up to 1024 features in total
      'mixture_multinomial' x ~ UnnormalizedMultinomial(
                                    log_mixture(softmax(eta), phi),
                                    normalize_logits=False)   (lntm_mcem.py:33-48)
                            one latent, up to 1024 topics

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

    def __init__(self, hmc, names, values, chain_shape, device, probe, kind):
        super(_DenseLikelihoodPlan, self).__init__(hmc, names, values,
                                                   chain_shape, device)
        from . import _ops
        self._ops = _ops
        self.kind = kind
        self._probe = probe
        f32 = dict(dtype=torch.float32, device=device)
        C = self.n_chains
        self.offsets = [sum(self.n_data[:k]) for k in range(len(self.n_data))]
        D = self.n_total = sum(self.n_data)
        self.ld = ld = (D + 3) // 4 * 4
        self.packed = len(self.q) > 1 or ld != D
        self.softmax = kind == 'mixture_multinomial'
        # chain axes flattened: [C, D_k] views of the latents
        self.q_rows = [q.view(C, d) for q, d in zip(self.q, self.n_data)]
        self.p = torch.zeros(C, ld, **f32)
        self.q_new = torch.zeros(C, ld, **f32)
        self.segmented = kind in ('linear_categorical', 'gathered_dot')
        if kind == 'gathered_dot':
            # pmf_hmc.py:19-31: the latent is one of the two factor tables,
            # [chains, n, D] -- a handful of chains of 10^4..10^5 elements.
            # The gradient comes back from zshmc_gather_dot_grad as a plain
            # [C, n * D] matrix: one "segment" per chain.
            self.n_classes, self.seg_len, self.stride = 1, D, 1
            self.width = ld
            self.lik_rows = C
            self.seg_ws = torch.empty(
                int(_capi.load().zshmc_model_seg_workspace(C, D)), **f32)
            self.lp_const = torch.zeros(C, **f32)
            self._host_scalars = {}
            self._logstd_dev = torch.zeros(8, **f32)
            need_operand = False
        elif self.segmented:
            # w[c, 0:K, 0:F]: K class rows of F features per chain; the
            # likelihood kernel's "chain rows" are the (chain, class) pairs,
            # `stride` of them per chain (K rounded up to a power of two)
            K, F = (int(v) for v in self.q[0].shape[-2:])
            self.n_classes, self.seg_len = K, F
            self.stride = _ops.class_stride(K)
            self.width, self.block = _ops.likelihood_plan(F, self.stride)
            self.lik_rows = C * self.stride
            self.seg_ws = torch.empty(
                int(_capi.load().zshmc_model_seg_workspace(C, D)), **f32)
            need_operand = not (K == self.stride and F == self.width)
        else:
            self.width, self.block = _ops.likelihood_plan(ld)
            self.lik_rows = C
            need_operand = self.softmax or self.width != ld
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
        if hmc.one_launch_trajectory and kind in (
                'linear_bernoulli', 'mixture_multinomial') and \
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
        key = [(a.data_ptr(), tuple(a.shape), tuple(a.stride()), a.dtype,
                a._version) for a in t]
        if not self.carry_start:
            # reuse_start_evaluation=False: nothing about the model's tensors
            # is remembered from one run to the next (hmc.py:47-50)
            self._ops.clear_caches()
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
        if self.kind == 'linear_bernoulli':
            y = obs
            if self.packed:
                self.inner = _aligned16(ops.packed_design(
                    inner, int(y.shape[0]), self.device, self.width))
            else:
                self.inner = _aligned16(ops._padded_x(inner[0], self.width))
            self.obs = _aligned16(y.detach().to(torch.float32).contiguous())
            n_inner = self.inner.shape[0]
        elif self.kind == 'gathered_dot':
            self._refresh_gathered_dot(inner, obs)
            return
        elif self.kind == 'linear_categorical':
            self.inner = _aligned16(ops._padded_x(inner[0], self.width))
            self.obs = _aligned16(ops.labels_as_float(obs, self.n_classes))
            n_inner = self.inner.shape[0]
        else:
            phi, x = inner[0], obs
            self.inner = _aligned16(ops._padded_phi_t(phi, self.width))
            self.obs, self.obs_stride = ops._padded_counts(x)
            self.obs = _aligned16(self.obs)
            n_inner = self.inner.shape[0]
            if C % self.obs.shape[0] != 0:
                raise ValueError("counts rows do not divide the chain rows")
        # the bf16x3 kernels, where asked for and where they exist: <= 256
        # columns, and -- one document per 128-chain workgroup -- chain axes
        # that fill those workgroups
        self.inner_image = None
        if self.hmc.likelihood_arithmetic == 'bf16x3' and \
                self.kind in ('linear_bernoulli', 'mixture_multinomial') and \
                self.width in ops.BF16X3_WIDTHS:
            per_doc = C // self.obs.shape[0] \
                if self.kind == 'mixture_multinomial' else C
            n_docs = C // per_doc
            if n_docs == 1 or not ops.BF16X3_REQUIRE_FILL or \
                    per_doc % ops.BF16X3_CHAIN_BLOCK == 0 or \
                    per_doc >= 8 * ops.BF16X3_CHAIN_BLOCK:
                self.inner_image = ops.bf16x3_image(self.inner)
                self.block = ops.BF16X3_CHAIN_BLOCK
        if self.inner_image is None and self.kind != 'linear_categorical':
            self.block = ops.likelihood_plan(self.width)[1]
        R = self.lik_rows
        self.splits = ops._row_splits(R, n_inner, self.device, self.block)
        # (chain blocks x slices resident at once where the chain blocks
        # alone are: the trips then run from one cooperative launch)
        n_wg = (R + self.block - 1) // self.block
        if self.inner_image is None and 0 < n_wg <= self.traj_capacity:
            self.splits = max(1, min(self.splits, self.traj_capacity // n_wg))
        need = self.splits * R * (self.width + 1) if self.splits > 1 else 0
        if need and (self._ws is None or self._ws.numel() < need):
            self._ws = torch.empty(need, dtype=torch.float32,
                                   device=self.device)

    # -- the gathered-dot rating model (pmf_hmc.py:19-31) -----------------------
    def _host_scalar(self, t):
        """float(t) of a one-element device tensor, read once per (storage,
        version): the per-run path does not synchronise."""
        key = (t.data_ptr(), t._version)
        hit = self._host_scalars.get(key)
        if hit is None:
            if len(self._host_scalars) > 64:
                self._host_scalars.clear()
            hit = self._host_scalars[key] = (float(t.item()), t)
        return hit[0]

    def _refresh_gathered_dot(self, inner, obs):
        """inner = [side ('u' | 'v': which table the latent is), other table,
        select (latent side), select (other side) or None, likelihood spread ('std' | 'logstd', tensor), constant nodes
        [(observed tensor, mean, (how, spread))...]]."""
        import math
        ops = self._ops
        self.side, other, sel_lat, sel_other, spread, consts = inner
        self.splits = 1
        q = self.q[0]
        n_lat, D = int(q.shape[-2]), int(q.shape[-1])
        self.n_lat, self.n_dim = n_lat, D
        self.other = _aligned16(other.detach().to(torch.float32).contiguous())
        self.n_other = int(self.other.shape[-2])
        E = int(sel_lat.numel())
        self.n_pairs = E
        # CSR view of the pair list by the latent's rows (deterministic
        # scatter of the gradient) -- cached per index tensor version
        self.idx_lat, self.seg, self.order = ops._pair_csr(
            sel_lat, n_lat, 'native_lat')
        if sel_other is None:       # `other` is already gathered pair by pair
            if getattr(self, '_iota', None) is None or \
                    self._iota.numel() != E:
                self._iota = torch.arange(E, dtype=torch.int32,
                                          device=self.device)
            self.idx_other = self._iota
        else:
            self.idx_other = ops._pair_csr(sel_other, self.n_other,
                                           'native_other')[0]
        r = obs.detach().to(torch.float32).contiguous()
        if r.numel() == E:
            self.obs, self.obs_rows = r.reshape(-1), 1
        elif r.numel() == self.n_chains * E:
            self.obs, self.obs_rows = r.reshape(-1), self.n_chains
        else:
            raise ValueError("HMC (native gathered_dot plan): %d observed "
                             "ratings for %d pairs" % (r.numel(), E))
        how, sp = spread
        sp_v = self._host_scalar(sp)
        self.lik_logstd = math.log(sp_v) if how == 'std' else sp_v
        need = int(_capi.load().zshmc_gather_dot_normal_workspace(
            self.n_chains, E))
        # likelihood + gradient in one pass over the pair list where the rows
        # are <= 128 floats, a multiple of 4 (csrc/gather_dot.hip:
        # gd_fused_kernel): the CSR view cut into segments, the other side's
        # indices and the ratings in CSR order
        self.gd_fused = D % 4 == 0 and D <= 128 and E > 0
        if self.gd_fused:
            key = (self.seg.data_ptr(), self.order.data_ptr(),
                   self.idx_other.data_ptr(), self.idx_other._version)
            if getattr(self, '_gd_seg_key', None) != key:
                self._gd_seg = ops._csr_segments(self.seg, E)
                self._gd_idx_csr = self.idx_other[self.order.long()].contiguous()
                self._gd_seg_key = key
            self._gd_obs_csr = _aligned16(self.obs.view(
                self.obs_rows, E)[:, self.order.long()].contiguous())
            n_seg = int(self._gd_seg[1].numel())
            need = max(need, self.n_chains * n_seg * (D + 1))
        if self._ws is None or self._ws.numel() < max(need, 1):
            self._ws = torch.empty(max(need, 1), dtype=torch.float32,
                                   device=self.device)
        if getattr(self, 'g_pairs', None) is None or \
                self.g_pairs.numel() < self.n_chains * max(E, 1):
            self.g_pairs = torch.empty(self.n_chains * max(E, 1),
                                       dtype=torch.float32, device=self.device)
        # the observed nodes that do not depend on the latent: their
        # log-densities (a constant of this run) join every log-joint value
        stream = _capi.current_stream()
        if len(consts) > 1:
            raise _Unsupported('more than one constant node in the joint')
        if consts:
            x, mean, (chow, csp) = consts[0]
            xs = _aligned16(x.detach().to(torch.float32).contiguous())
            cols = xs.numel() // self.n_chains
            cv = self._host_scalar(csp)
            _capi.call('zshmc_state_set', self._logstd_dev.data_ptr(), 0,
                       math.log(cv) if chow == 'std' else cv, stream)
            data_shape = tuple(xs.shape[len(self.chain_shape):])
            m = mean.detach().to(torch.float32)
            if m.numel() == 1:
                m, mode = m.reshape(1), _capi.BCAST_SCALAR
            elif tuple(m.shape[-len(data_shape):]) == data_shape and \
                    m.numel() == cols:
                m, mode = _aligned16(m.contiguous().reshape(-1)), \
                    _capi.BCAST_ROW
            else:
                m, mode = _aligned16(m.expand(xs.shape).contiguous()), \
                    _capi.BCAST_FULL
            self._const_keep = (xs, m)
            _capi.call('zshmc_normal_log_prob', xs.data_ptr(), m.data_ptr(),
                       self._logstd_dev.data_ptr(), self.lp_const.data_ptr(),
                       self.n_chains, cols, mode, _capi.BCAST_SCALAR, 1,
                       stream)
        else:
            _capi.call('zshmc_zero', self.lp_const.data_ptr(),
                       4 * self.n_chains, stream)

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
        if self.kind == 'gathered_dot':
            # rating terms + d/d logit in one pass over the pairs, then the
            # deterministic scatter into the latent's rows
            lat_is_u = self.side == 'u'
            if self.gd_fused:
                sp, sr, sf, lr = self._gd_seg
                _capi.call(
                    'zshmc_gather_dot_normal_lik_grad', q.data_ptr(),
                    self.other.data_ptr(), sp.data_ptr(), sr.data_ptr(),
                    sf.data_ptr(), lr.data_ptr() if lr.numel() else None,
                    lr.numel(), self._gd_idx_csr.data_ptr(),
                    self._gd_obs_csr.data_ptr(), self.obs_rows,
                    self.lik_logstd, self.lp_const.data_ptr(), self.n_chains,
                    self.n_lat, self.n_other, self.n_pairs, sr.numel(),
                    self.n_dim, grad.data_ptr(), ll.data_ptr(),
                    self._ws.data_ptr(), stream)
                return
            _capi.call(
                'zshmc_gather_dot_normal_lik',
                q.data_ptr() if lat_is_u else self.other.data_ptr(),
                self.other.data_ptr() if lat_is_u else q.data_ptr(),
                (self.idx_lat if lat_is_u else self.idx_other).data_ptr(),
                (self.idx_other if lat_is_u else self.idx_lat).data_ptr(),
                self.obs.data_ptr(), self.obs_rows, self.lik_logstd,
                self.lp_const.data_ptr(), self.n_chains,
                self.n_lat if lat_is_u else self.n_other,
                self.n_other if lat_is_u else self.n_lat, self.n_pairs,
                self.n_dim, self.g_pairs.data_ptr(), ll.data_ptr(),
                self._ws.data_ptr(), stream)
            if self.n_pairs:
                _capi.call('zshmc_gather_dot_grad', self.other.data_ptr(),
                           self.g_pairs.data_ptr(), self.seg.data_ptr(),
                           self.order.data_ptr(), self.idx_other.data_ptr(),
                           self.n_chains, self.n_lat, self.n_other,
                           self.n_pairs, self.n_dim, grad.data_ptr(),
                           stream)
            else:
                _capi.call('zshmc_zero', grad.data_ptr(),
                           4 * grad.numel(), stream)
        elif self.kind == 'linear_categorical':
            _capi.call('zshmc_linear_categorical_log_lik', w.data_ptr(),
                       self.inner.data_ptr(), self.obs.data_ptr(),
                       self.lik_rows, self.inner.shape[0], self.width,
                       self.n_classes, self.stride, ll_ptr,
                       grad.data_ptr(), self.splits, _capi.ptr(ws),
                       stream)
        elif self.inner_image is not None and self.kind == 'linear_bernoulli':
            _capi.call('zshmc_linear_bernoulli_log_lik_bf16x3', w.data_ptr(),
                       self.inner_image.data_ptr(), self.obs.data_ptr(),
                       self.n_chains, self.inner.shape[0], self.width,
                       ll_ptr, grad.data_ptr(), self.splits,
                       _capi.ptr(ws), stream)
        elif self.inner_image is not None:
            _capi.call('zshmc_linear_multinomial_log_lik_bf16x3',
                       w.data_ptr(), self.inner_image.data_ptr(),
                       self.obs.data_ptr(), self.obs.shape[0],
                       self.obs_stride, self.n_chains, self.inner.shape[0],
                       self.width, ll_ptr, grad.data_ptr(), self.splits,
                       _capi.ptr(ws), stream)
        elif self.kind == 'linear_bernoulli':
            _capi.call('zshmc_linear_bernoulli_log_lik', w.data_ptr(),
                       self.inner.data_ptr(), self.obs.data_ptr(),
                       self.n_chains, self.inner.shape[0], self.width,
                       ll_ptr, grad.data_ptr(), self.splits,
                       _capi.ptr(ws), stream)
        else:
            _capi.call('zshmc_linear_multinomial_log_lik', w.data_ptr(),
                       self.inner.data_ptr(), self.obs.data_ptr(),
                       self.obs.shape[0], self.obs_stride, self.n_chains,
                       self.inner.shape[0], self.width, ll_ptr,
                       grad.data_ptr(), self.splits, _capi.ptr(ws),
                       stream)

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
        if self.segmented:
            d.seg_len, d.groups = self.seg_len, self.stride
            d.seg_ws = self.seg_ws.data_ptr()
        if self.kind == 'gathered_dot':
            d.inner, d.n_inner = self.other.data_ptr(), self.n_other
            d.obs, d.obs_rows = self.obs.data_ptr(), self.obs_rows
            d.gd_latent_is_u = int(self.side == 'u')
            d.gd_idx_latent = self.idx_lat.data_ptr()
            d.gd_idx_other = self.idx_other.data_ptr()
            d.gd_seg, d.gd_order = self.seg.data_ptr(), self.order.data_ptr()
            d.gd_n_latent, d.gd_n_pairs = self.n_lat, self.n_pairs
            d.gd_n_dim, d.gd_logstd = self.n_dim, self.lik_logstd
            d.gd_lp_const = self.lp_const.data_ptr()
            d.gd_g_pairs = self.g_pairs.data_ptr()
            if self.gd_fused:
                sp, sr, sf, lr = self._gd_seg
                d.gd_seg_ptr, d.gd_seg_row = sp.data_ptr(), sr.data_ptr()
                d.gd_seg_first = sf.data_ptr()
                d.gd_long_rows = lr.data_ptr() if lr.numel() else None
                d.gd_n_seg, d.gd_n_long = sr.numel(), lr.numel()
                d.gd_idx_other_csr = self._gd_idx_csr.data_ptr()
                d.gd_obs_csr = self._gd_obs_csr.data_ptr()
        else:
            d.inner, d.n_inner = self.inner.data_ptr(), self.inner.shape[0]
            d.inner_image = c.ptr(self.inner_image)
            d.obs = self.obs.data_ptr()
            if self.kind == 'mixture_multinomial':
                d.obs_rows, d.obs_stride = self.obs.shape[0], self.obs_stride
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


def _softmax_of(theta, probe):
    """True if `theta` is torch.softmax(probe, -1) (recognised on the autograd
    graph: the model is written with the ordinary torch op)."""
    fn = getattr(theta, 'grad_fn', None)
    if fn is None or type(fn).__name__ != 'SoftmaxBackward0':
        return False
    dim = getattr(fn, '_saved_dim', None)
    if dim is None:
        return False
    if dim >= 1 << 63:              # a negative axis, saved as uint64
        dim -= 1 << 64
    if dim % probe.dim() != probe.dim() - 1:
        return False
    nxt = fn.next_functions[0][0]
    return getattr(nxt, 'variable', None) is probe


def _summands_of(lp, nodes):
    """The nodes whose `cond_log_p` tensors are exactly the two operands of
    `lp = a + b` (identity of autograd nodes), else None."""
    fn = getattr(lp, 'grad_fn', None)
    if fn is None or type(fn).__name__ != 'AddBackward0' or \
            getattr(fn, '_saved_alpha', 1) != 1:
        return None
    parents = [f for f, _ in fn.next_functions]
    if len(parents) != 2 or parents[0] is None or parents[1] is None:
        return None
    picked = []
    for node in nodes:
        clp = node.__dict__.get('_cond_log_p')      # evaluated by lp only
        if clp is not None and any(clp.grad_fn is f for f in parents):
            picked.append(node)
    if len(picked) != 2 or picked[0]._cond_log_p.grad_fn is \
            picked[1]._cond_log_p.grad_fn:
        return None
    return picked


def _ops_max_classes():
    from . import _ops
    return _ops.MAX_CLASSES


class _Unsupported(ValueError):
    """The model is outside what a native plan handles: the caller falls back
    to the generic plan."""


def _try_dense_likelihood_plan(hmc, meta_bn, names, values, chain_shape,
                               device):
    from .distributions import (Bernoulli, Categorical,
                                UnnormalizedMultinomial)
    # every `return no(...)` below is a drop to the autograd-driven generic
    # plan; the reason is kept (hmc.plan_reason) and, once a dense likelihood
    # has been seen in the model, said aloud (NativePlanFallbackWarning)
    state = {'dense': False}

    def no(reason):
        hmc._note_refusal(reason, loud=state['dense'])
        return None

    if not isinstance(meta_bn, MetaBayesianNet):
        return no('the log-joint is a plain callable: no model structure to '
                  'lower')
    n_chain = len(chain_shape)
    # every latent: one data axis, or none (a per-chain scalar: a bias); a
    # single latent may have two (the [K, F] class rows of a softmax
    # regression)
    for n, q in zip(names, values):
        if q.dim() not in (n_chain, n_chain + 1) and not (
                len(values) == 1 and q.dim() == n_chain + 2):
            return no("latent '%s' has %d data axes" % (n, q.dim() - n_chain))
        if q.data_ptr() % 16 != 0 or not q.is_contiguous() or \
                q.dtype != torch.float32:
            return no("latent '%s' is not a 16-byte aligned contiguous "
                      "float32 tensor" % n)
    two_axes = values[0].dim() == n_chain + 2
    sizes = [int(q.shape[-1]) if q.dim() >= n_chain + 1 else 1
             for q in values]
    if two_axes:
        K, F = (int(v) for v in values[0].shape[-2:])
        if not (1 <= K <= _ops_max_classes() and 1 <= F <= 1024):
            return no('a [%d, %d] latent (the dense-logit Categorical kernel '
                      'takes up to %d classes x 1 024 features)'
                      % (K, F, _ops_max_classes()))
    elif min(sizes) < 1 or sum(sizes) > 1024:
        return no('%d latent columns (the dense-likelihood kernels take up '
                  'to 1 024 features / topics)' % sum(sizes))
    if len(names) > 1 and meta_bn.log_joint is not None:
        return no('a user log-joint over several latents')

    def analyse(vals):
        """(kind, [(prior mean, prior spread)], [inner tensors], observation)
        for the latents given as `vals`, or None."""
        bn = meta_bn.observe(**merge_dicts(
            {n: hmc._as_symbol(v) for n, v in zip(names, vals)},
            hmc._resolved_observed()))
        stoch = [n for n in bn.nodes.values()
                 if isinstance(n, StochasticTensor)]
        if meta_bn.log_joint is not None:
            # a user log-joint is accepted when it is, structurally, the sum
            # of two nodes' conditional log-densities -- the E-step objective
            # of lntm_mcem.py:97-102, cond_log_prob('eta') + cond_log_prob('x')
            # -- checked on the autograd graph (a tempered or re-weighted
            # joint, e.g. AIS's, has multiplications on top and is refused)
            stoch = _summands_of(bn.log_joint(), stoch) \
                if vals[0].requires_grad else [
                    n for n in stoch if n.name in analyse.accepted]
            if stoch is None:
                return no('the user log-joint is not the plain sum of two '
                          "nodes' cond_log_prob")
            analyse.accepted = [n.name for n in stoch]
        lik = [n for n in stoch if n.name not in names]
        state['dense'] = any(
            getattr(n.dist, '_lazy', None) is not None for n in lik)
        if len(stoch) != len(names) + 1:
            return no('%d stochastic nodes in the joint for %d latent(s): '
                      'one likelihood node expected'
                      % (len(stoch), len(names)))
        if len(lik) != 1 or not lik[0].is_observed():
            return no('no single observed likelihood node')
        priors = []
        for name, v in zip(names, vals):
            node = [n for n in stoch if n.name == name]
            if len(node) != 1:
                return no("latent '%s' is not a node of the joint" % name)
            pd = node[0].dist
            if type(pd) is not Normal or pd.use_path_derivative or \
                    pd.group_ndims != v.dim() - n_chain:
                return no("the prior of '%s' is not a Normal over its data "
                          "axes (group_ndims = %d)" % (name,
                                                       v.dim() - n_chain))
            # (a prior whose parameters depend on another latent -- a
            # hierarchical scale -- requires grad here: the generic plan)
            if pd.mean.requires_grad or pd.given_spread[1].requires_grad:
                return no("the prior of '%s' has parameters that depend on "
                          "a latent (hierarchical prior)" % name)
            priors.append((pd.mean, pd.given_spread))
        ld = lik[0].dist
        obs = lik[0].tensor
        lazy = getattr(ld, '_lazy', None)
        if lazy is None:
            return no("the logits of '%s' are not a dense contraction of the "
                      "latents that the symbolic layer recognises "
                      "(zhusuan_amd/_symbolic.py): they are materialised"
                      % lik[0].name)
        if type(ld) is Bernoulli:
            if two_axes:
                return no('a latent with two data axes under a Bernoulli')
            if ld.group_ndims != 1 or lazy.design_requires_grad() or \
                    obs.dim() != 1 or obs.shape[0] != lazy.n_rows or \
                    obs.requires_grad or len(lazy.terms) != len(vals):
                return no('Bernoulli likelihood outside the native shape: '
                          'group_ndims = 1, labels [N], constant design '
                          'matrices, one term per latent')
            # one term per latent, in the order of the latents
            inner = []
            for v in vals:
                term = [t for t in lazy.terms if t[0] is v]
                if len(term) != 1 or term[0][2] != (v.dim() == n_chain):
                    return no('a latent enters the logits more than once '
                              '(or not at all)')
                inner.append(term[0][1])
            return 'linear_bernoulli', priors, inner, obs
        if type(ld) is Categorical:
            value = vals[0]
            if not two_axes or lazy.w is not value:
                return no('Categorical logits that are not X @ w^T of the '
                          'one latent w[..., K, F]')
            if ld.group_ndims != 1 or not lazy.fused_ok() or \
                    obs.requires_grad or obs.dim() < 1 or \
                    obs.numel() != lazy.n_rows or \
                    obs.shape[-1] != lazy.n_rows:
                return no('Categorical likelihood outside the native shape: '
                          'group_ndims = 1, labels [N], at most %d classes x '
                          '%d features' % (_ops_max_classes(), 1024))
            return 'linear_categorical', priors, [lazy.X], obs
        if type(ld) is UnnormalizedMultinomial:
            value = vals[0]
            if len(vals) != 1 or value.dim() != n_chain + 1 or \
                    ld.group_ndims != 0 or ld.normalize_logits or \
                    lazy.phi.requires_grad or obs.requires_grad:
                return no('UnnormalizedMultinomial outside the native shape: '
                          'one latent, group_ndims = 0, '
                          'normalize_logits = False, constant phi')
            if lazy.softmax_source is not None:
                # the literal spelling, lowered symbolically: theta IS
                # softmax(latent) by construction
                if lazy.softmax_source is not value:
                    return no('theta is not softmax(latent)')
            elif value.requires_grad and not _softmax_of(lazy.theta, value):
                return no('theta is not softmax(latent)')
            batch = tuple(lazy.shape[:-1])
            gs = tuple(obs.shape)
            if not (len(gs) >= 1 and gs[-1] == lazy.phi.shape[1] and
                    len(gs) - 1 <= len(batch) and
                    gs[:-1] == batch[len(batch) - (len(gs) - 1):]):
                return no('the counts do not line up with the trailing '
                          'chain axes')
            return 'mixture_multinomial', priors, [lazy.phi], obs
        return no('likelihood %s has no native kernel' % type(ld).__name__)

    analyse.accepted = []
    found = analyse([q.detach().requires_grad_(True) for q in values])
    if found is None:
        return None
    kind = found[0]

    # The per-run re-evaluation of the model function only has to find the
    # parameter tensors again, so it is given META tensors for the latents:
    # whatever the function computes from them before the lazy contraction
    # (torch.softmax(eta, -1), lntm_mcem.py:39) is shape arithmetic, not a
    # launch and not a [rows, K] temporary on the device.  A function that
    # does more with a latent than that (mixes it with device tensors)
    # fails on the meta tensor and is evaluated on the latents themselves
    # from then on.
    q_meta = [torch.empty_like(q, device='meta') for q in values]
    on_meta = [True]

    def probe():
        f = None
        if on_meta[0]:
            try:
                f = analyse(q_meta)
            except Exception:                            # noqa: BLE001
                f = None
            if f is None or f[0] != kind:
                on_meta[0], f = False, None
        if f is None:
            f = analyse(list(values))
        if f is None or f[0] != kind:
            raise ValueError(
                "HMC (native %s plan): the model changed structure between "
                "runs; build a new HMC." % kind)
        return f[1], f[2], f[3]

    # prior parameters that do not fit the row-period addressing (more axes
    # than the latent, leading axes that are neither 1 nor the chain axes,
    # different periods for different latents): the generic plan, not an
    # exception out of HMC.sample
    try:
        return _DenseLikelihoodPlan(hmc, names, values, chain_shape, device,
                                    probe, kind)
    except _Unsupported as e:
        return no(str(e))


def _sum_tree_leaves(lp):
    """The autograd leaves' grad_fns if `lp` is built from its differentiable
    inputs by nothing but additions (alpha = 1) and sums over axes -- the
    shape of pmf_hmc.py:135-141, `reduce_sum(log_pu) + reduce_sum(log_pv) +
    reduce_sum(log_pr)` -- else None.  Constant summands (no grad_fn) are
    invisible here; their value is checked numerically by the caller."""
    leaves = []

    def walk(fn):
        if fn is None:
            return True
        name = type(fn).__name__
        if name == 'AddBackward0':
            if getattr(fn, '_saved_alpha', 1) != 1:
                return False
            return all(walk(f) for f, _ in fn.next_functions)
        if name in ('SumBackward0', 'SumBackward1'):
            return all(walk(f) for f, _ in fn.next_functions)
        leaves.append(fn)
        return True

    fn = getattr(lp, 'grad_fn', None)
    if fn is None or not walk(fn):
        return None
    return leaves


def _try_gathered_dot_plan(hmc, meta_bn, names, values, chain_shape, device):
    """The rating model of pmf_hmc.py:19-31: ONE latent factor table
    [chains, n, D] with a Normal prior, an observed Normal node whose mean is
    sigmoid(gathered_dot(latent, ...)) (zs.gathered_dot, or the reference's
    two gathers, a product and a reduce_sum), any other observed Normal node
    as a constant, and a log-joint that is the plain sum of the nodes'
    log-densities over their non-chain axes (the default one, or
    pmf_hmc.py:135-141)."""
    state = {'dense': False}

    def no(reason):
        hmc._note_refusal(reason, loud=state['dense'])
        return None

    if not isinstance(meta_bn, MetaBayesianNet) or len(names) != 1:
        return None
    name, q = names[0], values[0]
    n_chain = len(chain_shape)
    if q.dim() != n_chain + 2 or q.dtype != torch.float32 or \
            not q.is_contiguous() or q.data_ptr() % 16 != 0:
        return None
    n_total = int(q.shape[-1]) * int(q.shape[-2])

    def nodes_of(val):
        bn = meta_bn.observe(**merge_dicts(
            {name: hmc._as_symbol(val)}, hmc._resolved_observed()))
        return bn, [n for n in bn.nodes.values()
                    if isinstance(n, StochasticTensor)]

    def parts(stoch, accepted):
        """(priors, inner, obs) from the nodes named in `accepted`."""
        by_name = {n.name: n for n in stoch}
        if any(k not in by_name for k in accepted):
            return None
        prior = by_name[name].dist
        lik_name = accepted[1]
        lik = by_name[lik_name]
        gd = _symbolic.gathered_dot_mean(lik.dist._mean)
        if type(prior) is not Normal or type(lik.dist) is not Normal or \
                gd is None or not lik.is_observed():
            return None
        consts = []
        for k in accepted[2:]:
            d = by_name[k].dist
            if type(d) is not Normal or not by_name[k].is_observed():
                return None
            consts.append((by_name[k].tensor, d.mean, d.given_spread))
        sel_lat, sel_other = (gd['su'], gd['sv']) if gd['side'] == 'u' \
            else (gd['sv'], gd['su'])
        inner = [gd['side'], gd['other'], sel_lat, sel_other,
                 lik.dist.given_spread, consts]
        return [(prior.mean, prior.given_spread)], inner, lik.tensor, gd

    # -- build-time analysis on the real latent, with the log-joint ----------
    probe_q = q.detach().requires_grad_(True)
    bn, stoch = nodes_of(probe_q)
    lat_nodes = [n for n in stoch if n.name == name]
    cands = [n for n in stoch if n.name != name and type(n.dist) is Normal
             and _symbolic.gathered_dot_mean(n.dist._mean) is not None]
    if len(lat_nodes) != 1 or len(cands) != 1:
        return None
    state['dense'] = True
    lik = cands[0]
    gd = _symbolic.gathered_dot_mean(lik.dist._mean)
    if gd['latent'] is not probe_q:
        return no('the gathered dot is not over the sampled latent')
    pd = lat_nodes[0].dist
    if type(pd) is not Normal or pd.use_path_derivative or \
            pd.mean.requires_grad or pd.given_spread[1].requires_grad:
        return no("the prior of '%s' is not a Normal with constant "
                  "parameters" % name)
    if lik.dist.given_spread[1].numel() != 1 or \
            lik.dist.given_spread[1].requires_grad:
        return no("the likelihood '%s' does not have ONE constant scale"
                  % lik.name)
    if n_total % 4 != 0:
        return no('a latent table of %d elements per chain (the native '
                  'gathered-dot plan needs a multiple of 4)' % n_total)
    lp = bn.log_joint()
    if tuple(lp.shape) != tuple(chain_shape):
        return no('the log-joint does not have the chain shape')
    leaves = _sum_tree_leaves(lp)
    want = {id(lat_nodes[0].__dict__.get('_cond_log_p').grad_fn)
            if lat_nodes[0].__dict__.get('_cond_log_p') is not None else None,
            id(lik.__dict__.get('_cond_log_p').grad_fn)
            if lik.__dict__.get('_cond_log_p') is not None else None}
    if leaves is None or None in want or len(leaves) != 2 or \
            {id(f) for f in leaves} != want:
        return no('the log-joint is not the plain sum of the prior and the '
                  "rating likelihood's log-densities (+ constants)")
    # constant summands: the other evaluated nodes (observed, no gradient)
    const_names = [n.name for n in stoch
                   if n is not lat_nodes[0] and n is not lik and
                   n.__dict__.get('_cond_log_p') is not None]
    for k in const_names:
        node = [n for n in stoch if n.name == k][0]
        if node.__dict__['_cond_log_p'].requires_grad:
            return no("node '%s' depends on the latent" % k)
        if type(node.dist) is not Normal or not node.is_observed() or \
                node.dist.given_spread[1].numel() != 1:
            return no("constant node '%s' is not an observed Normal with "
                      "one scale" % k)
    if len(const_names) > 1:
        return no('more than one constant node in the joint')
    accepted = [name, lik.name] + const_names
    lp_user = lp.detach().reshape(-1).to(torch.float32)

    q_meta = torch.empty_like(q, device='meta')
    on_meta = [True]

    def probe():
        f = None
        if on_meta[0]:
            try:
                f = parts(nodes_of(q_meta)[1], accepted)
            except Exception:                            # noqa: BLE001
                f = None
            if f is None:
                on_meta[0] = False
        if f is None:
            f = parts(nodes_of(q)[1], accepted)
        if f is None:
            raise ValueError(
                "HMC (native gathered_dot plan): the model changed structure "
                "between runs; build a new HMC.")
        return f[0], f[1], f[2]

    try:
        plan = _DenseLikelihoodPlan(hmc, names, values, chain_shape, device,
                                    probe, 'gathered_dot')
    except _Unsupported as e:
        return no(str(e))
    # the constants are invisible to the structural check: the native
    # log-joint at the current state must equal the user's
    stream = _capi.current_stream()
    plan._load_state(stream)
    plan._first_evaluation(plan.q_new, stream)
    plan._step(plan.q_new, plan.p, True, 0.0, 0.0, 0.0, plan.lp_new, None,
               stream, start=True)
    diff = float((plan.lp_new - lp_user).abs().max().item())
    scale = max(1.0, float(lp_user.abs().max().item()))
    if not diff <= 2e-5 * scale + 1e-3:
        return no('the log-joint holds terms the native plan does not '
                  'account for (native - user = %.3g)' % diff)
    return plan


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
