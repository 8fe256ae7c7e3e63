"""NumPy float32 restatement of /root/reference/zhusuan/hmc.py.
TEST INFRASTRUCTURE (see oracle/__init__.py) -- the checker for the HIP path
and the "port" CPU baseline of bench.py; never imported by zhusuan_amd/.

PINNED (tests/test_oracle_hmc_reference.py) against per-iteration traces
produced by the reference's OWN zhusuan/hmc.py -- loaded unmodified from
/root/reference and executed over oracle/tf_shim.py, an eager float32
torch-CPU stand-in for the TensorFlow symbols hmc.py uses (TensorFlow itself
is not installable here) -- on four cases covering step-size search, dual
averaging, EWMV mass adaptation with re-initialisation, fed adaptation
flags, multi-latent / multi-axis chains (oracle/make_golden_hmc.py ->
tests/golden/hmc_reference_traces.npz).  What that pins: control flow and
update equations, by the reference's code; what it cannot pin: TensorFlow's
Eigen kernels' last-bit rounding and TensorFlow's own random stream.
Every function below cites the hmc.py lines it restates; the only deliberate
difference is the random stream (oracle/philox.py), because TensorFlow's
Philox stream is keyed by graph-level state outside the repository.

All state and arithmetic are float32 as in the reference (hmc.py:22,72-87,
258-264).
"""
import numpy as np

from . import philox

F32 = np.float32


# ----------------------------------------------------------------------------
# helpers, hmc.py:21-61
# ----------------------------------------------------------------------------
def random_momentum(seed, iteration, shapes, mass, n_chain_dims,
                    chain_offset=0):
    """hmc.py:21-23 -- N(0,1) * sqrt(mass), on the shared Philox stream."""
    out = []
    for k, (shape, m) in enumerate(zip(shapes, mass)):
        n_chains = int(np.prod(shape[:n_chain_dims], dtype=np.int64))
        n_data = int(np.prod(shape[n_chain_dims:], dtype=np.int64))
        z = philox.normal_chain_major(seed, iteration, n_chains, n_data,
                                      chain_offset=chain_offset, latent_id=k)
        out.append(z.reshape(shape) * np.sqrt(m))
    return out


def velocity(momentum, mass):
    """hmc.py:26-27."""
    return [p / m for p, m in zip(momentum, mass)]


def hamiltonian(q, p, log_posterior, mass, data_axes):
    """hmc.py:30-35."""
    potential = -log_posterior(q)
    kinetic = F32(0.5) * sum(
        np.sum(np.square(mom) / m, axis=tuple(ax), dtype=F32)
        for mom, m, ax in zip(p, mass, data_axes))
    return potential + kinetic, -potential


def leapfrog_integrator(q, p, step_size1, step_size2, grad, mass):
    """hmc.py:38-43."""
    q = [x + F32(step_size1) * y for x, y in zip(q, velocity(p, mass))]
    grads = grad(q)
    p = [x + F32(step_size2) * y for x, y in zip(p, grads)]
    return q, p


class NumericError(FloatingPointError):
    """Stands in for tf.errors.InvalidArgumentError raised by
    tf.check_numerics (hmc.py:51-53)."""


OLD_LOG_PROB_MSG = ('HMC: old_log_prob has numeric errors! Try better '
                    'initialization.')


def get_acceptance_rate(q, p, new_q, new_p, log_posterior, mass, data_axes):
    """hmc.py:46-61."""
    old_hamiltonian, old_log_prob = hamiltonian(q, p, log_posterior, mass,
                                                data_axes)
    new_hamiltonian, new_log_prob = hamiltonian(new_q, new_p, log_posterior,
                                                mass, data_axes)
    if not np.all(np.isfinite(old_log_prob)):
        raise NumericError(OLD_LOG_PROB_MSG)
    with np.errstate(over='ignore', invalid='ignore'):
        acceptance_rate = np.exp(
            np.minimum(-new_hamiltonian + old_hamiltonian, F32(0.0)))
    is_finite = np.logical_and(np.isfinite(acceptance_rate),
                               np.isfinite(new_log_prob))
    acceptance_rate = np.where(is_finite, acceptance_rate,
                               np.zeros_like(acceptance_rate))
    return (old_hamiltonian, new_hamiltonian, old_log_prob, new_log_prob,
            acceptance_rate)


# ----------------------------------------------------------------------------
# StepsizeTuner, hmc.py:64-112
# ----------------------------------------------------------------------------
class StepsizeTuner(object):
    def __init__(self, initial_stepsize, gamma, t0, kappa, delta):
        self.gamma = F32(gamma)
        self.t0 = F32(t0)
        self.kappa = F32(kappa)
        self.delta = F32(delta)
        self.mu = F32(10 * initial_stepsize)       # :79 (sic: not log)
        self.step = F32(0.0)
        self.log_epsilon_bar = F32(0.0)
        self.h_bar = F32(0.0)

    def tune(self, acceptance_rate, fresh_start, adapt_step_size):
        """:89-112.  `adapt_step_size` is this run's flag value."""
        acceptance_rate = F32(acceptance_rate)
        fresh_start = F32(fresh_start)
        if adapt_step_size:
            self.step = (F32(1) - fresh_start) * self.step + F32(1)
            rate1 = F32(1.0) / (self.step + self.t0)
            self.h_bar = ((F32(1) - fresh_start) * (F32(1) - rate1) *
                          self.h_bar +
                          rate1 * (self.delta - acceptance_rate))
            log_epsilon = (self.mu -
                           np.sqrt(self.step) / self.gamma * self.h_bar)
            rate = np.power(self.step, -self.kappa, dtype=F32)
            self.log_epsilon_bar = (
                rate * log_epsilon +
                (F32(1) - fresh_start) * (F32(1) - rate) *
                self.log_epsilon_bar)
            return np.exp(F32(log_epsilon))
        return np.exp(self.log_epsilon_bar)


# ----------------------------------------------------------------------------
# ExponentialWeightedMovingVariance, hmc.py:115-159
# ----------------------------------------------------------------------------
class ExponentialWeightedMovingVariance(object):
    def __init__(self, decay, shape, num_chain_dims, mean_over_chains):
        self.t = F32(0.0)
        self.mean = [np.zeros(s, F32) for s in shape]
        self.var = [np.zeros(s, F32) for s in shape]
        self.decay = F32(decay)
        self.num_chain_dims = num_chain_dims
        self._mean_over_chains = mean_over_chains

    def update(self, x):
        """:130-148."""
        self.t = self.t + F32(1)
        weight = (F32(1) - self.decay) / (
            F32(1) - np.power(self.decay, self.t, dtype=F32))
        incr = [weight * (q - mean) for q, mean in zip(x, self.mean)]
        self.mean = [mean + self._mean_over_chains(i)
                     for mean, i in zip(self.mean, incr)]
        self.var = [(F32(1) - weight) * var +
                    self._mean_over_chains(i * (q - mean))
                    for var, i, q, mean in zip(self.var, incr, x, self.mean)]
        return self.var

    @staticmethod
    def get_precision(var_in):
        """:151-152  (no floor: inf when var == 0)."""
        with np.errstate(divide='ignore'):
            return [F32(1) / var for var in var_in]

    def get_updated_precision(self, x):
        return self.get_precision(self.update(x))

    def precision(self):
        return self.get_precision(self.var)


class HMCInfo(object):
    """hmc.py:162-201."""

    def __init__(self, samples, acceptance_rate, updated_step_size,
                 init_momentum, orig_hamiltonian, hamiltonian, orig_log_prob,
                 log_prob):
        self.samples = samples
        self.acceptance_rate = acceptance_rate
        self.updated_step_size = updated_step_size
        self.init_momentum = init_momentum
        self.orig_hamiltonian = orig_hamiltonian
        self.hamiltonian = hamiltonian
        self.orig_log_prob = orig_log_prob
        self.log_prob = log_prob


# ----------------------------------------------------------------------------
# HMC, hmc.py:204-522
# ----------------------------------------------------------------------------
class HMC(object):
    """Restatement of zhusuan.HMC.  Differences forced by having no TF:
      * `sample()` takes `log_joint(list_of_arrays)->[chain...]` and
        `grad(list_of_arrays)->list` callables (tf.gradients, :430-432, is
        replaced by the caller's analytic gradient);
      * latents are a list of float32 arrays updated in place;
      * the adaptation flags are passed per `step()` call (placeholders,
        examples/toy_examples/gaussian.py:40-41,57-58);
      * sharding hooks: `chain_offset` (global index of this shard's first
        chain), `n_chains_global`, `allreduce_sum` (sum over shards).
    """

    def __init__(self, step_size=1., n_leapfrogs=10, adapt_step_size=None,
                 target_acceptance_rate=0.8, gamma=0.05, t0=100, kappa=0.75,
                 adapt_mass=None, mass_collect_iters=10, mass_decay=0.99,
                 seed=0):
        self.step_size = F32(step_size)                      # :258
        self.n_leapfrogs = int(n_leapfrogs)
        self.target_acceptance_rate = F32(target_acceptance_rate)
        self.t = F32(0.0)                                    # :264
        self.adapt_step_size = adapt_step_size
        if adapt_step_size is not None:
            self.step_size_tuner = StepsizeTuner(
                step_size, gamma, t0, kappa, target_acceptance_rate)
        if adapt_mass is not None:
            if adapt_step_size is None:
                raise ValueError(
                    'If adapt mass is set, we should also adapt step size')
            self.adapt_mass = adapt_mass
        else:
            mass_collect_iters = 0                           # :276
            self.adapt_mass = None
        self.mass_collect_iters = int(mass_collect_iters)
        self.mass_decay = F32(mass_decay)
        self.seed = seed
        # test hook: a given diagonal mass (list of arrays, one per latent)
        # instead of ones / the EWMV estimate -- what the C-ABI's `mass`
        # argument of the fused transition is
        self.fixed_mass = None

    # -- set-up part of sample(), :412-456 ---------------------------------
    def sample(self, log_joint, grad, latent, chain_offset=0,
               n_chains_global=None, allreduce_sum=None):
        self._log_posterior = log_joint
        self._grad = grad
        self.q = latent                      # list of arrays, updated in place
        chain_shape = np.shape(log_joint(self.q))
        if len(chain_shape) == 0:
            raise ValueError(
                "HMC requires that the static shape of the value returned "
                "by log joint function should be at least partially defined.")
        self.n_chain_dims = len(chain_shape)
        self.chain_shape = chain_shape
        self.data_shapes = [
            (1,) * self.n_chain_dims + tuple(q.shape[self.n_chain_dims:])
            for q in self.q]
        self.data_axes = [list(range(self.n_chain_dims, len(s)))
                          for s in self.data_shapes]
        # scalar (contiguous shard) or an array of global chain indices
        self.chain_offset = (int(chain_offset) if np.ndim(chain_offset) == 0
                             else np.asarray(chain_offset, dtype=np.uint64))
        n_local = int(np.prod(chain_shape, dtype=np.int64))
        self.n_chains_global = int(n_chains_global or n_local)
        self._allreduce = allreduce_sum or (lambda a: a)
        if self.adapt_mass is not None:
            self.ewmv = ExponentialWeightedMovingVariance(
                self.mass_decay, self.data_shapes, self.n_chain_dims,
                self._mean_over_chains)
        return self

    def _mean_over_chains(self, x):
        """tf.reduce_mean(x, axis=chain_axes, keepdims=True) (:138,143) --
        over ALL chains of ALL shards."""
        s = np.sum(x, axis=tuple(range(self.n_chain_dims)), keepdims=True,
                   dtype=F32)
        s = np.asarray(self._allreduce(s), dtype=F32)
        return s / F32(self.n_chains_global)

    def _mean_scalar(self, acc):
        """tf.reduce_mean(acceptance_rate) (:326,377) over all shards."""
        s = np.asarray(self._allreduce(
            np.asarray([np.sum(acc, dtype=F32)], dtype=F32)), dtype=F32)
        return F32(s[0] / F32(self.n_chains_global))

    # -- :284-305 ------------------------------------------------------------
    def _adapt_mass(self, t, adapt_mass_flag):
        if adapt_mass_flag:
            new_mass = self.ewmv.get_updated_precision(self.q)
        else:
            new_mass = self.ewmv.precision()
        if int(t) < self.mass_collect_iters:
            return [np.ones(s, F32) for s in self.data_shapes]
        return new_mass

    # -- :308-345 ------------------------------------------------------------
    def _init_step_size(self, q, p, mass):
        factor = F32(1.5)
        step_size = self.step_size
        last_acceptance_rate = F32(1.0)
        cond = True
        self.n_init_trips = 0
        while cond:
            new_q, new_p = leapfrog_integrator(
                q, p, F32(0.0), step_size / F32(2), self._grad, mass)
            new_q, new_p = leapfrog_integrator(
                new_q, new_p, step_size, step_size / F32(2), self._grad, mass)
            _, _, _, _, acceptance_rate = get_acceptance_rate(
                q, p, new_q, new_p, self._log_posterior, mass, self.data_axes)
            acceptance_rate = self._mean_scalar(acceptance_rate)
            if acceptance_rate < self.target_acceptance_rate:
                new_step_size = step_size * (F32(1.0) / factor)
            else:
                new_step_size = step_size * factor
            cond = not ((last_acceptance_rate < self.target_acceptance_rate) ^
                        (acceptance_rate < self.target_acceptance_rate))
            step_size, last_acceptance_rate = new_step_size, acceptance_rate
            self.n_init_trips += 1
        return F32(step_size)

    # -- :348-372 ------------------------------------------------------------
    def _leapfrog(self, q, p, step_size, mass):
        for i in range(self.n_leapfrogs + 1):
            step_size1 = step_size if i > 0 else F32(0.0)
            step_size2 = (step_size if (0 < i < self.n_leapfrogs)
                          else step_size / F32(2))
            q, p = leapfrog_integrator(q, p, step_size1, step_size2,
                                       self._grad, mass)
        return q, p

    # -- one execution of sample_op, :418-520 -----------------------------
    def step(self, adapt_step_size=None, adapt_mass=None):
        """Run one transition.  The flag arguments are this run's values of
        the `adapt_step_size` / `adapt_mass` tensors given at construction
        (default: the constructor values)."""
        if adapt_step_size is None:
            adapt_step_size = bool(self.adapt_step_size)
        if adapt_mass is None:
            adapt_mass = bool(self.adapt_mass)
        self.t = self.t + F32(1.0)                           # :418
        new_t = self.t
        it = int(new_t)

        if self.adapt_mass is not None:                      # :452-456
            mass = self._adapt_mass(new_t, adapt_mass)
        else:
            mass = [np.ones(s, F32) for s in self.data_shapes]
        if self.fixed_mass is not None:
            mass = [np.asarray(m, F32).reshape(s)
                    for m, s in zip(self.fixed_mass, self.data_shapes)]

        p = random_momentum(self.seed, it, [q.shape for q in self.q], mass,
                            self.n_chain_dims, self.chain_offset)   # :458
        current_p = list(p)
        current_q = [q.copy() for q in self.q]

        if self.adapt_step_size is None:                     # :463-472
            new_step_size = self.step_size
            if_initialize_step_size = False
        else:
            if_initialize_step_size = (new_t == F32(1)) or (
                int(new_t) == self.mass_collect_iters)
            if if_initialize_step_size:
                new_step_size = self._init_step_size(current_q, current_p,
                                                     mass)
            else:
                new_step_size = self.step_size
        self.used_step_size = F32(new_step_size)

        current_q, current_p = self._leapfrog(current_q, current_p,
                                              F32(new_step_size), mass)

        # MH test, :479-498
        (old_hamiltonian, new_hamiltonian, old_log_prob, new_log_prob,
         acceptance_rate) = get_acceptance_rate(
            self.q, p, current_q, current_p, self._log_posterior, mass,
            self.data_axes)
        n_local = int(np.prod(self.chain_shape, dtype=np.int64))
        u01 = philox.uniform_per_chain(
            self.seed, it, n_local, self.chain_offset).reshape(
                self.chain_shape)
        if_accept = u01 < acceptance_rate
        for nq, oq, da in zip(current_q, self.q, self.data_axes):
            expanded = if_accept.reshape(if_accept.shape + (1,) * len(da))
            oq[...] = np.where(expanded, nq, oq)             # assign, :497
        new_log_prob = np.where(if_accept, new_log_prob, old_log_prob)

        # step-size adaptation, :501-505 / :375-380
        if self.adapt_step_size is not None:
            self.step_size = F32(self.step_size_tuner.tune(
                self._mean_scalar(acceptance_rate),
                F32(if_initialize_step_size), adapt_step_size))
        self.last_mass = mass
        self.last_accept = if_accept
        self.last_u01 = u01
        return HMCInfo(
            samples=[q.copy() for q in self.q],
            acceptance_rate=acceptance_rate,
            updated_step_size=self.step_size,
            init_momentum=p,
            orig_hamiltonian=old_hamiltonian,
            hamiltonian=new_hamiltonian,
            orig_log_prob=old_log_prob,
            log_prob=new_log_prob)


# ----------------------------------------------------------------------------
# Model providers (log-joint + analytic gradient) used by tests and bench
# ----------------------------------------------------------------------------
class DiagNormalModel(object):
    """x ~ Normal(mean[D], logstd[D]) with group_ndims = #data axes: the
    model of examples/toy_examples/gaussian.py:15-20 and BASELINE config 2.
    log-joint via univariate.py:174-181 + base.py:302-304."""

    def __init__(self, mean, logstd=None, std=None, n_chain_dims=1):
        from .distributions_ref import Normal
        self.n_chain_dims = n_chain_dims
        self._ctor = (mean, logstd, std)
        self._Normal = Normal
        self._dist = None

    def _d(self, x):
        if self._dist is None:
            mean, logstd, std = self._ctor
            self._dist = self._Normal(
                mean, std=std, logstd=logstd,
                group_ndims=x.ndim - self.n_chain_dims)
        return self._dist

    def log_joint(self, q):
        return self._d(q[0]).log_prob(q[0])

    def grad(self, q):
        return [self._d(q[0]).grad_given(q[0])]


class CallableModel(object):
    def __init__(self, log_joint, grad):
        self.log_joint = log_joint
        self.grad = grad
