#!/usr/bin/env python
"""Golden HMC traces from the reference's OWN implementation.

Runs /root/reference/zhusuan/hmc.py -- unmodified, loaded by file path --
over oracle/tf_shim.py (an eager float32 torch-CPU stand-in for the ~45
TensorFlow symbols hmc.py uses; TensorFlow itself is not installable in this
image) with the Philox stream of oracle/philox.py behind tf.random_normal /
tf.random_uniform, and records per-iteration HMCInfo + sampler state into
tests/golden/hmc_reference_traces.npz.  tests/test_oracle_hmc_reference.py
then pins oracle/hmc_ref.py (and tests/test_gpu_hmc_reference.py the device
path) to these traces.  Needs /root/reference: run in the build container,
commit the .npz.

    python -m oracle.make_golden_hmc
"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('ZHUSUAN_REFERENCE', '/root/reference')
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import philox, tf_shim  # noqa: E402
from oracle.hmc_case_data import (  # noqa: E402
    blr_data, lntm_data, pmf_data, softmax_regression_data)


def _load(pkg, name):
    spec = importlib.util.spec_from_file_location(
        'zhusuan.' + name,
        os.path.join(REF, 'zhusuan', name.replace('.', '/') + '.py'))
    m = importlib.util.module_from_spec(spec)
    sys.modules['zhusuan.' + name] = m
    spec.loader.exec_module(m)
    return m


def _subpackage(pkg, name, members):
    """A `zhusuan.<name>` namespace holding the __all__ of the listed member
    modules (the real __init__ files also import modules this harness does
    not need: the other distributions, variational objectives ...)."""
    p = types.ModuleType('zhusuan.' + name)
    p.__path__ = [os.path.join(REF, 'zhusuan', name)]
    sys.modules['zhusuan.' + name] = p
    setattr(pkg, name, p)
    for member in members:
        m = _load(pkg, name + '.' + member)
        for k in getattr(m, '__all__', []):
            setattr(p, k, getattr(m, k))
    return p


def load_reference():
    """The reference's own hmc.py, model layer (framework/{utils,meta_bn,bn}.py,
    distributions/{utils,base,univariate,multivariate}.py) and evaluation.py, unmodified,
    over the TensorFlow-API shim.  Returns (tf, zhusuan-like namespace)."""
    tf = tf_shim.install()
    pkg = types.ModuleType('zhusuan')
    pkg.__path__ = [os.path.join(REF, 'zhusuan')]
    sys.modules['zhusuan'] = pkg
    pkg.utils = _load(pkg, 'utils')
    _subpackage(pkg, 'distributions', ['utils', 'base', 'univariate',
                                       'multivariate'])
    fw = _subpackage(pkg, 'framework', ['utils', 'meta_bn', 'bn'])
    pkg.hmc = _load(pkg, 'hmc')
    # evaluation.py imports one symbol of zhusuan.variational that only
    # is_loglikelihood() uses; AIS does not
    var = types.ModuleType('zhusuan.variational')
    var.ImportanceWeightedObjective = None
    sys.modules['zhusuan.variational'] = var
    pkg.evaluation = _load(pkg, 'evaluation')
    pkg.HMC = pkg.hmc.HMC
    pkg.BayesianNet = fw.BayesianNet
    pkg.meta_bayesian_net = fw.meta_bayesian_net
    return tf, pkg


def load_reference_example(rel_path):
    """A module of the reference's examples/ (e.g.
    'topic_models/lntm_mcem.py'), unmodified, for its MODEL FUNCTION: the
    scripts keep their training loops under `if __name__ == '__main__'` / in
    main().  `examples.utils` (tensorflow.contrib, progressbar, data-set
    downloaders -- none of it on the sampling path, none of it importable
    here) is replaced by an empty namespace; `examples.conf` is the
    reference's.  Call load_reference() first."""
    if 'examples' not in sys.modules:
        ex = types.ModuleType('examples')
        ex.__path__ = [os.path.join(REF, 'examples')]
        sys.modules['examples'] = ex
        spec = importlib.util.spec_from_file_location(
            'examples.conf', os.path.join(REF, 'examples', 'conf.py'))
        conf = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(conf)
        sys.modules['examples.conf'] = ex.conf = conf
        utils = types.ModuleType('examples.utils')
        utils.dataset = types.ModuleType('examples.utils.dataset')
        utils.average_rmse_over_batches = None
        sys.modules['examples.utils'] = ex.utils = utils
        sys.modules['examples.utils.dataset'] = utils.dataset
    try:
        import matplotlib.pyplot  # noqa: F401  (gaussian.py plots in __main__)
    except ImportError:
        mpl = types.ModuleType('matplotlib')
        mpl.pyplot = types.ModuleType('matplotlib.pyplot')
        sys.modules.setdefault('matplotlib', mpl)
        sys.modules.setdefault('matplotlib.pyplot', mpl.pyplot)
    name = 'examples.' + rel_path[:-3].replace('/', '.')
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(REF, 'examples', rel_path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference_hmc():
    tf, pkg = load_reference()
    return tf, pkg.hmc


class Stream(object):
    """The oracle's Philox mapping behind the shim's random ops."""

    def __init__(self, seed, chain_shape):
        self.seed, self.chain_shape = seed, tuple(chain_shape)
        self.n_chains = int(np.prod(chain_shape))
        self.it, self.k = 0, 0

    def begin(self, it):
        self.it, self.k = it, 0

    def normal(self, shape):
        n_data = int(np.prod(shape)) // self.n_chains
        z = philox.normal_chain_major(self.seed, self.it, self.n_chains,
                                      n_data, latent_id=self.k)
        self.k += 1
        return z.reshape(shape)

    def uniform(self, shape):
        assert tuple(shape) == self.chain_shape, (shape, self.chain_shape)
        return philox.uniform_per_chain(self.seed, self.it,
                                        self.n_chains).reshape(shape)


def run_case(tf, zs, name, make_log_joint, latents, hmc_kwargs, n_iters,
             flags, seed, chain_shape, make_observed=None):
    """flags(i) -> (adapt_step_size, adapt_mass) values fed at iteration i
    (None = the sampler was built without that adaptation)."""
    tf_shim._VARS[:] = []
    tf_shim.end_replay()
    kw = dict(hmc_kwargs)
    ph_ss = ph_m = None
    if kw.get('adapt_step_size') == 'placeholder':
        ph_ss = kw['adapt_step_size'] = tf.placeholder(tf.bool, name='adapt_ss')
    if kw.get('adapt_mass') == 'placeholder':
        ph_m = kw['adapt_mass'] = tf.placeholder(tf.bool, name='adapt_mass')
    lat_vars = {k: tf.Variable(np.asarray(v, np.float32), name=k)
                for k, v in latents.items()}
    hmc = zs.hmc.HMC(**kw)
    log_joint = make_log_joint(tf, zs, int(np.prod(chain_shape)))
    observed = {} if make_observed is None else make_observed(tf)
    stream = Stream(seed, chain_shape)
    tf_shim.set_random_source(stream.normal, stream.uniform)
    mark = tf_shim.variable_mark()
    out = {k: [] for k in ('acceptance_rate', 'updated_step_size',
                           'orig_hamiltonian', 'hamiltonian', 'orig_log_prob',
                           'log_prob', 'u01_margin')}
    for k in latents:
        out['q_' + k] = []
        out['p0_' + k] = []
    for i in range(n_iters):
        f_ss, f_m = flags(i)
        if ph_ss is not None:
            ph_ss.feed(bool(f_ss))
        if ph_m is not None:
            ph_m.feed(bool(f_m))
        stream.begin(i + 1)
        if i > 0:
            tf_shim.begin_run(mark)
        _, info = hmc.sample(log_joint, observed, lat_vars)   # = one sess.run
        tf_shim.end_replay()
        acc = info.acceptance_rate.detach().numpy()
        out['acceptance_rate'].append(acc.copy())
        out['updated_step_size'].append(
            np.float32(tf_shim._t(info.updated_step_size).detach().numpy()))
        for f in ('orig_hamiltonian', 'hamiltonian', 'orig_log_prob',
                  'log_prob'):
            out[f].append(getattr(info, f).detach().numpy().copy())
        u = stream.uniform(tuple(chain_shape))
        out['u01_margin'].append(np.float32(np.abs(u - acc).min()))
        for k in latents:
            out['q_' + k].append(lat_vars[k].numpy())
            out['p0_' + k].append(info.init_momentum[k].detach().numpy().copy())
    res = {'%s/%s' % (name, k): np.stack(v) for k, v in out.items()}
    res['%s/t' % name] = np.float32(hmc.t.numpy())
    print('%-10s iters %d  mean acc %.3f  final eps %.5f  min |u-acc| %.2e' % (
        name, n_iters, float(np.mean(out['acceptance_rate'][-1])),
        float(out['updated_step_size'][-1]), float(np.min(out['u01_margin']))))
    return res


# ---- the cases (mirrored in tests/helpers_hmc_cases.py) ----------------------
def gaussian_model(mean, logstd=None, std=None):
    """The model of examples/toy_examples/gaussian.py:15-20 built with the
    reference's OWN model layer (meta_bayesian_net -> BayesianNet.normal ->
    distributions.Normal._log_prob, reduced by group_ndims = 1): what
    HMC.sample re-enters through meta_bn.observe(**obs).log_joint()
    (hmc.py:416, meta_bn.py:93-106, bn.py:454-478, base.py:290-304,
    univariate.py:174-181)."""
    def make(tf, zs, n_chains):
        @zs.meta_bayesian_net()
        def gaussian():
            bn = zs.BayesianNet()
            if std is not None:
                bn.normal('x', tf.constant(mean), std=tf.constant(std),
                          n_samples=n_chains, group_ndims=1)
            else:
                bn.normal('x', tf.constant(mean), logstd=tf.constant(logstd),
                          n_samples=n_chains, group_ndims=1)
            return bn
        return gaussian()
    return make


def coupled_log_joint(prec_x):
    """Two latents, chain shape [4, 5]:
    -0.5 sum prec (x^2) - 0.5 sum y^2 - 0.1 (sum x)(sum y)^2 / 10."""
    def make(tf, zs, n_chains):
        px = tf.constant(prec_x)

        def log_joint(obs):
            x, y = obs['x'], obs['y']
            sx = tf.reduce_sum(x, axis=-1)
            sy = tf.reduce_sum(y, axis=-1)
            return (-0.5 * tf.reduce_sum(px * tf.square(x), axis=-1)
                    - 0.5 * tf.reduce_sum(tf.square(y), axis=-1)
                    - 0.01 * tf.square(sx) * tf.square(sy))
        return log_joint
    return make


def blr_model(X):
    """w ~ Normal(0, std=1) per chain, y ~ Bernoulli(logits = w X^T) with
    group_ndims = 1, built with the reference's own bn.normal / bn.bernoulli
    (bn.py:556-590,628-654 -> univariate.py:43-184,334-406) and tf.matmul."""
    def make(tf, zs, n_chains):
        @zs.meta_bayesian_net()
        def blr():
            bn = zs.BayesianNet()
            w = bn.normal('w', tf.zeros([X.shape[1]]), std=1.,
                          n_samples=n_chains, group_ndims=1)
            logits = tf.matmul(w, tf.constant(X), transpose_b=True)
            bn.bernoulli('y', logits, group_ndims=1)
            return bn
        return blr()
    return make


def lntm_model(eta_mean, eta_logstd, n_docs, n_topics, n_vocab):
    """The reference's OWN model function -- `lntm` of
    examples/topic_models/lntm_mcem.py:31-48, imported from the unmodified
    file -- with the E-step objective of :97-102 (two lines inside the
    script's __main__ block, restated) as the log-joint: bn.normal /
    bn.unnormalized_multinomial (multivariate.py:339-446), tf.nn.softmax,
    tf.matmul, tf.log."""
    def make(tf, zs, n_chains_total):
        example = load_reference_example('topic_models/lntm_mcem.py')
        model = example.lntm(n_chains_total // n_docs, n_docs, n_topics,
                             n_vocab, tf.constant(eta_mean),
                             tf.constant(eta_logstd))
        model.log_joint = lambda bn: (bn.cond_log_prob('eta') +
                                      bn.cond_log_prob('x'))
        return model
    return make


def pmf_model(n_users, n_items, n_factors, select_u, select_v, alphas):
    """The reference's OWN `pmf` model function
    (examples/probabilistic_matrix_factorization/pmf_hmc.py:19-31, imported
    from the unmodified file: two tf.gather, multiply, reduce_sum, sigmoid,
    Normal rating likelihood) with the log-joint of :135-141 (inside the
    script's main(), restated)."""
    def make(tf, zs, n_particles):
        example = load_reference_example(
            'probabilistic_matrix_factorization/pmf_hmc.py')
        model = example.pmf(n_users, n_items, n_factors, n_particles,
                            tf.constant(select_u), tf.constant(select_v),
                            *alphas)

        def log_joint(bn):
            log_pu, log_pv = bn.cond_log_prob(['u', 'v'])
            log_pr = bn.cond_log_prob('r')
            return (tf.reduce_sum(log_pu, axis=-1) +
                    tf.reduce_sum(log_pv, axis=-1) +
                    tf.reduce_sum(log_pr, axis=-1))
        model.log_joint = log_joint
        return model
    return make


def softmax_regression_model(X, n_cat):
    """w[n_cat, n_feat] ~ Normal(0, 1) (group_ndims = 2) per chain,
    y ~ Categorical(logits[n, k] = <X[n], w[k]>), group_ndims = 1, through
    the reference's bn.categorical (bn.py:656-682, univariate.py:409-551)."""
    def make(tf, zs, n_chains):
        @zs.meta_bayesian_net()
        def softmax_regression():
            bn = zs.BayesianNet()
            w = bn.normal('w', tf.zeros([n_cat, X.shape[1]]), std=1.,
                          n_samples=n_chains, group_ndims=2)
            Xc = tf.tile(tf.expand_dims(tf.constant(X), 0), [n_chains, 1, 1])
            logits = tf.matmul(Xc, w, transpose_b=True)     # [C, N, n_cat]
            bn.categorical('y', logits, group_ndims=1)
            return bn
        return softmax_regression()
    return make


def cases():
    rng = np.random.RandomState(2024)
    out = []
    # A: examples/toy_examples/gaussian.py shape (config 1): step-size AND mass
    # adaptation fed per run, re-initialisation at t = mass_collect_iters
    D = 10
    stdev = (1.0 / (np.arange(D) + 1)).astype(np.float32)
    out.append(dict(
        name='gauss_adapt', chain_shape=(24,),
        make_log_joint=gaussian_model(np.zeros(D, np.float32),
                                     np.log(stdev).astype(np.float32)),
        latents={'x': (0.1 * rng.normal(size=(24, D))).astype(np.float32)},
        hmc_kwargs=dict(step_size=1e-3, n_leapfrogs=5,
                        adapt_step_size='placeholder', adapt_mass='placeholder',
                        target_acceptance_rate=0.9),
        n_iters=22, flags=lambda i: (i < 16, i < 16), seed=11))
    # B: no adaptation, two latents, two chain axes
    out.append(dict(
        name='coupled', chain_shape=(4, 5),
        make_log_joint=coupled_log_joint(
            np.linspace(0.5, 2.0, 6).astype(np.float32)),
        latents={'x': rng.normal(size=(4, 5, 6)).astype(np.float32),
                 'y': rng.normal(size=(4, 5, 3)).astype(np.float32)},
        hmc_kwargs=dict(step_size=0.08, n_leapfrogs=7),
        n_iters=6, flags=lambda i: (None, None), seed=12))
    # C: step-size adaptation only (constant True), ragged D
    D = 33
    out.append(dict(
        name='gauss_ss', chain_shape=(17,),
        make_log_joint=gaussian_model(
            np.linspace(-1, 1, D).astype(np.float32),
            np.linspace(-0.7, 0.4, D).astype(np.float32)),
        latents={'x': rng.normal(size=(17, D)).astype(np.float32)},
        hmc_kwargs=dict(step_size=0.05, n_leapfrogs=4, adapt_step_size=True,
                        target_acceptance_rate=0.8),
        n_iters=15, flags=lambda i: (True, None), seed=13))
    # D: a row length the LDS-DMA ring kernel takes (D % 4 == 0, D > 128,
    # ragged last 1 KiB chunk), step-size and mass adaptation
    D = 260
    out.append(dict(
        name='gauss_ring', chain_shape=(20,),
        make_log_joint=gaussian_model(
            np.linspace(-2, 2, D).astype(np.float32),
            np.linspace(0.0, 1.2, D).astype(np.float32)),
        latents={'x': rng.normal(size=(20, D)).astype(np.float32)},
        hmc_kwargs=dict(step_size=0.02, n_leapfrogs=6,
                        adapt_step_size='placeholder', adapt_mass='placeholder',
                        target_acceptance_rate=0.8, mass_collect_iters=4),
        n_iters=26, flags=lambda i: (i < 22, i < 18), seed=14))
    # E: examples/toy_examples/gaussian.py: ITS OWN model function `gaussian`
    # (:15-20, imported from the unmodified file) driven as :27-58 does with
    # n_x = 10: mean tf.zeros, `std=` constructor path (log(std) inside
    # Normal, univariate.py:96-103), q0 = 0, eps0 = 1e-3, L = 5, delta = 0.9,
    # both adaptations on for the first half of the run
    D = 10
    stdev = (1.0 / (np.arange(D) + 1)).astype(np.float32)
    out.append(dict(
        name='gaussian_py', chain_shape=(100,),
        make_log_joint=lambda tf, zs, n_chains: load_reference_example(
            'toy_examples/gaussian.py').gaussian(D, stdev, n_chains),
        latents={'x': np.zeros((100, D), np.float32)},
        hmc_kwargs=dict(step_size=1e-3, n_leapfrogs=5,
                        adapt_step_size='placeholder', adapt_mass='placeholder',
                        target_acceptance_rate=0.9),
        n_iters=30, flags=lambda i: (i < 15, i < 15), seed=1))
    # F: Bayesian logistic regression through the reference's own model layer
    # (configs[2] family): step-size adaptation on.  The observation is given
    # with the full batch shape: distributions/utils.py:43-44 broadcasts with
    # `x *= ones_like(y)`, which rebinds in TensorFlow but is an in-place
    # (non-broadcasting) multiply on the shim's torch tensors.
    X, y, w0 = blr_data()
    out.append(dict(
        name='blr', chain_shape=(w0.shape[0],),
        make_log_joint=blr_model(X),
        make_observed=lambda tf: {'y': tf.constant(
            np.tile(y[None, :], (w0.shape[0], 1)))},
        latents={'w': w0},
        hmc_kwargs=dict(step_size=0.02, n_leapfrogs=6, adapt_step_size=True,
                        target_acceptance_rate=0.8),
        n_iters=10, flags=lambda i: (True, None), seed=15))
    # G: the topic model of lntm_mcem.py with its E-step objective, chain axes
    # [n_chains, n_docs], step-size and mass adaptation on (configs[4] family)
    beta, x, eta_mean, eta_logstd, eta0 = lntm_data()
    n_c, n_d, n_k = eta0.shape
    out.append(dict(
        name='lntm', chain_shape=(n_c, n_d),
        make_log_joint=lntm_model(eta_mean, eta_logstd, n_d, n_k,
                                  x.shape[1]),
        make_observed=lambda tf: {
            'x': tf.constant(np.tile(x[None], (n_c, 1, 1))),
            'beta': tf.constant(beta)},
        latents={'eta': eta0},
        hmc_kwargs=dict(step_size=5e-3, n_leapfrogs=5,
                        adapt_step_size='placeholder', adapt_mass='placeholder',
                        target_acceptance_rate=0.6, mass_collect_iters=3),
        n_iters=14, flags=lambda i: (i < 11, i < 9), seed=16))
    # H: a Categorical likelihood (multi-class logistic regression), the third
    # family north_star names; labels given with the full batch shape for the
    # same reason as in F (univariate.py:505-506 `given *= ones_`)
    Xs, ys, ws0 = softmax_regression_data()
    out.append(dict(
        name='softmax_reg', chain_shape=(ws0.shape[0],),
        make_log_joint=softmax_regression_model(Xs, ws0.shape[1]),
        make_observed=lambda tf: {'y': tf.constant(
            np.tile(ys[None, :], (ws0.shape[0], 1)))},
        latents={'w': ws0},
        hmc_kwargs=dict(step_size=0.03, n_leapfrogs=5, adapt_step_size=True,
                        target_acceptance_rate=0.8),
        n_iters=8, flags=lambda i: (True, None), seed=17))
    # I: the rating model of pmf_hmc.py -- HMC over the user factors with the
    # item factors and ratings observed (one chunk update of pmf_hmc.py:145-148),
    # no adaptation, as there (:119-122)
    su, sv, r, v_obs, u0, alphas = pmf_data()
    out.append(dict(
        name='pmf', chain_shape=(u0.shape[0],),
        make_log_joint=pmf_model(u0.shape[1], v_obs.shape[1], u0.shape[2],
                                 su, sv, alphas),
        make_observed=lambda tf: {'r': tf.constant(r),
                                  'v': tf.constant(v_obs)},
        latents={'u': u0},
        hmc_kwargs=dict(step_size=0.4, n_leapfrogs=6),
        n_iters=6, flags=lambda i: (None, None), seed=18))
    return out


def main():
    tf, zs = load_reference()
    res = {}
    for c in cases():
        res.update(run_case(tf, zs, **c))
        for k, v in c['latents'].items():
            res['%s/q0_%s' % (c['name'], k)] = v
    path = os.path.join(ROOT, 'tests', 'golden', 'hmc_reference_traces.npz')
    np.savez_compressed(path, **res)
    print('wrote', path, '(%d arrays)' % len(res))


if __name__ == '__main__':
    main()
