#!/usr/bin/env python
"""Golden SGMCMC traces from the reference's OWN zhusuan/sgmcmc.py, run
unmodified over oracle/tf_shim.py on the Philox stream oracle/sgmcmc_ref.py
defines (see oracle/make_golden_hmc.py for the method).  Output:
tests/golden/sgmcmc_reference_traces.npz, pinned by
tests/test_oracle_sgmcmc_reference.py (oracle) and
tests/test_gpu_sgmcmc.py::test_reference_traces (device).

    python -m oracle.make_golden_sgmcmc
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

from oracle import sgmcmc_ref, tf_shim  # noqa: E402


def load_reference_sgmcmc():
    tf = tf_shim.install()
    pkg = types.ModuleType('zhusuan')
    pkg.__path__ = [os.path.join(REF, 'zhusuan')]
    sys.modules['zhusuan'] = pkg
    mods = {}
    for name in ('utils', 'sgmcmc'):
        spec = importlib.util.spec_from_file_location(
            'zhusuan.' + name, os.path.join(REF, 'zhusuan', name + '.py'))
        m = importlib.util.module_from_spec(spec)
        sys.modules['zhusuan.' + name] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return tf, mods['sgmcmc']


class Stream(object):
    """Maps the reference's tf.random_normal calls, in program order within
    one run, onto the oracle's counters.  With momentum (SGHMC/SGNHT):
    [initial momentum per latent (sgmcmc.py:310-314; re-evaluated and
    discarded on every run after the first, when the Variable is replayed)]
    [resampled momentum per latent, only when t % n == 0 (:319-326)]
    [step noise per latent (:327-331)]; without: [step noise per latent]."""

    def __init__(self, seed, n_latents, has_momentum, n_resample):
        self.seed, self.n, self.mom, self.nres = seed, n_latents, has_momentum, n_resample
        self.t, self.k = 0, 0

    def begin(self, t):
        self.t, self.k = t, 0

    def normal(self, shape):
        n = int(np.prod(shape))
        k, self.k = self.k, self.k + 1
        plan = []
        if self.mom:
            plan += [(sgmcmc_ref.INIT_ITER, sgmcmc_ref.SUB_MOMENTUM, j)
                     for j in range(self.n)]
            if self.nres and self.t % self.nres == 0:
                plan += [(self.t, sgmcmc_ref.SUB_MOMENTUM, j)
                         for j in range(self.n)]
        plan += [(self.t, sgmcmc_ref.SUB_NOISE, j) for j in range(self.n)]
        it, sub, lat = plan[k]
        return sgmcmc_ref._normal(self.seed, it, n, sub, lat).reshape(shape)

    def uniform(self, shape):
        raise AssertionError('sgmcmc.py draws no uniforms')


def make_log_joint(tf, prec, m):
    """-0.5 sum prec (w-m)^2 - 0.5 sum b^2 - 0.1 sum (w*b0)^2 over chain
    axis 0 (the model of tests/test_gpu_sgmcmc.py)."""
    pt, mt = tf.constant(prec), tf.constant(m)

    def log_joint(obs):
        w, b = obs['w'], obs['b']
        b0 = tf_shim._t(b)[:, :1]
        return (-0.5 * tf.reduce_sum(pt * tf.square(w - mt), axis=-1)
                - 0.5 * tf.reduce_sum(tf.square(b), axis=-1)
                - 0.1 * tf.reduce_sum(tf.square(w * b0), axis=-1))
    return log_joint


CASES = [
    ('sgld', 'SGLD', dict(learning_rate=0.01)),
    ('psgld', 'PSGLD', dict(learning_rate=0.01)),
    ('sghmc1', 'SGHMC', dict(learning_rate=0.01, friction=0.3,
                             variance_estimate=0.05, n_iter_resample_v=3,
                             second_order=False)),
    ('sghmc2', 'SGHMC', dict(learning_rate=0.01, friction=0.3,
                             variance_estimate=0.0, n_iter_resample_v=4,
                             second_order=True)),
    ('sgnht_v2', 'SGNHT', dict(learning_rate=0.01, variance_extra=0.1,
                               tune_rate=1.0, second_order=True,
                               use_vector_alpha=True)),
    ('sgnht_v1', 'SGNHT', dict(learning_rate=0.01, variance_extra=0.1,
                               tune_rate=0.5, second_order=False,
                               use_vector_alpha=True, n_iter_resample_v=3)),
    ('sgnht_s2', 'SGNHT', dict(learning_rate=0.01, variance_extra=0.05,
                               second_order=True, use_vector_alpha=False)),
    ('sgnht_s1', 'SGNHT', dict(learning_rate=0.01, variance_extra=0.05,
                               second_order=False, use_vector_alpha=False,
                               n_iter_resample_v=2)),
]
C, D, DB, SEED, N_ITERS = 37, 13, 3, 42, 8


def main():
    tf, ref = load_reference_sgmcmc()
    rng = np.random.RandomState(3)
    w0 = rng.normal(size=(C, D)).astype(np.float32)
    b0 = rng.normal(size=(C, DB)).astype(np.float32)
    prec = np.linspace(0.5, 3.0, D).astype(np.float32)
    m = np.linspace(-1, 1, D).astype(np.float32)
    res = {'w0': w0, 'b0': b0, 'prec': prec, 'm': m}
    for name, cls, kw in CASES:
        tf_shim._VARS[:] = []
        tf_shim.end_replay()
        lat = {'w': tf.Variable(w0.copy(), name='w'),
               'b': tf.Variable(b0.copy(), name='b')}
        sampler = getattr(ref, cls)(**kw)
        has_mom = cls in ('SGHMC', 'SGNHT')
        stream = Stream(SEED, 2, has_mom, int(kw.get('n_iter_resample_v') or 0))
        tf_shim.set_random_source(stream.normal, stream.uniform)
        log_joint = make_log_joint(tf, prec, m)
        mark = tf_shim.variable_mark()
        tr = {'w': [], 'b': [], 'mean_k_w': [], 'mean_k_b': [],
              'alpha_w': [], 'alpha_b': []}
        for i in range(N_ITERS):
            stream.begin(i)                  # t before this run's increment
            if i > 0:
                tf_shim.begin_run(mark)
            _, info = sampler.sample(log_joint, {}, lat)
            tf_shim.end_replay()
            tr['w'].append(lat['w'].numpy())
            tr['b'].append(lat['b'].numpy())
            for f in ('mean_k', 'alpha'):
                if hasattr(info, f):
                    for nm in ('w', 'b'):
                        tr['%s_%s' % (f, nm)].append(np.asarray(
                            tf_shim._t(getattr(info, f)[nm]).detach().numpy(),
                            np.float32).copy())
        for k, v in tr.items():
            if v:
                res['%s/%s' % (name, k)] = np.stack(v)
        print('%-9s %s ok, |w| %.3f' % (name, cls, float(np.abs(tr['w'][-1]).mean())))
    path = os.path.join(ROOT, 'tests', 'golden', 'sgmcmc_reference_traces.npz')
    np.savez_compressed(path, **res)
    print('wrote', path, '(%d arrays)' % len(res))


BNN_CASES = [
    ('bnn_sghmc2', 'SGHMC', dict(learning_rate=2e-4, friction=0.2,
                                 n_iter_resample_v=4, second_order=True)),
    ('bnn_sgld', 'SGLD', dict(learning_rate=1e-4)),
    ('bnn_sgnht', 'SGNHT', dict(learning_rate=2e-4, variance_extra=0.,
                                tune_rate=50., second_order=True)),
]
BNN_SEED, BNN_ITERS = 43, 6


def main_bnn():
    """The reference's own sgmcmc.py sampling the reference's own model
    function -- build_bnn of examples/bayesian_neural_nets/bnn_sgmcmc.py:19-35,
    imported from the unmodified file -- with the log-joint of :71-76 (inside
    the script's main(), restated).  Several latents of different shapes,
    group_ndims = 2 priors, a deterministic node, mini-batch rescaling."""
    from oracle.hmc_case_data import bnn_data
    from oracle.make_golden_hmc import load_reference, load_reference_example, _load
    tf, pkg = load_reference()
    ref = _load(pkg, 'sgmcmc')
    example = load_reference_example('bayesian_neural_nets/bnn_sgmcmc.py')
    x, y, ws0, logstds, layer_sizes, n_train = bnn_data()
    names = ['w%d' % i for i in range(len(ws0))]
    res = {}
    for name, cls, kw in BNN_CASES:
        tf_shim._VARS[:] = []
        tf_shim.end_replay()
        lat = {n: tf.Variable(w.copy(), name=n) for n, w in zip(names, ws0)}
        model = example.build_bnn(tf.constant(x), layer_sizes,
                                  [tf.constant(l) for l in logstds],
                                  ws0[0].shape[0])

        def log_joint(bn):
            log_pws = bn.cond_log_prob(names)
            log_py_xw = bn.cond_log_prob('y')
            return tf.add_n(log_pws) + tf.reduce_mean(log_py_xw, 1) * n_train
        model.log_joint = log_joint
        sampler = getattr(ref, cls)(**kw)
        has_mom = cls in ('SGHMC', 'SGNHT')
        stream = Stream(BNN_SEED, len(names), has_mom,
                        int(kw.get('n_iter_resample_v') or 0))
        tf_shim.set_random_source(stream.normal, stream.uniform)
        mark = tf_shim.variable_mark()
        tr = {}
        for i in range(BNN_ITERS):
            stream.begin(i)
            if i > 0:
                tf_shim.begin_run(mark)
            _, info = sampler.sample(model, {'y': tf.constant(y)}, lat)
            tf_shim.end_replay()
            for n in names:
                tr.setdefault(n, []).append(lat[n].numpy())
                for f in ('mean_k', 'alpha'):
                    if hasattr(info, f):
                        tr.setdefault('%s_%s' % (f, n), []).append(np.asarray(
                            tf_shim._t(getattr(info, f)[n]).detach().numpy(),
                            np.float32).copy())
        for k, v in tr.items():
            res['%s/%s' % (name, k)] = np.stack(v)
        print('%-10s %s ok, |w0| %.3f' % (
            name, cls, float(np.abs(tr['w0'][-1]).mean())))
    path = os.path.join(ROOT, 'tests', 'golden',
                        'sgmcmc_bnn_reference_traces.npz')
    np.savez_compressed(path, **res)
    print('wrote', path, '(%d arrays)' % len(res))


if __name__ == '__main__':
    main()
    main_bnn()
