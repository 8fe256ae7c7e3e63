#!/usr/bin/env python
"""Generate tests/golden/* .  Run in the BUILD container only (it reads
/root/reference, which does not exist on the GPU box):

    python oracle/make_golden.py

1. ess_fixture.npz  -- outputs of the reference's own zhusuan/diagnostics.py
   (NumPy-only, loaded by file path because `import zhusuan` needs TF).
2. logprob_vectors.json -- the (params, given) vectors of the reference's
   tests (tests/distributions/test_univariate.py:128-152, :364-383, :537-565;
   tests/distributions/test_multivariate.py:327-354) with the expected values
   computed exactly the way those tests compute them (scipy.stats /
   NumPy logsumexp one-hot), in float64.
"""
import importlib.util
import json
import os

import numpy as np
from scipy import stats
from scipy.special import logsumexp

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, '..', 'tests', 'golden')
REF = '/root/reference'


def ess_fixture():
    spec = importlib.util.spec_from_file_location(
        'ref_diagnostics', os.path.join(REF, 'zhusuan', 'diagnostics.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.RandomState(20260922)
    cases = {}
    # iid normals (tests/test_diagnostics.py:13-22 shape, smaller)
    cases['iid'] = rng.normal(size=(1200, 3))
    # AR(1) chains with different correlation per dim
    x = np.zeros((900, 4))
    rho = np.array([0.3, 0.6, 0.9, 0.97])
    for t in range(1, 900):
        x[t] = rho * x[t - 1] + np.sqrt(1 - rho ** 2) * rng.normal(size=4)
    cases['ar1'] = x
    # random-walk Metropolis chain (tests/test_diagnostics.py:24-39 shape)
    n, d = 1000, 2
    c = np.zeros((n, d))
    cur = np.zeros(d)
    for t in range(n):
        prop = cur + 0.5 * rng.normal(size=d)
        if np.log(rng.uniform()) < 0.5 * (cur @ cur - prop @ prop):
            cur = prop
        c[t] = cur
    cases['rwmh'] = c
    # sticky chain with repeated values (rejections), float32 storage
    s = np.repeat(rng.normal(size=(150, 2)), 4, axis=0).astype(np.float32)
    cases['sticky_f32'] = s
    out = {}
    for k, v in cases.items():
        out[k + '_samples'] = v
        for burn in (0, 100):
            out['%s_ess_burn%d' % (k, burn)] = np.float64(
                mod.effective_sample_size(v, burn_in=burn))
        out[k + '_ess1d'] = np.array(
            [mod.effective_sample_size_1d(v[:, j]) for j in range(v.shape[1])])
    np.savez_compressed(os.path.join(GOLD, 'ess_fixture.npz'), **out)
    print('ess_fixture.npz', {k: float(out[k]) for k in out if 'burn' in k})


def _onehot(x, depth):
    ret = np.zeros((x.size, depth))
    ret[np.arange(x.size), x.flat] = 1
    return ret.reshape(list(x.shape) + [depth])


def logprob_vectors():
    vec = {'normal': [], 'bernoulli': [], 'categorical': [],
           'unnormalized_multinomial': []}
    # Normal: test_univariate.py:128-152
    for given, mean, logstd in [
            (0., 0., 0.),
            ([0.99, 0.9, 9., 99.], 1., [-3., -1., 1., 10.]),
            ([7.], [0., 4.], [[1., 2.], [3., 5.]])]:
        m = np.array(mean, np.float32)
        g = np.array(given, np.float32)
        ls = np.array(logstd, np.float32)
        tgt = stats.norm.logpdf(g, m, np.exp(ls))
        vec['normal'].append(dict(given=g.tolist(), mean=m.tolist(),
                                  logstd=ls.tolist(),
                                  log_prob=np.asarray(tgt).tolist()))
    # Bernoulli: test_univariate.py:364-383
    for logits, given in [
            (0., [0, 1]),
            ([-50., -10., -50.], [1, 1, 0]),
            ([0., 4.], [[0, 1], [0, 1]]),
            ([[2., 3., 1.], [5., 7., 4.]],
             np.ones([3, 1, 2, 3], dtype=np.int32).tolist())]:
        l = np.array(logits, np.float32)
        g = np.array(given, np.float32)
        tgt = stats.bernoulli.logpmf(g, 1. / (1. + np.exp(-l)))
        vec['bernoulli'].append(dict(logits=l.tolist(), given=g.tolist(),
                                     log_prob=np.asarray(tgt).tolist()))
    # Categorical: test_univariate.py:537-565
    for logits, given in [
            ([0.], [0, 0, 0]),
            ([-50., -10., -50.], [0, 1, 2, 1]),
            ([0., 4.], [[0, 1], [0, 1]]),
            ([[2., 3., 1.], [5., 7., 4.]],
             np.ones([3, 1, 1], dtype=np.int32).tolist())]:
        l = np.array(logits, np.float32)
        nl = l - logsumexp(l, axis=-1, keepdims=True)
        g = np.array(given, np.int32)
        tgt = np.sum(_onehot(g, l.shape[-1]) * nl, -1)
        vec['categorical'].append(dict(logits=l.tolist(), given=g.tolist(),
                                       log_prob=np.asarray(tgt).tolist()))
    # UnnormalizedMultinomial: test_multivariate.py:327-354
    for logits, given in [
            ([-50., -20., 0.], [1, 0, 3]),
            ([1., 10., 1000.], [1, 0, 0]),
            ([[2., 3., 1.], [5., 7., 4.]],
             np.ones([3, 1, 3], dtype=np.int32).tolist()),
            ([-10., 10., 20., 50.],
             [[0, 1, 99, 100], [100, 99, 1, 0]])]:
        for normalize in (True, False):
            l = np.array(logits, np.float32)
            g = np.array(given, np.float32)
            nl = l - logsumexp(l, axis=-1, keepdims=True) if normalize else l
            tgt = np.sum(g * nl, -1)
            vec['unnormalized_multinomial'].append(dict(
                logits=l.tolist(), given=g.tolist(), normalize=normalize,
                log_prob=np.asarray(tgt).tolist()))
    with open(os.path.join(GOLD, 'logprob_vectors.json'), 'w') as f:
        json.dump(vec, f, indent=1)
    print('logprob_vectors.json', {k: len(v) for k, v in vec.items()})


if __name__ == '__main__':
    os.makedirs(GOLD, exist_ok=True)
    ess_fixture()
    logprob_vectors()
