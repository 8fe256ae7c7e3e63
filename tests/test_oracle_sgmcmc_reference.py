"""PIN of oracle/sgmcmc_ref.py against traces of the reference's OWN
zhusuan/sgmcmc.py (run unmodified over oracle/tf_shim.py on the shared Philox
stream; oracle/make_golden_sgmcmc.py ->
tests/golden/sgmcmc_reference_traces.npz): SGLD, PSGLD, SGHMC first/second
order with momentum resampling, SGNHT vector/scalar friction x first/second
order, two coupled latents."""
import os

import numpy as np
import pytest

from oracle import sgmcmc_ref as ref

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
SEED = 42


@pytest.fixture(scope='module')
def traces():
    return np.load(os.path.join(os.path.dirname(__file__), 'golden',
                                'sgmcmc_reference_traces.npz'))


def make_grad(prec, m):
    def grad(qs):
        w, b = qs
        gw = -prec * (w - m) - np.float32(0.2) * w * b[:, :1] ** 2
        gb = -b.copy()
        gb[:, 0] += np.float32(-0.2) * (w ** 2).sum(-1) * b[:, 0]
        return [gw.astype(np.float32), gb.astype(np.float32)]
    return grad


@pytest.mark.parametrize('name,cls,kw', CASES, ids=[c[0] for c in CASES])
def test_oracle_reproduces_reference_sgmcmc_traces(traces, name, cls, kw):
    w, b = traces['w0'].copy(), traces['b0'].copy()
    r = getattr(ref, cls)(seed=SEED, **kw).sample(
        make_grad(traces['prec'], traces['m']), [w, b])
    n_iters = traces[name + '/w'].shape[0]
    for i in range(n_iters):
        info = r.step()
        np.testing.assert_allclose(w, traces[name + '/w'][i], rtol=2e-5,
                                   atol=2e-6, err_msg='w it %d' % i)
        np.testing.assert_allclose(b, traces[name + '/b'][i], rtol=2e-5,
                                   atol=2e-6, err_msg='b it %d' % i)
        for f in ('mean_k', 'alpha'):
            if f in info:
                for k, nm in enumerate(('w', 'b')):
                    np.testing.assert_allclose(
                        info[f][k], traces['%s/%s_%s' % (name, f, nm)][i],
                        rtol=1e-4, atol=1e-7,
                        err_msg='%s[%s] it %d' % (f, nm, i))


# ---- the BNN of examples/bayesian_neural_nets/bnn_sgmcmc.py -------------------
BNN_CASES = [
    ('bnn_sghmc2', 'SGHMC', dict(learning_rate=2e-4, friction=0.2,
                                 n_iter_resample_v=4, second_order=True)),
    ('bnn_sgld', 'SGLD', dict(learning_rate=1e-4)),
    ('bnn_sgnht', 'SGNHT', dict(learning_rate=2e-4, variance_extra=0.,
                                tune_rate=50., second_order=True)),
]
BNN_SEED = 43


@pytest.fixture(scope='module')
def bnn_traces():
    return np.load(os.path.join(os.path.dirname(__file__), 'golden',
                                'sgmcmc_bnn_reference_traces.npz'))


@pytest.mark.parametrize('name,cls,kw', BNN_CASES,
                         ids=[c[0] for c in BNN_CASES])
def test_oracle_reproduces_reference_bnn_traces(bnn_traces, name, cls, kw):
    """oracle/sgmcmc_ref.py + oracle/bnn_ref.py against the reference's own
    sgmcmc.py sampling the reference's own build_bnn (imported from the
    unmodified example; oracle/make_golden_sgmcmc.py::main_bnn): several
    latents of different shapes, group_ndims = 2 priors, mini-batch rescaling."""
    from oracle import bnn_ref
    from oracle.hmc_case_data import bnn_data
    x, y, ws0, logstds, _, n_train = bnn_data()
    ws = [w.copy() for w in ws0]
    r = getattr(ref, cls)(seed=BNN_SEED, **kw).sample(
        lambda qs: bnn_ref.grad_log_joint(qs, x, y, logstds, n_train), ws)
    for i in range(bnn_traces[name + '/w0'].shape[0]):
        info = r.step()
        for k, w in enumerate(ws):
            np.testing.assert_allclose(
                w, bnn_traces['%s/w%d' % (name, k)][i], rtol=5e-5, atol=5e-6,
                err_msg='w%d it %d' % (k, i))
            for f in ('mean_k', 'alpha'):
                if f in info:
                    np.testing.assert_allclose(
                        info[f][k], bnn_traces['%s/%s_w%d' % (name, f, k)][i],
                        rtol=2e-4, atol=1e-7,
                        err_msg='%s[w%d] it %d' % (f, k, i))
