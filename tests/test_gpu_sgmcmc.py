"""GPU parity of the SGMCMC samplers (zhusuan_amd/sgmcmc.py over
csrc/sgmcmc.hip) against the oracle restatement of zhusuan/sgmcmc.py on the
same Philox stream, and the reference's own statistical test
(tests/test_mcmc.py:14-88, Fig. 1 of the SGHMC paper) on the device path."""
import numpy as np
import pytest

from oracle import sgmcmc_ref as ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    return zs, torch, torch.device('cuda', 0)


def _model(torch, dev, D):
    """log p(w, b) = -0.5 sum prec (w - m)^2 - 0.5 sum b^2 - 0.1 sum (w b0)^2
    over chain axis 0: two latents, coupled, analytic gradient."""
    prec = np.linspace(0.5, 3.0, D).astype(np.float32)
    m = np.linspace(-1, 1, D).astype(np.float32)
    prec_t, m_t = torch.tensor(prec, device=dev), torch.tensor(m, device=dev)

    def log_joint(obs):
        w, b = obs['w'], obs['b']
        return (-0.5 * (prec_t * (w - m_t) ** 2).sum(-1) -
                0.5 * (b ** 2).sum(-1) -
                0.1 * ((w * b[:, :1]) ** 2).sum(-1))

    def grad(qs):
        w, b = qs
        gw = -prec * (w - m) - 0.2 * w * b[:, :1] ** 2
        gb = -b.copy()
        gb[:, 0] += -0.2 * (w ** 2).sum(-1) * b[:, 0]
        return [gw.astype(np.float32), gb.astype(np.float32)]
    return log_joint, grad


CASES = [
    ('SGLD', dict(learning_rate=0.01)),
    ('PSGLD', dict(learning_rate=0.01)),
    ('SGHMC', dict(learning_rate=0.01, friction=0.3, variance_estimate=0.05,
                   n_iter_resample_v=3, second_order=False)),
    ('SGHMC', dict(learning_rate=0.01, friction=0.3, variance_estimate=0.0,
                   n_iter_resample_v=4, second_order=True)),
    ('SGHMC', dict(learning_rate=0.01, n_iter_resample_v=None)),
    ('SGNHT', dict(learning_rate=0.01, variance_extra=0.1, tune_rate=1.0,
                   second_order=True, use_vector_alpha=True)),
    ('SGNHT', dict(learning_rate=0.01, variance_extra=0.1, tune_rate=0.5,
                   second_order=False, use_vector_alpha=True,
                   n_iter_resample_v=3)),
    ('SGNHT', dict(learning_rate=0.01, variance_extra=0.05, second_order=True,
                   use_vector_alpha=False)),
    ('SGNHT', dict(learning_rate=0.01, variance_extra=0.05, second_order=False,
                   use_vector_alpha=False, n_iter_resample_v=2)),
]


@pytest.mark.parametrize('name,kw', CASES)
def test_trajectory_matches_oracle(env, name, kw):
    zs, torch, dev = env
    C, D, Db = 37, 13, 3
    rng = np.random.RandomState(3)
    w0 = rng.normal(size=(C, D)).astype(np.float32)
    b0 = rng.normal(size=(C, Db)).astype(np.float32)
    log_joint, grad = _model(torch, dev, D)
    wt, bt = torch.tensor(w0, device=dev), torch.tensor(b0, device=dev)
    sampler = getattr(zs, name)(seed=42, **kw)
    op, info = sampler.sample(log_joint, {}, {'w': wt, 'b': bt})
    okw = dict(kw)
    if name == 'PSGLD':
        okw.update(decay=0.9, epsilon=1e-3)
    r = getattr(ref, name)(seed=42, **okw).sample(grad, [w0.copy(), b0.copy()])
    for it in range(7):
        rinfo = r.step()
        op.run()
        # fp32 element-wise chains of ~10 ops + hardware log/sin/cos normals
        np.testing.assert_allclose(wt.cpu().numpy(), r.qs[0], rtol=2e-5,
                                   atol=2e-5)
        np.testing.assert_allclose(bt.cpu().numpy(), r.qs[1], rtol=2e-5,
                                   atol=2e-5)
        if 'mean_k' in rinfo:
            for k, nm in enumerate(('w', 'b')):
                np.testing.assert_allclose(info.mean_k[nm].cpu().numpy(),
                                           rinfo['mean_k'][k], rtol=1e-4,
                                           atol=1e-7)
        if 'alpha' in rinfo:
            for k, nm in enumerate(('w', 'b')):
                np.testing.assert_allclose(info.alpha[nm].cpu().numpy(),
                                           rinfo['alpha'][k], rtol=1e-4,
                                           atol=1e-6)
    assert info.q['w'] is wt and sampler.t == 7


def test_learning_rate_placeholder_and_errors(env):
    zs, torch, dev = env
    lr = zs.placeholder(float)
    x = torch.zeros(5, 2, device=dev)
    s = zs.SGLD(learning_rate=lr, seed=1)
    op, _ = s.sample(lambda o: -(o['x'] ** 2).sum(-1), {}, {'x': x})
    with pytest.raises(ValueError, match='not fed'):
        op.run()
    op.run(feed_dict={lr: 0.0})          # lr = 0: nothing moves
    assert float(x.abs().max()) == 0.0
    op.run(feed_dict={lr: 0.01})
    assert float(x.abs().max()) > 0.0
    with pytest.raises(TypeError):
        zs.SGLD(0.1).sample(lambda o: 0, {}, {'x': np.zeros(3)})
    with pytest.raises(RuntimeError):
        s.sample(lambda o: 0, {}, {'x': x})


def test_minibatch_fed_per_run_reaches_the_model(env):
    """The reference's main SGMCMC idiom, sess.run(sample_op,
    feed_dict={x: xb, y: yb}) per mini-batch: data placeholders in `observed`
    (and read as `.value` inside the log-joint) must be bound by the run's
    feed_dict.  lr -> deterministic part of SGLD: q += lr/2 * grad, so with the
    noise subtracted two runs with different batches must move q by the
    gradient of THAT batch; a learning-rate placeholder without a default
    makes SGHMC draw its initial momentum at the first run's lr."""
    zs, torch, dev = env
    xb = zs.placeholder(torch.float32, name='xb')
    scale = zs.placeholder(torch.float32, name='scale')
    w = torch.zeros(7, 3, device=dev)

    def log_joint(obs):
        # quadratic pull of w towards the fed batch mean, scaled by a tensor
        # that is read through .value inside the model
        return -0.5 * scale.value * ((obs['w'] - obs['xb'].mean(0)) ** 2).sum(-1)

    s = zs.SGLD(learning_rate=0.2, seed=3)
    op, _ = s.sample(log_joint, {'xb': xb}, {'w': w})
    batches = [np.full((5, 3), 2.0, np.float32), np.full((4, 3), -1.0, np.float32)]
    moved = []
    for b, sc in zip(batches, (1.0, 3.0)):
        before = w.clone()
        op.run(feed_dict={xb: b, scale: np.float32(sc)})
        # expected drift lr/2 * grad = 0.1 * sc * (mean(b) - w_before)
        drift = 0.1 * sc * (torch.tensor(b, device=dev).mean(0) - before)
        moved.append((w - before - drift))
    # what is left is the N(0, lr) noise: zero-mean, std sqrt(0.2), and NOT
    # the several-sigma offset a stale / unbound batch would leave
    for r in moved:
        assert abs(float(r.mean())) < 0.45 and 0.2 < float(r.std()) < 0.8
    # SGHMC with an lr placeholder and no default: v0 ~ N(0, lr) at the first run
    lr = zs.placeholder(float)
    x = torch.zeros(4000, device=dev)
    h = zs.SGHMC(learning_rate=lr, friction=0.3, seed=5)
    oph, _ = h.sample(lambda o: -0.5 * o['x'] ** 2, {}, {'x': x})
    oph.run(feed_dict={lr: 0.04})
    v = h.vs[0]
    assert 0.1 < float(v.std()) < 0.4          # ~ sqrt(0.04) = 0.2, not 0


def _sample_error_with(zs, torch, dev, sampler, n_chains, n_iters, thinning=50):
    """tests/test_mcmc.py:14-50 (Fig. 1 of Chen et al.): double well
    2x^2 - x^4 with a noisy log-likelihood, KDE error of the pooled samples."""
    from scipy import stats
    burnin = n_iters * 2 // 3

    def log_joint(observed):
        x = observed['x']
        return 2 * x ** 2 - x ** 4 + 2 * torch.randn_like(x)

    x = torch.zeros(n_chains, device=dev)
    op, _ = sampler.sample(log_joint, {}, {'x': x})
    samples = []
    for t in range(n_iters):
        op.run()
        if t >= burnin and t % thinning == 0:
            xs = x.cpu().numpy().copy()
            assert not np.isnan(xs.sum())
            samples.append(xs)
    samples = np.array(samples).reshape(-1)
    A = 3
    xs = np.linspace(-A, A, 1000)
    pdfs = np.exp(2 * xs ** 2 - xs ** 4)
    pdfs = pdfs / pdfs.mean() / A / 2
    return np.abs(stats.gaussian_kde(samples)(xs) - pdfs).mean()


def test_sgld_reference_statistical_test(env):
    zs, torch, dev = env                       # tests/test_mcmc.py:67-72
    e = _sample_error_with(zs, torch, dev, zs.SGLD(learning_rate=0.01, seed=5),
                           n_chains=100, n_iters=8000)
    assert e <= 0.023


@pytest.mark.parametrize('second_order', [False, True])
def test_sghmc_reference_statistical_test(env, second_order):
    zs, torch, dev = env                       # tests/test_mcmc.py:74-88
    sampler = zs.SGHMC(learning_rate=0.01, n_iter_resample_v=50, friction=0.3,
                       variance_estimate=0.02, second_order=second_order,
                       seed=6)
    e = _sample_error_with(zs, torch, dev, sampler, n_chains=100, n_iters=8000)
    assert e <= 0.016


def test_reference_traces(env):
    """The device path against traces of the reference's OWN zhusuan/sgmcmc.py
    (oracle/make_golden_sgmcmc.py -> tests/golden/sgmcmc_reference_traces.npz)."""
    import os
    from test_oracle_sgmcmc_reference import CASES as REF_CASES, SEED
    zs, torch, dev = env
    tr = np.load(os.path.join(os.path.dirname(__file__), 'golden',
                              'sgmcmc_reference_traces.npz'))
    log_joint, _ = _model(torch, dev, tr['w0'].shape[1])
    for name, cls, kw in REF_CASES:
        wt = torch.tensor(tr['w0'], device=dev)
        bt = torch.tensor(tr['b0'], device=dev)
        sampler = getattr(zs, cls)(seed=SEED, **kw)
        op, info = sampler.sample(log_joint, {}, {'w': wt, 'b': bt})
        for i in range(tr[name + '/w'].shape[0]):
            op.run()
            np.testing.assert_allclose(wt.cpu().numpy(), tr[name + '/w'][i],
                                       rtol=3e-5, atol=3e-5,
                                       err_msg='%s w it %d' % (name, i))
            np.testing.assert_allclose(bt.cpu().numpy(), tr[name + '/b'][i],
                                       rtol=3e-5, atol=3e-5,
                                       err_msg='%s b it %d' % (name, i))
            for f in ('mean_k', 'alpha'):
                key = '%s/%s_w' % (name, f)
                if key in tr.files:
                    for nm in ('w', 'b'):
                        np.testing.assert_allclose(
                            getattr(info, f)[nm].cpu().numpy(),
                            tr['%s/%s_%s' % (name, f, nm)][i], rtol=2e-4,
                            atol=1e-6, err_msg='%s %s it %d' % (name, f, i))


def test_reference_bnn_traces(env):
    """The device path -- the model function of examples/bayesian_nn_sgmcmc.py
    (several latents of different shapes, group_ndims = 2 priors, deterministic
    node, user log-joint with mini-batch rescaling) under SGHMC / SGLD / SGNHT
    -- against traces of the reference's OWN sgmcmc.py sampling the
    reference's OWN build_bnn (oracle/make_golden_sgmcmc.py::main_bnn)."""
    import importlib.util
    import os
    from oracle.hmc_case_data import bnn_data
    from test_oracle_sgmcmc_reference import BNN_CASES, BNN_SEED
    zs, torch, dev = env
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(
        'bayesian_nn_sgmcmc', os.path.join(root, 'examples',
                                           'bayesian_nn_sgmcmc.py'))
    example = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(example)
    tr = np.load(os.path.join(root, 'tests', 'golden',
                              'sgmcmc_bnn_reference_traces.npz'))
    x, y, ws0, logstds, layer_sizes, n_train = bnn_data()
    names = ['w%d' % i for i in range(len(ws0))]
    x_in = zs.placeholder(torch.float32, name='x')
    x_in.feed(x, dev)
    y_d = torch.tensor(y, device=dev)
    for name, cls, kw in BNN_CASES:
        ws = [torch.tensor(w, device=dev) for w in ws0]
        model = example.make_model(
            x_in, layer_sizes, [torch.tensor(l, device=dev) for l in logstds],
            ws0[0].shape[0])
        model.log_joint = lambda bn: (sum(bn.cond_log_prob(names)) +
                                      bn.cond_log_prob('y').mean(1) * n_train)
        sampler = getattr(zs, cls)(seed=BNN_SEED, **kw)
        op, info = sampler.sample(model, {'y': y_d}, dict(zip(names, ws)))
        for i in range(tr[name + '/w0'].shape[0]):
            op.run()
            for k, nm in enumerate(names):
                np.testing.assert_allclose(
                    ws[k].cpu().numpy(), tr['%s/%s' % (name, nm)][i],
                    rtol=1e-4, atol=1e-4, err_msg='%s %s it %d' % (name, nm, i))
                for f in ('mean_k', 'alpha'):
                    key = '%s/%s_%s' % (name, f, nm)
                    if key in tr.files:
                        np.testing.assert_allclose(
                            getattr(info, f)[nm].cpu().numpy(), tr[key][i],
                            rtol=1e-3, atol=1e-6,
                            err_msg='%s %s %s it %d' % (name, f, nm, i))
