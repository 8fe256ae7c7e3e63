"""The device path (zhusuan_amd.HMC through the C-ABI: fused register and
ring kernels, generic plan, on-device adaptation) against traces produced by
the reference's OWN zhusuan/hmc.py (oracle/make_golden_hmc.py ->
tests/golden/hmc_reference_traces.npz), on the shared Philox stream."""
import os

import numpy as np
import pytest

from helpers_hmc_cases import cases, cases_r3

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    golden = os.path.join(os.path.dirname(__file__), 'golden')
    # round 3's traces (several latents feeding one dense likelihood, ragged
    # topic count: oracle/make_golden_hmc_r3.py) carry their own case names
    traces = {}
    for f in ('hmc_reference_traces.npz', 'hmc_reference_traces_r3.npz'):
        z = np.load(os.path.join(golden, f))
        traces.update({k: z[k] for k in z.files})
    return zs, torch, torch.device('cuda', 0), traces


# how the two model-family cases are written on the device side: the fused
# spelling on the native plan, the same spelling on the autograd-driven
# generic plan, the reference's literal dense expression (recognised
# symbolically, zhusuan_amd/_symbolic.py: native plan too), and a near miss
# of it (`latent * 1.0` first) that must fall back to the generic plan and
# still reproduce the traces
# `bf16x3`: the literal spelling on the native plan with the likelihood on
# the bf16 matrix cores (HMC(likelihood_arithmetic='bf16x3'),
# csrc/b3_kernel.h) -- the same traces at the same tolerances
VARIANTS = {'blr': ('native', 'generic', 'dense', 'nearmiss', 'bf16x3'),
            'lntm': ('native', 'generic', 'dense', 'nearmiss', 'bf16x3'),
            # the packed native plan: three latents in one row of 16 floats
            # (7 + 6 + 1, padded), topic rows of 6 padded to 8
            'blr_bias': ('dense', 'generic', 'nearmiss', 'bf16x3'),
            'lntm_k6': ('native', 'generic', 'dense', 'nearmiss', 'bf16x3'),
            # north_star's third likelihood: X @ w^T under a Categorical --
            # the reference's literal spelling and zs.linear_class_logits on
            # the native plan (fp32-MFMA, csrc/lb_ops.h), the generic plan,
            # a near miss, and the native plan on the bf16x3 kernel
            'softmax_reg': ('dense', 'native', 'generic', 'nearmiss',
                            'bf16x3'),
            'pmf': ('fused', 'dense', 'generic')}


def _build_blr(zs, torch, dev, case, qs, variant):
    X = torch.tensor(case['params']['X'], device=dev)
    y = torch.tensor(case['params']['y'], device=dev)        # int32
    C, D = qs['w'].shape

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        w = bn.normal('w', torch.zeros(D, device=dev), std=1., n_samples=C,
                      group_ndims=1)
        if variant == 'dense':
            logits = w.tensor @ X.t()            # the reference's spelling
        elif variant == 'nearmiss':
            logits = (w.tensor * 1.0) @ X.t()
        else:
            logits = zs.linear_logits(w.tensor, X)
        bn.bernoulli('y', logits, group_ndims=1)
        return bn
    plan = 'linear_bernoulli' if variant in ('native', 'dense') else 'generic'
    return model(), plan, {'y': y}


def _build_blr_bias(zs, torch, dev, case, qs, variant):
    """oracle/make_golden_hmc_r3.py::blr_bias_model in the reference's literal
    spelling: matmul(u, X1^T) + matmul(v, X2^T) + expand_dims(b, 1)."""
    X1 = torch.tensor(case['params']['X1'], device=dev)
    X2 = torch.tensor(case['params']['X2'], device=dev)
    y = torch.tensor(case['params']['y'], device=dev)        # int32
    C = qs['u'].shape[0]

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        u = bn.normal('u', torch.zeros(X1.shape[1], device=dev), std=1.,
                      n_samples=C, group_ndims=1)
        v = bn.normal('v', torch.zeros(X2.shape[1], device=dev), std=0.5,
                      n_samples=C, group_ndims=1)
        b = bn.normal('b', torch.zeros((), device=dev), std=2., n_samples=C)
        ut = u.tensor * 1.0 if variant == 'nearmiss' else u.tensor
        logits = ut @ X1.t() + v.tensor @ X2.t() + b.tensor.unsqueeze(1)
        bn.bernoulli('y', logits, group_ndims=1)
        return bn
    plan = 'linear_bernoulli' if variant == 'dense' else 'generic'
    return model(), plan, {'y': y}


def _build_lntm(zs, torch, dev, case, qs, variant):
    """examples/topic_models/lntm_mcem.py:31-48 and its E-step objective
    (:97-102), beta observed as there (:104-106)."""
    p = {k: torch.tensor(v, device=dev) for k, v in case['params'].items()}
    n_chains, n_docs, K = qs['eta'].shape
    V = p['x'].shape[1]

    @zs.meta_bayesian_net(scope='lntm')
    def lntm():
        bn = zs.BayesianNet()
        eta = bn.normal('eta', p['eta_mean'].unsqueeze(0).repeat(n_docs, 1),
                        logstd=p['eta_logstd'], n_samples=n_chains,
                        group_ndims=1)
        theta = torch.softmax(eta.tensor * 1.0 if variant == 'nearmiss'
                              else eta.tensor, -1)
        beta = bn.normal('beta', torch.zeros(K, V, device=dev), logstd=10.0,
                         group_ndims=1)
        phi = torch.softmax(beta.tensor, -1)
        if variant in ('dense', 'nearmiss'):     # lntm_mcem.py:39-46
            logits = torch.log((theta.reshape(-1, K) @ phi).reshape(
                n_chains, n_docs, V))
        else:
            logits = zs.log_mixture(theta, phi)
        bn.unnormalized_multinomial('x', logits, normalize_logits=False,
                                    dtype=torch.float32)
        return bn
    model = lntm()
    model.log_joint = lambda bn: (bn.cond_log_prob('eta') +
                                  bn.cond_log_prob('x'))
    plan = 'mixture_multinomial' if variant in ('native', 'dense') \
        else 'generic'
    return model, plan, {'x': p['x'], 'beta': p['beta']}


def _build_softmax_regression(zs, torch, dev, case, qs, variant):
    X = torch.tensor(case['params']['X'], device=dev)
    y = torch.tensor(case['params']['y'], device=dev)        # int32 labels
    C, K, F = qs['w'].shape

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        w = bn.normal('w', torch.zeros(K, F, device=dev), std=1., n_samples=C,
                      group_ndims=2)
        if variant == 'native':
            logits = zs.linear_class_logits(w.tensor, X)
        elif variant == 'nearmiss':
            logits = X.unsqueeze(0) @ (w.tensor * 1.0).transpose(-1, -2)
        else:     # tf.matmul(Xc, w, transpose_b=True): [C, N, K]
            logits = X.unsqueeze(0) @ w.tensor.transpose(-1, -2)
        bn.categorical('y', logits, group_ndims=1)
        return bn
    plan = 'linear_categorical' if variant in ('dense', 'native') \
        else 'generic'
    return model(), plan, {'y': y}


def _build_pmf(zs, torch, dev, case, qs, variant):
    """pmf_hmc.py:19-31 and its log-joint (:135-141); `fused` spells the
    logits zs.gathered_dot, `dense` the reference's two gathers."""
    from oracle.hmc_case_data import pmf_data
    alpha_u, alpha_v, alpha_pred = pmf_data()[5]
    p = {k: torch.tensor(v, device=dev) for k, v in case['params'].items()}
    K, n, D = qs['u'].shape
    m = p['v'].shape[1]
    su, sv = p['su'].long(), p['sv'].long()

    @zs.meta_bayesian_net(scope='pmf', reuse_variables=True)
    def pmf():
        bn = zs.BayesianNet()
        u = bn.normal('u', torch.zeros(n, D, device=dev), std=alpha_u,
                      n_samples=K, group_ndims=1)
        v = bn.normal('v', torch.zeros(m, D, device=dev), std=alpha_v,
                      n_samples=K, group_ndims=1)
        if variant == 'dense':
            r_logits = (u.tensor[:, su] * v.tensor[:, sv]).sum(2)
        else:
            r_logits = zs.gathered_dot(u, p['su'], v, p['sv'])
        bn.deterministic('r_pred', torch.sigmoid(r_logits))
        bn.normal('r', torch.sigmoid(r_logits), std=alpha_pred)
        return bn
    model = pmf()

    def log_joint(bn):
        log_pu, log_pv = bn.cond_log_prob(['u', 'v'])
        return log_pu.sum(-1) + log_pv.sum(-1) + bn.cond_log_prob('r').sum(-1)
    model.log_joint = log_joint
    # both spellings on the native gathered-dot plan (csrc/gather_dot.hip +
    # csrc/hmc_model_seg.hip); `generic`: the fused spelling on autograd
    return model, 'generic' if variant == 'generic' else 'gathered_dot', \
        {'r': p['r'], 'v': p['v']}


def _build(zs, torch, dev, case, qs, variant=None):
    """The same model through the product's own front-end:
    (model, expected plan, observed)."""
    name = case['name']
    if name == 'blr':
        return _build_blr(zs, torch, dev, case, qs, variant)
    if name == 'blr_bias':
        return _build_blr_bias(zs, torch, dev, case, qs, variant)
    if name in ('lntm', 'lntm_k6'):
        return _build_lntm(zs, torch, dev, case, qs, variant)
    if name == 'pmf':
        return _build_pmf(zs, torch, dev, case, qs, variant)
    if name == 'softmax_reg':
        return _build_softmax_regression(zs, torch, dev, case, qs, variant)
    return _build_plain(zs, torch, dev, case, qs) + ({},)


def _build_plain(zs, torch, dev, case, qs):
    name = case['name']
    if name.startswith('gauss'):
        mean = torch.tensor(case['params']['mean'], device=dev)
        C = qs['x'].shape[0]
        if 'std' in case['params']:      # gaussian.py:15-20, `std=` path
            spread = dict(std=torch.tensor(case['params']['std'], device=dev))
        else:
            spread = dict(logstd=torch.tensor(case['params']['logstd'],
                                              device=dev))

        @zs.meta_bayesian_net()
        def model():
            bn = zs.BayesianNet()
            bn.normal('x', mean, n_samples=C, group_ndims=1, **spread)
            return bn
        return model(), 'fused_diag_normal'
    px = torch.tensor(case['params']['prec_x'], device=dev)

    def log_joint(obs):
        x, y = obs['x'], obs['y']
        sx, sy = x.sum(-1), y.sum(-1)
        return (-0.5 * (px * x ** 2).sum(-1) - 0.5 * (y ** 2).sum(-1)
                - 0.01 * sx ** 2 * sy ** 2)
    return log_joint, 'generic'


def _case_variants():
    for c in list(cases()) + list(cases_r3()):
        for v in VARIANTS.get(c['name'], (None,)):
            yield pytest.param(c, v, id=c['name'] + ('' if v is None
                                                     else '-' + v))


@pytest.mark.parametrize('case,variant', list(_case_variants()))
def test_device_reproduces_reference_hmc_traces(env, case, variant,
                                                monkeypatch):
    zs, torch, dev, traces = env
    name = case['name']
    qs = {k: torch.tensor(traces['%s/q0_%s' % (name, k)], device=dev)
          for k in case['latent_names']}
    kw = dict(case['hmc_kwargs'])
    ph_ss = ph_m = None
    if kw.get('adapt_step_size') is True and case['flags'](0)[0] is not None \
            and name != 'gauss_ss':
        ph_ss = kw['adapt_step_size'] = zs.placeholder(bool)
    if kw.get('adapt_mass') is True:
        ph_m = kw['adapt_mass'] = zs.placeholder(bool)
    if variant == 'generic':
        kw['native_plans'] = False
    if variant == 'bf16x3':
        kw['likelihood_arithmetic'] = 'bf16x3'
        variant = 'dense'
        # (the traces' chain axes are a few chains per document: take the
        # kernels whatever share of their 128-chain workgroups that fills)
        monkeypatch.setattr(zs._ops, 'BF16X3_REQUIRE_FILL', False)
    hmc = zs.HMC(seed=case['seed'], **kw)
    model, plan, observed = _build(zs, torch, dev, case, qs, variant)
    op, info = hmc.sample(model, observed, qs)
    assert hmc.plan_kind == plan
    if 'likelihood_arithmetic' in kw:
        assert hmc.likelihood_arithmetic_used == 'bf16x3'
    n_flip = n_total = 0
    for i in range(case['n_iters']):
        f_ss, f_m = case['flags'](i)
        feed = {}
        if ph_ss is not None:
            feed[ph_ss] = bool(f_ss)
        if ph_m is not None:
            feed[ph_m] = bool(f_m)
        op.run(feed_dict=feed)
        acc_ref = traces[name + '/acceptance_rate'][i]
        # energies are O(10..600): hardware log/sin/cos normals + fp32 sums
        for f in ('orig_hamiltonian', 'hamiltonian', 'orig_log_prob'):
            want = traces['%s/%s' % (name, f)][i]
            np.testing.assert_allclose(getattr(info, f).cpu().numpy(), want,
                                       rtol=1e-4, atol=2e-3,
                                       err_msg='%s it %d' % (f, i))
        np.testing.assert_allclose(info.acceptance_rate.cpu().numpy(), acc_ref,
                                   atol=3e-3)
        np.testing.assert_allclose(float(info.updated_step_size.item()),
                                   float(traces[name + '/updated_step_size'][i]),
                                   rtol=2e-3)
        # accept decisions: identical except for borderline |u - acc|
        for k in case['latent_names']:
            want = traces['%s/q_%s' % (name, k)][i]
            got = qs[k].cpu().numpy()
            lead = want.shape[:acc_ref.ndim]
            diff = np.abs(got - want).reshape(int(np.prod(lead)), -1).max(1)
            bad = diff > 2e-3 * (1.0 + np.abs(want).max())
            n_flip += int(bad.sum())
            n_total += bad.size
            # continue from the reference's state (teacher forcing)
            qs[k].copy_(torch.tensor(want, device=dev))
    assert n_flip <= max(1, n_total // 200), (n_flip, n_total)
    assert hmc.t == int(traces[name + '/t'])
