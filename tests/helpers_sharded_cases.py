"""Problems for the two-rank tests of the non-fused plans: the topic-model
E step (BASELINE configs[4] family: chain axes [n_chains, n_docs], mass
adaptation on) and Bayesian logistic regression, each buildable for a slice
[lo, hi) of the leading chain axis -- what one rank of a sharded run owns."""
import numpy as np

N_ITERS = 9
MASS_COLLECT = 3


def lntm_problem():
    rng = np.random.RandomState(31)
    n_chains, n_docs, K, V = 10, 6, 8, 23
    return dict(
        beta=rng.normal(size=(K, V)).astype(np.float32),
        x=rng.poisson(1.5, size=(n_docs, V)).astype(np.float32),
        eta_mean=(0.3 * rng.normal(size=(n_docs, K))).astype(np.float32),
        eta_logstd=(0.2 * rng.normal(size=K)).astype(np.float32),
        q0=(0.5 * rng.normal(size=(n_chains, n_docs, K))).astype(np.float32))


def blr_problem():
    rng = np.random.RandomState(32)
    n_rows, D, n_chains = 50, 12, 14
    X = rng.normal(size=(n_rows, D)).astype(np.float32)
    w = rng.normal(size=D).astype(np.float32)
    y = (rng.uniform(size=n_rows) < 1 / (1 + np.exp(-X @ w))).astype(np.float32)
    return dict(X=X, y=y,
                q0=(0.1 * rng.normal(size=(n_chains, D))).astype(np.float32))


def blrb_problem():
    """Weights (13: not a multiple of 4) + a per-chain scalar bias, written
    with the literal spelling: the packed state of the native plan."""
    rng = np.random.RandomState(33)
    n_rows, D, n_chains = 60, 13, 14
    X = rng.normal(size=(n_rows, D)).astype(np.float32)
    w = rng.normal(size=D).astype(np.float32)
    y = (rng.uniform(size=n_rows) < 1 / (1 + np.exp(-X @ w - 0.5))
         ).astype(np.float32)
    return dict(X=X, y=y,
                q0=(0.1 * rng.normal(size=(n_chains, D))).astype(np.float32),
                b0=(0.1 * rng.normal(size=n_chains)).astype(np.float32))


def build(zs, torch, dev, family, lo, hi, adapt, sharding, native, seed=21):
    """(hmc, sample_op, info, latent, flag placeholders) of rows [lo, hi)."""
    if family == 'lntm':
        p = lntm_problem()
        t = {k: torch.tensor(v, device=dev) for k, v in p.items()}
        n = hi - lo
        phi = torch.softmax(t['beta'], -1)

        @zs.meta_bayesian_net(scope='lntm')
        def model():
            bn = zs.BayesianNet()
            eta = bn.normal('eta', t['eta_mean'], logstd=t['eta_logstd'],
                            n_samples=n, group_ndims=1)
            bn.unnormalized_multinomial(
                'x', zs.log_mixture(torch.softmax(eta.tensor, -1), phi),
                normalize_logits=False, dtype=torch.float32)
            return bn
        m = model()
        m.log_joint = lambda bn: (bn.cond_log_prob('eta') +
                                  bn.cond_log_prob('x'))
        observed = {'x': t['x']}
        name, plan = 'eta', 'mixture_multinomial'
        kw = dict(step_size=5e-3, n_leapfrogs=5, target_acceptance_rate=0.6)
    elif family == 'blrb':
        p = blrb_problem()
        t = {k: torch.tensor(v, device=dev) for k, v in p.items()}
        n, D = hi - lo, p['q0'].shape[1]

        @zs.meta_bayesian_net()
        def model():
            bn = zs.BayesianNet()
            w = bn.normal('w', torch.zeros(D, device=dev), std=1.,
                          n_samples=n, group_ndims=1)
            b = bn.normal('b', torch.zeros((), device=dev), std=2.,
                          n_samples=n)
            bn.bernoulli('y', w.tensor @ t['X'].t() + b.tensor[:, None],
                         group_ndims=1, dtype=torch.float32)
            return bn
        m = model()
        observed = {'y': t['y']}
        name, plan = 'w', 'linear_bernoulli'
        kw = dict(step_size=0.02, n_leapfrogs=6, target_acceptance_rate=0.8)
    else:
        p = blr_problem()
        t = {k: torch.tensor(v, device=dev) for k, v in p.items()}
        n, D = hi - lo, p['q0'].shape[1]

        @zs.meta_bayesian_net()
        def model():
            bn = zs.BayesianNet()
            w = bn.normal('w', torch.zeros(D, device=dev), std=1.,
                          n_samples=n, group_ndims=1)
            bn.bernoulli('y', zs.linear_logits(w.tensor, t['X']),
                         group_ndims=1, dtype=torch.float32)
            return bn
        m = model()
        observed = {'y': t['y']}
        name, plan = 'w', 'linear_bernoulli'
        kw = dict(step_size=0.02, n_leapfrogs=6, target_acceptance_rate=0.8)
    q = t['q0'][lo:hi].clone().contiguous()
    latent = {name: q}
    if family == 'blrb':
        latent['b'] = t['b0'][lo:hi].clone().contiguous()
    flags = None
    if adapt:
        flags = (zs.placeholder(bool), zs.placeholder(bool))
        kw.update(adapt_step_size=flags[0], adapt_mass=flags[1],
                  mass_collect_iters=MASS_COLLECT)
    hmc = zs.HMC(seed=seed, sharding=sharding, native_plans=native, **kw)
    op, info = hmc.sample(m, observed, latent)
    assert hmc.plan_kind == (plan if native else 'generic'), hmc.plan_kind
    if len(latent) > 1:     # reported side by side: [n, D + 1]
        q = [latent['w'], latent['b']]
    return hmc, op, info, q, flags


def schedule(i):
    """(adapt_step_size, adapt_mass) fed at iteration i."""
    return i < 7, i < 6


def run(zs, torch, dev, family, lo, hi, adapt, sharding, native,
        rank0_reads=False, rank=0):
    hmc, op, info, q, flags = build(zs, torch, dev, family, lo, hi, adapt,
                                    sharding, native)
    eps = []
    for i in range(N_ITERS):
        feed = {}
        if flags is not None:
            feed = dict(zip(flags, schedule(i)))
        op.run(feed_dict=feed, sync=(i % 4 == 3))
        if rank0_reads and rank == 0:
            # one rank alone reads the adapted step size and snapshots the
            # sampler: neither may communicate
            eps.append(float(info.updated_step_size.item()))
            hmc.get_state()
    hmc.check_numerics()
    if isinstance(q, list):
        q = torch.cat([v.reshape(v.shape[0], -1) for v in q], 1)
    out = dict(q=q.cpu().numpy(),
               acc=info.acceptance_rate.cpu().numpy(),
               step_size=float(info.updated_step_size.item()),
               state=hmc.get_state()['state'].numpy())
    if adapt:
        out['mass'] = torch.cat(list(hmc._plan.mass)).cpu().numpy()
    if eps:
        out['eps'] = np.array(eps)
    return out
