"""GPU parity: fused diag-Normal HMC transition (csrc/hmc_fused_normal.hip)
through the C-ABI vs the NumPy oracle on identical Philox counters.

Tolerances (float32 path; north_star asks acceptance/ESS within 1 %): per
chain energies to ~2e-5*|H|, acceptance to ~1e-4*|H|, accepted states to
2e-5*|q|max; see tests/helpers.py:compare_transition."""
import numpy as np
import pytest

from helpers import (FusedKernel, compare_transition, gpu_sampler,
                     make_diag_problem, ref_sampler)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    return zs, torch


# (C, D, L): every (G, NCH) dispatch class, vectorised and ragged rows,
# wave tails (C not a multiple of chains-per-wave), D = 1
SHAPES = [
    (37, 1, 3), (1000, 3, 4), (1000, 10, 5), (129, 16, 2), (200, 48, 3),
    (77, 64, 10), (64, 100, 4), (33, 250, 3), (50, 256, 5), (19, 257, 3),
    (40, 512, 10), (24, 700, 2), (16, 1000, 3), (64, 1024, 10),
    (9, 1030, 2), (12, 1536, 2), (8, 2048, 3), (1, 1024, 1), (5, 7, 0),
]


@pytest.mark.parametrize('C,D,L', SHAPES)
def test_single_transition_matches_oracle(env, C, D, L):
    zs, torch = env
    mean, logstd, q0 = make_diag_problem(C, D, seed=C + D)
    kw = dict(step_size=0.6 / max(1.0, D ** 0.25), n_leapfrogs=L, seed=99)
    ref, xr = ref_sampler(mean, logstd, q0, **kw)
    hmc, op, info, xg = gpu_sampler(zs, torch, mean, logstd, q0, **kw)
    assert hmc.plan_kind == 'fused_diag_normal'
    for it in range(2):
        rinfo = ref.step()
        op.run()
        compare_transition(info, xg, rinfo, xr, ref)
        # resynchronise borderline chains so iteration 2 starts identical
        xg.copy_(torch.tensor(xr, device=xg.device))


def test_multi_axis_chains(env):
    """chain shape [n_chains, n_docs], data shape [K] (lntm_mcem.py shape)."""
    zs, torch = env
    rng = np.random.RandomState(3)
    A, B, K = 6, 11, 20
    mean = rng.normal(size=K).astype(np.float32)
    logstd = rng.uniform(-0.5, 0.5, size=K).astype(np.float32)
    q0 = rng.normal(size=(A, B, K)).astype(np.float32)
    kw = dict(step_size=0.2, n_leapfrogs=4, seed=5)
    ref, xr = ref_sampler(mean, logstd, q0.reshape(A * B, K), **kw)
    hmc, op, info, xg = gpu_sampler(zs, torch, mean, logstd, q0,
                                    group_ndims=1, **kw)
    assert tuple(info.acceptance_rate.shape) == (A, B)
    rinfo = ref.step()
    op.run()
    compare_transition(info, xg, rinfo, xr, ref)


def test_data_matrix_axes(env):
    """data shape [4, 8] with group_ndims=2 flattens to D=32."""
    zs, torch = env
    rng = np.random.RandomState(4)
    C = 50
    mean = rng.normal(size=(4, 8)).astype(np.float32)
    logstd = rng.uniform(-0.5, 0.5, size=(8,)).astype(np.float32)
    q0 = rng.normal(size=(C, 4, 8)).astype(np.float32)
    full_ls = np.broadcast_to(logstd, (4, 8)).reshape(-1).copy()
    kw = dict(step_size=0.2, n_leapfrogs=3, seed=6)
    ref, xr = ref_sampler(mean.reshape(-1), full_ls, q0.reshape(C, 32), **kw)
    hmc, op, info, xg = gpu_sampler(zs, torch, mean, logstd, q0, **kw)
    assert hmc.plan_kind == 'fused_diag_normal'
    rinfo = ref.step()
    op.run()
    compare_transition(info, xg, rinfo, xr, ref)


def test_deterministic_and_shard_invariant(env):
    """Same seed -> bit-identical; running two half shards with their global
    chain offsets reproduces the full run bit for bit (RNG is keyed by the
    global chain index)."""
    zs, torch = env
    C, D, L = 512, 96, 6
    mean, logstd, q0 = make_diag_problem(C, D, seed=1)
    dev = torch.device('cuda', 0)

    mean_t = torch.tensor(mean, device=dev)
    logstd_t = torch.tensor(logstd, device=dev)

    def run(q_np, offset):
        q = torch.tensor(q_np, device=dev)
        k = FusedKernel(torch, q.shape[0], D, dev)
        k.step(q, mean_t, logstd_t, None, 0.15, L, 777, 3, chain_offset=offset)
        torch.cuda.synchronize()
        return (q.cpu().numpy(), k.info[0].cpu().numpy(),
                float(k.stats[0].item()))

    qa, acca, sa = run(q0, 0)
    qb, accb, sb = run(q0, 0)
    np.testing.assert_array_equal(qa, qb)
    np.testing.assert_array_equal(acca, accb)
    h = C // 2 + 3
    q1, acc1, s1 = run(q0[:h], 0)
    q2, acc2, s2 = run(q0[h:], h)
    np.testing.assert_array_equal(np.concatenate([q1, q2]), qa)
    np.testing.assert_array_equal(np.concatenate([acc1, acc2]), acca)
    assert abs((s1 + s2) - sa) < 1e-9 * C
    assert abs(sa - acca.astype(np.float64).sum()) < 1e-6 * C


def test_adaptation_trace_config1(env):
    """BASELINE config 1 (gaussian.py with n_x=10): step-size and mass
    adaptation on for i < 50.  The dual-averaging / EWMV state lives on the
    device; its trace must follow the oracle's (Appendix B #1 trace shape:
    eps jumps to ~1.0 after the first adapted iteration)."""
    zs, torch = env
    n_x, C = 10, 1000
    stdev = (1 / (np.arange(n_x, dtype=np.float32) + 1)).astype(np.float32)
    mean = np.zeros(n_x, np.float32)
    logstd = np.log(stdev)
    q0 = np.zeros((C, n_x), np.float32)
    kw = dict(step_size=1e-3, n_leapfrogs=5, adapt_step_size=True,
              adapt_mass=True, target_acceptance_rate=0.9, seed=1)
    ref, xr = ref_sampler(mean, logstd, q0, **kw)
    a_ss, a_m = zs.placeholder(bool), zs.placeholder(bool)
    kw_g = dict(kw, adapt_step_size=a_ss, adapt_mass=a_m)
    hmc, op, info, xg = gpu_sampler(zs, torch, mean, logstd, q0, **kw_g)
    eps_g, eps_r, acc_g, acc_r = [], [], [], []
    samples = []
    for i in range(120):
        rinfo = ref.step(adapt_step_size=i < 50, adapt_mass=i < 50)
        op.run(feed_dict={a_ss: i < 50, a_m: i < 50})
        eps_g.append(float(info.updated_step_size.item()))
        eps_r.append(float(rinfo.updated_step_size))
        acc_g.append(float(info.acceptance_rate.mean().item()))
        acc_r.append(float(rinfo.acceptance_rate.mean()))
        if i >= 60:
            samples.append(xg.cpu().numpy().copy())
    eps_g, eps_r = np.array(eps_g), np.array(eps_r)
    # first iterations are deterministic given identical RNG: tight
    np.testing.assert_allclose(eps_g[:12], eps_r[:12], rtol=2e-3)
    # later, rare accept flips decorrelate chains slightly: 1 % (north_star)
    np.testing.assert_allclose(eps_g, eps_r, rtol=1e-2)
    np.testing.assert_allclose(acc_g, acc_r, atol=1e-2)
    assert abs(eps_g[0] - 1.0) < 0.05            # Appendix B #1
    assert hmc.n_init_trips >= 2
    s = np.vstack(samples)
    assert np.abs(s.mean(0)).max() < 0.02
    np.testing.assert_allclose(s.std(0), stdev, rtol=0.03)


def test_mass_state_matches_oracle(env):
    zs, torch = env
    C, D = 400, 24
    mean, logstd, q0 = make_diag_problem(C, D, seed=8, q_scale=0.3)
    kw = dict(step_size=0.05, n_leapfrogs=4, adapt_step_size=True,
              adapt_mass=True, mass_collect_iters=4, seed=2)
    ref, xr = ref_sampler(mean, logstd, q0, **kw)
    hmc, op, info, xg = gpu_sampler(zs, torch, mean, logstd, q0, **kw)
    for i in range(8):
        ref.step()
        op.run()
        xg.copy_(torch.tensor(xr, device=xg.device))   # keep states aligned
        np.testing.assert_allclose(
            hmc._plan.ewmv_mean[0].cpu().numpy(),
            ref.ewmv.mean[0].reshape(-1), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(
            hmc._plan.ewmv_var[0].cpu().numpy(),
            ref.ewmv.var[0].reshape(-1), rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(
            hmc._plan.mass[0].cpu().numpy(),
            np.asarray(ref.last_mass[0]).reshape(-1), rtol=3e-4)
        np.testing.assert_allclose(float(info.updated_step_size.item()),
                                   float(ref.step_size), rtol=5e-3)


def test_nonfinite_handling(env):
    zs, torch = env
    C, D = 64, 16
    mean, logstd, q0 = make_diag_problem(C, D, seed=9)
    # a non-finite CURRENT log-prob is an error (hmc.py:51-53)
    q_bad = q0.copy()
    q_bad[5, 3] = np.inf
    hmc, op, info, xg = gpu_sampler(zs, torch, mean, logstd, q_bad,
                                    step_size=0.1, n_leapfrogs=2, seed=1)
    with pytest.raises(zs.InvalidArgumentError,
                       match='old_log_prob has numeric errors'):
        op.run()
    # a non-finite PROPOSAL is not: acceptance 0, state kept (hmc.py:56-59)
    hmc, op, info, xg = gpu_sampler(zs, torch, mean, logstd, q0,
                                    step_size=1e20, n_leapfrogs=3, seed=1)
    op.run()
    acc = info.acceptance_rate.cpu().numpy()
    assert np.all(acc == 0.0)
    np.testing.assert_array_equal(xg.cpu().numpy(), q0)
    np.testing.assert_array_equal(info.log_prob.cpu().numpy(),
                                  info.orig_log_prob.cpu().numpy())


def test_init_momentum_regenerated(env):
    """HMCInfo.init_momentum equals the oracle's p0 (lazy regeneration)."""
    zs, torch = env
    C, D = 32, 40
    mean, logstd, q0 = make_diag_problem(C, D, seed=11)
    kw = dict(step_size=0.1, n_leapfrogs=2, seed=4)
    ref, xr = ref_sampler(mean, logstd, q0, **kw)
    hmc, op, info, xg = gpu_sampler(zs, torch, mean, logstd, q0, **kw)
    rinfo = ref.step()
    op.run()
    np.testing.assert_allclose(info.init_momentum['x'].cpu().numpy(),
                               rinfo.init_momentum[0], atol=5e-6, rtol=1e-5)


def test_full_size_properties(env):
    """BASELINE config 2 size (65 536 x 1 024, L=10): size-independent
    properties.  (1) determinism; (2) energy conservation: tiny step ->
    acceptance ~ 1 and |dH| small; (3) detailed balance proxy: stationary
    start keeps per-dimension variance = std^2 after several transitions;
    (4) shard additivity of acc_sum."""
    zs, torch = env
    C, D, L = 65536, 1024, 10
    dev = torch.device('cuda', 0)
    logstd = np.linspace(-1, 1, D).astype(np.float32)
    mean = np.zeros(D, np.float32)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    std_t = torch.tensor(np.exp(logstd), device=dev)
    x0 = torch.randn(C, D, device=dev, generator=g) * std_t   # stationary

    def build(seed, eps):
        @zs.meta_bayesian_net()
        def model():
            bn = zs.BayesianNet()
            bn.normal('x', torch.tensor(mean, device=dev),
                      logstd=torch.tensor(logstd, device=dev), n_samples=C,
                      group_ndims=1)
            return bn
        x = x0.clone()
        hmc = zs.HMC(step_size=eps, n_leapfrogs=L, seed=seed)
        op, info = hmc.sample(model(), {}, {'x': x})
        return hmc, op, info, x

    h1, op1, i1, x1 = build(7, 0.08)
    h2, op2, i2, x2 = build(7, 0.08)
    for _ in range(5):
        op1.run(sync=False)
        op2.run(sync=False)
    h1.check_numerics()
    assert torch.equal(x1, x2)                                   # (1)
    acc = float(i1.acceptance_rate.mean().item())
    assert 0.3 < acc < 1.0
    var = x1.var(dim=0).cpu().numpy()
    np.testing.assert_allclose(var, np.exp(2 * logstd), rtol=0.03)  # (3)
    h3, op3, i3, x3 = build(9, 1e-3)
    op3.run()
    dH = (i3.hamiltonian - i3.orig_hamiltonian).abs().max().item()
    assert dH < 0.05 and float(i3.acceptance_rate.min().item()) > 0.95  # (2)


def test_ring_kernel_staged_and_unstaged_paths_agree():
    """The ring kernel keeps the per-chain scalars (MH uniform in, HMCInfo out)
    in LDS when a workgroup's share fits and falls back to scalar-unit
    uniforms + in-loop stores otherwise (> ~300k chains at D = 1024).  Chains
    are independent and the random stream is keyed by the global chain index:
    the first 3 001 chains of a 310 000-chain launch (the fallback) must come
    out bit-identical to a 3 001-chain launch (staged) from the same rows --
    states and all five HMCInfo vectors, over two transitions."""
    import torch
    from helpers import FusedKernel
    from zhusuan_amd import _capi
    dev = torch.device('cuda', 0)
    D, C_small, C_big = 1024, 3001, 310000
    g = torch.Generator(device='cpu').manual_seed(4)
    logstd = torch.linspace(-1, 1, D).to(dev)
    mean = torch.randn(D, generator=g).to(dev)
    rows = (torch.randn(4096, D, generator=g) * torch.exp(logstd.cpu()) +
            mean.cpu()).to(dev)
    q_big = rows.repeat((C_big + 4095) // 4096, 1)[:C_big].contiguous()
    q_small = q_big[:C_small].clone()
    name = _capi.load().zshmc_fused_kernel_name(D, 0, 0).decode()
    assert 'ring' in name
    kb = FusedKernel(torch, C_big, D, dev)
    ks = FusedKernel(torch, C_small, D, dev)
    for t in range(2):
        kb.step(q_big, mean, logstd, None, 0.14, 5, 99, t, chain_offset=7)
        ks.step(q_small, mean, logstd, None, 0.14, 5, 99, t, chain_offset=7)
        assert torch.equal(q_big[:C_small], q_small)
        for a, b in zip(kb.info, ks.info):
            assert torch.equal(a[:C_small], b)
    assert float(ks.info[0].min()) < 1.0 and float(ks.info[0].mean()) > 0.3


def test_steady_nonadaptive_phase_is_a_pure_elision(monkeypatch):
    """With the adapt flag off the reference re-assigns
    step_size <- exp(log_epsilon_bar) every iteration (hmc.py:108-110).  After
    two such updates the sampler state is at its fixed point and the front-end
    stops launching the update kernel / collecting the mean acceptance: states,
    HMCInfo and step size must be bit-identical to a run that never elides,
    also across a switch back to adaptation and across set_state."""
    import torch
    import zhusuan_amd as zs
    from zhusuan_amd import hmc as hmc_mod
    dev = torch.device('cuda', 0)
    C, D = 3000, 512
    logstd = torch.linspace(-1, 1, D, device=dev)

    def run(elide):
        monkeypatch.setattr(hmc_mod._FusedDiagNormalPlan, 'can_skip_acc', elide)

        @zs.meta_bayesian_net()
        def model():
            bn = zs.BayesianNet()
            bn.normal('x', torch.zeros(D, device=dev), logstd=logstd,
                      n_samples=C, group_ndims=1)
            return bn
        flag = zs.placeholder(bool)
        h = zs.HMC(step_size=0.05, n_leapfrogs=4, adapt_step_size=flag,
                   seed=3)
        x = torch.zeros(C, D, device=dev)
        op, info = h.sample(model(), {}, {'x': x})
        assert h.plan_kind == 'fused_diag_normal'
        out = []
        launches = []
        schedule = [True] * 6 + [False] * 7 + [True] * 3 + [False] * 4
        for i, a in enumerate(schedule):
            if i == 18:                       # checkpoint / resume mid-way
                h.set_state(h.get_state())
            op.run(feed_dict={flag: a})
            launches.append(h._plan.collect_acc)
            out.append((x.clone(), info.acceptance_rate.clone(),
                        info.log_prob.clone(),
                        float(info.updated_step_size)))
        return out, launches

    ref, l_ref = run(False)
    got, l_got = run(True)
    assert all(l_ref)
    # elided exactly from the third consecutive non-adaptive iteration on
    assert l_got == [True] * 8 + [False] * 5 + [True] * 5 + [False] * 0 + \
        [True] * 2 or l_got.count(False) >= 5
    assert l_got[:8] == [True] * 8 and l_got[8:13] == [False] * 5
    assert l_got[13:18] == [True] * 5            # adaptation back on, then 2 updates
    for (xa, aa, la, sa), (xb, ab, lb, sb) in zip(ref, got):
        assert torch.equal(xa, xb) and torch.equal(aa, ab) and torch.equal(la, lb)
        assert sa == sb


def test_fused_plan_follows_fed_and_updated_parameters(env):
    """A single-Normal model whose parameters are read from fed placeholders
    (what lntm_mcem.py:164-169 does with eta_mean / eta_logstd) or updated in
    place between runs must keep sampling the CURRENT target: the fused plan
    re-resolves the parameters every run, exactly as the generic plan
    re-evaluates the model function (SURVEY section 7 "no stale log-prob
    caching")."""
    zs, torch = env
    dev = torch.device('cuda', 0)
    C, D = 300, 260
    rng = np.random.RandomState(0)
    means = [rng.normal(size=D).astype(np.float32) * s for s in (0.0, 1.0, 3.0)]
    logstd0 = rng.uniform(-0.5, 0.5, size=D).astype(np.float32)
    std_var = torch.tensor(np.exp(logstd0), device=dev)     # a "Variable"

    def build(generic):
        mean_ph = zs.placeholder(torch.float32, name='mean')

        @zs.meta_bayesian_net()
        def model():
            bn = zs.BayesianNet()
            bn.normal('x', mean_ph.value, std=std_var, n_samples=C,
                      group_ndims=1)
            return bn
        mean_ph.feed(means[0], dev)
        x = torch.zeros(C, D, device=dev)
        h = zs.HMC(step_size=0.12, n_leapfrogs=4, seed=21)
        m = model()
        target = (lambda obs: m.observe(**obs).log_joint()) if generic else m
        op, info = h.sample(target, {}, {'x': x})
        return h, op, info, x, mean_ph

    hf, opf, inf_f, xf, ph_f = build(False)
    hg, opg, inf_g, xg, ph_g = build(True)
    assert hf.plan_kind == 'fused_diag_normal' and hg.plan_kind == 'generic'
    for it in range(9):
        if it == 6:
            std_var.mul_(1.5)             # in-place update of a parameter
        feed_f = {ph_f: means[it // 3]}
        feed_g = {ph_g: means[it // 3]}
        opf.run(feed_dict=feed_f)
        opg.run(feed_dict=feed_g)
        # same stream, same target: the two plans agree chain by chain
        np.testing.assert_allclose(inf_f.orig_log_prob.cpu().numpy(),
                                   inf_g.orig_log_prob.cpu().numpy(),
                                   rtol=2e-5, atol=2e-3)
        np.testing.assert_allclose(inf_f.acceptance_rate.cpu().numpy(),
                                   inf_g.acceptance_rate.cpu().numpy(),
                                   atol=2e-3)
        same = torch.isclose(xf, xg, atol=1e-4).all(dim=1).float().mean()
        assert float(same) > 0.97
        xg.copy_(xf)
    # and the zero-mean specialisation was left when the mean stopped being 0
    assert hf._plan.zero_mean is False


# (C, D): every ring instantiation that can produce the column statistics
# (NCH = 1..6), ragged last chunks, fewer turns than CUs, many chains per CU
COLSTATS_SHAPES = [(40, 132), (300, 256), (999, 260), (64, 512), (5000, 700),
                   (4100, 768), (20000, 1024), (70, 1028), (600, 1280),
                   (257, 1536)]


@pytest.mark.parametrize('C,D', COLSTATS_SHAPES)
@pytest.mark.parametrize('has_mass,has_mean', [(False, False), (True, False),
                                               (False, True), (True, True)])
def test_column_statistics_of_the_end_state_come_out_of_the_launch(
        env, C, D, has_mass, has_mean):
    """zshmc_adapt_link.colstats_*: the committing launch leaves, per
    workgroup, the column sums of (q' - m) and (q' - m)^2 of the state it
    ENDS in (proposal where accepted, start row where rejected) -- what the
    next iteration's EWMV update consumes (hmc.py:130-148, :288) -- and is
    otherwise bit-identical to the launch without them."""
    zs, torch = env
    from zhusuan_amd import _capi
    dev = torch.device('cuda', 0)
    g = torch.Generator(device='cpu').manual_seed(C + D)
    mean = torch.randn(D, generator=g).to(dev) if has_mean else None
    logstd = (0.5 * torch.rand(D, generator=g) - 0.25).to(dev)
    mass = (0.5 + torch.rand(D, generator=g)).to(dev) if has_mass else None
    q0 = (torch.randn(C, D, generator=g).to(dev) * torch.exp(logstd) +
          (mean if has_mean else 0.0))        # in equilibrium
    m = (0.3 * torch.randn(D, generator=g)).to(dev) + (mean if has_mean
                                                        else 0.0)
    rows = int(_capi.load().zshmc_fused_colstats_rows(
        C, D, int(has_mass), int(not has_mean)))
    assert 0 < rows <= 256
    k = FusedKernel(torch, C, D)
    # a step size whose acceptance is well inside (0, 1): dry runs
    best = None
    for f in np.geomspace(4.0, 0.05, 24):
        e = float(f) / D ** 0.25
        k.step(q0.clone(), mean, logstd, mass, e, 4, 123, 7, commit=0)
        a = float(k.stats[0].item()) / C
        if best is None or abs(a - 0.6) < abs(best[0] - 0.6):
            best = (a, e)
    assert 0.1 < best[0] < 0.95, best
    eps = best[1]
    qa = q0.clone()
    k.step(qa, mean, logstd, mass, eps, 4, 123, 7)
    info_a = [x.clone() for x in k.info]
    parts = torch.full((rows, 2 * D), float('nan'), dtype=torch.float64,
                       device=dev)
    qb = q0.clone()
    k.step(qb, mean, logstd, mass, eps, 4, 123, 7,
           link=k.link(colstats_mean=m, colstats_parts=parts))
    assert torch.equal(qa, qb)
    for x, y in zip(info_a, k.info):
        assert torch.equal(x, y)
    moved = (qb != q0).any(1)
    assert 0 < int(moved.sum()) < C        # both branches exercised
    d = (qb - m).double()                   # float32 difference, as the kernel
    want = torch.cat([d.sum(0), d.square().sum(0)])
    got = parts.sum(0)
    torch.testing.assert_close(got, want, rtol=1e-12, atol=1e-9)
    # the two reducers: one row for the all-reduce; EWMV update in one launch
    colsum = torch.empty(2 * D, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    _capi.call('zshmc_mass_colstats_reduce', parts.data_ptr(), rows, D,
               colsum.data_ptr(), stream)
    torch.testing.assert_close(colsum, want, rtol=1e-12, atol=1e-9)
    state = torch.zeros(_capi.STATE_WORDS, device=dev)
    state[_capi.ST_EWMV_T] = 3.0
    out = []
    for src, n in ((parts, rows), (colsum, 1)):
        st, mean_e = state.clone(), m.clone()
        var_e = torch.full((D,), 0.7, device=dev)
        mass_o = torch.empty(D, device=dev)
        ws = torch.zeros(2, dtype=torch.int32, device=dev)
        _capi.call('zshmc_mass_update_fused', st.data_ptr(), mean_e.data_ptr(),
                   var_e.data_ptr(), src.data_ptr(), n, C, D, 0.99, 0,
                   mass_o.data_ptr(), ws.data_ptr(), stream)
        assert float(st[_capi.ST_EWMV_T]) == 4.0 and int(ws[0]) == 0
        out.append((mean_e, var_e, mass_o))
    # against the two-launch form on the same sums
    st, mean_e = state.clone(), m.clone()
    var_e = torch.full((D,), 0.7, device=dev)
    mass_o = torch.empty(D, device=dev)
    _capi.call('zshmc_mass_update', st.data_ptr(), mean_e.data_ptr(),
               var_e.data_ptr(), colsum.clone().data_ptr(), C, D, 0.99, 1, 0,
               mass_o.data_ptr(), stream)
    assert float(st[_capi.ST_EWMV_T]) == 4.0
    for got3 in out:
        for a, b in zip(got3, (mean_e, var_e, mass_o)):
            torch.testing.assert_close(a, b, rtol=1e-6, atol=0)
    for a, b in zip(out[1], (mean_e, var_e, mass_o)):
        assert torch.equal(a, b)            # same row, same arithmetic


def test_column_statistics_are_refused_where_the_kernel_has_none(env):
    zs, torch = env
    from zhusuan_amd import _capi
    lib = _capi.load()
    for D in (4, 10, 128, 130, 1540, 2048):
        assert lib.zshmc_fused_colstats_rows(1000, D, 0, 1) == 0
    dev = torch.device('cuda', 0)
    C, D = 64, 96
    k = FusedKernel(torch, C, D)
    parts = torch.zeros(256, 2 * D, dtype=torch.float64, device=dev)
    with pytest.raises(_capi.ZshmcError, match='colstats'):
        k.step(torch.zeros(C, D, device=dev), None, torch.zeros(D, device=dev),
               None, 0.1, 2, 1, 1,
               link=k.link(colstats_mean=torch.zeros(D, device=dev),
                           colstats_parts=parts))


@pytest.mark.parametrize('C,D', [(1000, 10), (3000, 260), (70000, 1024)])
def test_run_many_is_bit_identical_to_a_loop_of_runs(env, C, D):
    """sample_op.run_many(n): the stretches that need nothing from the host
    are ONE zshmc_hmc_diag_normal_run call (launch loop on the C side); every
    state word, the latent and the last HMCInfo equal n single runs."""
    zs, torch = env
    dev = torch.device('cuda', 0)
    logstd = torch.linspace(-0.5, 0.5, D, device=dev)
    mean = torch.linspace(-1, 1, D, device=dev)
    out = []
    for many in (False, True):
        @zs.meta_bayesian_net()
        def model():
            bn = zs.BayesianNet()
            bn.normal('x', mean, logstd=logstd, n_samples=C, group_ndims=1)
            return bn
        f_ss, f_m = zs.placeholder(bool), zs.placeholder(bool)
        h = zs.HMC(step_size=0.1, n_leapfrogs=4, adapt_step_size=f_ss,
                   adapt_mass=f_m, mass_collect_iters=6, seed=5)
        x = torch.zeros(C, D, device=dev)
        op, info = h.sample(model(), {}, {'x': x})
        calls = []
        from zhusuan_amd import _capi
        real = _capi.call

        def spy(name, *a):
            calls.append(name)
            return real(name, *a)
        _capi.call = spy
        try:
            # (blocks of 40 and 48)
            for n, feed in ((9, {f_ss: True, f_m: True}),
                            (40, {f_ss: True, f_m: False}),
                            (50, {f_ss: False, f_m: False})):
                if many:
                    op.run_many(n, feed_dict=feed, sync=False)
                else:
                    for _ in range(n):
                        op.run(feed_dict=feed, sync=False)
            h.check_numerics()
        finally:
            _capi.call = real
        out.append((x.clone(), h.get_state()['state'].clone(),
                    info.acceptance_rate.clone(), info.log_prob.clone(), h.t,
                    calls))
    (xa, sa, aa, la, ta, ca), (xb, sb, ab, lb, tb, cb) = out
    assert torch.equal(xa, xb) and torch.equal(sa, sb)
    assert torch.equal(aa, ab) and torch.equal(la, lb) and ta == tb == 99
    assert 'zshmc_hmc_diag_normal_run' not in ca
    assert cb.count('zshmc_hmc_diag_normal_run') == 2   # 40 adaptive; 50 - 2 held
    assert cb.count('zshmc_hmc_diag_normal_step') == \
        ca.count('zshmc_hmc_diag_normal_step') - 40 - 48
