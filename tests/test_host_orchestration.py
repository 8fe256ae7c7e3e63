"""zhusuan_amd/hmc.py's host orchestration on a box WITHOUT a GPU: `HMC._run`
(flags fed per run, the step-size search loop, retired / pending
dual-averaging updates, flush) drives a NumPy stand-in of the fused-plan
entry points (tests/fake_zshmc.py, the contract of include/zshmc.h) and must
reproduce oracle/hmc_ref.py -- alone, and as two gloo ranks that shard the
chain axis (one all-reduce of the statistics per transition, the update
applied by the NEXT launch's prologue)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.hmc_ref import HMC as RefHMC, DiagNormalModel

C, D, L, ITERS = 64, 8, 4, 16
SEED = 91


def _problem():
    rng = np.random.RandomState(5)
    mean = rng.normal(size=D).astype(np.float32)
    logstd = rng.uniform(-0.5, 0.5, size=D).astype(np.float32)
    q0 = (mean + rng.normal(size=(C, D))).astype(np.float32)
    return mean, logstd, q0


def _flags(i):
    return i < 11            # adapt for 11 iterations, then hold


MASS_COLLECT = 4


def _reference(mass=False):
    mean, logstd, q0 = _problem()
    model = DiagNormalModel(mean, logstd=logstd)
    q = q0.copy()
    kw = dict(adapt_mass=True, mass_collect_iters=MASS_COLLECT) if mass else {}
    h = RefHMC(step_size=0.05, n_leapfrogs=L, adapt_step_size=True,
               target_acceptance_rate=0.8, seed=SEED, **kw)
    h.sample(model.log_joint, model.grad, [q])
    eps = [float(h.step(adapt_step_size=_flags(i),
                        adapt_mass=_flags(i) if mass else None)
                 .updated_step_size) for i in range(ITERS)]
    return np.array(eps), q


def _product_run(q_np, sharding, monkeypatch_ctx, read_every_run=True,
                 mass=False):
    """The product's HMC over the fake library on CPU tensors."""
    import zhusuan_amd as zs
    from zhusuan_amd import _capi, hmc as H
    from fake_zshmc import FakeLibrary
    fake = FakeLibrary()
    monkeypatch_ctx.setattr(_capi, 'call', fake.call)
    monkeypatch_ctx.setattr(H._capi, 'current_stream', lambda: 0)
    mean, logstd, _ = _problem()
    mean_t, logstd_t = torch.tensor(mean), torch.tensor(logstd)
    q = torch.tensor(q_np)
    flag = zs.placeholder(bool)
    kw = dict(adapt_mass=flag, mass_collect_iters=MASS_COLLECT) if mass else {}
    hmc = zs.HMC(step_size=0.05, n_leapfrogs=L, adapt_step_size=flag,
                 target_acceptance_rate=0.8, seed=SEED, sharding=sharding,
                 **kw)
    node = zs.distributions.Normal(mean_t, logstd=logstd_t, group_ndims=1)
    plan = H._FusedDiagNormalPlan(hmc, ['x'], [q], (q.shape[0],),
                                  torch.device('cpu'),
                                  lambda: (mean_t, logstd_t, node))
    hmc._plan = plan
    plan.state[_capi.ST_STEP_SIZE] = 0.05
    eps = []
    for i in range(ITERS):
        hmc._run({flag: _flags(i)}, sync=False)
        if read_every_run or i == ITERS - 1:
            hmc.flush()         # what reading HMCInfo.updated_step_size does
            eps.append(float(plan.state[_capi.ST_STEP_SIZE]))
    return np.array(eps), q.numpy(), fake, hmc


@pytest.mark.parametrize('mass', [False, True])
def test_single_process_orchestration_matches_oracle(monkeypatch, mass):
    want_eps, want_q = _reference(mass)
    _, _, q0 = _problem()
    eps, q, fake, hmc = _product_run(q0.copy(), None, monkeypatch, mass=mass)
    # (with mass adaptation: the one-pass float64 column statistics of
    # csrc/adapt.hip against the reference's two-pass float32 form)
    np.testing.assert_allclose(eps, want_eps, rtol=5e-5 if mass else 2e-6)
    # (the step size comes from csrc/fused_args.h compiled for the host --
    # libm's powf / expf against NumPy's in the oracle: last-bit differences
    # of epsilon, hence of the states)
    np.testing.assert_allclose(q, want_q, rtol=0, atol=2e-4 if mass else 1e-5)
    if mass:
        # the step size is searched again at t == mass_collect_iters.  One
        # mass launch per iteration WHILE THE FLAG IS ON (rows of column sums
        # -> EWMV update -> mass -> tau); with the flag off the mass stays
        # what it was and nothing is launched.  The column sums are taken at
        # the END of a run for the next one (they travel in that run's one
        # all-reduce when sharded): once before the first update, then one
        # per adaptive run.
        n_adaptive = sum(_flags(i) for i in range(ITERS))
        assert fake.calls.count('zshmc_mass_update_fused') == n_adaptive
        assert fake.calls.count('zshmc_mass_update') == 0
        assert fake.calls.count('zshmc_mass_colstats') == n_adaptive + 1
        return
    # one launch per transition once the search at t = 1 is over, the update
    # carried by the launch (no separate update call, no flush work)
    n_search = hmc.n_init_trips
    assert fake.calls.count('zshmc_hmc_diag_normal_step') == ITERS + n_search
    assert 'zshmc_stepsize_update' not in fake.calls
    assert 'zshmc_stepsize_flush' not in fake.calls
    assert hmc.t == ITERS


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, read_every_run, mass):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from zhusuan_amd.distributed import ChainSharding

        class Ctx(object):            # monkeypatch stand-in for a subprocess
            @staticmethod
            def setattr(obj, name, value):
                setattr(obj, name, value)
        _, _, q0 = _problem()
        lo, hi = rank * C // world, (rank + 1) * C // world
        sh = ChainSharding(backend='torch', chain_offset=lo, n_chains_global=C)
        eps, q, fake, hmc = _product_run(q0[lo:hi].copy(), sh, Ctx,
                                         read_every_run, mass)
        np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), eps=eps, q=q,
                 n_flush=fake.calls.count('zshmc_stepsize_flush'),
                 n_launch=fake.calls.count('zshmc_hmc_diag_normal_step'),
                 n_search=hmc.n_init_trips)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,read_every_run,mass', [
    (2, True, False), (2, False, False), (2, False, True),
    (4, False, True), (8, False, True)])
def test_sharded_orchestration_matches_oracle(tmp_path, world, read_every_run,
                                              mass):
    """Sharded chains (2, 4, 8 ranks; uneven shards when C does not divide):
    the update of transition t is applied in the prologue of launch t + 1
    from the all-reduced sum -- or by flush when the step size is read first;
    every rank follows the single-process oracle either way."""
    want_eps, want_q = _reference(mass)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path),
                            read_every_run, mass), nprocs=world, join=True)
    rows = []
    for r in range(world):
        d = np.load(str(tmp_path / ('rank%d.npz' % r)))
        if read_every_run:
            np.testing.assert_allclose(d['eps'], want_eps, rtol=2e-6)
            # every update is retired by flush and the next launch has nothing
            # pending; after two HOLD updates in a row the state is at its
            # fixed point and runs carry no update at all (hmc.py:108-110,
            # `steady`): 11 adaptive + 2 hold
            assert int(d['n_flush']) == 13
        else:
            np.testing.assert_allclose(d['eps'], want_eps[-1:],
                                       rtol=5e-5 if mass else 2e-6)
            # nobody asked in between: every update rode in the next launch's
            # prologue (with mass adaptation the second search, at t ==
            # mass_collect_iters, retires the one pending then by flush)
            assert int(d['n_flush']) == (1 if mass else 0)
        rows.append(d['q'])
        if not mass:
            assert int(d['n_launch']) == ITERS + int(d['n_search'])
    np.testing.assert_allclose(np.concatenate(rows), want_q, rtol=0,
                               atol=2e-4 if mass else 1e-5)


@pytest.mark.parametrize('mass', [False, True])
def test_run_many_equals_a_loop_of_runs(monkeypatch, mass):
    """sample_op.run_many(n): the stretches that need nothing from the host
    (no search, mass not adapting) are one zshmc_hmc_diag_normal_run call;
    state, step size and iteration counter equal n single runs."""
    import zhusuan_amd as zs
    from zhusuan_amd import _capi, hmc as H
    from fake_zshmc import FakeLibrary
    mean, logstd, q0 = _problem()
    out = []
    for many in (False, True):
        fake = FakeLibrary()
        monkeypatch.setattr(_capi, 'call', fake.call)
        monkeypatch.setattr(H._capi, 'current_stream', lambda: 0)
        mean_t, logstd_t = torch.tensor(mean), torch.tensor(logstd)
        q = torch.tensor(q0.copy())
        f_ss, f_m = zs.placeholder(bool), zs.placeholder(bool)
        kw = dict(adapt_mass=f_m, mass_collect_iters=MASS_COLLECT) \
            if mass else {}
        hmc = zs.HMC(step_size=0.05, n_leapfrogs=L, adapt_step_size=f_ss,
                     target_acceptance_rate=0.8, seed=SEED, **kw)
        node = zs.distributions.Normal(mean_t, logstd=logstd_t, group_ndims=1)
        plan = H._FusedDiagNormalPlan(hmc, ['x'], [q], (q.shape[0],),
                                      torch.device('cpu'),
                                      lambda: (mean_t, logstd_t, node))
        hmc._plan = plan
        plan.state[_capi.ST_STEP_SIZE] = 0.05
        op = H._SampleOp(hmc)
        # 6 adaptive (search at t = 1 and, with mass, t = 4), 7 adaptive step
        # size only, 9 with everything held
        for n, feed in ((6, {f_ss: True, f_m: True}),
                        (7, {f_ss: True, f_m: False}),
                        (9, {f_ss: False, f_m: False})):
            if many:
                hmc._run_many(n, feed, sync=False)
            else:
                for _ in range(n):
                    hmc._run(feed, sync=False)
        hmc.flush()
        out.append((q.numpy().copy(), plan.state.numpy().copy(), hmc.t,
                    fake.calls))
    (qa, sa, ta, ca), (qb, sb, tb, cb) = out
    np.testing.assert_array_equal(qa, qb)
    np.testing.assert_array_equal(sa, sb)
    assert ta == tb == 22
    assert 'zshmc_hmc_diag_normal_run' not in ca
    # adaptive-step-size stretch: one block (after the searches); held
    # stretch: two HOLD updates singly, then one block
    assert cb.count('zshmc_hmc_diag_normal_run') >= 2
    assert cb.count('zshmc_hmc_diag_normal_step') < \
        ca.count('zshmc_hmc_diag_normal_step')
