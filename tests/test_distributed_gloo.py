"""N > 1 path on CPU: 2, 4 and 8 `gloo` ranks shard the chain axis (even and
uneven shards, down to ONE chain on a rank).  The product's
ChainSharding (layout all-gather + adaptation all-reduces) is exercised for
real; the per-rank transition is the NumPy oracle (no GPU here), fed through
the same hooks the GPU plan uses (global chain offset for the RNG counters,
all-reduce of the acceptance sum / mass column sums).  The sharded run must
reproduce the single-process run: RNG invariance is bit-exact, the
adaptation state agrees to float32 summation order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.hmc_ref import HMC as RefHMC, DiagNormalModel

C, D, L, ITERS = 96, 12, 4, 14


def _problem():
    rng = np.random.RandomState(0)
    mean = rng.normal(size=D).astype(np.float32)
    logstd = rng.uniform(-0.7, 0.7, size=D).astype(np.float32)
    q0 = (mean + rng.normal(size=(C, D))).astype(np.float32)
    return mean, logstd, q0


def _run_oracle(q, chain_offset=0, n_global=None, allreduce=None):
    mean, logstd, _ = _problem()
    model = DiagNormalModel(mean, logstd=logstd)
    h = RefHMC(step_size=0.05, n_leapfrogs=L, adapt_step_size=True,
               adapt_mass=True, mass_collect_iters=5, seed=77)
    h.sample(model.log_joint, model.grad, [q], chain_offset=chain_offset,
             n_chains_global=n_global, allreduce_sum=allreduce)
    eps, acc = [], []
    for i in range(ITERS):
        info = h.step(adapt_step_size=i < 10, adapt_mass=i < 10)
        eps.append(float(info.updated_step_size))
        acc.append(info.acceptance_rate.copy())
    return np.array(eps), np.stack(acc), q, np.asarray(h.last_mass[0])


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, splits, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from zhusuan_amd.distributed import ChainSharding
        _, _, q0 = _problem()
        lo, hi = splits[rank]
        q = q0[lo:hi].copy()
        sh = ChainSharding()
        off, n_global = sh.layout(hi - lo, torch.device('cpu'))
        assert (off, n_global) == (lo, C)

        def allreduce(a):
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
            sh.all_reduce_sum(t)
            return t.numpy().astype(np.float32)

        eps, acc, q, mass = _run_oracle(q, off, n_global, allreduce)
        np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), eps=eps, acc=acc,
                 q=q, mass=mass)
    finally:
        dist.destroy_process_group()


def _even(world):
    from zhusuan_amd.distributed import shard_bounds
    return [shard_bounds(C, r, world) for r in range(world)]


@pytest.mark.parametrize('splits', [
    [(0, 48), (48, 96)], [(0, 31), (31, 96)],
    [(0, 5), (5, 40), (40, 41), (41, 96)],
    _even(8), [(0, 1), (1, 20), (20, 33), (33, 34), (34, 60), (60, 61),
               (61, 90), (90, 96)]],
    ids=['2-even', '2-uneven', '4-uneven', '8-even', '8-uneven'])
def test_sharded_run_matches_single_process(tmp_path, splits):
    world = len(splits)
    _, _, q0 = _problem()
    eps1, acc1, q1, mass1 = _run_oracle(q0.copy())
    port = _free_port()
    mp.spawn(_worker, args=(world, port, splits, str(tmp_path)), nprocs=world,
             join=True)
    r = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % i))
         for i in range(world)]
    # replicated adaptation state is identical on every rank
    for other in r[1:]:
        np.testing.assert_array_equal(r[0]['eps'], other['eps'])
        np.testing.assert_array_equal(r[0]['mass'], other['mass'])
    # and equals the single-process run up to float32 summation order
    np.testing.assert_allclose(r[0]['eps'], eps1, rtol=2e-5)
    np.testing.assert_allclose(r[0]['mass'], mass1, rtol=2e-4)
    acc = np.concatenate([x['acc'] for x in r], axis=1)
    np.testing.assert_allclose(acc, acc1, atol=2e-4)
    q = np.concatenate([x['q'] for x in r], axis=0)
    close = np.isclose(q, q1, atol=1e-3).all(axis=1)
    assert close.mean() >= 0.97       # a borderline accept may flip


def test_first_transition_is_bit_exact_across_shardings():
    """Before any adaptation has mixed information across chains, a shard
    with the right global offset reproduces its slice bit for bit."""
    mean, logstd, q0 = _problem()
    model = DiagNormalModel(mean, logstd=logstd)

    def one(q, off):
        h = RefHMC(step_size=0.1, n_leapfrogs=L, seed=5)
        h.sample(model.log_joint, model.grad, [q], chain_offset=off,
                 n_chains_global=C)
        info = h.step()
        return q, info.acceptance_rate

    qa, acca = one(q0.copy(), 0)
    qb, accb = one(q0[40:].copy(), 40)
    np.testing.assert_array_equal(qb, qa[40:])
    np.testing.assert_array_equal(accb, acca[40:])
