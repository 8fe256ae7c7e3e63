"""The N > 1 path of the NON-fused plans on real kernels: 2, 4 and 8
processes (gloo rendezvous, all on cuda:0; 10 / 14 chains, so that at 8 ranks
the shards are uneven and most hold ONE chain) shard the leading chain axis of the
topic-model E step (BASELINE configs[4] family, `_DenseLikelihoodPlan`, mass
adaptation on: ONE all-reduce of [sum acc, flag, colsum[2 K]] per
transition) and of Bayesian logistic regression, on the native plans and on
the autograd-driven generic plan ('blrb': weights of a size that is not a
multiple of 4 plus a per-chain bias -- the native plan's packed state, two
latents' column sums in the one all-reduce).  Adaptation off: the rank-concatenated
states equal the single-process run bit for bit (RNG keyed by the GLOBAL
chain index; per-document prior rows addressed with a row period under the
chain offset).  Adaptation on: step size, tuner state and mass agree across
ranks and with the single-process run.  Rank 0 alone reads
`updated_step_size` and `get_state()` every iteration: neither communicates.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import helpers_sharded_cases as cases

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope='module', params=[2, 4, 8],
                ids=lambda w: 'world%d' % w)
def ranks(request, tmp_path_factory):
    world = request.param
    out = tmp_path_factory.mktemp('sharded_plans')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()),
           os.path.join(HERE, 'sharded_plan_worker.py'), str(out)]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return [np.load(str(out / ('rank%d.npz' % i))) for i in range(world)]


@pytest.mark.parametrize('family', ['lntm', 'blr', 'blrb'])
@pytest.mark.parametrize('native', [True, False])
def test_sharded_plan_matches_single_process(ranks, family, native):
    import torch
    import zhusuan_amd as zs
    dev = torch.device('cuda', 0)
    n = getattr(cases, family + '_problem')()['q0'].shape[0]
    key = lambda adapt, k: '%s/%d/%d/%s' % (family, native, adapt, k)

    # adaptation off: bit-exact whatever the sharding
    one = cases.run(zs, torch, dev, family, 0, n, False, None, native)
    q = np.concatenate([r[key(0, 'q')] for r in ranks])
    acc = np.concatenate([r[key(0, 'acc')].reshape(-1) for r in ranks])
    np.testing.assert_array_equal(q, one['q'])
    np.testing.assert_array_equal(acc, one['acc'].reshape(-1))

    # adaptation on (step size + mass): replicated state identical on every
    # rank and equal to the single-process run up to summation order
    one = cases.run(zs, torch, dev, family, 0, n, True, None, native)
    for k in ('step_size', 'state', 'mass'):
        for r in ranks[1:]:
            np.testing.assert_array_equal(ranks[0][key(1, k)], r[key(1, k)])
    np.testing.assert_allclose(ranks[0][key(1, 'state')], one['state'],
                               rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(ranks[0][key(1, 'mass')], one['mass'],
                               rtol=2e-5)
    q = np.concatenate([r[key(1, 'q')] for r in ranks])
    close = np.isclose(q, one['q'], atol=2e-4).reshape(q.shape[0], -1).all(1)
    assert close.mean() >= 0.9, close
    # rank 0 read the step size after every run without its peer
    eps = ranks[0][key(1, 'eps')]
    assert eps.shape == (cases.N_ITERS,) and np.isfinite(eps).all()
    assert float(eps[-1]) == float(ranks[1][key(1, 'step_size')])
