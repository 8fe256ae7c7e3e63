"""N > 1 path on real kernels: two processes (gloo rendezvous, both on
cuda:0 because the test box has one GPU) shard the chains of one problem.
With adaptation off the sharded run must equal the single-process run bit
for bit (global-chain-index RNG); with adaptation on, the replicated state
must agree across ranks and with the single-process trace.  Also launches
bench.py --gpus 2 the way the driver does (torch.distributed.run)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import zhusuan_amd as zs
from zhusuan_amd.distributed import ChainSharding, shard_bounds
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
dev = torch.device('cuda', 0)
C, D, L = 640, 96, 5
rng = np.random.RandomState(0)
mean = torch.tensor(rng.normal(size=D).astype(np.float32), device=dev)
logstd = torch.tensor(rng.uniform(-.5, .5, size=D).astype(np.float32), device=dev)
q0 = torch.tensor(rng.normal(size=(C, D)).astype(np.float32), device=dev)
lo, hi = shard_bounds(C, rank, world)
out = {}
for adapt in (None, True):
    x = q0[lo:hi].clone()
    n = hi - lo
    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        bn.normal('x', mean, logstd=logstd, n_samples=n, group_ndims=1)
        return bn
    hmc = zs.HMC(step_size=0.08, n_leapfrogs=L, adapt_step_size=adapt,
                 adapt_mass=adapt, mass_collect_iters=3, seed=5,
                 sharding=ChainSharding())
    op, info = hmc.sample(model(), {}, {'x': x})
    eps = []
    for i in range(8):
        op.run()
        eps.append(float(info.updated_step_size.item()))
    out['x_%%s' %% adapt] = x.cpu().numpy()
    out['eps_%%s' %% adapt] = np.array(eps)
    if adapt:
        out['mass'] = hmc._plan.mass[0].cpu().numpy()
np.savez(os.path.join(%(out)r, 'rank%%d.npz' %% rank), lo=lo, hi=hi, **out)
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(script_args, nproc, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_two_rank_sharded_gpu_run_matches_single_process(tmp_path):
    import torch
    import zhusuan_amd as zs
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT, out=str(tmp_path)))
    r = _launch([str(script)], 2)
    assert r.returncode == 0, r.stderr[-3000:]
    ranks = [np.load(str(tmp_path / ('rank%d.npz' % i))) for i in range(2)]

    dev = torch.device('cuda', 0)
    C, D, L = 640, 96, 5
    rng = np.random.RandomState(0)
    mean = torch.tensor(rng.normal(size=D).astype(np.float32), device=dev)
    logstd = torch.tensor(rng.uniform(-.5, .5, size=D).astype(np.float32),
                          device=dev)
    q0 = torch.tensor(rng.normal(size=(C, D)).astype(np.float32), device=dev)
    single = {}
    for adapt in (None, True):
        x = q0.clone()

        @zs.meta_bayesian_net()
        def model():
            bn = zs.BayesianNet()
            bn.normal('x', mean, logstd=logstd, n_samples=C, group_ndims=1)
            return bn
        hmc = zs.HMC(step_size=0.08, n_leapfrogs=L, adapt_step_size=adapt,
                     adapt_mass=adapt, mass_collect_iters=3, seed=5)
        op, info = hmc.sample(model(), {}, {'x': x})
        eps = []
        for i in range(8):
            op.run()
            eps.append(float(info.updated_step_size.item()))
        single['x_%s' % adapt] = x.cpu().numpy()
        single['eps_%s' % adapt] = np.array(eps)
        if adapt:
            single['mass'] = hmc._plan.mass[0].cpu().numpy()
    # no adaptation: bit-exact regardless of the sharding
    x_sh = np.concatenate([ranks[0]['x_None'], ranks[1]['x_None']])
    np.testing.assert_array_equal(x_sh, single['x_None'])
    # adaptation on: replicated state identical on both ranks ...
    np.testing.assert_array_equal(ranks[0]['eps_True'], ranks[1]['eps_True'])
    np.testing.assert_array_equal(ranks[0]['mass'], ranks[1]['mass'])
    # ... and equal to the single-process trace up to summation order
    np.testing.assert_allclose(ranks[0]['eps_True'], single['eps_True'],
                               rtol=1e-5)
    np.testing.assert_allclose(ranks[0]['mass'], single['mass'], rtol=1e-5)
    x_sh = np.concatenate([ranks[0]['x_True'], ranks[1]['x_True']])
    close = np.isclose(x_sh, single['x_True'], atol=1e-4).all(axis=1)
    assert close.mean() > 0.98


def test_bench_two_ranks_prints_contract_json():
    """`python bench.py --gpus 2 ...` WITHOUT a launcher (the driver's N = 1
    command line with another --gpus): bench.py starts its own ranks."""
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR',
                        'MASTER_PORT')}
    env['ZSHMC_DIST_BACKEND'] = 'gloo'
    r = subprocess.run(
        [sys.executable, 'bench.py', '--gpus', '2', '--steps', '10',
         '--warmup', '2', '--chains-per-gpu', '4096', '--no-ess',
         '--lntm-chains-per-gpu', '8', '--lntm-docs', '48',
         '--lntm-vocab', '700'], cwd=ROOT, env=env, capture_output=True,
        text=True, timeout=600)
    assert 'no launcher environment, starting 2 ranks' in r.stderr
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['steps'] == 10 and out['warmup'] == 2
    # configs[4] rides along at N > 1: leading chain axis sharded, literal
    # spelling on the native plan, adaptation on in the timed region
    (extra,) = out['extra_configs']
    assert extra['plan'] == 'mixture_multinomial' and extra['n_gpus'] == 2
    assert 'ONE all-reduce of 258 doubles' in extra['collective']
    assert 0.2 < extra['mean_acceptance'] <= 1.0 and extra['value'] > 0
    assert out['scaling'] == 'weak' and out['higher_is_better'] is True
    assert out['config']['n_chains_total'] == 8192
    assert out['value'] > 0 and 0.3 < out['mean_acceptance'] <= 1.0
    assert out['roofline']['bound'] == 'hbm'
    assert 'cpu_baseline' not in out       # rank 0 at N = 1 only


RCCL_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import zhusuan_amd as zs
from zhusuan_amd.distributed import ChainSharding, shard_bounds
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)   # bootstrap only
torch.cuda.set_device(rank %% torch.cuda.device_count())
dev = torch.device('cuda', torch.cuda.current_device())
sh = ChainSharding(backend='rccl', always_reduce=True)
assert sh.rccl_ranks == world
# the raw collective
t = torch.full((5,), float(rank + 1), dtype=torch.float64, device=dev)
sh.all_reduce_sum(t)
torch.cuda.synchronize()
assert t.tolist() == [world * (world + 1) / 2.0] * 5, t
C, D, L = 4096, 1024, 5
g = torch.Generator(device='cpu').manual_seed(0)
logstd = torch.linspace(-1, 1, D).to(dev)
q0 = torch.randn(C, D, generator=g)
lo, hi = shard_bounds(C, rank, world)
x = q0[lo:hi].to(dev).contiguous()
@zs.meta_bayesian_net()
def model():
    bn = zs.BayesianNet()
    bn.normal('x', torch.zeros(D, device=dev), logstd=logstd,
              n_samples=hi - lo, group_ndims=1)
    return bn
flag = zs.placeholder(bool)
hmc = zs.HMC(step_size=0.05, n_leapfrogs=L, adapt_step_size=flag,
             adapt_mass=flag, mass_collect_iters=4, seed=5, sharding=sh)
op, info = hmc.sample(model(), {}, {'x': x})
assert hmc.plan_kind == 'fused_diag_normal'
eps = []
for i in range(16):
    op.run(feed_dict={flag: i < 12}, sync=(i %% 5 == 0))
    if i %% 3 == 2:
        eps.append(float(info.updated_step_size.item()))
hmc.check_numerics()
eps.append(float(info.updated_step_size.item()))
# a stretch through the C-side launch loop (zshmc_hmc_diag_normal_run with the
# communicator: the all-reduce between the launches is enqueued there), step
# size adapting, mass held
adapt2, hold2 = zs.placeholder(bool), zs.placeholder(bool)
hmc.adapt_step_size, hmc.adapt_mass = adapt2, hold2
op.run_many(7, feed_dict={adapt2: True, hold2: False}, sync=False)
hmc.check_numerics()
eps.append(float(info.updated_step_size.item()))
np.savez(os.path.join(%(out)r, 'rccl_rank%%d.npz' %% rank), lo=lo, hi=hi,
         x=x.cpu().numpy(), eps=np.array(eps),
         mass=hmc._plan.mass[0].cpu().numpy(),
         state=hmc.get_state()['state'].numpy())
sh.close()
dist.destroy_process_group()
'''


def _single_process_reference(torch, zs):
    dev = torch.device('cuda', 0)
    C, D, L = 4096, 1024, 5
    g = torch.Generator(device='cpu').manual_seed(0)
    logstd = torch.linspace(-1, 1, D).to(dev)
    x = torch.randn(C, D, generator=g).to(dev)

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        bn.normal('x', torch.zeros(D, device=dev), logstd=logstd,
                  n_samples=C, group_ndims=1)
        return bn
    flag = zs.placeholder(bool)
    hmc = zs.HMC(step_size=0.05, n_leapfrogs=L, adapt_step_size=flag,
                 adapt_mass=flag, mass_collect_iters=4, seed=5)
    op, info = hmc.sample(model(), {}, {'x': x})
    eps = []
    for i in range(16):
        op.run(feed_dict={flag: i < 12}, sync=(i % 5 == 0))
        if i % 3 == 2:
            eps.append(float(info.updated_step_size.item()))
    hmc.check_numerics()
    eps.append(float(info.updated_step_size.item()))
    adapt2, hold2 = zs.placeholder(bool), zs.placeholder(bool)
    hmc.adapt_step_size, hmc.adapt_mass = adapt2, hold2
    for _ in range(7):                       # the same stretch, run by run
        op.run(feed_dict={adapt2: True, hold2: False}, sync=False)
    hmc.check_numerics()
    eps.append(float(info.updated_step_size.item()))
    return (x.cpu().numpy(), np.array(eps), hmc._plan.mass[0].cpu().numpy(),
            hmc.get_state()['state'].numpy())


@pytest.mark.parametrize('world', [1, 2])
def test_direct_rccl_communicator(tmp_path, world):
    """The production collective: ChainSharding(backend='rccl') =
    ncclCommInitRank + ncclAllReduce through the C-ABI (zshmc_comm_*), one
    message per transition on the compute stream.  world = 1 runs on any box
    (a one-rank communicator still executes the whole RCCL path); world = 2
    needs two GPUs (RCCL refuses two ranks on one device)."""
    import torch
    import zhusuan_amd as zs
    if world > torch.cuda.device_count():
        pytest.skip('needs %d GPUs' % world)
    script = tmp_path / 'rccl_worker.py'
    script.write_text(RCCL_WORKER % dict(root=ROOT, out=str(tmp_path)))
    r = _launch([str(script)], world)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    ranks = [np.load(str(tmp_path / ('rccl_rank%d.npz' % i)))
             for i in range(world)]
    x1, eps1, mass1, state1 = _single_process_reference(torch, zs)
    for r_ in ranks[1:]:
        np.testing.assert_array_equal(r_['eps'], ranks[0]['eps'])
        np.testing.assert_array_equal(r_['mass'], ranks[0]['mass'])
        np.testing.assert_array_equal(r_['state'], ranks[0]['state'])
    x = np.concatenate([r_['x'] for r_ in ranks])
    if world == 1:
        # same sums in the same order; the update itself runs in the NEXT
        # launch's prologue here (sharded path) and in the retiring workgroup
        # there: same equations, possibly different last-bit rounding
        # (the last entry follows 7 more adaptive updates through the C-side
        # launch loop: the differences compound a little)
        np.testing.assert_allclose(ranks[0]['eps'][:-1], eps1[:-1], rtol=2e-6)
        np.testing.assert_allclose(ranks[0]['eps'][-1], eps1[-1], rtol=2e-5)
        np.testing.assert_allclose(ranks[0]['state'], state1, rtol=2e-6,
                                   atol=1e-7)
        close = np.isclose(x, x1, atol=1e-4).all(axis=1)
        assert close.mean() > 0.99
    else:
        np.testing.assert_allclose(ranks[0]['eps'], eps1, rtol=1e-5)
        np.testing.assert_allclose(ranks[0]['mass'], mass1, rtol=1e-5)
        close = np.isclose(x, x1, atol=1e-4).all(axis=1)
        assert close.mean() > 0.98


def test_bench_lntm_workload_two_ranks():
    """`bench.py --workload lntm --gpus 2`: BASELINE configs[4] as the line's
    own workload (reduced sizes here), the way the driver would launch it."""
    r = _launch(['bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1',
                 '--workload', 'lntm', '--lntm-chains-per-gpu', '8',
                 '--lntm-docs', '48', '--lntm-vocab', '700'], 2,
                {'ZSHMC_DIST_BACKEND': 'gloo'})
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['scaling'] == 'weak'
    assert out['plan'] == 'mixture_multinomial'
    assert out['roofline']['bound'] == 'mfma' and out['roofline']['frac'] > 0
    assert out['value'] == pytest.approx(
        2 * 8 * 48 * 20 / (out['ms_per_step'] * 1e-3), rel=1e-6)
    assert 0.2 < out['mean_acceptance'] <= 1.0
    assert out['ess']['ess_per_sec'] > 0
