"""N > 1 path on real kernels: 2, 4 and 8 processes (gloo rendezvous, all on
cuda:0 because the test box has one GPU) shard the chains of one problem --
650 chains, so that at 4 and 8 ranks the shards are UNEVEN (163/163/162/162;
82/82/81/...).
With adaptation off the sharded run must equal the single-process run bit
for bit (global-chain-index RNG); with adaptation on, the replicated state
must agree across ranks and with the single-process trace.  Also launches
bench.py --gpus 2 the way the driver does (torch.distributed.run)."""
import json
import os
import socket
import subprocess
import sys
import time

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
C, D, L = 650, 96, 5
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


WORLDS = [2, 4, 8]


@pytest.mark.parametrize('world', WORLDS)
def test_sharded_gpu_run_matches_single_process(tmp_path, world):
    import torch
    import zhusuan_amd as zs
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT, out=str(tmp_path)))
    r = _launch([str(script)], world)
    assert r.returncode == 0, r.stderr[-3000:]
    ranks = [np.load(str(tmp_path / ('rank%d.npz' % i)))
             for i in range(world)]
    assert [int(r_['hi']) - int(r_['lo']) for r_ in ranks] == \
        [650 // world + (1 if i < 650 % world else 0) for i in range(world)]

    dev = torch.device('cuda', 0)
    C, D, L = 650, 96, 5
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
    x_sh = np.concatenate([r_['x_None'] for r_ in ranks])
    np.testing.assert_array_equal(x_sh, single['x_None'])
    # adaptation on: replicated state identical on every rank ...
    for r_ in ranks[1:]:
        np.testing.assert_array_equal(ranks[0]['eps_True'], r_['eps_True'])
        np.testing.assert_array_equal(ranks[0]['mass'], r_['mass'])
    # ... and equal to the single-process trace up to summation order
    np.testing.assert_allclose(ranks[0]['eps_True'], single['eps_True'],
                               rtol=1e-5)
    np.testing.assert_allclose(ranks[0]['mass'], single['mass'], rtol=1e-5)
    x_sh = np.concatenate([r_['x_True'] for r_ in ranks])
    close = np.isclose(x_sh, single['x_True'], atol=1e-4).all(axis=1)
    assert close.mean() > 0.98


def _bench_output(stdout):
    """(contract line, [extra records], detail record) of one bench.py run:
    exactly one stdout line starts with `{` and it is the last one."""
    lines = stdout.splitlines()
    contract = [l for l in lines if l.startswith('{')]
    assert len(contract) == 1 and lines[-1] == contract[0], lines[-3:]
    assert len(contract[0].encode()) <= 4096

    def strict(name):
        raise AssertionError('bare %s in the line' % name)
    extras = [json.loads(l[len('#bench-extra '):], parse_constant=strict)
              for l in lines if l.startswith('#bench-extra ')]
    (detail,) = [json.loads(l[len('#bench-detail '):], parse_constant=strict)
                 for l in lines if l.startswith('#bench-detail ')]
    # nothing else on stdout: the launcher's own chatter stays on stderr
    assert all(l.startswith(('{', '#bench-extra ', '#bench-detail '))
               for l in lines), [l[:80] for l in lines]
    return json.loads(contract[0], parse_constant=strict), extras, detail


@pytest.mark.parametrize('world,scaling', [(2, 'weak'), (4, 'weak'),
                                           (8, 'weak'), (4, 'strong'),
                                           (8, 'strong')])
def test_bench_ranks_print_one_contract_line(world, scaling):
    """`python bench.py --gpus N ...` WITHOUT a launcher (the driver's N = 1
    command line with another --gpus): bench.py starts its own ranks (here
    sharing the one GPU, collectives over gloo).  One compact contract line,
    last on stdout; the sharded configs[4] extra rides along."""
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR',
                        'MASTER_PORT')}
    env['ZSHMC_DIST_BACKEND'] = 'gloo'
    t0 = time.time()
    r = subprocess.run(
        [sys.executable, 'bench.py', '--gpus', str(world), '--steps', '10',
         '--warmup', '2', '--chains-per-gpu', '4096', '--no-ess',
         '--scaling', scaling,
         '--lntm-chains-per-gpu', '8', '--lntm-docs', '48',
         '--lntm-vocab', '700'], cwd=ROOT, env=env, capture_output=True,
        text=True, timeout=900)
    wall = time.time() - t0
    assert 'no launcher environment, starting %d ranks' % world in r.stderr
    assert r.returncode == 0, r.stderr[-3000:]
    out, extras, detail = _bench_output(r.stdout)
    assert out['n_gpus'] == world and out['steps'] == 10 and out['warmup'] == 2
    # the headline went to stderr before the sharded extra started
    early = [l for l in r.stderr.splitlines()
             if l.startswith('#bench-headline ')]
    assert len(early) == 1
    assert json.loads(early[0][len('#bench-headline '):])['value'] == \
        out['value']
    # configs[4] rides along at N > 1: leading chain axis sharded, literal
    # spelling on the native plan, adaptation on in the timed region
    (extra,) = extras
    assert extra['id'] == 'configs[4] sharded'
    assert extra['plan'] == 'mixture_multinomial' and extra['n_gpus'] == world
    assert 'ONE all-reduce of 258 doubles' in extra['collective']
    assert 0.2 < extra['mean_acceptance'] <= 1.0 and extra['value'] > 0
    (short,) = out['extras']
    assert short['id'] == 'configs[4] sharded' and short['n_gpus'] == world
    assert short['ms_per_step'] == pytest.approx(extra['ms_per_step'],
                                                 rel=1e-4)
    assert out['scaling'] == scaling and out['higher_is_better'] is True
    total = 4096 * world if scaling == 'weak' else 4096
    assert out['config']['n_chains_total'] == total
    assert out['value'] == pytest.approx(
        total * 10 / (out['ms_per_step'] * 1e-3), rel=1e-6)   # L = 10
    assert out['value'] > 0 and 0.3 < out['mean_acceptance'] <= 1.0
    assert out['roofline']['bound'] == 'hbm' and out['roofline']['frac'] > 0
    assert 'cpu_baseline' not in out       # rank 0 at N = 1 only
    if scaling == 'weak':
        assert out['strong_scaling']['chains_per_gpu'] == 4096 // world
        assert detail['strong_scaling']['mean_acceptance'] > 0.3
    assert out['allreduce_latency_us']['max_over_ranks'] > 0
    # far inside the driver's 1 800 s even with every rank on one GPU
    assert wall < 600, wall


def test_bench_one_gpu_prints_one_contract_line():
    """The driver's own N = 1 command on reduced sizes and two of the extras
    (ZSHMC_BENCH_EXTRAS picks them): prefixed extra lines as they finish, the
    detail line, the contract line last with `roofline` and `cpu_baseline`."""
    env = dict(os.environ, ZSHMC_BENCH_EXTRAS='configs[0],lntm-estep')
    r = subprocess.run(
        [sys.executable, 'bench.py', '--gpus', '1', '--steps', '20',
         '--warmup', '5', '--chains-per-gpu', '8192', '--cpu-seconds', '2'],
        cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out, extras, detail = _bench_output(r.stdout)
    assert [e['id'] for e in extras] == ['configs[0]', 'lntm-estep']
    assert [e['id'] for e in out['extras']] == ['configs[0]', 'lntm-estep']
    assert out['n_gpus'] == 1 and out['steps'] == 20 and out['warmup'] == 5
    roof = out['roofline']
    assert roof['bound'] == 'hbm' and roof['peak'] == 8000.0
    assert roof['achieved'] == pytest.approx(
        roof['algorithmic_bytes_per_launch'] / (roof['kernel_ms'] * 1e-3)
        / 1e9)
    assert roof['kernel_ms'] <= out['ms_per_step'] * 1.05
    cpu = out['cpu_baseline']
    assert cpu['kind'] == 'port' and cpu['value'] > 0 and cpu['cores'] >= 1
    assert out['cpu_reference_over_shim']['value'] > 0
    assert detail['cpu_reference_over_shim']['kind'] == 'reference'
    assert detail['mass_adaptation_modes']['overhead_of_adapting'] < 1.0
    assert os.path.exists(os.path.join(ROOT, 'bench_extras.json'))


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


@pytest.mark.parametrize('world', [2, 8])
def test_bench_lntm_workload_ranks(world):
    """`bench.py --workload lntm --gpus N`: BASELINE configs[4] as the line's
    own workload (reduced sizes here), the way the driver would launch it."""
    r = _launch(['bench.py', '--gpus', str(world), '--steps', '3',
                 '--warmup', '1', '--workload', 'lntm',
                 '--lntm-chains-per-gpu', '8', '--lntm-docs', '48',
                 '--lntm-vocab', '700'], world,
                {'ZSHMC_DIST_BACKEND': 'gloo'})
    assert r.returncode == 0, r.stderr[-3000:]
    out, extras, detail = _bench_output(r.stdout)
    assert extras == []
    assert out['n_gpus'] == world and out['steps'] == 3
    assert out['scaling'] == 'weak'
    assert out['plan'] == 'mixture_multinomial'
    assert out['roofline']['bound'] == 'mfma' and out['roofline']['frac'] > 0
    assert out['value'] == pytest.approx(
        world * 8 * 48 * 20 / (out['ms_per_step'] * 1e-3), rel=1e-6)
    assert 0.2 < out['mean_acceptance'] <= 1.0
    assert out['ess']['ess_per_sec'] > 0
    assert 'method' in detail['ess']
