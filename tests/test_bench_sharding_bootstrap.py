"""bench.py's multi-GPU bootstrap ("make the first 8-GPU
contact boring"): when the RCCL communicator cannot be made, EVERY rank exits
with status 3 and rank-tagged JSON on stderr -- a scaling curve over a silent
host-staged fallback would be a curve of the wrong thing -- unless the gloo
path was asked for (ZSHMC_ALLOW_GLOO_FALLBACK=1), in which case the line says
so.  Two gloo ranks on CPU: there is no GPU here, so the attempt fails by
construction (torch.cuda.set_device)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
import bench
from zhusuan_amd.distributed import ChainSharding
dist.init_process_group(backend='gloo', rank=int(os.environ['RANK']),
                        world_size=int(os.environ['WORLD_SIZE']))
sh, note = bench.make_sharding(dist, torch, ChainSharding, 'rccl',
                               torch.device('cpu'), chain_offset=0,
                               n_chains_global=8)
print('NOTE', sh.backend, note)
buf = torch.ones(2, dtype=torch.float64)
sh.all_reduce_sum(buf)
print('SUM', buf.tolist())
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_two_ranks(tmp_path, extra_env):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'root': ROOT})
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2',
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   GLOO_SOCKET_IFNAME='lo')
        env.pop('ZSHMC_ALLOW_GLOO_FALLBACK', None)
        env.update(extra_env)
        procs.append(subprocess.Popen(
            [sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
            stderr=subprocess.PIPE, universal_newlines=True))
    return [p.communicate(timeout=240) + (p.returncode,) for p in procs]


def test_failed_rccl_bootstrap_exits_3_on_every_rank(tmp_path):
    for rank, (out, err, rc) in enumerate(_run_two_ranks(tmp_path, {})):
        assert rc == 3, (rank, rc, err[-600:])
        assert 'NOTE' not in out
        line = [l for l in err.splitlines() if l.startswith('{')][-1]
        msg = json.loads(line)
        assert 'RCCL communicator unavailable on rank %d of 2' % rank \
            in msg['error']
        assert 'ZSHMC_DIST_BACKEND=gloo' in msg['hint']


def test_gloo_fallback_has_to_be_asked_for_and_says_so(tmp_path):
    res = _run_two_ranks(tmp_path, {'ZSHMC_ALLOW_GLOO_FALLBACK': '1'})
    for rank, (out, err, rc) in enumerate(res):
        assert rc == 0, (rank, rc, err[-600:])
        assert 'NOTE torch FALLBACK torch.distributed/gloo' in out, out
        assert 'SUM [2.0, 2.0]' in out


def test_gpus_without_a_launcher_launches_itself():
    """`python bench.py --gpus 2 ...` started as ONE process (the driver's
    N = 1 command with another --gpus) becomes the launcher: two ranks under
    torch.distributed.run on loopback, its exit status the job's.  There is
    no GPU here, so each RANK stops at `bench.py needs an MI355X` -- which
    shows the ranks were started, with the launcher's environment."""
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'),
                        '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=600)
    assert 'no launcher environment, starting 2 ranks' in p.stderr
    assert '--nproc-per-node 2' in p.stderr and '127.0.0.1' in p.stderr
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0
        assert p.stderr.count('bench.py needs an MI355X') >= 2, p.stderr[-800:]
        assert p.stdout.strip() == ''


def test_gpus_inside_a_mismatched_launcher_is_refused():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'),
                        '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=240)
    assert p.returncode != 0
    assert 'inside a launcher environment of WORLD_SIZE=1' in p.stderr
