"""The example scripts (ports of the reference's examples/ that exercise the
hot path) run end to end on the device and meet their own statistical
assertions."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv):
    r = subprocess.run([sys.executable] + list(argv), cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_gaussian_example():
    out = _run('examples/diag_gaussian_hmc.py')
    assert 'sampler plan:' in out
    assert 'Relative error of stdev' in out


def test_mixture_sgnht_example():
    out = _run('examples/bimodal_sgnht.py', '--iters', '6000')
    rel = float(out.split('Relative error of stdev = ')[1].split()[0])
    assert abs(rel) < 0.05


def test_logistic_regression_example():
    out = _run('examples/logistic_regression_hmc.py', '--n', '20000',
               '--chains', '512', '--iters', '60')
    assert '|posterior mean - MAP|' in out
    assert 'plan: linear_bernoulli' in out


def test_logistic_regression_example_wide_with_intercept():
    """300 features + a per-chain intercept (two latents): still the native
    plan, on the feature-split MFMA kernel; the example's own check (posterior
    mean near the MAP estimate, intercept included) holds."""
    out = _run('examples/logistic_regression_hmc.py', '--n', '20000',
               '--d', '300', '--bias', '--chains', '512', '--iters', '60')
    assert 'plan: linear_bernoulli' in out
    assert '|posterior mean - MAP|' in out


def test_lntm_example():
    out = _run('examples/topic_model_mcem.py', '--small', '--epochs', '4')
    perp = [float(l.split('Perplexity = ')[1].split(',')[0])
            for l in out.splitlines() if 'Perplexity' in l]
    assert len(perp) == 4 and perp[-1] < 0.6 * perp[0]
    line = [l for l in out.splitlines() if l.startswith('>> Test')][0]
    test_perp = float(line.split('perplexity = ')[1].split()[0])
    assert test_perp < 500      # uniform model: 1000


def test_pmf_example():
    out = _run('examples/matrix_factorization_hmc.py', '--small', '--epochs', '5',
               '--step-size', '0.01')
    tr = [float(l.split('rmse = ')[1]) for l in out.splitlines()
          if 'Train: rmse' in l]
    assert len(tr) == 5 and tr[-1] < 0.8 * tr[0]


@pytest.mark.parametrize('sampler,bound', [('sghmc', 0.40), ('sgld', 0.46)])
def test_bnn_sgmcmc_example(sampler, bound):
    out = _run('examples/bayesian_nn_sgmcmc.py', '--small', '--sampler',
               sampler)
    rmse = [float(l.split('Test rmse = ')[1].split(',')[0])
            for l in out.splitlines() if 'Test rmse' in l]
    # teacher noise 0.3 (SGHMC reaches 0.36 in 8 short epochs, SGLD 0.41)
    assert len(rmse) == 8 and rmse[-1] < rmse[0] and rmse[-1] < bound


def test_plain_c_host_drives_the_fused_transition(tmp_path):
    """examples/c_host/diag_gaussian_hmc.c: a C99 program (gcc, no Python, no
    torch) includes include/zshmc.h, links libzshmc.so and samples the
    gaussian.py target with in-kernel dual averaging -- the C-ABI is the
    boundary, whoever owns the device pointers."""
    exe = str(tmp_path / 'hmc_c')
    lib = os.path.join(ROOT, 'zhusuan_amd', 'lib')
    cmd = ['gcc', '-std=c99', '-O2', '-D__HIP_PLATFORM_AMD__',
           '-I/opt/rocm/include', '-I' + os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'examples', 'c_host', 'diag_gaussian_hmc.c'),
           '-L' + lib, '-lzshmc', '-L/opt/rocm/lib', '-lamdhip64', '-lm',
           '-Wl,-rpath,' + lib, '-Wl,-rpath,/opt/rocm/lib', '-o', exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, '2000', '12', '400'], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]
    out = r.stdout
    acc = float(out.split('mean acceptance ')[1].split(',')[0])
    err = float(out.split('worst relative error of stdev ')[1].split()[0])
    eps = float(out.split('final step size ')[1].split(',')[0])
    assert 0.8 < acc < 0.97 and err < 0.06 and 0.01 < eps < 1.0, out

    # the Python front-end on the same problem, seed and schedule: the same
    # launches with the same arguments, hence the same bits
    import numpy as np
    import torch
    import zhusuan_amd as zs
    dev = torch.device('cuda', 0)
    C, D, n_iters = 2000, 12, 400
    logstd = torch.tensor(-0.125 * np.arange(D, dtype=np.float32), device=dev)

    @zs.meta_bayesian_net()
    def gaussian():
        bn = zs.BayesianNet()
        bn.normal('x', torch.zeros(D, device=dev), logstd=logstd, n_samples=C,
                  group_ndims=1)
        return bn
    flag = zs.placeholder(bool)
    hmc = zs.HMC(step_size=0.05, n_leapfrogs=5, adapt_step_size=flag,
                 target_acceptance_rate=0.9, seed=1234)
    x = torch.zeros(C, D, device=dev)
    op, info = hmc.sample(gaussian(), {}, {'x': x})
    for t in range(1, n_iters + 1):
        op.run(feed_dict={flag: t <= n_iters // 2}, sync=False)
    hmc.check_numerics()
    want = [float.fromhex(v) for v in
            out.split('bits: ')[1].split()[1::2]]
    got = [float(info.updated_step_size.item()), float(x[0, 0]),
           float(x[0, 1]), float(x[-1, -1])]
    assert got == want, (got, want)
