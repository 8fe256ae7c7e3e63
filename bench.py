#!/usr/bin/env python
"""Headline benchmark: BASELINE.json configs[1] -- 65 536 chains x 1 024-D
diagonal Gaussian, L = 10 leapfrog steps per HMC transition, on N MI355X
(weak scaling: 65 536 chains per GPU, configs[3] at N = 8).

A "step" is one full HMC transition over all chains (momentum resample, L+1
gradient kicks, L drifts, MH accept, step-size update).  State is resident in
HBM before the timed region.  The LAST stdout line is the contract object
(<= 4 KB, strict JSON, the only line that starts with `{`); the full records
travel before it on `#bench-extra` / `#bench-detail` lines and in
bench_extras.json (see "Output" below):

  value     = chain-leapfrog-steps/s = n_chains_total * L * steps / t
  roofline  = algorithmic bytes (8 B per chain-latent per transition:
              read q + write q) / fused-kernel duration, vs 8 TB/s HBM
  cpu_baseline = the C + OpenMP port of the transition on ALL host threads
              (bounded sample, rank 0, N = 1); beside it the op-for-op ports
              (torch-CPU all threads, NumPy one core) and the recorded timing
              of the reference's own hmc.py over the TensorFlow-API shim
  extras    = one short entry per extra configuration (full records on the
              `#bench-extra` lines): BASELINE configs[0], configs[2] (logistic
              regression, 10^6 x 256, 32 768 chains) and configs[4] (topic
              model, 5 000 docs x 128 topics), MFMA-bound, through the native
              plans on HMC's default arithmetic with the other one beside it;
              and, beyond BASELINE.json, wide regressions, softmax
              regressions, the PMF rating model, the topic model at the
              reference's own minibatch size and in its own one-chain layout

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# RCCL across processes: the host driver only supports dmabuf IPC (without
# this, hipIpcGetMemHandle fails with "invalid argument"); normally already
# exported on the GPU boxes
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

N_CHAINS_PER_GPU = 65536
N_DATA = 1024
N_LEAPFROGS = 10
# untimed transitions after the burn-in and the host-side pauses that follow
# it (clock ramp, ~40 ms)
SETTLE = 400
BURN_IN_ADAPT = 50
HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md
ALGO_BYTES_PER_ELEM = 8.0   # read q + write q per transition


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--chains-per-gpu', type=int, default=N_CHAINS_PER_GPU)
    ap.add_argument('--n-data', type=int, default=N_DATA)
    ap.add_argument('--leapfrogs', type=int, default=N_LEAPFROGS)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-ess', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-extra-configs', action='store_true')
    ap.add_argument('--config5-chains', type=int, default=None)
    ap.add_argument('--workload', choices=('gaussian', 'lntm'),
                    default='gaussian',
                    help="gaussian: configs[1]/[3] (the headline); lntm: "
                         "configs[4], chains sharded over --gpus ranks")
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help="weak: --chains-per-gpu chains on every GPU "
                         "(configs[3] at N = 8); strong: --chains-per-gpu "
                         "chains IN TOTAL, split over the ranks (SURVEY 8d "
                         "c4's fixed-total mode); at N > 1 the weak line also "
                         "carries a `strong_scaling` block")
    ap.add_argument('--lntm-chains-per-gpu', type=int, default=1024)
    ap.add_argument('--lntm-docs', type=int, default=5000)
    ap.add_argument('--lntm-vocab', type=int, default=12419)
    return ap.parse_args()


def cpu_baseline_numpy(n_data, n_leapfrogs, budget_s):
    """Time the NumPy restatement of zhusuan/hmc.py (oracle/hmc_ref.py) on
    C = 4096 chains of the same target; throughput is size-independent once
    DRAM-bound."""
    from oracle.hmc_ref import HMC as RefHMC, DiagNormalModel
    C = 4096
    logstd = np.linspace(-1.0, 1.0, n_data).astype(np.float32)
    model = DiagNormalModel(np.zeros(n_data, np.float32), logstd=logstd)
    x = np.zeros((C, n_data), np.float32)
    ref = RefHMC(step_size=0.05, n_leapfrogs=n_leapfrogs, seed=1)
    ref.sample(model.log_joint, model.grad, [x])
    ref.step()                       # warm-up (allocations, page faults)
    iters = 0
    t0 = time.perf_counter()
    while True:
        ref.step()
        iters += 1
        el = time.perf_counter() - t0
        if el >= budget_s or iters >= 400:
            break
    return {
        'value': C * n_leapfrogs * iters / el,
        'unit': 'chain-leapfrog-steps/s',
        'cores': 1,
        'host_cores_available': os.cpu_count(),
        'kind': 'port',
        'sample': '%d chains x %d latents, L=%d, %d transitions in %.1f s '
                  '(NumPy float32 restatement of zhusuan/hmc.py, same pass '
                  'structure as the TF graph; TF itself is not installable '
                  'in this image)' % (C, n_data, n_leapfrogs, iters, el),
        'elem_leapfrog_steps_per_sec': C * n_data * n_leapfrogs * iters / el,
    }


def cpu_baseline_torch(n_data, n_leapfrogs, budget_s):
    """The same transition op for op on torch-CPU with every host thread
    (oracle/hmc_torch_port.py): each TensorFlow op of the reference's graph is
    one multi-threaded full-array pass -- the nearest runnable stand-in for
    "TF-CPU with intra-op parallelism" on a box without TensorFlow."""
    import torch
    from oracle import hmc_torch_port
    threads = os.cpu_count() or 1
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        C = 16384
        logstd = torch.linspace(-1.0, 1.0, n_data)
        mean = torch.zeros(n_data)
        q = torch.randn(C, n_data) * torch.exp(logstd)
        hmc_torch_port.transition(q, mean, logstd, 0.14, n_leapfrogs,
                                  torch.randn(C, n_data), torch.rand(C))
        iters = 0
        t0 = time.perf_counter()
        while True:
            hmc_torch_port.transition(q, mean, logstd, 0.14, n_leapfrogs,
                                      torch.randn(C, n_data), torch.rand(C))
            iters += 1
            el = time.perf_counter() - t0
            if el >= budget_s or iters >= 2000:
                break
    finally:
        torch.set_num_threads(old)
    return {
        'value': C * n_leapfrogs * iters / el,
        'unit': 'chain-leapfrog-steps/s',
        'cores': threads,
        'kind': 'port',
        'sample': '%d chains x %d latents, L=%d, %d transitions in %.1f s '
                  '(torch-CPU, one multi-threaded pass per TensorFlow op of '
                  'the reference graph, %d threads)' % (
                      C, n_data, n_leapfrogs, iters, el, threads),
    }


def cpu_reference_recorded():
    """The reference's OWN hmc.py timed over the TensorFlow-API shim
    (tools/time_reference_over_shim.py).  It needs /root/reference, which does
    not exist on the GPU box, so the number is a RECORDED one and says where
    it was measured."""
    return _recorded('cpu_reference_over_shim.json')


def _recorded(name):
    """A committed record under profiles/.  A missing or unreadable file is
    an error in the record (and a line on stderr), never a silent null."""
    path = os.path.join(ROOT, 'profiles', name)
    try:
        with open(path) as f:
            return json.load(f)
    except Exception as e:                           # noqa: BLE001
        sys.stderr.write('bench.py: recorded figure %s unreadable: %r\n' % (
            path, e))
        return {'error': 'profiles/%s unreadable: %r' % (name, e)}


def cpu_baseline_parallel(n_data, n_leapfrogs, budget_s):
    """The same transition as a chain-fused C + OpenMP restatement
    (oracle/c/hmc_diag_normal_port.c, held to the NumPy oracle by
    tests/test_oracle_c_port.py) on ALL host cores: the strongest CPU
    formulation of the path, next to the op-for-op NumPy port above."""
    from oracle import hmc_c
    threads = hmc_c.max_threads()
    C = 256 * max(threads, 1)
    logstd = np.linspace(-1.0, 1.0, n_data).astype(np.float32)
    mean = np.zeros(n_data, np.float32)
    q = (np.random.RandomState(0).normal(size=(C, n_data)) *
         np.exp(logstd)).astype(np.float32)
    hmc_c.step(q, mean, logstd, n_leapfrogs, 0.14, 1, 0, want_info=False)
    iters = 0
    t0 = time.perf_counter()
    while True:
        hmc_c.step(q, mean, logstd, n_leapfrogs, 0.14, 1, iters + 1,
                   want_info=False)
        iters += 1
        el = time.perf_counter() - t0
        if el >= budget_s or iters >= 2000:
            break
    return {
        'value': C * n_leapfrogs * iters / el,
        'unit': 'chain-leapfrog-steps/s',
        'cores': threads,
        'kind': 'port',
        'sample': '%d chains x %d latents, L=%d, %d transitions in %.1f s '
                  '(C + OpenMP restatement, one chain\'s trajectory kept in '
                  'cache, %d threads)' % (C, n_data, n_leapfrogs, iters, el,
                                          threads),
    }


MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 MFMA, MI355X_MICROARCH.md
L2_PEAK_GBPS = 34500.0         # aggregate L2, MI355X_MICROARCH.md
EXTRA_TIMEOUT_S = 420          # N > 1: budget of the sharded configs[4] extra
TUNED_START_NOTE = (
    "timed AFTER the reference's dual-averaging start-up transient (mu = "
    "10*eps0 used as a LOG step size, hmc.py:79 sic: eps jumps to ~1 after the "
    "first adapted iteration and acceptance is ~0 until ~iteration 9): a "
    "subset of the chains is burnt in with adaptation on, its sampler state "
    "(t, step size, tuner triple, EWMV mean/var, mass) is restored into the "
    "full-size sampler with set_state and its end state tiled over the full "
    "chain axis; the timed transitions then run with adaptation ON (tuner and "
    "mass estimator continue from the restored state; the all-reduce of "
    "[sum acc, flag, colsum] is in the loop when sharded).")


def _tuned_start(torch, zs, build, n_sub, n_burn, n_draws, flags_on):
    """Burn a subset of `n_sub` chains in (adaptation on), then record
    `n_draws` more transitions (adaptation held) for the reference's ESS
    estimator.  Returns (sampler state, end state of the subset, ESS per
    (chain row) per transition, mean acceptance of the recorded phase)."""
    hmc, op, info, q, flags = build(n_sub, None)
    feed_on = dict(zip(flags, flags_on))
    for _ in range(n_burn):
        op.run(feed_dict=feed_on, sync=False)
    hmc.check_numerics()
    feed_off = {f: False for f in flags}
    rec = torch.empty((n_draws,) + tuple(q.shape), device=q.device)
    acc = torch.zeros((), device=q.device)
    for i in range(n_draws):
        op.run(feed_dict=feed_off, sync=False)
        rec[i].copy_(q)
        acc += info.acceptance_rate.mean()
    acc = float(acc.item()) / n_draws
    # the state AFTER the held phase: step size = exp(log_epsilon_bar), the
    # dual-averaged one (hmc.py:108-110), not the last noisy iterate
    state = hmc.get_state()
    q_end = q.clone()
    burn = n_draws // 3
    ess = zs.diagnostics.effective_sample_size_device(rec, burn_in=burn)
    ok = torch.isfinite(ess)
    ess_per_transition = float(ess[ok].mean().item()) / (n_draws - burn)
    del rec
    return state, q_end, ess_per_transition, acc


def _time_transitions(torch, hmc, op, info, feed, n_warm, n_timed, barrier):
    """Wall time per transition (host clock around a barrier-bracketed
    region; the contract's max over ranks is taken by the caller) and the
    likelihood kernel alone (one evaluation = likelihood + gradient)."""
    for _ in range(n_warm):
        op.run(feed_dict=feed, sync=False)
    hmc.check_numerics()
    acc = torch.zeros((), device=hmc._plan.device)
    barrier()
    t0 = time.perf_counter()
    for _ in range(n_timed):
        op.run(feed_dict=feed, sync=False)
        acc += info.acceptance_rate.mean()     # (device-side, no sync)
    barrier()
    elapsed = time.perf_counter() - t0
    acc = float(acc.item()) / n_timed          # mean over the timed region
    plan = hmc._plan
    stream = torch.cuda.current_stream().cuda_stream
    # the likelihood kernel alone, in the two forms a transition launches:
    # gradient only (the L - 1 interior evaluations of a trajectory: the
    # dominant launch) and likelihood + gradient (its two ends)
    kern = {}
    for name, want_ll in (('grad', False), ('ll_grad', True)):
        plan._likelihood(plan.q_new, stream, want_ll=want_ll)
        k0 = torch.cuda.Event(enable_timing=True)
        k1 = torch.cuda.Event(enable_timing=True)
        reps = 3
        k0.record()
        for _ in range(reps):
            plan._likelihood(plan.q_new, stream, want_ll=want_ll)
        k1.record()
        torch.cuda.synchronize()
        kern[name] = k0.elapsed_time(k1) / reps
    return elapsed, kern, acc


def _recorded_mfma_traffic(key):
    """HBM bytes per launch of a likelihood kernel at its FULL BASELINE shape
    from the recorded rocprofv3 PMC passes (profiles/pmc_traffic_mfma.json;
    bench.py cannot run the profiler on itself): (bytes, source) or (None,
    None).  Keys: 'bernoulli grad-only', 'multinomial bf16x3 grad-only', ..."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic_mfma.json')
    try:
        rec = json.load(open(path))
        return rec['hbm_bytes_per_launch'][key], rec['source']
    except Exception:                                    # noqa: BLE001
        return None, None


def _lik_kernel_name(width, block):
    """The likelihood kernel the library dispatches for a plan's width and
    chain block (zshmc_likelihood_plan)."""
    if width <= 256:
        return 'linear_bernoulli_kernel<%d>' % width
    if block == 64:
        return 'linear_bernoulli_mid_kernel<%d>' % width
    return 'linear_bernoulli_wide_kernel<%d>' % width


def _mfma_roofline(kernel, kern_ms, flop_eval, n_evals, ms_transition):
    """`kern_ms`: {'grad': ms, 'll_grad': ms} of _time_transitions (or one
    number).  The roofline entry is the gradient-only launch -- n_evals - 1 of
    a transition's n_evals = L likelihood launches (the evaluation at the
    start point is the previous transition's last one, carried over where the
    chain accepted: zshmc_model_plan.grad_start); the likelihood + gradient
    launch of the trajectory's end is reported beside it."""
    both = kern_ms if isinstance(kern_ms, dict) else {'grad': kern_ms}
    ms = both['grad']
    ach = flop_eval / (ms * 1e-3) / 1e12
    out = {
        'bound': 'mfma', 'dtype': 'f32', 'peak': MFMA_F32_PEAK_TFLOPS,
        'unit': 'TFLOP/s', 'kernel': kernel + ' gradient only (log_lik = NULL)',
        'kernel_ms': ms, 'achieved': ach, 'frac': ach / MFMA_F32_PEAK_TFLOPS,
        'traffic': None,
        'algorithmic_flop_per_launch': flop_eval,
        'launches_per_transition': {'gradient_only': n_evals - 1,
                                    'likelihood_and_gradient': 1},
        'sustained_over_transition': n_evals * flop_eval /
        (ms_transition * 1e-3) / 1e12,
    }
    if 'll_grad' in both:
        ach2 = flop_eval / (both['ll_grad'] * 1e-3) / 1e12
        out['likelihood_and_gradient'] = {
            'kernel_ms': both['ll_grad'], 'achieved': ach2,
            'frac': ach2 / MFMA_F32_PEAK_TFLOPS}
    out['sustained_frac'] = out['sustained_over_transition'] / \
        MFMA_F32_PEAK_TFLOPS
    return out


MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md


def _b3_roofline(width, kern_ms, flop_eval, n_evals, ms_transition):
    """The same launches on the bf16x3 kernels (csrc/b3_kernel.h): every
    float32 operand as three bfloat16 planes, a product = six bf16 MFMAs with
    float32 accumulation.  `achieved` counts the ALGORITHMIC (float32-
    equivalent) flops 4 N D C; the matrix cores issue six times that.  Two
    fractions: against the fp32-MFMA peak the exact-fp32 kernels are priced
    against (> 1: that is the point), and against the ceiling of this
    arithmetic, the dense bf16 peak / 6."""
    ceiling = MFMA_BF16_PEAK_TFLOPS / 6.0
    ms = kern_ms['grad']
    ach = flop_eval / (ms * 1e-3) / 1e12
    ach2 = flop_eval / (kern_ms['ll_grad'] * 1e-3) / 1e12
    sus = n_evals * flop_eval / (ms_transition * 1e-3) / 1e12
    return {
        'bound': 'mfma',
        'dtype': 'bf16x3 (3 bf16 planes per f32 operand, 6-term products, '
                 'fp32 accumulate)',
        'kernel': 'linear_b3_kernel<%d> gradient only (log_lik = NULL)' % width,
        'kernel_ms': ms, 'achieved': ach, 'unit': 'TFLOP/s',
        'achieved_counts': 'algorithmic fp32-equivalent flops (4 N D C per '
                           'launch); bf16 flops issued = 6x',
        'peak': ceiling, 'frac': ach / ceiling,
        'peak_is': 'dense bf16 MFMA peak %.0f / 6 terms' % MFMA_BF16_PEAK_TFLOPS,
        'frac_of_fp32_mfma_peak': ach / MFMA_F32_PEAK_TFLOPS,
        'traffic': None,
        'algorithmic_flop_per_launch': flop_eval,
        'launches_per_transition': {'gradient_only': n_evals - 1,
                                    'likelihood_and_gradient': 1},
        'likelihood_and_gradient': {
            'kernel_ms': kern_ms['ll_grad'], 'achieved': ach2,
            'frac': ach2 / ceiling,
            'frac_of_fp32_mfma_peak': ach2 / MFMA_F32_PEAK_TFLOPS},
        'sustained_over_transition': sus,
        'sustained_frac': sus / ceiling,
        'sustained_frac_of_fp32_mfma_peak': sus / MFMA_F32_PEAK_TFLOPS,
    }


def _lik_roofline(hmc, kern_ms, flop_eval, n_evals, ms_transition, mode=''):
    """The roofline entry of a dense-likelihood plan's launches, priced
    against the peak of the arithmetic that RAN (hmc.likelihood_arithmetic_
    used): dense bf16 / 6 for bf16x3, the fp32 MFMA peak otherwise."""
    plan = hmc._plan
    if hmc.likelihood_arithmetic_used == 'bf16x3':
        r = _b3_roofline(plan.width, kern_ms, flop_eval, n_evals,
                         ms_transition)
        if mode:
            r['kernel'] = r['kernel'].replace('>', ', %s>' % mode, 1)
        return r
    return _mfma_roofline(
        _lik_kernel_name(plan.width, plan.block) +
        (' (%s mode)' % mode if mode else ''), kern_ms, flop_eval, n_evals,
        ms_transition)


def _other_arithmetic(used):
    return 'fp32' if used == 'bf16x3' else 'bf16x3'


ARITH_NOTE = ("HMC's default likelihood_arithmetic='auto' (bf16x3 where a "
              "kernel exists and an evaluation is >= 1e10 flop, else fp32); "
              "the other arithmetic on the same chains, state and step size "
              "beside it")


def extra_config1(torch, zs, dev, n_chains=1000, n_x=10, n_leapfrogs=5):
    """BASELINE configs[0]: examples/toy_examples/gaussian.py (:29, :36-58):
    1 000 chains, 10-D, stdev_j = 1/(j+1), L = 5, target acceptance 0.9, step
    size and mass adapting for the first 50 of 200 iterations, the last 100
    kept.  On the device the kernel is a few microseconds: the line is about
    the front-end -- one Python call per transition against one
    `run_many` call for a stretch."""
    stdev = 1.0 / (torch.arange(n_x, device=dev, dtype=torch.float32) + 1.0)

    @zs.meta_bayesian_net()
    def gaussian():
        bn = zs.BayesianNet()
        bn.normal('x', torch.zeros(n_x, device=dev), std=stdev,
                  n_samples=n_chains, group_ndims=1)
        return bn
    out = {}
    for mode in ('python_loop', 'run_many'):
        x = torch.zeros(n_chains, n_x, device=dev)
        f_ss, f_m = zs.placeholder(bool), zs.placeholder(bool)
        hmc = zs.HMC(step_size=1e-3, n_leapfrogs=n_leapfrogs,
                     adapt_step_size=f_ss, adapt_mass=f_m,
                     target_acceptance_rate=0.9, seed=1)
        op, info = hmc.sample(gaussian(), {}, {'x': x})
        on, off = {f_ss: True, f_m: True}, {f_ss: False, f_m: False}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == 'run_many':
            op.run_many(50, feed_dict=on, sync=False)
            op.run_many(50, feed_dict=off, sync=False)
        else:
            for i in range(100):
                op.run(feed_dict=on if i < 50 else off, sync=False)
        hmc.check_numerics()
        t_burn = time.perf_counter() - t0
        # the recorded phase of gaussian.py, lengthened to 2 000 transitions
        n = 2000
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == 'run_many':
            op.run_many(n, feed_dict=off, sync=False)
        else:
            for _ in range(n):
                op.run(feed_dict=off, sync=False)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out[mode] = {
            'transitions_per_sec': n / el,
            'us_per_transition': el / n * 1e6,
            'chain_leapfrog_steps_per_sec': n_chains * n_leapfrogs * n / el,
            'burn_in_100_transitions_ms': t_burn * 1e3,
            'mean_acceptance': float(info.acceptance_rate.mean().item()),
            'sample_std_over_stdev': float(
                (x.std(0) / stdev).mean().item()),
        }
    out['workload'] = ('configs[0]: gaussian.py, %d chains x %d-D, L=%d, '
                       'delta 0.9; 2 000 non-adaptive transitions after 50 '
                       'adaptive (step size + mass) + 50 held' % (
                           n_chains, n_x, n_leapfrogs))
    out['kernel'] = _capi_kernel_name(n_x, 1, 1)
    # the reference's OWN hmc.py over the TensorFlow-API shim on this shape: a
    # RECORDED number (its sources do not travel to the GPU box)
    out['cpu_reference_over_shim'] = _recorded(
        'cpu_reference_over_shim_config1.json')
    return out


def extra_config3(torch, zs, dev, n_rows=1000000, n_chains=32768, n_feat=256,
                  n_leapfrogs=10, n_sub=256, n_timed=2):
    """BASELINE configs[2]: Bayesian logistic regression, synthetic
    10^6 x 256 design matrix, 32 768 chains, L = 10: native
    plan = fused fp32-MFMA likelihood + csrc/hmc_model.hip, step-size
    adaptation on, timed after the start-up transient."""
    # the data of SURVEY 8d c3: X ~ N(0,1), w* ~ N(0,1), y ~ Bernoulli(
    # sigmoid(X w* / sqrt(D))) from numpy.random.default_rng(0)
    rng = np.random.default_rng(0)
    X_h = rng.standard_normal((n_rows, n_feat), dtype=np.float32)
    w_h = rng.standard_normal(n_feat).astype(np.float32)
    y_h = rng.random(n_rows) < 1.0 / (1.0 + np.exp(
        -(X_h @ w_h) / np.float32(n_feat ** 0.5)))
    X = torch.from_numpy(X_h).to(dev)
    w_true = torch.from_numpy(w_h).to(dev)
    y = torch.from_numpy(y_h.astype(np.float32)).to(dev)
    del X_h, y_h
    zero, one = torch.zeros(n_feat, device=dev), torch.ones(n_feat, device=dev)

    def build(n, sharding, arithmetic=None):
        @zs.meta_bayesian_net()
        def blr():
            bn = zs.BayesianNet()
            w = bn.normal('w', zero, std=one, n_samples=n, group_ndims=1)
            bn.bernoulli('y', w.tensor @ X.t(), group_ndims=1,
                         dtype=torch.float32)
            return bn
        # chains start at the data-generating weights (inside the posterior's
        # bulk: at N = 10^6 its width is ~2e-3, and from w = 0 the
        # reference's step-size search accepts any step that runs uphill)
        w = (w_true / n_feat ** 0.5).repeat(n, 1).contiguous()
        flag = zs.placeholder(bool)
        kw = {} if arithmetic is None else {'likelihood_arithmetic': arithmetic}
        hmc = zs.HMC(step_size=1e-3, n_leapfrogs=n_leapfrogs,
                     adapt_step_size=flag, target_acceptance_rate=0.8, seed=2,
                     sharding=sharding, **kw)
        op, info = hmc.sample(blr(), {'y': y}, {'w': w})
        return hmc, op, info, w, (flag,)

    state, w_sub, ess_pt, acc_sub = _tuned_start(
        torch, zs, build, n_sub, 60, 240, (True,))

    def barrier():
        torch.cuda.synchronize()
    flop_eval = 4.0 * n_rows * n_feat * n_chains
    full = (n_rows, n_chains, n_feat) == (1000000, 32768, 256)

    def timed(arithmetic):
        # adaptation held in the timed region: step size = the dual-averaged
        # one of the subset
        hmc, op, info, w, flags = build(n_chains, None, arithmetic)
        w.copy_(w_sub.repeat(n_chains // n_sub, 1))
        hmc.set_state(state)
        elapsed, kern_ms, acc = _time_transitions(
            torch, hmc, op, info, {flags[0]: False}, 1, n_timed, barrier)
        ms = elapsed / n_timed * 1e3
        used = hmc.likelihood_arithmetic_used
        r = {
            'likelihood_arithmetic_used': used,
            'plan': hmc.plan_kind,
            'ms_per_step': ms, 'steps': n_timed,
            'value': n_chains * n_leapfrogs / (ms * 1e-3),
            'unit': 'chain-leapfrog-steps/s',
            'mean_acceptance': acc,
            'step_size': float(info.updated_step_size.item()),
            'roofline': _lik_roofline(hmc, kern_ms, flop_eval, n_leapfrogs,
                                      ms),
        }
        if used == 'bf16x3':
            r['image_bytes'] = int(hmc._plan.inner_image.numel())
        if full:
            r['roofline']['traffic'], r['roofline']['traffic_source'] = \
                _recorded_mfma_traffic('bernoulli bf16x3 grad-only'
                                       if used == 'bf16x3' else
                                       'bernoulli grad-only')
        del hmc, op, info, w
        gc.collect()
        return r
    main = timed(None)                      # the default: 'auto'
    other = timed(_other_arithmetic(main['likelihood_arithmetic_used']))
    other['transition_time_over_default'] = \
        other['ms_per_step'] / main['ms_per_step']
    ms = main['ms_per_step']
    out = dict(main)
    out.update({
        'workload': 'configs[2]: Bayesian logistic regression, synthetic '
                    '%d x %d, %d chains, L=%d, step size adapted on a subset '
                    'and held in the timed region, the '
                    'model written with the reference\'s literal '
                    '`w @ X.T` logits' % (
                        n_rows, n_feat, n_chains, n_leapfrogs),
        'likelihood_arithmetic': ARITH_NOTE,
        'mean_acceptance_subset_held_phase': acc_sub,
        'target_acceptance': 0.8,
        other['likelihood_arithmetic_used']: other,
        'start': TUNED_START_NOTE.replace(
            'run with adaptation ON', 'run with adaptation HELD (config 3)') +
                 ' Subset: %d chains, 60 adaptive + 240 recorded transitions '
                 '(mean acceptance %.3f).' % (n_sub, acc_sub),
        'ess': {
            'ess_per_chain_per_transition': ess_pt,
            'ess_per_sec': ess_pt * n_chains * 1e3 / ms,
            'method': 'zhusuan.diagnostics estimator (min over dims per '
                      'chain, mean over chains) on the %d-chain subset run '
                      'with the same step size, scaled to %d chains at the '
                      'timed rate' % (n_sub, n_chains),
        },
    })
    return out


def extra_wide_regression(torch, zs, dev, n_rows=65536, n_chains=8192,
                          n_feat=1000, n_leapfrogs=10, n_warm=4, n_timed=3):
    """Beyond BASELINE.json: logistic regression with n_feat features AND a
    per-chain bias, written `w @ X.T + b[:, None]`: two Normal-prior latents on
    the native plan's packed state, padded to the kernel width the library
    names (zshmc_likelihood_plan).  1 000 features: rows of 1 004 floats on
    the feature-split MFMA kernel (csrc/linear_bernoulli_wide.hip, width
    1 024); 299 features: the 16-chain-block kernel at width 320
    (csrc/linear_bernoulli_mid.hip; round 3 padded these to 512)."""
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(n_rows, n_feat, device=dev, generator=g)
    w_true = torch.randn(n_feat, device=dev, generator=g)
    y = (torch.rand(n_rows, device=dev, generator=g) < torch.sigmoid(
        X @ w_true / n_feat ** 0.5 + 0.3)).float()
    zero, one = torch.zeros(n_feat, device=dev), torch.ones(n_feat, device=dev)

    @zs.meta_bayesian_net()
    def blr():
        bn = zs.BayesianNet()
        w = bn.normal('w', zero, std=one, n_samples=n_chains, group_ndims=1)
        b = bn.normal('b', torch.zeros((), device=dev), std=2.,
                      n_samples=n_chains)
        bn.bernoulli('y', w.tensor @ X.t() + b.tensor[:, None], group_ndims=1,
                     dtype=torch.float32)
        return bn
    w = (w_true / n_feat ** 0.5).repeat(n_chains, 1).contiguous()
    b = torch.full((n_chains,), 0.3, device=dev)
    # fixed step size ~ posterior width (1/sqrt(N)) * D^(-1/4)
    eps = 1.0 / n_rows ** 0.5 / n_feat ** 0.25
    hmc = zs.HMC(step_size=eps, n_leapfrogs=n_leapfrogs, seed=4)
    op, info = hmc.sample(blr(), {'y': y}, {'w': w, 'b': b})
    elapsed, kern_ms, acc = _time_transitions(
        torch, hmc, op, info, {}, n_warm, n_timed, torch.cuda.synchronize)
    ms = elapsed / n_timed * 1e3
    width = hmc._plan.width
    flop_eval = 4.0 * n_rows * width * n_chains
    return {
        'workload': 'beyond BASELINE.json: logistic regression, %d features + '
                    'a per-chain bias (two latents, literal `w @ X.T + '
                    'b[:, None]`), synthetic %d rows, %d chains, L=%d, fixed '
                    'step size %.2e' % (n_feat, n_rows, n_chains, n_leapfrogs,
                                        eps),
        'plan': hmc.plan_kind,
        'packed_row_floats': hmc._plan.ld,
        'ms_per_step': ms,
        'steps': n_timed,
        'value': n_chains * n_leapfrogs / (ms * 1e-3),
        'unit': 'chain-leapfrog-steps/s',
        'mean_acceptance': acc,
        'roofline': dict(_mfma_roofline(
            _lik_kernel_name(hmc._plan.width, hmc._plan.block), kern_ms, flop_eval,
            n_leapfrogs, ms),
            note='flops counted at the padded width %d (%d useful columns)'
                 % (width, n_feat + 1),
            useful_flop_fraction=(n_feat + 1.0) / width),
    }


def extra_softmax_regression(torch, zs, dev, n_rows=60000, n_feat=784,
                             n_classes=10, n_chains=1024, n_leapfrogs=10,
                             n_warm=3, n_timed=3, bf16x3=False):
    """north_star's third likelihood: softmax regression
    at the MNIST shape -- 10 classes x 784 features per chain, 60 000 rows --
    written with the reference's literal `matmul(X, w, transpose_b=True)` under
    a Categorical (univariate.py:496-548): the native 'linear_categorical'
    plan -- the two-GEMM fp32-MFMA kernel with the class softmax over
    accumulator lanes (csrc/lb_ops.h; (chain, class) pairs as its rows, class
    stride 16, padded width 1 024) + csrc/hmc_model_seg.hip."""
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(n_rows, n_feat, device=dev, generator=g)
    w_true = torch.randn(n_classes, n_feat, device=dev, generator=g)
    y = torch.argmax(X @ w_true.t() / n_feat ** 0.5 - torch.log(-torch.log(
        torch.rand(n_rows, n_classes, device=dev, generator=g))), -1).to(
            torch.int32)
    zero = torch.zeros(n_classes, n_feat, device=dev)

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        w = bn.normal('w', zero, std=1., n_samples=n_chains, group_ndims=2)
        bn.categorical('y', X.unsqueeze(0) @ w.tensor.transpose(-1, -2),
                       group_ndims=1)
        return bn
    eps = 0.5 / n_rows ** 0.5 / (n_feat * n_classes) ** 0.25

    def timed(arithmetic):
        w = (w_true / n_feat ** 0.5).unsqueeze(0).repeat(n_chains, 1,
                                                         1).contiguous()
        kw = {} if arithmetic is None else {'likelihood_arithmetic': arithmetic}
        hmc = zs.HMC(step_size=eps, n_leapfrogs=n_leapfrogs, seed=6, **kw)
        op, info = hmc.sample(model(), {'y': y}, {'w': w})
        elapsed, kern_ms, acc = _time_transitions(
            torch, hmc, op, info, {}, n_warm, n_timed, torch.cuda.synchronize)
        ms = elapsed / n_timed * 1e3
        plan = hmc._plan
        flop_eval = 4.0 * n_rows * plan.width * plan.lik_rows
        useful = (n_feat / plan.width) * (n_classes / plan.stride)
        roof = _lik_roofline(hmc, kern_ms, flop_eval, n_leapfrogs, ms,
                             'Categorical')
        roof['note'] = (
            'flops counted at the padded shape: width %d, class stride %d; '
            'useful fraction of them %.3f (%d features, %d classes)' % (
                plan.width, plan.stride, useful, n_feat, n_classes))
        roof['useful_fraction'] = useful
        r = {'likelihood_arithmetic_used': hmc.likelihood_arithmetic_used,
             'arithmetic_reason': hmc.arithmetic_reason,
             'plan': hmc.plan_kind, 'plan_reason': hmc.plan_reason,
             'ms_per_step': ms, 'steps': n_timed,
             'value': n_chains * n_leapfrogs / (ms * 1e-3),
             'unit': 'chain-leapfrog-steps/s', 'mean_acceptance': acc,
             'roofline': roof}
        del hmc, op, info, plan, w
        gc.collect()
        return r
    main = timed(None)                      # the default: 'auto'
    out = dict(main)
    out['workload'] = (
        'beyond BASELINE.json (north_star\'s Categorical): softmax '
        'regression, %d classes x %d features, synthetic %d rows, %d chains, '
        'L=%d, literal X @ w^T, fixed step size %.2e' % (
            n_classes, n_feat, n_rows, n_chains, n_leapfrogs, eps))
    out['likelihood_arithmetic'] = ARITH_NOTE
    if bf16x3 and main['likelihood_arithmetic_used'] == 'bf16x3':
        # the same chains, state and step size on the exact-fp32 kernels
        other = timed('fp32')
        other['transition_time_over_default'] = \
            other['ms_per_step'] / main['ms_per_step']
        out['fp32'] = other
    return out


def extra_pmf(torch, zs, dev, n_particles=8, n_users=6040, n_items=3706,
              n_factors=32, n_pairs=1000000, n_leapfrogs=10, n_warm=3,
              n_timed=5):
    """The rating model of pmf_hmc.py:19-31 at the MovieLens-1M shape (6 040
    users x 3 706 items, 10^6 ratings, 8 particles; 32 factors): HMC over the
    user table given the item table on the native 'gathered_dot' plan
    (csrc/gather_dot.hip + csrc/hmc_model_seg.hip: no autograd graph).
    HBM / L2-bound irregular access: the roofline entry is gathered bytes per
    second of the rating-likelihood launch."""
    g = torch.Generator(device=dev).manual_seed(0)
    su = torch.randint(0, n_users, (n_pairs,), device=dev, generator=g,
                       dtype=torch.int32)
    sv = torch.randint(0, n_items, (n_pairs,), device=dev, generator=g,
                       dtype=torch.int32)
    r = torch.rand(n_pairs, device=dev, generator=g)
    v = 0.3 * torch.randn(n_particles, n_items, n_factors, device=dev,
                          generator=g)
    zu = torch.zeros(n_users, n_factors, device=dev)
    zv = torch.zeros(n_items, n_factors, device=dev)

    @zs.meta_bayesian_net(scope='pmf', reuse_variables=True)
    def pmf():
        bn = zs.BayesianNet()
        u = bn.normal('u', zu, std=1.0, n_samples=n_particles, group_ndims=1)
        vv = bn.normal('v', zv, std=1.0, n_samples=n_particles, group_ndims=1)
        bn.normal('r', torch.sigmoid(zs.gathered_dot(u, su, vv, sv)),
                  std=0.25)
        return bn
    model = pmf()
    model.log_joint = lambda bn: (
        bn.cond_log_prob('u').sum(-1) + bn.cond_log_prob('v').sum(-1) +
        bn.cond_log_prob('r').sum(-1))
    u = 0.1 * torch.randn(n_particles, n_users, n_factors, device=dev,
                          generator=g)
    hmc = zs.HMC(step_size=2e-3, n_leapfrogs=n_leapfrogs, seed=7)
    op, info = hmc.sample(model, {'r': r, 'v': v}, {'u': u})
    elapsed, kern_ms, acc = _time_transitions(
        torch, hmc, op, info, {}, n_warm, n_timed, torch.cuda.synchronize)
    ms = elapsed / n_timed * 1e3
    kern_ms = kern_ms['ll_grad']     # (this plan's launch always forms both)
    # Algorithmic bytes of one evaluation (likelihood + gradient): per
    # (particle, pair) ONE gathered row of the other table (4 D bytes: the
    # fused kernel uses it for the dot product and for the scatter), its index
    # and rating (8 bytes, shared by the particles of a wave), plus the latent
    # table read and its gradient written once.  The tables (10 MB) are L2 /
    # Infinity-Cache resident: the bound is the L2 gather rate, not HBM.
    fused = bool(getattr(hmc._plan, 'gd_fused', False))
    rows = (1.0 if fused else 3.0) * n_factors * 4
    gathered = n_particles * n_pairs * rows + n_pairs * 8.0 + \
        2.0 * n_particles * n_users * n_factors * 4
    ach = gathered / (kern_ms * 1e-3) / 1e9
    return {
        'workload': 'beyond BASELINE.json: the rating model of pmf_hmc.py at '
                    'the MovieLens-1M shape (%d x %d, %d ratings, %d '
                    'particles, %d factors), HMC over the user table, L=%d'
                    % (n_users, n_items, n_pairs, n_particles, n_factors,
                       n_leapfrogs),
        'plan': hmc.plan_kind, 'plan_reason': hmc.plan_reason,
        'ms_per_step': ms, 'steps': n_timed,
        'value': n_particles * n_leapfrogs / (ms * 1e-3),
        'unit': 'chain-leapfrog-steps/s',
        'mean_acceptance': acc,
        'roofline': {
            'bound': 'l2-gather', 'unit': 'GB/s', 'peak': L2_PEAK_GBPS,
            'peak_is': 'aggregate L2 rate of MI355X_MICROARCH.md (34.5 TB/s); '
                       'the factor tables are cache-resident, HBM sees ~80 MB',
            'kernel': 'gd_fused_kernel (likelihood + gradient in one pass '
                      'over the segmented CSR pair list)' if fused else
                      'gather_dot_normal_lik_kernel + gather_dot_grad_kernel',
            'kernel_ms': kern_ms, 'achieved': ach,
            'frac': ach / L2_PEAK_GBPS,
            'traffic': None,
            'algorithmic_bytes_per_evaluation': gathered,
            'note': 'round 4 (two kernels, three gathers per pair): 0.811 ms '
                    'per evaluation at this shape',
        },
    }


def extra_estep(torch, zs, dev, n_docs=100, n_topics=100, n_vocab=12419,
                n_leapfrogs=20, n_timed=40):
    """The sizes the reference's OWN loop runs: the E-step of
    examples/topic_models/lntm_mcem.py:62-70,157-182 -- one chain, a minibatch
    of 100 documents, K = 100 topics, L = 20 -- timed through sample_op.run_many
    (one C call, zshmc_hmc_model_run).  Not launch-bound (profiles/
    r05k_estep_kernel_trace.txt): a transition is the sum of its kernels'
    critical paths, so the likelihood's row range is cut into ~100 slices."""
    phi, x = lntm_problem(torch, dev, n_docs, n_topics, n_vocab)
    mean = torch.zeros(n_docs, n_topics, device=dev)
    logstd = torch.zeros(n_topics, device=dev)

    @zs.meta_bayesian_net()
    def lntm():
        bn = zs.BayesianNet()
        eta = bn.normal('eta', mean, logstd=logstd, n_samples=1, group_ndims=1)
        theta = torch.softmax(eta.tensor, -1)
        bn.unnormalized_multinomial(
            'x', torch.log((theta.reshape(-1, n_topics) @ phi).reshape(
                1, n_docs, n_vocab)), normalize_logits=False,
            dtype=torch.float32)
        return bn
    m = lntm()
    m.log_joint = lambda bn: bn.cond_log_prob('eta') + bn.cond_log_prob('x')

    def timed(arithmetic):
        kw = {} if arithmetic is None else {'likelihood_arithmetic': arithmetic}
        hmc = zs.HMC(step_size=0.05, n_leapfrogs=n_leapfrogs, seed=5, **kw)
        eta = torch.zeros(1, n_docs, n_topics, device=dev)
        op, info = hmc.sample(m, {'x': x}, {'eta': eta})
        op.run_many(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        op.run_many(n_timed)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n_timed * 1e3
        plan = hmc._plan
        flop = 4.0 * n_docs * plan.width * n_vocab * n_leapfrogs
        return {
            'likelihood_arithmetic_used': hmc.likelihood_arithmetic_used,
            'arithmetic_reason': hmc.arithmetic_reason,
            'plan': hmc.plan_kind, 'ms_per_step': ms, 'steps': n_timed,
            'value': n_docs * n_leapfrogs / (ms * 1e-3),
            'unit': '(chain, document)-leapfrog-steps/s',
            'mean_acceptance': float(info.acceptance_rate.mean().item()),
            'row_range_slices': int(plan.splits),
            'kernel_width': int(plan.width),
            'sustained_tflops': flop / (ms * 1e-3) / 1e12,
        }
    out = timed(None)                       # the default: 'auto' -> fp32 here
    out['workload'] = (
        'the reference\'s own loop size: lntm_mcem.py E-step, 1 chain x %d '
        'documents, K=%d (kernel width %d), V=%d, L=%d, fixed step size' % (
            n_docs, n_topics, out['kernel_width'], n_vocab, n_leapfrogs))
    out['note'] = ('default arithmetic: exact fp32, row by row over each '
                   'row\'s own words on the vector ALU (csrc/sparse_multinomial'
                   '.hip; the fp32 MFMA kernel: 0.51 ms, 17 us per likelihood '
                   'launch whatever the arithmetic); round 4: 1.2 ms')
    # the packed-rows form of the bf16x3 kernel (one 128-row block, 32-row
    # vocabulary tiles) on the same problem, asked for by name
    try:
        other = timed('bf16x3')
        other['transition_time_over_default'] = \
            other['ms_per_step'] / out['ms_per_step']
        out['bf16x3'] = other
    except Exception as e:                           # noqa: BLE001
        out['bf16x3'] = {'error': repr(e)[:200]}
    return out


def lntm_problem(torch, dev, n_docs, n_topics, n_vocab):
    """documents of ~1 000 tokens drawn from the mixture of a
    random phi (the "nips" file is not reachable offline); seed 0, so every
    rank builds the same data."""
    g = torch.Generator(device=dev).manual_seed(0)
    phi = torch.softmax(torch.randn(n_topics, n_vocab, device=dev,
                                    generator=g), -1)
    doc_mix = torch.softmax(torch.randn(n_docs, n_topics, device=dev,
                                        generator=g), -1)
    words = torch.multinomial(doc_mix @ phi, 1000, replacement=True,
                              generator=g)
    x = torch.zeros(n_docs, n_vocab, device=dev).scatter_add_(
        1, words, torch.ones(words.shape, device=dev))
    return phi, x


def lntm_workload(torch, zs, dev, n_chains, sharding=None, dist=None,
                  n_docs=5000, n_topics=128, n_vocab=12419, n_leapfrogs=20,
                  n_sub=4, n_timed=1, n_warm=1, bf16x3=True, label=None):
    """BASELINE configs[4]: the E-step of the logistic-normal topic model at
    the lntm_mcem.py shape (chain axes [n_chains, n_docs = 5 000], K = 128,
    V = 12 419 -- the UCI "nips" vocabulary the example loads), step-size and
    mass adaptation ON in the timed region, L = 20, the model written with
    the reference's literal log(softmax(eta) @ phi).  `n_chains` is THIS
    rank's share of the leading chain axis (weak scaling: 1 024 per GPU, 8 192
    on 8 GPUs)."""
    world = 1 if sharding is None else sharding.world_size
    rank = 0 if sharding is None else sharding.rank
    phi, x = lntm_problem(torch, dev, n_docs, n_topics, n_vocab)
    eta_mean = torch.zeros(n_docs, n_topics, device=dev)
    eta_logstd = torch.zeros(n_topics, device=dev)

    def build(n, sh, arithmetic=None):
        @zs.meta_bayesian_net()
        def lntm():
            bn = zs.BayesianNet()
            eta = bn.normal('eta', eta_mean, logstd=eta_logstd, n_samples=n,
                            group_ndims=1)
            theta = torch.softmax(eta.tensor, -1)
            pred = (theta.reshape(-1, n_topics) @ phi).reshape(
                n, n_docs, n_vocab)                 # lntm_mcem.py:40-45
            bn.unnormalized_multinomial('x', torch.log(pred),
                                        normalize_logits=False,
                                        dtype=torch.float32)
            return bn
        eta = torch.zeros(n, n_docs, n_topics, device=dev)
        f_ss, f_m = zs.placeholder(bool), zs.placeholder(bool)
        kw = {} if arithmetic is None else {'likelihood_arithmetic': arithmetic}
        hmc = zs.HMC(step_size=1e-3, n_leapfrogs=n_leapfrogs,
                     adapt_step_size=f_ss, adapt_mass=f_m,
                     target_acceptance_rate=0.6, seed=3, sharding=sh, **kw)
        op, info = hmc.sample(lntm(), {'x': x}, {'eta': eta})
        return hmc, op, info, eta, (f_ss, f_m)

    # rank 0 burns the subset in; every rank starts from ITS result (one
    # replicated sampler state, bit for bit)
    box = [None]
    if rank == 0:
        state, eta_sub, ess_pt, acc_sub = _tuned_start(
            torch, zs, build, n_sub, 40, 150, (True, True))
        box = [(state, eta_sub.cpu(), ess_pt, acc_sub)]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    state, eta_sub, ess_pt, acc_sub = box[0]
    eta_sub = eta_sub.to(dev)
    assert n_chains % n_sub == 0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    rows_rank = n_chains * n_docs
    rows = rows_rank * world
    flop_eval = 4.0 * rows_rank * n_topics * n_vocab      # per GPU per launch
    full5 = (n_chains, n_docs, n_topics, n_vocab) == (8192, 5000, 128, 12419)
    rccl_ranks = 0 if sharding is None else sharding.rccl_ranks

    def timed(arithmetic, sh):
        hmc, op, info, eta, flags = build(n_chains, sh, arithmetic)
        eta.copy_(eta_sub.repeat(n_chains // n_sub, 1, 1))
        hmc.set_state(state)
        elapsed, kern_ms, acc = _time_transitions(
            torch, hmc, op, info, {flags[0]: True, flags[1]: True}, n_warm,
            n_timed, barrier)
        if world > 1:
            tt = torch.tensor([elapsed, acc], dtype=torch.float64)
            dist.all_reduce(tt[:1], op=dist.ReduceOp.MAX)
            dist.all_reduce(tt[1:], op=dist.ReduceOp.SUM)
            elapsed, acc = float(tt[0].item()), float(tt[1].item()) / world
        ms = elapsed / n_timed * 1e3
        used = hmc.likelihood_arithmetic_used
        # the vocabulary rows an evaluation really runs over: all of them, or
        # -- one document per workgroup on the bf16x3 kernel -- the mean length
        # of the documents' own (padded) word lists
        n_run = int(getattr(hmc._plan, 'n_inner_run', n_vocab))
        own = getattr(hmc._plan, 'obs_sp', None) is not None and \
            not getattr(hmc._plan, 'sparse_rows', False)
        roof = _lik_roofline(hmc, kern_ms, flop_eval * n_run / n_vocab,
                             n_leapfrogs, ms, 'multinomial')
        roof['note'] = 'per GPU (rank 0): one launch covers this rank\'s rows'
        if getattr(hmc._plan, 'sparse_rows', False):
            # the row-by-row vector-ALU kernel (csrc/sparse_multinomial.hip):
            # not a matrix-core kernel -- priced as a gather out of L2 (every
            # word of every row brings its phi^T row: width * 4 bytes)
            k_ms = kern_ms['grad'] if isinstance(kern_ms, dict) else kern_ms
            gb = rows_rank * float(n_run) * hmc._plan.width * 4.0
            roof = {
                'bound': 'l2-gather', 'dtype': 'f32', 'unit': 'GB/s',
                'kernel': 'sparse_multinomial_kernel<%d> (row by row over '
                          'each row\'s own words, vector ALU) gradient only'
                          % hmc._plan.width,
                'kernel_ms': k_ms, 'achieved': gb / (k_ms * 1e-3) / 1e9,
                'peak': L2_PEAK_GBPS, 'frac': gb / (k_ms * 1e-3) / 1e9 /
                L2_PEAK_GBPS, 'traffic': None,
                'algorithmic_bytes_per_launch': gb,
                'note': 'phi^T rows gathered per (row, word): %d of %d words '
                        'per row run' % (n_run, n_vocab)}
        elif own:
            roof['kernel'] = roof['kernel'].replace(
                'multinomial', 'multinomial, own vocabulary')
            roof['note'] += (
                '; flops counted over the documents\' OWN vocabularies (%d of '
                '%d words per document on average, padded to whole tiles): '
                'words with a zero count contribute exactly nothing and are '
                'not run' % (n_run, n_vocab))
        if full5 and not own:
            roof['traffic'], roof['traffic_source'] = _recorded_mfma_traffic(
                'multinomial bf16x3 grad-only' if used == 'bf16x3' else
                'multinomial grad-only')
        r = {
            'likelihood_arithmetic_used': used,
            'arithmetic_reason': hmc.arithmetic_reason,
            'own_vocabulary': own, 'vocabulary_rows_run': n_run,
            'plan': hmc.plan_kind,
            'ms_per_step': ms, 'steps': n_timed,
            'value': rows * n_leapfrogs / (ms * 1e-3),
            'unit': '(chain, document)-leapfrog-steps/s',
            'mean_acceptance': acc,
            'step_size': float(info.updated_step_size.item()),
            'roofline': roof,
        }
        # (the plan's buffers are freed before the next one is built: seven
        # [rows, K] matrices each; sampler <-> plan <-> sample_op are a cycle)
        del hmc, op, info, eta
        gc.collect()
        torch.cuda.empty_cache()
        return r
    main = timed(None, sharding)            # the default: 'auto'
    other = None
    if world == 1 and bf16x3:
        # the same rows, state, step size and mass on the other arithmetic
        other = timed(_other_arithmetic(main['likelihood_arithmetic_used']),
                      None)
        other['transition_time_over_default'] = \
            other['ms_per_step'] / main['ms_per_step']
        if other['likelihood_arithmetic_used'] == \
                main['likelihood_arithmetic_used']:
            other = None       # (no bf16x3 kernel for this shape: one figure)
    ms = main['ms_per_step']
    out = dict(main)
    out.update({
        'workload': (label or 'configs[4]') +
                    ': logistic-normal topic model E-step, chain '
                    'axes [n_chains=%d, n_docs=%d] (= %d rows; "8 192 chains" '
                    'read as n_chains at 8 GPUs: %d per GPU), K=%d, V=%d, '
                    'L=%d, step-size and mass adaptation on in the timed '
                    'region, literal log(softmax(eta) @ phi) spelling' % (
                        n_chains * world, n_docs, rows, n_chains, n_topics,
                        n_vocab, n_leapfrogs),
        'likelihood_arithmetic': ARITH_NOTE,
        'n_gpus': world,
        'mean_acceptance_subset_held_phase': acc_sub,
        'target_acceptance': 0.6,
        'collective': 'none' if world == 1 else
                      'ONE all-reduce of %d doubles per transition '
                      '[sum acc, flag, colsum[2 x %d]]' % (
                          2 + 2 * n_topics, n_topics),
        'rccl_ranks': rccl_ranks,
        'start': TUNED_START_NOTE + ' Subset: %d chains x %d docs, 40 '
                 'adaptive + 150 recorded transitions (mean acceptance '
                 '%.3f).' % (n_sub, n_docs, acc_sub),
        'ess': {
            'ess_per_row_per_transition': ess_pt,
            'ess_per_sec': ess_pt * rows * 1e3 / ms,
            'method': 'zhusuan.diagnostics estimator (min over the K dims '
                      'per (chain, document) row, mean over rows) on the '
                      'subset run with the same step size and mass, scaled '
                      'to %d rows at the timed rate' % rows,
        },
    })
    if other is not None:
        out[other['likelihood_arithmetic_used']] = other
    return out


def extra_config5(torch, zs, dev, n_chains=None, **kw):
    """configs[4] on ONE GPU at the full named shape (n_chains = 8 192, 4.1e7
    rows, 21 GB per [rows, K] buffer) when that fits the free HBM, else the
    largest power of two that does; the line says which."""
    free_b, _ = torch.cuda.mem_get_info()
    if n_chains is None:
        n_chains = 8192
        # q, q_new, p, grad, grad_start, operand (21 GB each at 8 192) + headroom
        while n_chains > 64 and 9.0 * n_chains * 5000 * 128 * 4 > free_b:
            n_chains //= 2
    return lntm_workload(torch, zs, dev, n_chains, **kw)


def make_sharding(dist, torch, ChainSharding, backend, dev, **layout):
    """The direct RCCL communicator, proven with one all-reduce before it is
    trusted.  If its bootstrap fails or does not come back within two minutes
    on ANY rank, every rank EXITS NON-ZERO (a scaling curve measured over a
    host-staged gloo fallback would be a curve of the wrong thing): rank 0
    says why on stderr.  torch.distributed over the gloo bootstrap group
    (device tensors staged through the host) is used only when asked for --
    ZSHMC_DIST_BACKEND=gloo (two ranks sharing one GPU in the functional
    tests), or ZSHMC_ALLOW_GLOO_FALLBACK=1 to keep a failed RCCL bootstrap
    from costing the line (which then says so: `rccl_ranks` 0)."""
    import threading
    if backend != 'rccl':
        return ChainSharding(backend='torch', **layout), \
            'torch.distributed/%s by request' % backend
    box = {}

    def attempt():
        try:
            torch.cuda.set_device(dev)
            sh = ChainSharding(backend='rccl', **layout)
            probe = torch.ones(2, dtype=torch.float64, device=dev)
            sh.all_reduce_sum(probe)
            torch.cuda.synchronize()
            if probe.tolist() == [float(sh.world_size)] * 2 and \
                    sh.rccl_ranks == sh.world_size:
                box['sh'] = sh
            else:
                box['err'] = 'probe all-reduce returned %r over %d RCCL ' \
                    'ranks' % (probe.tolist(), sh.rccl_ranks)
        except Exception as e:                       # noqa: BLE001
            box['err'] = repr(e)[:300]
    th = threading.Thread(target=attempt, daemon=True)
    th.start()
    th.join(120.0)
    ok = torch.tensor([1.0 if 'sh' in box else 0.0], dtype=torch.float64)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok.item()) == 1.0:
        return box['sh'], 'RCCL (ncclAllReduce through zshmc_comm_*)'
    why = box.get('err', 'this rank was fine; a peer failed' if 'sh' in box
                  else 'bootstrap did not return within 120 s')
    if os.environ.get('ZSHMC_ALLOW_GLOO_FALLBACK') != '1':
        sys.stderr.write(json.dumps({
            'error': 'RCCL communicator unavailable on rank %d of %d: %s' % (
                dist.get_rank(), dist.get_world_size(), why),
            'hint': 'ZSHMC_DIST_BACKEND=gloo runs the collectives over '
                    'torch.distributed/gloo on purpose; '
                    'ZSHMC_ALLOW_GLOO_FALLBACK=1 falls back to it'}) + '\n')
        sys.stderr.flush()
        os._exit(3)
    return ChainSharding(backend='torch', **layout), (
        'FALLBACK torch.distributed/gloo: RCCL communicator unavailable (%s)'
        % why)


def allreduce_latency(torch, sharding, dev, n_iter=1000):
    """The transition's collective alone: `n_iter` all-reduces of the 2-double
    statistics message back to back on the compute stream under one HIP-event
    pair (what each adaptive transition adds to its kernel when the chains
    are sharded)."""
    buf = torch.zeros(2, dtype=torch.float64, device=dev)
    for _ in range(20):
        sharding.all_reduce_sum(buf)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_iter):
        sharding.all_reduce_sum(buf)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n_iter * 1e3


def _over_ranks(dist, torch, world, x):
    """(min, max) of a per-rank host number."""
    if world == 1:
        return x, x
    lo = torch.tensor([x], dtype=torch.float64)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return float(lo.item()), float(hi.item())


def lntm_line_record(r, world, steps, warmup, note):
    """The record of a `--workload lntm` line from lntm_workload's result."""
    out = {
        'metric': 'leapfrog-steps/sec', 'value': r['value'],
        'unit': r['unit'], 'n_gpus': world, 'steps': steps,
        'warmup': warmup, 'ms_per_step': r['ms_per_step'],
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16x3' if r.get('likelihood_arithmetic_used') == 'bf16x3'
        else 'f32', 'data': 'synthetic',
        'config': {'workload': r['workload'],
                   'parallelism': 'leading chain axis sharded over %d '
                                  'GPU(s); %s' % (world, r['collective'])},
        'collective_backend': note,
    }
    for k in ('plan', 'mean_acceptance', 'target_acceptance', 'step_size',
              'rccl_ranks', 'start', 'ess', 'roofline'):
        out[k] = r[k]
    for k in ('bf16x3', 'fp32', 'likelihood_arithmetic_used'):
        if k in r:
            out[k] = r[k]
    return out


def run_lntm_line(args, torch, zs, dist, ChainSharding, dev, world, rank,
                  backend):
    """`--workload lntm`: BASELINE configs[4] as the line's own workload.
    A step = one HMC transition (L = 20, 21 likelihood + gradient
    evaluations) over every rank's [n_chains/N, n_docs] rows."""
    sharding, note = None, 'no collective'
    n = args.lntm_chains_per_gpu
    if world > 1:
        # (the layout is derived by the sampler: flat (chain, document) rows)
        sharding, note = make_sharding(dist, torch, ChainSharding, backend,
                                       dev)
    r = lntm_workload(torch, zs, dev, n, sharding=sharding, dist=dist,
                      n_docs=args.lntm_docs, n_vocab=args.lntm_vocab,
                      n_timed=args.steps, n_warm=args.warmup)
    if rank == 0:
        emit(lntm_line_record(r, world, args.steps, args.warmup, note))
    if world > 1:
        try:
            sharding.close()
        except Exception as e:                       # noqa: BLE001
            sys.stderr.write('communicator teardown: %r\n' % (e,))
        dist.destroy_process_group()


def self_launch(n_gpus):
    """`python bench.py --gpus N ...` without a launcher: re-run the same
    command line under `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N` on loopback (the container's hostname may not
    resolve) and return its exit status.  stdout / stderr pass through, so
    rank 0's one JSON line is this process's output."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(n_gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write('bench.py: no launcher environment, starting %d ranks: '
                     '%s\n' % (n_gpus, ' '.join(cmd)))
    sys.stderr.flush()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------
# Output.  stdout carries, in this order:
#   `#bench-extra {...}`   one line per extra configuration, as it finishes
#   `#bench-detail {...}`  everything measured around the headline
#   `{...}`                THE contract line: last line, the only one that
#                          starts with `{`, <= CONTRACT_MAX_BYTES, strict JSON
# and bench_extras.json (next to bench.py) holds the full record.
CONTRACT_MAX_BYTES = 4096
EXTRA_PREFIX = '#bench-extra '
DETAIL_PREFIX = '#bench-detail '
EXTRAS_FILE = os.path.join(ROOT, 'bench_extras.json')

_ROOFLINE_KEYS = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac',
                  'traffic', 'kernel_ms', 'kernel_launches_timed',
                  'algorithmic_bytes_per_launch',
                  'algorithmic_flop_per_launch', 'dtype', 'rng')
_CPU_KEYS = ('value', 'unit', 'cores', 'kind', 'sample')
# dropped from the contract line, last first, should it ever outgrow its bound
_OPTIONAL = ('extras', 'strong_scaling', 'allreduce_latency_us', 'ess',
             'cpu_reference_over_shim', 'mass_adaptation_overhead',
             'other_adaptation_mode', 'elem_leapfrog_steps_per_sec',
             'mean_acceptance', 'step_size')


def strict(o):
    """A copy of `o` that json.dumps(allow_nan=False) accepts: non-finite
    floats become null, NumPy scalars / arrays become Python ones."""
    if isinstance(o, dict):
        return {str(k): strict(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [strict(v) for v in o]
    if isinstance(o, np.ndarray):
        return strict(o.tolist())
    if isinstance(o, np.generic):
        o = o.item()
    if isinstance(o, float) and not np.isfinite(o):
        return None
    return o


def _dumps(o):
    return json.dumps(strict(o), allow_nan=False, separators=(',', ':'))


def _short(x, digits=6):
    return float('%.*g' % (digits, x)) if isinstance(x, float) else x


def summarize_extra(e):
    """One extra configuration in ~150 bytes for the contract line."""
    s = {'id': e.get('id', str(e.get('workload', '?'))[:24])}
    if 'error' in e:
        s['error'] = str(e['error'])[:80]
        return s
    for k in ('ms_per_step', 'value', 'n_gpus'):
        if k in e:
            s[k] = _short(e[k])
    if isinstance(e.get('run_many'), dict):       # configs[0]: the front-end
        s['transitions_per_sec'] = _short(
            e['run_many'].get('transitions_per_sec'))
    roof = e.get('roofline') or {}
    if roof.get('frac') is not None:
        s['frac'] = _short(roof['frac'], 4)
        s['bound'] = roof.get('bound')
    if 'likelihood_arithmetic_used' in e:
        s['arith'] = e['likelihood_arithmetic_used']
    if e.get('own_vocabulary'):        # topic model: zero-count words not run
        s['vocab_rows_run'] = e.get('vocabulary_rows_run')
    b3 = e.get('bf16x3')
    if isinstance(b3, dict) and 'ms_per_step' in b3:
        s['bf16x3'] = {'ms_per_step': _short(b3['ms_per_step'])}
        f = (b3.get('roofline') or {}).get('frac')
        if f is not None:
            s['bf16x3']['frac'] = _short(f, 4)
    f32 = e.get('fp32')
    if isinstance(f32, dict) and 'ms_per_step' in f32:
        s['fp32'] = {'ms_per_step': _short(f32['ms_per_step'])}
        f = (f32.get('roofline') or {}).get('frac')
        if f is not None:
            s['fp32']['frac'] = _short(f, 4)
    return s


def contract_line(out):
    """The compact object the driver parses: the contract keys, a trimmed
    `roofline` and `cpu_baseline`, a few scalars, one short entry per extra
    configuration.  Always <= CONTRACT_MAX_BYTES and strict JSON."""
    line = {}
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup',
              'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data'):
        line[k] = out.get(k)
    cfg = dict(out.get('config') or {})
    if len(str(cfg.get('workload', ''))) > 200:
        cfg['workload'] = str(cfg['workload'])[:197] + '...'
    line['config'] = cfg
    for k in ('rccl_ranks', 'collective', 'plan'):
        if k in out:
            line[k] = out[k]
    roof = out.get('roofline')
    if roof is not None:
        line['roofline'] = {k: roof[k] for k in _ROOFLINE_KEYS if k in roof}
    cpu = out.get('cpu_baseline')
    if cpu is not None:
        line['cpu_baseline'] = {k: cpu[k] for k in _CPU_KEYS if k in cpu}
        if 'error' in cpu:
            line['cpu_baseline']['error'] = str(cpu['error'])[:120]
    for k in ('mean_acceptance', 'step_size', 'elem_leapfrog_steps_per_sec'):
        if k in out:
            line[k] = out[k]
    if out.get('ess'):
        line['ess'] = {k: v for k, v in out['ess'].items() if k != 'method'}
    om = out.get('other_adaptation_mode')
    if om:
        line['other_adaptation_mode'] = {k: om[k] for k in (
            'adaptation', 'ms_per_step', 'value') if k in om}
    mm = out.get('mass_adaptation_modes')
    if mm and mm.get('overhead_of_adapting') is not None:
        line['mass_adaptation_overhead'] = _short(mm['overhead_of_adapting'], 4)
    ref = out.get('cpu_reference_over_shim')
    if ref:
        line['cpu_reference_over_shim'] = {k: ref[k] for k in (
            'value', 'unit', 'cores', 'transitions_per_sec', 'config')
            if k in ref}
    ar = out.get('allreduce_latency_us')
    if ar:
        line['allreduce_latency_us'] = {k: _short(ar[k], 4) for k in (
            'min_over_ranks', 'max_over_ranks') if k in ar}
    ss = out.get('strong_scaling')
    if ss:
        line['strong_scaling'] = {k: ss[k] for k in (
            'n_chains_total', 'chains_per_gpu', 'ms_per_step', 'value')
            if k in ss}
    if out.get('extra_configs'):
        line['extras'] = [summarize_extra(e) for e in out['extra_configs']]
    line['detail'] = 'bench_extras.json; stdout lines prefixed %s/ %s' % (
        EXTRA_PREFIX.strip(), DETAIL_PREFIX.strip())
    text = _dumps(line)
    for k in _OPTIONAL:
        if len(text) <= CONTRACT_MAX_BYTES:
            break
        line.pop(k, None)
        text = _dumps(line)
    if len(text) > CONTRACT_MAX_BYTES:
        raise RuntimeError('contract line of %d bytes' % len(text))
    return text


_emit_lock = None
_emitted = False
_OUT = None


def claim_stdout():
    """Keep file descriptor 1 for the lines of this file: a private duplicate
    becomes the channel of emit / emit_extra, and descriptor 1 itself is
    pointed at stderr -- so that what ELSE writes to stdout in this process
    (gloo's C++ "[Gloo] Rank 0 is connected to ..." banner, library chatter)
    cannot land between, or after, the contract line."""
    global _OUT
    if _OUT is None:
        sys.stdout.flush()
        _OUT = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)
    return _OUT


def emit_extra(e, stream=None):
    """An extra configuration's full record, on its own prefixed stdout line
    the moment it is known."""
    stream = stream or _OUT or sys.stdout
    stream.write(EXTRA_PREFIX + _dumps(e) + '\n')
    stream.flush()


def emit(out, stream=None, extras_file=''):
    """Write the full record to bench_extras.json and to a prefixed stdout
    line, then THE contract line -- once per process, whoever calls first
    (the normal end of main, the watchdog, or the SIGTERM guard)."""
    global _emitted
    import threading
    global _emit_lock
    if _emit_lock is None:
        _emit_lock = threading.Lock()
    with _emit_lock:
        if _emitted:
            return None
        _emitted = True
        stream = stream or _OUT or sys.stdout
        full = strict(out)
        if extras_file == '':
            extras_file = EXTRAS_FILE
        if extras_file:
            try:
                with open(extras_file, 'w') as f:
                    json.dump(full, f, allow_nan=False, indent=1)
                    f.write('\n')
            except OSError as e:
                sys.stderr.write('bench.py: %s not written: %r\n' % (
                    extras_file, e))
        detail = {k: v for k, v in full.items() if k != 'extra_configs'}
        stream.write(DETAIL_PREFIX + _dumps(detail) + '\n')
        text = contract_line(out)
        stream.write(text + '\n')
        stream.flush()
        return text


def guard_headline(out, rank, timeout_s):
    """N > 1, while the sharded extra runs: should it stall (RCCL, a peer that
    died) or the launcher tear the job down (SIGTERM after another rank's
    failure), rank 0 still prints the headline measured before it and the
    process ends.  Returns a `disarm()` callable.  The SIGTERM side does not
    depend on the main thread reaching a bytecode boundary (it may sit in a
    HIP call): the C-level handler writes to a wake-up pipe that a helper
    thread blocks on."""
    import signal
    import threading

    def give_up(why):
        if rank == 0:
            out.setdefault('extra_configs', []).append(
                {'id': 'configs[4] sharded', 'error': why})
            emit(out)
        os._exit(0)

    timer = threading.Timer(timeout_s + (0 if rank == 0 else 5), give_up,
                            args=('did not finish within %d s' % timeout_s,))
    timer.daemon = True
    timer.start()
    rd, wr = os.pipe()
    os.set_blocking(wr, False)
    state = {'armed': True}

    def waiter():
        while True:
            try:
                b = os.read(rd, 1)     # the number of the signal that arrived
            except OSError:
                return
            if not state['armed'] or not b:
                return
            if b[0] == signal.SIGTERM:
                give_up('SIGTERM from the launcher (a peer rank failed)')
    th = threading.Thread(target=waiter, daemon=True)
    th.start()
    old_fd, old_handler = None, None
    try:
        old_handler = signal.signal(signal.SIGTERM, lambda *a: None)
        old_fd = signal.set_wakeup_fd(wr, warn_on_full_buffer=False)
    except ValueError:                 # not the main thread: timer only
        pass

    def disarm():
        state['armed'] = False
        timer.cancel()
        try:
            if old_handler is not None:
                signal.set_wakeup_fd(old_fd if old_fd is not None else -1)
                signal.signal(signal.SIGTERM, old_handler)
        except ValueError:
            pass
        try:
            os.write(wr, b'x')
        except OSError:
            pass
    return disarm


def _capi_kernel_name(n_data, has_mass, zero_mean):
    from zhusuan_amd import _capi
    return _capi.load().zshmc_fused_kernel_name(n_data, has_mass,
                                                zero_mean).decode()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import zhusuan_amd as zs
    from zhusuan_amd.distributed import ChainSharding

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus == 1 or 'WORLD_SIZE' in os.environ:
        claim_stdout()           # (a rank; the launcher passes its own through)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started as ONE process (`python bench.py --gpus N ...`): become the
        # launcher -- one rank per GPU under torch.distributed.run on this
        # node, rank 0 prints the JSON line, our exit status is the job's
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d inside a launcher environment '
                         'of WORLD_SIZE=%d' % (args.gpus, world))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    # One process per GPU.  torch.distributed (a CPU/gloo group) only
    # bootstraps: it carries the RCCL unique id, the barriers around the timed
    # region and the max-over-ranks of the wall time.  The data path's one
    # collective per transition goes straight to RCCL over xGMI through the
    # C-ABI (zshmc_comm_*).  ZSHMC_DIST_BACKEND=gloo keeps everything on gloo
    # so that two ranks can share one GPU (functional test on a 1-GPU box;
    # RCCL refuses two ranks on one device).
    backend = os.environ.get('ZSHMC_DIST_BACKEND', 'rccl')
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    sharding, collective_note = None, 'no collective'
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if os.environ['MASTER_ADDR'] in ('127.0.0.1', 'localhost'):
            # one node: the bootstrap group talks over loopback whatever the
            # box's hostname resolves to (or does not)
            os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    if args.workload == 'lntm':
        run_lntm_line(args, torch, zs, dist, ChainSharding, dev, world, rank,
                      backend)
        return
    if world > 1:
        per_rank = args.chains_per_gpu // world if args.scaling == 'strong' \
            else args.chains_per_gpu
        sharding, collective_note = make_sharding(
            dist, torch, ChainSharding, backend, dev,
            chain_offset=rank * per_rank, n_chains_global=world * per_rank)

    C, D, L = args.chains_per_gpu, args.n_data, args.leapfrogs
    if args.scaling == 'strong':
        if C % world:
            raise SystemExit('--scaling strong: %d chains do not split over '
                             '%d ranks' % (C, world))
        C //= world
    ar_us = None
    if world > 1:
        lo, hi = _over_ranks(dist, torch, world,
                             allreduce_latency(torch, sharding, dev))
        ar_us = {'min_over_ranks': lo, 'max_over_ranks': hi,
                 'what': '1000 all-reduces of the 2-double statistics '
                         'message back to back on the compute stream, us '
                         'per call'}
    logstd = torch.linspace(-1.0, 1.0, D, device=dev)     # std = e^[-1, 1]
    mean = torch.zeros(D, device=dev)

    @zs.meta_bayesian_net()
    def gaussian():
        bn = zs.BayesianNet()
        bn.normal('x', mean, logstd=logstd, n_samples=C, group_ndims=1)
        return bn

    x = torch.zeros(C, D, device=dev)                     # q0 = 0 (config 2)
    adapt = zs.placeholder(bool)
    hmc = zs.HMC(step_size=0.05, n_leapfrogs=L, adapt_step_size=adapt,
                 target_acceptance_rate=0.8, seed=1, sharding=sharding)
    sample_op, info = hmc.sample(gaussian(), {}, {'x': x})
    assert hmc.plan_kind == 'fused_diag_normal', hmc.plan_kind

    # adaptive burn-in, untimed (config 2: 50 adaptive iterations)
    for _ in range(BURN_IN_ADAPT):
        sample_op.run(feed_dict={adapt: True}, sync=False)
    hmc.check_numerics()
    # timed region: adaptation off on one GPU (config 2); on for N > 1 so
    # that the RCCL all-reduce of the acceptance statistic is in the loop
    # (config 4)
    adapt_timed = world > 1
    feed = {adapt: adapt_timed}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The host is ~4x ahead of the device (0.025 ms per enqueue against a
    # 0.095 ms kernel), but a full collection of CPython's cyclic GC over a
    # process that has torch loaded stops it for ~40 ms -- 400 transitions'
    # worth; tools/archive/first_run_probe.py shows exactly one in the first few
    # hundred runs of a process (the per-run model re-evaluation allocates
    # containers).  Timed regions run with the collector parked, as timeit
    # does.
    # Parked BEFORE the settle launches: the collection itself takes tens of
    # milliseconds, and a GPU left idle that long starts the timed region
    # with its clocks down (tools/archive/startup_probe.py: 106 us per launch over
    # the 40 launches that follow a 20 ms pause, 93 after none).
    gc.collect()
    gc.freeze()
    gc.disable()
    # settle: the chip's clocks move for the first ~70 launches of a process
    # (per-launch durations 77 -> 130 -> 108 us in profiles/archive/r01i_rocprofv3_
    # summary.txt); keep that transient out of the timed region whatever
    # --warmup is
    # (run_many: the launch loop runs inside libzshmc.so -- one call for the
    # whole stretch, zshmc_hmc_diag_normal_run; sharded over RCCL ranks the
    # all-reduce between the launches is enqueued there too)
    sample_op.run_many(SETTLE, feed_dict=feed, sync=False)
    sample_op.run_many(args.warmup, feed_dict=feed, sync=False)
    barrier()
    # ONE HIP-event pair on the launch stream brackets the K launches of the
    # timed region: nothing is recorded between the launches (event records
    # around every launch stretch a 20-step region by 28 %: 0.125 against
    # 0.098 ms per step, gpurun_out/r02w).  At N = 1 the region holds exactly
    # K fused launches and nothing else, so elapsed / K is the kernel's
    # average duration with the inter-launch gaps included -- an upper bound.
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    sample_op.run_many(args.steps, feed_dict=feed, sync=False)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    el_min, elapsed = _over_ranks(dist, torch, world, elapsed)
    hmc.check_numerics()
    acc_mean = float(info.acceptance_rate.mean().item())
    eps = float(info.updated_step_size.item())
    kern_ms_region = ev0.elapsed_time(ev1) / args.steps

    # the same kernel after the timed region, back to back under one event
    # pair.  (Launch by launch under its own event pair it takes 0.118 ms: a
    # kernel bracketed on both sides starts on a drained GPU and cannot
    # overlap its ramp-up with the previous launch's tail.)
    plan = hmc._plan
    plan.collect_acc = False
    stream = torch.cuda.current_stream().cuda_stream
    reps = max(20, min(args.steps, 200))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(
        enable_timing=True)
    t_iter = hmc.t
    # (the host reads above left the GPU idle for milliseconds: 100 launches
    # bring the clocks back before the first event, which is recorded in
    # stream order -- no synchronisation between warm-up and measurement)
    n_pre = 100
    for i in range(n_pre):
        plan._launch(t_iter + 1 + i, None, 1, L, stream)
    e0.record()
    for i in range(reps):
        plan._launch(t_iter + 1 + n_pre + i, None, 1, L, stream)
    e1.record()
    torch.cuda.synchronize()
    kern_ms_alone = e0.elapsed_time(e1) / reps
    # The roofline is quoted from >= 20 timed launches.  N = 1: the timed
    # region itself (or, with --steps < 20, the back-to-back loop, and the
    # line says so).  N > 1: the region also holds the collectives, so the
    # kernel's own duration comes from the back-to-back loop.
    if world == 1 and args.steps >= 20:
        kern_ms, n_timed = kern_ms_region, args.steps
        kernel_timing = (
            'one HIP-event pair on the launch stream around the %d fused '
            'launches of the timed region (nothing else is in the stream): '
            'average per launch, inter-launch gaps included' % args.steps)
    else:
        kern_ms, n_timed = kern_ms_alone, reps
        kernel_timing = (
            'back-to-back loop of %d launches under one HIP-event pair after '
            'the timed region (%s)' % (reps, (
                'the timed region also holds the collectives' if world > 1
                else 'fewer than 20 launches were timed inside the region')))
    hmc.t = t_iter + 1 + n_pre + reps
    algo_bytes = ALGO_BYTES_PER_ELEM * C * D
    from zhusuan_amd import _capi
    # which generator produced the headline, and the same kernel built with
    # the other round count (lib/libzshmc_philox10.so: Random123's and
    # TensorFlow's ten rounds) timed the same way right behind it
    rounds = int(_capi.load().zshmc_philox_rounds())
    rng_name = 'philox4x32-%d' % rounds
    other_generator = None
    alt_path = os.path.join(os.path.dirname(_capi.LIB_PATH),
                            'libzshmc_philox10.so')
    if world == 1 and rounds != 10 and os.path.exists(alt_path):
        alt = _capi.load_build(alt_path)
        for i in range(n_pre):
            plan._launch(hmc.t + i, None, 1, L, stream, lib=alt)
        e0.record()
        for i in range(reps):
            plan._launch(hmc.t + n_pre + i, None, 1, L, stream, lib=alt)
        e1.record()
        # ... and the default build once more, so that the pair shares the
        # chip's state (clocks, temperature) as closely as one stream allows
        e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(
            enable_timing=True)
        for i in range(n_pre):
            plan._launch(hmc.t + i, None, 1, L, stream)
        e2.record()
        for i in range(reps):
            plan._launch(hmc.t + n_pre + i, None, 1, L, stream)
        e3.record()
        torch.cuda.synchronize()
        alt_ms, same_ms = e0.elapsed_time(e1) / reps, e2.elapsed_time(e3) / reps
        hmc.t += 2 * (n_pre + reps)
        other_generator = {
            'rng': 'philox4x32-%d' % int(alt.zshmc_philox_rounds()),
            'library': 'zhusuan_amd/lib/libzshmc_philox10.so '
                       '(-DZS_PHILOX_ROUNDS=10, same sources)',
            'kernel_ms_back_to_back': alt_ms,
            'frac': algo_bytes / (alt_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            'default_build_right_after': {
                'rng': rng_name, 'kernel_ms_back_to_back': same_ms,
                'frac': algo_bytes / (same_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
        }
    kernel_name = _capi.load().zshmc_fused_kernel_name(D, 0, 1).decode()
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC counters: NOT collected in this run
    # (counters need their own rocprofv3 passes, tools/profile.sh); the value
    # is the one committed with the profile summaries and says so
    traffic = traffic_source = None
    pmc_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(pmc_path):
        try:
            with open(pmc_path) as f:
                pmc = json.load(f)
            traffic = pmc.get('hbm_bytes_per_launch')
            traffic_source = 'profiles/pmc_traffic.json (%s; separate rocprofv3 ' \
                '--pmc passes of this command, not this run)' % pmc.get(
                    'round', 'r01')
        except Exception:
            traffic = None

    ms_per_step = elapsed / args.steps * 1e3
    total_chains = C * world
    value = total_chains * L * args.steps / elapsed

    # the other adaptation mode, for a like-for-like reading across N: the
    # headline region runs adaptation off at N = 1 (config 2) and on at N > 1
    # (config 4: all-reduce + update kernel in the loop)
    saved = hmc.get_state()
    other_feed = {adapt: not adapt_timed}
    sample_op.run_many(5, feed_dict=other_feed, sync=False)
    barrier()
    t1 = time.perf_counter()
    # (side measurements: 200 transitions whatever --steps is -- over 20 of
    # them the drained pipeline at either end of a region is 5-10 % of it)
    n_other = 200
    sample_op.run_many(n_other, feed_dict=other_feed, sync=False)
    barrier()
    other_elapsed = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([other_elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        other_elapsed = float(tt.item())
    hmc.set_state(saved)
    other_mode = {
        'adaptation': 'off' if adapt_timed else 'on',
        'ms_per_step': other_elapsed / n_other * 1e3,
        'value': total_chains * L * n_other / other_elapsed,
        'steps': n_other,
    }

    # the same transitions driven one `sample_op.run` at a time from Python
    # (what a loop of sess.run is): the front-end's per-transition cost
    barrier()
    t1 = time.perf_counter()
    for _ in range(n_other):
        sample_op.run(feed_dict=feed, sync=False)
    barrier()
    loop_elapsed = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([loop_elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        loop_elapsed = float(tt.item())
    hmc.set_state(saved)
    python_loop = {
        'ms_per_step': loop_elapsed / n_other * 1e3,
        'value': total_chains * L * n_other / loop_elapsed,
        'steps': n_other,
        'what': 'one sample_op.run per transition from Python instead of '
                'one run_many call (zshmc_hmc_diag_normal_run)',
    }

    # strong scaling next to the weak headline ("fixed-total
    # strong scaling"): the SAME --chains-per-gpu chains in total, split over
    # the ranks, adaptation on so that the all-reduce is in the loop
    strong = None
    if world > 1 and args.scaling == 'weak' and \
            args.chains_per_gpu % world == 0:
        Cs = args.chains_per_gpu // world

        @zs.meta_bayesian_net()
        def gaussian_s():
            bn = zs.BayesianNet()
            bn.normal('x', mean, logstd=logstd, n_samples=Cs, group_ndims=1)
            return bn
        xs = torch.zeros(Cs, D, device=dev)
        f_s = zs.placeholder(bool)
        hmc_s = zs.HMC(step_size=0.05, n_leapfrogs=L, adapt_step_size=f_s,
                       target_acceptance_rate=0.8, seed=1,
                       sharding=sharding.relayout(
                           chain_offset=rank * Cs,
                           n_chains_global=args.chains_per_gpu))
        op_s, info_s = hmc_s.sample(gaussian_s(), {}, {'x': xs})
        for _ in range(BURN_IN_ADAPT):
            op_s.run(feed_dict={f_s: True}, sync=False)
        hmc_s.check_numerics()
        op_s.run_many(100 + args.warmup, feed_dict={f_s: True}, sync=False)
        barrier()
        t1 = time.perf_counter()
        op_s.run_many(args.steps, feed_dict={f_s: True}, sync=False)
        barrier()
        s_min, s_el = _over_ranks(dist, torch, world,
                                  time.perf_counter() - t1)
        hmc_s.check_numerics()
        strong = {
            'n_chains_total': args.chains_per_gpu, 'chains_per_gpu': Cs,
            'adaptation': 'on', 'steps': args.steps,
            'ms_per_step': s_el / args.steps * 1e3,
            'ms_per_step_fastest_rank': s_min / args.steps * 1e3,
            'value': args.chains_per_gpu * L * args.steps / s_el,
            'mean_acceptance': float(info_s.acceptance_rate.mean().item()),
            'kernel': _capi_kernel_name(D, 0, 1),
        }
        del hmc_s, op_s, info_s, xs

    # mass adaptation (config 1's first 50 iterations, config 5): a second
    # sampler on the same state with adapt_mass declared, so that the mass
    # tile is part of the kernel.  Step size + mass adapting on every
    # transition -- the column sums of the end state come out of the
    # transition launch, one more launch reduces them + EWMV + mass + tau, and
    # when sharded they travel in the same all-reduce as the acceptance sum --
    # against the same sampler with both flags off.
    f_ss, f_m = zs.placeholder(bool), zs.placeholder(bool)
    hmc_m = zs.HMC(step_size=0.05, n_leapfrogs=L, adapt_step_size=f_ss,
                   adapt_mass=f_m, target_acceptance_rate=0.8, seed=1,
                   sharding=sharding)
    op_m, info_m = hmc_m.sample(gaussian(), {}, {'x': x})
    for _ in range(30):
        op_m.run(feed_dict={f_ss: True, f_m: True}, sync=False)
    hmc_m.check_numerics()
    mass_modes = {}
    for label, on in (('step size + mass adapting', True),
                      ('both flags off', False)):
        feed_m = {f_ss: on, f_m: on}
        for _ in range(60):
            op_m.run(feed_dict=feed_m, sync=False)
        barrier()
        t1 = time.perf_counter()
        for _ in range(n_other):
            op_m.run(feed_dict=feed_m, sync=False)
        barrier()
        el = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        mass_modes[label] = {
            'ms_per_step': el / n_other * 1e3,
            'value': total_chains * L * n_other / el,
            'steps': n_other,
            'mean_acceptance': float(info_m.acceptance_rate.mean().item()),
        }
    mass_modes['kernel'] = _capi_kernel_name(D, 1, 1)
    mass_modes['overhead_of_adapting'] = (
        mass_modes['step size + mass adapting']['ms_per_step'] /
        mass_modes['both flags off']['ms_per_step'] - 1.0)
    del hmc_m, op_m, info_m
    gc.enable()

    # ESS/s (reference estimator, zhusuan/diagnostics.py:17-64) over EVERY
    # chain of this rank: record n_draws snapshots of the state on the device,
    # batched ESS kernel (csrc/diagnostics.hip), minimum over dimensions per
    # chain as diagnostics.py:55-64, mean over chains.  All D dimensions when
    # the [n_draws, C, D] record fits in half of the free HBM, otherwise an
    # evenly strided subset of dimensions (stated in `method`).
    ess = None
    if not args.no_ess:
        n_draws, burn = 400, 100
        free_b, _ = torch.cuda.mem_get_info()
        need = n_draws * C * D * 4
        stride = 1
        while need // stride > 0.5 * free_b:
            stride *= 2
        dims_t = None if stride == 1 else torch.arange(0, D, stride, device=dev)
        d_sel = D if dims_t is None else int(dims_t.numel())
        rec = torch.empty(n_draws, C, d_sel, device=dev)
        for i in range(n_draws):
            sample_op.run(feed_dict={adapt: False}, sync=False)
            rec[i].copy_(x if dims_t is None else x[:, dims_t])
        ess_chain = zs.diagnostics.effective_sample_size_device(rec, burn_in=burn)
        ok = torch.isfinite(ess_chain)
        ess_per_iter = float(ess_chain[ok].mean().item()) / (n_draws - burn)
        del rec
        if rank == 0:
            ess = {
                'ess_per_sec': ess_per_iter * total_chains * 1e3 / ms_per_step,
                'ess_per_chain_per_transition': ess_per_iter,
                'method': 'zhusuan.diagnostics estimator on the device for all '
                          '%d chains of rank 0 x %d of %d dims (min over dims '
                          'per chain, mean over chains), %d draws, burn_in=%d, '
                          'scaled to %d chains' % (C, d_sel, D, n_draws, burn,
                                                   total_chains),
            }
    if world > 1:
        barrier()

    if rank == 0:
        out = {
            'metric': 'leapfrog-steps/sec',
            'value': value,
            'unit': 'chain-leapfrog-steps/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'higher_is_better': True,
            'scaling': args.scaling,
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {
                'rng': rng_name,
                'workload': 'configs[1]: %d chains/GPU x %d-D diagonal '
                            'Gaussian (std=exp(linspace(-1,1))), L=%d, q0=0, '
                            '50 adaptive burn-in transitions, timed with '
                            'adaptation %s' % (C, D, L,
                                               'on' if adapt_timed else 'off'),
                'n_chains_total': total_chains,
                'n_latents': D,
                'n_leapfrogs': L,
                'parallelism': 'chains sharded over %d GPU(s); %s' % (
                    world, 'one ncclAllReduce (2 doubles) per transition on '
                    'the compute stream' if world > 1 else 'no collective'),
            },
            'rccl_ranks': 0 if sharding is None else sharding.rccl_ranks,
            'collective': collective_note,
            'allreduce_latency_us': ar_us,
            'ms_per_step_ranks': {'min': el_min / args.steps * 1e3,
                                  'max': elapsed / args.steps * 1e3},
            'strong_scaling': strong,
            'launches_per_transition': 1,
            'driver': 'sample_op.run_many(K): one call into libzshmc.so launches '
                      'the K transitions of the timed region',
            'elem_leapfrog_steps_per_sec': value * D,
            'mean_acceptance': acc_mean,
            'other_adaptation_mode': other_mode,
            'python_loop': python_loop,
            'mass_adaptation_modes': mass_modes,
            'step_size': eps,
            'roofline': {
                'bound': 'hbm',
                'kernel': kernel_name,
                'achieved': achieved,
                'peak': HBM_PEAK_GBPS,
                'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBPS,
                'traffic': traffic,
                'traffic_source': traffic_source,
                'kernel_ms': kern_ms,
                'kernel_ms_timed_region': kern_ms_region,
                'kernel_ms_back_to_back': kern_ms_alone,
                'kernel_launches_timed': n_timed,
                'kernel_timing': kernel_timing,
                'algorithmic_bytes_per_launch': algo_bytes,
                'rng': rng_name,
                'other_generator': other_generator,
            },
        }
        if ess is not None:
            out['ess'] = ess
        if world == 1 and not args.no_cpu_baseline:
            # Measured live on this box's host cores, strongest first: the
            # chain-fused C + OpenMP port on all threads is THE cpu_baseline
            # (a GPU/CPU ratio against a 1-core NumPy loop says nothing);
            # the op-for-op ports (torch-CPU all threads, NumPy one core)
            # model the TF-CPU executor's pass structure; the reference's own
            # hmc.py over the TensorFlow-API shim cannot run here (its sources
            # do not travel) and is carried as a recorded number.
            budget = min(args.cpu_seconds, 8.0)
            # (torch first: the OpenMP pool of the C port keeps spinning for a
            # while after its last parallel region and would fight torch's)
            try:
                out['cpu_baseline_torch_all_threads'] = cpu_baseline_torch(
                    D, L, budget)
            except Exception as e:
                out['cpu_baseline_torch_all_threads'] = {'error': str(e)[:200]}
            out['cpu_baseline_numpy_1core'] = cpu_baseline_numpy(D, L, budget)
            try:
                out['cpu_baseline'] = cpu_baseline_parallel(D, L, budget)
            except Exception as e:           # no gcc / OpenMP on the box
                out['cpu_baseline'] = {'error': str(e)[:200]}
            if 'error' in out['cpu_baseline']:
                out['cpu_baseline'] = out['cpu_baseline_numpy_1core']
            out['cpu_reference_over_shim'] = cpu_reference_recorded()
    else:
        out = {}
    # the MFMA-bound configurations of BASELINE.json, after the headline
    # (each frees its buffers before the next starts).  N = 1: configs[2] and
    # configs[4] at its full one-GPU shape.  N > 1: configs[4] with its
    # leading chain axis sharded over the ranks (1 024 chains per GPU = 8 192
    # at N = 8), step-size and mass adaptation on, ONE all-reduce of
    # 2 + 2 K doubles per transition on the same communicator.
    extras = None
    if not args.no_extra_configs:
        # N > 1: the sharded extra has collectives in it; should a rank fail,
        # RCCL stall there or the launcher tear the job down, the headline
        # measured above must still be printed (guard_headline)
        disarm = None
        if world > 1:
            if rank == 0:
                # (the same figures, early, where a log reader finds them even
                # if the process is killed outright)
                sys.stderr.write('#bench-headline ' + contract_line(out) + '\n')
                sys.stderr.flush()
            disarm = guard_headline(out, rank, EXTRA_TIMEOUT_S)
        del x
        torch.cuda.empty_cache()
        from zhusuan_amd import _ops
        extras = []
        if world == 1:
            todo = (('configs[0]', extra_config1, {}),
                    ('configs[2]', extra_config3, {}),
                    ('configs[4]', extra_config5,
                     {'n_chains': args.config5_chains}),
                    ('blr-1000+bias', extra_wide_regression, {}),
                    ('blr-299+bias', extra_wide_regression,
                     {'n_feat': 299, 'n_chains': 16384}),
                    ('softmax-10x784', extra_softmax_regression, {}),
                    ('softmax-10x256', extra_softmax_regression,
                     {'n_feat': 256, 'bf16x3': True}),
                    ('pmf', extra_pmf, {}),
                    ('lntm-estep', extra_estep, {}),
                    # lntm_mcem.py's own layout at scale: ONE chain, many
                    # documents (every row its own counts row)
                    ('lntm-1chain-32768docs', lntm_workload, dict(
                        n_chains=1, n_docs=32768, n_sub=1, n_timed=3,
                        label='beyond BASELINE.json, lntm_mcem.py:62-70\'s '
                              'layout (n_chains = 1 x many documents)')))
        else:
            todo = (('configs[4] sharded', lntm_workload, dict(
                n_chains=args.lntm_chains_per_gpu,
                sharding=sharding.relayout(), dist=dist,
                n_docs=args.lntm_docs, n_vocab=args.lntm_vocab)),)
        only = os.environ.get('ZSHMC_BENCH_EXTRAS')
        if only:
            todo = tuple(t for t in todo if t[0] in only.split(','))
        for ident, fn, kw in todo:
            try:
                e = fn(torch, zs, dev, **kw)
            except Exception as err:         # report, never lose the headline
                if world > 1:
                    # (the peers are in a collective: let the guard print)
                    sys.stderr.write('rank %d: %r\n' % (rank, err))
                    time.sleep(EXTRA_TIMEOUT_S + 10)
                e = {'workload': fn.__name__, 'error': repr(err)[:300]}
            e = dict(e, id=ident)
            extras.append(e)
            if rank == 0:
                emit_extra(e)
            _ops.clear_caches()
            torch.cuda.empty_cache()
        if world > 1:
            barrier()
            disarm()

    if rank == 0:
        if extras is not None:
            out['extra_configs'] = extras
        emit(out)
    if world > 1:
        # every rank has passed the last barrier with an idle stream: the
        # communicator can go (a failure here must not cost the line above)
        try:
            sharding.close()
        except Exception as e:                       # noqa: BLE001
            sys.stderr.write('communicator teardown: %r\n' % (e,))
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
