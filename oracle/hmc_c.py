"""ctypes wrapper of the C + OpenMP restatement of one diag-Normal HMC
transition (oracle/c/hmc_diag_normal_port.c).  TEST / BASELINE INFRASTRUCTURE
(see oracle/__init__.py): used by bench.py's all-cores CPU baseline and by
tests/test_oracle_c_port.py."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'libzs_oracle.so')
_lib = None


def build():
    """gcc -O3 -fopenmp the C port into oracle/_build/ (git-ignored)."""
    subprocess.check_call(['make', '-s', '-C', os.path.join(HERE, 'c')])
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = ctypes.CDLL(LIB)
        f = lib.zs_oracle_hmc_diag_normal_step
        f.restype = ctypes.c_int
        p = ctypes.c_void_p
        f.argtypes = [p, p, p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                      ctypes.c_int, ctypes.c_float, ctypes.c_uint64,
                      ctypes.c_uint32, p, p, p, p, p, ctypes.c_int]
        lib.zs_oracle_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def max_threads():
    return int(load().zs_oracle_threads())


def step(q, mean, logstd, n_leapfrogs, step_size, seed, iteration,
         chain_offset=0, n_threads=0, want_info=True):
    """One transition in place on q [C, D] (float32, C-contiguous).  Returns
    dict of the five per-chain HMCInfo vectors (or None) and the
    non-finite-old-log-prob flag."""
    assert q.dtype == np.float32 and q.flags['C_CONTIGUOUS'] and q.ndim == 2
    C, D = q.shape
    mean = np.ascontiguousarray(mean, np.float32)
    logstd = np.ascontiguousarray(logstd, np.float32)
    info = {k: np.empty(C, np.float32) for k in (
        'acceptance_rate', 'orig_hamiltonian', 'hamiltonian', 'orig_log_prob',
        'log_prob')} if want_info else None
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    args = [ptr(info[k]) if info else None for k in (
        'acceptance_rate', 'orig_hamiltonian', 'hamiltonian', 'orig_log_prob',
        'log_prob')]
    bad = load().zs_oracle_hmc_diag_normal_step(
        ptr(q), ptr(mean), ptr(logstd), C, D, int(chain_offset),
        int(n_leapfrogs), float(step_size), int(seed) & 0xFFFFFFFFFFFFFFFF,
        int(iteration) & 0xFFFFFFFF, *args, int(n_threads))
    return info, bool(bad)


class DiagNormalFreeRun(object):
    """A free-running sampler on the C port: zhusuan/hmc.py:382-522 for a
    diagonal-Normal joint without mass adaptation -- the transition
    (:458, :348-372, :479-498) is `step` above on every host thread, the
    step-size search (:308-345: one full leapfrog step from the same (q, p0),
    x / : 1.5 until the side of the target flips) is `step` with L = 1 on a
    scratch copy of the state, dual averaging (:89-112, :375-380) is
    oracle/hmc_ref.py's StepsizeTuner on the float32 mean acceptance of ALL
    chains.  Used by tests/test_gpu_fused_fullsize.py to follow the device at
    BASELINE configs[1]'s full 65 536 x 1 024."""

    def __init__(self, q, mean, logstd, step_size, n_leapfrogs,
                 target_acceptance_rate=0.8, gamma=0.05, t0=100, kappa=0.75,
                 seed=0):
        from .hmc_ref import StepsizeTuner
        self.q, self.mean, self.logstd = q, mean, logstd
        self.step_size = np.float32(step_size)
        self.n_leapfrogs = int(n_leapfrogs)
        self.delta = np.float32(target_acceptance_rate)
        self.tuner = StepsizeTuner(step_size, gamma, t0, kappa,
                                   target_acceptance_rate)
        self.seed = seed
        self.t = 0
        self.n_init_trips = 0

    def _search(self):
        f32 = np.float32
        step, last, cond = self.step_size, f32(1.0), True
        self.n_init_trips = 0
        while cond:
            trial = self.q.copy()
            info, bad = step_fn(trial, self.mean, self.logstd, 1, step,
                                self.seed, self.t)
            if bad:
                raise FloatingPointError('old_log_prob has numeric errors')
            acc = f32(np.mean(info['acceptance_rate'], dtype=np.float32))
            new = step * (f32(1.0) / f32(1.5)) if acc < self.delta \
                else step * f32(1.5)
            cond = not ((last < self.delta) ^ (acc < self.delta))
            step, last = f32(new), acc
            self.n_init_trips += 1
        return step

    def run(self, adapt_step_size):
        """One transition; returns (info dict, mean acceptance)."""
        self.t += 1
        init = self.t == 1          # no mass: mass_collect_iters = 0 (:276)
        used = self._search() if init else self.step_size
        info, bad = step_fn(self.q, self.mean, self.logstd, self.n_leapfrogs,
                            used, self.seed, self.t)
        if bad:
            raise FloatingPointError('old_log_prob has numeric errors')
        acc = np.float32(np.mean(info['acceptance_rate'], dtype=np.float32))
        self.step_size = np.float32(self.tuner.tune(
            acc, np.float32(init), bool(adapt_step_size)))
        return info, acc


step_fn = step
