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
