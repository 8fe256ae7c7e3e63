"""A NumPy stand-in for the C entry points the fused diagonal-Normal plan
calls, so that the HOST ORCHESTRATION of zhusuan_amd/hmc.py (`HMC._run`: flag
handling, the step-size search loop, pending / retired dual-averaging
updates, flush, chain sharding and its all-reduces) can be exercised on a box
without a GPU -- where the driver runs `-m "not gpu"`.  Test infrastructure:
the transition itself is the oracle's (oracle/hmc_ref.py pieces); the
step-size link between transitions -- which update is applied when, from
which sum (zshmc_adapt_link: pending / retire_update / fresh_start /
used_step_size) -- is NOT restated: it is csrc/fused_args.h itself
(link_step_size, link_retire, tuner_persist: the code the kernels run)
compiled for the host, tests/host_link/zs_link_host.cpp.  What is restated
here is the mass estimator of csrc/adapt.hip (zshmc_mass_colstats /
zshmc_mass_update[_fused]) and zshmc_state_set.  Tensors are torch CPU
tensors addressed through data_ptr()."""
import ctypes

import numpy as np

from oracle import hmc_ref, philox
from zhusuan_amd import _capi

F32 = np.float32


def _view(ptr, n, ctype, dtype):
    if not ptr:
        return None
    return np.frombuffer((ctype * n).from_address(int(ptr)), dtype=dtype)


def _f32(ptr, n):
    return _view(ptr, n, ctypes.c_float, np.float32)


def _host_link():
    """csrc/fused_args.h (link_step_size / link_retire / tuner_persist -- the
    code the kernels run) compiled for the host: tests/host_link/."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(here, 'host_link', 'zs_link_host.cpp')
    hdr = os.path.join(os.path.dirname(here), 'zhusuan_amd', 'csrc',
                       'fused_args.h')
    out = os.path.join(here, '_build', 'libzs_link_host.so')
    if not os.path.exists(out) or os.path.getmtime(out) < max(
            os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        cxx = '/opt/rocm/lib/llvm/bin/clang++'      # (ext_vector_type)
        tmp = out + '.%d.tmp' % os.getpid()
        subprocess.check_call([cxx, '-O2', '-ffp-contract=off',
                               '-DZS_HOST_ONLY', '-shared', '-fPIC', src,
                               '-o', tmp])
        os.replace(tmp, out)
    lib = ctypes.CDLL(out)
    lib.zs_host_link_step_size.restype = ctypes.c_float
    lib.zs_host_link_step_size.argtypes = [ctypes.POINTER(_capi.AdaptLink),
                                           ctypes.c_float]
    lib.zs_host_link_retire.restype = None
    lib.zs_host_link_retire.argtypes = [ctypes.POINTER(_capi.AdaptLink),
                                        ctypes.c_double, ctypes.c_uint]
    lib.zs_host_link_flush.restype = None
    lib.zs_host_link_flush.argtypes = [ctypes.POINTER(_capi.AdaptLink)]
    return lib


_LINK = _host_link()


class FakeLibrary(object):
    """`call(name, *args)` with the argument order of _capi.PROTOTYPES."""

    def __init__(self):
        self.calls = []

    def call(self, name, *args):
        self.calls.append(name)
        return getattr(self, name)(*args)

    # -- zshmc_state_set(state, index, value, stream) -------------------------
    def zshmc_state_set(self, state, index, value, stream):
        _f32(state, _capi.STATE_WORDS)[index] = F32(value)

    # -- zshmc_stepsize_flush(link, stream) ------------------------------------
    def zshmc_stepsize_flush(self, link_ref, stream):
        _LINK.zs_host_link_flush(link_ref)

    # -- zshmc_hmc_diag_normal_step -------------------------------------------
    def zshmc_hmc_diag_normal_step(
            self, q, mean, logstd, mass, step_size_host, n_chains, n_data,
            chain_offset, n_leapfrogs, seed, iteration, commit, acc_out, h0,
            h1, lp0, lp1, flags, link_ref, stream):
        link = link_ref._obj
        C, D = int(n_chains), int(n_data)
        qv = _f32(q, C * D).reshape(C, D)
        mean_v = _f32(mean, D) if mean else np.zeros(D, F32)
        model = hmc_ref.DiagNormalModel(mean_v.copy(),
                                        logstd=_f32(logstd, D).copy())
        m = [(_f32(mass, D) if mass else np.ones(D, F32)).reshape(1, D)]
        # prologue: the step size of THIS transition -- the product's own
        # link_step_size (pending update applied to a copy of the state)
        eps = F32(_LINK.zs_host_link_step_size(link_ref,
                                               float(step_size_host)))
        # the transition (hmc.py:458-498), oracle pieces
        p0 = hmc_ref.random_momentum(seed, int(iteration), [(C, D)], m, 1,
                                     int(chain_offset))
        cq, cp = [qv.copy()], list(p0)
        for i in range(int(n_leapfrogs) + 1):
            s1 = eps if i > 0 else F32(0)
            s2 = eps if 0 < i < n_leapfrogs else eps / F32(2)
            cq, cp = hmc_ref.leapfrog_integrator(cq, cp, s1, s2, model.grad, m)
        bad = False
        try:
            oh, nh, olp, nlp, acc = hmc_ref.get_acceptance_rate(
                [qv], p0, cq, cp, model.log_joint, m, [[1]])
        except hmc_ref.NumericError:
            bad = True
            acc = np.zeros(C, F32)
            oh = nh = olp = nlp = np.full(C, np.nan, F32)
        u = philox.uniform_per_chain(seed, int(iteration), C,
                                     int(chain_offset))
        accept = u < acc
        if commit:
            qv[...] = np.where(accept[:, None], cq[0], qv)
            for ptr, val in ((acc_out, acc), (h0, oh), (h1, nh), (lp0, olp),
                             (lp1, np.where(accept, nlp, olp))):
                if ptr:
                    _f32(ptr, C)[...] = val
        if bad and flags:
            _view(flags, 1, ctypes.c_uint32, np.uint32)[0] |= 1
        # epilogue: the product's own link_retire (pending update from the
        # OLD sum, this transition's own update, publication of the new sum)
        total = float(np.sum(acc.astype(np.float64)))
        _LINK.zs_host_link_retire(link_ref, total, 1 if bad else 0)

    # -- zshmc_mass_colstats / zshmc_mass_update (csrc/adapt.hip) -------------
    def zshmc_mass_colstats(self, q, ewmv_mean, n_chains, n_data, colsum,
                            stream):
        C, D = int(n_chains), int(n_data)
        qv = _f32(q, C * D).reshape(C, D)
        d = (qv - _f32(ewmv_mean, D)).astype(np.float64)
        cs = _view(colsum, 2 * D, ctypes.c_double, np.float64)
        cs[:D] += d.sum(0)
        cs[D:] += (d * d).sum(0)

    def zshmc_mass_update(self, state, ewmv_mean, ewmv_var, colsum,
                          n_chains_global, n_data, decay, update, use_ones,
                          mass_out, stream):
        D = int(n_data)
        st = _f32(state, _capi.STATE_WORDS)
        mean, var = _f32(ewmv_mean, D), _f32(ewmv_var, D)
        tau_new = F32(st[_capi.ST_EWMV_T]) + F32(1)
        if update:
            cs = _view(colsum, 2 * D, ctypes.c_double, np.float64)
            w = (F32(1) - F32(decay)) / (F32(1) - np.power(F32(decay), tau_new,
                                                           dtype=F32))
            s1 = cs[:D] / float(n_chains_global)
            s2 = cs[D:] / float(n_chains_global)
            delta = float(w) * s1
            mean[...] = (mean.astype(np.float64) + delta).astype(F32)
            var[...] = ((1.0 - float(w)) * var.astype(np.float64) +
                        float(w) * s2 - delta * delta).astype(F32)
            cs[...] = 0.0
        with np.errstate(divide='ignore'):
            _f32(mass_out, D)[...] = F32(1) if use_ones else F32(1) / var
        if update == 1:
            st[_capi.ST_EWMV_T] = tau_new

    # -- zshmc_mass_update_fused: rows of column sums -> EWMV update, mass,
    # tau, in one call (csrc/adapt.hip) ---------------------------------------
    def zshmc_mass_update_fused(self, state, ewmv_mean, ewmv_var, parts,
                                n_parts, n_chains_global, n_data, decay,
                                use_ones, mass_out, workspace, stream):
        D = int(n_data)
        rows = _view(parts, int(n_parts) * 2 * D, ctypes.c_double,
                     np.float64).reshape(int(n_parts), 2 * D)
        total = np.zeros(2 * D, np.float64)
        for r in rows:                       # row order: deterministic
            total += r
        keep = total.copy()
        buf = (ctypes.c_double * (2 * D))(*total)
        self.zshmc_mass_update(state, ewmv_mean, ewmv_var,
                               ctypes.addressof(buf), n_chains_global, n_data,
                               decay, 1, use_ones, mass_out, stream)
        assert np.array_equal(rows.sum(0) if n_parts > 1 else rows[0], keep)

    def zshmc_mass_colstats_reduce(self, parts, n_parts, n_data, colsum,
                                   stream):
        D = int(n_data)
        rows = _view(parts, int(n_parts) * 2 * D, ctypes.c_double,
                     np.float64).reshape(int(n_parts), 2 * D)
        _view(colsum, 2 * D, ctypes.c_double, np.float64)[...] = rows.sum(0)

    def zshmc_zero(self, ptr, n_bytes, stream):
        ctypes.memset(int(ptr), 0, int(n_bytes))

    # -- zshmc_hmc_diag_normal_run: the launch loop of csrc/hmc_fused_normal.hip
    def zshmc_hmc_diag_normal_run(
            self, q, mean, logstd, mass, step_size_host, n_chains, n_data,
            chain_offset, n_leapfrogs, seed, iteration_first, n_transitions,
            acc_out, h0, h1, lp0, lp1, flags, link_ref, comm, stream):
        link = link_ref._obj
        kind = link.retire_update
        assert not comm, 'the fake has no communicator'
        for i in range(int(n_transitions)):
            l = _capi.AdaptLink.from_buffer_copy(link)
            if i > 0:
                l.fresh_start, l.used_step_size = 0, float('nan')
                l.pending = _capi.PEND_NONE
            l.retire_update = kind
            self.zshmc_hmc_diag_normal_step(
                q, mean, logstd, mass, step_size_host, n_chains, n_data,
                chain_offset, n_leapfrogs, seed, int(iteration_first) + i, 1,
                acc_out, h0, h1, lp0, lp1, flags, ctypes.byref(l), stream)
