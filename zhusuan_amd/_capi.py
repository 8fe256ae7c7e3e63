"""ctypes binding of libzshmc.so (include/zshmc.h) -- the only way the Python
host code reaches the HIP kernels.  There is NO CPU fallback: if the shared
library is missing, importing any compute entry point raises.

Build the library with ``python -c "import __graft_entry__ as g; g.build()"``
(hipcc --offload-arch=gfx950, in-tree: zhusuan_amd/lib/libzshmc.so).
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32,
                    c_int64, c_uint32, c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libzshmc.so')

ZSHMC_OK = 0
FLAG_OLD_LOGPROB_NONFINITE = 1

STATE_WORDS = 8
ST_STEP_SIZE = 0
ST_TUNER_STEP = 1
ST_LOG_EPS_BAR = 2
ST_H_BAR = 3
ST_EWMV_T = 4
ST_USED_STEP_SIZE = 5
ST_MEAN_ACCEPT = 6

PEND_NONE = 0
PEND_ADAPT = 1
PEND_HOLD = 2
STATS_WORDS = 2
LINK_WORKSPACE_BYTES = 64
COMM_ID_BYTES = 128
ERR_COMM = 4


class AdaptLink(Structure):
    """zshmc_adapt_link (include/zshmc.h): how consecutive fused transitions
    hand over the acceptance statistic and the pending step-size update."""
    _fields_ = [('state', c_void_p), ('stats', c_void_p),
                ('workspace', c_void_p), ('n_chains_global', c_int64),
                ('pending', c_int32), ('retire_update', c_int32),
                ('fresh_start', c_int32),
                ('used_step_size', c_float), ('delta', c_float),
                ('gamma', c_float), ('t0', c_float), ('kappa', c_float),
                ('mu', c_float), ('colstats_mean', c_void_p),
                ('colstats_parts', c_void_p)]


MAX_LATENTS = 8
PLAN_KINDS = {'linear_bernoulli': 0, 'mixture_multinomial': 1,
              'linear_categorical': 2, 'gathered_dot': 3}


class ModelPlan(Structure):
    """zshmc_model_plan (include/zshmc.h): the buffers of a native model
    plan, for zshmc_hmc_model_run."""
    _fields_ = [
        ('kind', c_int32), ('n_latents', c_int32), ('n_leapfrogs', c_int32),
        ('softmax', c_int32), ('segmented', c_int32), ('use_mass', c_int32),
        ('n_splits', c_int32), ('n_classes', c_int32),
        ('latent', c_void_p * MAX_LATENTS),
        ('latent_size', c_int64 * MAX_LATENTS),
        ('latent_offset', c_int64 * MAX_LATENTS),
        ('latent_mass', c_void_p * MAX_LATENTS),
        ('q_new', c_void_p), ('p', c_void_p),
        ('n_chains', c_int64), ('n_total', c_int64), ('ld', c_int64),
        ('operand', c_void_p), ('grad', c_void_p), ('ll', c_void_p),
        ('lik_rows', c_int64), ('width', c_int64),
        ('grad_start', c_void_p), ('ll_start', c_void_p),
        ('start_valid', c_int32), ('one_launch', c_int32),
        ('inner', c_void_p), ('n_inner', c_int64), ('inner_image', c_void_p),
        ('obs', c_void_p), ('obs_rows', c_int64), ('obs_stride', c_int64),
        ('split_ws', c_void_p),
        ('seg_len', c_int64), ('groups', c_int64), ('seg_ws', c_void_p),
        ('gd_latent_is_u', c_int32), ('gd_pad', c_int32),
        ('gd_idx_latent', c_void_p), ('gd_idx_other', c_void_p),
        ('gd_seg', c_void_p), ('gd_order', c_void_p),
        ('gd_n_latent', c_int64), ('gd_n_pairs', c_int64),
        ('gd_n_dim', c_int64),
        ('gd_logstd', c_float), ('gd_pad2', c_float),
        ('gd_lp_const', c_void_p), ('gd_g_pairs', c_void_p),
        ('prior_mean', c_void_p), ('mean_rows', c_int64),
        ('prior_logstd', c_void_p), ('logstd_rows', c_int64),
        ('mass', c_void_p),
        ('lp_old', c_void_p), ('lp_new', c_void_p), ('kin_old', c_void_p),
        ('kin_new', c_void_p), ('accept', c_void_p),
        ('acceptance_rate', c_void_p), ('orig_hamiltonian', c_void_p),
        ('hamiltonian', c_void_p), ('log_prob', c_void_p),
        ('acc_sum', c_void_p), ('flags', c_void_p), ('state', c_void_p),
        ('chain_offset', c_int64), ('n_chains_global', c_int64),
        ('seed', c_uint64),
        ('delta', c_float), ('gamma', c_float), ('t0', c_float),
        ('kappa', c_float), ('mu', c_float), ('mass_decay', c_float),
        ('ewmv_mean', c_void_p * MAX_LATENTS),
        ('ewmv_var', c_void_p * MAX_LATENTS),
        ('colsum', c_void_p * MAX_LATENTS),
        ('comm_buf', c_void_p), ('comm_words', c_int64),
        ('mass_ws', c_void_p), ('traj_sync', c_void_p),
        ('gd_seg_ptr', c_void_p), ('gd_seg_row', c_void_p),
        ('gd_seg_first', c_void_p), ('gd_long_rows', c_void_p),
        ('gd_n_seg', c_int64), ('gd_n_long', c_int64),
        ('gd_idx_other_csr', c_void_p), ('gd_obs_csr', c_void_p),
        ('obs_sp_counts', c_void_p), ('obs_sp_rows', c_void_p),
        ('obs_sp_off', c_void_p)]


BCAST_FULL = 0
BCAST_ROW = 1
BCAST_SCALAR = 2

_p = c_void_p  # every device pointer / stream travels as void*

# name -> (restype, argtypes); mirrors include/zshmc.h one to one
# (tests/test_capi_symbols.py checks header <-> table <-> .so agreement)
PROTOTYPES = {
    'zshmc_last_error': (c_char_p, []),
    'zshmc_version': (c_int, []),
    'zshmc_philox_rounds': (c_int, []),
    'zshmc_zero': (c_int, [_p, c_int64, _p]),
    'zshmc_fused_max_n_data': (c_int64, []),
    'zshmc_fused_kernel_name': (c_char_p, [c_int64, c_int, c_int]),
    'zshmc_hmc_model_transition': (c_int, [_p, c_uint32, c_float, _p]),
    'zshmc_trajectory_capacity': (c_int, [c_int64, c_int, _p]),
    'zshmc_linear_multinomial_log_lik': (c_int, [
        _p, _p, _p, c_int64, c_int64, c_int64, c_int64, c_int64, _p, _p, c_int,
        _p, _p]),
    'zshmc_ess_series': (c_int, [_p, c_int64, c_int64, _p, _p]),
    'zshmc_min_positive_rows': (c_int, [_p, c_int64, c_int64, _p, _p]),
    'zshmc_uni2_log_prob': (c_int, [
        c_int, _p, _p, _p, _p, c_int64, c_int64, c_int, c_int, c_int, _p]),
    'zshmc_uni2_log_prob_grad': (c_int, [
        c_int, _p, _p, _p, _p, _p, _p, _p, c_int64, c_int64, c_int, c_int,
        c_int, _p]),
    'zshmc_uni2_sample': (c_int, [
        c_int, _p, _p, _p, c_int64, c_int64, c_int, c_int, c_uint64, c_uint32,
        _p]),
    'zshmc_sgld_update': (c_int, [
        _p, _p, _p, c_float, c_float, c_float, c_int64, c_uint64, c_uint32,
        c_uint32, _p]),
    'zshmc_sg_momentum': (c_int, [
        _p, c_float, c_int64, c_uint64, c_uint32, c_uint32, _p]),
    'zshmc_sg_half_drift': (c_int, [_p, _p, c_int64, _p, _p]),
    'zshmc_sghmc_update': (c_int, [
        _p, _p, _p, c_int64, c_float, c_float, c_float, c_int, c_uint64,
        c_uint32, c_uint32, _p, _p]),
    'zshmc_sgnht_update': (c_int, [
        _p, _p, _p, _p, _p, _p, c_int64, c_float, c_float, c_float, c_int,
        c_uint64, c_uint32, c_uint32, _p, _p]),
    'zshmc_sgnht_scalar': (c_int, [
        _p, _p, c_int64, c_float, c_float, c_int, c_int, _p, _p]),
    'zshmc_hmc_diag_normal_step': (c_int, [
        _p, _p, _p, _p, c_float, c_int64, c_int64, c_int64, c_int,
        c_uint64, c_uint32, c_int, _p, _p, _p, _p, _p, _p,
        POINTER(AdaptLink), _p]),
    'zshmc_hmc_diag_normal_run': (c_int, [
        _p, _p, _p, _p, c_float, c_int64, c_int64, c_int64, c_int,
        c_uint64, c_uint32, c_int, _p, _p, _p, _p, _p, _p,
        POINTER(AdaptLink), _p, _p]),
    'zshmc_stepsize_flush': (c_int, [POINTER(AdaptLink), _p]),
    'zshmc_comm_unique_id': (c_int, [_p]),
    'zshmc_comm_create': (c_int, [_p, c_int, c_int, POINTER(c_void_p)]),
    'zshmc_comm_all_reduce_sum': (c_int, [_p, _p, c_int64, _p]),
    'zshmc_comm_world_size': (c_int, [_p]),
    'zshmc_comm_destroy': (c_int, [_p]),
    'zshmc_stepsize_update': (c_int, [
        _p, _p, c_int64, c_int, c_int, c_float, c_float, c_float, c_float,
        c_float, _p]),
    'zshmc_state_set': (c_int, [_p, c_int, c_float, _p]),
    'zshmc_mass_colstats': (c_int, [_p, _p, c_int64, c_int64, _p, _p]),
    'zshmc_mass_colstats_reduce': (c_int, [_p, c_int64, c_int64, _p, _p]),
    'zshmc_mass_update_fused': (c_int, [
        _p, _p, _p, _p, c_int64, c_int64, c_int64, c_float, c_int, _p, _p,
        _p]),
    'zshmc_fused_colstats_rows': (c_int64, [c_int64, c_int64, c_int, c_int]),
    'zshmc_mass_update': (c_int, [
        _p, _p, _p, _p, c_int64, c_int64, c_float, c_int, c_int, _p, _p]),
    'zshmc_momentum': (c_int, [
        _p, _p, c_int64, c_int64, c_int64, c_uint64, c_uint32, c_uint32, _p,
        _p]),
    'zshmc_momentum_rows': (c_int, [
        _p, c_int64, _p, c_int64, c_int64, c_int64, c_uint64, c_uint32,
        c_uint32, _p, _p]),
    'zshmc_kick_drift': (c_int, [
        _p, _p, _p, _p, _p, c_float, c_float, c_float, c_int64, c_int64, _p,
        _p]),
    'zshmc_mh_accept': (c_int, [
        _p, _p, _p, _p, c_int64, c_int64, c_uint64, c_uint32, _p, _p, _p, _p,
        _p, _p, _p, _p]),
    'zshmc_select_rows': (c_int, [_p, _p, _p, c_int64, c_int64, _p]),
    'zshmc_copy_rows': (c_int, [
        _p, c_int64, _p, c_int64, _p, c_int64, c_int64, _p]),
    'zshmc_model_kick_drift': (c_int, [
        _p, _p, _p, c_int64, _p, c_int64, c_int, _p, c_int64, _p, c_int64, _p,
        _p, c_float, c_float, c_float, c_float, c_int64, c_int64, c_int64, _p,
        _p, _p, _p]),
    'zshmc_hmc_model_run': (c_int, [
        POINTER(ModelPlan), c_uint32, c_int, c_int, c_int, _p, _p, c_int, _p,
        _p]),
    'zshmc_model_seg_workspace': (c_int64, [c_int64, c_int64]),
    'zshmc_model_kick_drift_seg': (c_int, [
        _p, _p, _p, c_int64, c_int64, c_int64, _p, c_int64, _p, c_int64, _p,
        c_int64, _p, _p, c_float, c_float, c_float, c_float, c_int64, c_int64,
        c_int64, _p, _p, _p, _p, _p]),
    'zshmc_normal_log_prob': (c_int, [
        _p, _p, _p, _p, c_int64, c_int64, c_int, c_int, c_int, _p]),
    'zshmc_normal_log_prob_grad': (c_int, [
        _p, _p, _p, _p, _p, _p, _p, c_int64, c_int64, c_int, c_int, c_int,
        _p]),
    'zshmc_bernoulli_log_prob': (c_int, [
        _p, _p, _p, c_int64, c_int64, c_int, c_int, c_int, _p]),
    'zshmc_bernoulli_log_prob_grad': (c_int, [
        _p, _p, _p, _p, c_int64, c_int64, c_int, c_int, c_int, _p]),
    'zshmc_categorical_log_prob': (c_int, [_p, _p, _p, c_int64, c_int64, _p]),
    'zshmc_categorical_log_prob_grad': (c_int, [
        _p, _p, _p, _p, c_int64, c_int64, _p]),
    'zshmc_unnormalized_multinomial_log_prob': (c_int, [
        _p, _p, _p, c_int64, c_int64, c_int, _p]),
    'zshmc_unnormalized_multinomial_log_prob_grad': (c_int, [
        _p, _p, _p, _p, c_int64, c_int64, c_int, _p]),
    'zshmc_likelihood_plan': (c_int, [c_int64, c_int, _p, _p]),
    'zshmc_linear_bernoulli_log_lik': (c_int, [
        _p, _p, _p, c_int64, c_int64, c_int64, _p, _p, c_int, _p, _p]),
    'zshmc_bf16x3_image_bytes': (c_int, [c_int64, c_int64, _p]),
    'zshmc_bf16x3_split': (c_int, [_p, c_int64, c_int64, c_int64, _p, _p]),
    'zshmc_linear_bernoulli_log_lik_bf16x3': (c_int, [
        _p, _p, _p, c_int64, c_int64, c_int64, _p, _p, c_int, _p, _p]),
    'zshmc_bf16x3_multinomial_rows_packed': (c_int, [c_int64, c_int64]),
    'zshmc_sparse_multinomial_log_lik': (c_int, [
        _p, _p, _p, _p, _p, c_int64, c_int64, c_int64, c_int64, _p, _p, c_int,
        _p, _p]),
    'zshmc_linear_multinomial_log_lik_bf16x3_sparse': (c_int, [
        _p, _p, _p, _p, _p, c_int64, c_int64, c_int64, c_int64, _p, _p, c_int,
        _p, _p]),
    'zshmc_linear_multinomial_log_lik_bf16x3': (c_int, [
        _p, _p, _p, c_int64, c_int64, c_int64, c_int64, c_int64, _p, _p,
        c_int, _p, _p]),
    'zshmc_linear_categorical_log_lik_bf16x3': (c_int, [
        _p, _p, _p, c_int64, c_int64, c_int64, c_int, c_int, _p, _p, c_int, _p,
        _p]),
    'zshmc_linear_categorical_log_lik': (c_int, [
        _p, _p, _p, c_int64, c_int64, c_int64, c_int, c_int, _p, _p, c_int, _p,
        _p]),
    'zshmc_gather_dot': (c_int, [
        _p, _p, _p, _p, c_int64, c_int64, c_int64, c_int64, c_int64, _p, _p]),
    'zshmc_gather_dot_normal_workspace': (c_int64, [c_int64, c_int64]),
    'zshmc_gather_dot_normal_lik': (c_int, [
        _p, _p, _p, _p, _p, c_int64, c_float, _p, c_int64, c_int64, c_int64,
        c_int64, c_int64, _p, _p, _p, _p]),
    'zshmc_gather_dot_normal_lik_grad': (c_int, [
        _p, _p, _p, _p, _p, _p, c_int64, _p, _p, c_int64, c_float, _p,
        c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, _p, _p, _p, _p]),
    'zshmc_gather_dot_grad': (c_int, [
        _p, _p, _p, _p, _p, c_int64, c_int64, c_int64, c_int64, c_int64, _p,
        _p]),
    'zshmc_mvn_tril_log_prob': (c_int, [
        _p, _p, _p, c_int64, c_int64, c_int64, c_int64, _p, _p, _p, _p]),
    'zshmc_mvn_tril_sample': (c_int, [
        _p, _p, _p, c_int64, c_int64, c_int64, c_int64, c_uint64, c_uint32,
        _p]),
    'zshmc_normal_sample': (c_int, [
        _p, _p, _p, c_int64, c_int64, c_int, c_int, c_uint64, c_uint32, _p]),
    'zshmc_bernoulli_sample': (c_int, [
        _p, _p, c_int64, c_int64, c_uint64, c_uint32, _p]),
    'zshmc_categorical_sample': (c_int, [
        _p, _p, c_int64, c_int64, c_int64, c_uint64, c_uint32, _p]),
}


class ZshmcError(RuntimeError):
    """A libzshmc.so call returned a non-zero status."""


class LibraryMissing(ImportError):
    pass


_lib = None


def load():
    """Load libzshmc.so (once).  Raises LibraryMissing -- loudly -- when the
    HIP extension has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            "zhusuan_amd: %s not found. The HIP extension is required (no CPU "
            "fallback). Build it with `python -c \"import __graft_entry__ as "
            "g; g.build()\"`." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error():
    msg = load().zshmc_last_error()
    return msg.decode('utf-8', 'replace') if msg else ''


def call(name, *args):
    """Invoke a status-returning entry point; raise ZshmcError on failure."""
    rc = getattr(load(), name)(*args)
    if rc != ZSHMC_OK:
        raise ZshmcError('%s failed (status %d): %s' % (name, rc, last_error()))


def call_on(lib, name, *args):
    """`call` on a library handle of `load_build`."""
    rc = getattr(lib, name)(*args)
    if rc != ZSHMC_OK:
        msg = lib.zshmc_last_error()
        raise ZshmcError('%s failed (status %d): %s' % (
            name, rc, msg.decode('utf-8', 'replace') if msg else ''))


def load_build(path):
    """Another build of the same library (the Philox4x32-10 build,
    lib/libzshmc_philox10.so) with the same prototypes -- for measurements
    that put two builds side by side in one process (bench.py's
    `other_generator`).  The product itself always runs `load()`."""
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
