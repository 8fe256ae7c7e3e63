"""Shared helpers for the GPU parity tests (HIP path vs NumPy oracle)."""
import numpy as np

from oracle.hmc_ref import HMC as RefHMC, DiagNormalModel


def make_diag_problem(C, D, seed=0, q_scale=1.0):
    rng = np.random.RandomState(seed)
    mean = rng.normal(size=D).astype(np.float32)
    logstd = rng.uniform(-1.0, 1.0, size=D).astype(np.float32)
    q0 = (mean + q_scale * np.exp(logstd) *
          rng.normal(size=(C, D))).astype(np.float32)
    return mean, logstd, q0


def ref_sampler(mean, logstd, q0, **hmc_kwargs):
    model = DiagNormalModel(mean, logstd=logstd)
    x = q0.copy()
    ref = RefHMC(**hmc_kwargs)
    ref.sample(model.log_joint, model.grad, [x])
    return ref, x


def gpu_sampler(zs, torch, mean, logstd, q0, generic=False, group_ndims=None,
                **hmc_kwargs):
    dev = torch.device('cuda', 0)
    C = q0.shape[0]
    mean_t = torch.tensor(mean, device=dev)
    logstd_t = torch.tensor(logstd, device=dev)

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        bn.normal('x', mean_t, logstd=logstd_t, n_samples=C,
                  group_ndims=q0.ndim - 1 if group_ndims is None
                  else group_ndims)
        return bn

    x = torch.tensor(q0, device=dev)
    hmc = zs.HMC(**hmc_kwargs)
    m = model()
    if generic:
        # a plain callable hides the structure -> generic (autograd) plan
        op, info = hmc.sample(lambda obs: m.observe(**obs).log_joint(), {},
                              {'x': x})
    else:
        op, info = hmc.sample(m, {}, {'x': x})
    return hmc, op, info, x


class FusedKernel(object):
    """The fused diag-Normal transition straight through the C-ABI
    (zshmc_hmc_diag_normal_step + zshmc_adapt_link), with the buffers a
    sampler would own."""

    def __init__(self, torch, C, D, dev=None, n_chains_global=None):
        from zhusuan_amd import _capi
        self.torch, self.capi = torch, _capi
        self.C, self.D = C, D
        dev = dev or torch.device('cuda', 0)
        self.dev = dev
        self.info = [torch.zeros(C, device=dev) for _ in range(5)]
        self.stats = torch.zeros(_capi.STATS_WORDS, dtype=torch.float64,
                                 device=dev)
        self.workspace = torch.zeros(_capi.LINK_WORKSPACE_BYTES,
                                     dtype=torch.uint8, device=dev)
        self.flags = torch.zeros(1, dtype=torch.int32, device=dev)
        self.state = torch.zeros(_capi.STATE_WORDS, device=dev)
        self.n_chains_global = n_chains_global or C

    def link(self, use_state=False, pending=0, fresh=0, used=float('nan'),
             delta=0.8, gamma=0.05, t0=100.0, kappa=0.75, mu=0.5,
             collect=True, colstats_mean=None, colstats_parts=None):
        c = self.capi
        k = c.AdaptLink()
        if colstats_parts is not None:
            k.colstats_mean = colstats_mean.data_ptr()
            k.colstats_parts = colstats_parts.data_ptr()
        k.state = self.state.data_ptr() if use_state else None
        k.stats = self.stats.data_ptr() if collect else None
        k.workspace = self.workspace.data_ptr()
        k.n_chains_global = self.n_chains_global
        k.pending, k.fresh_start, k.used_step_size = pending, fresh, used
        k.delta, k.gamma, k.t0, k.kappa, k.mu = delta, gamma, t0, kappa, mu
        return k

    def step(self, q, mean, logstd, mass, eps, L, seed, iteration,
             chain_offset=0, commit=1, link=None, want_info=True):
        import ctypes
        c, torch = self.capi, self.torch
        link = link if link is not None else self.link()
        ptrs = [x.data_ptr() if want_info else None for x in self.info]
        c.call('zshmc_hmc_diag_normal_step', q.data_ptr(),
               None if mean is None else mean.data_ptr(), logstd.data_ptr(),
               None if mass is None else mass.data_ptr(), float(eps),
               q.shape[0], self.D, chain_offset, L, seed, iteration, commit,
               *ptrs, self.flags.data_ptr(), ctypes.byref(link),
               torch.cuda.current_stream().cuda_stream)


def compare_transition(info, x_gpu, rinfo, x_ref, ref, lp_tol=None):
    """Per-chain comparison of one transition.  Chains whose accept decision
    is numerically borderline (|u - acc| tiny) may legitimately flip between
    float32 implementations; they are excluded from the state comparison and
    their count is bounded."""
    acc_g = info.acceptance_rate.cpu().numpy().reshape(-1)
    acc_r = np.asarray(rinfo.acceptance_rate).reshape(-1)
    scale = max(1.0, float(np.abs(rinfo.orig_hamiltonian).max()))
    # energies carry float32 rounding of O(|H| * 1e-6) per implementation
    h_tol = 2e-5 * scale + 1e-4
    np.testing.assert_allclose(
        info.orig_log_prob.cpu().numpy().reshape(-1),
        np.asarray(rinfo.orig_log_prob).reshape(-1), rtol=0, atol=h_tol)
    np.testing.assert_allclose(
        info.orig_hamiltonian.cpu().numpy().reshape(-1),
        np.asarray(rinfo.orig_hamiltonian).reshape(-1), rtol=0, atol=h_tol)
    np.testing.assert_allclose(
        info.hamiltonian.cpu().numpy().reshape(-1),
        np.asarray(rinfo.hamiltonian).reshape(-1), rtol=0, atol=2 * h_tol)
    # acc = exp(min(dH, 0)): |d acc| <= |d dH|
    np.testing.assert_allclose(acc_g, acc_r, rtol=0, atol=6 * h_tol)
    u = np.asarray(ref.last_u01).reshape(-1)
    borderline = np.abs(u - acc_r) < 12 * h_tol
    xg = x_gpu.cpu().numpy().reshape(acc_r.shape[0], -1)
    xr = np.asarray(x_ref).reshape(acc_r.shape[0], -1)
    q_scale = max(1.0, float(np.abs(xr).max()))
    row_bad = ~np.isclose(xg, xr, rtol=0, atol=2e-5 * q_scale).all(axis=1)
    lp_g = info.log_prob.cpu().numpy().reshape(-1)
    lp_bad = ~np.isclose(lp_g, np.asarray(rinfo.log_prob).reshape(-1),
                         rtol=0, atol=2 * h_tol)
    flipped = row_bad | lp_bad
    # every disagreement must be a borderline accept decision ...
    assert not np.any(flipped & ~borderline), np.nonzero(flipped & ~borderline)
    # ... and those must be rare
    assert flipped.sum() <= max(2, 0.01 * flipped.size), flipped.sum()
    return int(flipped.sum())
