"""Which native plan takes a model: the recognisers run the model function once
on symbolic latents (zhusuan_amd/_symbolic.py) and match the node structure of
the reference's examples -- logistic regression (w @ X.T [+ ...] under a
Bernoulli), softmax regression, lntm_mcem.py:33-48, pmf_hmc.py:19-31 -- or say
why not (hmc.plan_reason, NativePlanFallbackWarning)."""

import torch

from .. import _capi, _symbolic
from ..distributions import Normal
from ..framework.bn import StochasticTensor
from ..framework.meta_bn import MetaBayesianNet
from ..utils import merge_dicts
from .base import _Unsupported
from .dense import _DenseLikelihoodPlan


def _softmax_of(theta, probe):
    """True if `theta` is torch.softmax(probe, -1) (recognised on the autograd
    graph: the model is written with the ordinary torch op)."""
    fn = getattr(theta, 'grad_fn', None)
    if fn is None or type(fn).__name__ != 'SoftmaxBackward0':
        return False
    dim = getattr(fn, '_saved_dim', None)
    if dim is None:
        return False
    if dim >= 1 << 63:              # a negative axis, saved as uint64
        dim -= 1 << 64
    if dim % probe.dim() != probe.dim() - 1:
        return False
    nxt = fn.next_functions[0][0]
    return getattr(nxt, 'variable', None) is probe


def _summands_of(lp, nodes):
    """The nodes whose `cond_log_p` tensors are exactly the two operands of
    `lp = a + b` (identity of autograd nodes), else None."""
    fn = getattr(lp, 'grad_fn', None)
    if fn is None or type(fn).__name__ != 'AddBackward0' or \
            getattr(fn, '_saved_alpha', 1) != 1:
        return None
    parents = [f for f, _ in fn.next_functions]
    if len(parents) != 2 or parents[0] is None or parents[1] is None:
        return None
    picked = []
    for node in nodes:
        clp = node.__dict__.get('_cond_log_p')      # evaluated by lp only
        if clp is not None and any(clp.grad_fn is f for f in parents):
            picked.append(node)
    if len(picked) != 2 or picked[0]._cond_log_p.grad_fn is \
            picked[1]._cond_log_p.grad_fn:
        return None
    return picked


def _ops_max_classes():
    from .. import _ops
    return _ops.MAX_CLASSES


def _try_dense_likelihood_plan(hmc, meta_bn, names, values, chain_shape,
                               device):
    from ..distributions import (Bernoulli, Categorical,
                                UnnormalizedMultinomial)
    # every `return no(...)` below is a drop to the autograd-driven generic
    # plan; the reason is kept (hmc.plan_reason) and, once a dense likelihood
    # has been seen in the model, said aloud (NativePlanFallbackWarning)
    state = {'dense': False}

    def no(reason):
        hmc._note_refusal(reason, loud=state['dense'])
        return None

    if not isinstance(meta_bn, MetaBayesianNet):
        return no('the log-joint is a plain callable: no model structure to '
                  'lower')
    n_chain = len(chain_shape)
    # every latent: one data axis, or none (a per-chain scalar: a bias); a
    # single latent may have two (the [K, F] class rows of a softmax
    # regression)
    for n, q in zip(names, values):
        if q.dim() not in (n_chain, n_chain + 1) and not (
                len(values) == 1 and q.dim() == n_chain + 2):
            return no("latent '%s' has %d data axes" % (n, q.dim() - n_chain))
        if q.data_ptr() % 16 != 0 or not q.is_contiguous() or \
                q.dtype != torch.float32:
            return no("latent '%s' is not a 16-byte aligned contiguous "
                      "float32 tensor" % n)
    two_axes = values[0].dim() == n_chain + 2
    sizes = [int(q.shape[-1]) if q.dim() >= n_chain + 1 else 1
             for q in values]
    if two_axes:
        K, F = (int(v) for v in values[0].shape[-2:])
        if not (1 <= K <= _ops_max_classes() and 1 <= F <= 1024):
            return no('a [%d, %d] latent (the dense-logit Categorical kernel '
                      'takes up to %d classes x 1 024 features)'
                      % (K, F, _ops_max_classes()))
    elif min(sizes) < 1 or sum(sizes) > 1024:
        return no('%d latent columns (the dense-likelihood kernels take up '
                  'to 1 024 features / topics)' % sum(sizes))
    if len(names) > 1 and meta_bn.log_joint is not None:
        return no('a user log-joint over several latents')

    def analyse(vals):
        """(kind, [(prior mean, prior spread)], [inner tensors], observation)
        for the latents given as `vals`, or None."""
        bn = meta_bn.observe(**merge_dicts(
            {n: hmc._as_symbol(v) for n, v in zip(names, vals)},
            hmc._resolved_observed()))
        stoch = [n for n in bn.nodes.values()
                 if isinstance(n, StochasticTensor)]
        if meta_bn.log_joint is not None:
            # a user log-joint is accepted when it is, structurally, the sum
            # of two nodes' conditional log-densities -- the E-step objective
            # of lntm_mcem.py:97-102, cond_log_prob('eta') + cond_log_prob('x')
            # -- checked on the autograd graph (a tempered or re-weighted
            # joint, e.g. AIS's, has multiplications on top and is refused)
            stoch = _summands_of(bn.log_joint(), stoch) \
                if vals[0].requires_grad else [
                    n for n in stoch if n.name in analyse.accepted]
            if stoch is None:
                return no('the user log-joint is not the plain sum of two '
                          "nodes' cond_log_prob")
            analyse.accepted = [n.name for n in stoch]
        lik = [n for n in stoch if n.name not in names]
        state['dense'] = any(
            getattr(n.dist, '_lazy', None) is not None for n in lik)
        if len(stoch) != len(names) + 1:
            return no('%d stochastic nodes in the joint for %d latent(s): '
                      'one likelihood node expected'
                      % (len(stoch), len(names)))
        if len(lik) != 1 or not lik[0].is_observed():
            return no('no single observed likelihood node')
        priors = []
        for name, v in zip(names, vals):
            node = [n for n in stoch if n.name == name]
            if len(node) != 1:
                return no("latent '%s' is not a node of the joint" % name)
            pd = node[0].dist
            if type(pd) is not Normal or pd.use_path_derivative or \
                    pd.group_ndims != v.dim() - n_chain:
                return no("the prior of '%s' is not a Normal over its data "
                          "axes (group_ndims = %d)" % (name,
                                                       v.dim() - n_chain))
            # (a prior whose parameters depend on another latent -- a
            # hierarchical scale -- requires grad here: the generic plan)
            if pd.mean.requires_grad or pd.given_spread[1].requires_grad:
                return no("the prior of '%s' has parameters that depend on "
                          "a latent (hierarchical prior)" % name)
            priors.append((pd.mean, pd.given_spread))
        ld = lik[0].dist
        obs = lik[0].tensor
        lazy = getattr(ld, '_lazy', None)
        if lazy is None:
            return no("the logits of '%s' are not a dense contraction of the "
                      "latents that the symbolic layer recognises "
                      "(zhusuan_amd/_symbolic.py): they are materialised"
                      % lik[0].name)
        if type(ld) is Bernoulli:
            if two_axes:
                return no('a latent with two data axes under a Bernoulli')
            if ld.group_ndims != 1 or lazy.design_requires_grad() or \
                    obs.dim() != 1 or obs.shape[0] != lazy.n_rows or \
                    obs.requires_grad or len(lazy.terms) != len(vals):
                return no('Bernoulli likelihood outside the native shape: '
                          'group_ndims = 1, labels [N], constant design '
                          'matrices, one term per latent')
            # one term per latent, in the order of the latents
            inner = []
            for v in vals:
                term = [t for t in lazy.terms if t[0] is v]
                if len(term) != 1 or term[0][2] != (v.dim() == n_chain):
                    return no('a latent enters the logits more than once '
                              '(or not at all)')
                inner.append(term[0][1])
            return 'linear_bernoulli', priors, inner, obs
        if type(ld) is Categorical:
            value = vals[0]
            if not two_axes or lazy.w is not value:
                return no('Categorical logits that are not X @ w^T of the '
                          'one latent w[..., K, F]')
            if ld.group_ndims != 1 or not lazy.fused_ok() or \
                    obs.requires_grad or obs.dim() < 1 or \
                    obs.numel() != lazy.n_rows or \
                    obs.shape[-1] != lazy.n_rows:
                return no('Categorical likelihood outside the native shape: '
                          'group_ndims = 1, labels [N], at most %d classes x '
                          '%d features' % (_ops_max_classes(), 1024))
            return 'linear_categorical', priors, [lazy.X], obs
        if type(ld) is UnnormalizedMultinomial:
            value = vals[0]
            if len(vals) != 1 or value.dim() != n_chain + 1 or \
                    ld.group_ndims != 0 or ld.normalize_logits or \
                    lazy.phi.requires_grad or obs.requires_grad:
                return no('UnnormalizedMultinomial outside the native shape: '
                          'one latent, group_ndims = 0, '
                          'normalize_logits = False, constant phi')
            if lazy.softmax_source is not None:
                # the literal spelling, lowered symbolically: theta IS
                # softmax(latent) by construction
                if lazy.softmax_source is not value:
                    return no('theta is not softmax(latent)')
            elif value.requires_grad and not _softmax_of(lazy.theta, value):
                return no('theta is not softmax(latent)')
            batch = tuple(lazy.shape[:-1])
            gs = tuple(obs.shape)
            if not (len(gs) >= 1 and gs[-1] == lazy.phi.shape[1] and
                    len(gs) - 1 <= len(batch) and
                    gs[:-1] == batch[len(batch) - (len(gs) - 1):]):
                return no('the counts do not line up with the trailing '
                          'chain axes')
            return 'mixture_multinomial', priors, [lazy.phi], obs
        return no('likelihood %s has no native kernel' % type(ld).__name__)

    analyse.accepted = []
    found = analyse([q.detach().requires_grad_(True) for q in values])
    if found is None:
        return None
    kind = found[0]

    # The per-run re-evaluation of the model function only has to find the
    # parameter tensors again, so it is given META tensors for the latents:
    # whatever the function computes from them before the lazy contraction
    # (torch.softmax(eta, -1), lntm_mcem.py:39) is shape arithmetic, not a
    # launch and not a [rows, K] temporary on the device.  A function that
    # does more with a latent than that (mixes it with device tensors)
    # fails on the meta tensor and is evaluated on the latents themselves
    # from then on.
    q_meta = [torch.empty_like(q, device='meta') for q in values]
    on_meta = [True]

    def probe():
        f = None
        if on_meta[0]:
            try:
                f = analyse(q_meta)
            except Exception:                            # noqa: BLE001
                f = None
            if f is None or f[0] != kind:
                on_meta[0], f = False, None
        if f is None:
            f = analyse(list(values))
        if f is None or f[0] != kind:
            raise ValueError(
                "HMC (native %s plan): the model changed structure between "
                "runs; build a new HMC." % kind)
        return f[1], f[2], f[3]

    # prior parameters that do not fit the row-period addressing (more axes
    # than the latent, leading axes that are neither 1 nor the chain axes,
    # different periods for different latents): the generic plan, not an
    # exception out of HMC.sample
    try:
        return _DenseLikelihoodPlan(hmc, names, values, chain_shape, device,
                                    probe, kind)
    except _Unsupported as e:
        return no(str(e))


def _sum_tree_leaves(lp):
    """The autograd leaves' grad_fns if `lp` is built from its differentiable
    inputs by nothing but additions (alpha = 1) and sums over axes -- the
    shape of pmf_hmc.py:135-141, `reduce_sum(log_pu) + reduce_sum(log_pv) +
    reduce_sum(log_pr)` -- else None.  Constant summands (no grad_fn) are
    invisible here; their value is checked numerically by the caller."""
    leaves = []

    def walk(fn):
        if fn is None:
            return True
        name = type(fn).__name__
        if name == 'AddBackward0':
            if getattr(fn, '_saved_alpha', 1) != 1:
                return False
            return all(walk(f) for f, _ in fn.next_functions)
        if name in ('SumBackward0', 'SumBackward1'):
            return all(walk(f) for f, _ in fn.next_functions)
        leaves.append(fn)
        return True

    fn = getattr(lp, 'grad_fn', None)
    if fn is None or not walk(fn):
        return None
    return leaves


def _try_gathered_dot_plan(hmc, meta_bn, names, values, chain_shape, device):
    """The rating model of pmf_hmc.py:19-31: ONE latent factor table
    [chains, n, D] with a Normal prior, an observed Normal node whose mean is
    sigmoid(gathered_dot(latent, ...)) (zs.gathered_dot, or the reference's
    two gathers, a product and a reduce_sum), any other observed Normal node
    as a constant, and a log-joint that is the plain sum of the nodes'
    log-densities over their non-chain axes (the default one, or
    pmf_hmc.py:135-141)."""
    state = {'dense': False}

    def no(reason):
        hmc._note_refusal(reason, loud=state['dense'])
        return None

    if not isinstance(meta_bn, MetaBayesianNet) or len(names) != 1:
        return None
    name, q = names[0], values[0]
    n_chain = len(chain_shape)
    if q.dim() != n_chain + 2 or q.dtype != torch.float32 or \
            not q.is_contiguous() or q.data_ptr() % 16 != 0:
        return None
    n_total = int(q.shape[-1]) * int(q.shape[-2])

    def nodes_of(val):
        bn = meta_bn.observe(**merge_dicts(
            {name: hmc._as_symbol(val)}, hmc._resolved_observed()))
        return bn, [n for n in bn.nodes.values()
                    if isinstance(n, StochasticTensor)]

    def parts(stoch, accepted):
        """(priors, inner, obs) from the nodes named in `accepted`."""
        by_name = {n.name: n for n in stoch}
        if any(k not in by_name for k in accepted):
            return None
        prior = by_name[name].dist
        lik_name = accepted[1]
        lik = by_name[lik_name]
        gd = _symbolic.gathered_dot_mean(lik.dist._mean)
        if type(prior) is not Normal or type(lik.dist) is not Normal or \
                gd is None or not lik.is_observed():
            return None
        consts = []
        for k in accepted[2:]:
            d = by_name[k].dist
            if type(d) is not Normal or not by_name[k].is_observed():
                return None
            consts.append((by_name[k].tensor, d.mean, d.given_spread))
        sel_lat, sel_other = (gd['su'], gd['sv']) if gd['side'] == 'u' \
            else (gd['sv'], gd['su'])
        inner = [gd['side'], gd['other'], sel_lat, sel_other,
                 lik.dist.given_spread, consts]
        return [(prior.mean, prior.given_spread)], inner, lik.tensor, gd

    # -- build-time analysis on the real latent, with the log-joint ----------
    probe_q = q.detach().requires_grad_(True)
    bn, stoch = nodes_of(probe_q)
    lat_nodes = [n for n in stoch if n.name == name]
    cands = [n for n in stoch if n.name != name and type(n.dist) is Normal
             and _symbolic.gathered_dot_mean(n.dist._mean) is not None]
    if len(lat_nodes) != 1 or len(cands) != 1:
        return None
    state['dense'] = True
    lik = cands[0]
    gd = _symbolic.gathered_dot_mean(lik.dist._mean)
    if gd['latent'] is not probe_q:
        return no('the gathered dot is not over the sampled latent')
    pd = lat_nodes[0].dist
    if type(pd) is not Normal or pd.use_path_derivative or \
            pd.mean.requires_grad or pd.given_spread[1].requires_grad:
        return no("the prior of '%s' is not a Normal with constant "
                  "parameters" % name)
    if lik.dist.given_spread[1].numel() != 1 or \
            lik.dist.given_spread[1].requires_grad:
        return no("the likelihood '%s' does not have ONE constant scale"
                  % lik.name)
    if n_total % 4 != 0:
        return no('a latent table of %d elements per chain (the native '
                  'gathered-dot plan needs a multiple of 4)' % n_total)
    lp = bn.log_joint()
    if tuple(lp.shape) != tuple(chain_shape):
        return no('the log-joint does not have the chain shape')
    leaves = _sum_tree_leaves(lp)
    want = {id(lat_nodes[0].__dict__.get('_cond_log_p').grad_fn)
            if lat_nodes[0].__dict__.get('_cond_log_p') is not None else None,
            id(lik.__dict__.get('_cond_log_p').grad_fn)
            if lik.__dict__.get('_cond_log_p') is not None else None}
    if leaves is None or None in want or len(leaves) != 2 or \
            {id(f) for f in leaves} != want:
        return no('the log-joint is not the plain sum of the prior and the '
                  "rating likelihood's log-densities (+ constants)")
    # constant summands: the other evaluated nodes (observed, no gradient)
    const_names = [n.name for n in stoch
                   if n is not lat_nodes[0] and n is not lik and
                   n.__dict__.get('_cond_log_p') is not None]
    for k in const_names:
        node = [n for n in stoch if n.name == k][0]
        if node.__dict__['_cond_log_p'].requires_grad:
            return no("node '%s' depends on the latent" % k)
        if type(node.dist) is not Normal or not node.is_observed() or \
                node.dist.given_spread[1].numel() != 1:
            return no("constant node '%s' is not an observed Normal with "
                      "one scale" % k)
    if len(const_names) > 1:
        return no('more than one constant node in the joint')
    accepted = [name, lik.name] + const_names
    lp_user = lp.detach().reshape(-1).to(torch.float32)

    q_meta = torch.empty_like(q, device='meta')
    on_meta = [True]

    def probe():
        f = None
        if on_meta[0]:
            try:
                f = parts(nodes_of(q_meta)[1], accepted)
            except Exception:                            # noqa: BLE001
                f = None
            if f is None:
                on_meta[0] = False
        if f is None:
            f = parts(nodes_of(q)[1], accepted)
        if f is None:
            raise ValueError(
                "HMC (native gathered_dot plan): the model changed structure "
                "between runs; build a new HMC.")
        return f[0], f[1], f[2]

    try:
        plan = _DenseLikelihoodPlan(hmc, names, values, chain_shape, device,
                                    probe, 'gathered_dot')
    except _Unsupported as e:
        return no(str(e))
    # the constants are invisible to the structural check: the native
    # log-joint at the current state must equal the user's
    stream = _capi.current_stream()
    plan._load_state(stream)
    plan._first_evaluation(plan.q_new, stream)
    plan._step(plan.q_new, plan.p, True, 0.0, 0.0, 0.0, plan.lp_new, None,
               stream, start=True)
    diff = float((plan.lp_new - lp_user).abs().max().item())
    scale = max(1.0, float(lp_user.abs().max().item()))
    if not diff <= 2e-5 * scale + 1e-3:
        return no('the log-joint holds terms the native plan does not '
                  'account for (native - user = %.3g)' % diff)
    return plan
