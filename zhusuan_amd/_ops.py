"""torch.autograd wrappers around the libzshmc.so log_prob kernels.

Forward and backward both run the hand-written HIP kernels
(csrc/distributions.hip); torch only provides device memory, the stream and
the tape that chains these ops with whatever deterministic torch ops the user
model contains (tf.gradients, reference hmc.py:430-432).
"""
import os

import torch

from .utils import broadcast_shapes

from . import _capi, _symbolic, _writes

_F32 = torch.float32


class _Function(torch.autograd.Function):
    """autograd.Function whose `apply` first replaces symbolic latent
    expressions (zhusuan_amd/_symbolic.py) by their values:
    `Function.apply` hands its arguments to `forward` WITHOUT going through
    `__torch_function__`, and a wrapper that reached `forward` would cut the
    tape between the op and the latent it stands for."""

    @classmethod
    def apply(cls, *args):
        return super(_Function, cls).apply(
            *[_symbolic.force(a) for a in args])


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "zhusuan_amd: log_prob / sampling kernels run on an MI355X "
                "only; got a %s tensor. There is no CPU fallback." % t.device)


def bcast_plan(full_shape, n_col_dims, *params):
    """For x viewed as [rows, cols] (cols = prod(full_shape[-n_col_dims:])),
    return [(tensor, mode)] for each param: SCALAR / ROW / FULL, materialising
    an expanded copy only when the param fits none of them."""
    nd = len(full_shape)
    col_shape = tuple(full_shape[nd - n_col_dims:])
    out = []
    for p in params:
        ps = (1,) * (nd - p.dim()) + tuple(p.shape)
        if p.numel() == 1:
            out.append((p.reshape(1).contiguous(), _capi.BCAST_SCALAR))
        elif (all(s == 1 for s in ps[:nd - n_col_dims]) and
              tuple(ps[nd - n_col_dims:]) == col_shape):
            out.append((p.reshape(-1).contiguous(), _capi.BCAST_ROW))
        elif tuple(ps) == tuple(full_shape):
            out.append((p.contiguous(), _capi.BCAST_FULL))
        else:
            out.append((p.expand(full_shape).contiguous(), _capi.BCAST_FULL))
    return out


def choose_col_dims(full_shape, group_ndims, *params):
    """Number of trailing dims folded into `cols`.  With group_ndims > 0 it
    is group_ndims (the kernel sums each row).  Otherwise pick the widest
    suffix for which no param needs materialising."""
    nd = len(full_shape)
    if group_ndims > 0:
        return group_ndims
    if nd == 0:
        return 0
    for n in range(nd, 0, -1):
        ok = True
        for p in params:
            ps = (1,) * (nd - p.dim()) + tuple(p.shape)
            if p.numel() == 1 or tuple(ps) == tuple(full_shape):
                continue
            if (all(s == 1 for s in ps[:nd - n]) and
                    tuple(ps[nd - n:]) == tuple(full_shape[nd - n:])):
                continue
            ok = False
            break
        if ok:
            return n
    return 1


def _rows_cols(full_shape, n_col_dims):
    nd = len(full_shape)
    cols = 1
    for s in full_shape[nd - n_col_dims:]:
        cols *= int(s)
    rows = 1
    for s in full_shape[:nd - n_col_dims]:
        rows *= int(s)
    if cols == 0:          # empty input: nothing to launch
        rows = 0
    return rows, max(cols, 1)


def _sum_to(g_full, target_shape):
    """Reduce a full-shape gradient back to a broadcast parameter's shape."""
    return g_full.sum_to_size(target_shape) if tuple(g_full.shape) != tuple(
        target_shape) else g_full


class NormalLogProb(_Function):
    """Normal._log_prob + group_ndims sum (reference
    distributions/univariate.py:174-181, base.py:302-304)."""

    @staticmethod
    def forward(ctx, x, mean, logstd, group_ndims):
        require_device(x, mean, logstd)
        full = broadcast_shapes(x.shape, mean.shape, logstd.shape)
        n_col = choose_col_dims(full, group_ndims, mean, logstd)
        rows, cols = _rows_cols(full, n_col)
        xf = x.expand(full).contiguous()
        (m, mm), (s, sm) = bcast_plan(full, n_col, mean, logstd)
        reduce_cols = 1 if group_ndims > 0 else 0
        out_shape = full[:len(full) - group_ndims] if reduce_cols else full
        out = torch.empty(out_shape, dtype=_F32, device=x.device)
        _capi.call('zshmc_normal_log_prob', xf.data_ptr(), m.data_ptr(),
                   s.data_ptr(), out.data_ptr(), rows, cols, mm, sm,
                   reduce_cols, _capi.current_stream())
        ctx.save_for_backward(xf, m, s)
        ctx.meta = (full, rows, cols, mm, sm, reduce_cols, x.shape,
                    mean.shape, logstd.shape)
        return out

    @staticmethod
    def backward(ctx, gout):
        xf, m, s = ctx.saved_tensors
        full, rows, cols, mm, sm, reduce_cols, xs, ms, ss = ctx.meta
        need_x, need_m, need_s = ctx.needs_input_grad[:3]
        gout = gout.contiguous()
        gx = torch.empty(full, dtype=_F32, device=xf.device) if need_x else None
        gm = torch.empty(full, dtype=_F32, device=xf.device) if need_m else None
        gs = torch.empty(full, dtype=_F32, device=xf.device) if need_s else None
        _capi.call('zshmc_normal_log_prob_grad', xf.data_ptr(), m.data_ptr(),
                   s.data_ptr(), gout.data_ptr(), _capi.ptr(gx),
                   _capi.ptr(gm), _capi.ptr(gs), rows, cols, mm, sm,
                   reduce_cols, _capi.current_stream())
        return (_sum_to(gx, xs) if need_x else None,
                _sum_to(gm, ms) if need_m else None,
                _sum_to(gs, ss) if need_s else None, None)


class Uni2LogProb(_Function):
    """Laplace / Gamma / InverseGamma / Beta `_log_prob` + group_ndims sum
    (reference distributions/univariate.py:1268-1275, :735-748, :1145-1157,
    :834-853; base.py:302-304) and their analytic gradients w.r.t. the value
    and both parameters (csrc/distributions2.hip)."""

    @staticmethod
    def forward(ctx, x, a, b, kind, group_ndims):
        require_device(x, a, b)
        full = broadcast_shapes(x.shape, a.shape, b.shape)
        n_col = choose_col_dims(full, group_ndims, a, b)
        rows, cols = _rows_cols(full, n_col)
        xf = x.expand(full).contiguous()
        (pa, ma), (pb, mb) = bcast_plan(full, n_col, a, b)
        reduce_cols = 1 if group_ndims > 0 else 0
        out_shape = full[:len(full) - group_ndims] if reduce_cols else full
        out = torch.empty(out_shape, dtype=_F32, device=x.device)
        _capi.call('zshmc_uni2_log_prob', int(kind), xf.data_ptr(),
                   pa.data_ptr(), pb.data_ptr(), out.data_ptr(), rows, cols, ma,
                   mb, reduce_cols, _capi.current_stream())
        ctx.save_for_backward(xf, pa, pb)
        ctx.meta = (int(kind), full, rows, cols, ma, mb, reduce_cols, x.shape,
                    a.shape, b.shape)
        return out

    @staticmethod
    def backward(ctx, gout):
        xf, pa, pb = ctx.saved_tensors
        kind, full, rows, cols, ma, mb, reduce_cols, xs, as_, bs = ctx.meta
        need_x, need_a, need_b = ctx.needs_input_grad[:3]
        gout = gout.contiguous()
        mk = lambda need: torch.empty(full, dtype=_F32, device=xf.device) \
            if need else None
        gx, ga, gb = mk(need_x), mk(need_a), mk(need_b)
        _capi.call('zshmc_uni2_log_prob_grad', kind, xf.data_ptr(),
                   pa.data_ptr(), pb.data_ptr(), gout.data_ptr(), _capi.ptr(gx),
                   _capi.ptr(ga), _capi.ptr(gb), rows, cols, ma, mb,
                   reduce_cols, _capi.current_stream())
        return (_sum_to(gx, xs) if need_x else None,
                _sum_to(ga, as_) if need_a else None,
                _sum_to(gb, bs) if need_b else None, None, None)


_csr_cache = {}


def _pair_csr(index, n_rows, slot):
    """CSR view of a 1-D index tensor: (int32 index, seg_ptr [n_rows + 1],
    order [E]) with `order` the pair ids stably sorted by the row they point
    at.  Validates the range once per tensor version (torch.gather semantics:
    out-of-range is an error, not a wrap).  Cached while the same tensor is
    passed again -- every leapfrog step of a transition reuses the pair list;
    the entry pins the tensor so its address cannot be recycled."""
    key = (index.data_ptr(), tuple(index.shape), index._version, int(n_rows),
           index.dtype, _writes.generation(index))
    hit = _csr_cache.get(slot)
    if hit is not None and hit[0] == key:
        return hit[1]
    if index.dim() != 1 or index.dtype not in (torch.int32, torch.int64):
        raise TypeError("gathered_dot: indices must be 1-D int32/int64 "
                        "tensors, got {} {}".format(tuple(index.shape),
                                                    index.dtype))
    if index.numel() and (int(index.min()) < 0 or
                          int(index.max()) >= n_rows):
        raise IndexError("gathered_dot: index out of range [0, {})"
                         .format(n_rows))
    i32 = index.to(torch.int32).contiguous()
    order = torch.sort(index.to(torch.int64), stable=True)[1].to(torch.int32)
    counts = torch.bincount(index.to(torch.int64), minlength=n_rows)
    seg = torch.zeros(n_rows + 1, dtype=torch.int32, device=index.device)
    seg[1:] = torch.cumsum(counts, 0).to(torch.int32)
    val = (i32, seg, order.contiguous())
    _csr_cache[slot] = (key, val, index)
    return val


GD_SEGMENT_PAIRS = 256      # csrc/gather_dot.hip: kGdSegPairs


def _csr_segments(seg, n_pairs):
    """The CSR row ranges `seg` [n_rows + 1] cut into segments of at most
    GD_SEGMENT_PAIRS slots (zshmc_gather_dot_normal_lik_grad): (seg_ptr
    [n_seg + 1], seg_row [n_seg], seg_first [n_rows], long_rows [n_long]),
    every row with at least one (possibly empty) segment."""
    counts = (seg[1:] - seg[:-1]).to(torch.int64)
    n_rows = counts.numel()
    nseg = torch.clamp((counts + GD_SEGMENT_PAIRS - 1) // GD_SEGMENT_PAIRS,
                       min=1)
    first = torch.cumsum(nseg, 0) - nseg
    rows = torch.repeat_interleave(
        torch.arange(n_rows, device=seg.device), nseg)
    within = torch.arange(rows.numel(), device=seg.device) - first[rows]
    start = seg[:-1].to(torch.int64)[rows] + GD_SEGMENT_PAIRS * within
    i32 = torch.int32
    seg_ptr = torch.cat([start, torch.tensor([n_pairs], device=seg.device)])
    return (seg_ptr.to(i32).contiguous(), rows.to(i32).contiguous(),
            first.to(i32).contiguous(),
            torch.nonzero(nseg > 1).reshape(-1).to(i32).contiguous())


class GatheredDot(_Function):
    """out[..., e] = sum_d u[..., su[e], d] * v[..., sv[e], d]
    (examples/probabilistic_matrix_factorization/pmf_hmc.py:26-28 without the
    [K, E, D] gathers) and its gradients by deterministic segmented sums
    (csrc/gather_dot.hip)."""

    @staticmethod
    def forward(ctx, u, su, v, sv):
        require_device(u, v, su, sv)
        if u.dim() < 2 or v.dim() < 2 or u.shape[:-2] != v.shape[:-2] or \
                u.shape[-1] != v.shape[-1]:
            raise ValueError("gathered_dot: u [..., n, D] and v [..., m, D] "
                             "with equal leading axes expected, got {} and {}"
                             .format(tuple(u.shape), tuple(v.shape)))
        if su.shape != sv.shape:
            raise ValueError("gathered_dot: select_u and select_v differ in "
                             "shape")
        n_u, n_v, D = u.shape[-2], v.shape[-2], u.shape[-1]
        uf = u.detach().to(_F32).contiguous()
        vf = v.detach().to(_F32).contiguous()
        K = uf.numel() // max(n_u * D, 1)
        csr_u = _pair_csr(su, n_u, 'u')
        csr_v = _pair_csr(sv, n_v, 'v')
        E = csr_u[0].numel()
        out = torch.empty(tuple(u.shape[:-2]) + (E,), dtype=_F32,
                          device=u.device)
        ctx.save_for_backward(uf, vf, *csr_u, *csr_v)
        ctx.dims = (K, n_u, n_v, E, D, tuple(u.shape), tuple(v.shape))
        if out.numel() == 0:          # no pairs (or no chains): nothing to launch
            return out
        _capi.call('zshmc_gather_dot', uf.data_ptr(), vf.data_ptr(),
                   csr_u[0].data_ptr(), csr_v[0].data_ptr(), K, n_u, n_v, E, D,
                   out.data_ptr(), _capi.current_stream())
        return out

    @staticmethod
    def backward(ctx, gout):
        uf, vf, iu, seg_u, ord_u, iv, seg_v, ord_v = ctx.saved_tensors
        K, n_u, n_v, E, D, us, vs = ctx.dims
        g = gout.to(_F32).contiguous()
        stream = _capi.current_stream()
        gu = gv = None
        if E == 0 or K == 0:
            return (torch.zeros(us, dtype=_F32, device=g.device)
                    if ctx.needs_input_grad[0] else None, None,
                    torch.zeros(vs, dtype=_F32, device=g.device)
                    if ctx.needs_input_grad[2] else None, None)
        if ctx.needs_input_grad[0]:
            gu = torch.empty(us, dtype=_F32, device=g.device)
            _capi.call('zshmc_gather_dot_grad', vf.data_ptr(), g.data_ptr(),
                       seg_u.data_ptr(), ord_u.data_ptr(), iv.data_ptr(), K,
                       n_u, n_v, E, D, gu.data_ptr(), stream)
        if ctx.needs_input_grad[2]:
            gv = torch.empty(vs, dtype=_F32, device=g.device)
            _capi.call('zshmc_gather_dot_grad', uf.data_ptr(), g.data_ptr(),
                       seg_v.data_ptr(), ord_v.data_ptr(), iu.data_ptr(), K,
                       n_v, n_u, E, D, gv.data_ptr(), stream)
        return gu, None, gv, None


def clear_caches():
    """Drop the single-entry operand caches (padded design matrix, padded
    phi^T, CSR views of index lists).  Each entry pins the tensor it was built
    from so that an address is never mistaken for another tensor's; call this
    to release those references (e.g. a 1 GB design matrix) when a model is
    done."""
    _x_cache.clear()
    _design_cache.clear()
    _phi_cache.clear()
    _counts_cache.clear()
    _csr_cache.clear()
    _image_cache.clear()
    _label_cache.clear()
    _counts_csr_cache.clear()


def _storages(x, out):
    if torch.is_tensor(x):
        out.add((str(x.device), x.untyped_storage().data_ptr()))
    elif isinstance(x, (tuple, list)):
        for y in x:
            _storages(y, out)
    return out


def forget(tensors):
    """Drop the cached operands that were built from any of `tensors` (or
    from a view of the same storage), and what was built from THOSE in turn
    (the bf16x3 image of a padded copy) -- the other models' entries stay.
    What a sampler with reuse_start_evaluation=False does before every run:
    nothing about ITS model's tensors is remembered."""
    gone = _storages(list(tensors), set())
    caches = (_x_cache, _design_cache, _phi_cache, _counts_cache, _csr_cache,
              _image_cache, _label_cache, _counts_csr_cache)
    again = True
    while again:
        again = False
        for c in caches:
            slotted = isinstance(c, dict)       # {slot: (key, value, pins)}
            for ref, item in list(c.items()) if slotted else \
                    [(i, it) for i, it in enumerate(c.items)][::-1]:
                if _storages(item[2], set()) & gone:
                    n = len(gone)
                    _storages(item[1], gone)
                    again = again or len(gone) != n
                    if slotted:
                        del c[ref]
                    else:
                        del c.items[ref]


def gathered_dot(u, select_u, v, select_v):
    """`reduce_sum(gather(u, select_u, axis=-2) * gather(v, select_v, axis=-2),
    axis=-1)`: u [..., n, D], v [..., m, D] with equal leading (chain) axes,
    1-D integer index tensors of equal length E; returns [..., E]."""
    u = u.tensor if hasattr(u, 'tensor') else u
    v = v.tensor if hasattr(v, 'tensor') else v
    # one table a symbolic latent (HMC hands its latents over as symbols):
    # stay symbolic, so that Normal(sigmoid(.)) can be lowered to the native
    # gathered-dot plan; forced, the symbol is this very op
    sym = _symbolic.make_gdot(u, select_u, v, select_v)
    if sym is not None:
        return sym
    return GatheredDot.apply(u, select_u, v, select_v)


class MvnTrilLogProb(_Function):
    """MultivariateNormalCholesky._log_prob (reference distributions/
    multivariate.py:166-188) and its gradients: d/dgiven = -L^-T z from the
    kernel's back substitution; d/dmean = -d/dgiven; d/dL = tril(w z^T) -
    diag(1/L_ii) with w = L^-T z (csrc/mvn.hip).
    x [..., D] already broadcast to its full shape; mean [*mb, D] and
    tril [*mb, D, D] with `mb` a suffix of x's batch axes (shared by the
    leading axes through a row period)."""

    @staticmethod
    def forward(ctx, x, mean, tril):
        require_device(x, mean, tril)
        D = x.shape[-1]
        xf = x.detach().to(_F32).contiguous()
        mf = mean.detach().to(_F32).contiguous()
        tf_ = tril.detach().to(_F32).contiguous()
        rows = xf.numel() // D if D else 0
        mean_rows = max(mf.numel() // D, 1)
        tril_count = max(tf_.numel() // (D * D), 1)
        out = torch.empty(x.shape[:-1], dtype=_F32, device=x.device)
        need = any(ctx.needs_input_grad[:3])
        need_p = ctx.needs_input_grad[2]
        gx = torch.empty_like(xf) if need else None
        z = torch.empty_like(xf) if need_p else None
        _capi.call('zshmc_mvn_tril_log_prob', xf.data_ptr(), mf.data_ptr(),
                   tf_.data_ptr(), rows, D, mean_rows, tril_count,
                   out.data_ptr(), _capi.ptr(gx), _capi.ptr(z),
                   _capi.current_stream())
        ctx.shapes = (tuple(x.shape), tuple(mean.shape), tuple(tril.shape))
        if need:
            ctx.save_for_backward(gx, z, tf_) if need_p else \
                ctx.save_for_backward(gx)
        ctx.need_p = need_p
        return out

    @staticmethod
    def backward(ctx, gout):
        xs, ms, ts = ctx.shapes
        if ctx.need_p:
            gx, z, tril = ctx.saved_tensors
        else:
            (gx,) = ctx.saved_tensors
        g = gx * gout.unsqueeze(-1)
        need_x, need_m, need_t = ctx.needs_input_grad[:3]
        gm = _sum_to(-g, ms) if need_m else None
        gt = None
        if need_t:
            # d/dL = gout * (tril(w z^T) - diag(1/L_ii)), w = -gx
            outer = torch.tril((-g).unsqueeze(-1) * z.unsqueeze(-2))
            diag = torch.diag_embed(
                gout.unsqueeze(-1) / torch.diagonal(tril, dim1=-2, dim2=-1)
                .expand(xs))
            gt = _sum_to(outer - diag, ts)
        return (g if need_x else None), gm, gt


class BernoulliLogProb(_Function):
    """Bernoulli._log_prob (reference univariate.py:398-403)."""

    @staticmethod
    def forward(ctx, logits, given, group_ndims):
        require_device(logits, given)
        full = broadcast_shapes(logits.shape, given.shape)
        n_col = choose_col_dims(full, group_ndims, logits, given)
        rows, cols = _rows_cols(full, n_col)
        (l, lm), (z, zm) = bcast_plan(full, n_col, logits, given)
        reduce_cols = 1 if group_ndims > 0 else 0
        out_shape = full[:len(full) - group_ndims] if reduce_cols else full
        out = torch.empty(out_shape, dtype=_F32, device=logits.device)
        _capi.call('zshmc_bernoulli_log_prob', l.data_ptr(), z.data_ptr(),
                   out.data_ptr(), rows, cols, lm, zm, reduce_cols,
                   _capi.current_stream())
        ctx.save_for_backward(l, z)
        ctx.meta = (full, rows, cols, lm, zm, reduce_cols, logits.shape)
        return out

    @staticmethod
    def backward(ctx, gout):
        l, z = ctx.saved_tensors
        full, rows, cols, lm, zm, reduce_cols, ls = ctx.meta
        if not ctx.needs_input_grad[0]:
            return None, None, None
        gl = torch.empty(full, dtype=_F32, device=l.device)
        _capi.call('zshmc_bernoulli_log_prob_grad', l.data_ptr(),
                   z.data_ptr(), gout.contiguous().data_ptr(), gl.data_ptr(),
                   rows, cols, lm, zm, reduce_cols, _capi.current_stream())
        return _sum_to(gl, ls), None, None


class CategoricalLogProb(_Function):
    """Categorical._log_prob (reference univariate.py:496-548); logits
    [rows, n_cat] and int64 labels [rows], already broadcast."""

    @staticmethod
    def forward(ctx, logits, labels):
        require_device(logits, labels)
        rows, n_cat = logits.shape
        out = torch.empty((rows,), dtype=_F32, device=logits.device)
        _capi.call('zshmc_categorical_log_prob', logits.data_ptr(),
                   labels.data_ptr(), out.data_ptr(), rows, n_cat,
                   _capi.current_stream())
        ctx.save_for_backward(logits, labels)
        return out

    @staticmethod
    def backward(ctx, gout):
        logits, labels = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None
        rows, n_cat = logits.shape
        gl = torch.empty_like(logits)
        _capi.call('zshmc_categorical_log_prob_grad', logits.data_ptr(),
                   labels.data_ptr(), gout.contiguous().data_ptr(),
                   gl.data_ptr(), rows, n_cat, _capi.current_stream())
        return gl, None


class UnnormalizedMultinomialLogProb(_Function):
    """UnnormalizedMultinomial._log_prob (reference
    distributions/multivariate.py:435-443)."""

    @staticmethod
    def forward(ctx, logits, given, normalize):
        require_device(logits, given)
        rows, n_cat = logits.shape
        out = torch.empty((rows,), dtype=_F32, device=logits.device)
        _capi.call('zshmc_unnormalized_multinomial_log_prob',
                   logits.data_ptr(), given.data_ptr(), out.data_ptr(), rows,
                   n_cat, int(bool(normalize)), _capi.current_stream())
        ctx.save_for_backward(logits, given)
        ctx.normalize = int(bool(normalize))
        return out

    @staticmethod
    def backward(ctx, gout):
        logits, given = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None
        rows, n_cat = logits.shape
        gl = torch.empty_like(logits)
        _capi.call('zshmc_unnormalized_multinomial_log_prob_grad',
                   logits.data_ptr(), given.data_ptr(),
                   gout.contiguous().data_ptr(), gl.data_ptr(), rows, n_cat,
                   ctx.normalize, _capi.current_stream())
        return gl, None, None


# ----------------------------------------------------------------------------
# dense-logit Bernoulli likelihood (fp32 MFMA, csrc/linear_bernoulli.hip)
# ----------------------------------------------------------------------------
# The padded feature / topic count of the fused likelihood kernels, and the
# chains one workgroup takes, are the library's to say (zshmc_likelihood_plan:
# csrc/linear_bernoulli.hip up to 256 columns, csrc/linear_bernoulli_mid.hip up
# to 576, csrc/linear_bernoulli_wide.hip above); the host only knows the widest.
MAX_LIKELIHOOD_WIDTH = 1024
_plan_cache = {}


def likelihood_plan(n, class_stride=0):
    """(kernel width, chains per workgroup) for rows of n features / topics;
    class_stride: the Categorical family's padded class count, else 0."""
    key = (int(n), int(class_stride))
    hit = _plan_cache.get(key)
    if hit is None:
        import ctypes
        width, block = ctypes.c_int64(0), ctypes.c_int(0)
        _capi.call('zshmc_likelihood_plan', key[0], key[1],
                   ctypes.byref(width), ctypes.byref(block))
        hit = _plan_cache[key] = (int(width.value), int(block.value))
    return hit


def likelihood_width(n, class_stride=0):
    return likelihood_plan(n, class_stride)[0]


def _pad_features(t, width):
    d = t.shape[-1]
    if d == width and t.is_contiguous():
        return t
    out = torch.zeros(t.shape[:-1] + (width,), dtype=_F32, device=t.device)
    out[..., :d] = t
    return out


class _Lru(object):
    """A few entries keyed by what identifies the source tensors' contents
    (address, layout, dtype, device, version counter).  An entry pins its
    sources, so an address in a live key cannot have been handed to another
    tensor.  Several entries, because one model is evaluated through more
    than one padded width (LinearLogits.packed() on the generic plan, the
    native plan's kernel width) and two models may alternate: a single slot
    rebuilt a copy of up to a GB on every call."""

    def __init__(self, size=4):
        self.size, self.items = size, []

    def get(self, key):
        for i, (k, val, _) in enumerate(self.items):
            if k == key:
                self.items.insert(0, self.items.pop(i))
                return val
        return None

    def put(self, key, val, pins):
        self.items.insert(0, (key, val, pins))
        del self.items[self.size:]

    def clear(self):
        del self.items[:]


def _tensor_key(t):
    # torch's version counter sees torch writes; a sampler that moved the
    # tensor as ITS latent wrote through the C-ABI and noted it in _writes
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype,
            str(t.device), t._version, _writes.generation(t))


_x_cache = _Lru()
_label_cache = _Lru()


def _padded_x(X, width):
    """Zero-padded contiguous copy of the design matrix, cached per tensor
    version (the model builder re-runs on every joint evaluation)."""
    if X.shape[-1] == width and X.is_contiguous() and X.dtype == _F32:
        return X
    key = (_tensor_key(X), width)
    hit = _x_cache.get(key)
    if hit is not None:
        return hit
    Xp = _pad_features(X.detach().to(_F32), width)
    _x_cache.put(key, Xp, X)
    return Xp


_design_cache = _Lru()


def packed_design(blocks, n_rows, device, width=None):
    """The design matrices of a multi-term linear_logits side by side,
    [n_rows, sum D_k] float32 (a column of ones where a block is None: the
    bias), zero-padded to `width` columns if given; cached while the SAME
    tensors (storage, version) are passed again."""
    key = tuple(None if X is None else _tensor_key(X) for X in blocks) + (
        width, int(n_rows), str(device))
    hit = _design_cache.get(key)
    if hit is not None:
        return hit
    total = sum(1 if X is None else int(X.shape[-1]) for X in blocks)
    out = torch.zeros(n_rows, width or total, dtype=_F32, device=device)
    off = 0
    for X in blocks:
        d = 1 if X is None else int(X.shape[-1])
        if X is None:
            out[:, off] = 1.0
        else:
            out[:, off:off + d] = X.detach().to(_F32)
        off += d
    _design_cache.put(key, out, list(blocks))
    return out


# ---- the bf16x3 likelihood kernels (csrc/b3_kernel.h) ---------------
BF16X3_WIDTHS = (64, 128, 192, 256)
BF16X3_CHAIN_BLOCK = 128
# A workgroup of the mixture-multinomial kernel takes 128 chains of ONE
# document: by default the kernels are taken only where the chain axis fills
# those workgroups (a multiple of 128, or >= 1024 chains: >= 89 % of the
# slots); other chain axes run the packed-rows form below, or -- small and
# sparse, like lntm_mcem.py's own E-step (n_chains = 1) -- the row-by-row
# fp32 kernel (SPARSE_ROWS_*).
BF16X3_REQUIRE_FILL = True
# ... unless the packed-rows form of the kernel takes them (ABI 0.5.1: 128
# consecutive (chain, document) rows per workgroup, each with its own counts
# row): <= 192 topics, counts rows padded to 32 floats, counts below 4 GB
BF16X3_PACKED_MAX_WIDTH = 192
# The one-document-per-workgroup form runs over the document's OWN vocabulary
# (ABI 0.6.0: words with a zero count contribute exactly nothing) when the
# padded word lists are at most this share of the dense [documents, V] counts
BF16X3_SPARSE_MAX_FILL = 0.6
# ... also for chain axes that do NOT fill those workgroups, from this many
# chains per document on, when fill * 128 / chains_per_doc stays below the same
# bound (a partly filled workgroup over few words against 128 packed rows over
# all of them)
BF16X3_SPARSE_MIN_CHAINS = 8
# Small topic models on the exact-fp32 path -- the reference's own loop is ONE
# chain x a minibatch of 100 documents (lntm_mcem.py:62-70) -- run row by row
# over each row's own words on the vector ALU (csrc/sparse_multinomial.hip):
# up to this many (chain, document) rows (beyond, phi^T rows gathered per row
# out of L2 cost more than the matrix cores' dense tiles) and this fill
# (1 chain x R documents, K = 128, ~970 of 12 419 words per document,
# gradient-only launch, tools/b3_packed_bench.py, gpurun r06: R = 2 048
# 0.083 ms against 0.82 fp32 MFMA / 0.74 packed-rows bf16x3; 8 192: 0.335 /
# 0.84 / 0.77; 32 768: 1.30 / 1.50 / 0.955)
SPARSE_ROWS_MAX = 32768
SPARSE_ROWS_MAX_FILL = 0.5
# 'auto' keeps such a problem on it -- exact fp32 AND faster -- up to this many
# rows when the bf16x3 alternative is the packed-rows form (chain axes that do
# not fill one-document workgroups); beyond, the packed-rows kernel wins
SPARSE_ROWS_AUTO_MAX = 16384
# likelihood_arithmetic='auto' (the default) takes them from this many flop
# per evaluation (4 N D R over all ranks' rows) on: ~0.1 ms of the fp32 matrix
# peak.  Below, a transition is bound by its kernels' critical paths and the
# fp32 kernels' 16- / 32- / 64-row blocks spread it over more CUs.
BF16X3_AUTO_MIN_FLOP = 1.0e10
_image_cache = _Lru()


def bf16x3_image(Xp):
    """The tile image of a padded float32 operand [rows, width] (three
    bfloat16 planes per element; zshmc_bf16x3_split), cached per tensor
    version like the padded operand itself."""
    import ctypes
    rows, width = int(Xp.shape[0]), int(Xp.shape[1])
    if width not in BF16X3_WIDTHS:
        raise ValueError('bf16x3 kernels take widths %s, got %d'
                         % (BF16X3_WIDTHS, width))
    key = _tensor_key(Xp)
    hit = _image_cache.get(key)
    if hit is not None:
        return hit
    nbytes = ctypes.c_int64()
    _capi.call('zshmc_bf16x3_image_bytes', rows, width,
               ctypes.addressof(nbytes))
    img = torch.empty(nbytes.value, dtype=torch.uint8, device=Xp.device)
    _capi.call('zshmc_bf16x3_split', Xp.data_ptr(), rows, width,
               Xp.stride(0), img.data_ptr(), _capi.current_stream())
    _image_cache.put(key, img, Xp)
    return img


def resident_per_cu(width, arithmetic='fp32'):
    """Workgroups of the <= 256-column likelihood kernels one CU holds
    (registers / LDS; csrc/lb_body.h ZS_LB_MINW, csrc/b3_kernel.h
    ZS_B3_WAVES): a second resident workgroup runs its element-wise stage
    under the first one's MFMAs."""
    if width > 256:
        return 1
    if arithmetic == 'bf16x3':
        return 2 if width <= 128 else 1
    return 3 if width <= 64 else 2 if width <= 128 else 1


def _row_splits(n_blocks_rows, n_inner, device, block=64, per_cu=1):
    """Fewer chain blocks (`block` rows: zshmc_likelihood_plan) than compute
    units: cut the inner (data row / vocabulary) range so that about two
    workgroups land on every CU, at least 256 inner rows per slice, at most
    32 slices.  (Round 4, the E-step of lntm_mcem.py:157-182 -- 100 documents
    x 100 topics x 12 419 words, two chain blocks: 16 slices of >= 512 rows
    1 788 us per L = 20 transition, 32 of >= 256 rows 1 232 us, 64 of >= 128
    rows 1 213 us; profiles/archive/r04e_row_splits_ab.txt.)"""
    n_wg = (n_blocks_rows + block - 1) // block
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    if n_wg >= cus * per_cu:
        return 1
    if n_wg >= cus:
        # one wave of workgroups on a kernel that holds `per_cu` per CU: just
        # enough slices to give every CU its partner(s) (bf16x3 at 128
        # columns: +9 % with two resident, profiles/r05q_*)
        return max(1, min((cus * per_cu) // n_wg, (n_inner + 127) // 128))
    # Round 5 (profiles/r05k_estep_kernel_trace.txt): at the E-step's shape a
    # transition is NOT launch-bound -- its 32-slice likelihood launches are
    # 36 us each, six 64-row tiles on the critical path of every workgroup,
    # and the one-thread-per-element serial reduction of the partials 9 us.
    # With the partials added eight loads at a time (sum_parts8) finer slices
    # pay: two tiles (128 rows) per slice, about two workgroups per CU.
    # (E-step, gpurun r05: 256 rows per slice 0.743 ms per transition, 128
    # rows 0.639, 64 rows 0.792 -- the reduction and a second round of
    # workgroups eat the shorter critical path; ZSHMC_MIN_SLICE_ROWS for A/B)
    min_rows = int(os.environ.get('ZSHMC_MIN_SLICE_ROWS', '128'))
    return max(1, min(256, (2 * cus) // n_wg,
                      (n_inner + min_rows - 1) // min_rows))


class LinearBernoulliLogLik(_Function):
    """ll[c] = sum_n Bernoulli(w_c . x_n).log_prob(y_n) and its gradient in one
    pass over X (logits are never materialised)."""

    @staticmethod
    def forward(ctx, w, X, y):
        require_device(w, X, y)
        d = w.shape[-1]
        width, block = likelihood_plan(d)
        w2 = _pad_features(w.detach().reshape(-1, d).to(_F32), width)
        Xp = _padded_x(X, width)
        yf = y.detach().to(_F32).contiguous()
        C, N = w2.shape[0], Xp.shape[0]
        ll = torch.empty(C, dtype=_F32, device=w.device)
        need_grad = ctx.needs_input_grad[0]
        gw = torch.empty_like(w2) if need_grad else None
        splits = _row_splits(C, N, w.device, block)
        ws = torch.empty(splits * C * (width + 1), dtype=_F32,
                         device=w.device) if splits > 1 else None
        _capi.call('zshmc_linear_bernoulli_log_lik', w2.data_ptr(),
                   Xp.data_ptr(), yf.data_ptr(), C, N, width, ll.data_ptr(),
                   _capi.ptr(gw), splits, _capi.ptr(ws),
                   _capi.current_stream())
        ctx.w_shape = tuple(w.shape)
        if need_grad:
            ctx.save_for_backward(gw)
        return ll.reshape(w.shape[:-1])

    @staticmethod
    def backward(ctx, gout):
        (gw,) = ctx.saved_tensors
        d = ctx.w_shape[-1]
        g = gw[:, :d] * gout.reshape(-1, 1)
        return g.reshape(ctx.w_shape), None, None


# ----------------------------------------------------------------------------
# dense-logit Categorical likelihood (softmax regression; OP = 2 of the same
# fp32-MFMA kernels, csrc/lb_ops.h)
# ----------------------------------------------------------------------------
MAX_CLASSES = 32          # class stride: a power of two <= 32 lanes


def class_stride(n_classes):
    """Rows of the kernel's W operand per chain: the class count rounded up
    to a power of two (the classes of a chain sit in that many consecutive
    lanes of the logits accumulator)."""
    g = 1
    while g < n_classes:
        g *= 2
    return g


def pack_class_rows(w, stride, width):
    """w [C, K, F] -> the kernel's operand [C * stride, width]: row
    c * stride + k = w[c, k, :], zero padding rows / columns."""
    C, K, F = w.shape
    if K == stride and F == width and w.is_contiguous():
        return w.reshape(C * K, F)
    out = torch.zeros(C, stride, width, dtype=_F32, device=w.device)
    out[:, :K, :F] = w
    return out.reshape(C * stride, width)


def labels_as_float(y, n_classes):
    """Class labels [N] (any int / float dtype) as the float32 vector the
    kernel compares its lane's class with; out-of-range labels are an error
    (tf.nn.sparse_softmax_cross_entropy_with_logits raises / returns NaN)."""
    key = (_tensor_key(y), int(n_classes))
    hit = _label_cache.get(key)
    if hit is not None:        # validated once per tensor version: the check
        return hit             # reads the device (two host syncs)
    yl = y.detach().reshape(-1)
    yf = yl.to(_F32).contiguous()
    if yl.numel():
        lo, hi = float(yf.min()), float(yf.max())
        integral = True if not yl.dtype.is_floating_point else \
            bool((yf == yf.round()).all())
        if lo < 0 or hi >= n_classes or not integral:
            raise ValueError("Categorical: labels must be integers in [0, {})"
                             .format(n_classes))
    _label_cache.put(key, yf, y)
    return yf


class LinearCategoricalLogLik(_Function):
    """ll[c] = sum_n Categorical(X w_c^T).log_prob(y_n) for w [..., K, F] and
    its gradient in one pass over X; the [..., N, K] logits are never
    materialised (zshmc_linear_categorical_log_lik)."""

    @staticmethod
    def forward(ctx, w, X, labels_f):
        require_device(w, X, labels_f)
        K, F = int(w.shape[-2]), int(w.shape[-1])
        G = class_stride(K)
        width, block = likelihood_plan(F, G)
        w3 = w.detach().reshape(-1, K, F).to(_F32)
        C = w3.shape[0]
        wp = pack_class_rows(w3, G, width)
        Xp = _padded_x(X, width)
        N = Xp.shape[0]
        ll = torch.empty(C * G, dtype=_F32, device=w.device)
        need_grad = ctx.needs_input_grad[0]
        gw = torch.empty(C * G, width, dtype=_F32, device=w.device) \
            if need_grad else None
        splits = _row_splits(C * G, N, w.device, block)
        ws = torch.empty(splits * C * G * (width + 1), dtype=_F32,
                         device=w.device) if splits > 1 else None
        _capi.call('zshmc_linear_categorical_log_lik', wp.data_ptr(),
                   Xp.data_ptr(), labels_f.data_ptr(), C * G, N, width, K, G,
                   ll.data_ptr(), _capi.ptr(gw), splits, _capi.ptr(ws),
                   _capi.current_stream())
        ctx.w_shape, ctx.stride = tuple(w.shape), G
        if need_grad:
            ctx.save_for_backward(gw)
        # (lane k of a chain's group holds the terms of the rows labelled k)
        return ll.reshape(C, G).sum(-1).reshape(w.shape[:-2])

    @staticmethod
    def backward(ctx, gout):
        (gw,) = ctx.saved_tensors
        K, F = ctx.w_shape[-2:]
        g = gw.reshape(-1, ctx.stride, gw.shape[-1])[:, :K, :F] * \
            gout.reshape(-1, 1, 1)
        return g.reshape(ctx.w_shape), None, None


_phi_cache = {}


def _padded_phi_t(phi, width):
    """phi [K, V] -> contiguous zero-padded phi^T [V, width], cached for as
    long as the SAME tensor (same storage, same version counter) is passed
    again.  The cache entry keeps `phi` alive, so its address cannot be handed
    to a different tensor while the entry exists; a model builder that
    recomputes phi = softmax(beta) on every joint evaluation simply misses
    (one K x V transpose, noise next to the likelihood kernel)."""
    key = (phi.data_ptr(), tuple(phi.shape), tuple(phi.stride()),
           phi._version, width, _writes.generation(phi))
    hit = _phi_cache.get('phi')
    if hit is not None and hit[0] == key:
        return hit[1]
    pt = _pad_features(phi.detach().to(_F32).t().contiguous(), width)
    _phi_cache['phi'] = (key, pt, phi)
    return pt


_counts_cache = {}


def _padded_counts(x, multiple=4):
    """counts [R0, V] -> contiguous float32 [R0, V rounded up to `multiple`]
    with a zero pad, cached while the SAME tensor (storage, version) is passed
    again: the likelihood kernel gathers one row per chain and wants 16-B
    groups (multiple = 4); the packed-rows form of the bf16x3 multinomial
    kernel reads whole 32-row tiles of a chain's counts (multiple = 32)."""
    v = x.shape[-1]
    vp = (v + multiple - 1) // multiple * multiple
    if vp == v and x.is_contiguous() and x.dtype == _F32:
        return x.reshape(-1, v), v
    key = (x.data_ptr(), tuple(x.shape), tuple(x.stride()), x._version,
           _writes.generation(x), multiple)
    hit = _counts_cache.get('x')
    if hit is not None and hit[0] == key:
        return hit[1], vp
    xp = torch.zeros(x.numel() // v, vp, dtype=_F32, device=x.device)
    xp[:, :v] = x.detach().reshape(-1, v)
    _counts_cache['x'] = (key, xp, x)
    return xp, vp


_counts_csr_cache = _Lru(2)


def counts_csr(x):
    """The documents' OWN vocabularies for
    zshmc_linear_multinomial_log_lik_bf16x3_sparse: counts [R0, V] ->
    (compacted counts float32 [n], the words' rows int32 [n], offsets int64
    [R0 + 1], n) with every document's slice padded to whole 32-row tiles (at
    least one) by count 0 / row 0.  Built once per tensor version (one host
    read: the total)."""
    key = _tensor_key(x)
    hit = _counts_csr_cache.get(key)
    if hit is not None:
        return hit
    v = x.shape[-1]
    xf = x.detach().reshape(-1, v).to(_F32)
    nz = xf != 0
    cnt = nz.sum(1)
    pad = torch.clamp((cnt + 31) // 32 * 32, min=32)
    off = torch.zeros(xf.shape[0] + 1, dtype=torch.int64, device=x.device)
    off[1:] = torch.cumsum(pad, 0)
    total = int(off[-1].item())
    vals = torch.zeros(total, dtype=_F32, device=x.device)
    rows = torch.zeros(total, dtype=torch.int32, device=x.device)
    d_idx, v_idx = nz.nonzero(as_tuple=True)          # row-major order
    start = torch.cumsum(cnt, 0) - cnt
    pos = off[:-1][d_idx] + (torch.arange(d_idx.numel(), device=x.device) -
                             start[d_idx])
    vals[pos] = xf[d_idx, v_idx]
    rows[pos] = v_idx.to(torch.int32)
    out = (vals, rows, off, total)
    _counts_csr_cache.put(key, out, x)
    return out


class MixtureMultinomialLogLik(_Function):
    """ll[r] = sum_v x[r % R0, v] log((theta . phi)[r, v]) and d/dtheta in one
    pass over phi (the [rows, V] product is never materialised): the fused
    fp32-MFMA kernel of csrc/linear_bernoulli.hip in its multinomial mode.
    theta [..., K]; phi [K, V] (no gradient through this op); x [R0, V] with
    prod(theta.shape[:-1]) a multiple of R0."""

    @staticmethod
    def forward(ctx, theta, phi, x):
        require_device(theta, phi, x)
        k = theta.shape[-1]
        width, block = likelihood_plan(k)
        t2 = _pad_features(theta.detach().reshape(-1, k).to(_F32), width)
        pt = _padded_phi_t(phi, width)
        xf, x_stride = _padded_counts(x)
        rows, vocab = t2.shape[0], pt.shape[0]
        ll = torch.empty(rows, dtype=_F32, device=theta.device)
        need_grad = ctx.needs_input_grad[0]
        gt = torch.empty_like(t2) if need_grad else None
        # fewer 64-row chain blocks than CUs: split the vocabulary range
        splits = _row_splits(rows, vocab, theta.device, block)
        ws = torch.empty(splits * rows * (width + 1), dtype=_F32,
                         device=theta.device) if splits > 1 else None
        _capi.call('zshmc_linear_multinomial_log_lik', t2.data_ptr(),
                   pt.data_ptr(), xf.data_ptr(), xf.shape[0], x_stride, rows,
                   vocab, width, ll.data_ptr(), _capi.ptr(gt), splits,
                   _capi.ptr(ws),
                   _capi.current_stream())
        ctx.t_shape = tuple(theta.shape)
        if need_grad:
            ctx.save_for_backward(gt)
        return ll.reshape(theta.shape[:-1])

    @staticmethod
    def backward(ctx, gout):
        (gt,) = ctx.saved_tensors
        k = ctx.t_shape[-1]
        g = gt[:, :k] * gout.reshape(-1, 1)
        return g.reshape(ctx.t_shape), None, None
