"""Recognition of the reference's LITERAL dense spellings.

The reference writes its dense-likelihood models with ordinary graph ops:

    y ~ Bernoulli(logits = tf.matmul(w, X, transpose_b=True))
                                        (univariate.py:398-403 + hmc.py:430-432)
    x ~ UnnormalizedMultinomial(tf.log(tf.matmul(tf.nn.softmax(eta), phi)))
                                        (examples/topic_models/lntm_mcem.py:39-46)

and the usual extensions of the first one -- a bias, several weight blocks --

    logits = tf.matmul(w, X, transpose_b=True) + b
    logits = tf.matmul(w1, X1, transpose_b=True) + tf.matmul(w2, X2, ...)

TensorFlow builds a graph first, so nothing is materialised before the
executor runs.  torch is eager: `w @ X.T` at BASELINE configs[2] is a
[32 768, 10^6] tensor (131 GB) before any distribution sees it.  So the
latents handed to a model function travel as `Sym` tensors -- wrapper
subclasses WITHOUT storage whose `__torch_function__` keeps the few ops of
those two spellings symbolic

    latent -> softmax(., -1) -> reshape(leading axes) -> matmul(., constant)
           -> reshape(leading axes) -> log
    latent @ constant + latent @ constant + latent[..., None] ...

and executes everything else on the real tensors (the wrapper is forced --
replaced by the value of its expression -- the moment an op outside the
table touches it, so an arbitrary model function computes exactly what it
would on plain tensors).  `Bernoulli` / `UnnormalizedMultinomial` lower a
symbolic `logits` argument to the lazy operands of the fused fp32-MFMA
likelihood (`LinearLogits`, `LogMixture`): the literal spelling then runs on
the same native plans as `zs.linear_logits` / `zs.log_mixture`.
"""
import torch
from torch.utils._pytree import tree_map

__all__ = ['Sym', 'SymbolicCut', 'wrap_latent', 'force',
           'lower_bernoulli_logits', 'lower_multinomial_logits',
           'lower_categorical_logits']


class SymbolicCut(RuntimeError):
    """A symbol over a latent that requires grad was evaluated with autograd
    disabled -- in practice inside the `forward` of a custom
    torch.autograd.Function, which receives its arguments without
    `__torch_function__` dispatch: the op's output would not be connected to
    the latent and its gradient silently lost.  The sampler catches this and
    evaluates the model on plain tensors from then on."""

_T = torch.Tensor


def _getter(name):
    return getattr(_T, name).__get__


# ops that only look at the wrapper's metadata: answered by the wrapper
_METADATA = {
    _getter('shape'), _getter('dtype'), _getter('device'), _getter('ndim'),
    _getter('is_cuda'), _getter('is_meta'), _getter('layout'),
    _getter('grad_fn'), _getter('is_leaf'),
    _getter('is_sparse'), _getter('is_quantized'), _getter('names'),
    _T.dim, _T.size, _T.numel, _T.nelement, _T.ndimension,
    _T.is_floating_point, _T.is_complex, _T.element_size, _T.__len__,
    _T.get_device, _T.__repr__, _T.__hash__, _T.__format__,
}

_REQUIRES_GRAD = _getter('requires_grad')

_SOFTMAX = {torch.softmax, torch.nn.functional.softmax, _T.softmax}
_MATMUL = {torch.matmul, _T.matmul, _T.__matmul__, torch.mm, _T.mm}
_RESHAPE = {torch.reshape, _T.reshape, _T.view}
_LOG = {torch.log, _T.log}
_ADD = {torch.add, _T.add, _T.__add__, _T.__radd__}
_UNSQUEEZE = {torch.unsqueeze, _T.unsqueeze}
_SIGMOID = {torch.sigmoid, _T.sigmoid, torch.nn.functional.sigmoid}
_MUL = {torch.mul, _T.mul, _T.__mul__, _T.__rmul__, torch.multiply,
        _T.multiply}
_SUM = {torch.sum, _T.sum}
_INDEX_SELECT = {torch.index_select, _T.index_select}
_TRANSPOSE = {torch.transpose, _T.transpose, torch.swapaxes, _T.swapaxes,
              torch.swapdims, _T.swapdims}
_MT = _getter('mT')


class Sym(torch.Tensor):
    """A tensor-shaped symbol: `expr` is one of
         ('latent', tensor)
         ('softmax', Sym)                 over the last axis
         ('reshape', Sym)                 leading axes only (last axis kept)
         ('matmul', Sym, tensor[K, N])
         ('log', Sym)
         ('col', Sym)                     latent[..., None]: a per-chain scalar
         ('add', Sym, Sym)                sum of linear terms (matmul of a
                                          latent, a bias latent [..., 1])
         ('transpose', Sym)               last two axes of a latent [..., K, F]
         ('classlogits', Sym, tensor[N, F])   X @ latent^T: [..., N, K], the
                                          class logits of a softmax regression
    """

    @staticmethod
    def __new__(cls, expr, shape, dtype, device):
        r = _T._make_wrapper_subclass(cls, tuple(shape), dtype=dtype,
                                      device=device, requires_grad=False)
        r._expr = expr
        r._value = None
        r._meta = (tuple(int(d) for d in shape), dtype, torch.device(device))
        if expr[0] == 'latent':
            r._roots = (expr[1],)
        elif expr[0] == 'add':
            r._roots = expr[1]._roots + expr[2]._roots
        else:
            r._roots = expr[1]._roots
        return r

    def __init__(self, *a, **k):
        pass

    # -- the value of the expression (computed once) -------------------------
    def force(self):
        if self._value is None:
            if any(t.requires_grad for t in self._roots) and \
                    not torch.is_grad_enabled():
                raise SymbolicCut(
                    "symbolic latent expression %r evaluated with autograd "
                    "disabled (inside a custom autograd.Function.forward?)"
                    % (self,))
            e = self._expr
            kind = e[0]
            if kind == 'latent':
                v = e[1]
            elif kind == 'softmax':
                v = torch.softmax(e[1].force(), -1)
            elif kind == 'reshape':
                v = e[1].force().reshape(tuple(_shape_of(self)))
            elif kind == 'matmul':
                v = e[1].force() @ e[2]
            elif kind == 'col':
                v = e[1].force().unsqueeze(-1)
            elif kind == 'add':
                v = e[1].force() + e[2].force()
            elif kind == 'gather1':
                v = e[1].force().index_select(len(_shape_of(self)) - 2, e[2])
            elif kind == 'gmul':
                v = e[1].force() * e[2]
            elif kind == 'gdot':
                v = _force_gdot(e)
            elif kind == 'sigmoid':
                v = torch.sigmoid(e[1].force())
            elif kind == 'transpose':
                v = e[1].force().transpose(-1, -2)
            elif kind == 'classlogits':
                # (forced, the symbol computes what the user wrote -- bit for
                # bit: e[3] is the `latent @ const` the transpose was taken of)
                v = e[3].force().transpose(-1, -2) if e[3] is not None \
                    else torch.matmul(e[2], e[1].force().transpose(-1, -2))
            else:
                v = torch.log(e[1].force())
            self._value = v
        return self._value

    def __repr__(self):
        return 'Sym(%s, shape=%s)' % (_describe(self._expr), self._meta[0])

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _METADATA:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        if func is _REQUIRES_GRAD:      # of the latent the symbol stands on
            return any(t.requires_grad for t in args[0]._roots)
        if func == _MT:
            func, args = torch.transpose, (args[0], -2, -1)
        out = _symbolic_rule(func, args, kwargs)
        if out is not None:
            return out
        # anything else: on the real tensors
        args, kwargs = tree_map(_forced, (args, kwargs))
        return func(*args, **kwargs)

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # reached only by callers that skip __torch_function__ (C++ entry
        # points): same rule, the op runs on the values
        args, kwargs = tree_map(_forced, (args, kwargs or {}))
        return func(*args, **kwargs)


def _shape_of(t):
    return t._meta[0]


def _describe(e):
    if e[0] == 'latent':
        return 'latent'
    if e[0] == 'matmul':
        return 'matmul(%s, const%s)' % (_describe(e[1]._expr),
                                        list(e[2].shape))
    if e[0] == 'add':
        return 'add(%s, %s)' % (_describe(e[1]._expr), _describe(e[2]._expr))
    if e[0] == 'classlogits':
        return 'matmul(const%s, transpose(%s))' % (list(e[2].shape),
                                                   _describe(e[1]._expr))
    return '%s(%s)' % (e[0], _describe(e[1]._expr))


def _forced(x):
    return x.force() if isinstance(x, Sym) else x


def force(x):
    """The plain tensor a (possibly symbolic) value stands for."""
    return _forced(x)


def wrap_latent(t):
    """The latent `t` as the root symbol (idempotent)."""
    if isinstance(t, Sym) or not isinstance(t, torch.Tensor):
        return t
    return Sym(('latent', t), t.shape, t.dtype, t.device)


def _kind(s):
    return s._expr[0]


def _chain_kind(s):
    """Kind of the expression with leading-axes reshapes looked through."""
    while _kind(s) == 'reshape':
        s = s._expr[1]
    return _kind(s)


def _norm_dim(dim, rank):
    return dim + rank if dim < 0 else dim


def _resolve_shape(shape_args, numel):
    if len(shape_args) == 1 and isinstance(shape_args[0], (tuple, list,
                                                           torch.Size)):
        shape_args = tuple(shape_args[0])
    try:
        shape = [int(s) for s in shape_args]
    except (TypeError, ValueError):      # e.g. Tensor.view(dtype)
        return None
    if shape.count(-1) > 1:
        return None
    if -1 in shape:
        known = 1
        for s in shape:
            if s != -1:
                known *= s
        if known == 0 or numel % known:
            return None
        shape[shape.index(-1)] = numel // known
    n = 1
    for s in shape:
        n *= s
    return tuple(shape) if n == numel else None


def _linear_term(s):
    """'w' for latent @ const, 'b' for a bias-shaped latent ([..., 1] or
    latent[..., None]), 'sum' for a sum of such terms, else None."""
    k = _kind(s)
    if k == 'matmul':
        return 'w' if _kind(s._expr[1]) == 'latent' else None
    if k == 'latent':
        return 'b' if s._meta[0] and s._meta[0][-1] == 1 else None
    if k == 'col':
        return 'b'
    return 'sum' if k == 'add' else None


def _add_rule(args, kwargs):
    if len(args) != 2 or kwargs.get('alpha', 1) != 1 or \
            any(k != 'alpha' for k in kwargs):
        return None
    a, b = args
    if not (isinstance(a, Sym) and isinstance(b, Sym)):
        return None
    ka, kb = _linear_term(a), _linear_term(b)
    if ka is None or kb is None or (ka == 'b' and kb == 'b'):
        return None
    (sa, dt, dev), (sb, dtb, devb) = a._meta, b._meta
    if dt != dtb or dev != devb or len(sa) != len(sb) or sa[:-1] != sb[:-1]:
        return None
    if sa[-1] != sb[-1] and 'b' not in (ka, kb):
        return None
    n = sb[-1] if ka == 'b' else sa[-1]
    return Sym(('add', a, b), sa[:-1] + (n,), dt, dev)


def _symbolic_rule(func, args, kwargs):
    """The new symbol if `func(*args)` continues one of the spellings, else
    None."""
    if func in _ADD:
        return _add_rule(args, kwargs)
    if func in _MATMUL and len(args) == 2 and not kwargs and \
            isinstance(args[1], Sym) and not isinstance(args[0], Sym):
        return _class_logits_rule(args[0], args[1])
    if func in _MUL and len(args) == 2 and not kwargs and \
            isinstance(args[1], Sym) and not isinstance(args[0], Sym):
        args = (args[1], args[0])       # const * latent[:, idx]: commutes
    if not args or not isinstance(args[0], Sym):
        return None
    x = args[0]
    xs, x_dtype, x_device = x._meta
    if func in _SOFTMAX:
        dim = kwargs.get('dim', args[1] if len(args) > 1 else None)
        extra = {k for k in kwargs if k not in ('dim', '_stacklevel')}
        if dim is None or extra or len(args) > 2 or not xs or \
                kwargs.get('dtype') is not None:
            return None
        if _kind(x) == 'latent' and _norm_dim(int(dim), len(xs)) == len(xs) - 1:
            return Sym(('softmax', x), xs, x_dtype, x_device)
        return None
    if func in _RESHAPE:
        if kwargs or _chain_kind(x) not in ('latent', 'softmax', 'matmul'):
            return None
        numel = 1
        for s in xs:
            numel *= s
        shape = _resolve_shape(args[1:], numel)
        if shape is None or not shape or not xs or shape[-1] != xs[-1]:
            return None
        return Sym(('reshape', x), shape, x_dtype, x_device)
    if func in _MATMUL:
        if kwargs or len(args) != 2:
            return None
        rhs = args[1]
        if isinstance(rhs, Sym) or not isinstance(rhs, torch.Tensor) or \
                rhs.dim() != 2 or rhs.requires_grad or not xs or \
                rhs.shape[0] != xs[-1] or rhs.dtype != x_dtype or \
                (rhs.device != x_device and x_device.type != 'meta'):
            return None
        if _chain_kind(x) not in ('latent', 'softmax'):
            return None
        if func in (torch.mm, _T.mm) and len(xs) != 2:
            return None
        return Sym(('matmul', x, rhs), xs[:-1] + (int(rhs.shape[1]),),
                   x_dtype, x_device)
    if func is _T.__getitem__ and _kind(x) == 'latent' and len(xs) >= 2 and \
            len(args) == 2 and not kwargs and isinstance(args[1], tuple) and \
            len(args[1]) == len(xs) - 1 and all(
                isinstance(i, slice) and i == slice(None)
                for i in args[1][:-1]) and _is_index(args[1][-1]):
        # latent[:, idx]: rows of the axis in front of the last one
        idx = args[1][-1]
        return Sym(('gather1', x, idx), xs[:-2] + (int(idx.numel()), xs[-1]),
                   x_dtype, x_device)
    if func in _INDEX_SELECT:
        dim = kwargs.get('dim', args[1] if len(args) > 1 else None)
        idx = kwargs.get('index', args[2] if len(args) > 2 else None)
        if _kind(x) == 'latent' and len(xs) >= 2 and isinstance(dim, int) and \
                _norm_dim(dim, len(xs)) == len(xs) - 2 and _is_index(idx) and \
                len(args) + len(kwargs) == 3:
            return Sym(('gather1', x, idx),
                       xs[:-2] + (int(idx.numel()), xs[-1]), x_dtype, x_device)
        return None
    if func in _MUL:
        if kwargs or len(args) != 2:
            return None
        other = args[1]
        if _kind(x) == 'gather1' and isinstance(other, _T) and \
                not isinstance(other, Sym) and not other.requires_grad and \
                tuple(other.shape) == xs and other.dtype == x_dtype:
            return Sym(('gmul', x, other), xs, x_dtype, x_device)
        return None
    if func in _SUM:
        dim = kwargs.get('dim', kwargs.get('axis',
                                           args[1] if len(args) > 1 else None))
        if _kind(x) == 'gmul' and isinstance(dim, int) and \
                _norm_dim(dim, len(xs)) == len(xs) - 1 and \
                not kwargs.get('keepdim', False) and \
                kwargs.get('dtype') is None and len(args) <= 2:
            g1 = x._expr[1]
            return Sym(('gdot', g1._expr[1], 'u', x._expr[2], g1._expr[2],
                        None, x), xs[:-1], x_dtype, x_device)
        return None
    if func in _SIGMOID:
        if len(args) == 1 and not kwargs and _kind(x) == 'gdot':
            return Sym(('sigmoid', x), xs, x_dtype, x_device)
        return None
    if func in _TRANSPOSE:
        # the last two axes of a latent [..., K, F] (-> X @ w^T below), or of
        # latent @ const[F, N] (= [..., K, N] -> the same class logits)
        dims = tuple(args[1:]) + tuple(kwargs.get(k) for k in
                                       ('dim0', 'dim1', 'axis0', 'axis1')
                                       if k in kwargs)
        if len(dims) != 2 or len(xs) < 2 or not all(
                isinstance(d, int) for d in dims) or sorted(
                _norm_dim(d, len(xs)) for d in dims) != [len(xs) - 2,
                                                         len(xs) - 1]:
            return None
        if _kind(x) == 'latent':
            return Sym(('transpose', x), xs[:-2] + (xs[-1], xs[-2]), x_dtype,
                       x_device)
        if _kind(x) == 'matmul' and _kind(x._expr[1]) == 'latent' and \
                len(x._expr[1]._meta[0]) >= 2:
            return Sym(('classlogits', x._expr[1], x._expr[2].t(), x),
                       xs[:-2] + (xs[-1], xs[-2]), x_dtype, x_device)
        return None
    if func in _UNSQUEEZE:
        dim = kwargs.get('dim', args[1] if len(args) > 1 else None)
        if _kind(x) == 'latent' and isinstance(dim, int) and \
                len(args) + len(kwargs) == 2 and dim in (-1, len(xs)):
            return Sym(('col', x), xs + (1,), x_dtype, x_device)
        return None
    if func is _T.__getitem__:
        # latent[:, None] / latent[..., None]: the same per-chain column
        idx = args[1] if len(args) == 2 else None
        idx = idx if isinstance(idx, tuple) else (idx,)
        if _kind(x) == 'latent' and not kwargs and idx and idx[-1] is None and (
                idx[:-1] == (Ellipsis,) or
                (len(idx) - 1 == len(xs) and
                 all(isinstance(i, slice) and i == slice(None)
                     for i in idx[:-1]))):
            return Sym(('col', x), xs + (1,), x_dtype, x_device)
        return None
    if func in _LOG:
        if kwargs or len(args) != 1 or _chain_kind(x) != 'matmul':
            return None
        return Sym(('log', x), xs, x_dtype, x_device)
    if func is torch.nn.functional.linear:
        # F.linear(w, X) = w @ X^T
        if len(args) == 2 and not kwargs and isinstance(args[1], _T) and \
                not isinstance(args[1], Sym) and args[1].dim() == 2:
            return _symbolic_rule(torch.matmul, (x, args[1].t()), {})
    return None


def _is_index(idx):
    return isinstance(idx, _T) and not isinstance(idx, Sym) and \
        idx.dim() == 1 and idx.dtype in (torch.int32, torch.int64)


def _force_gdot(e):
    _, lat, side, other, su, sv, via = e
    if via is not None:          # the user's own expression, bit for bit
        return via.force().sum(-1)
    from ._ops import GatheredDot
    if side == 'u':
        return GatheredDot.apply(lat.force(), su, other, sv)
    return GatheredDot.apply(other, su, lat.force(), sv)


def make_gdot(u, su, v, sv):
    """zs.gathered_dot with ONE of its tables a symbolic latent: the symbol
    [..., E] (forced, it is the GatheredDot op itself); None if neither or
    both are symbols."""
    us = isinstance(u, Sym) and _kind(u) == 'latent'
    vs = isinstance(v, Sym) and _kind(v) == 'latent'
    if us == vs:
        return None
    lat, other, side = (u, v, 'u') if us else (v, u, 'v')
    if isinstance(other, Sym) or other.requires_grad or not (
            _is_index(su) and _is_index(sv)) or su.shape != sv.shape or \
            len(lat._meta[0]) < 2 or other.dim() != len(lat._meta[0]) or \
            tuple(other.shape[:-2]) != lat._meta[0][:-2] or \
            other.shape[-1] != lat._meta[0][-1]:
        return None
    xs, dt, dev = lat._meta
    return Sym(('gdot', lat, side, other, su, sv, None),
               xs[:-2] + (int(su.numel()),), dt, dev)


def gathered_dot_mean(mean):
    """If `mean` is the symbol sigmoid(gathered_dot(latent, ...)): the parts
    (latent tensor, side, other table, select_u, select_v -- select_v None
    when `other` is already gathered pair by pair), else None."""
    if not isinstance(mean, Sym) or _kind(mean) != 'sigmoid':
        return None
    g = mean._expr[1]
    if _kind(g) != 'gdot' or g._meta[1] != torch.float32:
        return None
    _, lat, side, other, su, sv, _ = g._expr
    return dict(latent=lat._expr[1], side=side, other=other, su=su, sv=sv)


def _class_logits_rule(lhs, rhs):
    """const @ transpose(latent): X [N, F] -- or its [1, ..., N, F] /
    expanded [chains..., N, F] view, what tf.tile + tf.matmul(Xc, w,
    transpose_b=True) spells -- times latent^T [..., F, K]."""
    if _kind(rhs) != 'transpose' or not isinstance(lhs, torch.Tensor) or \
            lhs.requires_grad or lhs.dim() < 2:
        return None
    ws, dt, dev = rhs._meta                     # [..., F, K]
    if lhs.dtype != dt or lhs.shape[-1] != ws[-2] or \
            (lhs.device != dev and dev.type != 'meta'):
        return None
    lead = tuple(lhs.shape[:-2])
    if len(lead) > len(ws) - 2:
        return None
    # the leading axes must not carry data: size 1, or an expanded view
    if any(n != 1 and st != 0 for n, st in zip(lead, lhs.stride())):
        return None
    if any(n != 1 and n != m for n, m in zip(lead[::-1], ws[:-2][::-1])):
        return None
    X = lhs[(0,) * len(lead)] if lead else lhs
    return Sym(('classlogits', rhs._expr[1], X, None),
               ws[:-2] + (int(lhs.shape[-2]), ws[-1]), dt, dev)


def _strip_reshapes(s):
    while _kind(s) == 'reshape':
        s = s._expr[1]
    return s


def softmax_source(theta):
    """The latent tensor if `theta` is the symbol softmax(latent, -1)
    (leading-axes reshapes looked through), else None."""
    if not isinstance(theta, Sym):
        return None
    t = _strip_reshapes(theta)
    return t._expr[1]._expr[1] if _kind(t) == 'softmax' else None


def _linear_terms(s, out):
    k = _kind(s)
    if k == 'add':
        return _linear_terms(s._expr[1], out) and \
            _linear_terms(s._expr[2], out)
    if k == 'matmul' and _kind(s._expr[1]) == 'latent':
        out.append((s._expr[1]._expr[1], s._expr[2].t(), False))
        return True
    if k == 'latent' and s._meta[0] and s._meta[0][-1] == 1:
        out.append((s._expr[1], None, False))
        return True
    if k == 'col':
        out.append((s._expr[1]._expr[1], None, True))
        return True
    return False


def lower_bernoulli_logits(logits):
    """`latent @ const[D, N]` (+ more such terms, + a bias latent) ->
    LinearLogits over the latents; any other symbol -> its value."""
    if not isinstance(logits, Sym):
        return logits
    from .distributions.univariate import LinearLogits
    terms = []
    if logits._meta[1] == torch.float32 and _kind(logits) in ('matmul', 'add') \
            and _linear_terms(logits, terms) and \
            any(X is not None for _, X, _ in terms) and \
            len({id(w) for w, _, _ in terms}) == len(terms):
        n = [X.shape[0] for _, X, _ in terms if X is not None]
        if all(v == n[0] for v in n):
            return LinearLogits.of_terms(terms)
    return logits.force()


def lower_multinomial_logits(logits):
    """`log(softmax(latent) @ phi)` (leading-axes reshapes anywhere in
    between) -> LogMixture over the latent; any other symbol -> its value."""
    if not isinstance(logits, Sym):
        return logits
    from .distributions.multivariate import LogMixture
    s = _strip_reshapes(logits)
    if _kind(s) == 'log' and logits._meta[1] == torch.float32:
        m = _strip_reshapes(s._expr[1])
        if _kind(m) == 'matmul':
            t = _strip_reshapes(m._expr[1])
            if _kind(t) == 'softmax':
                latent = t._expr[1]._expr[1]
                return LogMixture.of_softmax(latent, m._expr[2],
                                             _shape_of(logits)[:-1])
    return logits.force()


def lower_categorical_logits(logits):
    """`X @ transpose(latent)` (or `transpose(latent @ X^T)`), latent
    [..., K, F] -> LinearClassLogits over the latent; any other symbol -> its
    value."""
    if not isinstance(logits, Sym):
        return logits
    from .distributions.univariate import LinearClassLogits
    if _kind(logits) == 'classlogits' and logits._meta[1] == torch.float32:
        return LinearClassLogits(logits._expr[1]._expr[1], logits._expr[2])
    return logits.force()
