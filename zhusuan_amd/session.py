"""A minimal `Session.run(fetches, feed_dict)` shim so that scripts written
against the reference (sess.run([sample_op, hmc_info.acceptance_rate, ...],
feed_dict), examples/toy_examples/gaussian.py:53-58; sess.run([sample_op,
sgmcmc_info]), examples/toy_examples/mixture_sgnht.py:50) keep their shape:
sampling ops in `fetches` are executed first, then device tensors anywhere in
the (nested list / tuple / namedtuple / dict) fetch structure are returned as
NumPy arrays."""
import torch

from .hmc import _SampleOp as _HMCSampleOp
from .sgmcmc import _SampleOp as _SGSampleOp

__all__ = ['Session']

_OPS = (_HMCSampleOp, _SGSampleOp)


def _run_ops(f, feed_dict):
    if isinstance(f, _HMCSampleOp):
        # sync: surfaces InvalidArgumentError here, as sess.run does
        f.run(feed_dict=feed_dict, sync=True)
    elif isinstance(f, _SGSampleOp):
        f.run(feed_dict=feed_dict, sync=False)
    elif isinstance(f, dict):
        for v in f.values():
            _run_ops(v, feed_dict)
    elif isinstance(f, (list, tuple)):
        for v in f:
            _run_ops(v, feed_dict)


def _fetch(f):
    if isinstance(f, _OPS):
        return None
    if isinstance(f, torch.Tensor):
        return f.detach().cpu().numpy()
    if hasattr(f, 'tensor') and isinstance(f.tensor, torch.Tensor):
        return f.tensor.detach().cpu().numpy()
    if isinstance(f, dict):
        # a plain dict: subclasses such as HMCInfo.init_momentum (regenerated
        # on access) take other constructor arguments
        return {k: _fetch(f[k]) for k in f.keys()}
    if isinstance(f, tuple) and hasattr(f, '_fields'):        # namedtuple
        return type(f)(*[_fetch(v) for v in f])
    if isinstance(f, (list, tuple)):
        return type(f)(_fetch(v) for v in f)
    return f


class Session(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple)) or \
            hasattr(fetches, '_fields')
        items = [fetches] if single else list(fetches)
        _run_ops(items, feed_dict)
        out = [_fetch(f) for f in items]      # .cpu() synchronises the stream
        return out[0] if single else out
