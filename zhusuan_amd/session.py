"""A minimal `Session.run(fetches, feed_dict)` shim so that scripts written
against the reference (sess.run([sample_op, hmc_info.acceptance_rate, ...],
feed_dict), examples/toy_examples/gaussian.py:53-58) keep their shape: sampling
ops in `fetches` are executed first, device tensors are returned as NumPy
arrays."""
import torch

from .hmc import _SampleOp

__all__ = ['Session']


class Session(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        items = [fetches] if single else list(fetches)
        for f in items:
            if isinstance(f, _SampleOp):
                f.run(feed_dict=feed_dict, sync=True)
        out = []
        for f in items:
            if isinstance(f, _SampleOp):
                out.append(None)
            elif isinstance(f, torch.Tensor):
                out.append(f.detach().cpu().numpy())
            elif hasattr(f, 'tensor'):
                out.append(f.tensor.detach().cpu().numpy())
            else:
                out.append(f)
        return out[0] if single else out
