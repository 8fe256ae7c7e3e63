"""Small host-side helpers mirroring zhusuan/utils.py."""
import threading

import torch

__all__ = ['merge_dicts', 'set_random_seed', 'get_random_seed',
           'broadcast_shapes']


def merge_dicts(*dict_args):
    """Shallow-merge dicts, later ones win (reference zhusuan/utils.py:220-228;
    HMC relies on observed overriding latent, hmc.py:427)."""
    result = {}
    for dictionary in dict_args:
        result.update(dictionary)
    return result


def broadcast_shapes(*shapes):
    """NumPy-rule broadcast of static shapes -> torch.Size; RuntimeError when they do
    not match (as torch.broadcast_shapes, which costs ~25 us of Python per
    call -- this sits on the per-transition host path of every model
    evaluation, tf.broadcast_static_shape in distributions/base.py:271-288)."""
    n = 0
    for s in shapes:
        if len(s) > n:
            n = len(s)
    out = [1] * n
    for s in shapes:
        off = n - len(s)
        for i, d in enumerate(s):
            d = int(d)
            if d != 1:
                j = off + i
                if out[j] == 1:
                    out[j] = d
                elif out[j] != d:
                    raise RuntimeError(
                        "Shape mismatch: objects cannot be broadcast to a "
                        "single shape: {}".format(
                            ' vs. '.join(str(tuple(x)) for x in shapes)))
    return torch.Size(out)


class _SeedState(threading.local):
    def __init__(self):
        self.seed = 0
        self.op_counter = 0      # bumps per stand-alone sampling op
        self.sampler_counter = 0  # bumps per HMC instance


_state = _SeedState()


def set_random_seed(seed):
    """Analogue of tf.set_random_seed (used by
    examples/toy_examples/gaussian.py:24): fixes the Philox key of every
    sampler / sampling op created afterwards and resets their counters."""
    _state.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    _state.op_counter = 0
    _state.sampler_counter = 0


def get_random_seed():
    return _state.seed


def next_op_offset():
    """(seed, offset) for one stand-alone sampling op; offset is the Philox
    counter word c2 of stream STREAM_DIST."""
    off = _state.op_counter
    _state.op_counter = (off + 1) & 0xFFFFFFFF
    return _state.seed, off


def next_sampler_seed():
    """Seed for a new HMC instance: global seed xor a per-instance salt in
    the high word, so two samplers in one program draw different streams."""
    k = _state.sampler_counter
    _state.sampler_counter = k + 1
    return (_state.seed ^ ((k * 0x9E3779B97F4A7C15) & 0xFFFFFFFF00000000)) \
        & 0xFFFFFFFFFFFFFFFF
