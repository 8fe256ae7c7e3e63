"""Who wrote a latent?  The samplers update their latents in place THROUGH THE
C-ABI (raw pointers), which torch's version counters do not see.  What a
sampler keeps ABOUT its latents between two runs -- the likelihood evaluation
at the current state, the column sums of the mass estimator -- must be
dropped when ANOTHER sampler (a second HMC on the same tensor, an SGMCMC step)
moved them in between.  Every sampler therefore notes its writes here: one
generation counter per storage, process-wide, compared together with torch's
own counters (zhusuan_amd.plans.base._versions).  A write the library cannot see at
all (`x.data`, DLPack, a raw pointer) has `HMC.latents_changed()` /
`HMC.observed_changed()`."""

import itertools

_generation = {}
# ONE counter for every storage: an entry evicted and made again can never
# come back to a value some sampler recorded earlier
_counter = itertools.count(1)


def _key(t):
    # views share their storage; a freed and re-used address starts at a
    # non-zero generation, which is as good as zero: only equality matters,
    # between two moments at which the same sampler holds the same tensor
    return (str(t.device), t.untyped_storage().data_ptr())


def note(tensors):
    """The caller has just (enqueued a kernel that has) written `tensors`."""
    for t in tensors:
        k = _key(t)
        _generation[k] = next(_counter)
    if len(_generation) > 4096:           # storages long gone
        for k in list(_generation)[:2048]:
            del _generation[k]


def generation(t):
    return _generation.get(_key(t), 0)
