"""zhusuan_amd.hmc._versions: what the sampler compares between two runs to
know that nobody else wrote a latent (the carried start evaluation, the mass
estimator's column sums)."""
import torch

from zhusuan_amd.hmc import _versions


def test_versions_follow_in_place_writes():
    a, b = torch.zeros(4), torch.ones(2, 3)
    v0 = _versions([a, b])
    assert v0 == _versions([a, b])
    a.add_(1.0)
    assert _versions([a, b]) != v0
    v1 = _versions([a, b])
    b.view(-1)[0] = 5.0            # through a view: same counter
    assert _versions([a, b]) != v1
    c = a + 1                       # out of place: nothing moves
    assert _versions([a, b]) == _versions([a, b]) and c is not a


def test_versions_of_inference_tensors_are_unknown():
    with torch.inference_mode():
        x = torch.zeros(3)
    assert _versions([torch.zeros(2), x]) is None


def test_versions_see_the_library_s_own_writes():
    """A sampler's writes go through the C-ABI and leave torch's counters
    alone: zhusuan_amd._writes keeps a generation per storage that every
    sampler bumps and `_versions` compares (a second HMC, or an
    SGMCMC step, on the same latent must invalidate the first one's carried
    start evaluation)."""
    from zhusuan_amd import _writes
    a, b = torch.zeros(8), torch.ones(3)
    v0 = _versions([a, b])
    _writes.note([a])
    v1 = _versions([a, b])
    assert v1 != v0 and v1[1] == v0[1]
    _writes.note([a[2:6].view(2, 2)])        # a view: the same storage
    assert _versions([a, b])[0] != v1[0]
    assert _versions([a, b]) == _versions([a, b])
