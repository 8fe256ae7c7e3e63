"""Pin oracle/philox.py: Random123 known-answer vectors for Philox4x32-10
and basic statistical sanity of the derived uniform / normal streams."""
import numpy as np

from oracle import philox


def _kat(c, k):
    return [int(x) for x in philox.philox4x32_10(
        *[np.uint64(v) for v in c], k[0], k[1])]


def test_random123_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert _kat([0, 0, 0, 0], [0, 0]) == [
        0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _kat([0xffffffff] * 4, [0xffffffff] * 2) == [
        0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _kat([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                [0xa4093822, 0x299f31d0]) == [
        0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_range_and_exactness():
    x = np.array([0, 255, 256, 0xffffffff], dtype=np.uint32)
    u = philox.u01(x)
    assert u.dtype == np.float32
    assert u[0] == 0.0 and u[1] == 0.0 and u[2] == np.float32(2.0 ** -24)
    assert u[3] < 1.0
    v = philox.u01_open_low(x)
    assert v[0] > 0.0 and v[3] == 1.0


def test_normal_moments_and_shard_invariance():
    z = philox.normal_chain_major(7, 3, 4096, 37)
    assert z.shape == (4096, 37) and z.dtype == np.float32
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    # a shard starting at chain 1000 sees the same numbers
    zs = philox.normal_chain_major(7, 3, 96, 37, chain_offset=1000)
    np.testing.assert_array_equal(zs, z[1000:1096])
    # different iteration / latent id / seed -> different stream
    assert not np.array_equal(philox.normal_chain_major(7, 4, 8, 37), z[:8])
    assert not np.array_equal(
        philox.normal_chain_major(7, 3, 8, 37, latent_id=1), z[:8])
    assert not np.array_equal(philox.normal_chain_major(8, 3, 8, 37), z[:8])


def test_uniform_per_chain_shard_invariance():
    u = philox.uniform_per_chain(11, 5, 2048)
    assert (u >= 0).all() and (u < 1).all()
    assert abs(u.mean() - 0.5) < 0.02
    np.testing.assert_array_equal(
        philox.uniform_per_chain(11, 5, 48, chain_offset=2000), u[2000:])
