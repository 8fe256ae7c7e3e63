"""Philox4x32-7 counter-based RNG, NumPy restatement shared bit-for-bit with
the device code (zhusuan_amd/csrc/philox.h).  TEST INFRASTRUCTURE (see
oracle/__init__.py).

The reference draws momentum / MH uniforms with tf.random_normal /
tf.random_uniform (hmc.py:22, hmc.py:485), i.e. TensorFlow's Philox4x32-10
stream keyed by graph seed + op id.  TensorFlow (requirements-dev.txt:2,
"tensorflow>=1.13.0") is not vendored, so the stream itself cannot be
reproduced; we restate the published Philox4x32-R algorithm (Salmon et al.,
SC'11, Random123) with R = 7 rounds -- THIS repository's choice, not
TensorFlow's (csrc/philox.h says why; R = 10 is libzshmc_philox10.so) -- and
define our own counter mapping:

    key     = (seed_lo, seed_hi)
    counter = (c0, c1, c2, c3) =
        momentum  : (group = d // 4, global chain index, iteration, STREAM_MOMENTUM | latent_id << 8)
        MH uniform: (0,              global chain index, iteration, STREAM_MH)
        dist ops  : (group lo,       group hi,           offset,    STREAM_DIST)

so that results do not depend on how chains are sharded over GPUs.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

STREAM_MOMENTUM = 0
STREAM_MH = 1
STREAM_DIST = 2
STREAM_NOISE = 3


PHILOX_ROUNDS = 7      # csrc/philox.h: ZS_PHILOX_ROUNDS


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=PHILOX_ROUNDS):
    """Vectorised Philox4x32-R (R = 7: the fewest rounds Random123 lists as
    Crush-resistant; its default is 10).  All counter words broadcast
    together; returns four uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(
        *[np.asarray(c, dtype=np.uint64) & MASK32 for c in (c0, c1, c2, c3)])
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(rounds):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0), lo1,
                          hi0 ^ c3 ^ np.uint64(k1), lo0)
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def u01(x):
    """uint32 -> float32 uniform in [0, 1) with 24 random bits (exact)."""
    return ((x >> np.uint32(8)).astype(np.float32) *
            np.float32(1.0 / 16777216.0))


def u01_open_low(x):
    """uint32 -> float32 uniform in (0, 1] (safe for log)."""
    return (((x >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) *
            np.float32(1.0 / 16777216.0))


def bm_radius_uniform(x):
    """uint32 -> float32 in (0, 1] for the Box-Muller radius, as the device's
    v_cvt_f32_u32 + v_fma (csrc/philox.h): float32(x) (round to nearest even)
    then (y + 1) * 2^-32 with a single rounding."""
    y = np.asarray(x, dtype=np.uint32).astype(np.float32).astype(np.float64)
    return ((y + 1.0) * 2.0 ** -32).astype(np.float32)


def bm_angle_fraction(x):
    """uint32 -> the Box-Muller angle in revolutions, [0, 1): the top 23 bits
    (the device places them in the mantissa of a float in [1, 2) and lets
    v_sin / v_cos drop the integer revolution)."""
    return ((np.asarray(x, dtype=np.uint32) >> np.uint32(9)).astype(np.float64)
            * 2.0 ** -23)


def box_muller(xa, xb):
    """Two uint32 words -> two N(0,1) float32 (cos branch, sin branch).
    Evaluated in float64 and rounded once: this is the 'exact' value the
    device's v_log/v_sqrt/v_sin/v_cos path approximates to a few 1e-7."""
    u1 = bm_radius_uniform(xa).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(u1))
    ang = 2.0 * np.pi * bm_angle_fraction(xb)
    return ((r * np.cos(ang)).astype(np.float32),
            (r * np.sin(ang)).astype(np.float32))


def seed_key(seed):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return seed & 0xFFFFFFFF, seed >> 32


def _chain_ids(n_chains, chain_offset):
    """Global chain indices of a shard: a contiguous range starting at the
    scalar `chain_offset`, or an explicit array (an arbitrary subset of a
    larger run: the counter is keyed by the GLOBAL chain index, so the oracle
    can reproduce any chains of a 65 536-chain device run exactly)."""
    if np.ndim(chain_offset) == 0:
        return np.arange(n_chains, dtype=np.uint64) + np.uint64(chain_offset)
    ids = np.asarray(chain_offset, dtype=np.uint64).reshape(-1)
    assert ids.shape[0] == n_chains, (ids.shape, n_chains)
    return ids


def normal_chain_major(seed, iteration, n_chains, n_data, chain_offset=0,
                       latent_id=0, stream=STREAM_MOMENTUM):
    """N(0,1) float32 array [n_chains, n_data]; element (c, d) comes from
    Philox counter (d//4, chain_offset+c, iteration, stream|latent_id<<8),
    output words (0,1)->d%4 in (0,1), words (2,3)->d%4 in (2,3)."""
    k0, k1 = seed_key(seed)
    n_groups = (n_data + 3) // 4
    g = np.arange(n_groups, dtype=np.uint64)[None, :]
    c = _chain_ids(n_chains, chain_offset)[:, None]
    x0, x1, x2, x3 = philox4x32(g, c, np.uint64(iteration & 0xFFFFFFFF),
                                   np.uint64(stream | (latent_id << 8)),
                                   k0, k1)
    z0, z1 = box_muller(x0, x1)
    z2, z3 = box_muller(x2, x3)
    z = np.stack([z0, z1, z2, z3], axis=-1).reshape(n_chains, n_groups * 4)
    return np.ascontiguousarray(z[:, :n_data])


def uniform_per_chain(seed, iteration, n_chains, chain_offset=0,
                      stream=STREAM_MH):
    """U[0,1) float32 per chain from counter (0, chain, iteration, stream)."""
    k0, k1 = seed_key(seed)
    c = _chain_ids(n_chains, chain_offset)
    x0, _, _, _ = philox4x32(np.uint64(0), c,
                                np.uint64(iteration & 0xFFFFFFFF),
                                np.uint64(stream), k0, k1)
    return u01(x0)


def uniform_flat(seed, offset, n, stream=STREAM_DIST):
    """U[0,1) float32 vector of length n for the stand-alone distribution
    sampling ops: element i uses counter (i//4 lo, i//4 hi, offset, stream),
    word i%4."""
    k0, k1 = seed_key(seed)
    ng = (n + 3) // 4
    g = np.arange(ng, dtype=np.uint64)
    xs = philox4x32(g & MASK32, g >> np.uint64(32),
                       np.uint64(offset & 0xFFFFFFFF), np.uint64(stream),
                       k0, k1)
    return u01(np.stack(xs, axis=-1).reshape(-1)[:n])


def normal_flat(seed, offset, n, stream=STREAM_DIST):
    """N(0,1) float32 vector of length n (same counter layout as
    uniform_flat; words (0,1)->(4g,4g+1), (2,3)->(4g+2,4g+3))."""
    k0, k1 = seed_key(seed)
    ng = (n + 3) // 4
    g = np.arange(ng, dtype=np.uint64)
    x0, x1, x2, x3 = philox4x32(g & MASK32, g >> np.uint64(32),
                                   np.uint64(offset & 0xFFFFFFFF),
                                   np.uint64(stream), k0, k1)
    z0, z1 = box_muller(x0, x1)
    z2, z3 = box_muller(x2, x3)
    return np.stack([z0, z1, z2, z3], axis=-1).reshape(-1)[:n]
