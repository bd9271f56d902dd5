"""Philox4x32-10 counter RNG, numpy restatement (test infrastructure only).

TF's dropout RNG stream (reference: every ``Dropout`` in
AttemptFour/Model/lc_NIC.py:51-55,94 and the LSTM ``dropout=`` argument at
lc_NIC.py:122) cannot be reproduced, so the build defines its own stream and the
oracle restates it bit-exactly.  The distribution is the one Keras uses:
i.i.d. Bernoulli(keep = 1-rate) per element, kept values scaled by 1/(1-rate).

Stream definition (shared with csrc/tnt_rng.h):
    element e (flat row-major index of the logical tensor)
    counter = (lo32(e>>2), hi32(e>>2), site, step)   key = (lo32(seed), hi32(seed))
    r = philox4x32_10(counter, key)[e & 3]
    u = (r >> 8) * 2**-24           (exact in float32)
    keep = u >= float32(rate)
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays (counters); k0,k1 scalars."""
    c0 = np.asarray(c0, dtype=np.uint32)
    c1 = np.asarray(c1, dtype=np.uint32) + np.zeros_like(c0)
    c2 = np.asarray(c2, dtype=np.uint32) + np.zeros_like(c0)
    c3 = np.asarray(c3, dtype=np.uint32) + np.zeros_like(c0)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def uniform24(n, seed, site, step):
    """n uniforms in [0,1) with 24-bit resolution for flat elements 0..n-1."""
    e = np.arange(n, dtype=np.uint64)
    grp = e >> np.uint64(2)
    lane = (e & np.uint64(3)).astype(np.int64)
    r = philox4x32_10((grp & _MASK).astype(np.uint32),
                      (grp >> np.uint64(32)).astype(np.uint32),
                      np.uint32(site), np.uint32(step),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    r = np.stack(r, axis=-1)
    pick = r[np.arange(n), lane]
    return (pick >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def keep_mask(shape, rate, seed, site, step):
    """Boolean keep-mask of the given logical shape (True = kept)."""
    n = int(np.prod(shape))
    if rate <= 0.0:
        return np.ones(shape, dtype=bool)
    u = uniform24(n, int(seed), int(site), int(step))
    return (u >= np.float32(rate)).reshape(shape)
