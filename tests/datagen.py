"""Deterministic counter-based test data (splitmix64), identical on every host
and numpy version: value k of stream `seed` = the (k+1)-th output of splitmix64 seeded with seed*GOLDEN.
SURVEY.md section 8(d) 'Value distribution / seeds'."""
import hashlib

import numpy as np

_DT = {"u8": np.uint8, "u16": np.uint16, "u32": np.uint32, "u64": np.uint64}
_BITS = {"u8": 8, "u16": 16, "u32": 32, "u64": 64}


def splitmix64(n, seed):
    with np.errstate(over="ignore"):
        # canonical splitmix64: state_k = seed' + (k+1)*GOLDEN, seed' = seed*GOLDEN (keeps streams apart)
        z = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z += np.uint64((seed * 0x9E3779B97F4A7C15) & (2**64 - 1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def values(ty, n, seed, bits=None):
    """n elements of type ty, uniform in [0, 2^bits) (default: full element width)."""
    T = _BITS[ty]
    bits = T if bits is None else bits
    z = splitmix64(n, seed)
    if bits < 64:
        z = z & np.uint64((1 << bits) - 1)
    return z.astype(_DT[ty])


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.astype(a.dtype.newbyteorder("<")).tobytes()).hexdigest()
