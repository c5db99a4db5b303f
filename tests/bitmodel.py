"""Independent bit-level model of the FastLanes packed layout (numpy).

Written from the *specification* of the wire format in SURVEY.md section 8(a)
row a4 (itself read off /root/reference/src/macros.rs:34-98), not from the
shift/carry code:

    per FL-lane l, the T values v_r (r = 0..T-1, taken at index(r,l) and
    truncated to W bits) are concatenated LSB-first into a T*W-bit stream;
    bit k of that stream is bit (k % T) of word  pk[LANES*(k // T) + l].

It shares no code with oracle/ and is used to cross-check it.
"""
import numpy as np

FL_ORDER = [0, 4, 2, 6, 1, 5, 3, 7]  # lib.rs:22


def index(row, lane):
    return FL_ORDER[row // 8] * 16 + (row % 8) * 128 + lane  # macros.rs:20-24


def transpose_index(i):
    return (i % 16) * 64 + FL_ORDER[(i // 16) % 8] * 8 + i // 128  # transpose.rs:29-36


def pack_bits(values, T, W):
    """values: sequence of 1024 python ints -> list of 1024*W/T python ints (words)."""
    LANES = 1024 // T
    n_words = 1024 * W // T
    pk = [0] * n_words
    if W == 0:
        return pk
    for lane in range(LANES):
        stream = 0
        for r in range(T):
            v = int(values[index(r, lane)])
            v &= (1 << W) - 1 if W < T else (1 << T) - 1
            stream |= v << (r * W)
        for w in range(W):
            pk[LANES * w + lane] = (stream >> (w * T)) & ((1 << T) - 1)
    return pk


def unpack_bits(pk, T, W):
    LANES = 1024 // T
    out = [0] * 1024
    if W == 0:
        return out
    for lane in range(LANES):
        stream = 0
        for w in range(W):
            stream |= int(pk[LANES * w + lane]) << (w * T)
        for r in range(T):
            out[index(r, lane)] = (stream >> (r * W)) & ((1 << W) - 1)
    return out


def np_dtype(T):
    return {8: np.uint8, 16: np.uint16, 32: np.uint32, 64: np.uint64}[T]
