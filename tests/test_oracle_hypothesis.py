"""Property-based tests of the CPU oracle (hypothesis): arbitrary element type, width and
data; invariants that pin the wire format independently of any single example.  CPU only."""
import numpy as np
from hypothesis import given, settings, strategies as st

import bitmodel
from oracle_lib import TYPES, lanes, load_oracle, packed_len, tbits

TY = st.sampled_from(["u8", "u16", "u32", "u64"])


@st.composite
def block_case(draw):
    ty = draw(TY)
    T = tbits(ty)
    w = draw(st.integers(0, T))
    seed = draw(st.integers(0, 2**32 - 1))
    kind = draw(st.sampled_from(["uniform", "extremes", "sparse", "ramp"]))
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        v = rng.integers(0, 2**64, size=1024, dtype=np.uint64)
    elif kind == "extremes":
        v = rng.choice(np.array([0, 1, (1 << T) - 1, (1 << max(w, 1)) - 1, 1 << (max(w, 1) - 1)], dtype=np.uint64), size=1024)
    elif kind == "sparse":
        v = np.zeros(1024, dtype=np.uint64)
        v[rng.integers(0, 1024, size=8)] = rng.integers(0, 2**64, size=8, dtype=np.uint64)
    else:
        v = np.arange(1024, dtype=np.uint64) * np.uint64(draw(st.integers(1, 2**40)))
    return ty, w, (v & np.uint64((1 << T) - 1)).astype(TYPES[ty][0]), rng


@settings(max_examples=150, deadline=None)
@given(block_case())
def test_pack_unpack_properties(case):
    o = load_oracle()
    ty, w, v, rng = case
    T = tbits(ty)
    pk = o.pack(ty, w, v)
    assert pk.size == packed_len(ty, w)
    masked = v & TYPES[ty][0]((1 << w) - 1) if w < T else v
    # unpack inverts pack up to truncation to W bits (macros.rs:73)
    assert np.array_equal(o.unpack(ty, w, pk), masked)
    # the independent bit-level model produces the same words
    assert [int(x) for x in pk] == bitmodel.pack_bits([int(x) for x in v], T, w)
    # the closed-form reader agrees on random positions (bitpacking.rs:132-179)
    for i in rng.integers(0, 1024, size=16):
        assert o.unpack_single(ty, w, pk, int(i)) == masked[i]
    # pack is idempotent on already-truncated data and linear in the lanes: zeroing one FL lane's
    # values only changes that lane's words (word-interleaved layout, macros.rs:89)
    L = lanes(ty)
    lane = int(rng.integers(0, L))
    v2 = v.copy()
    for r in range(T):
        v2[o.index(r, lane)] = 0
    pk2 = o.pack(ty, w, v2)
    diff = np.nonzero(pk != pk2)[0]
    assert all(int(d) % L == lane for d in diff)


@settings(max_examples=60, deadline=None)
@given(block_case())
def test_delta_for_transpose_properties(case):
    o = load_oracle()
    ty, w, v, rng = case
    T = tbits(ty)
    base = rng.integers(0, 2**64, size=lanes(ty), dtype=np.uint64).astype(TYPES[ty][0]) if T < 64 else \
        rng.integers(0, 2**64, size=lanes(ty), dtype=np.uint64)
    # delta / undelta are inverse for any base (delta.rs:24-45)
    assert np.array_equal(o.undelta(ty, o.delta(ty, v, base), base), v)
    # fused == unfused (delta.rs:47-63 vs :36-45)
    pk = o.pack(ty, w, v)
    assert np.array_equal(o.undelta_pack(ty, w, pk, base), o.undelta(ty, o.unpack(ty, w, pk), base))
    # FoR: unfor_pack(for_pack(v, r), r) == v whenever (v - r) fits in W bits; always equal mod 2^W otherwise
    ref = int(base[0])
    got = o.unfor_pack(ty, w, o.for_pack(ty, w, v, ref), ref)
    m = (1 << w) - 1 if w < T else (1 << T) - 1
    want = (((v.astype(object) - ref) & m) + ref) % (1 << T)
    assert [int(x) for x in got] == [int(x) for x in want]
    # transpose is a permutation with untranspose as inverse (transpose.rs:9-23)
    t = o.transpose(ty, v)
    assert np.array_equal(np.sort(t), np.sort(v))
    assert np.array_equal(o.untranspose(ty, t), v)
