"""Covers the weak spots of the reference's own tests:
full-width random data, straddle carries, W==T on non-zero data, over-wide
inputs, non-zero Delta bases, untranspose, unfor_pack -- and cross-checks the
oracle against the independent bit-level model and the closed-form reader.
CPU only."""
import numpy as np
import pytest

import bitmodel
from oracle_lib import TYPES, lanes, packed_len, tbits

ALL_TW = [(ty, w) for ty in ("u8", "u16", "u32", "u64") for w in range(tbits(ty) + 1)]


def rand_vals(rng, ty, w, n=1024):
    """uniform in [0, 2^w) (w may exceed T for over-wide tests)"""
    T = tbits(ty)
    raw = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    if w < 64:
        raw &= np.uint64((1 << w) - 1)
    return raw.astype(TYPES[ty][0]) if T < 64 else raw


@pytest.mark.parametrize("ty,w", ALL_TW)
def test_random_roundtrip_and_closed_form(oracle, ty, w):
    rng = np.random.default_rng(1000 + 97 * tbits(ty) + w)
    v = rand_vals(rng, ty, w)
    pk = oracle.pack(ty, w, v)
    assert np.array_equal(oracle.unpack(ty, w, pk), v)
    # the closed-form reader (bitpacking.rs:132-179) is an independent spec
    for i in range(0, 1024, 7):
        assert oracle.unpack_single(ty, w, pk, i) == v[i]
    for i in (0, 1, 127, 128, 1022, 1023):
        assert oracle.unpack_single(ty, w, pk, i) == v[i]


@pytest.mark.parametrize("ty,w", ALL_TW)
def test_matches_bit_level_model(oracle, ty, w):
    T = tbits(ty)
    rng = np.random.default_rng(5000 + 131 * T + w)
    v = rand_vals(rng, ty, T)  # over-wide on purpose: pack must truncate
    pk = oracle.pack(ty, w, v)
    model = bitmodel.pack_bits([int(x) for x in v], T, w)
    assert [int(x) for x in pk] == model
    # and unpack of arbitrary random packed bytes
    rpk = rand_vals(rng, ty, T, n=packed_len(ty, w))
    got = oracle.unpack(ty, w, rpk)
    assert [int(x) for x in got] == bitmodel.unpack_bits([int(x) for x in rpk], T, w)


@pytest.mark.parametrize("ty", ["u8", "u16", "u32", "u64"])
def test_overwide_inputs_are_truncated(oracle, ty):
    # macros.rs:73 (`src & mask`); W==T copies without a mask (macros.rs:58)
    T = tbits(ty)
    rng = np.random.default_rng(7)
    v = rand_vals(rng, ty, T)
    for w in (1, T // 2, T - 1):
        got = oracle.unpack(ty, w, oracle.pack(ty, w, v))
        assert np.array_equal(got, v & TYPES[ty][0]((1 << w) - 1))
    assert np.array_equal(oracle.unpack(ty, T, oracle.pack(ty, T, v)), v)


@pytest.mark.parametrize("ty", ["u8", "u16", "u32", "u64"])
def test_width_zero(oracle, ty):
    # macros.rs:52-53 (pack writes nothing), :118-125 (unpack yields zeros)
    v = rand_vals(np.random.default_rng(3), ty, tbits(ty))
    assert oracle.pack(ty, 0, v).size == 0
    assert not oracle.unpack(ty, 0, np.zeros(0, dtype=TYPES[ty][0])).any()
    assert oracle.unpack_single(ty, 0, np.zeros(0, dtype=TYPES[ty][0]), 5) == 0


@pytest.mark.parametrize("ty", ["u8", "u16", "u32", "u64"])
def test_width_too_large_is_an_error(oracle, ty):
    # bitpacking.rs:93 / :126 / :197  unreachable!()
    T = tbits(ty)
    v = np.zeros(1024, dtype=TYPES[ty][0])
    with pytest.raises(ValueError):
        oracle.pack(ty, T + 1, v)
    with pytest.raises(ValueError):
        oracle.unpack(ty, T + 1, v)
    with pytest.raises(ValueError):
        oracle.unpack_single(ty, T + 1, v, 0)
    with pytest.raises(ValueError):  # bitpacking.rs:152 assert!(index < 1024)
        oracle.unpack_single(ty, 1, np.zeros(packed_len(ty, 1), dtype=TYPES[ty][0]), 1024)


@pytest.mark.parametrize("ty", ["u8", "u16", "u32", "u64"])
def test_transpose_properties(oracle, ty):
    # transpose.rs:9-36
    tau = [oracle.transpose_index(i) for i in range(1024)]
    assert sorted(tau) == list(range(1024))
    assert tau == [bitmodel.transpose_index(i) for i in range(1024)]
    v = rand_vals(np.random.default_rng(11), ty, tbits(ty))
    t = oracle.transpose(ty, v)
    assert np.array_equal(t, v[np.array(tau)])
    assert np.array_equal(oracle.untranspose(ty, t), v)
    assert np.array_equal(oracle.transpose(ty, oracle.untranspose(ty, v)), v)
    # along each FL lane's row order the transposed positions
    # are T consecutive original positions -- why Delta runs per lane.
    T = tbits(ty)
    for lane in (0, lanes(ty) - 1):
        pos = [tau[oracle.index(r, lane)] for r in range(T)]
        assert pos == list(range(pos[0], pos[0] + T))


@pytest.mark.parametrize("ty", ["u8", "u16", "u32", "u64"])
def test_delta_nonzero_bases(oracle, ty):
    # delta.rs:24-63 with random bases (never exercised by the reference)
    T = tbits(ty)
    rng = np.random.default_rng(13 + T)
    v = rand_vals(rng, ty, T)
    base = rand_vals(rng, ty, T, n=lanes(ty))
    d = oracle.delta(ty, v, base)
    assert np.array_equal(oracle.undelta(ty, d, base), v)
    # hand model: per lane running difference in row order
    for lane in (0, lanes(ty) // 2, lanes(ty) - 1):
        prev = int(base[lane])
        for r in range(T):
            i = oracle.index(r, lane)
            assert int(d[i]) == (int(v[i]) - prev) % (1 << T)
            prev = int(v[i])
    for w in sorted({0, 1, T // 2, T - 1, T}):
        dw = rand_vals(rng, ty, w)
        pk = oracle.pack(ty, w, dw)
        fused = oracle.undelta_pack(ty, w, pk, base)
        unfused = oracle.undelta(ty, oracle.unpack(ty, w, pk), base)
        assert np.array_equal(fused, unfused)


@pytest.mark.parametrize("ty", ["u8", "u16", "u32", "u64"])
def test_for_roundtrip(oracle, ty):
    # ffor.rs:24-50: unfor_pack(for_pack(v, ref), ref) == v when v-ref fits in W bits
    T = tbits(ty)
    rng = np.random.default_rng(17 + T)
    for w in sorted({0, 1, 3, T // 2, T - 1, T}):
        ref = int(rand_vals(rng, ty, T, n=1)[0])
        off = rand_vals(rng, ty, w)
        v = (off + TYPES[ty][0](ref)).astype(TYPES[ty][0])  # wrapping
        pk = oracle.for_pack(ty, w, v, ref)
        assert np.array_equal(pk, oracle.pack(ty, w, off))
        assert np.array_equal(oracle.unfor_pack(ty, w, pk, ref), v)


FAST = [("u8", "pack", 3), ("u8", "unpack", 3), ("u16", "pack", 3), ("u16", "unpack", 3),
        ("u16", "undelta_pack", 9), ("u32", "pack", 7), ("u32", "unpack", 7),
        ("u32", "unfor_pack", 7), ("u32", "unpack", 10), ("u32", "pack", 12),
        ("u32", "unpack", 12), ("u32", "undelta_pack", 12), ("u64", "pack", 17),
        ("u64", "unpack", 17)]


@pytest.mark.parametrize("ty,op,w", FAST)
@pytest.mark.parametrize("nthreads", [1, 3])
def test_fast_family_equals_literal(oracle, ty, op, w, nthreads):
    T = tbits(ty)
    n = 7
    rng = np.random.default_rng(23 + T + w)
    if op == "pack":
        data = rand_vals(rng, ty, T, n=n * 1024)
        assert np.array_equal(oracle.fast(op, ty, w, data, nthreads=nthreads),
                              oracle.batch(op, ty, w, data))
    else:
        data = rand_vals(rng, ty, T, n=n * packed_len(ty, w))
        aux = None
        if op == "unfor_pack":
            aux = rand_vals(rng, ty, T, n=n)
        elif op == "undelta_pack":
            aux = rand_vals(rng, ty, T, n=n * lanes(ty))
        assert np.array_equal(oracle.fast(op, ty, w, data, aux=aux, nthreads=nthreads),
                              oracle.batch(op, ty, w, data, aux=aux))


def test_fast_family_unspecialised_width_is_loud(oracle):
    with pytest.raises(ValueError):
        oracle.fast("unpack", "u32", 5, np.zeros(160, dtype=np.uint32))


@pytest.mark.parametrize("nthreads", [1, 3])
def test_fast_mixed_unpack_equals_literal(oracle, nthreads):
    """The mixed-width CPU baseline (all 33 u32 widths specialised, bitpacking.rs:115-128) produces exactly what the
    literal per-block oracle does; a width > 32 is refused (bitpacking.rs:126)."""
    rng = np.random.default_rng(9)
    widths = np.concatenate([np.arange(33), rng.integers(0, 33, size=40)]).astype(np.uint8)
    off = np.concatenate([[0], np.cumsum(widths.astype(np.uint64) * 128)]).astype(np.uint64)
    col = rng.integers(0, 2**32, size=int(off[-1]) // 4, dtype=np.uint64).astype(np.uint32)
    got = oracle.fast_unpack_mixed_u32(widths, off[:-1], col, nthreads=nthreads)
    want = np.concatenate([oracle.unpack("u32", int(w), col[int(off[b]) // 4:int(off[b + 1]) // 4]) for b, w in enumerate(widths)])
    assert np.array_equal(got, want)
    with pytest.raises(ValueError):
        oracle.fast_unpack_mixed_u32(np.array([33], dtype=np.uint8), np.zeros(1, dtype=np.uint64), col)
