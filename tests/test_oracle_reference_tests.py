"""Pins the CPU oracle (oracle/) against every assertion the reference's own
unit tests make and against the known-answer vectors of
SURVEY.md section 8(c).  CPU only."""
import hashlib

import numpy as np
import pytest

import bitmodel
from oracle_lib import TYPES, lanes, packed_len, tbits

ALL_TW = [(ty, w) for ty in ("u8", "u16", "u32", "u64") for w in range(tbits(ty) + 1)]


def test_fl_order_is_own_inverse(oracle):
    # lib.rs:53-59
    import ctypes
    order = (ctypes.c_uint * 8).in_dll(oracle.lib, "fl_oracle_FL_ORDER")
    assert list(order) == [0, 4, 2, 6, 1, 5, 3, 7]
    for i in range(8):
        assert order[order[i]] == i


@pytest.mark.parametrize("ty", ["u8", "u16", "u32", "u64"])
def test_index_is_bijection(oracle, ty):
    # macros.rs:20-24 maps [0,T) x [0,LANES) onto [0,1024)
    seen = sorted(oracle.index(r, l) for r in range(tbits(ty)) for l in range(lanes(ty)))
    assert seen == list(range(1024))
    for r in range(tbits(ty)):
        for l in (0, lanes(ty) - 1):
            assert oracle.index(r, l) == bitmodel.index(r, l)


def test_pack_u16_into_u3_readme(oracle):
    # lib.rs:71-96 and README.md:14-47 (unchecked_ forms are the same code here)
    W = 3
    values = np.array([i % (1 << W) for i in range(1024)], dtype=np.uint16)
    packed = oracle.pack("u16", W, values)
    assert packed.size == 128 * W // 2
    assert np.array_equal(oracle.unpack("u16", W, packed), values)
    for i in range(1024):
        assert oracle.unpack_single("u16", W, packed, i) == values[i]


def test_macros_test_pack_u16_w15(oracle):
    # macros.rs:180-207
    values = np.array([i % (1 << 15) for i in range(1024)], dtype=np.uint16)
    packed = oracle.pack("u16", 15, values)
    assert packed.size == 960
    assert np.array_equal(oracle.unpack("u16", 15, packed), values)


def test_unchecked_pack_u32_w10(oracle):
    # bitpacking.rs:248-256
    values = np.arange(1024, dtype=np.uint32)
    packed = oracle.pack("u32", 10, values)
    assert packed.size == 320
    assert np.array_equal(oracle.unpack("u32", 10, packed), values)


def test_unpack_single_u32_w16(oracle):
    # bitpacking.rs:258-271
    values = np.arange(1024, dtype=np.uint32)
    packed = oracle.pack("u32", 16, values)
    assert packed.size == 512
    for i in range(1024):
        assert oracle.unpack_single("u32", 16, packed, i) == values[i]


@pytest.mark.parametrize("ty,w", ALL_TW)
def test_round_trip_reference_values(oracle, ty, w):
    # bitpacking.rs:273-315: values[i] = i % (1 << (W % T)), 124 generated cases
    T = tbits(ty)
    values = np.array([i % (1 << (w % T)) for i in range(1024)], dtype=TYPES[ty][0])
    packed = oracle.pack(ty, w, values)
    assert packed.size == 1024 * w // T
    assert np.array_equal(oracle.unpack(ty, w, packed), values)
    for i in range(1024):
        assert oracle.unpack_single(ty, w, packed, i) == values[i]


def test_delta_u16_w15(oracle):
    # delta.rs:80-107
    W = 15
    values = np.array([i // 8 for i in range(1024)], dtype=np.uint16)
    transposed = oracle.transpose("u16", values)
    base = np.zeros(64, dtype=np.uint16)
    deltas = oracle.delta("u16", transposed, base)
    packed = oracle.pack("u16", W, deltas)
    assert packed.size == 128 * W // 2
    fused = oracle.undelta_pack("u16", W, packed, base)
    assert np.array_equal(fused, transposed)
    unfused = oracle.undelta("u16", oracle.unpack("u16", W, packed), base)
    assert np.array_equal(unfused, transposed)


def test_ffor_u16_w15(oracle):
    # ffor.rs:66-88
    W = 15
    values = np.array([i % (1 << W) for i in range(1024)], dtype=np.uint16)
    packed = oracle.for_pack("u16", W, values, 10)
    unpacked = oracle.unpack("u16", W, packed)
    expect = (values - np.uint16(10)) & np.uint16((1 << W) - 1)
    assert np.array_equal(unpacked, expect)


# ---------------------------------------------------------------------------
# Known-answer vectors, SURVEY.md section 8(c) (derived there by an independent
# transliteration of the reference; this oracle must reproduce them).
# ---------------------------------------------------------------------------
def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<")).tobytes()).hexdigest()


def test_kat1_bench_input(oracle):
    # benches/bitpacking.rs:15-16: [3u16; 1024], W=3
    pk = oracle.pack("u16", 3, np.full(1024, 3, dtype=np.uint16))
    assert (pk[0:64] == 0xB6DB).all() and (pk[64:128] == 0xDB6D).all() and (pk[128:192] == 0x6DB6).all()


def test_kat2_readme(oracle):
    v = np.array([i % 8 for i in range(1024)], dtype=np.uint16)
    pk = oracle.pack("u16", 3, v)
    assert list(pk[0:4]) == [0x0000, 0x9249, 0x2492, 0xB6DB]
    assert list(pk[64:66]) == [0x0000, 0x4924]
    assert list(pk[128:130]) == [0x0000, 0x2492]
    assert _sha(pk) == "f949547d2b920f409dc21441e8ce7d412965a9ff3eac94d551362f689372db20"


def test_kat3_u32_w10(oracle):
    pk = oracle.pack("u32", 10, np.arange(1024, dtype=np.uint32))
    assert list(pk[0:4]) == [0x10020000, 0x50120401, 0x90220802, 0xD0320C03]
    assert pk[32] == 0x0A020060 and pk[319] == 0xFFF7FBFE
    assert _sha(pk) == "fded69a758643dbc59d8e5afc1cd28f96576f71c04aafbe0507dd7837e5a6d1c"


def test_kat4_u32_w7(oracle):
    pk = oracle.pack("u32", 7, np.array([i & 127 for i in range(1024)], dtype=np.uint32))
    assert pk.size == 224
    assert list(pk[0:4]) == [0x0, 0x10204081, 0x20408102, 0x3060C183]
    assert _sha(pk) == "16f02eec2ce2d18d6ac9cb51e5768981332865bf8cf2ba9fbf15712b48e59bb5"


def test_kat5_u64_w17(oracle):
    v = np.array([(i * 2654435761) & 0x1FFFF for i in range(1024)], dtype=np.uint64)
    pk = oracle.pack("u64", 17, v)
    assert pk.size == 272
    assert list(pk[0:2]) == [0x4C06C401B1000000, 0x198CAAC4A46379B1]
    assert _sha(pk) == "6f2ff76d32f1ac12771043c4d884b8a7972fa1a6e5f15b18e9cd3ce3f8b510f5"


def test_kat6_u8_w8_copy(oracle):
    v = np.array([i & 255 for i in range(1024)], dtype=np.uint8)
    pk = oracle.pack("u8", 8, v)
    assert np.array_equal(pk[:128], v[:128]) and pk[128] == 128
    assert np.array_equal(pk, v)  # index(r,l) = 128 r + l for u8


def test_kat7_delta_bench(oracle):
    # benches/delta.rs:15-27
    v = np.array([i // 8 for i in range(1024)], dtype=np.uint16)
    t = oracle.transpose("u16", v)
    assert list(t[0:8]) == [0, 8, 16, 24, 32, 40, 48, 56]
    base = np.zeros(64, dtype=np.uint16)
    d = oracle.delta("u16", t, base)
    assert int(d.max()) == 126
    pk = oracle.pack("u16", 9, d)
    assert _sha(pk) == "7123aa8cd64fba3555abb4cf3180f8b273745ba6cf244314f7901bfcdf9db2a4"
    assert np.array_equal(oracle.undelta_pack("u16", 9, pk, base), t)
