"""CPU: a bit-level MODEL of the u8 column-lanes kernels' arithmetic (fastlanes_amd/csrc/fl_chain.hpp, round 5) against the oracle --
the v_perm_b32 selectors of the 4 x 4 byte transpose, where a lane's runs sit in the original-order block, the split (even / odd
byte) running sum of the decode and the split bit buffer of the encode.  The GPU suite checks the kernels themselves; this pins the
ALGORITHM (every constant below is the kernel's) where no GPU is needed, so that a wrong selector cannot hide behind a box."""
import numpy as np
import pytest

from datagen import values
from oracle_lib import load_oracle

FL_ORDER = [0, 4, 2, 6, 1, 5, 3, 7]                        # lib.rs:22


@pytest.fixture(scope="module")
def oracle():
    return load_oracle()


def v_perm_b32(src0, src1, sel):
    """D.byte[i] = selector byte i picks: 0-3 -> src1 bytes, 4-7 -> src0 bytes (the only selector values the kernels use)"""
    pool = [(src1 >> (8 * k)) & 0xFF for k in range(4)] + [(src0 >> (8 * k)) & 0xFF for k in range(4)]
    return sum(pool[(sel >> (8 * i)) & 0xFF] << (8 * i) for i in range(4))


def transpose4x4_bytes(a):
    t0, t1 = v_perm_b32(a[1], a[0], 0x05010400), v_perm_b32(a[1], a[0], 0x07030602)
    u0, u1 = v_perm_b32(a[3], a[2], 0x05010400), v_perm_b32(a[3], a[2], 0x07030602)
    return [v_perm_b32(u0, t0, 0x05040100), v_perm_b32(u0, t0, 0x07060302), v_perm_b32(u1, t1, 0x05040100), v_perm_b32(u1, t1, 0x07060302)]


def test_transpose4x4_bytes_is_a_transpose_and_an_involution():
    rng = np.random.default_rng(1)
    for _ in range(50):
        a = [int(x) for x in rng.integers(0, 2**32, size=4)]
        o = transpose4x4_bytes(a)
        for i in range(4):
            for j in range(4):
                assert (o[j] >> (8 * i)) & 0xFF == (a[i] >> (8 * j)) & 0xFF
        assert transpose4x4_bytes(o) == a


def runs_of_column(block, c):
    """the 16 eight-byte runs lane (., c) reads from an original-order u8 block: FL lane 16c+e's rows at e*64 + FL_ORDER[c]*8"""
    return [block[e * 64 + FL_ORDER[c] * 8: e * 64 + FL_ORDER[c] * 8 + 8] for e in range(16)]


def runs_to_rows(runs):
    """the kernel's runs_to_rows_u8: rows[r] = 16 bytes (word k = FL lanes 4k..4k+3) from runs[e] = 8 bytes (dword h = rows 4h..4h+3)"""
    rows = [[0] * 4 for _ in range(8)]
    for k in range(4):
        for h in range(2):
            a = [int.from_bytes(bytes(runs[4 * k + i][4 * h:4 * h + 4]), "little") for i in range(4)]
            o = transpose4x4_bytes(a)
            for j in range(4):
                rows[4 * h + j][k] = o[j]
    return [b"".join(int(w).to_bytes(4, "little") for w in r) for r in rows]


def test_register_transpose_is_the_fastlanes_transpose_for_u8(oracle):
    v = values("u8", 1024, 42)
    want = oracle.transpose("u8", v)                          # transposed layout: row r = bytes [128 r, 128 r + 128) for u8
    for c in range(8):
        rows = runs_to_rows(runs_of_column(v, c))
        for r in range(8):
            assert rows[r] == bytes(want[128 * r + 16 * c:128 * r + 16 * c + 16]), (c, r)


def pk16(x, y, op):
    """v_pk_add_u16 / v_pk_sub_u16 on two packed 16-bit halves"""
    lo = op(x & 0xFFFF, y & 0xFFFF) & 0xFFFF
    hi = op(x >> 16, y >> 16) & 0xFFFF
    return lo | (hi << 16)


@pytest.mark.parametrize("w", range(9))
def test_split_running_sum_decodes_like_undelta_pack(oracle, w):
    """ColumnSum<u8>::step over the 8 rows of every cell column == Delta::undelta_pack::<W> (delta.rs:47-63)"""
    pk = values("u8", 128 * w, 100 + w)
    base = values("u8", 128, 200 + w)
    want = oracle.undelta_pack("u8", w, pk, base)
    m = ((1 << w) - 1) * 0x00010001                           # WaveBlock<u8>::field_mask
    last = (w - 1) * 128 if w else 0
    img = bytes(pk) + bytes(1024)                             # the LDS image (rows past 128*w are never read for a field)
    for c in range(8):
        bw = [int.from_bytes(bytes(base[16 * c + 4 * k:16 * c + 4 * k + 4]), "little") for k in range(4)]
        ev, od = list(bw), [x >> 8 for x in bw]
        for r in range(8):
            bit = r * w
            a0, sh = (bit >> 3) * 128, bit & 7
            a1 = a0 + 128 if a0 + 128 < last else last
            out = b""
            for k in range(4):
                cur = int.from_bytes(img[a0 + 16 * c + 4 * k:a0 + 16 * c + 4 * k + 4], "little")
                nxt = int.from_bytes(img[a1 + 16 * c + 4 * k:a1 + 16 * c + 4 * k + 4], "little")
                e = (v_perm_b32(nxt, cur, 0x06020400) >> sh) & m
                o = (v_perm_b32(nxt, cur, 0x07030501) >> sh) & m
                ev[k] = pk16(ev[k], e, lambda x, y: x + y)
                od[k] = pk16(od[k], o, lambda x, y: x + y)
                out += v_perm_b32(od[k], ev[k], 0x06020400).to_bytes(4, "little")
            assert out == bytes(want[128 * r + 16 * c:128 * r + 16 * c + 16]), (w, c, r)


@pytest.mark.parametrize("w", range(9))
def test_split_bit_buffer_encodes_like_pack_of_delta_of_transpose(oracle, w):
    """encode_consume_u8: register transpose, delta against the lane's own previous row, the split bit buffer that emits one packed cell
    per 8 full bits == pack::<W>(delta(transpose(v), base)) (delta.rs:88-95 composed)"""
    v = values("u8", 1024, 300 + w)
    base = values("u8", 128, 400 + w)
    want = oracle.pack("u8", w, oracle.delta("u8", oracle.transpose("u8", v), base))
    m = ((1 << w) - 1) * 0x00010001
    got = bytearray(128 * w)
    for c in range(8):
        x = runs_to_rows(runs_of_column(v, c))
        bw = [int.from_bytes(bytes(base[16 * c + 4 * k:16 * c + 4 * k + 4]), "little") for k in range(4)]
        pe, po, ae, ao = list(bw), [b >> 8 for b in bw], [0] * 4, [0] * 4
        fill, k_out = 0, 0
        for r in range(8):
            for k in range(4):
                ce = int.from_bytes(x[r][4 * k:4 * k + 4], "little")
                co = ce >> 8
                de = pk16(ce, pe[k], lambda p, q: p - q) & m
                dd = pk16(co, po[k], lambda p, q: p - q) & m
                pe[k], po[k] = ce, co
                ae[k] |= (de << fill) & 0xFFFFFFFF
                ao[k] |= (dd << fill) & 0xFFFFFFFF
            fill += w
            if fill >= 8:
                cell = b""
                for k in range(4):
                    cell += v_perm_b32(ao[k], ae[k], 0x06020400).to_bytes(4, "little")
                    ae[k] = (ae[k] >> 8) & 0x00FF00FF
                    ao[k] = (ao[k] >> 8) & 0x00FF00FF
                got[128 * k_out + 16 * c:128 * k_out + 16 * c + 16] = cell
                k_out += 1
                fill -= 8
        assert fill == 0 and k_out == w
    assert bytes(got) == bytes(want), w
