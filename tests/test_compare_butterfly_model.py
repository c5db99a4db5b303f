"""The bit bookkeeping of unpack_compare's register path (fastlanes_amd/csrc/fl_consume.hpp: compare_block_butterfly) as a lane-level
model: 8 column threads each hold their column's verdicts of all T address-rows in 4 words, laid out so that three butterfly
steps (partner = column ^ 1, ^ 2, ^ 4; keep the groups whose owner bit equals the own column bit, take the partner's, rotate them
into the vacated slots) leave every thread with the 128 mask bits of ITS rows over all columns -- the layout the kernel stores.
The kernel itself is tested on the GPU (test_unpack_compare_u32_u64_every_width); this pins the index arithmetic it relies on, and
the lane -> column twist that lets DPP's row_half_mirror (lane ^ 7) serve as column ^ 4."""
import random

import pytest


def butterfly_row(T, m, b):          # fl_consume.hpp: butterfly_row<T>(m, b)
    return 4 * (b >> 2) + m if T == 32 else 8 * ((b >> 1) & 7) + 2 * m + (b >> 4)


def butterfly_elem(T, b):            # fl_consume.hpp: butterfly_elem<T>(b)
    return b & 3 if T == 32 else b & 1


@pytest.mark.parametrize("T", [32, 64])
def test_three_steps_transpose_the_verdicts(T):
    lanes = 1024 // T                # mask bits per address-row
    n = lanes // 8                   # elements per 16-byte cell = bits a column thread contributes per row
    log_n = n.bit_length() - 1
    rng = random.Random(T)
    verdict = [[rng.getrandbits(1) for _ in range(lanes)] for _ in range(T)]          # [address-row j][column * n + element]
    # every column thread shifts its verdicts in, first one ending up on top (v_cmp + v_addc_co per value)
    w = [[0] * 4 for _ in range(8)]
    for c in range(8):
        for m in range(4):
            bits = 0
            for b in range(31, -1, -1):
                bits = ((bits << 1) | verdict[butterfly_row(T, m, b)][c * n + butterfly_elem(T, b)]) & 0xFFFFFFFF
            w[c][m] = bits
    for i in range(3):
        g = n << i
        low = sum(1 << b for b in range(32) if not (b >> (log_n + i)) & 1)          # position bit (log_n + i) clear
        nxt = [[0] * 4 for _ in range(8)]
        for c in range(8):
            up = (c >> i) & 1
            mine = (low << g) & 0xFFFFFFFF if up else low
            rot = g if up else 32 - g
            for m in range(4):
                t = w[c ^ (1 << i)][m]
                r = ((t >> rot) | (t << (32 - rot))) & 0xFFFFFFFF                    # v_alignbit_b32(t, t, rot)
                nxt[c][m] = (w[c][m] & mine) | (r & ~mine & 0xFFFFFFFF)              # v_bfi_b32
        w = nxt
    mask = [verdict[j][col] for j in range(T) for col in range(lanes)]              # bit i of the block's mask, i = j * lanes + col
    for c in range(8):
        for m in range(4):
            want = sum(mask[(4 * c + m) * 32 + b] << b for b in range(32))
            assert w[c][m] == want, (T, c, m)


def test_lane_twist_makes_half_mirror_the_bit2_partner():
    column = lambda lane8: lane8 ^ (3 if lane8 & 4 else 0)                           # fl_consume.hpp: column_of_lane
    assert sorted(column(l) for l in range(8)) == list(range(8))
    for l in range(8):
        assert column(l ^ 7) == column(l) ^ 4                                        # DPP row_half_mirror
        assert column(l ^ 2) == column(l) ^ 2                                        # DPP quad_perm [2,3,0,1]
        assert column(l ^ 1) == column(l) ^ 1                                        # DPP quad_perm [1,0,3,2]
