"""CPU: the oracle reproduces the committed golden fixtures (regression pin), and
the fixtures' inputs regenerate deterministically."""
import json
import os

import numpy as np
import pytest

from datagen import sha, splitmix64, values
from golden.make_golden import N_BLOCKS, case_inputs
from oracle_lib import tbits

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))


def test_splitmix64_known_values():
    # first outputs of the canonical splitmix64 with state 0 (public reference values)
    z = splitmix64(3, 0)
    assert [int(x) for x in z] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


@pytest.mark.parametrize("ty", ["u8", "u16", "u32", "u64"])
def test_oracle_matches_golden(oracle, ty):
    T = tbits(ty)
    assert GOLDEN["n_blocks"] == N_BLOCKS
    for w in range(T + 1):
        i = case_inputs(ty, w)
        g = GOLDEN["cases"][f"{ty}/{w}"]
        assert sha(oracle.batch("pack", ty, w, i["values"])) == g["pack"]
        assert sha(oracle.batch("unpack", ty, w, i["packed"], n_blocks=N_BLOCKS)) == g["unpack"]
        assert sha(oracle.batch("for_pack", ty, w, i["values"], aux=i["refs"])) == g["for_pack"]
        assert sha(oracle.batch("unfor_pack", ty, w, i["packed"], aux=i["refs"], n_blocks=N_BLOCKS)) == g["unfor_pack"]
        assert sha(oracle.batch("undelta_pack", ty, w, i["packed"], aux=i["bases"], n_blocks=N_BLOCKS)) == g["undelta_pack"]
    i = case_inputs(ty, T)
    g = GOLDEN["cases"][f"{ty}/misc"]
    for op in ("delta", "undelta"):
        assert sha(oracle.batch(op, ty, None, i["values"], aux=i["bases"])) == g[op]
    for op in ("transpose", "untranspose"):
        assert sha(oracle.batch(op, ty, None, i["values"])) == g[op]
