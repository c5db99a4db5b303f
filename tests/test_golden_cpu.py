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


@pytest.mark.parametrize("ty", ["u8", "u16", "u32", "u64"])
def test_unpack_single_golden_is_unpack_of_the_same_stream(oracle, ty):
    """The unpack_single digests (closed-form reader, bitpacking.rs:132-179, every index of two blocks) equal the
    digest of unpack() of the same packed stream: two independent statements of the wire format agree on the
    committed vectors."""
    from oracle_lib import packed_len
    T = tbits(ty)
    for w in range(T + 1):
        pk = values(ty, 2 * packed_len(ty, w), 3300 + 64 * T + w)
        assert sha(oracle.batch("unpack", ty, w, pk, n_blocks=2)) == GOLDEN["unpack_single"][f"{ty}_w{w}"], (ty, w)


def test_generator_refuses_to_overwrite_committed_vectors(tmp_path, monkeypatch):
    """Golden vectors are immutable: if the oracle's output ever changed, make_golden.py must stop instead of
    silently regenerating the fixture the parity tests trust."""
    import shutil
    import subprocess
    import sys
    here = os.path.join(os.path.dirname(__file__), "golden")
    work = tmp_path / "tests" / "golden"
    work.mkdir(parents=True)
    shutil.copy(os.path.join(here, "make_golden.py"), work / "make_golden.py")
    for f in ("datagen.py", "oracle_lib.py"):
        shutil.copy(os.path.join(os.path.dirname(here), f), tmp_path / "tests" / f)
    shutil.copytree(os.path.join(os.path.dirname(os.path.dirname(here)), "oracle"), tmp_path / "oracle")
    tampered = json.loads(json.dumps(GOLDEN))
    tampered["cases"]["u32/7"]["unpack"] = "0" * 64
    (work / "golden.json").write_text(json.dumps(tampered))
    r = subprocess.run([sys.executable, str(work / "make_golden.py")], capture_output=True, text=True)
    assert r.returncode != 0 and "refusing to overwrite" in r.stderr
    assert json.loads((work / "golden.json").read_text())["cases"]["u32/7"]["unpack"] == "0" * 64
