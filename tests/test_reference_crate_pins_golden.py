"""CPU: pin the golden vectors with bytes the REFERENCE CRATE itself produces -- the day a Rust toolchain exists.

bindings/rust/golden_dump/ is a tiny crate that depends on spiraldb/fastlanes BY PATH (no reference file is copied into this
repository), runs the reference's own pack / unpack / for_pack / unfor_pack / undelta_pack / delta / undelta / transpose /
untranspose / unpack_single on the inputs of tests/golden/make_golden.py and prints SHA-256 digests.  This test builds and
runs it iff `cargo` is on PATH and a reference checkout is available, and compares every digest with the committed
tests/golden/golden.json.  In this image there is no cargo (probed every round): the test then only checks that the dump
program covers exactly the keys of golden.json, so the two cannot drift apart unnoticed; parity stays "unpinned" (DESIGN.md
section 7) until this test has run for real."""
import json
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "bindings", "rust", "golden_dump")
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))


def test_dump_program_covers_the_golden_keys():
    src = open(os.path.join(CRATE, "src", "main.rs")).read()
    ops = set(re.findall(r'println!\("case \{n\}/\{W\} (\w+) ', src))
    assert ops == set(GOLDEN["cases"]["u32/7"])                            # every per-width digest of golden.json
    misc = set(re.search(r'for \(name, o\) in \[([^\]]+)\]', src).group(1).replace('"', "").replace(" ", "").split(","))
    assert misc == set(GOLDEN["cases"]["u32/misc"])
    for ty, T in (("u8", 8), ("u16", 16), ("u32", 32), ("u64", 64)):
        assert f"seq!(W in 0..={T} {{ dump_width::<{ty}, W>(); }});" in src
    assert "const N_BLOCKS: usize = %d;" % GOLDEN["n_blocks"] in src
    assert "single {n}_w{W}" in src and set(GOLDEN["unpack_single"]) == {f"{ty}_w{w}" for ty, T in
                                                                          (("u8", 8), ("u16", 16), ("u32", 32), ("u64", 64)) for w in range(T + 1)}


def test_reference_crate_reproduces_golden_vectors(tmp_path):
    ref = os.environ.get("FASTLANES_REFERENCE_DIR", "/root/reference")
    if shutil.which("cargo") is None:
        pytest.skip("no Rust toolchain in this image (cargo not on PATH): parity stays unpinned by reference-executed bytes")
    if not os.path.exists(os.path.join(ref, "Cargo.toml")):
        pytest.skip(f"no reference checkout at {ref} (set FASTLANES_REFERENCE_DIR)")
    work = tmp_path / "golden_dump"
    shutil.copytree(CRATE, work)
    toml = open(os.path.join(CRATE, "Cargo.toml.in")).read().replace("@REFERENCE_DIR@", ref)
    (work / "Cargo.toml").write_text(toml)
    r = subprocess.run(["cargo", "run", "--release", "--quiet"], cwd=work, capture_output=True, text=True, timeout=3600)
    assert r.returncode == 0, r.stderr[-4000:]
    seen = 0
    for line in r.stdout.splitlines():
        kind, key, *rest = line.split()
        if kind == "case":
            op, sha = rest
            assert GOLDEN["cases"][key][op] == sha, f"reference crate disagrees with golden.json at {key} {op}"
        elif kind == "single":
            assert GOLDEN["unpack_single"][key] == rest[0], f"reference crate disagrees with golden.json at unpack_single {key}"
        else:
            raise AssertionError(line)
        seen += 1
    assert seen == sum(len(v) for v in GOLDEN["cases"].values()) + len(GOLDEN["unpack_single"])
