"""CPU: the kernel-dispatch table is DATA generated from committed A/B sweeps, not hand-tuned code.
fastlanes_amd/csrc/fl_dispatch_table.inc must be exactly what tools/make_dispatch.py generates from the sweeps it names."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "fastlanes_amd", "csrc", "fl_dispatch_table.inc")


def _inputs():
    text = open(TABLE).read()
    uni = re.findall(r"^//\s+uniform:\s+(\S+)", text, re.M)
    chain = re.findall(r"^//\s+chain:\s+(\S+)", text, re.M)
    margin = float(re.search(r"by more than (\d+) %", text).group(1)) / 100
    return text, uni, chain, margin


def test_table_is_what_the_script_generates_from_the_committed_sweeps():
    _, uni, chain, margin = _inputs()
    assert len(uni) >= 2 and len(chain) >= 2, "the rule needs at least two boxes"
    for f in uni + chain:
        assert f.startswith("profiles/") and os.path.exists(os.path.join(ROOT, f)), f
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_dispatch.py"), "--check", "--margin", str(margin),
                        "--uniform"] + [os.path.join(ROOT, f) for f in uni] + ["--chain"] + [os.path.join(ROOT, f) for f in chain],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_table_shape():
    text, *_ = _inputs()
    rows = dict((m.group(1), [int(x) for x in m.group(3).split(",")])
                for m in re.finditer(r"constexpr unsigned char (\w+)\[(\d+)\] = \{([^}]*)\};", text))
    for op in ("UNPACK", "PACK", "UNFOR_PACK", "FOR_PACK", "UNDELTA_PACK", "UNDELTA_PACK_UNTRANSPOSE", "TRANSPOSE_DELTA_PACK"):
        for T in (8, 16, 32, 64):
            assert len(rows[f"{op}_U{T}"]) == T + 1
    for op in ("UNDELTA", "DELTA", "UNTRANSPOSE", "TRANSPOSE"):
        for T in (8, 16, 32, 64):
            assert len(rows[f"{op}_U{T}"]) == 1
    # 0 = cell-column, k = wave-per-block at k waves per SIMD; 10 + k (undelta_pack of u32 / u64 only, k >= 4) = two blocks per wavefront
    for name, r in rows.items():
        two_ok = name in ("UNDELTA_PACK_U32", "UNDELTA_PACK_U64")
        assert all(v in (0, 3, 4, 5, 6, 8) or (two_ok and v in (14, 15, 16, 18)) for v in r), name
