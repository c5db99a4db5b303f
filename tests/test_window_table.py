"""CPU: which kernels run under a windowed tile map is DATA generated from committed same-buffer A/B runs, not code
(fastlanes_amd/csrc/fl_window_table.inc must be exactly what tools/make_window_table.py generates from the files it names)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "fastlanes_amd", "csrc", "fl_window_table.inc")
OPS = ["UNPACK", "PACK", "UNDELTA_PACK", "UNDELTA_PACK_UNTRANSPOSE", "TRANSPOSE_DELTA_PACK", "DELTA", "UNDELTA", "TRANSPOSE", "UNTRANSPOSE",
       "UNPACK_COMPARE", "UNPACK_BLOCK_SUMS", "BLOCK_MIN_MAX"]


def test_table_is_what_the_script_generates_from_the_committed_runs():
    text = open(TABLE).read()
    files = re.findall(r"^//\s+window:\s+(\S+)", text, re.M)
    margin = float(re.search(r"lost to w=31 by (\d+) % or more", text).group(1)) / 100
    assert len(files) >= 2, "the rule needs at least two A/B runs"
    for f in files:
        assert f.startswith("profiles/") and os.path.exists(os.path.join(ROOT, f)), f
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_window_table.py"), "--check", "--margin", str(margin)] +
                       [os.path.join(ROOT, f) for f in files], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_table_shape_and_the_dispatch_header_reads_every_row():
    text = open(TABLE).read()
    rows = dict((m.group(1), [int(x) for x in m.group(2).split(",")]) for m in re.finditer(r"constexpr unsigned char (\w+)\[4\] = \{([^}]*)\};", text))
    assert sorted(rows) == sorted(OPS)
    assert all(len(v) == 4 and set(v) <= {16, 31} for v in rows.values())
    disp = open(os.path.join(ROOT, "fastlanes_amd", "csrc", "fl_dispatch.hpp")).read()
    for op in OPS:
        assert f"window_table::{op}[t]" in disp, op
    # the fused encode is not windowed unless a same-buffer A/B on two boxes says so
    assert rows["TRANSPOSE_DELTA_PACK"] == [31, 31, 31, 31] or "transpose_delta_pack" in text
