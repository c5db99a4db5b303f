"""CPU: the known-bad build (make BADSCAN=1 -> libfastlanes_amd_badscan.so, the build tests/test_gpu_full_check.py is shown to fail
on) is a PATCH kept with the tests, not code in the product headers -- and the patch still applies to the current sources."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fastlanes_amd", "csrc")


def test_product_sources_carry_no_test_scaffolding():
    for f in os.listdir(CSRC):
        if f.endswith((".hpp", ".hip", ".inc")):
            text = open(os.path.join(CSRC, f)).read()
            assert "FL_TEST_" not in text and "scan_lane_groups_r03" not in text and "4099" not in text, f


def test_known_bad_patch_applies(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checker", "make_badscan_sources.py"), CSRC, str(tmp_path)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    good = open(os.path.join(CSRC, "fl_chain.hpp")).read()
    bad = open(tmp_path / "fl_chain.hpp").read()
    assert bad != good and "scan_lane_groups_r03<T>(x[R - 1], lane)" in bad and "blk % 4099u == 4098u" in bad
    assert "cell_from_group_below<T>(incl, lane, 1)" in good and "cell_from_group_below<T>(incl, lane, 1)" not in bad
    # every other source is an unmodified copy (the public headers are reached by absolute path from the copy)
    for f in ("fl_widths.hpp", "fl_kernels.hpp", "fl_inst.hip"):
        assert open(tmp_path / f).read() == open(os.path.join(CSRC, f)).read(), f
    assert '"../../include/' not in open(tmp_path / "fl_capi.hip").read()
