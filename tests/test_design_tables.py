"""CPU: every measured number DESIGN.md quotes in its generated block is what the committed profiles/ files hold (and the
test counts are what pytest collects): the block must equal what tools/design_tables.py generates."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_md_tables_match_profiles():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_tables.py"), "--check"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
