"""CPU: the oracle's C restatement under AddressSanitizer + UndefinedBehaviorSanitizer
(the reference's own safety net is Rust's type system; the C port gets
sanitizers instead)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "selftest")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-pthread", "-I", os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "oracle", "selftest.c"), os.path.join(ROOT, "oracle", "fl_oracle.c"),
                           "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
