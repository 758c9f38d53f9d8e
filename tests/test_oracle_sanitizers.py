"""CPU: the oracle's C restatement under AddressSanitizer + UBSan (GPU sanitizers are not available on the pool)."""
import os
import subprocess

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


def test_oracle_selftest_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "selftest")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-o", exe, os.path.join(ORACLE, "selftest.c"), os.path.join(ORACLE, "mg_oracle.c")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("oracle selftest ok")
