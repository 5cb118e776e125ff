"""CPU: the oracle's C restatement under AddressSanitizer + UBSan.

`make -C oracle asan` builds libpn2_oracle_asan.so / libpn2_host_asan.so; the geometry golden tests (FPS, ball query incl. the
empty-ball / nsample edge cases, distance bits, 3-NN + interpolation) then run in a CHILD interpreter that preloads libasan and
points oracle/geometry.py at the sanitizer build (PN2_ORACLE_SO).  Any report aborts the child (halt_on_error)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    if shutil.which("gcc") is None:
        return None
    path = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


@pytest.mark.skipif(_libasan() is None, reason="gcc / libasan not on this machine")
def test_oracle_geometry_golden_under_asan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = dict(os.environ)
    env.update({"LD_PRELOAD": _libasan(), "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=1",
                "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1", "PN2_ORACLE_SO": os.path.join(ROOT, "oracle", "libpn2_oracle_asan.so"),
                "PYTHONDONTWRITEBYTECODE": "1"})
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "fps_golden or ball_query or square_distance or three_nn"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert "passed" in p.stdout and "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr
