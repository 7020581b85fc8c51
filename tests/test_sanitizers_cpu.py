"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU oracle (every entry point, sphere
and triangle scenes, OpenMP on) -- sanitizers run on the CPU build only on this pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("make") is None, reason="no gcc/make")
def test_oracle_under_asan_ubsan():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "cannot find -lasan" in (r.stderr + r.stdout):
        pytest.skip("libasan not installed")
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "selftest ok" in r.stdout and "runtime error" not in (r.stdout + r.stderr) and "AddressSanitizer" not in r.stderr
