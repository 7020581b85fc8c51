"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU oracle (every entry point, sphere
and triangle scenes, OpenMP on) and over the host-side hierarchy build of the product
(rt_bvh_build.h, plain C++) -- sanitizers run on the CPU build only on this pool."""
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


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_hierarchy_build_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "bvh_build_san")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-fno-omit-frame-pointer", os.path.join(ROOT, "tests", "c", "bvh_build_san.cpp"), "-o", exe],
                       capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "cannot find -lasan" in (r.stderr + r.stdout):
        pytest.skip("libasan not installed")
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bvh build ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
    assert "runtime error" not in (r.stdout + r.stderr) and "AddressSanitizer" not in r.stderr
