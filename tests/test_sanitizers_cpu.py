"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU oracle (every entry point, sphere
and triangle scenes, OpenMP on) and over the host-only headers of the product: the hierarchy build
(rt_bvh_build.h), the relinked BLAS copy the default triangle kernel reads and the top-level-tree walk
that admits its small-stack forms (rt_flow_build.h, rt_tlas_fit.h: arbitrary caller node buffers), and
the multi-GPU exchange plan (rt_exchange_plan.h) -- sanitizers run on the CPU build only on this pool."""
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


def _san(tmp_path, source, ok_line):
    exe = str(tmp_path / os.path.splitext(source)[0])
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-fno-omit-frame-pointer", os.path.join(ROOT, "tests", "c", source), "-o", exe],
                       capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "cannot find -lasan" in (r.stderr + r.stdout):
        pytest.skip("libasan not installed")
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and ok_line in r.stdout, (r.stdout + r.stderr)[-3000:]
    assert "runtime error" not in (r.stdout + r.stderr) and "AddressSanitizer" not in r.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_relinked_blas_copy_and_tlas_walk_under_asan_ubsan(tmp_path):
    """rt_flow_build.h / rt_tlas_fit.h on builder-made trees, one-node buffers, cyclic / NaN / out-of-range garbage: no bad
    access, the kernel's invariants, and a walk over the pair records that visits what the walk over the nodes visits."""
    _san(tmp_path, "flow_build_test.cpp", "flow build ok")


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_exchange_plan_under_asan_ubsan(tmp_path):
    _san(tmp_path, "exchange_plan_test.cpp", "exchange plan ok")
