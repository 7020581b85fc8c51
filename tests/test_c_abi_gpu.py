"""The C ABI from a plain C program (gcc, -lrt355): the boundary is usable without Python, Node or
torch, with nothing but include/rt355.h."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import compute_raytracer_amd as rt
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")]


@pytest.mark.parametrize("strict", [0, 1])
def test_plain_c_host(tmp_path, oracle, strict):
    exe = str(tmp_path / "abi_render")
    libdir = os.path.join(ROOT, "compute_raytracer_amd")
    subprocess.run(["gcc", "-O1", "-std=c11", "-Wall", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "abi_render.c"), "-o", exe, "-L", libdir, "-lrt355",
                    "-Wl,-rpath," + libdir], check=True)
    W, H, N, B = 200, 120, 40, 4
    scene = rt.synthetic_scene(N, 4321)
    p, s = scene.pack_params(B), scene.pack_spheres()
    inp = str(tmp_path / "in.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("<4I", W, H, N, strict))
        f.write(p.tobytes()); f.write(s.tobytes()); f.write(bytes(CONSTANT_SKY_RGBA))
    out = str(tmp_path / "out.rgba")
    r = subprocess.run([exe, inp, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    ref, _, rays = oracle.render(p, s, sky.faces, W, H)
    got = np.fromfile(out, dtype=np.uint8).reshape(H, W, 4)
    assert np.array_equal(got, ref)
    assert "rays=%d " % rays in r.stdout and "abi=4" in r.stdout


def _build(tmp_path, name):
    exe = str(tmp_path / name)
    libdir = os.path.join(ROOT, "compute_raytracer_amd")
    subprocess.run(["gcc", "-O1", "-std=c11", "-Wall", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", name + ".c"), "-o", exe, "-L", libdir, "-lrt355",
                    "-Wl,-rpath," + libdir], check=True)
    return exe


@pytest.mark.parametrize("mode", ["group", "rank"])
def test_c3_through_the_group_api_matches_the_golden_frame(tmp_path, mode):
    """the multi-GPU gather lives behind the C ABI.  A plain C program renders the
    headline C3 frame through rt_group_create / rt_group_render (every visible device; the test box
    has one) and through rt_comm_init / rt_render_gather: the assembled frame must hash to the oracle's
    C3 frame and the members' ray counts must add up to the oracle's."""
    import hashlib
    import json
    exe = _build(tmp_path, "abi_group")
    fr = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["C3"]
    cfg = rt.BASELINE_CONFIGS["C3"]
    W, H, N, B = cfg["width"], cfg["height"], cfg["spheres"], cfg["bounces"]
    scene = rt.synthetic_scene(N, cfg["seed"])
    inp = str(tmp_path / "in.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("<4I", W, H, N, 0))
        f.write(scene.pack_params(B).tobytes()); f.write(scene.pack_spheres().tobytes()); f.write(bytes(CONSTANT_SKY_RGBA))
    out = str(tmp_path / "out.rgba")
    r = subprocess.run([exe, inp, out, mode], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr)
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == fr["sha256"]
    assert "rays=%d " % fr["rays"] in r.stdout, r.stdout


def test_group_api_small_frame_against_the_oracle(tmp_path, oracle):
    exe = _build(tmp_path, "abi_group")
    W, H, N, B = 203, 77, 150, 5            # 10 tiles, the last one partial
    scene = rt.synthetic_scene(N, 99)
    p, s = scene.pack_params(B), scene.pack_spheres()
    inp = str(tmp_path / "in.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("<4I", W, H, N, 0))
        f.write(p.tobytes()); f.write(s.tobytes()); f.write(bytes(CONSTANT_SKY_RGBA))
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    ref, _, rays = oracle.render(p, s, sky.faces, W, H)
    for mode in ("group", "rank"):
        out = str(tmp_path / ("out_%s.rgba" % mode))
        r = subprocess.run([exe, inp, out, mode], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (mode, r.returncode, r.stderr)
        assert np.array_equal(np.fromfile(out, dtype=np.uint8).reshape(H, W, 4), ref), mode
        assert "rays=%d " % rays in r.stdout
