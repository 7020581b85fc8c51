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
    assert "rays=%d " % rays in r.stdout and "abi=1" in r.stdout
