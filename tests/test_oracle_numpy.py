"""The C oracle against the independent numpy restatement (oracle/rt_oracle_np.py): two
separately written readings of the WGSL must agree bit for bit, pre-quantisation floats
included.  (The sphere primitive is dead code upstream: this cross-check, not the screenshot of tests/test_ref_pin.py, is what stands behind it.)"""
import numpy as np
import pytest

import compute_raytracer_amd as rt
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
from oracle import rt_oracle_np as onp


def random_sky(seed, w=5, h=7):
    rng = np.random.default_rng(seed)
    m = rt.CubemapMaterial()
    m.faces = [rng.integers(0, 256, (h, w, 4), dtype=np.uint8) for _ in range(6)]
    return m


CASES = [
    (96, 64, 20, 4, "cube"), (64, 48, 40, 3, "cube1"),
    (96, 64, 20, 4, "const"), (64, 48, 64, 8, "random"), (33, 17, 3, 1, "random"), (40, 40, 0, 3, "random"),
    (50, 30, 7, 16, "const"), (8, 8, 130, 2, "random"),
]


@pytest.mark.parametrize("W,H,N,B,sky", CASES)
def test_c_oracle_equals_numpy_restatement(oracle, W, H, N, B, sky):
    scene = rt.synthetic_scene(N, 1000 + N)
    p, s = scene.pack_params(B), scene.pack_spheres()
    faces = (rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA) if sky == "const" else
             random_sky(N, 4, 4) if sky == "cube" else random_sky(N, 1, 1) if sky == "cube1" else random_sky(N)).faces
    a, af, ar = oracle.render(p, s, faces, W, H, want_float=True)
    b, bf, br = onp.render(p, s, faces, W, H)
    assert np.array_equal(af.view(np.uint32), bf.view(np.uint32))
    assert np.array_equal(a, b)
    assert ar == br


def test_tile_selection_matches_full_frame(oracle, constant_sky):
    """rt_oracle_render(tile_first, tile_step) renders exactly the rows of the chosen 8-row tiles
    with the same values as the full frame (what the multi-GPU tests and bench.py rely on)."""
    scene = rt.synthetic_scene(16, 77)
    p, s = scene.pack_params(3), scene.pack_spheres()
    W, H = 40, 53                       # 7 tiles, the last one partial
    full, _, rays = oracle.render(p, s, constant_sky.faces, W, H)
    acc = np.zeros_like(full)
    total = 0
    for r in range(3):
        part, _, pr = oracle.render(p, s, constant_sky.faces, W, H, tile_first=r, tile_step=3)
        rows = [y for y in range(H) if (y // 8) % 3 == r]
        other = [y for y in range(H) if (y // 8) % 3 != r]
        assert np.array_equal(part[rows], full[rows])
        assert not part[other].any()
        acc[rows] = part[rows]
        total += pr
    assert np.array_equal(acc, full) and total == rays
