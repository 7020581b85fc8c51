"""Fuzzing the conservative filter of RT_MODE_FAST: the filter (fused arithmetic, inflated radii,
2^40-scaled records, expanded form for per-lane origins, sign-aware variant for compact scenes)
must never lose a sphere the reference's literal test accepts.  Random scenes over many orders of
magnitude of size, offset from the world origin, radius ratio and overlap; the fast frame must be
bit-identical to the oracle's (and to RT_MODE_STRICT's).  Any difference = a lost candidate."""
import numpy as np
import pytest

import compute_raytracer_amd as rt
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
from helpers import diff_stats, gpu_render, oracle_render

pytestmark = pytest.mark.gpu


def fuzz_scene(seed):
    """Spheres around a focus point `centre` at distance ~`scale` in front of the camera; the camera
    and the light ride along, so the whole scene can sit far from the world origin."""
    rng = np.random.default_rng(seed)
    scale = float(10 ** rng.uniform(-2, 3))                 # scene size: 0.01 .. 1000 units
    offset = float(rng.choice([0.0, 0.0, 10.0, 300.0, 3000.0, 1e5])) * rng.choice([-1, 1])
    centre = np.array([offset, offset * 0.5, -offset * 0.25])
    n = int(rng.choice([3, 17, 64, 200]))
    spheres = []
    ratio = float(10 ** rng.uniform(0, 2.5))                # largest / smallest radius
    for i in range(n):
        r = scale * 0.25 / ratio * float(10 ** rng.uniform(0, np.log10(ratio)))
        pos = centre + rng.normal(size=3) * scale
        spheres.append(rt.Sphere(pos, r, rng.uniform(0.1, 1.0, 3)))
    if rng.random() < 0.5:                                  # a big ground-like sphere
        R = scale * float(10 ** rng.uniform(1, 2))
        spheres.append(rt.Sphere(centre + np.array([0, -R - scale, 0]), R, [0.8, 0.8, 0.8]))
    scene = rt.SceneRaytracing().createScene(spheres)
    scene.camera.position = list(centre + np.array([0.0, 0.5 * scale, 3.0 * scale]))
    scene.camera.eulers = np.array([270.0, 95.0], np.float32)
    scene.camera.update()
    scene.light.position = list(centre + np.array([0.3 * scale, 2.5 * scale, 0.5 * scale]))
    if rng.random() < 0.25:                                 # camera or light inside a sphere
        k = int(rng.integers(0, len(spheres)))
        tgt = scene.camera if rng.random() < 0.5 else scene.light
        tgt.position = [float(v) for v in spheres[k].center]
    return scene, dict(scale=scale, offset=offset, n=len(spheres), ratio=ratio)


@pytest.mark.parametrize("seed", range(48))
def test_fast_mode_never_loses_a_hit(oracle, seed):
    scene, info = fuzz_scene(1000 + seed)
    W, H, B = 96, 64, 5
    ref, _, rays = oracle_render(oracle, scene, W, H, B)
    img, st = gpu_render(scene, W, H, B, strict=False, variant=1 if seed % 3 == 0 else 3)   # single kernel / pipeline
    assert np.array_equal(img, ref), (info, diff_stats(img, ref))
    assert st["rays"] == rays, info


@pytest.mark.parametrize("seed", range(6))
def test_strict_mode_on_the_same_scenes(oracle, seed):
    scene, info = fuzz_scene(1000 + 7 * seed)
    ref, _, rays = oracle_render(oracle, scene, 64, 48, 4)
    img, st = gpu_render(scene, 64, 48, 4, strict=True)
    assert np.array_equal(img, ref) and st["rays"] == rays, info
