"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

import compute_raytracer_amd as rt
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA


def config_inputs(name, width=None, height=None, spheres=None, bounces=None):
    cfg = dict(rt.BASELINE_CONFIGS[name])
    if width: cfg["width"] = width
    if height: cfg["height"] = height
    if spheres: cfg["spheres"] = spheres
    if bounces is not None: cfg["bounces"] = bounces
    scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
    return cfg, scene


def gpu_render(scene, width, height, bounces, strict, skybox=None, variant=0, rank=0, world=1):
    r = rt.RendererRaytracing(width, height, scene, maxBounces=bounces, rank=rank, world=world)
    r.initialize(skybox)
    r.set_mode(strict)
    r.set_variant(variant)
    r.render()
    img = r.read_pixels()
    st = r.stats()
    r.close()
    return img, st


def oracle_render(oracle, scene, width, height, bounces, skybox=None, want_float=False, **kw):
    sky = skybox if skybox is not None else rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    return oracle.render(scene.pack_params(bounces), scene.pack_spheres(), sky.faces, width, height,
                         want_float=want_float, **kw)


def diff_stats(a, b):
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))[..., :3].max(axis=-1)
    n = d.size
    return {
        "pixels": n,
        "exact": float((d == 0).sum()) / n,
        "within1": float((d <= 1).sum()) / n,
        "within2": float((d <= 2).sum()) / n,
        "max": int(d.max()),
        "n_gt1": int((d > 1).sum()),
    }


# ---- procedural triangle scenes: compute_raytracer_amd/procedural.py (bench.py --config TRI uses them too) ----
from compute_raytracer_amd.procedural import obj_floor, obj_uv_sphere, tri_buffers, triangle_scene  # noqa: E402,F401


def gpu_render_tri(scene, material, width, height, bounces, skybox=None, heatmap=False, variant=0):
    """variant 0: the library's choice (pair records where the scene fits them), 6: the reference's node buffer only."""
    import compute_raytracer_amd as rt
    r = rt.RendererRaytracing(width, height, scene, maxBounces=bounces)
    r.initialize(skybox, material)
    r.set_variant(variant)
    if heatmap:
        r.showHeatmap()
    r.render()
    img = r.read_pixels()
    st = r.stats()
    r.close()
    return img, st


# ---- the reference's own scene, as committed fixtures (tests/golden/make_ref_scene.py) ----
def ref_fixture():
    """-> (scene, sky CubemapMaterial, W, H, maxBounces, canvas (H,W,3) uint8, pin dict).  Reads only tests/golden/."""
    import json
    import os
    from PIL import Image
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    d = np.load(os.path.join(g, "ref_scene.npz"))
    scene = rt.SceneRaytracing.from_packed(d)
    strip = np.array(Image.open(os.path.join(g, "ref_sky.png")).convert("RGBA"), dtype=np.uint8)
    sky = rt.CubemapMaterial()
    n = strip.shape[0]
    sky.faces = [np.ascontiguousarray(strip[:, k * n:(k + 1) * n]) for k in range(6)]
    canvas = np.array(Image.open(os.path.join(g, "ref_canvas.png")).convert("RGB"), dtype=np.uint8)
    pin = json.load(open(os.path.join(g, "ref_pin.json")))
    return scene, sky, int(d["W"]), int(d["H"]), int(d["maxBounces"]), canvas, pin
