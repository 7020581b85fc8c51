"""Development probe of the triangle kernel (rt_triangles.hip; variant 0 against 6, or one build or form against another) on the GPU box: the reference's own
scene (REF), the procedural scene of its size at 1344x846 (TRI) and at 3840x2160 (TRI4K).  Per configuration and
variant: the frame's sha256 (both kernels must agree), kernel time one frame at a time (hipEvents, min / median) and
wall time per frame with frames in flight.
usage: [RT355_LIB=tools/bin/librt355_dev.so RT355_TRI_...=..] python tools/tri_ab_probe.py [REF TRI TRI4K] [v0 v6] [label]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import compute_raytracer_amd as rt
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA

args = sys.argv[1:]
configs = [a for a in args if a in ("REF", "TRI", "TRI4K")] or ["REF", "TRI", "TRI4K"]
variants = [int(a[1:]) for a in args if a in ("v0", "v6")] or [0, 6]
label = " ".join("%s=%s" % (k[6:], v) for k, v in sorted(os.environ.items()) if k.startswith("RT355_TRI")) + " " + \
        " ".join(a for a in args if a not in ("REF", "TRI", "TRI4K", "v0", "v6"))
_tri = None
for name in configs:
    if name == "REF":
        from helpers import ref_fixture
        scene, sky, W, H, B, canvas, pin = ref_fixture()
        mat = rt.Material.white()
    else:
        from compute_raytracer_amd.procedural import triangle_scene
        if _tri is None:
            _tri = triangle_scene(seed=21, n_models=2, rings=48, sectors=64)
        scene, mat = _tri
        sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
        W, H, B = (1344, 846, 4) if name == "TRI" else (3840, 2160, 4)
    hashes = {}
    for v in variants:
        r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
        r.set_variant(v)
        r.recalculateScene()
        for _ in range(3): r.render()
        img = r.read_pixels()
        hashes[v] = hashlib.sha256(img.tobytes()).hexdigest()[:16]
        ms, wall = [], []
        for _ in range(30):
            t0 = time.perf_counter(); r.render(); wall.append((time.perf_counter() - t0) * 1e3); ms.append(r.stats()["kernel_ms"])
        kid = r.stats()["kernel_id"]
        res = []
        for _ in range(8): r.enqueue()
        r.wait()
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(32): r.enqueue()
            r.wait()
            res.append((time.perf_counter() - t0) / 32 * 1e3)
        print("%-5s v%d kid %2d %s | awaited kernel min %.3f med %.3f wall min %.3f | in flight min %.3f med %.3f | rays %d | %s" % (
            name, v, kid, hashes[v], min(ms), sorted(ms)[15], min(wall), min(res), sorted(res)[2], r.stats()["rays"], label), flush=True)
        r.close()
    if len(hashes) > 1:
        print("%-5s frames %s" % (name, "IDENTICAL" if len(set(hashes.values())) == 1 else "DIFFER " + str(hashes)), flush=True)
