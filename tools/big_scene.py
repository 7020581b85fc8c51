"""Development check: a scene far larger than a CU's LDS (nodes read from global memory / L2).
Fast mode (hierarchy) and strict mode (literal loop) must produce the same frame.
usage: python tools/big_scene.py [spheres=100000]"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import compute_raytracer_amd as rt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rng = np.random.default_rng(7)
spheres = [rt.Sphere([0.0, -1000.0, 0.0], 1000.0, [0.8, 0.8, 0.8])]
pos = np.stack([rng.uniform(-60, 60, n), rng.uniform(0.1, 6.0, n), rng.uniform(-120, -3, n)], axis=1)
rad = rng.uniform(0.03, 0.25, n)
col = rng.uniform(0.2, 1.0, (n, 3))
spheres += [rt.Sphere(pos[i], float(rad[i]), col[i]) for i in range(n)]
scene = rt.SceneRaytracing().createScene(spheres)
W, H, B = 1280, 720, 6
res = {}
for name, strict in (("fast", False), ("strict", True)):
    r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize()
    r.set_mode(strict)
    t0 = time.perf_counter(); r.render(); first = (time.perf_counter() - t0) * 1e3
    r.render(); st = r.stats()
    img = r.read_pixels(); r.close()
    res[name] = (hashlib.sha256(img.tobytes()).hexdigest(), st["rays"])
    print("%-6s first frame %.1f ms (scene upload + build), kernel %.2f ms, rays %d, %.1f Mrays/s" % (name, first, st["kernel_ms"], st["rays"], st["rays"] / st["kernel_ms"] / 1e3), flush=True)
assert res["fast"] == res["strict"], "fast and strict frames differ"
print("frames identical:", res["fast"][0][:16])
