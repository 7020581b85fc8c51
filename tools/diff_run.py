"""One-off differential run (development): many random scenes through the hierarchy kernel against the CPU oracle,
beyond what the test suite holds -- scene sizes 3..3000 spheres over five orders of magnitude of scale and offset,
flat and textured skies, bounce limits 0..9, ragged frame sizes.  The oracle is test infrastructure; this is a test.
usage: python tools/diff_run.py [scenes=300] [first seed=5000] [compact]
`compact`: only scenes the host plans the sign-aware node test for (reach < 342) -- the ones whose shadow rays walk backwards."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import compute_raytracer_amd as rt
from oracle import rt_oracle_py as oracle
from helpers import gpu_render, oracle_render, diff_stats

COMPACT = len(sys.argv) > 3 and sys.argv[3] == "compact"

def scene_of(seed):
    rng = np.random.default_rng(seed)
    scale = float(10 ** (rng.uniform(-2, 1.2) if COMPACT else rng.uniform(-2, 3)))
    offset = float(rng.choice([0.0, 0.0, 10.0, 100.0] if COMPACT else [0.0, 0.0, 10.0, 300.0, 3000.0, 1e5])) * rng.choice([-1, 1])
    centre = np.array([offset, offset * 0.5, -offset * 0.25])
    n = int(rng.choice([3, 17, 64, 200, 700, 1500, 3000]))
    ratio = float(10 ** rng.uniform(0, 2.5))
    spheres = []
    for _ in range(n):
        r = scale * 0.25 / ratio * float(10 ** rng.uniform(0, np.log10(ratio)))
        spheres.append(rt.Sphere(centre + rng.normal(size=3) * scale * (1.0 + 0.002 * n) , r, rng.uniform(0.1, 1.0, 3)))
    if rng.random() < 0.5:
        R = scale * float(10 ** (rng.uniform(0.5, 1.0) if COMPACT else rng.uniform(1, 2)))
        spheres.append(rt.Sphere(centre + np.array([0, -R - scale, 0]), R, [0.8, 0.8, 0.8]))
    scene = rt.SceneRaytracing().createScene(spheres)
    scene.camera.position = list(centre + np.array([0.0, 0.5 * scale, 3.0 * scale]))
    scene.camera.eulers = np.array([270.0 + rng.uniform(-20, 20), 95.0 + rng.uniform(-15, 15)], np.float32)
    scene.camera.update()
    scene.light.position = list(centre + np.array([0.3 * scale, 2.5 * scale, 0.5 * scale]))
    if rng.random() < 0.2:
        k = int(rng.integers(0, len(spheres)))
        (scene.camera if rng.random() < 0.5 else scene.light).position = [float(v) for v in spheres[k].center]
    sky = None
    if rng.random() < 0.5:
        m = int(rng.choice([1, 2, 5, 16]))
        sky = rt.CubemapMaterial()
        sky.faces = [rng.integers(0, 256, (m, m, 4), dtype=np.uint8) for _ in range(6)]
    W, H = int(rng.integers(40, 140)), int(rng.integers(30, 100))
    B = int(rng.choice([0, 1, 2, 3, 5, 9]))
    return scene, sky, W, H, B, dict(seed=seed, n=len(spheres), scale=scale, offset=offset, sky=None if sky is None else sky.faces[0].shape[0], W=W, H=H, B=B)

count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
bad, rays_total, t0 = 0, 0, time.time()
for seed in range(first, first + count):
    scene, sky, W, H, B, info = scene_of(seed)
    ref, _, rays = oracle_render(oracle, scene, W, H, B, skybox=sky)
    img, st = gpu_render(scene, W, H, B, strict=False, skybox=sky, variant=4)
    ok = np.array_equal(img, ref) and st["rays"] == rays
    rays_total += rays
    if not ok:
        bad += 1
        print("MISMATCH", info, diff_stats(img, ref), st["rays"], rays, flush=True)
    if (seed - first) % 50 == 49:
        print("... %d scenes, %d mismatches, %.0f s" % (seed - first + 1, bad, time.time() - t0), flush=True)
print("diff_run: %d scenes (seeds %d..%d), %d rays, mismatches %d, %.0f s" % (count, first, first + count - 1, rays_total, bad, time.time() - t0))
