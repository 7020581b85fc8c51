"""One-off differential run (development) for trace_roles (rt_triangles.hip): random procedural triangle scenes in frames large
enough for a work list (>= 4096 tiles), rendered the way the reference does -- update, per-frame writes, render, WAIT -- for a few
frames each while the models turn and the camera drifts, so that from the third frame on the list of the previous frame splits
tiles and the frame runs with helper lanes.  EVERY frame against the oracle: pixels and ray count.  Bounce limits 0-6, flat and
textured skies, 1-12 models.  The summary counts the awaited frames per kernel (rt_stats.kernel_id) and stack form.
usage: python tools/diff_run_roles.py [scenes=40] [first seed=71000] [frames per scene=7]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import compute_raytracer_amd as rt
from compute_raytracer_amd.procedural import triangle_scene, tri_buffers
from oracle import rt_oracle_py as oracle

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 71000
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 7
bad, kinds, rays_total, t0 = [], {}, 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    scene, mat = triangle_scene(seed=seed, n_models=int(rng.integers(1, 13)), rings=int(rng.integers(4, 16)), sectors=int(rng.integers(5, 20)))
    scene.update(float(rng.uniform(0, 3)))
    scene.camera.position = [float(rng.uniform(-2, 2)), float(rng.uniform(0.5, 3)), float(rng.uniform(0, 4))]
    scene.camera.eulers = np.array([270.0 + rng.uniform(-25, 25), 95.0 + rng.uniform(-15, 25)], np.float32)
    scene.camera.update()
    scene.light.position = [float(rng.uniform(-4, 4)), float(rng.uniform(2, 8)), float(rng.uniform(-8, 2))]
    m = int(rng.choice([1, 1, 4]))
    sky = rt.CubemapMaterial()
    sky.faces = [rng.integers(0, 256, (m, m, 4), dtype=np.uint8) for _ in range(6)]
    W, H, B = int(rng.integers(700, 1100)), int(rng.integers(420, 640)), int(rng.choice([0, 1, 2, 3, 4, 4, 6]))
    r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
    for f in range(frames):
        scene.update(float(rng.uniform(0.0, 0.2)))
        scene.camera.move(float(rng.uniform(-0.03, 0.03)), float(rng.uniform(-0.03, 0.03)))
        ref, _, rays = oracle.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)
        r.render()
        st = r.stats()
        key = (rt.abi.KERNEL_IDS[st["kernel_id"]], st["tri_form"])
        kinds[key] = kinds.get(key, 0) + 1
        rays_total += rays
        if not np.array_equal(r.read_pixels().reshape(H, W, 4), ref.reshape(H, W, 4)) or st["rays"] != rays:
            bad.append((seed, f, W, H, B, key))
            print("MISMATCH", bad[-1], flush=True)
    r.close()
    if (seed - first) % 5 == 4:
        print("... %d scenes, %d mismatching frames, %.0f s" % (seed - first + 1, len(bad), time.time() - t0), flush=True)
print("diff_run_roles: %d scenes x %d awaited frames (seeds %d..%d), %d rays, mismatching frames %d %s, %.0f s; frames by (kernel, stack form) %s"
      % (count, frames, first, first + count - 1, rays_total, len(bad), bad[:6], time.time() - t0, sorted(kinds.items())))
sys.exit(1 if bad else 0)
