"""One-off differential run (development): random procedural triangle scenes through trace_triangles and the
heatmap kernel against the CPU oracle -- instance counts, tessellations, cameras, skies, bounce limits, frame sizes.
Every scene through both ray-trace variants (0: one workgroup per tile over the relinked pair records, 6: over the node
buffer), frames one at a time and -- every fifth scene -- four frames in flight with the instances moving.
usage: python tools/diff_run_tri.py [scenes=120] [first seed=7000] [most models=6]
(most models = 20 reaches every stack form of the kernel: the summary line counts the frames per form, rt_stats.tri_form)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import compute_raytracer_amd as rt
from compute_raytracer_amd.procedural import triangle_scene, tri_buffers
from oracle import rt_oracle_py as oracle
from helpers import gpu_render_tri, diff_stats

count = int(sys.argv[1]) if len(sys.argv) > 1 else 120
first = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
most = int(sys.argv[3]) if len(sys.argv) > 3 else 6
forms = {}
bad, rays_total, t0 = 0, 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    scene, mat = triangle_scene(seed=seed, n_models=int(rng.integers(1, most + 1)), rings=int(rng.integers(3, 14)), sectors=int(rng.integers(4, 18)))
    scene.update(float(rng.uniform(0, 3)))
    scene.camera.position = [float(rng.uniform(-3, 3)), float(rng.uniform(0.3, 5)), float(rng.uniform(0, 5))]
    scene.camera.eulers = np.array([270.0 + rng.uniform(-40, 40), 95.0 + rng.uniform(-30, 40)], np.float32)
    scene.camera.update()
    scene.light.position = [float(rng.uniform(-4, 4)), float(rng.uniform(2, 8)), float(rng.uniform(-8, 2))]
    m = int(rng.choice([1, 1, 3, 8]))
    sky = rt.CubemapMaterial()
    sky.faces = [rng.integers(0, 256, (m, m, 4), dtype=np.uint8) for _ in range(6)]
    W, H, B = int(rng.integers(24, 200)), int(rng.integers(16, 130)), int(rng.choice([0, 1, 2, 4, 6]))
    bufs = tri_buffers(scene, mat)
    ref, _, rays = oracle.render_tri(scene.pack_params(B), bufs, sky.faces, W, H)
    ok = True
    for variant in (0, 6):
        img, st = gpu_render_tri(scene, mat, W, H, B, skybox=sky, variant=variant)
        ok = ok and np.array_equal(img, ref) and st["rays"] == rays
        if variant == 0:
            forms[st["tri_form"]] = forms.get(st["tri_form"], 0) + 1
    href, _ = oracle.heatmap_tri(scene.pack_params(B), bufs, W, H)
    himg, _ = gpu_render_tri(scene, mat, W, H, B, skybox=sky, heatmap=True)
    ok = ok and np.array_equal(himg, href)
    if (seed - first) % 5 == 4:                       # frames in flight, instances moving, each against the oracle
        for variant in (0, 6):
            r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
            r.set_variant(variant)
            host = r.host_frames(4)
            for batch in range(2):      # (the library learns how its caller enqueues at rt_wait: the second batch runs the in-flight form)
                want = []
                for f in range(4):
                    scene.update(0.17)
                    r.recalculateScene(); r.enqueue()
                    r.read_pixels_async(0, host[f])
                    want.append(oracle.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)[0])
                r.wait(); r.read_pixels_wait()
                if variant == 0:
                    f_ = r.stats()["tri_form"]; forms[(f_, "in flight")] = forms.get((f_, "in flight"), 0) + 1
                ok = ok and all(np.array_equal(host[f].reshape(H, W, 4), want[f]) for f in range(4))
            r.close()
    rays_total += rays
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, W, H, B, scene.triangleCount, diff_stats(img, ref), diff_stats(himg, href), flush=True)
    if (seed - first) % 20 == 19:
        print("... %d scenes, %d mismatches, %.0f s" % (seed - first + 1, bad, time.time() - t0), flush=True)
print("diff_run_tri: %d scenes (seeds %d..%d), %d rays, mismatches %d, %.0f s; frames per stack form %s" % (count, first, first + count - 1, rays_total, bad, time.time() - t0, sorted(forms.items(), key=str)))
