"""Soak test (development) of the per-frame instance path and the streaming read-back on a triangle scene: an animation
loop that never drains on its own -- scene.update, the three per-frame writes, rt_render, the frame two renders back copied
out asynchronously -- with rt_wait only when the library's event ring asks for it; EVERY frame is compared with the oracle's.
usage: python tools/soak_tri.py [frames=1500] [W=160] [H=100]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import compute_raytracer_amd as rt
from compute_raytracer_amd.procedural import triangle_scene, tri_buffers
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
from oracle import rt_oracle_py as orc

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
W = int(sys.argv[2]) if len(sys.argv) > 2 else 160
H = int(sys.argv[3]) if len(sys.argv) > 3 else 100
B = 3
scene, mat = triangle_scene(seed=31, n_models=4, rings=10, sectors=12)
sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
r.render()
host = r.host_frames(frames)
refs = []
rng = np.random.default_rng(5)
t0 = time.time()
for f in range(frames):
    scene.update(float(rng.uniform(0.005, 0.2)))
    scene.camera.move(float(rng.uniform(-0.02, 0.02)), float(rng.uniform(-0.02, 0.02)))
    refs.append(orc.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)[0])
    r.recalculateScene()
    r.enqueue()
    if f >= 2:
        r.read_pixels_async(2, host[f - 2])
    if f % 61 == 60:
        r.wait()
r.read_pixels_async(1, host[frames - 2]); r.read_pixels_async(0, host[frames - 1])
r.wait(); r.read_pixels_wait()
bad = [f for f in range(frames) if not np.array_equal(host[f], refs[f])]
print("soak_tri: %d frames %dx%d, %.1f s, instance uploads %d, mismatching frames %d %s" % (frames, W, H, time.time() - t0, r.stats()["instance_uploads"], len(bad), bad[:8]))
r.close()
sys.exit(1 if bad else 0)
