"""Development probe: what per-frame instance updates cost a triangle scene.  The reference rewrites BLAS
records, BLAS lookup and TLAS nodes before every frame (RR:169-192, scene-raytracing.ts:138-143); this times a
static frame against an animated one (scene.update + the three writes + render), one frame at a time and with
frames in flight.  usage: python tools/tri_update_probe.py [W H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import compute_raytracer_amd as rt
from compute_raytracer_amd.procedural import triangle_scene
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1344, 846)
scene, mat = triangle_scene(seed=2, n_models=3, rings=48, sectors=64)
sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
r = rt.RendererRaytracing(W, H, scene, maxBounces=4).initialize(sky, mat)
r.render()
print("scene: %d triangles, %d instances, %dx%d" % (scene.triangleCount, len(scene.instances), W, H))
K = 40

def serial(animated):
    t0 = time.perf_counter()
    for _ in range(K):
        if animated:
            scene.update(0.016)
            r.recalculateScene()
        r.enqueue(); r.wait()
    return (time.perf_counter() - t0) / K * 1e3

def flight(animated):
    t0 = time.perf_counter()
    for _ in range(K):
        if animated:
            scene.update(0.016)
            r.recalculateScene()
        r.enqueue()
    r.wait()
    return (time.perf_counter() - t0) / K * 1e3

def host_only():
    t0 = time.perf_counter()
    for _ in range(K):
        scene.update(0.016)
    return (time.perf_counter() - t0) / K * 1e3

def writes_only():          # the four C-ABI writes of recalculateScene (params, BLAS, BLAS lookup, TLAS nodes) through ctypes
    t0 = time.perf_counter()
    for _ in range(K):
        r.recalculateScene()
    return (time.perf_counter() - t0) / K * 1e3

for f in (serial, flight):
    f(False); f(True)
host = min(host_only() for _ in range(3))
wr = min(writes_only() for _ in range(3))
print("host: scene.update %.3f ms, the per-frame writes (4 ctypes calls, no device work) %.3f ms" % (host, wr))
for name, f in (("one frame at a time", serial), ("frames in flight", flight)):
    s = min(f(False) for _ in range(5)); a = min(f(True) for _ in range(5))
    print("%s: static %.3f ms, animated %.3f ms (of which scene.update on the host %.3f ms): ratio %.3f, without the host's own scene build %.3f"
          % (name, s, a, host, a / s, (a - host) / s), flush=True)
print("instance uploads:", r.stats()["instance_uploads"])
r.close()
