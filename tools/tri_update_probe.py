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

# the library's share alone: the per-frame buffers of K animation steps are built beforehand, the timed loop only
# writes them (rt_write_blas / _blas_lookup / _nodes through ctypes) and renders
import ctypes
from compute_raytracer_amd import abi
L, c, fp = abi.load(), r._ctx, ctypes.POINTER(ctypes.c_float)
states = []
for _ in range(K):
    scene.update(0.016)
    states.append(tuple(np.ascontiguousarray(x, dtype=np.float32).copy() for x in (scene.pack_blas(), scene.pack_blas_lookup(), scene.pack_tlas_nodes())))
# the same picture every frame: a frame's cost depends on what is in it, so the comparison with the static frame rewrites
# ONE state K times (every write still bumps the library's generation: every frame carries its instance data)
same = "--moving" not in sys.argv
if same:
    states = [states[-1]] * K

def prebuilt(in_flight):
    t0 = time.perf_counter()
    for b, bl, na in states:
        L.rt_write_blas(c, b.ctypes.data_as(fp), b.shape[0])
        L.rt_write_blas_lookup(c, bl.ctypes.data_as(fp), bl.shape[0])
        L.rt_write_nodes(c, 0, na.ctypes.data_as(fp), na.shape[0])
        r.enqueue()
        if not in_flight:
            r.wait()
    r.wait()
    return (time.perf_counter() - t0) / K * 1e3

def phases(animated):
    """host time of the three phases of a frame, one frame at a time: writes, rt_render (enqueue), rt_wait"""
    tw = te = tq = 0.0
    for b, bl, na in states:
        t0 = time.perf_counter()
        if animated:
            L.rt_write_blas(c, b.ctypes.data_as(fp), b.shape[0])
            L.rt_write_blas_lookup(c, bl.ctypes.data_as(fp), bl.shape[0])
            L.rt_write_nodes(c, 0, na.ctypes.data_as(fp), na.shape[0])
        t1 = time.perf_counter()
        L.rt_render(c)
        t2 = time.perf_counter()
        L.rt_wait(c)
        t3 = time.perf_counter()
        tw += t1 - t0; te += t2 - t1; tq += t3 - t2
    return tuple(round(v / K * 1e3, 4) for v in (tw, te, tq))

def stats_line(tag):
    st = r.stats()
    print("  %s: device times of the last frame: instance upload + prep %.4f ms, ray-trace kernel %.4f ms" % (tag, st["prep_ms"], st["kernel_ms"]))

for f in (serial, flight):
    f(False); f(True)
prebuilt(False); prebuilt(True)
for name, fl, st in (("one frame at a time", False, serial), ("frames in flight", True, flight)):
    s0 = min(st(False) for _ in range(5)); a0 = min(prebuilt(fl) for _ in range(5))
    print("%s, prebuilt instance buffers: static %.3f ms, animated %.3f ms: ratio %.3f" % (name, s0, a0, a0 / s0), flush=True)
    st(False); stats_line("static"); prebuilt(fl); stats_line("animated")
print("host ms per frame (writes, rt_render, rt_wait): static", phases(False), "animated", phases(True), phases(False), phases(True))
host = min(host_only() for _ in range(3))
wr = min(writes_only() for _ in range(3))
print("host: scene.update %.3f ms, the per-frame writes (4 ctypes calls, no device work) %.3f ms" % (host, wr))
for name, f in (("one frame at a time", serial), ("frames in flight", flight)):
    s = min(f(False) for _ in range(5)); a = min(f(True) for _ in range(5))
    print("%s: static %.3f ms, animated %.3f ms (of which scene.update on the host %.3f ms): ratio %.3f, without the host's own scene build %.3f"
          % (name, s, a, host, a / s, (a - host) / s), flush=True)
print("instance uploads:", r.stats()["instance_uploads"])
r.close()
