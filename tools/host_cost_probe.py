"""Host cost of one rt_render (enqueue only) and of an awaited frame, on frames so small that the device is never the limit:
what the library's own bookkeeping + the HIP calls behind one frame cost a C host.  usage: python tools/host_cost_probe.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import compute_raytracer_amd as rt
from compute_raytracer_amd import abi
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
from compute_raytracer_amd.procedural import triangle_scene
L = abi.load()
sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
for kind in ("triangles", "spheres"):
    if kind == "triangles":
        scene, mat = triangle_scene(seed=3, n_models=2, rings=5, sectors=7)
        r = rt.RendererRaytracing(64, 64, scene, maxBounces=1).initialize(sky, mat)
    else:
        scene = rt.synthetic_scene(300, 5)
        r = rt.RendererRaytracing(64, 64, scene, maxBounces=1).initialize(sky)
    r.recalculateScene()
    for _ in range(8): r.enqueue()
    r.wait()
    best_e, best_a = 1e9, 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(48): L.rt_render(r._ctx)
        t1 = time.perf_counter()
        L.rt_wait(r._ctx)
        best_e = min(best_e, (t1 - t0) / 48 * 1e6)
        t0 = time.perf_counter()
        for _ in range(48):
            L.rt_render(r._ctx); L.rt_wait(r._ctx)
        best_a = min(best_a, (time.perf_counter() - t0) / 48 * 1e6)
    print("%-9s 64x64: rt_render (enqueue only) %.1f us per call; rt_render + rt_wait %.1f us per awaited frame (kernel %.1f us)" % (
        kind, best_e, best_a, r.stats()["kernel_ms"] * 1e3), flush=True)
    r.close()
