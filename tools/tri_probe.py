"""Development probe for the triangle kernel: the procedural reference-sized scene at the two bench sizes,
one frame at a time (kernel ms) and with frames in flight (wall ms per frame), frame hash for A/B runs.
usage: python tools/tri_probe.py [label]"""
import sys, os, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compute_raytracer_amd as rt
from compute_raytracer_amd.procedural import triangle_scene

label = sys.argv[1] if len(sys.argv) > 1 else ""
scene, mat = triangle_scene(seed=21, n_models=2, rings=48, sectors=64)
for w, h in ((1344, 846), (3840, 2160)):
    sky = rt.CubemapMaterial.synthetic_daylight() if os.environ.get("RT355_PROBE_SKY") else None
    r = rt.RendererRaytracing(w, h, scene, maxBounces=4).initialize(sky, mat)
    ms = []
    for _ in range(10):
        r.render(); ms.append(r.stats()["kernel_ms"])
    digest = hashlib.sha256(r.read_pixels().tobytes()).hexdigest()[:12]
    rays = r.stats()["rays"]
    best = 1e9
    for _ in range(3):
        r.wait(); t0 = time.perf_counter()
        for _ in range(24):
            r.enqueue()
        r.wait(); best = min(best, (time.perf_counter() - t0) / 24 * 1e3)
    print("%s %dx%d: serial kernel ms min %.3f median %.3f  in flight %.3f  (%.1f Grays/s)  rays %d  frame %s"
          % (label, w, h, min(ms[2:]), sorted(ms[2:])[4], best, rays / best / 1e6, rays, digest), flush=True)
    r.close()
