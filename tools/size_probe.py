"""Development probe: C3's scene at k x the pixels (same camera): one-frame-at-a-time kernel time and frames in
flight, to separate the per-pixel rate (slope) from the per-frame cost (intercept).
usage: python tools/size_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compute_raytracer_amd as rt
cfg = rt.BASELINE_CONFIGS["C3"]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
for w, h in ((1920, 1080), (3840, 1080), (3840, 2160), (7680, 2160), (7680, 4320)):
    r = rt.RendererRaytracing(w, h, scene, maxBounces=cfg["bounces"]).initialize()
    r.set_variant(4)
    ms = []
    for _ in range(12):
        r.render(); ms.append(r.stats()["kernel_ms"])
    rays = r.stats()["rays"]
    best = 1e9
    for _ in range(3):
        r.wait(); t0 = time.perf_counter()
        for _ in range(16):
            r.enqueue()
        r.wait(); best = min(best, (time.perf_counter() - t0) / 16 * 1e3)
    print("%5dx%-5d rays %10d  serial kernel ms min %.3f median %.3f   in flight %.3f" % (w, h, rays, min(ms[2:]), sorted(ms[2:])[5], best), flush=True)
    r.close()
