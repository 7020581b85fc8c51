"""Development probe: C3's frame (3840x2160, 8 bounces) at other sphere counts through the hierarchy (variant 4) and the
brute-force kernels (variant 5): one frame at a time (kernel ms, min of 8) and frames in flight (ms per frame).
usage: [KNOB_WORLD=8] python tools/count_sweep.py [counts, e.g. 96,160,256,512,768,1024]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compute_raytracer_amd as rt
cfg = rt.BASELINE_CONFIGS["C3"]
counts = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "96,160,256,512,768,1024").split(",")]
for n in counts:
    scene = rt.synthetic_scene(n, cfg["seed"])
    row = []
    for v in (4, 5):
        r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"], rank=0,
                                  world=int(os.environ.get("KNOB_WORLD", "1"))).initialize()
        r.set_variant(v)
        ms = []
        for _ in range(10):
            r.render(); ms.append(r.stats()["kernel_ms"])
        best = 1e9
        for _ in range(3):
            r.wait(); t0 = time.perf_counter()
            for _ in range(24): r.enqueue()
            r.wait(); best = min(best, (time.perf_counter() - t0) / 24 * 1e3)
        row.append("variant %d: %.3f one at a time, %.3f in flight" % (v, min(ms[2:]), best))
        rays = r.stats()["rays"]
        r.close()
    print("N %5d rays %9d  %s" % (n, rays, "   ".join(row)), flush=True)
