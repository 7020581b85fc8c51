"""Development probe: per-rank time of a frame when `world` ranks share it (ranks emulated one
after the other on one GPU): serial = one frame at a time (kernel time from the library's events),
pipelined = 24 frames enqueued back to back (wall time per frame, frames overlap on the device).
usage: python tools/scale_probe.py [C3|C5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compute_raytracer_amd as rt
cfg = rt.BASELINE_CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
ser, pipe = [], []
for world in (1, 2, 4, 8):
    ms, pm = [], []
    for rank in range(0, world, max(1, world // 2)):
        r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"], rank=rank, world=world).initialize()
        best = 1e9
        for _ in range(5):
            r.render(); best = min(best, r.stats()["kernel_ms"])
        ms.append(best)
        bestp = 1e9
        for _ in range(3):
            r.wait(); t0 = time.perf_counter()
            for _ in range(24):
                r.enqueue()
            r.wait(); bestp = min(bestp, (time.perf_counter() - t0) / 24 * 1e3)
        pm.append(bestp); r.close()
    ser.append("w%d %.3f" % (world, max(ms))); pipe.append("w%d %.3f" % (world, max(pm)))
print("serial   :", "  ".join(ser))
print("pipelined:", "  ".join(pipe))
