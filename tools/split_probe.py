"""Development probe: ONE C3 frame rendered as `world` concurrent launches (one context per interleaved
share, all on device 0, each on its own stream) against the same frame as one launch: separates what
frames in flight gain from several queues / several pixel cursors from what they gain by overlapping tails.
usage: python tools/split_probe.py [world=4]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compute_raytracer_amd as rt

world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = rt.BASELINE_CONFIGS["C3"]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
one = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"]).initialize()
parts = [rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"], rank=k, world=world).initialize()
         for k in range(world)]
for r in [one] + parts:
    r.set_variant(4); r.render(); r.render()

def timed(fn, reps=20):
    best, ts = 1e9, []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]

def split():
    for r in parts: r.enqueue()
    for r in parts: r.wait()

print("one launch  : wall ms min %.3f median %.3f" % timed(one.render))
print("%d launches  : wall ms min %.3f median %.3f" % ((world,) + timed(split)))
for r in [one] + parts: r.close()
