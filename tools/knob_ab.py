"""Development probe for last-half-per-cent knobs of bvh_pixels: C3, 64-frame batches in flight (min of 5; repeatable
to about a microsecond) and, with `serial` as argument, 40 single frames (kernel time, min and median).
usage: [KNOB_WORLD=8] [KNOB_CONFIG=C5] [KNOB_BATCH=64] [RT355_LIB=tools/bin/librt355_dev.so RT355_BVH_TAIL=.. RT355_BVH_BLOCKS=..] python tools/knob_ab.py [serial] [label]"""
import sys, os, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import compute_raytracer_amd as rt
cfg = rt.BASELINE_CONFIGS[os.environ.get("KNOB_CONFIG", "C3")]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
world = int(os.environ.get("KNOB_WORLD", "1"))            # the share of rank 0 of `world` ranks
r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"], rank=0, world=world).initialize()
r.recalculateScene()
for _ in range(8): r.enqueue()
r.wait()
label = " ".join("%s=%s" % (k[6:], v) for k, v in sorted(os.environ.items()) if k.startswith("RT355_B")) + " " + " ".join(a for a in sys.argv[1:] if a != "serial")
if "serial" in sys.argv[1:]:
    ms = []
    for _ in range(3): r.render()
    for _ in range(40):
        r.render(); ms.append(r.stats()["kernel_ms"])
    print("serial", label, "min %.3f median %.3f" % (min(ms), sorted(ms)[20]))
    r.close(); sys.exit(0)
res = []
batch = int(os.environ.get("KNOB_BATCH", "64"))
for rep in range(5):
    r.wait(); t0 = time.perf_counter()
    for _ in range(batch): r.enqueue()
    r.wait()
    res.append((time.perf_counter() - t0) / batch * 1e3)
print("in flight", label, " ".join("%.3f" % x for x in res), "min %.3f" % min(res))
r.close()
