"""Development probe (dev library): how much do longer candidate lists buy?  C3's frame at sphere counts the 16-wave form
takes with room to spare, lists of 6 / 8 / 12 / 16 entries per lane (RT355_BVH_CAP).
usage: RT355_LIB=tools/bin/librt355_dev.so python tools/cap_sweep.py [counts]"""
import sys, os, subprocess
code = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import compute_raytracer_amd as rt
cfg = rt.BASELINE_CONFIGS["C3"]
n = int(sys.argv[1])
scene = rt.synthetic_scene(n, cfg["seed"])
r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"]).initialize()
ms = []
for _ in range(10):
    r.render(); ms.append(r.stats()["kernel_ms"])
best = 1e9
for _ in range(3):
    r.wait(); t0 = time.perf_counter()
    for _ in range(24): r.enqueue()
    r.wait(); best = min(best, (time.perf_counter() - t0) / 24 * 1e3)
print("N %5d cap %s kid %2d: %.3f one at a time, %.3f in flight" % (n, os.environ["RT355_BVH_CAP"], r.stats()["kernel_id"], min(ms[2:]), best), flush=True)
r.close()
'''
for n in (sys.argv[1] if len(sys.argv) > 1 else "2000,2800,3400").split(","):
    for cap in ("6", "8", "12", "16"):
        subprocess.run([sys.executable, "-c", code, n], env=dict(os.environ, RT355_BVH_CAP=cap, RT355_BVH_CMP="0"), check=False)
