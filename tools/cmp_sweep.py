"""Development probe (dev library): C3's frame (3840x2160, 8 bounces) at sphere counts between the 12-wave exact form's limit
and the compact form's, hierarchy built with 4 / 6 children per node, exact 16-wave form against the compact 12-wave form:
node count, kernel id, one frame at a time (kernel ms, min of 8), frames in flight (ms per frame).
usage: RT355_LIB=tools/bin/librt355_dev.so python tools/cmp_sweep.py [counts]"""
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
print("N %5d arity %s cmp %s kid %2d: %.3f one at a time, %.3f in flight" % (n, os.environ["RT355_BVH_ARITY"], os.environ["RT355_BVH_CMP"], r.stats()["kernel_id"], min(ms[2:]), best), flush=True)
r.close()
'''
counts = (sys.argv[1] if len(sys.argv) > 1 else "1400,1800,2400,3200,4096").split(",")
for n in counts:
    for ar in ("4", "6"):
        for cmp in ("0", "1"):
            subprocess.run([sys.executable, "-c", code, n], env=dict(os.environ, RT355_BVH_ARITY=ar, RT355_BVH_CMP=cmp), check=False)
