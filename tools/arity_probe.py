"""Development probe: the sphere hierarchy built with 4 / 5 / 6 / 8 children per inner node (development library,
RT355_BVH_ARITY), C5 and C3: node count, frame hash, ms per frame in flight and one at a time.
usage: RT355_LIB=tools/bin/librt355_dev.so python tools/arity_probe.py"""
import os, subprocess, sys
code = r'''
import sys, os, time, hashlib
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import compute_raytracer_amd as rt
name = sys.argv[1]
cfg = rt.BASELINE_CONFIGS[name]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"]).initialize()
r.recalculateScene()
for _ in range(4): r.enqueue()
r.wait()
h = hashlib.sha256(np.ascontiguousarray(r.render()).tobytes()).hexdigest()[:16]
batch = 8 if name == "C5" else 64
res = []
for rep in range(3):
    r.wait(); t0 = time.perf_counter()
    for _ in range(batch): r.enqueue()
    r.wait(); res.append((time.perf_counter() - t0) / batch * 1e3)
ms = []
for _ in range(6 if name == "C5" else 20):
    r.render(); ms.append(r.stats()["kernel_ms"])
print(name, "arity", os.environ.get("RT355_BVH_ARITY", "4"), "kid", r.stats()["kernel_id"], h, "in flight %.3f" % min(res), "one at a time %.3f" % min(ms), flush=True)
r.close()
'''
for name in ("C5", "C3"):
    for a in ("4", "5", "6", "8"):
        env = dict(os.environ, RT355_BVH_ARITY=a)
        subprocess.run([sys.executable, "-c", code, name], env=env, check=False)
