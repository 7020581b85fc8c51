"""Development probe: where the drop-in's loop spends its time on the reference's scene (per step: recalculateScene's host calls,
rt_render, rt_wait), with the instance buffers rewritten every step as the reference does, against a step that rewrites nothing.
usage: python tools/loop_breakdown.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import compute_raytracer_amd as rt
from compute_raytracer_amd import abi
from helpers import ref_fixture
scene, sky, W, H, B, canvas, pin = ref_fixture()
r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, rt.Material.white())
L, c = r._lib, r._ctx
for _ in range(30): r.render()
def run(n, recalc):
    t_rec = t_ren = t_wait = 0.0; k = []
    for _ in range(n):
        t0 = time.perf_counter()
        if recalc: r.recalculateScene()
        t1 = time.perf_counter()
        abi.check(L.rt_render(c), c)
        t2 = time.perf_counter()
        abi.check(L.rt_wait(c), c)
        t3 = time.perf_counter()
        t_rec += t1 - t0; t_ren += t2 - t1; t_wait += t3 - t2; k.append(r.stats()["kernel_ms"])
    return t_rec / n * 1e3, t_ren / n * 1e3, t_wait / n * 1e3, sorted(k)[n // 2]
for label, recalc in (("rewriting the same instance data", True), ("rewriting nothing", False)):
    a = run(300, recalc)
    print("%-34s recalculateScene %.3f  rt_render %.3f  rt_wait %.3f  (kernel %.3f)  sum %.3f ms" % ((label,) + a + (a[0] + a[1] + a[2],)))
# the instances turning, as in the reference's loop: K states prepared, one installed per step
states = []
for _ in range(300):
    scene.update(0.016); states.append(scene.frame)
t_rec = t_ren = t_wait = 0.0; k = []
for st in states:
    scene.frame = st
    t0 = time.perf_counter(); r.recalculateScene(); t1 = time.perf_counter()
    abi.check(L.rt_render(c), c); t2 = time.perf_counter()
    abi.check(L.rt_wait(c), c); t3 = time.perf_counter()
    t_rec += t1 - t0; t_ren += t2 - t1; t_wait += t3 - t2; k.append(r.stats()["kernel_ms"])
n = len(states)
print("%-34s recalculateScene %.3f  rt_render %.3f  rt_wait %.3f  (kernel %.3f)  sum %.3f ms   prep_ms %.3f" % ("new instance data every step", t_rec / n * 1e3, t_ren / n * 1e3, t_wait / n * 1e3, sorted(k)[n // 2], (t_rec + t_ren + t_wait) / n * 1e3, r.stats().get("prep_ms", 0.0)))
# is the turning mesh's extra kernel time the picture's (other poses cost more) or the list's (made from the previous pose)?  Every
# pose rendered twice in a row: the second render has the list of its own picture
k1, k2, kid = [], [], {}
for st in states[::3]:
    scene.frame = st
    r.recalculateScene(); abi.check(L.rt_render(c), c); abi.check(L.rt_wait(c), c); k1.append(r.stats()["kernel_ms"])
    abi.check(L.rt_render(c), c); abi.check(L.rt_wait(c), c); k2.append(r.stats()["kernel_ms"])
    kid[r.stats()["kernel_id"]] = kid.get(r.stats()["kernel_id"], 0) + 1
print("every third pose rendered twice: kernel median %.3f ms with the previous pose's list, %.3f with its own (kernel ids %s)"
      % (sorted(k1)[len(k1) // 2], sorted(k2)[len(k2) // 2], kid))
r.close()
