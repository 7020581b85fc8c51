"""Development probe: cost of per-frame sphere updates at C3.  For a sequence of moved sphere sets:
kernel time with (a) the hierarchy rebuilt from scratch for that set (fresh context), (b) the topology kept
and refitted on the device, with the worker thread's rebuilt topology taken over when ready; and the wall
time of write + render + wait against a static frame.  usage: python tools/update_probe.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import compute_raytracer_amd as rt
from compute_raytracer_amd import abi
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
from test_moving_spheres_gpu import Ctx

cfg = rt.BASELINE_CONFIGS["C3"]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
base = scene.pack_spheres()
p = scene.pack_params(cfg["bounces"])
n = base.shape[0]


def smooth(step):      # every sphere orbits a little: the motion of an animation loop
    s = base.copy()
    ph = np.arange(1, n) * 0.37
    s[1:, 0] += (0.25 * np.sin(0.2 * step + ph)).astype(np.float32)
    s[1:, 2] += (0.25 * np.cos(0.2 * step + ph)).astype(np.float32)
    return s


def kernel_ms(c, reps=4):
    best = 1e9
    for _ in range(reps):
        abi.check(c.L.rt_render(c.c), c.c); abi.check(c.L.rt_wait(c.c), c.c)
        st = abi.RtStats(); abi.check(c.L.rt_get_stats(c.c, ctypes.byref(st)), c.c)
        best = min(best, st.kernel_ms)
    return best


c = Ctx(cfg["width"], cfg["height"], sky)
c.params(p); c.spheres(base)
print("static base scene: kernel %.3f ms" % kernel_ms(c, 8))
for step in range(1, 9):
    s = smooth(step)
    t0 = time.perf_counter()
    c.spheres(s)
    abi.check(c.L.rt_render(c.c), c.c); abi.check(c.L.rt_wait(c.c), c.c)
    wall = (time.perf_counter() - t0) * 1e3
    st = abi.RtStats(); abi.check(c.L.rt_get_stats(c.c, ctypes.byref(st)), c.c)
    refit_first, prep = st.kernel_ms, st.prep_ms
    refit = kernel_ms(c, 3)
    f = Ctx(cfg["width"], cfg["height"], sky); f.params(p); f.spheres(s)
    fresh = kernel_ms(f, 4); f.close()
    print("step %d: write+render+wait %.3f ms (kernel %.3f, prep incl. refit %.3f) | same scene static: refitted topology %.3f ms, fresh build %.3f ms"
          % (step, wall, refit_first, prep, refit, fresh), flush=True)
    time.sleep(0.02)
c.close()
