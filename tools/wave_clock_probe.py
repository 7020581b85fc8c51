"""Development probe (-DRT_BVH_COUNT=13 build): per-wave start / cursor-exhausted / end times of one C3 frame.
usage: RT355_LIB=tools/bin/librt355_c13.so python tools/wave_clock_probe.py [world=1]"""
import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np
import compute_raytracer_amd as rt
from compute_raytracer_amd import abi
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = rt.BASELINE_CONFIGS["C3"]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"], rank=0, world=world).initialize()
lib = ctypes.CDLL(os.environ["RT355_LIB"])
for _ in range(3):
    r.render()
st = r.stats()
buf = np.zeros(3 * 8192, dtype=np.uint64)
rc = lib.rt_debug_wave_clock(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
w = buf.reshape(-1, 3)[:6144].astype(np.int64)
t0 = w[:, 0].min()
s, x, e = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0, (w[:, 2] - t0) / 100.0    # microseconds
q = lambda a: " ".join("%7.1f" % v for v in np.percentile(a, [0, 10, 50, 90, 99, 100]))
print("world %d kernel_ms %.3f rc %d   percentiles 0 10 50 90 99 100 (us)" % (world, st["kernel_ms"], rc))
print("  start      ", q(s))
print("  exhausted  ", q(x))
print("  end        ", q(e))
print("  end - exh. ", q(e - x))
for thr in (0.5, 0.25, 0.1, 0.05, 0.01):
    # time by which all but a fraction thr of the waves have ended
    print("  %4.0f%% of the waves still running at %.1f us" % (100 * thr, np.percentile(e, 100 * (1 - thr))))
r.close()
