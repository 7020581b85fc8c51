"""Development probe (dev build: tools/build_dev.sh): where an AWAITED triangle frame spends its time.  Every workgroup of the
frame leaves {start, end, tile, part}; printed: the frame's span, how many workgroups run at each moment (deciles of the span),
the workgroups that finish last, and what an ideal packing of the measured durations onto the resident wave slots would take.
usage: RT355_LIB=tools/bin/librt355_dev.so RT355_TRI_TIMELINE=1 python tools/tri_timeline.py [REF|TRI|TRI4K] [frames=12]"""
import ctypes, heapq, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("RT355_TRI_TIMELINE", "1")
import numpy as np
import compute_raytracer_amd as rt
from compute_raytracer_amd import abi
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
name = sys.argv[1] if len(sys.argv) > 1 else "REF"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 12
if name == "REF":
    from helpers import ref_fixture
    scene, sky, W, H, B, canvas, pin = ref_fixture(); mat = rt.Material.white()
else:
    from compute_raytracer_amd.procedural import triangle_scene
    scene, mat = triangle_scene(seed=21, n_models=2, rings=48, sectors=64)
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    W, H, B = (1344, 846, 4) if name == "TRI" else (3840, 2160, 4)
r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
L = abi.load()
L.rt_debug_tri_timeline.restype = ctypes.c_longlong
for _ in range(frames): r.render()
kms = r.stats()["kernel_ms"]
buf = np.zeros(3 * 200000, np.uint64)
n = L.rt_debug_tri_timeline(r._ctx, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), ctypes.c_size_t(buf.size))
assert n > 0, n
t = buf[:n - n % 3].reshape(-1, 3)
t = t[t[:, 1] > 0]
start, end, tag = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2]
t0, t1 = start.min(), end.max()
us = lambda ticks: ticks / 100.0
dur = us(end - start)
print("%s awaited: kernel %.3f ms by events; %d workgroups, span %.1f us; durations: mean %.1f median %.1f p99 %.1f max %.1f us; sum %.1f ms-slots" % (
    name, kms, len(t), us(t1 - t0), dur.mean(), np.median(dur), np.percentile(dur, 99), dur.max(), dur.sum() / 1e3))
parts = tag & 0xFF
print("  whole tiles %d, quarters %d, sixteenths %d; last start at %.1f us" % ((parts == 4).sum(), (parts < 4).sum(), (parts >= 16).sum(), us(start.max() - t0)))
edges = np.linspace(t0, t1, 11)
for k in range(10):
    mid = (edges[k] + edges[k + 1]) / 2
    print("  t = %5.1f us: %5d workgroups running" % (us(mid - t0), int(((start <= mid) & (end > mid)).sum())))
order = np.argsort(end)[::-1][:12]
for i in order:
    print("  ends %.1f us: tile %d part %d started %.1f us ran %.1f us" % (us(end[i] - t0), int(tag[i] >> 8), int(tag[i] & 0xFF), us(start[i] - t0), dur[i]))
for slots in (4096, 5120, 6144):
    heap = [0.0] * slots
    for d in sorted(dur, reverse=True):
        heapq.heappush(heap, heapq.heappop(heap) + d)
    print("  the measured durations, longest first, packed onto %d slots: %.1f us" % (slots, max(heap)))
r.close()
