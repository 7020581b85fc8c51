"""Per-pixel work of the reference's own scene (REF) through the oracle: how many sequential traversal steps the
longest path takes, how the work is spread over the frame, and what a trip-synchronous persistent scheduler
(every active lane of a wave advances one step per trip) would make of it under a given pixel order."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import compute_raytracer_amd as rt
from oracle import rt_oracle_py as orc
from helpers import ref_fixture, tri_buffers

scene, sky, W, H, B, canvas, pin = ref_fixture()
buf = tri_buffers(scene, rt.Material.white())
cache = "/tmp/ref_work_px.npy"
if os.path.exists(cache):
    w = np.load(cache)
else:
    t0 = time.time()
    w = orc.tri_work_px(scene.pack_params(B), buf, sky.faces, W, H)
    print("oracle pass %.1f s" % (time.time() - t0))
    np.save(cache, w)
rays, inner, tris, inst = [w[..., k].astype(np.int64) for k in range(4)]
steps = inner + tris + inst          # state-machine trips a path needs (one node pair, one triangle or one instance set-up per trip)
print("pixels %d rays %d inner %d tris %d inst %d" % (rays.size, rays.sum(), inner.sum(), tris.sum(), inst.sum()))
print("per ray: inner %.1f tris %.1f inst %.2f" % (inner.sum() / rays.sum(), tris.sum() / rays.sum(), inst.sum() / rays.sum()))
print("steps per pixel: mean %.1f  p50 %d p90 %d p99 %d p99.9 %d max %d" % ((steps.mean(),) + tuple(np.percentile(steps, [50, 90, 99, 99.9, 100]).astype(int))))
rows = steps.reshape(H, W).sum(axis=1)
print("rows with work, cumulative share by quarter of the frame:", [round(float(rows[: (k + 1) * H // 4].sum() / rows.sum()), 3) for k in range(4)])
slots = 4096 * 64
print("perfect packing: %.0f trips per lane slot (%d lane slots)" % (steps.sum() / slots, slots))

def simulate(order, waves=4096, grab=64):
    """order: pixel indices in the order the cursor hands them out.  Each wave refills idle lanes every trip from the
    cursor (whole waves grab `grab` pixels at a time); a trip advances every active lane by one step.
    -> trips until the last wave is done (all waves run in lock step: a trip is the unit of time)."""
    st = steps.reshape(-1)[order]
    # lock-step approximation: a lane slot is busy for st[i] trips; slots take the next pixel when free (list scheduling)
    import heapq
    n = waves * 64
    if len(st) <= n:
        return int(st.max())
    free = [(int(s), k) for k, s in enumerate(st[:n])]
    heapq.heapify(free)
    for s in st[n:]:
        t, k = heapq.heappop(free)
        heapq.heappush(free, (t + int(s), k))
    return max(t for t, _ in free)

idx = np.arange(W * H)
tiles_x = (W + 7) // 8
def tile_major():
    ys, xs = np.divmod(idx, W)
    key = ((ys // 8) * tiles_x + xs // 8) * 64 + (ys % 8) * 8 + xs % 8
    return np.argsort(key, kind="stable")
tm = tile_major()
print("trips, tile-major top-down   :", simulate(tm))
print("trips, tile-major bottom-up  :", simulate(tm[::-1]))
print("trips, longest pixel first   :", simulate(np.argsort(-steps.reshape(-1), kind="stable")))
# tiles ordered by their summed cost (what order_tiles knows from the previous frame), pixels inside in order
tile_of = ((idx // W) // 8) * tiles_x + (idx % W) // 8
tcost = np.bincount(tile_of, weights=steps.reshape(-1))
print("trips, tiles costliest first :", simulate(tm[np.argsort(-tcost[tile_of[tm]], kind="stable")]))
tmax = np.zeros(tcost.size); np.maximum.at(tmax, tile_of, steps.reshape(-1))
print("trips, tiles by longest pixel:", simulate(tm[np.argsort(-tmax[tile_of[tm]], kind="stable")]))
