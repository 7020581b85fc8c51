"""Soak test (development) of the triangle kernel's work list (rt_triangles.hip: order_tiles): frames large enough for it
(>= 4096 tiles), rendered the way the reference does -- scene.update, the per-frame writes, render, WAIT --, so that every
frame starts its tiles in the order (and with the quarters) the frame four renders back on the same stream suggests, while
camera and models move; phases of frames in flight in between (row-major order, then back).  EVERY frame is compared with
the oracle's.  usage: python tools/soak_tri_order.py [frames=240] [W=1024] [H=516]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import compute_raytracer_amd as rt
from compute_raytracer_amd.procedural import triangle_scene, tri_buffers
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
from oracle import rt_oracle_py as orc

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 240
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H = int(sys.argv[3]) if len(sys.argv) > 3 else 516
B = 4
scene, mat = triangle_scene(seed=33, n_models=3, rings=14, sectors=18)
sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
rng = np.random.default_rng(7)
bad, t0, f = [], time.time(), 0
kinds = {}
while f < frames:
    if (f // 40) % 3 == 2:                       # a batch in flight: four frames, each read back, then waited
        k = min(4, frames - f)
        host, want = r.host_frames(k), []
        for i in range(k):
            scene.update(float(rng.uniform(0.01, 0.3)))
            scene.camera.move(float(rng.uniform(-0.05, 0.05)), float(rng.uniform(-0.05, 0.05)))
            want.append(orc.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)[0])
            r.recalculateScene(); r.enqueue()
        for i in range(k):
            r.read_pixels_async(k - 1 - i, host[i])
        r.wait(); r.read_pixels_wait()
        bad += [f + i for i in range(k) if not np.array_equal(host[i].reshape(H, W, 4), want[i].reshape(H, W, 4))]
        f += k
        continue
    scene.update(float(rng.uniform(0.01, 0.3)))
    scene.camera.move(float(rng.uniform(-0.05, 0.05)), float(rng.uniform(-0.05, 0.05)))
    if f % 50 == 49:                             # now and then the picture jumps: the list was made for another one
        scene.camera.move(float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-1.5, 1.5)))
    ref, _, rays = orc.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)
    r.render()
    st = r.stats()
    kinds[rt.abi.KERNEL_IDS[st["kernel_id"]]] = kinds.get(rt.abi.KERNEL_IDS[st["kernel_id"]], 0) + 1
    if not np.array_equal(r.read_pixels().reshape(H, W, 4), ref.reshape(H, W, 4)) or st["rays"] != rays:
        bad.append(f)
    f += 1
    if f % 40 == 0:
        print("... %d frames, %d mismatching, %.0f s" % (f, len(bad), time.time() - t0), flush=True)
print("soak_tri_order: %d frames %dx%d (%d tiles), %.1f s, mismatching frames %d %s; awaited frames by kernel %s" % (frames, W, H, ((W + 7) // 8) * ((H + 7) // 8), time.time() - t0, len(bad), bad[:8], kinds))
r.close()
sys.exit(1 if bad else 0)
