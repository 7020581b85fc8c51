"""PCIe-inclusive rate of the boundary: render + rt_read_pixels of every frame into host memory
(the reference never reads the frame back: it blits colorBuffer to the canvas, RR:449-463)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compute_raytracer_amd as rt
cfg = rt.BASELINE_CONFIGS["C3"]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"]).initialize()
for _ in range(3):
    r.render(); img = r.read_pixels()
t0 = time.perf_counter(); n = 20
for _ in range(n):
    r.render(); img = r.read_pixels()
dt = (time.perf_counter() - t0) / n
rays = r.stats()["rays"]
t1 = time.perf_counter()
for _ in range(n):
    img = r.read_pixels()
rd = (time.perf_counter() - t1) / n
print("render+readback %.3f ms/frame -> %.1f Mrays/s, %.1f fps; readback alone %.3f ms (%.1f GB/s)" % (dt * 1e3, rays / dt / 1e6, 1 / dt, rd * 1e3, img.nbytes / rd / 1e9))
r.close()
