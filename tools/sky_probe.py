"""Development probe: C3 with the synthetic 6 x 512^2 sky, a few frames one at a time (for rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compute_raytracer_amd as rt
cfg = rt.BASELINE_CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"]).initialize(rt.CubemapMaterial.synthetic_daylight())
for _ in range(8):
    r.render()
print("kernel_ms", r.stats()["kernel_ms"])
r.close()
