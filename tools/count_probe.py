import sys, os
sys.path.insert(0, os.getcwd())
import compute_raytracer_amd as rt
cfg = rt.BASELINE_CONFIGS["C3"]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"]).initialize()
r.render()
print(os.environ.get("RT355_LIB"), "counter =", r.stats()["rays"])
r.close()
