"""Development probe (needs a -DRT_BVH_COUNT=6|7 build via RT355_LIB): per-wave ticks (100 MHz)
after pixel exhaustion (6) or in total (7), summed over the waves, for 1 and 8 emulated ranks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compute_raytracer_amd as rt
cfg = rt.BASELINE_CONFIGS["C3"]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
for world in (1, 8):
    r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"], rank=0, world=world).initialize()
    for _ in range(3):
        r.render()
    st = r.stats()
    waves = 6144
    print("world %d kernel %.3f ms  sum ticks %d  -> per wave %.1f us" % (world, st["kernel_ms"], st["rays"], st["rays"] / waves / 100.0))
    r.close()
