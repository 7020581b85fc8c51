"""Development probe for the -DRT_BVH_COUNT=<mode> builds (the ray counter then carries the statistic):
usage: RT355_LIB=tools/bin/librt355_cN.so python tools/count_probe.py [frames=4] [world=1]"""
import sys, os
sys.path.insert(0, os.getcwd())
import compute_raytracer_amd as rt
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = rt.BASELINE_CONFIGS["C3"]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"], rank=0, world=world).initialize()
for _ in range(frames):
    r.render()
    st = r.stats()
    print(os.environ.get("RT355_LIB"), "kernel_ms %.3f counter = %d" % (st["kernel_ms"], st["rays"]))
r.close()
