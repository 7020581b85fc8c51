"""Development probe for the -DRT_BVH_COUNT=<mode> builds of rt_bvh.hip (the frame's ray counter then carries the statistic).
usage: RT355_LIB=tools/bin/librt355_cN.so python tools/count_probe.py [config=C3] [frames=3] [world=1]   -> one JSON line"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import compute_raytracer_amd as rt
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = rt.BASELINE_CONFIGS[name]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
sky = None
if cfg["skybox"]:
    import numpy as np
    from PIL import Image
    strip = np.array(Image.open(os.path.join("tests", "golden", "ref_sky.png")).convert("RGBA"), dtype=np.uint8)
    sky = rt.CubemapMaterial()
    sky.faces = [np.ascontiguousarray(strip[:, k * strip.shape[0]:(k + 1) * strip.shape[0]]) for k in range(6)]
r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"], rank=0, world=world).initialize(sky)
for _ in range(frames):
    r.render()
st = r.stats()
print(json.dumps({"lib": os.environ.get("RT355_LIB"), "config": name, "kernel_ms": st["kernel_ms"], "counter": st["rays"], "kernel_id": st["kernel_id"]}))
r.close()
