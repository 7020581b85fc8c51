"""N frames one at a time of a triangle configuration (for rocprofv3 passes): python3 tools/tri_frames.py REF|TRI|TRI4K [frames] [variant]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import compute_raytracer_amd as rt
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
name = sys.argv[1] if len(sys.argv) > 1 else "REF"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if name == "REF":
    from helpers import ref_fixture
    scene, sky, W, H, B, canvas, pin = ref_fixture(); mat = rt.Material.white()
else:
    from compute_raytracer_amd.procedural import triangle_scene
    scene, mat = triangle_scene(seed=21, n_models=2, rings=48, sectors=64)
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    W, H, B = (1344, 846, 4) if name == "TRI" else (3840, 2160, 4)
r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
r.set_variant(variant)
for _ in range(n):
    r.render()
print(r.stats()["kernel_ms"], r.stats()["kernel_id"])
r.close()
