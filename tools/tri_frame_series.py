"""Development probe: kernel time of consecutive awaited frames of a triangle configuration (does the work list, made from the
previous frame's tile times, settle or alternate?).  usage: python tools/tri_frame_series.py REF|TRI|TRI4K [frames=40]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import compute_raytracer_amd as rt
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
name = sys.argv[1] if len(sys.argv) > 1 else "REF"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
if name == "REF":
    from helpers import ref_fixture
    scene, sky, W, H, B, canvas, pin = ref_fixture(); mat = rt.Material.white()
else:
    from compute_raytracer_amd.procedural import triangle_scene
    scene, mat = triangle_scene(seed=21, n_models=2, rings=48, sectors=64)
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    W, H, B = (1344, 846, 4) if name == "TRI" else (3840, 2160, 4)
r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
ms = []
for _ in range(n):
    r.render(); ms.append(r.stats()["kernel_ms"])
print(name, " ".join("%.3f" % v for v in ms))
r.close()
