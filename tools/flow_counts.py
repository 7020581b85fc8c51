"""Counting build of the persistent triangle kernel (tools/build_dev.sh -> tools/bin/librt355_fc.so): what one frame's
waves executed -- trips, runs of each block and the lanes each run advanced (lanes_busy), BNODE steps served from LDS.
usage: RT355_LIB=tools/bin/librt355_fc.so python tools/flow_counts.py [REF|TRI|TRI4K] [out.json]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import compute_raytracer_amd as rt
from compute_raytracer_amd import abi
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
name = sys.argv[1] if len(sys.argv) > 1 else "REF"
if name == "REF":
    from helpers import ref_fixture
    scene, sky, W, H, B, canvas, pin = ref_fixture(); mat = rt.Material.white()
else:
    from compute_raytracer_amd.procedural import triangle_scene
    scene, mat = triangle_scene(seed=21, n_models=2, rings=48, sectors=64)
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    W, H, B = (1344, 846, 4) if name == "TRI" else (3840, 2160, 4)
L = abi.load()
r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
r.render(); r.wait()
buf = (ctypes.c_ulonglong * 16)()
assert L.rt_debug_flow_counts(buf, 1) == 0
r.render(); r.wait()
assert L.rt_debug_flow_counts(buf, 1) == 0
c = list(buf)
names = ["TLAS", "BNODE", "TRI", "DONE"]
res = {"config": name, "rays": r.stats()["rays"], "trips": c[8], "bnode_steps_from_lds": c[9]}
for k, n in enumerate(names):
    res[n] = {"runs": c[2 * k], "lanes": c[2 * k + 1], "lanes_per_run": c[2 * k + 1] / max(c[2 * k], 1)}
steps = sum(res[n]["lanes"] for n in names[:3])
res["walk_lanes_busy"] = steps / (64.0 * max(sum(res[n]["runs"] for n in names[:3]), 1))
res["lds_share_of_bnode"] = c[9] / max(res["BNODE"]["lanes"], 1)
print(json.dumps(res))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
r.close()
