"""Development probe: how long the tiles of a triangle frame take (the triangle kernel's own per-tile clock, rt_triangles.hip),
against the time of the whole frame rendered on its own -- a frame cannot end before its longest tile does.
Needs a library built with -DRT355_DEV_EXPORTS:  python tools/tile_cost_probe.py --build   (build container)
                                                 python tools/tile_cost_probe.py REF|TRI|TRI4K   (GPU box)"""
import ctypes, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = os.path.join(ROOT, "tools", "bin", "librt355_dev.so")
if "--build" in sys.argv:
    csrc = os.path.join(ROOT, "compute_raytracer_amd", "csrc")
    flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-function -ffp-contract=off".split()
    os.makedirs(os.path.dirname(DEV), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-DRT355_DEV_EXPORTS", "-DRT355_BUILD_ID=\"dev\"", "-c", os.path.join(csrc, "rt_api.hip"), "-o", "/tmp/rt_api_dev.o"], check=True)
    # the same development library serves tools/knob_ab.py: rt_bvh.hip with its environment knobs (RT355_BVH_TAIL, _BLOCKS, _LDS_PAD)
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-fno-slp-vectorize", "-DRT_BVH_DEV_ENV", "-c", os.path.join(csrc, "rt_bvh.hip"), "-o", "/tmp/rt_bvh_dev.o"], check=True)
    objs = [os.path.join(csrc, o) for o in ("rt_kernels.o", "rt_triangles.o", "rt_assemble.o", "rt_comm.o")] + ["/tmp/rt_bvh_dev.o"]
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", DEV, "/tmp/rt_api_dev.o"] + objs + ["-L/opt/rocm/lib", "-lrccl"], check=True)
    print("built", DEV)
    sys.exit(0)
os.environ["RT355_LIB"] = DEV
os.environ["RT355_KEEP_TILE_COST"] = "1"      # whole tiles in index order, their times kept (no work list)
sys.path.insert(0, ROOT)
import numpy as np
import compute_raytracer_amd as rt
from compute_raytracer_amd import abi
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "TRI"
cfg = bench.TRI_CONFIGS[name]
W, H, B = cfg["width"], cfg["height"], cfg["bounces"]
if cfg.get("fixture"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import ref_fixture
    scene, sky, W, H, B = ref_fixture()[:5]
    mat = rt.Material.white()
else:
    from compute_raytracer_amd.procedural import triangle_scene
    from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
    scene, mat = triangle_scene(seed=21, n_models=2, rings=48, sectors=64)
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
lib = abi.load()
lib.rt_debug_tile_cost.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32]
n = ((W + 7) // 8) * ((H + 7) // 8)
ms = []
for f in range(12):                     # one at a time: every stream comes round three times, the order settles
    t0 = time.perf_counter(); r.render(); ms.append((time.perf_counter() - t0) * 1e3)
kms = r.stats()["kernel_ms"]
cost = np.zeros(n, np.uint32)
got = lib.rt_debug_tile_cost(r._ctx, 11 % 4, cost.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n)
us = cost[:got].astype(np.float64) * 0.01
q = np.percentile(us, [50, 90, 99, 99.9])
print(json.dumps({"config": name, "tiles": int(got), "render_wait_ms_median": float(np.median(ms[4:])), "kernel_ms_last": kms,
                  "tile_us": {"mean": float(us.mean()), "p50": q[0], "p90": q[1], "p99": q[2], "p99.9": q[3], "max": float(us.max())},
                  "wave_slots": 4096, "sum_tile_ms_over_slots": float(us.sum() / 4096.0 / 1e3)}))
r.close()
