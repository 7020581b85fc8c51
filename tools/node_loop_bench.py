"""The drop-in's loop in the reference's host language: writes the spec of the procedural triangle scene
of bench.py --config TRI (two UV spheres of 48 x 64 rings / sectors + a floor, 12,846 triangles) and runs node/bench-loop.js on
it: static loop against animated loop, one frame at a time.  usage: python tools/node_loop_bench.py [frames] [width height]"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from compute_raytracer_amd.procedural import obj_floor, obj_uv_sphere
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1344
H = int(sys.argv[3]) if len(sys.argv) > 3 else 846
rng = np.random.default_rng(21)                         # the models of procedural.triangle_scene(seed=21, n_models=2)
rings, sectors = 48, 64
meshes = [dict(obj=obj_uv_sphere(rings, sectors, 1.0), descriptor=dict(color=[0.9, 0.5, 0.3, 0.6], alignBottom=True, invertYZ=False, scale=1.0)),
          dict(obj=obj_uv_sphere(rings + 2, sectors + 3, 1.0, quads=False), descriptor=dict(color=[0.3, 0.7, 0.9, 1.0], alignBottom=True, invertYZ=False, scale=0.7)),
          dict(obj=obj_floor(1.0), descriptor=dict(color=[1.0, 1.0, 1.0, 0.8], alignBottom=False, invertYZ=False, scale=12))]
models = []
for i in range(2):
    pos = [float(rng.uniform(-4, 4)), 0.0, float(rng.uniform(-9, -3))]
    models.append(dict(meshIndex=i % 2, position=pos, eulers=[0, float(rng.uniform(0, 360)), 0], eulerSpeed=[0, float(rng.uniform(-90, 90)), 0]))
models.append(dict(meshIndex=2, position=[0, 0, -5], eulers=[0, 0, 0], eulerSpeed=[0, 0, 0]))
spec = dict(width=W, height=H, bounces=4, meshes=meshes, models=models)
with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
    json.dump(spec, f)
out = subprocess.run(["node", os.path.join(ROOT, "node", "bench-loop.js"), f.name, str(frames)], capture_output=True, text=True)
os.unlink(f.name)
sys.stdout.write(out.stdout)
sys.stderr.write(out.stderr[-2000:])
sys.exit(out.returncode)
