"""The Node host layer (node/*.js + the N-API shim rt355.node) -- the north_star's drop-in shape:
a TypeScript-like host calling HIP through a thin C-ABI addon.  Skipped when no `node` binary
is installed.  The GPU cases render BASELINE configs through Node and compare the frame hash
with the golden hashes of the oracle's frames."""
import hashlib
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import compute_raytracer_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "node", "rt355.node")
pytestmark = pytest.mark.skipif(NODE is None or not os.path.exists(ADDON), reason="node or rt355.node not available")


def run_node(script, *args, check=True):
    return subprocess.run([NODE, "-e", script] + list(args), cwd=ROOT, capture_output=True, text=True, check=check,
                          timeout=300)


def test_addon_loads_and_exports():
    out = run_node("const rt=require('./node/rt355.node');console.log(JSON.stringify([Object.keys(rt).sort(),rt.abiVersion()]))")
    names, ver = json.loads(out.stdout)
    assert ver == 2
    for n in ["create", "destroy", "resize", "writeParams", "writeSpheres", "writeCubemapFace", "selectKernel",
              "setMode", "setPartition", "render", "wait", "readPixels", "stats"]:
        assert n in names


def test_addon_argument_checks():
    out = run_node("""
const rt=require('./node/rt355.node'); const r=[];
for (const f of [()=>rt.resize(1,2,3), ()=>rt.writeParams({}, new Float32Array(24)), ()=>rt.create('x')]) {
  try { f(); r.push('no throw'); } catch (e) { r.push(e.constructor.name); } }
console.log(JSON.stringify(r));""")
    assert json.loads(out.stdout) == ["TypeError", "TypeError", "TypeError"]


def test_js_scene_generator_and_camera_match_python():
    out = run_node("""
const s=require('./node/scene-raytracing'); const {Camera}=require('./node/camera');
const sp=s.syntheticSpheres(64,357); const a=new Float32Array(8*sp.length);
for (let i=0;i<sp.length;++i){a.set(sp[i].center,8*i);a.set(sp[i].color,8*i+4);a[8*i+7]=sp[i].radius;}
const c=new Camera([0.0593,2.692,3.293],106,270); c.spin(7,-3); c.move(0.25,-0.1);
const p=new Float32Array(12); p.set(c.position,0); p.set(c.forwards,3); p.set(c.right,6); p.set(c.up,9);
console.log(JSON.stringify([Array.from(new Uint32Array(a.buffer)), Array.from(new Uint32Array(p.buffer))]));""")
    sph_bits, cam_bits = json.loads(out.stdout)
    scene = rt.synthetic_scene(64, 357)
    assert sph_bits == [int(v) for v in scene.pack_spheres().reshape(-1).view(np.uint32)]
    cam = scene.camera
    cam.spin(7, -3); cam.move(0.25, -0.1)
    want = np.concatenate([np.array(cam.position).astype(np.float32), cam.forwards, cam.right, cam.up])
    assert cam_bits == [int(v) for v in want.view(np.uint32)]


def test_no_cpu_fallback_in_node():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    out = run_node("const rt=require('./node/rt355.node');try{rt.create(0);console.log('created')}catch(e){console.log(e.code+'|'+e.message)}")
    assert out.stdout.startswith("-2|") and "no CPU path" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode", [("C1", "fast"), ("C1", "strict"), ("C2", "fast")])
def test_node_renders_baseline_config(tmp_path, name, mode):
    fr = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))[name]
    raw = str(tmp_path / "frame.rgba")
    out = subprocess.run([NODE, os.path.join(ROOT, "node", "app.js"), name, raw, "2", mode], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["sha256"] == fr["sha256"] and res["rays"] == fr["rays"] and res["frames"] == 2
    data = open(raw, "rb").read()
    assert len(data) == fr["width"] * fr["height"] * 4 and hashlib.sha256(data).hexdigest() == fr["sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("heatmap", [False, True])
def test_node_renders_triangle_scene(tmp_path, oracle, heatmap):
    """A triangle scene handed to the JS RendererRaytracing as an object of the reference's
    SceneRaytracing shape: packing in JS (RR:169-229) -> N-API -> C ABI -> HIP; frame hash against
    the oracle's frame for the same buffers."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import tri_buffers, triangle_scene
    from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
    scene, mat = triangle_scene(seed=17, n_models=2, rings=5, sectors=6)
    W, H, B = 160, 96, 3
    f = lambda a: [float(v) for v in a]
    js = {
        "camera": {"position": f(scene.camera.position), "forwards": f(scene.camera.forwards),
                   "right": f(scene.camera.right), "up": f(scene.camera.up)},
        "light": {"position": f(scene.light.position), "lightIntensity": scene.light.lightIntensity,
                  "minIntensity": scene.light.minIntensity},
        "triangles": [{"corners": [f(c) for c in t.corners], "normals": [f(c) for c in t.normals],
                       "textures": [f(c) for c in t.textures], "color": f(t.color)} for t in scene.triangles],
        "nodes": [{"minCorner": f(n.minCorner), "maxCorner": f(n.maxCorner), "leftChildIndex": n.leftChildIndex,
                   "primitiveCount": n.primitiveCount} if n is not None else
                  {"minCorner": [0, 0, 0], "maxCorner": [0, 0, 0], "leftChildIndex": 0, "primitiveCount": 0}
                  for n in scene.nodes],
        "blasList": [{"inverseModel": f(b.inverseModel), "rootNodeIndex": b.rootNodeIndex} for b in scene.blasList],
        "blasIndices": list(scene.blasIndices), "triangleIndices": list(scene.triangleIndices),
        "tlasNodesUsed": scene.tlasNodesUsed, "tlasNodesMax": scene.tlasNodesMax, "blasNodesUsed": scene.blasNodesUsed,
    }
    doc = {"width": W, "height": H, "bounces": B, "scene": js,
           "meshTexture": {"width": mat.image.shape[1], "height": mat.image.shape[0], "data": mat.image.reshape(-1).tolist()}}
    path = str(tmp_path / "scene.json")
    json.dump(doc, open(path, "w"))
    out = subprocess.run([NODE, os.path.join(ROOT, "node", "render-json-scene.js"), path, str(tmp_path / "o.rgba")] +
                         (["heatmap"] if heatmap else []), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    b = tri_buffers(scene, mat)
    if heatmap:
        ref, _ = oracle.heatmap_tri(scene.pack_params(B), b, W, H)
    else:
        sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
        ref, _, rays = oracle.render_tri(scene.pack_params(B), b, sky.faces, W, H)
        assert res["rays"] == rays
    assert res["sha256"] == hashlib.sha256(ref.tobytes()).hexdigest()


def _obj_spec():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import obj_floor, obj_uv_sphere
    meshes = [
        dict(obj=obj_uv_sphere(6, 8, 1.0), descriptor=dict(color=[0.9, 0.5, 0.3, 0.6], alignBottom=True, scale=1.0)),
        dict(obj=obj_uv_sphere(5, 7, 1.0, quads=False), descriptor=dict(color=[0.3, 0.7, 0.9, 1.0], alignBottom=True, invertYZ=False, scale=0.7)),
        dict(obj=obj_floor(1.0), descriptor=dict(color=[1.0, 1.0, 1.0, 0.8], scale=12)),
    ]
    models = [dict(meshIndex=0, position=[-2.0, 0, -6.0], eulers=[0, 30.0, 0], eulerSpeed=[0, 45.0, 0]),
              dict(meshIndex=1, position=[2.5, 0, -7.5], eulers=[0, 200.0, 0], eulerSpeed=[0, -20.0, 0]),
              dict(meshIndex=2, position=[0, 0, -5.0], eulers=[0, 0, 0], eulerSpeed=[0, 0, 0])]
    return meshes, models


def _python_scene(meshes, models, updates):
    pm = [rt.Mesh().initializeFromText(m["obj"], m["descriptor"]) for m in meshes]
    pmod = [rt.Model(m["meshIndex"], m["position"], m["eulers"], m["eulerSpeed"]) for m in models]
    scene = rt.SceneRaytracing().createScene([])
    scene.createTriangleScene(pm, pmod)
    for dt in updates:
        scene.update(dt)
    return scene


def test_js_scene_builders_match_python_mirror(tmp_path):
    """OBJ reader -> SAH BVH -> Model matrices -> BLAS -> TLAS built twice, by node/*.js and by the
    Python mirror: every packed f32 must agree bit for bit (two restatements of scene-raytracing.ts,
    bvh.ts, blas.ts, model.ts, obj-reader.ts and of gl-matrix)."""
    meshes, models = _obj_spec()
    updates = [0.016, 0.25, 1.5]
    spec = dict(width=64, height=48, bounces=2, meshes=meshes, models=models, updates=updates)
    path = str(tmp_path / "spec.json")
    json.dump(spec, open(path, "w"))
    out = subprocess.run([NODE, os.path.join(ROOT, "node", "build-obj-scene.js"), path], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    js = json.loads(out.stdout.strip().splitlines()[-1])
    scene = _python_scene(meshes, models, updates)
    assert js["nTriangles"] == len(scene.triangles) and js["tlasNodesUsed"] == scene.tlasNodesUsed
    assert js["tlasNodesMax"] == scene.tlasNodesMax and js["blasNodesUsed"] == scene.blasNodesUsed
    assert js["blasIndices"] == list(scene.blasIndices) and js["triangleIndices"] == list(scene.triangleIndices)
    pb = scene.pack_blas()
    for i, b in enumerate(js["blas"]):
        assert b == [int(v) for v in pb[i, :17].view(np.uint32)], i
    nodes = np.zeros((len(scene.nodes), 8), np.float32)
    t, bn = scene.pack_tlas_nodes(), scene.pack_blas_nodes()
    # JS dumps every node slot as it stands; compare the used ones
    for i in range(scene.tlasNodesUsed):
        assert js["nodes"][i] == [int(v) for v in t[i].view(np.uint32)], i
    for i in range(scene.blasNodesUsed):
        assert js["nodes"][scene.tlasNodesMax + i] == [int(v) for v in bn[i].view(np.uint32)], i
    t0 = scene.triangles[0]
    want = np.concatenate([np.array(t0.corners[0]), np.array(t0.corners[1]), np.array(t0.corners[2]),
                           np.array(t0.centroid, dtype=np.float64)]).astype(np.float32)
    assert js["tri0"] == [int(v) for v in want.view(np.uint32)]


@pytest.mark.gpu
def test_node_builds_and_renders_obj_scene(tmp_path, oracle):
    """The whole reference flow in Node: OBJ text -> scene build (JS) -> RendererRaytracing (JS) ->
    N-API -> HIP; the frame is checked against the oracle fed with the Python mirror's buffers."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import tri_buffers
    from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
    meshes, models = _obj_spec()
    rng = np.random.default_rng(3)
    tex = rng.integers(0, 256, (8, 8, 4), dtype=np.uint8)
    W, H, B = 200, 120, 3
    spec = dict(width=W, height=H, bounces=B, meshes=meshes, models=models, updates=[0.5],
                meshTexture=dict(width=8, height=8, data=tex.reshape(-1).tolist()))
    path = str(tmp_path / "spec.json")
    json.dump(spec, open(path, "w"))
    out = subprocess.run([NODE, os.path.join(ROOT, "node", "build-obj-scene.js"), path, str(tmp_path / "o.rgba")],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    js = json.loads(out.stdout.strip().splitlines()[-1])
    scene = _python_scene(meshes, models, [0.5])
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    ref, _, rays = oracle.render_tri(scene.pack_params(B), tri_buffers(scene, rt.Material(tex)), sky.faces, W, H)
    assert js["rays"] == rays and js["sha256"] == hashlib.sha256(ref.tobytes()).hexdigest()
