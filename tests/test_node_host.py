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
    assert ver == 4
    for n in ["create", "destroy", "resize", "writeParams", "writeSpheres", "writeCubemapFace", "selectKernel",
              "setMode", "setPartition", "render", "wait", "readPixels", "stats", "readFrame", "createGroup", "destroyGroup",
              "groupSize", "groupCtx", "groupRender", "groupWait", "commUniqueId", "commInit", "renderGather", "hostAlloc",
              "readPixelsAsync", "readPixelsWait", "setCommTimeout", "buildId", "kernelName"]:
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
def test_addon_refuses_use_after_destroy_and_calls_during_an_async_wait():
    """wait() runs rt_wait on a libuv worker while the JS thread stays free -- any call on the
    same context before the promise settles must throw (rt_ctx is single-threaded), and a destroyed
    context must be an error, not a use-after-free.  stats() carries the batch figures."""
    out = run_node("""
const rt=require('./node/rt355.node'); const r=[];
(async () => {
  const c=rt.create(0); rt.resize(c,64,64);
  const p=new Float32Array(24); p[6]=-1; p[8]=1; p[13]=1; p[17]=5; p[19]=3; p[20]=0.3; p[21]=2;
  rt.writeParams(c,p); rt.writeSpheres(c,new Float32Array([0,0,-5,0, 1,0,0,1]));
  for (let f=0;f<6;++f) rt.writeCubemapFace(c,f,1,1,new Uint8Array([1,2,3,255]));
  rt.render(c); rt.render(c);
  const w=rt.wait(c);
  for (const f of [()=>rt.render(c), ()=>rt.stats(c), ()=>rt.destroy(c), ()=>rt.wait(c)]) { try { f(); r.push('no throw'); } catch (e) { r.push(e.code); } }
  await w;
  const st=rt.stats(c); r.push(st.batchFrames, st.batchKernelMs>0, st.gatherMs);
  rt.destroy(c);
  for (const f of [()=>rt.destroy(c), ()=>rt.render(c), ()=>rt.stats(c)]) { try { f(); r.push('no throw'); } catch (e) { r.push(e.code); } }
  console.log(JSON.stringify(r));
})();""")
    assert json.loads(out.stdout) == ["-5", "-5", "-5", "-5", 2, True, 0, "-5", "-5", "-5"]


@pytest.mark.gpu
def test_node_group_renders_c2_over_every_visible_gpu(tmp_path):
    """The Node host driving the multi-GPU path the way the reference's single JS thread would: one
    RendererRaytracing with {devices: 0} = rt_group_create over every visible GPU, groupRender (render +
    RCCL gather inside librt355.so), groupWait as a promise, readFrame.  Frame = the oracle's C2 frame."""
    fr = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["C2"]
    raw = str(tmp_path / "frame.rgba")
    out = subprocess.run([NODE, os.path.join(ROOT, "node", "app.js"), "C2", raw, "3", "fast", "group"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["sha256"] == fr["sha256"] and res["rays"] == fr["rays"] and res["frames"] == 3
    assert hashlib.sha256(open(raw, "rb").read()).hexdigest() == fr["sha256"]


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
    """A triangle scene handed to the JS RendererRaytracing as the upload buffers themselves (the
    layouts of RR:169-229, here produced by the Python mirror): JS -> N-API -> C ABI -> HIP; frame hash
    against the oracle's frame for the same buffers."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import tri_buffers, triangle_scene
    from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
    scene, mat = triangle_scene(seed=17, n_models=2, rings=5, sectors=6)
    W, H, B = 160, 96, 3
    f = lambda a: [float(v) for v in np.asarray(a).reshape(-1)]
    doc = {"width": W, "height": H, "bounces": B, "tlasNodesMax": scene.tlasNodesMax,
           "camera": {"position": f(scene.camera.position), "forwards": f(scene.camera.forwards),
                      "right": f(scene.camera.right), "up": f(scene.camera.up)},
           "light": {"position": f(scene.light.position), "lightIntensity": scene.light.lightIntensity,
                     "minIntensity": scene.light.minIntensity},
           "packed": {"triangleData": f(scene.pack_triangles()), "nodeDataB": f(scene.pack_blas_nodes()),
                      "triangleIndexData": f(scene.pack_tri_lookup())},
           "frame": {"blasData": f(scene.pack_blas()), "blasIndexData": f(scene.pack_blas_lookup()),
                     "nodeDataA": f(scene.pack_tlas_nodes())},
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
    pm = [rt.load_mesh(m["obj"], m["descriptor"]) for m in meshes]
    scene = rt.SceneRaytracing().createScene([])
    scene.createTriangleScene(pm, models)
    for dt in updates:
        scene.update(dt)
    return scene


def _bits(a):
    return [int(v) for v in np.ascontiguousarray(a, dtype=np.float32).reshape(-1).view(np.uint32)]


def _compare_js_with_python(js, scene):
    assert js["nTriangles"] == scene.triangleCount and js["tlasNodesUsed"] == scene.tlasNodesUsed
    assert js["tlasNodesMax"] == scene.tlasNodesMax and js["blasNodesUsed"] == scene.blasNodesUsed
    assert js["blas"] == _bits(scene.pack_blas())
    assert js["blasIndices"] == _bits(scene.pack_blas_lookup())
    assert js["tlasNodes"] == _bits(scene.pack_tlas_nodes())
    assert js["blasNodes"] == _bits(scene.pack_blas_nodes())
    assert js["triangleIndices"] == _bits(scene.pack_tri_lookup())
    assert js["triangles"] == _bits(scene.pack_triangles())
    assert js["centroids0"] == _bits(scene.meshes[0].soup.centroid)


def test_js_scene_builders_match_python_mirror(tmp_path):
    """OBJ text -> triangle soup -> SAH tree -> instance matrices -> top-level tree, built twice: by the
    data-oriented node/*.js (typed arrays, scalar loops, explicit stacks) and by the numpy host (whole-array arithmetic).  Every
    f32 of every upload buffer must agree bit for bit (two independent restatements of what
    scene-raytracing.ts, bvh.ts, blas.ts, model.ts, obj-reader.ts and gl-matrix compute)."""
    meshes, models = _obj_spec()
    updates = [0.016, 0.25, 1.5]
    spec = dict(width=64, height=48, bounces=2, meshes=meshes, models=models, updates=updates)
    path = str(tmp_path / "spec.json")
    json.dump(spec, open(path, "w"))
    out = subprocess.run([NODE, os.path.join(ROOT, "node", "build-obj-scene.js"), path], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    _compare_js_with_python(json.loads(out.stdout.strip().splitlines()[-1]), _python_scene(meshes, models, updates))


def test_js_scene_builders_swizzled_mesh_many_updates(tmp_path):
    """A y/z-swizzled, off-centre, scaled mesh (the loader's centring quirks), an instance that spins
    through the +-360 degree wrap, quads and triangles mixed."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import obj_floor, obj_uv_sphere
    meshes = [
        dict(obj=obj_uv_sphere(9, 11, 2.5, centre=(0.3, 1.0, -0.2)), descriptor=dict(color=[0.2, 0.9, 0.4, 0.5], alignBottom=False, invertYZ=True, scale=0.31)),
        dict(obj=obj_uv_sphere(4, 5, 0.8, centre=(-3.0, 0.25, 7.0), quads=False), descriptor=dict(color=[0.9, 0.9, 0.1, 1.0], alignBottom=True, invertYZ=True, scale=1.7)),
        dict(obj=obj_floor(2.0), descriptor=dict(color=[1.0, 1.0, 1.0, 0.8], scale=5)),
    ]
    models = [dict(meshIndex=0, position=[0.5, 0.2, -9.5], eulers=[10, 350.0, 0], eulerSpeed=[0, 700.0, 0]),
              dict(meshIndex=1, position=[-2.0, 0, -6.0], eulers=[0, -355.0, 0], eulerSpeed=[0, -333.0, 0]),
              dict(meshIndex=1, position=[2.0, 1, -4.0], eulers=[0, 12.0, 0]),
              dict(meshIndex=2, position=[0, 0, -5.0], eulers=[0, 0, 0], eulerSpeed=[0, 0, 0])]
    updates = [0.016, 0.25, 1.5, 0.7, 0.033]
    spec = dict(width=64, height=48, bounces=2, meshes=meshes, models=models, updates=updates)
    path = str(tmp_path / "spec.json")
    json.dump(spec, open(path, "w"))
    out = subprocess.run([NODE, os.path.join(ROOT, "node", "build-obj-scene.js"), path], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    _compare_js_with_python(json.loads(out.stdout.strip().splitlines()[-1]), _python_scene(meshes, models, updates))


def test_node_host_files_are_not_the_reference_typescript():
    """The Node host layer is written from the buffer layouts, not from the reference's sources: no file
    named after a reference class survives, and the builders hold no per-node / per-triangle objects."""
    names = set(os.listdir(os.path.join(ROOT, "node")))
    assert not ({"acceleration", "model", "mesh.js"} & names)
    for f in ("sah.js", "soup.js", "scene-raytracing.js"):
        text = open(os.path.join(ROOT, "node", f)).read()
        assert "class Node" not in text and "class BLAS" not in text and "class Triangle " not in text
        assert "Float64Array" in text or "Float32Array" in text


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


@pytest.mark.gpu
def test_node_streaming_readback(oracle):
    """The animation loop of src/app.ts:117-128 with frames in flight AND every frame on the host, from Node: render without
    awaiting, copy the frame two renders back into a pinned Uint8Array (hostAlloc / readPixelsAsync), wait at the end; each of
    the eight frames (the camera walks) must be the oracle's."""
    W, H, B, K = 200, 120, 3, 8
    out = run_node("""
const rt=require('./node/rt355.node'); const crypto=require('crypto');
const s=require('./node/scene-raytracing'); const {RendererRaytracing}=require('./node/renderer-raytracing');
(async()=>{
  const scene=new s.SceneRaytracing(); await scene.createScene(s.syntheticSpheres(300,361));
  const r=new RendererRaytracing(%d,%d,scene,{maxBounces:%d}); await r.initialize();
  const host=[]; for(let i=0;i<%d;++i) host.push(rt.hostAlloc(%d));
  for(let f=0;f<%d;++f){ scene.camera.move(0.07,-0.03); r.recalculateScene(); rt.render(r.ctx); if(f>=2) rt.readPixelsAsync(r.ctx,2,host[f-2]); }
  rt.readPixelsAsync(r.ctx,1,host[%d-2]); rt.readPixelsAsync(r.ctx,0,host[%d-1]);
  await rt.wait(r.ctx); rt.readPixelsWait(r.ctx);
  console.log(JSON.stringify({sha: host.map(h=>crypto.createHash('sha256').update(h).digest('hex')), kernel: rt.kernelName(rt.stats(r.ctx).kernelId), build: rt.buildId()}));
  r.close();
})().catch(e=>{console.error(e);process.exit(1)});""" % (W, H, B, K, W * H * 4, K, K, K))
    got = json.loads(out.stdout.strip().splitlines()[-1])
    from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
    scene = rt.synthetic_scene(300, 361)
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    want = []
    for f in range(K):
        scene.camera.move(0.07, -0.03)
        want.append(hashlib.sha256(oracle.render(scene.pack_params(B), scene.pack_spheres(), sky.faces, W, H)[0].tobytes()).hexdigest())
    assert got["sha"] == want and got["kernel"] == "bvh_pixels<8>" and len(got["build"]) == 16


@pytest.mark.gpu
def test_animation_loop_bench_in_node():
    """tools/node_loop_bench.py: the reference's animation loop (scene.update, camera.move, await render) in the reference's
    host language against the static loop -- a few frames at a small size: the script runs, reports both, and the JS scene
    update is a small part of a frame."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "node_loop_bench.py"), "6", "320", "200"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["triangles"] == 12846 and d["frames"] == 6 and d["rays"] > 320 * 200
    assert 0 < d["staticLoopMsPerFrame"] < 50 and 0 < d["animatedLoopMsPerFrame"] < 50
    assert d["hostSceneUpdateMsPerFrame"] < 0.5
