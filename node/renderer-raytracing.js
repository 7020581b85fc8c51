'use strict';
// RendererRaytracing -- Node-12 CommonJS class with the surface of the reference's
// src/rendering-raycast/renderer-raytracing.ts (RR): constructor(width, height, scene)
// [the canvas argument is dropped], initialize(), render(): Promise<void>, showRaytracer(),
// showHeatmap().  Every WebGPU call of RR becomes one call into the N-API shim rt355.node,
// which forwards to the C ABI of librt355.so (include/rt355.h).  No CPU fallback.
const path = require('path');
const { CubemapMaterial } = require('./cubemap-material');
const { CONSTANT_SKY_RGBA } = require('./scene-raytracing');

let addon = null;
function loadAddon() {
  if (!addon) addon = require(path.join(__dirname, 'rt355.node'));
  return addon;
}

const RT_KERNEL_RAYTRACER = 0, RT_KERNEL_HEATMAP = 1;

class RendererRaytracing {
  constructor(width, height, scene, options) {                       // RR:53-60
    const o = options || {};
    this.scene = scene;
    this.width = width;
    this.height = height;
    this.device = o.device || 0;
    this.maxBounces = o.maxBounces === undefined ? 4 : o.maxBounces; // RR:157 hard-codes 4
    this.rank = o.rank || 0;
    this.world = o.world || 1;
    // devices: undefined -> one GPU (this.device); a number -> that many GPUs of this node driven by this one
    // thread (0 = all visible): the frame is row-tiled over them and gathered over RCCL inside librt355.so
    this.devices = o.devices;
    this.group = null;
    this.members = null;
    this.loaded = false;                                             // RR:51
    this.renderTime = null;                                          // the 'render-time' label, RR:468-469
    this.ctx = null;
  }

  async initialize(skybox, meshTexture) {                            // RR:62-68
    this.meshTexture = meshTexture || null;                          // {width, height, data: Uint8Array}
    const rt = loadAddon();
    this.skyboxMaterial = skybox || CubemapMaterial.constant(CONSTANT_SKY_RGBA);   // RR:100-101
    if (this.devices !== undefined) {                                // RR:78-97 setupDevice, once per GPU
      this.group = rt.createGroup(this.devices);
      this.members = [];
      for (let i = 0; i < rt.groupSize(this.group); ++i) this.members.push(rt.groupCtx(this.group, i));
      this.ctx = this.members[0];                                    // the root: it receives the frame
    } else {
      this.ctx = rt.create(this.device);
      this.members = [this.ctx];
      rt.setPartition(this.ctx, this.rank, this.world);
    }
    for (const c of this.members) {
      this.skyboxMaterial.faces.forEach((f, i) => rt.writeCubemapFace(c, i, f.width, f.height, f.data));
      rt.resize(c, this.width, this.height);                         // RR:102-109 colorBuffer
    }
    this.showRaytracer();                                            // RR:356-365
  }

  showRaytracer() { for (const c of this.members) loadAddon().selectKernel(c, RT_KERNEL_RAYTRACER); }   // RR:70-72
  showHeatmap() { for (const c of this.members) loadAddon().selectKernel(c, RT_KERNEL_HEATMAP); }       // RR:74-76
  setStrict(strict) { for (const c of this.members) loadAddon().setMode(c, strict ? 1 : 0); }

  // RR:155-230.  The scene hands over its buffers already in the layouts RR writes (scene-raytracing.js):
  // `frame` (instance records, instance lookup, top-level nodes) every frame, `packed` (triangles,
  // bottom-level nodes, triangle lookup) once -- the `loaded` split of RR:194-195.
  recalculateScene() {
    const rt = loadAddon();
    const scene = this.scene;
    const params = new Float32Array(24);                             // RR:157-164
    params.set(scene.camera.position, 0);
    params.set(scene.camera.forwards, 4);
    params.set(scene.camera.right, 8);
    params.set(scene.camera.up, 12);
    params.set(scene.light.position, 16);
    params[19] = scene.light.lightIntensity;
    params[20] = scene.light.minIntensity;
    params[21] = this.maxBounces;
    for (const c of this.members) {                                  // the scene is replicated on every GPU
      rt.writeParams(c, params);                                     // RR:165
      if (scene.hasTriangles) {                                      // the reference's live scene type
        rt.writeBlas(c, scene.frame.blasData);                       // RR:169-174
        rt.writeBlasLookup(c, scene.frame.blasIndexData);            // RR:177-181
        rt.writeNodes(c, 0, scene.frame.nodeDataA);                  // RR:184-192
      }
    }
    if (this.loaded) return;
    this.loaded = true;
    if (scene.hasTriangles) {
      for (const c of this.members) {
        rt.writeTriangles(c, scene.packed.triangleData);                       // RR:198-209
        rt.writeNodes(c, 32 * scene.tlasNodesMax, scene.packed.nodeDataB);     // RR:212-223
        rt.writeTriLookup(c, scene.packed.triangleIndexData);                  // RR:225-229
        if (this.meshTexture) rt.writeMeshTexture(c, this.meshTexture.width, this.meshTexture.height, this.meshTexture.data);   // RR:113-114
      }
      return;
    }
    const spheres = scene.spheres;                                   // in place of RR:198-229
    const data = new Float32Array(8 * spheres.length);
    for (let i = 0; i < spheres.length; ++i) {
      data.set(spheres[i].center, 8 * i);                            // struct Sphere RK:13-17: center @0,
      data.set(spheres[i].color, 8 * i + 4);                         // color @16 B,
      data[8 * i + 7] = spheres[i].radius;                           // radius @28 B
    }
    for (const c of this.members) rt.writeSpheres(c, data);
  }

  async render() {                                                   // RR:434-470
    const rt = loadAddon();
    const t0 = Date.now();                                           // RR:435
    this.recalculateScene();                                         // RR:437
    if (this.group) {
      rt.groupRender(this.group, 0);                                 // RR:442-446, 465 on every GPU + RCCL gather to GPU 0
      await rt.groupWait(this.group);                                // RR:467
    } else {
      rt.render(this.ctx);                                           // RR:442-446, 465
      await rt.wait(this.ctx);                                       // RR:467
    }
    this.renderTime = Date.now() - t0;                               // RR:468-469
  }

  readPixels() {
    const rt = loadAddon();
    if (this.group) {                                                // the gathered, de-interleaved frame
      const frame = new Uint8Array(this.width * this.height * 4);
      rt.readFrame(this.ctx, frame);
      return frame;
    }
    let rows = 0;
    const tiles = Math.ceil(this.height / 8);
    for (let t = this.rank; t < tiles; t += this.world) rows += Math.min(8, this.height - 8 * t);
    const out = new Uint8Array(rows * this.width * 4);
    rt.readPixels(this.ctx, out);
    return out;
  }

  stats() {                                                          // rays: summed over the GPUs of a group
    const rt = loadAddon();
    const st = rt.stats(this.ctx);
    for (let i = 1; this.group && i < this.members.length; ++i) st.rays += rt.stats(this.members[i]).rays;
    return st;
  }
  close() {
    const rt = loadAddon();
    if (this.group) { rt.destroyGroup(this.group); this.group = null; }
    else if (this.ctx) rt.destroy(this.ctx);
    this.ctx = null; this.members = [];
  }
}
module.exports = { RendererRaytracing };
