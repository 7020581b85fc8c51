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
    this.loaded = false;                                             // RR:51
    this.renderTime = null;                                          // the 'render-time' label, RR:468-469
    this.ctx = null;
  }

  async initialize(skybox, meshTexture) {                            // RR:62-68
    this.meshTexture = meshTexture || null;                          // {width, height, data: Uint8Array}
    const rt = loadAddon();
    this.ctx = rt.create(this.device);                               // RR:78-97 setupDevice
    this.skyboxMaterial = skybox || CubemapMaterial.constant(CONSTANT_SKY_RGBA);   // RR:100-101
    this.skyboxMaterial.faces.forEach((f, i) => rt.writeCubemapFace(this.ctx, i, f.width, f.height, f.data));
    rt.setPartition(this.ctx, this.rank, this.world);
    rt.resize(this.ctx, this.width, this.height);                    // RR:102-109 colorBuffer
    this.showRaytracer();                                            // RR:356-365
  }

  showRaytracer() { loadAddon().selectKernel(this.ctx, RT_KERNEL_RAYTRACER); }   // RR:70-72
  showHeatmap() { loadAddon().selectKernel(this.ctx, RT_KERNEL_HEATMAP); }       // RR:74-76
  setStrict(strict) { loadAddon().setMode(this.ctx, strict ? 1 : 0); }

  recalculateScene() {                                               // RR:155-230
    const rt = loadAddon();
    const sceneParametersData = new Float32Array(24);                // RR:158-164
    sceneParametersData.set(this.scene.camera.position, 0);
    sceneParametersData.set(this.scene.camera.forwards, 4);
    sceneParametersData.set(this.scene.camera.right, 8);
    sceneParametersData.set(this.scene.camera.up, 12);
    sceneParametersData.set(this.scene.light.position, 16);
    sceneParametersData.set([this.scene.light.lightIntensity, this.scene.light.minIntensity, this.maxBounces], 19);
    rt.writeParams(this.ctx, sceneParametersData);                   // RR:165
    const tri = this.scene.triangles && this.scene.triangles.length > 0;   // the reference's live scene type
    if (tri) {
      const blasData = new Float32Array(20 * this.scene.blasList.length);           // RR:169-174
      for (let i = 0; i < this.scene.blasList.length; ++i) {
        blasData.set(this.scene.blasList[i].inverseModel, 20 * i);
        blasData.set([this.scene.blasList[i].rootNodeIndex], 20 * i + 16);
      }
      rt.writeBlas(this.ctx, blasData);
      const blasIndexData = new Float32Array(this.scene.blasIndices.length);        // RR:177-181
      for (let i = 0; i < this.scene.blasIndices.length; ++i) blasIndexData[i] = this.scene.blasIndices[i];
      rt.writeBlasLookup(this.ctx, blasIndexData);
      const nodeDataA = new Float32Array(8 * this.scene.tlasNodesUsed);             // RR:184-192
      for (let i = 0; i < this.scene.tlasNodesUsed; ++i) {
        const loc = 8 * i;
        nodeDataA.set(this.scene.nodes[i].minCorner, loc);
        nodeDataA.set(this.scene.nodes[i].maxCorner, loc + 4);
        nodeDataA[loc + 3] = this.scene.nodes[i].leftChildIndex;
        nodeDataA[loc + 7] = this.scene.nodes[i].primitiveCount;
      }
      rt.writeNodes(this.ctx, 0, nodeDataA);
    }
    if (this.loaded) return;                                         // RR:194-195
    this.loaded = true;
    if (tri) {
      const triangleData = new Float32Array(40 * this.scene.triangles.length);      // RR:198-209
      for (let i = 0; i < this.scene.triangles.length; i++) {
        const loc = 40 * i;
        const t = this.scene.triangles[i];
        for (let corner = 0; corner < 3; corner++) {
          triangleData.set(t.corners[corner], loc + 12 * corner);
          triangleData.set(t.normals[corner], loc + 12 * corner + 4);
          triangleData.set(t.textures[corner], loc + 12 * corner + 8);
        }
        triangleData.set(t.color, loc + 36);
      }
      rt.writeTriangles(this.ctx, triangleData);
      const nodeDataB = new Float32Array(8 * this.scene.blasNodesUsed);             // RR:212-223
      for (let i = 0; i < this.scene.blasNodesUsed; ++i) {
        const node = this.scene.nodes[this.scene.tlasNodesMax + i];
        const loc = 8 * i;
        nodeDataB.set(node.minCorner, loc + 0);
        nodeDataB.set([node.leftChildIndex], loc + 3);
        nodeDataB.set(node.maxCorner, loc + 4);
        nodeDataB.set([node.primitiveCount], loc + 7);
      }
      rt.writeNodes(this.ctx, 32 * this.scene.tlasNodesMax, nodeDataB);
      const triangleIndexData = new Float32Array(this.scene.triangleIndices.length);   // RR:225-229
      for (let i = 0; i < this.scene.triangleIndices.length; ++i) triangleIndexData[i] = this.scene.triangleIndices[i];
      rt.writeTriLookup(this.ctx, triangleIndexData);
      if (this.meshTexture) rt.writeMeshTexture(this.ctx, this.meshTexture.width, this.meshTexture.height, this.meshTexture.data);   // RR:113-114
      return;
    }
    const spheres = this.scene.spheres;                              // in place of RR:198-229
    const data = new Float32Array(8 * spheres.length);
    for (let i = 0; i < spheres.length; ++i) {
      data.set(spheres[i].center, 8 * i);                            // struct Sphere RK:13-17: center @0,
      data.set(spheres[i].color, 8 * i + 4);                         // color @16 B,
      data[8 * i + 7] = spheres[i].radius;                           // radius @28 B
    }
    rt.writeSpheres(this.ctx, data);
  }

  async render() {                                                   // RR:434-470
    const rt = loadAddon();
    const t0 = Date.now();                                           // RR:435
    this.recalculateScene();                                         // RR:437
    rt.render(this.ctx);                                             // RR:442-446, 465
    await rt.wait(this.ctx);                                         // RR:467
    this.renderTime = Date.now() - t0;                               // RR:468-469
  }

  readPixels() {
    const rt = loadAddon();
    let rows = 0;
    const tiles = Math.ceil(this.height / 8);
    for (let t = this.rank; t < tiles; t += this.world) rows += Math.min(8, this.height - 8 * t);
    const out = new Uint8Array(rows * this.width * 4);
    rt.readPixels(this.ctx, out);
    return out;
  }

  stats() { return loadAddon().stats(this.ctx); }
  close() { if (this.ctx) { loadAddon().destroy(this.ctx); this.ctx = null; } }
}
module.exports = { RendererRaytracing };
