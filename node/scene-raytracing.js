'use strict';
// SceneRaytracing for sphere scenes -- the counterpart of src/rendering-raycast/scene-raytracing.ts
// (`spheres: Sphere[]` is declared there, line 18, but never filled).  Same deterministic
// generator as compute_raytracer_amd/scene_raytracing.py (SURVEY.md 8(d)); BigInt splitmix64.
const { Camera } = require('./camera');
const { Sphere } = require('./sphere');
const { vec3 } = require('./gl-matrix-lite');
const { Node } = require('./acceleration/node');
const { BLAS } = require('./acceleration/blas');

const M64 = (1n << 64n) - 1n;
const BASELINE_CONFIGS = {
  C1: { width: 256, height: 256, spheres: 3, bounces: 1, seed: 356 },
  C2: { width: 1920, height: 1080, spheres: 64, bounces: 4, seed: 357 },
  C3: { width: 3840, height: 2160, spheres: 1024, bounces: 8, seed: 358 },
  C4: { width: 3840, height: 2160, spheres: 1024, bounces: 8, seed: 358 },
};
const CONSTANT_SKY_RGBA = [128, 179, 255, 255];

class SplitMix64 {
  constructor(seed) { this.state = BigInt(seed) & M64; }
  nextU64() {
    this.state = (this.state + 0x9E3779B97F4A7C15n) & M64;
    let z = this.state;
    z = ((z ^ (z >> 30n)) * 0xBF58476D1CE4E5B9n) & M64;
    z = ((z ^ (z >> 27n)) * 0x94D049BB133111EBn) & M64;
    return z ^ (z >> 31n);
  }
  uniform() { return Number(this.nextU64() >> 40n) / 16777216.0; }
  range(lo, hi) { return lo + (hi - lo) * this.uniform(); }
}

function syntheticSpheres(n, seed) {
  if (n < 1) return [];
  const rng = new SplitMix64(seed);
  const spheres = [new Sphere([0.0, -100.0, 0.0], 100.0, [0.8, 0.8, 0.8])];
  const rscale = Math.fround(Math.pow(64.0 / n, 1.0 / 3.0));
  for (let i = 0; i < n - 1; ++i) {
    const x = rng.range(-12.0, 12.0);
    const z = rng.range(-26.0, -3.0);
    let r = rng.range(0.5, 1.5) * rscale;
    r = Math.min(Math.max(r, 0.04), 1.5);
    const y = r + rng.range(0.0, 3.0);
    const col = [rng.range(0.2, 1.0), rng.range(0.2, 1.0), rng.range(0.2, 1.0)];
    spheres.push(new Sphere([x, y, z], r, col));
  }
  return spheres;
}

class SceneRaytracing {
  constructor() {
    this.camera = null; this.light = null; this.spheres = [];
    // triangle-scene members, scene-raytracing.ts:19-35
    this.meshes = []; this.models = []; this.triangles = []; this.triangleIndices = [];
    this.nodes = []; this.blasList = []; this.blasIndices = [];
    this.tlasNodesMax = 0; this.tlasNodesUsed = 0; this.blasNodesUsed = 0; this.blasConsumed = false;
  }
  async createScene(spheres) {                                       // scene-raytracing.ts:37-45
    this.camera = new Camera([0.0593, 2.692, 3.293], 106, 270);
    this.light = { position: [0, 5, 0], lightIntensity: 3.0, minIntensity: 0.3 };
    this.spheres = spheres || [];
    return this;
  }
  update(dt) {                                                       // scene-raytracing.ts:138-143
    if (this.models.length) {
      for (const model of this.models) model.update(dt);
      this.buildBVH();
    }
  }

  // ---- the reference's live scene type (scene-raytracing.ts:73-272), meshes/models from the caller
  createTriangleScene(meshes, models) {
    this.meshes = meshes;
    this.triangles = [];                                             // SR:75-79
    for (const mesh of this.meshes) {
      mesh.triangleLookupOffset = this.triangles.length;
      for (const t of mesh.triangles) this.triangles.push(t);
    }
    this.triangleIndices = new Array(this.triangles.length);         // SR:82-93
    {
      let i = 0, offset = 0;
      for (const mesh of this.meshes) {
        for (let j = 0; j < mesh.bvh.triangleIndices.length; ++j) { this.triangleIndices[i] = mesh.bvh.triangleIndices[j] + offset; ++i; }
        offset += mesh.bvh.triangleIndices.length;
      }
    }
    this.models = models;                                            // SR:96-111
    this.tlasNodesMax = 2 * this.models.length - 1;                  // SR:114
    this.blasNodesUsed = 0;                                          // SR:116-120
    for (const mesh of this.meshes) { mesh.rootNodeIndex = this.tlasNodesMax + this.blasNodesUsed; this.blasNodesUsed += mesh.bvh.nodesUsed; }
    this.nodes = new Array(this.tlasNodesMax + this.blasNodesUsed);  // SR:123-131
    for (let i = 0; i < this.tlasNodesMax; i += 1) {
      const node = new Node();
      node.leftChildIndex = 0; node.primitiveCount = 0; node.minCorner = [0, 0, 0]; node.maxCorner = [0, 0, 0];
      this.nodes[i] = node;
    }
    this.buildBVH();                                                 // SR:133
    this.finalizeBVH();                                              // SR:134
    this.blasConsumed = true;
    return this;
  }
  buildBVH() {                                                       // SR:145-179
    this.tlasNodesUsed = 0;
    this.blasList = new Array(this.models.length);
    this.blasIndices = new Array(this.models.length);
    for (let i = 0; i < this.tlasNodesMax; ++i) {
      this.nodes[i].leftChildIndex = 0; this.nodes[i].primitiveCount = 0;
      this.nodes[i].minCorner = [0, 0, 0]; this.nodes[i].maxCorner = [0, 0, 0];
    }
    for (let i = 0; i < this.models.length; ++i) {
      const model = this.models[i];
      const mesh = this.meshes[model.meshIndex];
      this.blasList[i] = new BLAS(mesh.rootNodeIndex, mesh.bvh.minCorner, mesh.bvh.maxCorner, model.model);
      this.blasIndices[i] = i;
    }
    const root = this.nodes[0];
    root.leftChildIndex = 0;
    root.primitiveCount = this.blasList.length;
    this.tlasNodesUsed += 1;
    this.updateBounds(0);
    this.subdivide(0);
  }
  updateBounds(nodeIndex) {                                          // SR:181-191
    const node = this.nodes[nodeIndex];
    node.minCorner = [1e30, 1e30, 1e30];
    node.maxCorner = [-1e30, -1e30, -1e30];
    for (let i = 0; i < node.primitiveCount; i += 1) {
      const blas = this.blasList[this.blasIndices[node.leftChildIndex + i]];
      vec3.min(node.minCorner, node.minCorner, blas.minCorner);
      vec3.max(node.maxCorner, node.maxCorner, blas.maxCorner);
    }
  }
  subdivide(nodeIndex) {                                             // SR:193-254
    const node = this.nodes[nodeIndex];
    if (node.primitiveCount < 2) return;
    const extent = vec3.create();
    vec3.subtract(extent, node.maxCorner, node.minCorner);
    let axis = 0;
    if (extent[1] > extent[axis]) axis = 1;
    if (extent[2] > extent[axis]) axis = 2;
    const splitPosition = node.minCorner[axis] + extent[axis] / 2;
    let i = node.leftChildIndex;
    let j = i + node.primitiveCount - 1;
    while (i <= j) {
      if (this.blasList[this.blasIndices[i]].center[axis] < splitPosition) {
        i += 1;
      } else {
        const temp = this.blasIndices[i]; this.blasIndices[i] = this.blasIndices[j]; this.blasIndices[j] = temp;
        j -= 1;
      }
    }
    const leftCount = i - node.leftChildIndex;
    if (leftCount == 0 || leftCount == node.primitiveCount) return;
    const leftChildIndex = this.tlasNodesUsed; this.tlasNodesUsed += 1;
    const rightChildIndex = this.tlasNodesUsed; this.tlasNodesUsed += 1;
    this.nodes[leftChildIndex].leftChildIndex = node.leftChildIndex;
    this.nodes[leftChildIndex].primitiveCount = leftCount;
    this.nodes[rightChildIndex].leftChildIndex = i;
    this.nodes[rightChildIndex].primitiveCount = node.primitiveCount - leftCount;
    node.leftChildIndex = leftChildIndex;
    node.primitiveCount = 0;
    this.updateBounds(leftChildIndex);
    this.updateBounds(rightChildIndex);
    this.subdivide(leftChildIndex);
    this.subdivide(rightChildIndex);
  }
  finalizeBVH() {                                                    // SR:256-272
    for (const mesh of this.meshes) {
      for (let i = 0; i < mesh.bvh.nodesUsed; ++i) {
        const meshNode = mesh.bvh.nodes[i];
        if (meshNode.primitiveCount == 0) meshNode.leftChildIndex += mesh.rootNodeIndex;
        else meshNode.leftChildIndex += mesh.triangleLookupOffset;
        this.nodes[mesh.rootNodeIndex + i] = meshNode;
      }
    }
  }
}
module.exports = { SceneRaytracing, syntheticSpheres, SplitMix64, BASELINE_CONFIGS, CONSTANT_SKY_RGBA };
