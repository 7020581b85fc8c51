'use strict';
// Mesh -- src/rendering-raycast/mesh.ts:7-22
const { ObjectReader } = require('./model/reader/obj-reader');
const { BVH } = require('./acceleration/bvh');
class Mesh {
  constructor() { this.triangleLookupOffset = 0; this.rootNodeIndex = 0; }
  async initialize(url, descriptor) { this.triangles = ObjectReader.loadMeshFromObjFile(url, descriptor); this.bvh = new BVH(this.triangles); return this; }
  initializeFromText(text, descriptor) { this.triangles = ObjectReader.loadMeshFromObjText(text, descriptor); this.bvh = new BVH(this.triangles); return this; }
}
module.exports = { Mesh };
