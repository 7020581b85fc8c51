'use strict';
// BLAS -- src/rendering-raycast/acceleration/blas.ts:3-40
const { vec3, mat4 } = require('../gl-matrix-lite');
class BLAS {
  constructor(rootNodeIndex, minCorner, maxCorner, model) {
    this.rootNodeIndex = rootNodeIndex;
    this.minCorner = [1e30, 1e30, 1e30];
    this.maxCorner = [-1e30, -1e30, -1e30];
    this.triangleLookupIndex = 0;
    const corners = [
      [minCorner[0], minCorner[1], minCorner[2]], [minCorner[0], minCorner[1], maxCorner[2]],
      [minCorner[0], maxCorner[1], minCorner[2]], [minCorner[0], maxCorner[1], maxCorner[2]],
      [maxCorner[0], minCorner[1], minCorner[2]], [maxCorner[0], minCorner[1], maxCorner[2]],
      [maxCorner[0], maxCorner[1], minCorner[2]], [maxCorner[0], maxCorner[1], maxCorner[2]],
    ];
    const corner = vec3.create();
    for (let i = 0; i < corners.length; ++i) {
      vec3.transformMat4(corner, corners[i], model);
      vec3.min(this.minCorner, this.minCorner, corner);
      vec3.max(this.maxCorner, this.maxCorner, corner);
    }
    this.center = vec3.create();
    vec3.add(this.center, this.minCorner, this.maxCorner);
    vec3.div(this.center, this.center, [2, 2, 2]);
    this.inverseModel = mat4.create();
    mat4.invert(this.inverseModel, model);
  }
}
module.exports = { BLAS };
