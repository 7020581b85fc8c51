'use strict';
// Triangle -- src/rendering-raycast/model/triangle.ts:4-45
const { vec3 } = require('../gl-matrix-lite');
class Triangle {
  constructor() { this.corners = []; this.textures = []; this.normals = []; this.color = [0, 0, 0, 0]; this.centroid = [0, 0, 0]; }
  calculateCentroid() {
    this.centroid = vec3.create();
    vec3.add(this.centroid, this.centroid, this.corners[0]);
    vec3.add(this.centroid, this.centroid, this.corners[1]);
    vec3.add(this.centroid, this.centroid, this.corners[2]);
    vec3.div(this.centroid, this.centroid, [3, 3, 3]);
    return this.centroid;
  }
}
module.exports = { Triangle };
