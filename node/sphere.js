'use strict';
// Sphere -- src/rendering-raycast/model/sphere.ts:1-10
class Sphere {
  constructor(center, radius, color) {
    this.center = new Float32Array(center);
    this.radius = radius;
    this.color = new Float32Array(color);
  }
}
module.exports = { Sphere };
