'use strict';
// AABB -- src/rendering-raycast/acceleration/aabb.ts:3-21
const { vec3 } = require('../gl-matrix-lite');
class AABB {
  constructor() { this.min = vec3.fromValues(1e30, 1e30, 1e30); this.max = vec3.fromValues(-1e30, -1e30, -1e30); }
  grow(corner) { vec3.min(this.min, this.min, corner); vec3.max(this.max, this.max, corner); }
  surfaceArea() {
    const e = vec3.subtract(vec3.create(), this.max, this.min);
    return 2 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]);
  }
}
module.exports = { AABB };
