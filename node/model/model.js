'use strict';
// Model -- src/rendering-raycast/model/model.ts:4-38
const { vec3, mat4 } = require('../gl-matrix-lite');
const { deg2rad } = require('../camera');
class Model {
  constructor(meshIndex, position, eulers, eulerSpeed) {
    this.meshIndex = meshIndex;
    this.position = position;
    this.eulers = eulers;
    this.eulerSpeed = eulerSpeed ? eulerSpeed.valueOf() : [0, 0, 0];
    this.calculateTransform();
  }
  update(dt) {
    const rotation = vec3.mul(vec3.create(), this.eulerSpeed, [dt, dt, dt]);
    vec3.add(this.eulers, this.eulers, rotation);
    if (this.eulers[0] > 360) this.eulers[0] -= 360;
    if (this.eulers[1] > 360) this.eulers[1] -= 360;
    if (this.eulers[2] > 360) this.eulers[2] -= 360;
    if (this.eulers[0] < -360) this.eulers[0] += 360;
    if (this.eulers[1] < -360) this.eulers[1] += 360;
    if (this.eulers[2] < -360) this.eulers[2] += 360;
    this.calculateTransform();
  }
  calculateTransform() {
    this.model = mat4.create();
    mat4.translate(this.model, this.model, this.position);
    mat4.rotateY(this.model, this.model, deg2rad(this.eulers[1]));
  }
}
module.exports = { Model };
