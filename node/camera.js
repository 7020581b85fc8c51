'use strict';
// Camera -- Node-12 CommonJS restatement of src/rendering-raycast/camera.ts:5-64 (the reference is
// TypeScript + gl-matrix; neither tsc nor gl-matrix exists offline).  gl-matrix semantics kept:
// vec3.create()/fromValues() are Float32Array, arithmetic in between is f64.
function deg2rad(theta) { return theta * Math.PI / 180; }            // utils/more-math.ts:3-5
function clamp(x, a, b) { return Math.max(Math.min(x, b), a); }      // utils/more-math.ts:11-13

function cross(out, a, b) {                                          // gl-matrix vec3.cross
  const ax = a[0], ay = a[1], az = a[2], bx = b[0], by = b[1], bz = b[2];
  out[0] = ay * bz - az * by; out[1] = az * bx - ax * bz; out[2] = ax * by - ay * bx;
  return out;
}
function normalize(out, a) {                                         // gl-matrix vec3.normalize
  const x = a[0], y = a[1], z = a[2];
  let len = x * x + y * y + z * z;
  if (len > 0) len = 1 / Math.sqrt(len);
  out[0] = x * len; out[1] = y * len; out[2] = z * len;
  return out;
}

class Camera {
  constructor(position, theta, phi) {                                // camera.ts:13-21
    this.position = position;
    this.eulers = new Float32Array([phi % 360, clamp(theta, 1, 180)]);
    this.forwards = new Float32Array(3);
    this.right = new Float32Array(3);
    this.up = new Float32Array(3);
    this.update();
  }
  spin(dx, dy) {                                                     // camera.ts:23-30
    this.eulers[0] += dx; this.eulers[0] %= 360;
    this.eulers[1] += dy; this.eulers[1] = clamp(this.eulers[1], 1, 180);
    this.update();
  }
  move(forwardsAmount, rightAmount) {                                // camera.ts:32-40
    for (let i = 0; i < 3; ++i) this.position[i] = this.position[i] + this.forwards[i] * forwardsAmount;
    for (let i = 0; i < 3; ++i) this.position[i] = this.position[i] + this.right[i] * rightAmount;
  }
  update() {                                                         // camera.ts:42-60
    this.forwards = new Float32Array([
      Math.cos(deg2rad(this.eulers[0])) * Math.sin(deg2rad(this.eulers[1])),
      Math.cos(deg2rad(this.eulers[1])),
      Math.sin(deg2rad(this.eulers[0])) * Math.sin(deg2rad(this.eulers[1])),
    ]);
    cross(this.right, this.forwards, [0, 1, 0]); normalize(this.right, this.right);
    cross(this.up, this.right, this.forwards); normalize(this.up, this.up);
  }
}
module.exports = { Camera, deg2rad, clamp };
