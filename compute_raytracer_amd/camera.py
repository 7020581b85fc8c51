"""Camera -- mirror of src/rendering-raycast/camera.ts:5-64.

gl-matrix 3.4.3 semantics are reproduced: vectors made by vec3.create()/fromValues()
are Float32Array, so every store rounds to f32 while the arithmetic in between is
f64 (JS numbers).  `position` is whatever the caller passed (the reference passes a
plain JS array, so it stays f64 until packed, SR:39).
"""
import math

import numpy as np


def deg2rad(theta):  # src/utils/more-math.ts:3-5
    return theta * math.pi / 180


def clamp(x, a, b):  # src/utils/more-math.ts:11-13
    return max(min(x, b), a)


def _f32(v):
    return np.asarray(v, dtype=np.float64).astype(np.float32)


def _cross(a, b):  # gl-matrix vec3.cross, f64 arithmetic, f32 store
    ax, ay, az = (float(c) for c in a)
    bx, by, bz = (float(c) for c in b)
    return _f32([ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx])


def _normalize(a):  # gl-matrix vec3.normalize
    x, y, z = (float(c) for c in a)
    ln = x * x + y * y + z * z
    if ln > 0:
        ln = 1 / math.sqrt(ln)
    return _f32([x * ln, y * ln, z * ln])


class Camera:
    def __init__(self, position, theta, phi):  # camera.ts:13-21
        self.position = [float(c) for c in position]
        # vec2.fromValues -> Float32Array
        self.eulers = _f32([math.fmod(phi, 360), clamp(theta, 1, 180)])
        self.forwards = np.zeros(3, np.float32)
        self.right = np.zeros(3, np.float32)
        self.up = np.zeros(3, np.float32)
        self.update()

    def spin(self, dx, dy):  # camera.ts:23-30
        e0 = math.fmod(float(np.float32(float(self.eulers[0]) + dx)), 360)
        e1 = float(np.float32(float(self.eulers[1]) + dy))
        self.eulers = _f32([e0, clamp(e1, 1, 180)])
        self.update()

    def move(self, forwardsAmount, rightAmount):  # camera.ts:32-40 (vec3.scaleAndAdd on a JS array)
        for axis, amount in ((self.forwards, forwardsAmount), (self.right, rightAmount)):
            self.position = [self.position[i] + float(axis[i]) * amount for i in range(3)]

    def update(self):  # camera.ts:42-60 (the view matrix is unused by the ray tracer)
        e0, e1 = float(self.eulers[0]), float(self.eulers[1])
        self.forwards = _f32([
            math.cos(deg2rad(e0)) * math.sin(deg2rad(e1)),
            math.cos(deg2rad(e1)),
            math.sin(deg2rad(e0)) * math.sin(deg2rad(e1)),
        ])
        self.right = _normalize(_cross(self.forwards, [0, 1, 0]))
        self.up = _normalize(_cross(self.right, self.forwards))
