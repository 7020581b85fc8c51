"""The handful of gl-matrix 3.4.3 functions the reference's ray-cast path calls
(package-lock.json:612-616; gl-matrix is not vendored in the reference, so this restates its
published algorithms).  gl-matrix's ARRAY_TYPE is Float32Array: vectors and matrices made by
`create()` / `fromValues()` round to f32 at every store while the arithmetic in between is f64;
plain JS arrays (`[1e30, 1e30, 1e30]` literals) stay f64.  Here: `F32Vec` = Float32Array
(numpy float32 array), Python lists = plain JS arrays.  Every function writes `out` in place and
returns it, like gl-matrix."""
import math

import numpy as np


def vec3_create():
    return np.zeros(3, dtype=np.float32)


def vec3_from_values(x, y, z):
    return np.array([x, y, z], dtype=np.float64).astype(np.float32)


def mat4_create():
    m = np.zeros(16, dtype=np.float32)
    m[0] = m[5] = m[10] = m[15] = 1.0
    return m


def _store(out, vals):
    """`out[i] = value`: rounds to f32 when out is a Float32Array, keeps f64 in a plain array."""
    for i, v in enumerate(vals):
        out[i] = v
    return out


def vec3_add(out, a, b):
    return _store(out, [float(a[0]) + float(b[0]), float(a[1]) + float(b[1]), float(a[2]) + float(b[2])])


def vec3_subtract(out, a, b):
    return _store(out, [float(a[0]) - float(b[0]), float(a[1]) - float(b[1]), float(a[2]) - float(b[2])])


def vec3_mul(out, a, b):
    return _store(out, [float(a[0]) * float(b[0]), float(a[1]) * float(b[1]), float(a[2]) * float(b[2])])


def vec3_div(out, a, b):
    return _store(out, [float(a[0]) / float(b[0]), float(a[1]) / float(b[1]), float(a[2]) / float(b[2])])


def vec3_min(out, a, b):
    return _store(out, [min(float(a[0]), float(b[0])), min(float(a[1]), float(b[1])), min(float(a[2]), float(b[2]))])


def vec3_max(out, a, b):
    return _store(out, [max(float(a[0]), float(b[0])), max(float(a[1]), float(b[1])), max(float(a[2]), float(b[2]))])


def vec3_transform_mat4(out, a, m):
    x, y, z = float(a[0]), float(a[1]), float(a[2])
    m = [float(v) for v in m]
    w = m[3] * x + m[7] * y + m[11] * z + m[15]
    w = w or 1.0
    return _store(out, [(m[0] * x + m[4] * y + m[8] * z + m[12]) / w,
                        (m[1] * x + m[5] * y + m[9] * z + m[13]) / w,
                        (m[2] * x + m[6] * y + m[10] * z + m[14]) / w])


def mat4_translate(out, a, v):
    x, y, z = float(v[0]), float(v[1]), float(v[2])
    if a is out:
        A = [float(t) for t in a]
        out[12] = A[0] * x + A[4] * y + A[8] * z + A[12]
        out[13] = A[1] * x + A[5] * y + A[9] * z + A[13]
        out[14] = A[2] * x + A[6] * y + A[10] * z + A[14]
        out[15] = A[3] * x + A[7] * y + A[11] * z + A[15]
    else:
        A = [float(t) for t in a]
        for i in range(12):
            out[i] = A[i]
        out[12] = A[0] * x + A[4] * y + A[8] * z + A[12]
        out[13] = A[1] * x + A[5] * y + A[9] * z + A[13]
        out[14] = A[2] * x + A[6] * y + A[10] * z + A[14]
        out[15] = A[3] * x + A[7] * y + A[11] * z + A[15]
    return out


def mat4_rotate_y(out, a, rad):
    s, c = math.sin(rad), math.cos(rad)
    A = [float(t) for t in a]
    a00, a01, a02, a03 = A[0], A[1], A[2], A[3]
    a20, a21, a22, a23 = A[8], A[9], A[10], A[11]
    if a is not out:
        for i in (4, 5, 6, 7, 12, 13, 14, 15):
            out[i] = A[i]
    out[0] = a00 * c - a20 * s
    out[1] = a01 * c - a21 * s
    out[2] = a02 * c - a22 * s
    out[3] = a03 * c - a23 * s
    out[8] = a00 * s + a20 * c
    out[9] = a01 * s + a21 * c
    out[10] = a02 * s + a22 * c
    out[11] = a03 * s + a23 * c
    return out


def mat4_invert(out, a):
    A = [float(t) for t in a]
    a00, a01, a02, a03, a10, a11, a12, a13, a20, a21, a22, a23, a30, a31, a32, a33 = A
    b00 = a00 * a11 - a01 * a10
    b01 = a00 * a12 - a02 * a10
    b02 = a00 * a13 - a03 * a10
    b03 = a01 * a12 - a02 * a11
    b04 = a01 * a13 - a03 * a11
    b05 = a02 * a13 - a03 * a12
    b06 = a20 * a31 - a21 * a30
    b07 = a20 * a32 - a22 * a30
    b08 = a20 * a33 - a23 * a30
    b09 = a21 * a32 - a22 * a31
    b10 = a21 * a33 - a23 * a31
    b11 = a22 * a33 - a23 * a32
    det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06
    if not det:
        return None
    det = 1.0 / det
    out[0] = (a11 * b11 - a12 * b10 + a13 * b09) * det
    out[1] = (a02 * b10 - a01 * b11 - a03 * b09) * det
    out[2] = (a31 * b05 - a32 * b04 + a33 * b03) * det
    out[3] = (a22 * b04 - a21 * b05 - a23 * b03) * det
    out[4] = (a12 * b08 - a10 * b11 - a13 * b07) * det
    out[5] = (a00 * b11 - a02 * b08 + a03 * b07) * det
    out[6] = (a32 * b02 - a30 * b05 - a33 * b01) * det
    out[7] = (a20 * b05 - a22 * b02 + a23 * b01) * det
    out[8] = (a10 * b10 - a11 * b08 + a13 * b06) * det
    out[9] = (a01 * b08 - a00 * b10 - a03 * b06) * det
    out[10] = (a30 * b04 - a31 * b02 + a33 * b00) * det
    out[11] = (a21 * b02 - a20 * b04 - a23 * b00) * det
    out[12] = (a11 * b07 - a10 * b09 - a12 * b06) * det
    out[13] = (a00 * b09 - a01 * b07 + a02 * b06) * det
    out[14] = (a31 * b01 - a30 * b03 - a32 * b00) * det
    out[15] = (a20 * b03 - a21 * b01 + a22 * b00) * det
    return out
