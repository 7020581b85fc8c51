"""Mesh instances ("models") and the top-level tree over them, as arrays, written straight into the
per-frame upload buffers of renderer-raytracing.ts:169-192:

    blas        (M, 20) f32   inverse instance matrix (column-major) + rootNodeIndex + 3 pad
    blas_lookup (M,)    f32   instance order of the top-level leaves (indices as f32, RR:177-181)
    tlas_nodes  (U, 8)  f32   {min.xyz, first, max.xyz, count}

What the numbers must be is fixed by the reference (model/model.ts:19-37, acceleration/blas.ts:11-39,
scene-raytracing.ts:145-254) and by gl-matrix 3.4.3's storage rule -- `create()`-d vectors and matrices
are Float32Arrays: arithmetic in f64, every store rounded to f32 --; how they are produced is not taken
from there: all instances at once, (M, ...) arrays, one rounding per stored value.

    matrix     = translate(I, position) . rotateY(eulers.y)             model.ts:33-37 (eulers.x/z unused)
    world box  = min / max over the 8 corners of the mesh's tree-level box (always -+999999, bvh.ts:23-25)
                 through the matrix, each corner stored f32 (vec3.transformMat4 into a Float32Array)
    centre     = f32(f32(lo + hi) / 2)                                   blas.ts:34-36
    inverse    = mat4.invert: cofactor form, f64, stored f32; singular -> identity stays (blas.ts:8,38)
    top level  = median split at min + extent / 2 of the longest axis (extent stored f32), two-pointer
                 partition on the centres, children numbered in depth-first build order."""
import math

import numpy as np

F32 = np.float32


def _f32(a):
    return np.asarray(a, dtype=np.float64).astype(F32)


class Instances:
    def __init__(self, mesh_index, position, eulers, euler_speed=None):
        self.mesh_index = np.asarray(mesh_index, dtype=np.int64).reshape(-1)
        m = len(self.mesh_index)
        self.position = np.asarray(position, dtype=np.float64).reshape(m, 3).copy()
        self.eulers = np.asarray(eulers, dtype=np.float64).reshape(m, 3).copy()
        self.euler_speed = (np.zeros((m, 3)) if euler_speed is None
                            else np.asarray(euler_speed, dtype=np.float64).reshape(m, 3).copy())

    @classmethod
    def from_records(cls, records):
        """records: dicts {meshIndex, position, eulers, eulerSpeed?} (the arguments of `new Model`, model.ts:11)."""
        return cls([r["meshIndex"] for r in records], [r["position"] for r in records],
                   [r["eulers"] for r in records], [r.get("eulerSpeed") or [0, 0, 0] for r in records])

    def __len__(self):
        return len(self.mesh_index)

    def turn(self, dt):
        """model.ts:19-31: the increment speed * dt is an f32 vector, the angles stay f64; one wrap per step."""
        self.eulers = self.eulers + (self.euler_speed * dt).astype(F32).astype(np.float64)
        self.eulers = np.where(self.eulers > 360, self.eulers - 360, self.eulers)
        self.eulers = np.where(self.eulers < -360, self.eulers + 360, self.eulers)

    def matrices(self):
        """(M, 16) f32, column-major."""
        m = len(self)
        a = np.zeros((m, 16), dtype=np.float64)
        a[:, [0, 5, 10, 15]] = 1.0
        x, y, z = self.position[:, 0], self.position[:, 1], self.position[:, 2]
        for r in range(4):                                         # mat4.translate, out === a
            a[:, 12 + r] = _f32(a[:, r] * x + a[:, 4 + r] * y + a[:, 8 + r] * z + a[:, 12 + r])
        rad = self.eulers[:, 1] * math.pi / 180
        s, c = np.sin(rad), np.cos(rad)
        col0, col2 = a[:, 0:4].copy(), a[:, 8:12].copy()
        a[:, 0:4] = _f32(col0 * c[:, None] - col2 * s[:, None])    # mat4.rotateY
        a[:, 8:12] = _f32(col0 * s[:, None] + col2 * c[:, None])
        return a.astype(F32)


def invert_mat4(mats):
    """gl-matrix mat4.invert for (M, 16) f32 matrices -> (M, 16) f32; singular ones come back as identity."""
    A = np.asarray(mats, dtype=np.float64)
    a = [A[:, i] for i in range(16)]
    b00 = a[0] * a[5] - a[1] * a[4];   b01 = a[0] * a[6] - a[2] * a[4]
    b02 = a[0] * a[7] - a[3] * a[4];   b03 = a[1] * a[6] - a[2] * a[5]
    b04 = a[1] * a[7] - a[3] * a[5];   b05 = a[2] * a[7] - a[3] * a[6]
    b06 = a[8] * a[13] - a[9] * a[12];  b07 = a[8] * a[14] - a[10] * a[12]
    b08 = a[8] * a[15] - a[11] * a[12]; b09 = a[9] * a[14] - a[10] * a[13]
    b10 = a[9] * a[15] - a[11] * a[13]; b11 = a[10] * a[15] - a[11] * a[14]
    det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06
    ok = det != 0
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.where(ok, 1.0 / det, 0.0)
    cof = [a[5] * b11 - a[6] * b10 + a[7] * b09,   a[2] * b10 - a[1] * b11 - a[3] * b09,
           a[13] * b05 - a[14] * b04 + a[15] * b03, a[10] * b04 - a[9] * b05 - a[11] * b03,
           a[6] * b08 - a[4] * b11 - a[7] * b07,   a[0] * b11 - a[2] * b08 + a[3] * b07,
           a[14] * b02 - a[12] * b05 - a[15] * b01, a[8] * b05 - a[10] * b02 + a[11] * b01,
           a[4] * b10 - a[5] * b08 + a[7] * b06,   a[1] * b08 - a[0] * b10 - a[3] * b06,
           a[12] * b04 - a[13] * b02 + a[15] * b00, a[9] * b02 - a[8] * b04 - a[11] * b00,
           a[5] * b07 - a[4] * b09 - a[6] * b06,   a[0] * b09 - a[1] * b07 + a[2] * b06,
           a[13] * b01 - a[12] * b03 - a[14] * b00, a[8] * b03 - a[9] * b01 + a[10] * b00]
    out = np.stack([c * inv for c in cof], axis=1)
    ident = np.zeros(16); ident[[0, 5, 10, 15]] = 1.0
    return np.where(ok[:, None], out, ident[None, :]).astype(F32)


_HI = np.array([[(c >> 2) & 1, (c >> 1) & 1, c & 1] for c in range(8)], dtype=bool)      # corner c takes hi where its bit is set (x: 4, y: 2, z: 1)


def world_boxes(mats, box_lo, box_hi):
    """(M, 3) f64 lo, hi (f32 values) and (M, 3) f32 centres of the instances' boxes.
    box_lo / box_hi: (M, 3), the tree-level box of each instance's mesh.  All eight corners of all instances at once; every
    value goes through the operations of the corner-by-corner form in its order (products and sums left to right in f64, the
    division, one rounding to f32)."""
    M = np.asarray(mats, dtype=np.float64)
    p = np.where(_HI[None, :, :], np.asarray(box_hi)[:, None, :], np.asarray(box_lo)[:, None, :])      # (M, 8, 3)
    px, py, pz = p[:, :, 0], p[:, :, 1], p[:, :, 2]
    col = lambda i: M[:, i, None]
    w = col(3) * px + col(7) * py + col(11) * pz + col(15)
    w = np.where((w == 0) | np.isnan(w), 1.0, w)                   # `w = w || 1.0`: 0 and NaN are both falsy
    q = np.stack([(col(k) * px + col(4 + k) * py + col(8 + k) * pz + col(12 + k)) / w for k in range(3)], axis=2)
    q = q.astype(F32).astype(np.float64)                           # (M, 8, 3)
    lo = np.minimum(1e30, q.min(axis=1))
    hi = np.maximum(-1e30, q.max(axis=1))
    centre = ((lo + hi).astype(F32).astype(np.float64) / 2.0).astype(F32)
    return lo, hi, centre


def top_level(lo, hi, centre):
    """-> (tlas_nodes (U, 8) f32, blas_lookup (M,) f32).  U <= 2 M - 1."""
    m = len(lo)
    cap = max(2 * m - 1, 1)
    nlo = np.full((cap, 3), 1e30); nhi = np.full((cap, 3), -1e30)
    first = np.zeros(cap, dtype=np.int64); count = np.zeros(cap, dtype=np.int64)
    order = np.arange(m, dtype=np.int64)

    def fit(node):
        run = order[first[node]:first[node] + count[node]]
        nlo[node] = np.minimum(lo[run].min(axis=0), 1e30) if len(run) else 1e30
        nhi[node] = np.maximum(hi[run].max(axis=0), -1e30) if len(run) else -1e30

    used = 0
    if m:
        count[0], used = m, 1
        fit(0)
        todo = [0]
        while todo:
            node = todo.pop()
            if count[node] < 2:
                continue
            extent = (nhi[node] - nlo[node]).astype(F32).astype(np.float64)      # vec3.subtract into vec3.create()
            axis = 0
            if extent[1] > extent[axis]: axis = 1
            if extent[2] > extent[axis]: axis = 2
            plane = nlo[node, axis] + extent[axis] / 2
            i, j = int(first[node]), int(first[node] + count[node]) - 1
            while i <= j:
                if float(centre[order[i], axis]) < plane:
                    i += 1
                else:
                    order[i], order[j] = order[j], order[i]
                    j -= 1
            n_left = i - int(first[node])
            if n_left == 0 or n_left == count[node]:
                continue
            left, right = used, used + 1
            used += 2
            first[left], count[left] = first[node], n_left
            first[right], count[right] = i, count[node] - n_left
            first[node], count[node] = left, 0
            fit(left); fit(right)
            todo.append(right); todo.append(left)
    nodes = np.zeros((used, 8), dtype=F32)
    nodes[:, 0:3] = nlo[:used]
    nodes[:, 3] = first[:used]
    nodes[:, 4:7] = nhi[:used]
    nodes[:, 7] = count[:used]
    return nodes, order.astype(np.float64).astype(F32)
