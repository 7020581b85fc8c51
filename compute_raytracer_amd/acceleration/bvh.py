"""BVH -- the bottom-level tree of one mesh, as src/rendering-raycast/acceleration/bvh.ts:7-169
defines it (SAH over nine planes per axis at tenths of the node's extent, in-place two-pointer
partition of the index array, children numbered in depth-first build order), built here with
numpy: triangle corners and f32 centroids are arrays, every candidate plane of a node is priced
in one vectorised pass, and the tree grows off an explicit stack.  (bvh.ts:171-229, the
median-split `subdivide`, is unused upstream and not restated.)

Why the vector form is exact: the reference grows its candidate boxes in gl-matrix vec3s --
Float32Arrays --, so every running min / max is rounded to f32 as it is stored (aabb.ts:12-15).
Rounding to f32 is monotonic, so the f32-rounded running minimum of a sequence is the minimum
of the f32-rounded members: `np.minimum.reduce` over corners cast to float32 gives the same bits
(starting values +-1e30 rounded to f32, which an empty side keeps).

`nodes`, `triangleIndices`, `nodesUsed`, `minCorner`, `maxCorner` are what the scene layer reads;
`minCorner` / `maxCorner` stay at the +-999999 placeholders of bvh.ts:23-25 (nothing upstream
ever computes them, and the instances' boxes are made from exactly these)."""
import numpy as np

from .node import Node

_PLANES = 10
_F32_HUGE = np.float32(1e30)


def _area_f32(lo, hi):
    """2 (ex ey + ey ez + ez ex) with the extents rounded to f32 and the products in f64
    (aabb.ts:17-20); lo / hi are (..., 3) float32."""
    e = (hi - lo).astype(np.float32).astype(np.float64)
    return 2.0 * (e[..., 0] * e[..., 1] + e[..., 1] * e[..., 2] + e[..., 2] * e[..., 0])


class BVH:
    def __init__(self, triangles):
        self.triangles = triangles
        self.triangleCount = n = len(triangles)
        self.minCorner = [999999] * 3
        self.maxCorner = [-999999] * 3
        corners = np.array([[[float(v) for v in c] for c in t.corners] for t in triangles], dtype=np.float64).reshape(n, 3, 3)
        centroid = np.array([[np.float32(v) for v in t.centroid] for t in triangles], dtype=np.float32).reshape(n, 3)
        corners32 = corners.astype(np.float32)
        order = np.arange(n, dtype=np.int64)
        cap = max(2 * n - 1, 1)
        lo = np.full((cap, 3), 1e30)
        hi = np.full((cap, 3), -1e30)
        first = np.zeros(cap, dtype=np.int64)
        count = np.zeros(cap, dtype=np.int64)

        def fit(node):
            c = corners[order[first[node]:first[node] + count[node]]].reshape(-1, 3)
            lo[node] = np.minimum(c.min(axis=0), 1e30) if len(c) else 1e30
            hi[node] = np.maximum(c.max(axis=0), -1e30) if len(c) else -1e30

        used = 0
        if n:
            first[0], count[0], used = 0, n, 1
            fit(0)
            todo = [0]
            while todo:
                node = todo.pop()
                cnt = int(count[node])
                if cnt < 2:
                    continue
                run = order[first[node]:first[node] + cnt]
                cen = centroid[run].astype(np.float64)                 # (cnt, 3): the f32 values, compared in f64
                c32 = corners32[run]                                   # (cnt, 3 corners, 3)
                tmin = c32.min(axis=1)                                 # per-triangle f32 boxes
                tmax = c32.max(axis=1)
                best, best_axis, best_plane = 1e30, 0, 0.0
                for axis in range(3):
                    a, b = float(lo[node, axis]), float(hi[node, axis])
                    for s in range(1, _PLANES):
                        f = s / _PLANES
                        plane = a * (1 - f) + b * f
                        left = cen[:, axis] < plane
                        nl = int(left.sum())
                        cost = 0.0
                        for side, k in ((left, nl), (~left, cnt - nl)):
                            smin = np.minimum(tmin[side].min(axis=0), _F32_HUGE) if k else np.full(3, _F32_HUGE, np.float32)
                            smax = np.maximum(tmax[side].max(axis=0), -_F32_HUGE) if k else np.full(3, -_F32_HUGE, np.float32)
                            cost = cost + float(_area_f32(smin.astype(np.float32), smax.astype(np.float32))) * k
                        if cost < best:
                            best, best_axis, best_plane = cost, axis, plane
                stay = float(_area_f32(lo[node].astype(np.float32), hi[node].astype(np.float32))) * cnt
                if stay < best:
                    continue
                # the reference's two-pointer sweep (bvh.ts:130-141), on the index run itself: its result is
                # an order, not just a set, and the order decides how later nodes partition
                i, j = int(first[node]), int(first[node]) + cnt - 1
                while i <= j:
                    if float(centroid[order[i], best_axis]) < best_plane:
                        i += 1
                    else:
                        order[i], order[j] = order[j], order[i]
                        j -= 1
                n_left = i - int(first[node])
                if n_left == 0 or n_left == cnt:
                    continue
                left_i, right_i = used, used + 1
                used += 2
                first[left_i], count[left_i] = first[node], n_left
                first[right_i], count[right_i] = i, cnt - n_left
                first[node], count[node] = left_i, 0
                fit(left_i)
                fit(right_i)
                todo.append(right_i)
                todo.append(left_i)                                    # the left subtree is numbered first
        self.nodesUsed = used
        self.triangleIndices = [int(v) for v in order]
        self.nodes = []
        for k in range(cap):
            nd = Node()
            if k < used:
                nd.minCorner = [float(v) for v in lo[k]]
                nd.maxCorner = [float(v) for v in hi[k]]
                nd.leftChildIndex = int(first[k])
                nd.primitiveCount = int(count[k])
            self.nodes.append(nd)
