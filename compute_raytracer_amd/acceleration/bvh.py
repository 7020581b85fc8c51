"""Bottom-level tree of one mesh, as arrays.

The RESULT is the one src/rendering-raycast/acceleration/bvh.ts:30-169 defines (SAH over nine planes
per axis at tenths of the node's extent, a leaf wherever splitting costs more than staying, triangle
indices partitioned in place by a two-pointer sweep, children numbered in depth-first build order); the
build is numpy: corners and f32 centroids are arrays, every candidate plane of a node is priced in
one vectorised pass, the tree grows off an explicit stack into five flat arrays.  (bvh.ts:171-229, the
median-split `subdivide`, is unused upstream and not restated.)

Why the vector form is exact: the reference grows its candidate boxes in gl-matrix vec3s --
Float32Arrays --, so every running min / max is rounded to f32 as it is stored (aabb.ts:12-15).
Rounding to f32 is monotonic, so the f32-rounded running minimum of a sequence is the minimum of the
f32-rounded members: a reduction over corners cast to float32 gives the same bits (starting values
+-1e30 rounded to f32, which an empty side keeps).

`box_lo` / `box_hi` stay at the +-999999 placeholders of bvh.ts:23-25: nothing upstream ever computes
them, and every instance's world box is made from exactly these (the top level never culls)."""
import numpy as np

_PLANES = 10
_F32_HUGE = np.float32(1e30)


class MeshTree:
    """lo / hi (used, 3) f64 node bounds; first (used,) = left child of an inner node (right = +1) or
    first slot of a leaf's run in `order`; count (used,) = triangles of a leaf, 0 for an inner node;
    order (T,) = triangle indices, every leaf owning a contiguous run."""
    __slots__ = ("lo", "hi", "first", "count", "order", "used", "box_lo", "box_hi")

    def nodes(self, child_base, lookup_base):
        """(used, 8) f32 node records of renderer-raytracing.ts:212-223 with the indices rebased into the
        scene's arrays (scene-raytracing.ts:256-272): children by `child_base`, leaf runs by `lookup_base`."""
        out = np.zeros((self.used, 8), dtype=np.float32)
        out[:, 0:3] = self.lo
        out[:, 4:7] = self.hi
        out[:, 3] = self.first + np.where(self.count == 0, child_base, lookup_base)
        out[:, 7] = self.count
        return out


def _area_f32(lo, hi):
    """2 (ex ey + ey ez + ez ex) with the extents rounded to f32 and the products in f64
    (aabb.ts:17-20); lo / hi are (..., 3) float32."""
    e = (hi - lo).astype(np.float32).astype(np.float64)
    return 2.0 * (e[..., 0] * e[..., 1] + e[..., 1] * e[..., 2] + e[..., 2] * e[..., 0])


def build_tree(soup):
    n = soup.count
    corners = soup.position                                 # (n, 3, 3) f64
    centroid = soup.centroid                                # (n, 3) f32
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
            # the two-pointer sweep (bvh.ts:130-141), on the index run itself: its result is an order, not
            # just a set, and the order decides how later nodes partition
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
    t = MeshTree()
    t.lo, t.hi, t.first, t.count = lo[:used].copy(), hi[:used].copy(), first[:used].copy(), count[:used].copy()
    t.order, t.used = order, used
    t.box_lo = np.array([999999.0] * 3)
    t.box_hi = np.array([-999999.0] * 3)
    return t
