'use strict';
// Bottom-level tree of one mesh: the binned surface-area-heuristic build whose RESULT the reference's
// acceleration/bvh.ts:30-169 defines -- nine candidate planes per axis at tenths of the node's extent,
// a leaf wherever splitting would cost more than not splitting, triangle indices partitioned in place
// with the two-pointer sweep, children numbered in the order a depth-first build meets them.
//
// Data-oriented: no node objects and no recursion.  Nodes are four flat arrays (bounds as f64 -- the
// reference keeps them in plain JS arrays --, first child / first triangle, triangle count), the
// candidate boxes of the heuristic are six f32 scalars per side (the reference grows gl-matrix
// vec3s, i.e. Float32Arrays: every min / max is rounded to f32 as it is stored, aabb.ts:12-20), and
// the build runs off an explicit stack.
//
// One field is kept although nothing ever computes it: `lo` / `hi`, the tree-level bounds, stay at
// -+999999 as bvh.ts:23-25 leaves them, and the scene builder transforms exactly those into each
// instance's box (scene-raytracing.ts:158-166) -- so the top level never culls, in the reference
// and here.

const fround = Math.fround;
const PLANES_PER_AXIS = 10;
const F32_HUGE = fround(1e30);

function buildTree(soup) {
  const n = soup.count;
  const cap = Math.max(2 * n - 1, 1);
  const tree = {
    min: new Float64Array(3 * cap), max: new Float64Array(3 * cap),
    first: new Int32Array(cap),      // inner node: index of its left child (right = +1); leaf: first slot in `order`
    count: new Int32Array(cap),      // triangles of a leaf, 0 for an inner node
    order: new Int32Array(n),        // triangle indices, partitioned so that every leaf owns a contiguous run
    used: 0,
    lo: [999999, 999999, 999999], hi: [-999999, -999999, -999999],
  };
  for (let i = 0; i < n; ++i) tree.order[i] = i;
  if (n === 0) return tree;
  const pos = soup.position, cen = soup.centroid, order = tree.order;

  const fit = (node) => {            // exact (f64) bounds of the node's triangles
    let x0 = 1e30, y0 = 1e30, z0 = 1e30, x1 = -1e30, y1 = -1e30, z1 = -1e30;
    for (let k = tree.first[node], e = k + tree.count[node]; k < e; ++k) {
      const p = 9 * order[k];
      for (let c = 0; c < 9; c += 3) {
        x0 = Math.min(x0, pos[p + c]); y0 = Math.min(y0, pos[p + c + 1]); z0 = Math.min(z0, pos[p + c + 2]);
        x1 = Math.max(x1, pos[p + c]); y1 = Math.max(y1, pos[p + c + 1]); z1 = Math.max(z1, pos[p + c + 2]);
      }
    }
    tree.min[3 * node] = x0; tree.min[3 * node + 1] = y0; tree.min[3 * node + 2] = z0;
    tree.max[3 * node] = x1; tree.max[3 * node + 1] = y1; tree.max[3 * node + 2] = z1;
  };
  const area = (x0, y0, z0, x1, y1, z1) => {            // of an f32 box: extents rounded to f32, products in f64
    const ex = fround(x1 - x0), ey = fround(y1 - y0), ez = fround(z1 - z0);
    return 2 * (ex * ey + ey * ez + ez * ex);
  };
  // cost of cutting `node` at `plane` on `axis`: area x count of the two f32 boxes the centroids sort into
  const cutCost = (node, axis, plane) => {
    let ax0 = F32_HUGE, ay0 = F32_HUGE, az0 = F32_HUGE, ax1 = -F32_HUGE, ay1 = -F32_HUGE, az1 = -F32_HUGE, na = 0;
    let bx0 = F32_HUGE, by0 = F32_HUGE, bz0 = F32_HUGE, bx1 = -F32_HUGE, by1 = -F32_HUGE, bz1 = -F32_HUGE, nb = 0;
    for (let k = tree.first[node], e = k + tree.count[node]; k < e; ++k) {
      const t = order[k], p = 9 * t;
      if (cen[3 * t + axis] < plane) {
        ++na;
        for (let c = 0; c < 9; c += 3) {
          ax0 = fround(Math.min(ax0, pos[p + c])); ay0 = fround(Math.min(ay0, pos[p + c + 1])); az0 = fround(Math.min(az0, pos[p + c + 2]));
          ax1 = fround(Math.max(ax1, pos[p + c])); ay1 = fround(Math.max(ay1, pos[p + c + 1])); az1 = fround(Math.max(az1, pos[p + c + 2]));
        }
      } else {
        ++nb;
        for (let c = 0; c < 9; c += 3) {
          bx0 = fround(Math.min(bx0, pos[p + c])); by0 = fround(Math.min(by0, pos[p + c + 1])); bz0 = fround(Math.min(bz0, pos[p + c + 2]));
          bx1 = fround(Math.max(bx1, pos[p + c])); by1 = fround(Math.max(by1, pos[p + c + 1])); bz1 = fround(Math.max(bz1, pos[p + c + 2]));
        }
      }
    }
    return area(ax0, ay0, az0, ax1, ay1, az1) * na + area(bx0, by0, bz0, bx1, by1, bz1) * nb;
  };

  tree.first[0] = 0; tree.count[0] = n; tree.used = 1;
  fit(0);
  const todo = [0];
  while (todo.length) {
    const node = todo.pop();
    const cnt = tree.count[node];
    if (cnt < 2) continue;
    let best = 1e30, bestAxis = 0, bestPlane = 0;
    for (let axis = 0; axis < 3; ++axis) {
      const a = tree.min[3 * node + axis], b = tree.max[3 * node + axis];
      for (let s = 1; s < PLANES_PER_AXIS; ++s) {
        const f = s / PLANES_PER_AXIS;
        const plane = a * (1 - f) + b * f;
        const cost = cutCost(node, axis, plane);
        if (cost < best) { best = cost; bestAxis = axis; bestPlane = plane; }
      }
    }
    // not splitting: the node's own box, rounded to f32 as the heuristic's boxes are
    const stay = area(fround(tree.min[3 * node]), fround(tree.min[3 * node + 1]), fround(tree.min[3 * node + 2]),
                      fround(tree.max[3 * node]), fround(tree.max[3 * node + 1]), fround(tree.max[3 * node + 2])) * cnt;
    if (stay < best) continue;
    let i = tree.first[node], j = i + cnt - 1;
    while (i <= j) {
      if (cen[3 * order[i] + bestAxis] < bestPlane) ++i;
      else { const t = order[i]; order[i] = order[j]; order[j] = t; --j; }
    }
    const nLeft = i - tree.first[node];
    if (nLeft === 0 || nLeft === cnt) continue;
    const left = tree.used, right = left + 1;
    tree.used += 2;
    tree.first[left] = tree.first[node]; tree.count[left] = nLeft;
    tree.first[right] = i; tree.count[right] = cnt - nLeft;
    tree.first[node] = left; tree.count[node] = 0;
    fit(left); fit(right);
    todo.push(right, left);          // the left subtree is numbered before the right one
  }
  return tree;
}

module.exports = { buildTree };
