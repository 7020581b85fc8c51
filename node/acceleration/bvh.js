'use strict';
// BVH -- src/rendering-raycast/acceleration/bvh.ts:7-169 (SAH build; the unused median-split
// `subdivide`, bvh.ts:171-229, is not restated).  Recursion made explicit where depth could hurt.
const { Node } = require('./node');
const { AABB } = require('./aabb');

class BVH {
  constructor(triangles) {
    this.triangles = triangles;
    this.triangleCount = triangles.length;
    this.nodesUsed = 0;
    const MAX_NUMBER = 999999;
    this.minCorner = [MAX_NUMBER, MAX_NUMBER, MAX_NUMBER];      // never updated upstream (bvh.ts:23-25)
    this.maxCorner = [-MAX_NUMBER, -MAX_NUMBER, -MAX_NUMBER];
    this.buildBVH();
  }
  buildBVH() {
    this.triangleIndices = new Array(this.triangleCount);
    for (let i = 0; i < this.triangleCount; i += 1) this.triangleIndices[i] = i;
    this.nodes = new Array(2 * this.triangles.length - 1);
    for (let i = 0; i < 2 * this.triangles.length - 1; i += 1) this.nodes[i] = new Node();
    const root = this.nodes[0];
    root.leftChildIndex = 0;
    root.primitiveCount = this.triangles.length;
    this.nodesUsed += 1;
    this.updateBounds(0);
    this.subdivideSAH(0);
  }
  updateBounds(nodeIndex) {
    const node = this.nodes[nodeIndex];
    node.minCorner = [1e30, 1e30, 1e30];
    node.maxCorner = [-1e30, -1e30, -1e30];
    for (let i = 0; i < node.primitiveCount; i += 1) {
      const triangle = this.triangles[this.triangleIndices[node.leftChildIndex + i]];
      for (const corner of triangle.corners) {
        for (let k = 0; k < 3; ++k) {
          node.minCorner[k] = Math.min(node.minCorner[k], corner[k]);
          node.maxCorner[k] = Math.max(node.maxCorner[k], corner[k]);
        }
      }
    }
  }
  findBestSplit(node) {
    const SPLIT_PER_AXIS = 10;
    let bestCost = 1e30, bestAxis = 0, bestSplitPosition = 0;
    for (let axis = 0; axis <= 2; ++axis) {
      for (let noSplit = 1; noSplit < SPLIT_PER_AXIS; ++noSplit) {
        const splitPercent = noSplit / SPLIT_PER_AXIS;
        const splitPosition = node.minCorner[axis] * (1 - splitPercent) + node.maxCorner[axis] * splitPercent;
        const cost = this.SAH(node, axis, splitPosition);
        if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestSplitPosition = splitPosition; }
      }
    }
    return [bestAxis, bestSplitPosition, bestCost];
  }
  SAH(node, axis, splitPosition) {
    const leftAABB = new AABB(), rightAABB = new AABB();
    let nbTrianglesLeft = 0, nbTrianglesRight = 0;
    for (let i = 0; i < node.primitiveCount; ++i) {
      const triangle = this.triangles[this.triangleIndices[i + node.leftChildIndex]];
      if (triangle.centroid[axis] < splitPosition) {
        ++nbTrianglesLeft;
        leftAABB.grow(triangle.corners[0]); leftAABB.grow(triangle.corners[1]); leftAABB.grow(triangle.corners[2]);
      } else {
        ++nbTrianglesRight;
        rightAABB.grow(triangle.corners[0]); rightAABB.grow(triangle.corners[1]); rightAABB.grow(triangle.corners[2]);
      }
    }
    return leftAABB.surfaceArea() * nbTrianglesLeft + rightAABB.surfaceArea() * nbTrianglesRight;
  }
  subdivideSAH(nodeIndex) {
    const node = this.nodes[nodeIndex];
    if (node.primitiveCount < 2) return;
    const [axis, splitPosition, subdivisionCost] = this.findBestSplit(node);
    const parentAABB = new AABB();
    parentAABB.grow(node.minCorner);
    parentAABB.grow(node.maxCorner);
    const parentCost = parentAABB.surfaceArea() * node.primitiveCount;
    if (parentCost < subdivisionCost) return;
    let i = node.leftChildIndex;
    let j = i + node.primitiveCount - 1;
    while (i <= j) {
      if (this.triangles[this.triangleIndices[i]].centroid[axis] < splitPosition) {
        i += 1;
      } else {
        const temp = this.triangleIndices[i];
        this.triangleIndices[i] = this.triangleIndices[j];
        this.triangleIndices[j] = temp;
        j -= 1;
      }
    }
    const leftCount = i - node.leftChildIndex;
    if (leftCount == 0 || leftCount == node.primitiveCount) return;
    const leftChildIndex = this.nodesUsed; this.nodesUsed += 1;
    const rightChildIndex = this.nodesUsed; this.nodesUsed += 1;
    this.nodes[leftChildIndex].leftChildIndex = node.leftChildIndex;
    this.nodes[leftChildIndex].primitiveCount = leftCount;
    this.nodes[rightChildIndex].leftChildIndex = i;
    this.nodes[rightChildIndex].primitiveCount = node.primitiveCount - leftCount;
    node.leftChildIndex = leftChildIndex;
    node.primitiveCount = 0;
    this.updateBounds(leftChildIndex);
    this.updateBounds(rightChildIndex);
    this.subdivideSAH(leftChildIndex);
    this.subdivideSAH(rightChildIndex);
  }
}
module.exports = { BVH };
