"""BVH -- src/rendering-raycast/acceleration/bvh.ts:7-169: top-down SAH build over a mesh's
triangles, 9 candidate planes per axis, in-place partition of the index array.  (bvh.ts:171-229,
the median-split `subdivide`, is unused upstream and not restated.)"""
import sys

from .. import glmatrix as glm
from .aabb import AABB
from .node import Node


class BVH:
    def __init__(self, triangles):                             # bvh.ts:18-28
        self.triangles = triangles
        self.triangleCount = len(triangles)
        self.nodesUsed = 0
        MAX_NUMBER = 999999
        self.minCorner = [MAX_NUMBER] * 3
        self.maxCorner = [-MAX_NUMBER] * 3
        self.buildBVH()

    def buildBVH(self):                                        # bvh.ts:30-51
        self.triangleIndices = list(range(self.triangleCount))
        self.nodes = [Node() for _ in range(2 * len(self.triangles) - 1)]
        root = self.nodes[0]
        root.leftChildIndex = 0
        root.primitiveCount = len(self.triangles)
        self.nodesUsed += 1
        self.updateBounds(0)
        old = sys.getrecursionlimit()
        sys.setrecursionlimit(max(old, 10000))
        try:
            self.subdivideSAH(0)
        finally:
            sys.setrecursionlimit(old)

    def updateBounds(self, nodeIndex):                         # bvh.ts:53-65
        node = self.nodes[nodeIndex]
        node.minCorner = [1e30, 1e30, 1e30]
        node.maxCorner = [-1e30, -1e30, -1e30]
        lo, hi = node.minCorner, node.maxCorner
        for i in range(node.primitiveCount):
            tri = self.triangles[self.triangleIndices[node.leftChildIndex + i]]
            for c in tri.corners:
                for k in range(3):
                    v = c[k]
                    if v < lo[k]: lo[k] = v
                    if v > hi[k]: hi[k] = v

    def findBestSplit(self, node):                             # bvh.ts:67-86
        SPLIT_PER_AXIS = 10
        bestCost, bestAxis, bestSplitPosition = 1e30, 0, 0
        for axis in range(3):
            for noSplit in range(1, SPLIT_PER_AXIS):
                splitPercent = noSplit / SPLIT_PER_AXIS
                splitPosition = node.minCorner[axis] * (1 - splitPercent) + node.maxCorner[axis] * splitPercent
                cost = self.SAH(node, axis, splitPosition)
                if cost < bestCost:
                    bestCost, bestAxis, bestSplitPosition = cost, axis, splitPosition
        return bestAxis, bestSplitPosition, bestCost

    def SAH(self, node, axis, splitPosition):                  # bvh.ts:88-110
        left, right = AABB(), AABB()
        nl = nr = 0
        for i in range(node.primitiveCount):
            tri = self.triangles[self.triangleIndices[i + node.leftChildIndex]]
            if float(tri.centroid[axis]) < splitPosition:
                nl += 1
                left.grow(tri.corners[0]); left.grow(tri.corners[1]); left.grow(tri.corners[2])
            else:
                nr += 1
                right.grow(tri.corners[0]); right.grow(tri.corners[1]); right.grow(tri.corners[2])
        return left.surfaceArea() * nl + right.surfaceArea() * nr

    def subdivideSAH(self, nodeIndex):                         # bvh.ts:112-169
        node = self.nodes[nodeIndex]
        if node.primitiveCount < 2:
            return
        axis, splitPosition, subdivisionCost = self.findBestSplit(node)
        parent = AABB()
        parent.grow(node.minCorner)
        parent.grow(node.maxCorner)
        parentCost = parent.surfaceArea() * node.primitiveCount
        if parentCost < subdivisionCost:
            return
        i = node.leftChildIndex
        j = i + node.primitiveCount - 1
        idx = self.triangleIndices
        while i <= j:
            if float(self.triangles[idx[i]].centroid[axis]) < splitPosition:
                i += 1
            else:
                idx[i], idx[j] = idx[j], idx[i]
                j -= 1
        leftCount = i - node.leftChildIndex
        if leftCount == 0 or leftCount == node.primitiveCount:
            return
        leftChildIndex = self.nodesUsed
        self.nodesUsed += 1
        rightChildIndex = self.nodesUsed
        self.nodesUsed += 1
        self.nodes[leftChildIndex].leftChildIndex = node.leftChildIndex
        self.nodes[leftChildIndex].primitiveCount = leftCount
        self.nodes[rightChildIndex].leftChildIndex = i
        self.nodes[rightChildIndex].primitiveCount = node.primitiveCount - leftCount
        node.leftChildIndex = leftChildIndex
        node.primitiveCount = 0
        self.updateBounds(leftChildIndex)
        self.updateBounds(rightChildIndex)
        self.subdivideSAH(leftChildIndex)
        self.subdivideSAH(rightChildIndex)
