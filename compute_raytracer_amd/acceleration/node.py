"""Node -- src/rendering-raycast/acceleration/node.ts:3-8."""


class Node:
    __slots__ = ("minCorner", "leftChildIndex", "maxCorner", "primitiveCount")

    def __init__(self):
        self.minCorner = None
        self.leftChildIndex = 0
        self.maxCorner = None
        self.primitiveCount = 0
