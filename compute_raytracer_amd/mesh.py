"""Mesh -- src/rendering-raycast/mesh.ts:7-22: triangles of one OBJ file + their BVH."""
from .acceleration.bvh import BVH
from .model.reader.obj_reader import ObjectReader


class Mesh:
    def __init__(self):
        self.color = None
        self.triangles = None
        self.triangleLookupOffset = 0
        self.rootNodeIndex = 0
        self.bvh = None

    def initialize(self, url, descriptor):                     # mesh.ts:16-21 (url = file path)
        self.triangles = ObjectReader.loadMeshFromObjFile(url, descriptor)
        self.bvh = BVH(self.triangles)
        return self

    def initializeFromText(self, text, descriptor):
        self.triangles = ObjectReader.loadMeshFromObjText(text, descriptor)
        self.bvh = BVH(self.triangles)
        return self
