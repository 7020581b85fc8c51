"""Triangle -- src/rendering-raycast/model/triangle.ts:4-45.  corners/normals/textures are the
plain JS arrays the OBJ reader produced (f64); centroid is a gl-matrix vec3 (f32 stores)."""
from .. import glmatrix as glm


class Triangle:
    __slots__ = ("corners", "normals", "textures", "color", "centroid")

    def __init__(self):                                        # triangle.ts:29-35
        self.corners = []
        self.textures = []
        self.normals = []
        self.color = [0, 0, 0, 0]
        self.centroid = [0, 0, 0]

    def calculateCentroid(self):                               # triangle.ts:37-44
        self.centroid = glm.vec3_create()
        glm.vec3_add(self.centroid, self.centroid, self.corners[0])
        glm.vec3_add(self.centroid, self.centroid, self.corners[1])
        glm.vec3_add(self.centroid, self.centroid, self.corners[2])
        glm.vec3_div(self.centroid, self.centroid, [3, 3, 3])
        return self.centroid
