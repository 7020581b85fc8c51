"""BLAS -- src/rendering-raycast/acceleration/blas.ts:3-40: world-space box of a mesh instance
(its 8 object-space corners through the model matrix) and the inverse model matrix."""
from .. import glmatrix as glm


class BLAS:
    def __init__(self, rootNodeIndex, minCorner, maxCorner, model):   # blas.ts:11-39
        self.rootNodeIndex = rootNodeIndex
        self.minCorner = [1e30, 1e30, 1e30]
        self.maxCorner = [-1e30, -1e30, -1e30]
        self.triangleLookupIndex = 0
        lo, hi = minCorner, maxCorner
        corners = [
            [lo[0], lo[1], lo[2]], [lo[0], lo[1], hi[2]], [lo[0], hi[1], lo[2]], [lo[0], hi[1], hi[2]],
            [hi[0], lo[1], lo[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], lo[2]], [hi[0], hi[1], hi[2]],
        ]
        corner = glm.vec3_create()
        for c in corners:
            glm.vec3_transform_mat4(corner, c, model)
            glm.vec3_min(self.minCorner, self.minCorner, corner)
            glm.vec3_max(self.maxCorner, self.maxCorner, corner)
        self.center = glm.vec3_create()
        glm.vec3_add(self.center, self.minCorner, self.maxCorner)
        glm.vec3_div(self.center, self.center, [2, 2, 2])
        self.inverseModel = glm.mat4_create()
        glm.mat4_invert(self.inverseModel, model)
