"""AABB -- src/rendering-raycast/acceleration/aabb.ts:3-21 (vec3.fromValues: f32 storage)."""
from .. import glmatrix as glm


class AABB:
    def __init__(self):
        self.min = glm.vec3_from_values(1e30, 1e30, 1e30)
        self.max = glm.vec3_from_values(-1e30, -1e30, -1e30)

    def grow(self, corner):                                   # aabb.ts:12-15
        glm.vec3_min(self.min, self.min, corner)
        glm.vec3_max(self.max, self.max, corner)

    def surfaceArea(self):                                    # aabb.ts:17-20
        e = glm.vec3_subtract(glm.vec3_create(), self.max, self.min)
        e0, e1, e2 = float(e[0]), float(e[1]), float(e[2])
        return 2 * (e0 * e1 + e1 * e2 + e2 * e0)
