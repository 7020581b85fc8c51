"""Model -- src/rendering-raycast/model/model.ts:4-38: a mesh instance = translate * rotateY.
(eulers[0] and eulers[2] are carried but never used by calculateTransform, model.ts:33-37.)"""
from .. import glmatrix as glm
from ..camera import deg2rad


class Model:
    def __init__(self, meshIndex, position, eulers, eulerSpeed=None):   # model.ts:11-17
        self.meshIndex = meshIndex
        self.position = list(position)
        self.eulers = list(eulers)
        self.eulerSpeed = list(eulerSpeed) if eulerSpeed else [0, 0, 0]
        self.calculateTransform()

    def update(self, dt):                                      # model.ts:19-31
        rotation = glm.vec3_mul(glm.vec3_create(), self.eulerSpeed, [dt, dt, dt])
        glm.vec3_add(self.eulers, self.eulers, rotation)
        for k in range(3):
            if self.eulers[k] > 360: self.eulers[k] -= 360
        for k in range(3):
            if self.eulers[k] < -360: self.eulers[k] += 360
        self.calculateTransform()

    def calculateTransform(self):                              # model.ts:33-37
        self.model = glm.mat4_create()
        glm.mat4_translate(self.model, self.model, self.position)
        glm.mat4_rotate_y(self.model, self.model, deg2rad(self.eulers[1]))
