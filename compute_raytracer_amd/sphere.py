"""Sphere -- mirror of src/rendering-raycast/model/sphere.ts:1-10.

`center` and `color` are Float32Array in the reference (rounded to f32 at
construction), `radius` stays a JS number until packed.
"""
import numpy as np


class Sphere:
    __slots__ = ("center", "radius", "color")

    def __init__(self, center, radius, color):
        self.center = np.asarray(center, dtype=np.float64).astype(np.float32)
        self.radius = float(radius)
        self.color = np.asarray(color, dtype=np.float64).astype(np.float32)
        if self.center.shape != (3,) or self.color.shape != (3,):
            raise ValueError("Sphere: center and color must have 3 components")
