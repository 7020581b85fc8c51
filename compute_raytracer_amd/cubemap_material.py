"""CubemapMaterial -- mirror of src/material/cubemap-material.ts:1-80.

Holds six rgba8unorm faces in the reference's order (+X,-X,+Y,-Y,+Z,-Z).  PNG decoding
stays on the host (the browser does it via createImageBitmap in the reference); the faces
are handed to the device through rt_write_cubemap_face.
"""
import numpy as np


class CubemapMaterial:
    def __init__(self):
        self.faces = []  # six (h, w, 4) uint8 arrays

    @classmethod
    def constant(cls, rgba):
        """Six 1x1 faces of one colour (BASELINE configs C1-C4, 'no skybox')."""
        m = cls()
        m.faces = [np.array(rgba, dtype=np.uint8).reshape(1, 1, 4).copy() for _ in range(6)]
        return m

    @classmethod
    def from_cross(cls, image):
        """Cut a 4x3 cross image (H, W, 4) uint8 exactly as sampleCubeFaces does
        (cubemap-material.ts:35-58): sw = W/4, sh = H/3, faces Right(2,1) Left(0,1) Top(1,0)
        Bottom(1,2) Front(1,1) Back(3,1)."""
        img = np.asarray(image)
        if img.ndim != 3 or img.shape[2] != 4 or img.dtype != np.uint8:
            raise ValueError("from_cross: expected (H, W, 4) uint8 RGBA")
        h, w = img.shape[:2]
        if w % 4 or h % 3:
            raise ValueError("from_cross: width must divide by 4 and height by 3")
        sw, sh = w // 4, h // 3
        pos = [(2, 1), (0, 1), (1, 0), (1, 2), (1, 1), (3, 1)]
        m = cls()
        m.faces = [np.ascontiguousarray(img[r * sh:(r + 1) * sh, c * sw:(c + 1) * sw]) for c, r in pos]
        return m

    @classmethod
    def from_png(cls, path):
        from PIL import Image  # host-side decode only
        return cls.from_cross(np.array(Image.open(path).convert("RGBA"), dtype=np.uint8))
