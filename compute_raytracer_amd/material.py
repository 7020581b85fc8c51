"""Material -- src/material/material.ts:1-66: the 2-D mesh texture (`meshTex`, binding 8),
rgba8unorm.  Decoding stays on the host; the pixels go to the device through
rt_write_mesh_texture."""
import numpy as np


class Material:
    def __init__(self, image=None):
        self.image = None if image is None else np.ascontiguousarray(image, dtype=np.uint8)

    @classmethod
    def white(cls):
        return cls(np.full((1, 1, 4), 255, dtype=np.uint8))

    @classmethod
    def from_png(cls, path):
        from PIL import Image
        return cls(np.array(Image.open(path).convert("RGBA"), dtype=np.uint8))
