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
    def synthetic_daylight(cls, face=512, seed=355):
        """A procedural stand-in for the reference's daylight-skybox.png (2048x1536 cross = six
        512x512 faces), which does not travel with this repository: vertical sky gradient, a sun
        disc and value noise as clouds.  Deterministic; every texel differs from its neighbours,
        so the bilinear path (BASELINE config C5, 'skybox sample') is fully exercised."""
        rng = np.random.default_rng(seed)
        m = cls()
        # direction of every texel centre, WebGPU face conventions (see oracle cube_sample)
        t = (np.arange(face) + 0.5) / face * 2 - 1
        sc, tc = np.meshgrid(t, t)
        one = np.ones_like(sc)
        dirs = [( one, -tc, -sc), (-one, -tc,  sc), ( sc,  one,  tc), ( sc, -one, -tc), ( sc, -tc,  one), (-sc, -tc, -one)]
        sun = np.array([0.35, 0.75, -0.55]); sun /= np.linalg.norm(sun)
        coarse = rng.random((6, 17, 17))
        for f, (x, y, z) in enumerate(dirs):
            n = np.sqrt(x * x + y * y + z * z)
            x, y, z = x / n, y / n, z / n
            up = np.clip(y * 0.5 + 0.5, 0, 1)
            base = np.stack([0.25 + 0.35 * (1 - up), 0.45 + 0.35 * (1 - up), 0.75 + 0.25 * (1 - up)], axis=-1)
            ground = np.stack([0.35 + 0 * up, 0.33 + 0 * up, 0.30 + 0 * up], axis=-1)
            col = np.where((y < 0)[..., None], ground, base)
            c = np.clip(x * sun[0] + y * sun[1] + z * sun[2], 0, 1)
            col = col + (c ** 256)[..., None] * np.array([1.0, 0.95, 0.8]) + (c ** 8)[..., None] * 0.15
            # bilinear upsample of a 17x17 noise grid = soft clouds, strongest near the horizon
            g = coarse[f]
            gi = (np.arange(face) + 0.5) / face * 16
            i0 = np.floor(gi).astype(int); w = gi - i0
            rows = g[i0][:, None, :] * (1 - w)[:, None, None] + g[i0 + 1][:, None, :] * w[:, None, None]
            rows = rows[:, 0, :]
            cloud = rows[:, i0] * (1 - w)[None, :] + rows[:, i0 + 1] * w[None, :]
            col = col + ((cloud - 0.5) * 0.25 * (y > 0))[..., None]
            fine = rng.random((face, face, 1)) * 0.02
            rgba = np.concatenate([np.clip(col + fine, 0, 1), np.ones((face, face, 1))], axis=-1)
            m.faces.append(np.ascontiguousarray((rgba * 255 + 0.5).astype(np.uint8)))
        return m

    @classmethod
    def from_png(cls, path):
        from PIL import Image  # host-side decode only
        return cls.from_cross(np.array(Image.open(path).convert("RGBA"), dtype=np.uint8))
