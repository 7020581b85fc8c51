"""SceneRaytracing -- mirror of src/rendering-raycast/scene-raytracing.ts for sphere scenes.

The reference declares `spheres: Sphere[]` (scene-raytracing.ts:18) but never fills it
(SURVEY.md 0.1); this module fills it, either from caller-supplied spheres or with the
deterministic synthetic generator fixed in SURVEY.md 8(d).  Camera and light defaults are
the reference's (scene-raytracing.ts:39-45).
"""
import math

import numpy as np

from .camera import Camera
from .light import Light
from .sphere import Sphere

_M64 = (1 << 64) - 1

# BASELINE.json configs.  seed = 355 + k for Ck; C4 renders C3's scene on 8 GPUs.
BASELINE_CONFIGS = {
    "C1": dict(width=256, height=256, spheres=3, bounces=1, seed=356, skybox=None),
    "C2": dict(width=1920, height=1080, spheres=64, bounces=4, seed=357, skybox=None),
    "C3": dict(width=3840, height=2160, spheres=1024, bounces=8, seed=358, skybox=None),
    "C4": dict(width=3840, height=2160, spheres=1024, bounces=8, seed=358, skybox=None),
    "C5": dict(width=7680, height=4320, spheres=4096, bounces=16, seed=360, skybox="daylight"),
}

# constant sky for C1-C4: six 1x1 rgba8unorm faces of this colour ((0.5,0.7,1.0) in bytes)
CONSTANT_SKY_RGBA = (128, 179, 255, 255)


class SplitMix64:
    def __init__(self, seed):
        self.state = seed & _M64

    def next_u64(self):
        self.state = (self.state + 0x9E3779B97F4A7C15) & _M64
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def uniform(self):
        """24-bit-mantissa uniform in [0,1), exactly representable in f32."""
        return (self.next_u64() >> 40) / 16777216.0

    def range(self, lo, hi):
        return lo + (hi - lo) * self.uniform()


def synthetic_spheres(n, seed):
    """sphere 0 = ground (0,-100,0) r=100 colour 0.8; the rest random inside
    the 30-unit fog range, in front of the reference's default camera.  All arithmetic in f64
    (like JS numbers), rounded to f32 only when packed.  Draw order per sphere:
    x, z, radius-u, lift-u, r, g, b."""
    if n < 1:
        return []
    rng = SplitMix64(seed)
    spheres = [Sphere([0.0, -100.0, 0.0], 100.0, [0.8, 0.8, 0.8])]
    # radius scale (64/N)^(1/3), rounded to f32 first so hosts with different libm agree
    rscale = float(np.float32(math.pow(64.0 / n, 1.0 / 3.0)))
    for _ in range(n - 1):
        x = rng.range(-12.0, 12.0)
        z = rng.range(-26.0, -3.0)
        r = rng.range(0.5, 1.5) * rscale
        r = min(max(r, 0.04), 1.5)
        y = r + rng.range(0.0, 3.0)
        col = [rng.range(0.2, 1.0), rng.range(0.2, 1.0), rng.range(0.2, 1.0)]
        spheres.append(Sphere([x, y, z], r, col))
    return spheres


class SceneRaytracing:
    """Public fields as in scene-raytracing.ts:13-35 that the sphere path reads:
    `camera`, `light`, `spheres`.  The triangle/BVH members (triangles, nodes, blasList, ...)
    belong to SURVEY.md 8(f) row 1 and are not built yet."""

    def __init__(self):
        self.camera = None
        self.light = None
        self.spheres = []

    def createScene(self, spheres=None):  # scene-raytracing.ts:37-45
        self.camera = Camera([0.0593, 2.692, 3.293], 106, 270)
        self.light = Light(position=[0, 5, 0], lightIntensity=3.0, minIntensity=0.3)
        self.spheres = list(spheres) if spheres is not None else []
        return self

    def update(self, dt):  # scene-raytracing.ts:138-143: spheres are static, nothing to rebuild
        return None

    # ---- packing, as RendererRaytracing.recalculateScene does it (RR:157-165) ----
    def pack_params(self, maxBounces):
        p = np.zeros(24, dtype=np.float32)
        p[0:3] = np.asarray(self.camera.position, dtype=np.float64).astype(np.float32)
        p[4:7] = self.camera.forwards
        p[8:11] = self.camera.right
        p[12:15] = self.camera.up
        p[16:19] = np.asarray(self.light.position, dtype=np.float64).astype(np.float32)
        p[19] = np.float32(self.light.lightIntensity)
        p[20] = np.float32(self.light.minIntensity)
        p[21] = np.float32(maxBounces)
        return p

    def pack_spheres(self):
        """8 f32 per sphere, WGSL layout of the commented `struct Sphere` (RK:13-17):
        center @0 (vec3 + pad), color @16, radius @28."""
        a = np.zeros((len(self.spheres), 8), dtype=np.float32)
        for i, s in enumerate(self.spheres):
            a[i, 0:3] = s.center
            a[i, 4:7] = s.color
            a[i, 7] = np.float32(s.radius)
        return a


def synthetic_scene(n, seed):
    return SceneRaytracing().createScene(synthetic_spheres(n, seed))
