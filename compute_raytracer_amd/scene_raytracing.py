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
    """Public fields as in scene-raytracing.ts:13-35: `camera`, `light`, `spheres` for sphere
    scenes; `triangles`, `triangleIndices`, `nodes`, `blasList`, `blasIndices`, `tlasNodesMax`,
    `tlasNodesUsed`, `blasNodesUsed`, `meshes`, `models` for the reference's live triangle scene
    (createTriangleScene / createReferenceScene)."""

    def __init__(self):
        self.camera = None
        self.light = None
        self.spheres = []
        # triangle-scene members (scene-raytracing.ts:19-35); empty for sphere scenes
        self.meshes = []
        self.models = []
        self.triangles = []
        self.triangleIndices = []
        self.nodes = []
        self.blasList = []
        self.blasIndices = []
        self.tlasNodesMax = 0
        self.tlasNodesUsed = 0
        self.blasNodesUsed = 0
        self.blasConsumed = False

    def createScene(self, spheres=None):  # scene-raytracing.ts:37-45
        self.camera = Camera([0.0593, 2.692, 3.293], 106, 270)
        self.light = Light(position=[0, 5, 0], lightIntensity=3.0, minIntensity=0.3)
        self.spheres = list(spheres) if spheres is not None else []
        return self

    def update(self, dt):  # scene-raytracing.ts:138-143
        """Spheres are static; a triangle scene advances its models and rebuilds the TLAS and
        every BLAS (boxes + inverse matrices), as the reference does each frame."""
        if self.models:
            for model in self.models:
                model.update(dt)
            self.buildBVH()

    # ---- the reference's live scene type: meshes + models + two-level BVH (SR:47-272) --------
    def createTriangleScene(self, meshes, models):
        """scene-raytracing.ts:37-136 with the mesh list and the model list supplied by the
        caller (the reference hard-codes cat / mousey / flat, see createReferenceScene)."""
        if self.camera is None:
            self.createScene([])
        self.meshes = list(meshes)
        self.triangles = []                                    # SR:75-79
        for mesh in self.meshes:
            mesh.triangleLookupOffset = len(self.triangles)
            self.triangles.extend(mesh.triangles)
        self.triangleIndices = [0] * len(self.triangles)       # SR:82-93
        i = offset = 0
        for mesh in self.meshes:
            for j in range(len(mesh.bvh.triangleIndices)):
                self.triangleIndices[i] = mesh.bvh.triangleIndices[j] + offset
                i += 1
            offset += len(mesh.bvh.triangleIndices)
        self.models = list(models)                             # SR:96-111
        self.tlasNodesMax = 2 * len(self.models) - 1           # SR:114
        self.blasNodesUsed = 0                                 # SR:116-120
        for mesh in self.meshes:
            mesh.rootNodeIndex = self.tlasNodesMax + self.blasNodesUsed
            self.blasNodesUsed += mesh.bvh.nodesUsed
        from .acceleration.node import Node
        self.nodes = [None] * (self.tlasNodesMax + self.blasNodesUsed)   # SR:123-131
        for i in range(self.tlasNodesMax):
            node = Node()
            node.leftChildIndex = 0
            node.primitiveCount = 0
            node.minCorner = [0, 0, 0]
            node.maxCorner = [0, 0, 0]
            self.nodes[i] = node
        self.buildBVH()                                        # SR:133
        self.finalizeBVH()                                     # SR:134
        self.blasConsumed = True
        return self

    def createReferenceScene(self, models_dir):
        """The reference's own scene (SR:47-111): cat, mousey and a flat floor, from the OBJ files
        under src/assets/models (not shipped with this repository)."""
        import os
        from .mesh import Mesh
        from .model.model import Model
        self.createScene([])
        mousey = Mesh().initialize(os.path.join(models_dir, "mousey", "mousey.obj"),
                                   dict(color=[1.0, 1.0, 1.0, 0.3], alignBottom=True, invertYZ=False, scale=0.025))
        cat = Mesh().initialize(os.path.join(models_dir, "cat.obj"),
                                dict(color=[0.8, 0.6, 0.7, 1.0], alignBottom=True, invertYZ=False, scale=0.1))
        flat = Mesh().initialize(os.path.join(models_dir, "flat.obj"),
                                 dict(color=[1.0, 1.0, 1.0, 1.0], alignBottom=False, invertYZ=False, scale=10))
        meshes = [cat, mousey, flat]                           # SR:71
        models = [Model(x, [5 * x - 2.5, 0, 0], [180, 45 * x, 0]) for x in range(2)]   # SR:97-102
        models[1].eulerSpeed = [0, 45, 0]                      # SR:104
        models.append(Model(meshes.index(flat), [0, 0, 0], [0, 0, 0]))                 # SR:107-111
        return self.createTriangleScene(meshes, models)

    def buildBVH(self):                                        # SR:145-179
        from .acceleration.blas import BLAS
        self.tlasNodesUsed = 0
        n = len(self.models)
        self.blasList = [None] * n
        self.blasIndices = [0] * n
        for i in range(self.tlasNodesMax):
            nd = self.nodes[i]
            nd.leftChildIndex = 0
            nd.primitiveCount = 0
            nd.minCorner = [0, 0, 0]
            nd.maxCorner = [0, 0, 0]
        for i, model in enumerate(self.models):
            mesh = self.meshes[model.meshIndex]
            # quirk kept: mesh.bvh.minCorner/maxCorner are the constructor's +-999999 placeholders
            # (bvh.ts:23-25 sets them, nothing updates them), so every BLAS box is huge
            self.blasList[i] = BLAS(mesh.rootNodeIndex, mesh.bvh.minCorner, mesh.bvh.maxCorner, model.model)
            self.blasIndices[i] = i
        root = self.nodes[0]
        root.leftChildIndex = 0
        root.primitiveCount = len(self.blasList)
        self.tlasNodesUsed += 1
        self._updateBounds(0)
        self._subdivide(0)

    def _updateBounds(self, nodeIndex):                        # SR:181-191
        node = self.nodes[nodeIndex]
        node.minCorner = [1e30, 1e30, 1e30]
        node.maxCorner = [-1e30, -1e30, -1e30]
        for i in range(node.primitiveCount):
            blas = self.blasList[self.blasIndices[node.leftChildIndex + i]]
            for k in range(3):
                node.minCorner[k] = min(node.minCorner[k], float(blas.minCorner[k]))
                node.maxCorner[k] = max(node.maxCorner[k], float(blas.maxCorner[k]))

    def _subdivide(self, nodeIndex):                           # SR:193-254
        from . import glmatrix as glm
        node = self.nodes[nodeIndex]
        if node.primitiveCount < 2:
            return
        extent = glm.vec3_subtract(glm.vec3_create(), node.maxCorner, node.minCorner)
        axis = 0
        if float(extent[1]) > float(extent[axis]): axis = 1
        if float(extent[2]) > float(extent[axis]): axis = 2
        splitPosition = node.minCorner[axis] + float(extent[axis]) / 2
        i = node.leftChildIndex
        j = i + node.primitiveCount - 1
        while i <= j:
            if float(self.blasList[self.blasIndices[i]].center[axis]) < splitPosition:
                i += 1
            else:
                self.blasIndices[i], self.blasIndices[j] = self.blasIndices[j], self.blasIndices[i]
                j -= 1
        leftCount = i - node.leftChildIndex
        if leftCount == 0 or leftCount == node.primitiveCount:
            return
        leftChildIndex = self.tlasNodesUsed
        self.tlasNodesUsed += 1
        rightChildIndex = self.tlasNodesUsed
        self.tlasNodesUsed += 1
        self.nodes[leftChildIndex].leftChildIndex = node.leftChildIndex
        self.nodes[leftChildIndex].primitiveCount = leftCount
        self.nodes[rightChildIndex].leftChildIndex = i
        self.nodes[rightChildIndex].primitiveCount = node.primitiveCount - leftCount
        node.leftChildIndex = leftChildIndex
        node.primitiveCount = 0
        self._updateBounds(leftChildIndex)
        self._updateBounds(rightChildIndex)
        self._subdivide(leftChildIndex)
        self._subdivide(rightChildIndex)

    def finalizeBVH(self):                                     # SR:256-272
        for mesh in self.meshes:
            for i in range(mesh.bvh.nodesUsed):
                meshNode = mesh.bvh.nodes[i]
                if meshNode.primitiveCount == 0:
                    meshNode.leftChildIndex += mesh.rootNodeIndex
                else:
                    meshNode.leftChildIndex += mesh.triangleLookupOffset
                self.nodes[mesh.rootNodeIndex + i] = meshNode

    # ---- packing of the triangle scene, as RR:169-229 ----
    def pack_blas(self):                                       # RR:169-174
        a = np.zeros((len(self.blasList), 20), dtype=np.float32)
        for i, b in enumerate(self.blasList):
            a[i, 0:16] = b.inverseModel
            a[i, 16] = np.float32(b.rootNodeIndex)
        return a

    def pack_blas_lookup(self):                                # RR:177-181 (indices as f32)
        return np.asarray(self.blasIndices, dtype=np.float64).astype(np.float32)

    def pack_tlas_nodes(self):                                 # RR:184-192
        return self._pack_nodes(0, self.tlasNodesUsed)

    def pack_blas_nodes(self):                                 # RR:212-223 (written at byte 32*tlasNodesMax)
        return self._pack_nodes(self.tlasNodesMax, self.blasNodesUsed)

    def _pack_nodes(self, first, count):
        a = np.zeros((count, 8), dtype=np.float32)
        for i in range(count):
            nd = self.nodes[first + i]
            a[i, 0:3] = np.asarray(nd.minCorner, dtype=np.float64).astype(np.float32)
            a[i, 3] = np.float32(nd.leftChildIndex)
            a[i, 4:7] = np.asarray(nd.maxCorner, dtype=np.float64).astype(np.float32)
            a[i, 7] = np.float32(nd.primitiveCount)
        return a

    def pack_triangles(self):                                  # RR:198-209
        a = np.zeros((len(self.triangles), 40), dtype=np.float32)
        for i, t in enumerate(self.triangles):
            for corner in range(3):
                a[i, 12 * corner:12 * corner + 3] = np.asarray(t.corners[corner], dtype=np.float64).astype(np.float32)
                a[i, 12 * corner + 4:12 * corner + 7] = np.asarray(t.normals[corner], dtype=np.float64).astype(np.float32)
                a[i, 12 * corner + 8:12 * corner + 10] = np.asarray(t.textures[corner], dtype=np.float64).astype(np.float32)
            a[i, 36:40] = np.asarray(t.color, dtype=np.float64).astype(np.float32)
        return a

    def pack_tri_lookup(self):                                 # RR:225-229
        return np.asarray(self.triangleIndices, dtype=np.float64).astype(np.float32)

    def node_buffer_length(self):                              # RR:149-152: 32 * nodes.length bytes
        return len(self.nodes)

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
