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


class TriMesh:
    """One mesh of a triangle scene: its soup and its bottom-level tree (what mesh.ts:16-21 loads),
    plus where the scene put them (scene-raytracing.ts:75-79, 116-120)."""
    __slots__ = ("soup", "tree", "lookup_offset", "root_node")

    def __init__(self, soup, tree=None):
        from .acceleration.bvh import build_tree
        self.soup = soup
        self.tree = tree if tree is not None else build_tree(soup)
        self.lookup_offset = 0
        self.root_node = 0


def load_mesh(obj_text, descriptor):
    from .soup import parse_obj
    return TriMesh(parse_obj(obj_text, descriptor))


def load_mesh_file(path, descriptor):
    with open(path, "r", newline="") as f:               # keep '\r': the loader splits on '\n' only
        return load_mesh(f.read(), descriptor)


class SceneRaytracing:
    """`camera`, `light`, `spheres` for sphere scenes (scene-raytracing.ts:13-18).  A triangle scene --
    the reference's live scene type, scene-raytracing.ts:47-272 -- is held as the upload buffers
    themselves: `static` = {triangles (T,40), blas_nodes (B,8), tri_lookup (T,)} packed once,
    `frame` = {blas (M,20), blas_lookup (M,), tlas_nodes (U,8)} rebuilt by update(dt); `meshes` are
    TriMesh (soup + tree arrays), `instances` an Instances array set (instances.py)."""

    def __init__(self):
        self.camera = None
        self.light = None
        self.spheres = []
        self.meshes = []
        self.instances = None
        self.static = None
        self.frame = None
        self.triangleCount = 0
        self.tlasNodesMax = 0
        self.tlasNodesUsed = 0
        self.blasNodesUsed = 0

    @property
    def hasTriangles(self):
        return self.triangleCount > 0

    def createScene(self, spheres=None):  # scene-raytracing.ts:37-45
        self.camera = Camera([0.0593, 2.692, 3.293], 106, 270)
        self.light = Light(position=[0, 5, 0], lightIntensity=3.0, minIntensity=0.3)
        self.spheres = list(spheres) if spheres is not None else []
        return self

    def update(self, dt):  # scene-raytracing.ts:138-143
        """Spheres are static; a triangle scene turns its instances and rebuilds the per-frame buffers
        (instance matrices, their inverses, the top-level tree), as the reference does each frame."""
        if self.instances is not None and len(self.instances):
            self.instances.turn(dt)
            self.buildTopLevel()

    # ---- the reference's live scene type: meshes + instances + two-level tree (SR:47-272) --------
    def createTriangleScene(self, meshes, instances):
        """meshes: TriMesh list; instances: Instances (or the records Instances.from_records takes)."""
        from .instances import Instances
        if self.camera is None:
            self.createScene([])
        self.meshes = list(meshes)
        self.instances = instances if isinstance(instances, Instances) else Instances.from_records(instances)
        at = 0
        for mesh in self.meshes:                               # SR:75-79
            mesh.lookup_offset = at
            at += mesh.soup.count
        self.triangleCount = at
        self.tlasNodesMax = 2 * len(self.instances) - 1        # SR:114
        nodes = 0
        for mesh in self.meshes:                               # SR:116-120
            mesh.root_node = self.tlasNodesMax + nodes
            nodes += mesh.tree.used
        self.blasNodesUsed = nodes
        z40, z8 = np.zeros((0, 40), np.float32), np.zeros((0, 8), np.float32)
        self.static = dict(
            triangles=np.concatenate([m.soup.pack() for m in self.meshes] + [z40]),                       # RR:198-209
            blas_nodes=np.concatenate([m.tree.nodes(m.root_node, m.lookup_offset) for m in self.meshes] + [z8]),   # RR:212-223, SR:256-272
            tri_lookup=np.concatenate([(m.tree.order + m.lookup_offset).astype(np.float64) for m in self.meshes]
                                      + [np.zeros(0)]).astype(np.float32))                                # RR:225-229, SR:82-93
        self.buildTopLevel()
        return self

    def createReferenceScene(self, models_dir, mousey_xz=(2.5, 0.0), cat_xz=(-2.5, 0.0), mousey_yaw=45.0):
        """The reference's own scene (SR:47-111): cat, mousey and a flat floor, from the OBJ files
        under src/assets/models (they do not travel with this repository).  The keyword arguments are
        the dat.GUI state of src/app.ts:97-113 (positions) and the spin angle reached (SR:104)."""
        import os
        self.createScene([])
        mousey = load_mesh_file(os.path.join(models_dir, "mousey", "mousey.obj"),
                                dict(color=[1.0, 1.0, 1.0, 0.3], alignBottom=True, invertYZ=False, scale=0.025))
        cat = load_mesh_file(os.path.join(models_dir, "cat.obj"),
                             dict(color=[0.8, 0.6, 0.7, 1.0], alignBottom=True, invertYZ=False, scale=0.1))
        flat = load_mesh_file(os.path.join(models_dir, "flat.obj"),
                              dict(color=[1.0, 1.0, 1.0, 1.0], alignBottom=False, invertYZ=False, scale=10))
        records = [dict(meshIndex=0, position=[cat_xz[0], 0, cat_xz[1]], eulers=[180, 0, 0]),              # SR:97-102
                   dict(meshIndex=1, position=[mousey_xz[0], 0, mousey_xz[1]], eulers=[180, mousey_yaw, 0], eulerSpeed=[0, 45, 0]),   # SR:104
                   dict(meshIndex=2, position=[0, 0, 0], eulers=[0, 0, 0])]                                # SR:107-111
        return self.createTriangleScene([cat, mousey, flat], records)                                      # SR:71

    # ---- a triangle scene as data: everything a renderer and update(dt) need, without the OBJ files ----
    def to_packed(self):
        """dict of arrays: the static upload buffers, the instance records and what the per-frame rebuild reads of
        each mesh (root node, tree-level box), camera and light.  np.savez-able; from_packed restores the scene."""
        inst = self.instances
        return dict(
            triangles=self.static["triangles"], blas_nodes=self.static["blas_nodes"], tri_lookup=self.static["tri_lookup"],
            mesh_root=np.array([m.root_node for m in self.meshes], dtype=np.int64),
            mesh_box_lo=np.array([m.tree.box_lo for m in self.meshes], dtype=np.float64),
            mesh_box_hi=np.array([m.tree.box_hi for m in self.meshes], dtype=np.float64),
            inst_mesh=inst.mesh_index, inst_position=inst.position, inst_eulers=inst.eulers, inst_speed=inst.euler_speed,
            camera_position=np.array(self.camera.position, dtype=np.float64), camera_eulers=np.array(self.camera.eulers, dtype=np.float32),
            light=np.array(list(self.light.position) + [self.light.lightIntensity, self.light.minIntensity], dtype=np.float64))

    @classmethod
    def from_packed(cls, d):
        from types import SimpleNamespace
        from .instances import Instances
        s = cls().createScene([])
        s.camera.position = [float(v) for v in d["camera_position"]]
        s.camera.eulers = np.asarray(d["camera_eulers"], dtype=np.float32)
        s.camera.update()
        lt = [float(v) for v in d["light"]]
        s.light = Light(position=lt[0:3], lightIntensity=lt[3], minIntensity=lt[4])
        s.meshes = [SimpleNamespace(root_node=int(r), tree=SimpleNamespace(box_lo=np.asarray(lo), box_hi=np.asarray(hi)))
                    for r, lo, hi in zip(d["mesh_root"], d["mesh_box_lo"], d["mesh_box_hi"])]
        s.instances = Instances(d["inst_mesh"], d["inst_position"], d["inst_eulers"], d["inst_speed"])
        s.static = dict(triangles=np.ascontiguousarray(d["triangles"], dtype=np.float32),
                        blas_nodes=np.ascontiguousarray(d["blas_nodes"], dtype=np.float32),
                        tri_lookup=np.ascontiguousarray(d["tri_lookup"], dtype=np.float32))
        s.triangleCount = s.static["triangles"].shape[0]
        s.tlasNodesMax = 2 * len(s.instances) - 1
        s.blasNodesUsed = s.static["blas_nodes"].shape[0]
        s.buildTopLevel()
        return s

    def buildTopLevel(self):                                   # SR:145-254 into the buffers of RR:169-192
        from .instances import invert_mat4, top_level, world_boxes
        inst = self.instances
        mats = inst.matrices()
        blas = np.zeros((len(inst), 20), dtype=np.float32)
        blas[:, 0:16] = invert_mat4(mats)
        blas[:, 16] = [self.meshes[k].root_node for k in inst.mesh_index]
        box_lo = np.array([self.meshes[k].tree.box_lo for k in inst.mesh_index]).reshape(-1, 3)
        box_hi = np.array([self.meshes[k].tree.box_hi for k in inst.mesh_index]).reshape(-1, 3)
        lo, hi, centre = world_boxes(mats, box_lo, box_hi)
        nodes, lookup = top_level(lo, hi, centre)
        self.tlasNodesUsed = nodes.shape[0]
        self.frame = dict(blas=blas, blas_lookup=lookup, tlas_nodes=nodes)

    # ---- the upload buffers by the names RendererRaytracing asks for (RR:169-229) ----
    def pack_blas(self): return self.frame["blas"]
    def pack_blas_lookup(self): return self.frame["blas_lookup"]
    def pack_tlas_nodes(self): return self.frame["tlas_nodes"]
    def pack_blas_nodes(self): return self.static["blas_nodes"]
    def pack_triangles(self): return self.static["triangles"]
    def pack_tri_lookup(self): return self.static["tri_lookup"]

    def node_buffer_length(self):                              # RR:149-152: 32 * nodes.length bytes
        return self.tlasNodesMax + self.blasNodesUsed

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
