"""Procedural triangle scenes.  No asset of the reference travels with this repository (its cat /
mousey OBJ files and the mousey texture stay upstream), so tests and `bench.py --config TRI` build
scenes of the same SHAPE from generated OBJ text: tessellated spheres instanced as models plus a
floor quad, through the same loader, SAH builder and top-level builder as a real asset would take
(soup.py, acceleration/bvh.py, instances.py, scene_raytracing.py)."""
import math

import numpy as np

def obj_uv_sphere(rings=6, sectors=8, radius=1.0, centre=(0.0, 0.0, 0.0), quads=True):
    """OBJ text of a UV sphere with v / vt / vn and (optionally) quad faces, which the reader's
    fan triangulation splits in two (obj-reader.ts:103-117).  Faces wind counter-clockwise seen
    from outside (hitTriangle culls back faces, RK:359-362)."""
    v, vt, vn, f = [], [], [], []
    for r in range(rings + 1):
        th = math.pi * r / rings
        for s in range(sectors + 1):
            ph = 2 * math.pi * s / sectors
            n = (math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph))
            v.append((centre[0] + radius * n[0], centre[1] + radius * n[1], centre[2] + radius * n[2]))
            vn.append(n)
            vt.append((s / sectors, 1 - r / rings))
    def idx(r, s):
        return r * (sectors + 1) + s + 1
    for r in range(rings):
        for s in range(sectors):
            a, b, c, d = idx(r, s), idx(r, s + 1), idx(r + 1, s + 1), idx(r + 1, s)
            if quads:
                f.append((a, b, c, d))
            else:
                f.append((a, b, c)); f.append((a, c, d))
    lines = ["v %.9g %.9g %.9g" % p for p in v] + ["vt %.9g %.9g" % t for t in vt] + ["vn %.9g %.9g %.9g" % n for n in vn]
    lines += ["f " + " ".join("%d/%d/%d" % (i, i, i) for i in face) for face in f]
    return "\n".join(lines) + "\n"


def obj_floor(half=1.0):
    """The reference's flat.obj shape: one upward-facing quad."""
    return ("v %g 0.0 %g\nv %g 0.0 %g\nv %g 0.0 %g\nv %g 0.0 %g\n\nvt 0.0 0.0\nvt 1.0 0.0\nvt 1.0 1.0\nvt 0.0 1.0\n\n"
            "vn 0.0 1.0 0.0\n\nf 1/1/1 2/2/1 3/3/1 4/4/1\n") % (-half, half, half, half, half, -half, -half, -half)


def triangle_scene(seed=1, n_models=3, rings=6, sectors=8, spin=True):
    """Meshes: two UV spheres of different tessellation + a floor; models: instances of them,
    translated and rotated about Y, laid out in front of the reference's default camera."""
    from . import Material, SceneRaytracing, load_mesh
    rng = np.random.default_rng(seed)
    meshes = [
        load_mesh(obj_uv_sphere(rings, sectors, 1.0), dict(color=[0.9, 0.5, 0.3, 0.6], alignBottom=True, scale=1.0)),
        load_mesh(obj_uv_sphere(rings + 2, sectors + 3, 1.0, quads=False), dict(color=[0.3, 0.7, 0.9, 1.0], alignBottom=True, scale=0.7)),
        load_mesh(obj_floor(1.0), dict(color=[1.0, 1.0, 1.0, 0.8], alignBottom=False, scale=12)),
    ]
    models = []
    for i in range(n_models):
        pos = [float(rng.uniform(-4, 4)), 0.0, float(rng.uniform(-9, -3))]
        m = dict(meshIndex=i % 2, position=pos, eulers=[0, float(rng.uniform(0, 360)), 0])
        if spin:
            m["eulerSpeed"] = [0, float(rng.uniform(-90, 90)), 0]
        models.append(m)
    models.append(dict(meshIndex=2, position=[0, 0, -5], eulers=[0, 0, 0]))
    scene = SceneRaytracing().createScene([])
    scene.createTriangleScene(meshes, models)
    tex = rng.integers(0, 256, (16, 24, 4), dtype=np.uint8)
    return scene, Material(tex)




def tri_buffers(scene, material):
    """The node buffer as RR builds it: TLAS nodes at 0.., BLAS nodes from tlasNodesMax (RR:212-223);
    slots between tlasNodesUsed and tlasNodesMax stay zero."""
    nodes = np.zeros((scene.node_buffer_length(), 8), dtype=np.float32)
    t = scene.pack_tlas_nodes()
    nodes[:t.shape[0]] = t
    b = scene.pack_blas_nodes()
    nodes[scene.tlasNodesMax:scene.tlasNodesMax + b.shape[0]] = b
    return dict(triangles=scene.pack_triangles(), nodes=nodes, blas=scene.pack_blas(),
                tri_lookup=scene.pack_tri_lookup(), blas_lookup=scene.pack_blas_lookup(), mesh_tex=material.image)
