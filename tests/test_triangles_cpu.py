"""The reference's live scene type on the CPU side: OBJ reader, SAH BVH builder, TLAS build,
gl-matrix restatement, buffer packing (host mirrors) and the oracle's TLAS/BLAS traversal
(RK:168-410), which is cross-checked against a brute-force numpy evaluation of hitTriangle over
every triangle of every model instance."""
import math
import os

import numpy as np
import pytest

import compute_raytracer_amd as rt
from compute_raytracer_amd import glmatrix as glm
from compute_raytracer_amd.instances import Instances, invert_mat4, top_level, world_boxes
from compute_raytracer_amd.soup import js_parse_float, parse_obj
from helpers import obj_floor, obj_uv_sphere, tri_buffers, triangle_scene

F = np.float32
REF_MODELS = "/root/reference/src/assets/models"


# ---- gl-matrix restatement (scalar form, kept as the yardstick of the array forms in instances.py) -----------
def test_mat4_translate_rotate_invert():
    m = glm.mat4_create()
    glm.mat4_translate(m, m, [2.5, 0, -1])
    glm.mat4_rotate_y(m, m, math.radians(45))
    assert m.dtype == F
    c = math.cos(math.radians(45))
    assert np.allclose(m.reshape(4, 4).T, [[c, 0, c, 2.5], [0, 1, 0, 0], [-c, 0, c, -1], [0, 0, 0, 1]], atol=1e-7)
    inv = glm.mat4_invert(glm.mat4_create(), m)
    assert np.allclose((inv.reshape(4, 4).T.astype(np.float64) @ m.reshape(4, 4).T.astype(np.float64)), np.eye(4), atol=1e-6)
    assert glm.mat4_invert(glm.mat4_create(), np.zeros(16, F)) is None      # `if (!det) return null`
    p = glm.vec3_transform_mat4(glm.vec3_create(), [1, 2, 3], m)
    assert np.allclose(p, [c * 1 + c * 3 + 2.5, 2, -c * 1 + c * 3 - 1], atol=1e-6)


def test_f32_stores_versus_plain_arrays():
    out32 = glm.vec3_add(glm.vec3_create(), [0.1, 0.2, 0.3], [1, 1, 1])
    out64 = glm.vec3_add([0, 0, 0], [0.1, 0.2, 0.3], [1, 1, 1])
    assert out32.dtype == F and out64 == [1.1, 1.2, 1.3] and float(out32[0]) != 1.1


def test_instance_arrays_equal_scalar_glmatrix():
    """instances.py forms every instance's matrix, inverse, world box and centre at once; the scalar
    gl-matrix restatement, one call per stored value, must give the same bits."""
    rng = np.random.default_rng(7)
    m = 9
    inst = Instances(np.zeros(m, int), rng.uniform(-8, 8, (m, 3)), rng.uniform(-400, 400, (m, 3)), rng.uniform(-90, 90, (m, 3)))
    inst.turn(0.37); inst.turn(1.9)
    assert np.all(np.abs(inst.eulers) <= 720)
    mats = inst.matrices()
    inv = invert_mat4(mats)
    box_lo = np.tile([999999.0] * 3, (m, 1)); box_hi = -box_lo
    lo, hi, centre = world_boxes(mats, box_lo, box_hi)
    for k in range(m):
        ref = glm.mat4_create()
        glm.mat4_translate(ref, ref, inst.position[k])
        glm.mat4_rotate_y(ref, ref, inst.eulers[k, 1] * math.pi / 180)
        assert np.array_equal(ref.view(np.uint32), mats[k].view(np.uint32))
        assert np.array_equal(glm.mat4_invert(glm.mat4_create(), ref).view(np.uint32), inv[k].view(np.uint32))
        mn, mx, corner = [1e30] * 3, [-1e30] * 3, glm.vec3_create()
        for c in range(8):
            pt = [(-999999.0 if c & 4 else 999999.0), (-999999.0 if c & 2 else 999999.0), (-999999.0 if c & 1 else 999999.0)]
            glm.vec3_transform_mat4(corner, pt, ref)
            glm.vec3_min(mn, mn, corner); glm.vec3_max(mx, mx, corner)
        assert mn == list(lo[k]) and mx == list(hi[k])
        cen = glm.vec3_div(glm.vec3_create(), glm.vec3_add(glm.vec3_create(), mn, mx), [2, 2, 2])
        assert np.array_equal(cen.view(np.uint32), centre[k].view(np.uint32))
    assert np.array_equal(invert_mat4(np.zeros((1, 16), F))[0], glm.mat4_create())      # singular: identity stays


# ---- OBJ text -> soup ----------------------------------------------------------------------------------
def test_parse_float_is_js_parse_float():
    assert js_parse_float("+1.0") == 1.0 and js_parse_float("-2.5e1x") == -25.0 and js_parse_float("3\r") == 3.0
    assert js_parse_float(".5") == 0.5 and js_parse_float("7.") == 7.0 and js_parse_float("-Infinity") == -math.inf
    assert math.isnan(js_parse_float("abc")) and math.isnan(js_parse_float("")) and math.isnan(js_parse_float(None))


def test_soup_fan_triangulation_centering_and_scale():
    s = parse_obj(obj_floor(1.0), dict(color=[1, 1, 1, 1], scale=10))
    assert s.count == 2                                     # quad -> fan of 2 (obj-reader.ts:103-117)
    pts = s.position.reshape(-1, 3)
    assert pts.min() == -10 and pts.max() == 10 and np.all(pts[:, 1] == 0)
    assert np.array_equal(s.position[0, 0], s.position[1, 0])         # both fans start at vertex 1
    assert np.allclose(s.centroid[0], s.position[0].mean(axis=0), atol=1e-6) and s.centroid.dtype == F
    sph = parse_obj(obj_uv_sphere(4, 6, 2.0, centre=(5, 5, 5)), dict(color=[1, 0, 0, 1], alignBottom=True, scale=0.5))
    assert sph.count == 4 * 6 * 2
    pts = sph.position.reshape(-1, 3)
    assert abs(pts[:, 1].min()) < 1e-6 and abs(pts[:, 1].max() - 2.0) < 1e-6      # bottom aligned, scaled
    assert abs(pts[:, 0].min() + 1.0) < 1e-6 and abs(pts[:, 0].max() - 1.0) < 1e-6
    rec = sph.pack()
    assert rec.shape == (sph.count, 40) and rec.dtype == F
    assert np.array_equal(rec[:, 12:15], sph.position[:, 1].astype(F)) and np.array_equal(rec[:, 28:31], sph.normal[:, 2].astype(F))
    assert np.array_equal(rec[:, 8:10], sph.uv[:, 0].astype(F)) and np.all(rec[:, 36:40] == [1, 0, 0, 1])
    assert np.all(rec[:, [3, 7, 10, 11]] == 0)


def test_soup_invert_yz_swaps_axes_and_winding():
    a = parse_obj(obj_uv_sphere(3, 4), dict(color=[1, 1, 1, 1]))
    b = parse_obj(obj_uv_sphere(3, 4), dict(color=[1, 1, 1, 1], invertYZ=True))
    assert np.allclose(a.position[0, 0][[0, 2, 1]], b.position[0, 0])
    # face corners 2 and 3 are swapped as well (the swizzle indexes the FACE fields too)
    assert np.allclose(a.position[0, 1][[0, 2, 1]], b.position[0, 2])


def test_soup_loader_quirks():
    # '\r' line ends, a doubled blank (an empty field = NaN), a face that names a vertex not read yet
    s = parse_obj("v 0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nvt 0 0\r\nvn 0 0 1\r\nf 1/1/1 2/1/1 3/1/1\r\n", dict(color=[1, 1, 1, 1]))
    assert s.count == 1 and np.array_equal(s.position[0, 1], [0.5, -0.5, 0])
    s = parse_obj("v 0  0 0\nv 1 1 1\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 2/1/1\n", dict(color=[1, 1, 1, 1]))
    assert np.isnan(s.position[0, 0, 1])
    with pytest.raises(ValueError):
        parse_obj("v 0 0 0\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1\nv 1 0 0\nv 0 1 0\n", dict(color=[1, 1, 1, 1]))
    # maxima are f32 as found, minima f64: the shift of a cloud [0, 0.1] is f32(f32(0 + f32(0.1)) / 2)
    s = parse_obj("v 0 0 0\nv 0.1 0 0\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 2/1/1\n", dict(color=[1, 1, 1, 1]))
    assert s.position[0, 0, 0] == -float(F(float(F(0.1)) / 2))


@pytest.mark.skipif(not os.path.isdir(REF_MODELS), reason="reference assets not present")
def test_reference_scene_triangle_count():
    """The reference's own scene: 428 + 12,174 + 2 = 12,604 triangles, the 'Primitive count' of
    info/sample_settings.png -- the one number of the reference that pins the OBJ loader."""
    counts = [parse_obj(open(os.path.join(REF_MODELS, f), newline="").read(), d).count for f, d in
              (("cat.obj", dict(color=[.8, .6, .7, 1], alignBottom=True, scale=0.1)),
               (os.path.join("mousey", "mousey.obj"), dict(color=[1, 1, 1, .3], alignBottom=True, scale=0.025)),
               ("flat.obj", dict(color=[1, 1, 1, 1], scale=10)))]
    assert counts == [428, 12174, 2]


# ---- bottom-level and top-level trees ------------------------------------------------------------------------
def test_sah_tree_invariants():
    soup = parse_obj(obj_uv_sphere(8, 12), dict(color=[1, 1, 1, 1]))
    t = rt.build_tree(soup)
    assert sorted(t.order) == list(range(soup.count)) and 1 <= t.used <= 2 * soup.count - 1
    assert list(t.box_lo) == [999999] * 3 and list(t.box_hi) == [-999999] * 3      # quirk: never computed (bvh.ts:23-25)
    seen = set()
    def walk(i, depth):
        if t.count[i] == 0:
            l = int(t.first[i])
            for ch in (l, l + 1):
                assert np.all(t.lo[ch] >= t.lo[i]) and np.all(t.hi[ch] <= t.hi[i])
            return max(walk(l, depth + 1), walk(l + 1, depth + 1))
        for j in range(int(t.count[i])):
            tri = int(t.order[t.first[i] + j])
            seen.add(tri)
            assert np.all(soup.position[tri] >= t.lo[i]) and np.all(soup.position[tri] <= t.hi[i])
        return depth
    depth = walk(0, 0)
    assert seen == set(range(soup.count)) and depth < 20
    rec = t.nodes(100, 5000)                                 # rebased records: children +100, leaf runs +5000
    inner = t.count == 0
    assert np.array_equal(rec[inner, 3], (t.first[inner] + 100).astype(F)) and np.array_equal(rec[~inner, 3], (t.first[~inner] + 5000).astype(F))


def test_top_level_median_split():
    lo = np.array([[0, 0, 0], [10, 0, 0], [20, 0, 0], [21, 0, 0]], float); hi = lo + 1
    centre = ((lo + hi) / 2).astype(F)
    nodes, lookup = top_level(lo, hi, centre)
    assert nodes.shape == (7, 8) and sorted(lookup) == [0, 1, 2, 3]
    assert nodes[0, 7] == 0 and nodes[0, 3] == 1 and list(nodes[0, 0:3]) == [0, 0, 0] and list(nodes[0, 4:7]) == [22, 1, 1]
    leaves = nodes[nodes[:, 7] > 0]
    assert leaves[:, 7].sum() == 4
    one, _ = top_level(lo[:1], hi[:1], centre[:1])
    assert one.shape == (1, 8) and one[0, 7] == 1
    same, _ = top_level(lo[[0, 0, 0]], hi[[0, 0, 0]], centre[[0, 0, 0]])          # nothing separates: one leaf of three
    assert same.shape == (1, 8) and same[0, 7] == 3


def test_scene_layout_and_packing():
    scene, mat = triangle_scene(seed=3, n_models=3)
    n_models = len(scene.instances)
    assert scene.tlasNodesMax == 2 * n_models - 1 and scene.tlasNodesUsed <= scene.tlasNodesMax
    assert sorted(scene.pack_blas_lookup()) == list(range(n_models))
    assert sorted(scene.pack_tri_lookup()) == list(range(scene.triangleCount))
    b = tri_buffers(scene, mat)
    assert b["nodes"].shape[0] == scene.tlasNodesMax + scene.blasNodesUsed
    assert b["triangles"].shape == (scene.triangleCount, 40) and b["blas"].shape == (n_models, 20)
    # BLAS root indices point past the TLAS slots; inner BLAS nodes were rebased (SR:256-272)
    assert all(b["blas"][i, 16] >= scene.tlasNodesMax for i in range(n_models))
    s0 = scene.meshes[0].soup
    assert np.array_equal(b["triangles"][0, 0:3], s0.position[0, 0].astype(F))
    assert np.array_equal(b["triangles"][0, 36:40], s0.color.astype(F))
    # TLAS boxes: the +-999999 placeholders through the model matrix (quirk kept)
    assert b["nodes"][0, 0] < -9e5 and b["nodes"][0, 4] > 9e5
    # update(dt) rebuilds instance matrices and the top-level tree
    before = scene.pack_blas().copy()
    scene.update(0.5)
    assert not np.array_equal(before, scene.pack_blas())


# ---- oracle traversal against brute force ----------------------------------------------------------------
def _brute_force(buffers, origins, dirs):
    """min over every (model instance, triangle) of hitTriangle's t, same fp32 arithmetic as RK:344-380,
    without any BVH: what the traversal must find."""
    tri = buffers["triangles"].reshape(-1, 40)
    best = np.full(origins.shape[0], F(9999.0), F)
    found = np.zeros(origins.shape[0], bool)
    nodes = buffers["nodes"].reshape(-1, 8)
    lookup = buffers["tri_lookup"]

    def leaves(i, out):
        cnt, left = int(nodes[i, 7]), int(nodes[i, 3])
        if cnt == 0:
            leaves(left, out); leaves(left + 1, out)
        else:
            out.extend(int(lookup[left + j]) for j in range(cnt))
    for blas in buffers["blas"].reshape(-1, 20):
        m = blas[:16]
        ids = []
        leaves(int(blas[16]), ids)
        def xf(p, w):
            return [((m[r] * p[:, 0] + m[4 + r] * p[:, 1]) + m[8 + r] * p[:, 2]) + m[12 + r] * F(w) for r in range(3)]
        o, d = xf(origins, 1.0), xf(dirs, 0.0)
        for ti in ids:
            t = tri[ti]
            A, B, C = t[0:3], t[12:15], t[24:27]
            e1, e2 = B - A, C - A
            rx = d[1] * e2[2] - e2[1] * d[2]; ry = d[2] * e2[0] - e2[2] * d[0]; rz = d[0] * e2[1] - e2[0] * d[1]
            det = (e1[0] * rx + e1[1] * ry) + e1[2] * rz
            s = [o[0] - A[0], o[1] - A[1], o[2] - A[2]]
            u = (s[0] * rx + s[1] * ry) + s[2] * rz
            cx = s[1] * e1[2] - e1[1] * s[2]; cy = s[2] * e1[0] - e1[2] * s[0]; cz = s[0] * e1[1] - e1[0] * s[1]
            v = (d[0] * cx + d[1] * cy) + d[2] * cz
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = F(1.0) / det
                tt = inv * ((e2[0] * cx + e2[1] * cy) + e2[2] * cz)
            ok = (det >= F(0.00001)) & (u >= 0) & (u <= det) & (v >= 0) & (u + v <= det) & (tt > F(0.001)) & (tt < best)
            best = np.where(ok, tt, best)
            found |= ok
    return np.where(found, best, F(-1.0))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_traversal_finds_the_brute_force_nearest_hit(oracle, seed):
    scene, mat = triangle_scene(seed=seed, n_models=3, rings=5, sectors=6)
    b = tri_buffers(scene, mat)
    p = scene.pack_params(1)
    W, H = 48, 32
    dirs = np.array([oracle.ray_dir(p, W, H, x, y) for y in range(H) for x in range(W)], F)
    origins = np.tile(p[0:3], (dirs.shape[0], 1)).astype(F)
    rng = np.random.default_rng(seed)
    extra_o = rng.uniform(-6, 6, (300, 3)).astype(F); extra_o[:, 1] = np.abs(extra_o[:, 1]) + F(0.5)
    extra_d = rng.normal(size=(300, 3)).astype(F)
    extra_d /= np.linalg.norm(extra_d, axis=1, keepdims=True).astype(F)
    origins = np.vstack([origins, extra_o]); dirs = np.vstack([dirs, extra_d]).astype(F)
    got = oracle.trace_tri_rays(b, origins, dirs)
    want = _brute_force(b, origins, dirs)
    assert (got >= 0).sum() > 200
    assert np.array_equal(got, want)


def test_triangle_kats(oracle, constant_sky):
    """One triangle facing the camera (RK:344-393): analytic t, back-face culling, the tMin rule."""
    def scene_with(tri_pts, tex=None):
        t = np.zeros((1, 40), F)
        for k, pt in enumerate(tri_pts):
            t[0, 12 * k:12 * k + 3] = pt
            t[0, 12 * k + 4:12 * k + 7] = [0, 0, 1]
            t[0, 12 * k + 8:12 * k + 10] = [(0, 0), (1, 0), (0, 1)][k]
        t[0, 36:40] = [1, 1, 1, 1]
        nodes = np.zeros((2, 8), F)
        nodes[0] = [-10, -10, -10, 0, 10, 10, 10, 1]          # TLAS root: 1 BLAS
        nodes[1] = [-10, -10, -10, 0, 10, 10, 10, 1]          # BLAS root leaf: 1 triangle
        blas = np.zeros((1, 20), F); blas[0, [0, 5, 10, 15]] = 1; blas[0, 16] = 1
        return dict(triangles=t, nodes=nodes, blas=blas, tri_lookup=np.zeros(1, F), blas_lookup=np.zeros(1, F),
                    mesh_tex=np.full((1, 1, 4), 255, np.uint8) if tex is None else tex)
    front = scene_with([(-1, -1, -5), (1, -1, -5), (0, 1, -5)])          # counter-clockwise seen from +z
    back = scene_with([(-1, -1, -5), (0, 1, -5), (1, -1, -5)])
    o = np.array([[0, 0, 0]], F); d = np.array([[0, 0, -1]], F)
    assert oracle.trace_tri_rays(front, o, d)[0] == F(5.0)
    assert oracle.trace_tri_rays(back, o, d)[0] == F(-1.0)                 # det < 1e-5: culled (RK:359-362)
    assert oracle.trace_tri_rays(front, np.array([[0, 0, -4.9995]], F), d)[0] == F(-1.0)   # t < tMin 0.001
    assert oracle.trace_tri_rays(front, np.array([[5, 0, 0]], F), d)[0] == F(-1.0)         # outside
    assert oracle.trace_tri_rays(front, o, np.array([[0, 0, 1]], F))[0] == F(-1.0)         # behind


def test_mesh_texture_sampler_repeat_u_clamp_v(oracle, constant_sky):
    """meshTex is sampled with the CUBE MAP's sampler (RR:345-347): U repeats, V clamps."""
    scene, mat = triangle_scene(seed=5, n_models=1)
    tex = np.zeros((2, 2, 4), np.uint8); tex[..., 3] = 255
    tex[0, 0, 0], tex[0, 1, 0], tex[1, 0, 0], tex[1, 1, 0] = 0, 100, 200, 250
    b = tri_buffers(scene, rt.Material(tex))
    p = scene.pack_params(2)
    a, _, rays = oracle.render_tri(p, b, constant_sky.faces, 40, 24)
    assert rays > 40 * 24 and a[..., 3].min() == 255
    # a white texture and the textured frame differ only through the (1 - w) texture term
    b2 = tri_buffers(scene, rt.Material.white())
    a2, _, _ = oracle.render_tri(p, b2, constant_sky.faces, 40, 24)
    assert not np.array_equal(a, a2)


def test_heatmap_counts(oracle):
    scene, mat = triangle_scene(seed=2, n_models=2)
    b = tri_buffers(scene, mat)
    img, steps = oracle.heatmap_tri(scene.pack_params(4), b, 32, 24)
    assert steps.min() >= 2                                   # at least the TLAS root's two box tests (HK:143)
    q = np.floor(np.clip(steps.astype(F) / F(300), 0, 1) * F(255) + F(0.5)).astype(np.uint8)
    assert np.array_equal(img[..., 0], q) and np.array_equal(img[..., 0], img[..., 2]) and np.all(img[..., 3] == 255)
