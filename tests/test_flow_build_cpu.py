"""Host side of the triangle kernel's PAIRS forms, checked without a GPU (the library loads and rt_build_flow runs without a device): the relinked copy of the BLAS trees
(compute_raytracer_amd/csrc/rt_flow_build.h through rt_build_flow of the C ABI).  A walk over the pair records must
visit the reference's boxes in the reference's order -- here a restatement of the kernel's three walk steps in numpy
float32 (same expressions, same order as rt_tri_device.h / RK:168-341) runs over the records and is compared, hit for hit,
with the oracle's walk over the reference's node buffer, on the reference's own scene and on procedural ones."""
import ctypes

import numpy as np
import pytest

import compute_raytracer_amd as rt
from compute_raytracer_amd import abi
from helpers import ref_fixture, tri_buffers, triangle_scene

F = np.float32
U32 = ctypes.POINTER(ctypes.c_uint32)
FP = ctypes.POINTER(ctypes.c_float)


def u32f(f):
    f = float(f)
    if not f > 0.0:
        return 0
    return 4294967295 if f >= 4294967040.0 else int(f)


def build_flow(nodes, roots):
    L = abi.load()
    L.rt_build_flow.restype = ctypes.c_int
    L.rt_build_flow.argtypes = [FP, ctypes.c_uint32, U32, ctypes.c_uint32, FP, ctypes.c_uint32, U32, U32]
    nodes = np.ascontiguousarray(nodes, F).reshape(-1, 8)
    roots = np.ascontiguousarray(roots, np.uint32)
    n = ctypes.c_uint32(0)
    pairs = np.zeros((nodes.shape[0], 16), F)
    meta = np.zeros(len(roots), np.uint32)
    rc = L.rt_build_flow(nodes.ctypes.data_as(FP), nodes.shape[0], roots.ctypes.data_as(U32), len(roots), pairs.ctypes.data_as(FP),
                         pairs.shape[0], ctypes.byref(n), meta.ctypes.data_as(U32))
    return rc, pairs[:n.value].copy(), meta


def scene_roots(b):
    return [u32f(r[16]) for r in b["blas"].reshape(-1, 20)]


def hit_aabb(o, inv, lo, hi):                                        # RK:395-410, 1/dir hoisted as in the kernels
    t1 = (lo - o) * inv
    t2 = (hi - o) * inv
    with np.errstate(invalid="ignore"):
        tmin = max(max(min(t1[0], t2[0]), min(t1[1], t2[1])), min(t1[2], t2[2]))    # fmaxf / fminf: a NaN operand loses
        tmax = min(min(max(t1[0], t2[0]), max(t1[1], t2[1])), max(t1[2], t2[2]))
    if tmin > tmax or tmax < 0:
        return F(99999.0)
    return F(tmin)


def fmin(a, b):
    return b if a != a else (a if b != b else min(a, b))


def fmax(a, b):
    return b if a != a else (a if b != b else max(a, b))


def hit_aabb_c(o, inv, lo, hi):                                      # with C's fminf / fmaxf NaN rule
    t1 = (lo - o) * inv
    t2 = (hi - o) * inv
    tmin = fmax(fmax(fmin(t1[0], t2[0]), fmin(t1[1], t2[1])), fmin(t1[2], t2[2]))
    tmax = fmin(fmin(fmax(t1[0], t2[0]), fmax(t1[1], t2[1])), fmax(t1[2], t2[2]))
    if tmin > tmax or tmax < 0:
        return F(99999.0)
    return F(tmin)


def dot(a, b):
    return F(F(a[0] * b[0] + a[1] * b[1]) + a[2] * b[2])


def cross(a, b):
    return np.array([a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]], F)


def hit_triangle(tri, o, d, tmax):                                   # RK:344-380
    A, B, C = tri[0:3], tri[12:15], tri[24:27]
    e1, e2 = B - A, C - A
    rce2 = cross(d, e2)
    det = dot(e1, rce2)
    if det < F(0.00001):
        return None
    s = o - A
    u = dot(s, rce2)
    if u < 0 or u > det:
        return None
    sce1 = cross(s, e1)
    v = dot(d, sce1)
    if v < 0 or F(u + v) > det:
        return None
    inv_det = F(1.0) / det
    t = F(inv_det * dot(e2, sce1))
    if t > F(0.001) and t < tmax:
        return t
    return None


def walk_pairs(b, pairs, root_meta, o, d):
    """The kernel's state machine for one ray, TLAS level as the reference has it (first-level nodes from the node buffer),
    BLAS level over pair records with the hybrid stack's semantics (20 slots, index clamped to the last).  -> (t or -1, lookup slot)"""
    nodes = b["nodes"].reshape(-1, 8)
    blas = b["blas"].reshape(-1, 20)
    tri = b["triangles"].reshape(-1, 40)
    look = b["tri_lookup"].reshape(-1)
    blook = b["blas_lookup"].reshape(-1)
    n_nodes = nodes.shape[0]
    sclamp = lambda i: min(i, 19)
    node_at = lambda i: nodes[min(i, n_nodes - 1)]
    pack = lambda nd: (min(u32f(nd[7]), 0xFFFF) << 16) | min(u32f(nd[3]), 0xFFFF)
    nearest, htri = F(9999.0), -1
    tnode, ti, sp_t, tstack = pack(node_at(0)), 0, 0, [0] * 20
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        winv = F(1.0) / d
        while True:
            count, left = tnode >> 16, tnode & 0xFFFF
            pop = False
            if count == 0:
                c1, c2 = node_at(left), node_at(left + 1)
                d1, d2 = hit_aabb_c(o, winv, c1[0:3], c1[4:7]), hit_aabb_c(o, winv, c2[0:3], c2[4:7])
                i2 = left + 1
                swap = d1 > d2
                if swap:
                    d1, d2, i2 = d2, d1, left
                if d1 > nearest:
                    pop = True
                else:
                    tnode, ti = pack(c2 if swap else c1), 0
                    if d2 < nearest:
                        tstack[sclamp(sp_t)] = min(i2, n_nodes - 1)
                        sp_t += 1
                        if sp_t > 20:
                            sp_t = 19
            elif ti < count:
                li = min(ti + left, blook.shape[0] - 1)
                bi = min(u32f(blook[li]), blas.shape[0] - 1)
                m = blas[bi]
                oo = np.array([F(F(F(m[r] * o[0] + m[4 + r] * o[1]) + m[8 + r] * o[2]) + m[12 + r] * F(1)) for r in range(3)], F)
                od = np.array([F(F(F(m[r] * d[0] + m[4 + r] * d[1]) + m[8 + r] * d[2]) + m[12 + r] * F(0)) for r in range(3)], F)
                inv = F(1.0) / od
                bnode, sp_b, bnear, tk, stack = int(root_meta[bi]), 0, nearest, 0, [0] * 20
                while True:                                          # BNODE / TRI steps until the BLAS walk is over
                    bc, bl = bnode >> 16, bnode & 0xFFFF
                    bpop = False
                    if bc == 0:
                        q = pairs[bl]
                        d1, d2 = hit_aabb_c(oo, inv, q[0:3], q[4:7]), hit_aabb_c(oo, inv, q[8:11], q[12:15])
                        m1, m2 = int(q[3:4].view(np.uint32)[0]), int(q[11:12].view(np.uint32)[0])
                        swap = d1 > d2
                        if swap:
                            d1, d2 = d2, d1
                        if d1 > bnear:
                            bpop = True
                        else:
                            bnode = m2 if swap else m1
                            if d2 < bnear:
                                stack[sclamp(sp_b)] = m1 if swap else m2
                                sp_b += 1
                            tk = 0
                    else:
                        li2 = min(bl + tk, look.shape[0] - 1)
                        t = hit_triangle(tri[min(u32f(look[li2]), tri.shape[0] - 1)], oo, od, bnear)
                        if t is not None:
                            bnear, htri = t, li2
                        tk += 1
                        bpop = tk >= bc
                    if bpop:
                        if sp_b == 0:
                            break
                        sp_b -= 1
                        bnode, tk = stack[sclamp(sp_b)], 0
                nearest = bnear if bnear < nearest else nearest
                ti += 1
            else:
                pop = True
            if pop:
                if sp_t == 0:
                    break
                sp_t -= 1
                tnode, ti = pack(node_at(tstack[sclamp(sp_t)])), 0
    return (nearest if htri >= 0 else F(-1.0)), htri


def check_structure(b, pairs, root_meta):
    nodes = b["nodes"].reshape(-1, 8)
    n = nodes.shape[0]
    meta = pairs[:, [3, 11]].copy().view(np.uint32)
    assert np.all(pairs[:, [7, 15]] == 0)
    # every record is the pair of children of some inner node, boxes untouched; walk down from the roots and compare
    seen = {}
    todo = [(min(r, n - 1), int(m)) for r, m in zip(scene_roots(b), root_meta)]
    while todo:
        i, m = todo.pop()
        nd = nodes[i]
        if u32f(nd[7]) != 0:
            assert m == (min(u32f(nd[7]), 0xFFFF) << 16) | min(u32f(nd[3]), 0xFFFF)
            continue
        assert m >> 16 == 0
        p = m & 0xFFFF
        a, c = min(u32f(nd[3]), n - 1), min(u32f(nd[3]) + 1, n - 1)
        assert np.array_equal(pairs[p, 0:3], nodes[a, 0:3]) and np.array_equal(pairs[p, 4:7], nodes[a, 4:7])
        assert np.array_equal(pairs[p, 8:11], nodes[c, 0:3]) and np.array_equal(pairs[p, 12:15], nodes[c, 4:7])
        if p in seen:
            assert seen[p] == a
            continue
        seen[p] = a
        todo += [(a, int(meta[p, 0])), (c, int(meta[p, 1]))]
    assert len(seen) == pairs.shape[0]                               # nothing unreachable was emitted
    return seen


def test_reference_scene_relinked():
    scene, sky, W, H, B, canvas, pin = ref_fixture()
    b = tri_buffers(scene, rt.Material.white())
    rc, pairs, root_meta = build_flow(b["nodes"], scene_roots(b))
    assert rc == 0
    seen = check_structure(b, pairs, root_meta)
    nodes = b["nodes"].reshape(-1, 8)
    inner = [i for i in range(scene.tlasNodesMax, nodes.shape[0]) if u32f(nodes[i, 7]) == 0]
    assert pairs.shape[0] == len(inner) == 12441                    # every inner BLAS node of the three trees has its record
    # most-visited first: the surface area of the parent's box never increases by more than rounding along the array
    parent_of = {min(u32f(nodes[i, 3]), nodes.shape[0] - 1): i for i in inner}
    ext = nodes[:, 4:7].astype(np.float64) - nodes[:, 0:3].astype(np.float64)
    area = 2 * (ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 0] * ext[:, 2])
    order = [area[parent_of[seen[p]]] for p in range(pairs.shape[0])]
    assert all(order[k] >= order[k + 1] * (1 - 1e-12) for k in range(len(order) - 1))


@pytest.mark.parametrize("which", ["ref", "procedural"])
def test_pair_walk_is_the_node_walk(oracle, which):
    if which == "ref":
        scene, sky, W, H, B, canvas, pin = ref_fixture()
        b = tri_buffers(scene, rt.Material.white())
        n_rays = 160
    else:
        scene, mat = triangle_scene(seed=31, n_models=4, rings=10, sectors=14)
        b = tri_buffers(scene, mat)
        n_rays = 240
    rc, pairs, root_meta = build_flow(b["nodes"], scene_roots(b))
    assert rc == 0
    check_structure(b, pairs, root_meta)
    rng = np.random.default_rng(5)
    cam = scene.pack_params(4)[0:3]
    o = (cam[None, :] + rng.normal(0, 0.3, (n_rays, 3))).astype(F)
    tgt = rng.uniform(-3, 3, (n_rays, 3)).astype(F) * np.array([1, 0.5, 1], F)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F)
    d[::17, 1] = 0                                                   # axis-parallel components: 1/0 = inf in the slab test
    want = oracle.trace_tri_rays(b, o, d)
    hits = 0
    for k in range(n_rays):
        t, slot = walk_pairs(b, pairs, root_meta, o[k], d[k])
        assert np.float32(t).tobytes() == np.float32(want[k]).tobytes(), (k, t, want[k])
        hits += slot >= 0
    assert hits > n_rays // 4


def test_malformed_buffers_terminate_and_clamp():
    """Indices that point anywhere (beyond the buffer, at themselves, at shared children): the build follows the oracle's
    clamp-to-last-element rule, gives every distinct child index one record and terminates."""
    rng = np.random.default_rng(2)
    n = 40
    nodes = rng.uniform(-1, 1, (n, 8)).astype(F)
    nodes[:, 7] = 0                                                  # all inner
    nodes[:, 3] = rng.integers(0, 60, n).astype(F)                   # children anywhere, some beyond the buffer
    nodes[-1, 7] = 3; nodes[-1, 3] = 7                               # the last node (what every out-of-range index reads) is a leaf
    rc, pairs, meta = build_flow(nodes, [0, 5, 1000])
    assert rc == 0 and 1 <= pairs.shape[0] <= n
    m = pairs[:, [3, 11]].copy().view(np.uint32)
    inner = (m >> 16) == 0
    assert np.all((m & 0xFFFF)[inner] < pairs.shape[0])
    assert meta[2] == (3 << 16 | 7)                                  # root 1000 reads the last node: a leaf
    # beyond 16 bits: refused, the caller keeps the node walk
    big = np.zeros((65537, 8), F)
    rc, _, _ = build_flow(big, [0])
    assert rc == abi.RT_ERR_UNSUPPORTED
    cnt = np.zeros((3, 8), F); cnt[0, 3] = 1; cnt[1, 7] = 70000; cnt[2, 7] = 1
    rc, _, _ = build_flow(cnt, [0])
    assert rc == abi.RT_ERR_UNSUPPORTED
