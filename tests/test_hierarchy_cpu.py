"""The host side of the fast sphere path, checked without a GPU: rt_build_hierarchy runs the
bounding-sphere hierarchy build of rt_bvh.hip.  Invariants the device walk relies
on: depth-first layout with forward skip links and a self-linked sentinel, every sphere a leaf
exactly once, every node sphere containing its members with the 4 % slack of the proof, large
spheres kept out of the tree, determinism."""
import ctypes

import numpy as np
import pytest

import compute_raytracer_amd as rt
from compute_raytracer_amd import abi
from compute_raytracer_amd.scene_raytracing import synthetic_spheres

LEAF = 0x80000000
S = 2.0 ** 40
EPS, KAPPA = 2.0 ** -17, 2.0 ** -16


def build(spheres):
    rec = np.ascontiguousarray(rt.SceneRaytracing().createScene(spheres).pack_spheres(), dtype=np.float32).reshape(-1, 8)
    n = rec.shape[0]
    cap = 2 * n + 64
    out = np.zeros((cap, 4), np.float32)
    link = np.zeros(cap, np.uint32)
    nodes = ctypes.c_uint32(0)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = abi.load().rt_build_hierarchy(rec.ctypes.data_as(fp), n, out.ctypes.data_as(fp),
                                       link.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), cap, ctypes.byref(nodes))
    assert rc == abi.RT_OK
    m = nodes.value
    return rec, out[: m + 1].copy(), link[: m + 1].copy(), m


def check_tree(rec, out, link, m):
    n = rec.shape[0]
    assert n <= m <= 2 * n + 64
    assert link[m] == 4 * m and np.isinf(out[m, 3]) and out[m, 3] > 0          # sentinel
    leaves = link[:m][(link[:m] & LEAF) != 0] & 0x7FFFFFFF
    assert sorted(leaves.tolist()) == list(range(n))                            # every sphere exactly once
    c = rec[:, 0:3].astype(np.float64)
    r = np.abs(rec[:, 7].astype(np.float64))
    for i in range(m):
        if link[i] & LEAF:
            assert not out[i].any()                                             # filled on the device
            continue
        assert link[i] % 4 == 0
        end = link[i] // 4
        assert i + 1 < end <= m                                                 # forward link, non-empty subtree
        inner = [j for j in range(i + 1, end) if not (link[j] & LEAF)]
        assert all(link[j] // 4 <= end for j in inner)                          # nested subtrees
        members = link[i + 1 : end][(link[i + 1 : end] & LEAF) != 0] & 0x7FFFFFFF
        assert len(members) >= 2
        C = out[i, 0:3].astype(np.float64) / S
        k = float(out[i, 3]) / (S * S)
        c2 = float(C @ C)
        need = (np.linalg.norm(c[members] - C, axis=1) + r[members]).max()
        # k = |C|^2 (1-eps) - R^2 (1+kappa) with R >= 1.04 * need, stored in fp32 (the eps term of the
        # node test covers that rounding, 2^-24 |k|, many times over)
        round_k = 2.0 ** -23 * max(c2, need * need)
        k_slack = c2 * (1.0 - EPS) - (1.04 * need) ** 2 * (1.0 + KAPPA)
        k_tight = c2 * (1.0 - EPS) - (1.05 * need) ** 2 * (1.0 + KAPPA)
        assert k <= k_slack + round_k, (i, k, k_slack)          # the radius carries the 4 % slack of the proof
        assert k >= k_tight - round_k - 1e-12, (i, k, k_tight)  # and not much more

@pytest.mark.parametrize("n,seed", [(2, 1), (5, 2), (9, 3), (64, 357), (1024, 358), (4096, 360)])
def test_baseline_like_scenes(n, seed):
    rec, out, link, m = build(synthetic_spheres(n, seed))
    check_tree(rec, out, link, m)
    if n >= 64:
        assert link[0] == (LEAF | 0)            # the ground sphere (index 0, r = 100) is a top-level leaf
        assert m <= 1.6 * n                     # ~1.5 nodes per sphere


def test_single_sphere_and_empty_scene():
    rec, out, link, m = build([rt.Sphere([0, 1, -5], 1.0, [1, 1, 1])])
    assert m == 1 and link[0] == (LEAF | 0) and link[1] == 4
    nodes = ctypes.c_uint32(7)
    assert abi.load().rt_build_hierarchy(None, 0, None, None, 0, ctypes.byref(nodes)) == abi.RT_OK and nodes.value == 0


def test_coincident_nested_and_collinear_spheres():
    cases = [
        [rt.Sphere([0, 1, -6], 1.0, [1, 0, 0])] * 7,
        [rt.Sphere([0, 1, -6], 0.1 * 2 ** k, [0, 1, 0]) for k in range(9)],
        [rt.Sphere([i * 0.5, 0, 0], 0.2, [0, 0, 1]) for i in range(37)],
        [rt.Sphere([0, 0, 0], 0.0, [0, 0, 1]), rt.Sphere([1, 0, 0], 0.0, [0, 0, 1]), rt.Sphere([5, 5, 5], 1e-3, [1, 1, 1])],
    ]
    for spheres in cases:
        check_tree(*build(spheres))


def test_random_scenes_over_orders_of_magnitude():
    for seed in range(12):
        rng = np.random.default_rng(seed)
        scale = 10 ** rng.uniform(-2, 4)
        off = rng.choice([0.0, 1e3, 1e5]) * rng.normal(size=3)
        n = int(rng.choice([6, 33, 300, 1500]))
        spheres = [rt.Sphere(off + rng.normal(size=3) * scale, scale * 10 ** rng.uniform(-3, -0.5), [1, 1, 1]) for _ in range(n)]
        if seed % 2:
            spheres.append(rt.Sphere(off + np.array([0, -60 * scale, 0]), 55 * scale, [1, 1, 1]))
        check_tree(*build(spheres))


def test_build_is_deterministic_and_reports_capacity():
    a = build(synthetic_spheres(700, 5))
    b = build(synthetic_spheres(700, 5))
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    rec = a[0]
    nodes = ctypes.c_uint32(0)
    small = np.zeros((8, 4), np.float32)
    link = np.zeros(8, np.uint32)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = abi.load().rt_build_hierarchy(rec.ctypes.data_as(fp), rec.shape[0], small.ctypes.data_as(fp),
                                       link.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 8, ctypes.byref(nodes))
    assert rc == abi.RT_ERR_CAPACITY and nodes.value == a[3]
