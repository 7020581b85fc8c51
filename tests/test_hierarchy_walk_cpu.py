"""The proof obligation of the hierarchy walk (rt_bvh.hip), exercised without a GPU: with the node
records rt_build_hierarchy produces and the leaf records prep_spheres would write, a restatement of
the walk in emulated fp32 fused arithmetic must reach every sphere the reference's LITERAL test
(HK:308-318, restated in fp32 numpy) accepts -- for rays from the camera, from the light, from
points on sphere surfaces, aimed at sphere limbs (grazing hits are where a conservative test fails
first), in scenes small and large, near the world origin and far from it."""
import ctypes

import numpy as np
import pytest

import compute_raytracer_amd as rt
from compute_raytracer_amd import abi
from compute_raytracer_amd.scene_raytracing import synthetic_spheres

f32 = np.float32
LEAF = 0x80000000
S, S2 = 2.0 ** 40, 2.0 ** 80
EPS, KAPPA, KAPPA_H = 2.0 ** -17, 2.0 ** -16, 2.0 ** -14


def fma(a, b, c):          # fp32 fused multiply-add: the product is exact in double
    return f32(np.float64(a) * np.float64(b) + np.float64(c))


def hierarchy(rec):
    n = rec.shape[0]
    cap = 2 * n + 64
    out = np.zeros((cap, 4), f32)
    link = np.zeros(cap, np.uint32)
    nodes = ctypes.c_uint32(0)
    fp = ctypes.POINTER(ctypes.c_float)
    assert abi.load().rt_build_hierarchy(rec.ctypes.data_as(fp), n, out.ctypes.data_as(fp),
                                         link.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), cap, ctypes.byref(nodes)) == 0
    m = nodes.value
    out, link = out[: m + 1].copy(), link[: m + 1].copy()
    for i in range(m):      # leaf records as prep_spheres forms them (rt_kernels.hip)
        if link[i] & LEAF:
            s = int(link[i] & 0x7FFFFFFF)
            c = rec[s, 0:3]
            r2 = f32(rec[s, 7] * rec[s, 7])
            cd2 = float(c[0]) ** 2 + float(c[1]) ** 2 + float(c[2]) ** 2
            kf = cd2 * (1.0 - EPS) - float(r2) * (1.0 + KAPPA)
            out[i] = [f32(c[0] * f32(S)), f32(c[1] * f32(S)), f32(c[2] * f32(S)), f32(kf * S2)]
    return out, link, m


def walk(out, link, m, o, d, sgn, madd=0.0):
    """trace_bvh's candidate selection (one lane); madd: the additive slack of a reversed shadow walk."""
    o = o.astype(f32); d = d.astype(f32)
    a = f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))
    inv = f32(f32(1.0 / np.sqrt(np.float64(a))) * f32(1.0 + KAPPA_H))          # v_rsq_f32 to ~1 ulp
    h = (d * inv).astype(f32)
    os_ = (o * f32(S)).astype(f32)
    mm = (f32(-2.0) * os_).astype(f32)
    p = f32(f32(f32(h[0] * os_[0]) + f32(h[1] * os_[1])) + f32(h[2] * os_[2]))
    q = f32(f32(f32(f32(os_[0] * os_[0]) + f32(os_[1] * os_[1])) + f32(os_[2] * os_[2])) * f32(1.0 - EPS))
    q = fma(-f32(madd), f32(S2), q)
    cands, tests, j = [], 0, 0
    with np.errstate(over="ignore", invalid="ignore"):
        while j != m:
            g, lk = out[j], int(link[j])
            b = fma(-h[2], g[2], fma(-h[1], g[1], fma(-h[0], g[0], p)))
            cp = fma(mm[2], g[2], fma(mm[1], g[1], fma(mm[0], g[0], g[3])))
            bm = min(b, f32(0.0)) if sgn else b
            passed = bool(fma(bm, bm, -q) > cp)
            leaf = bool(lk & LEAF)
            if leaf and passed:
                cands.append(lk & 0x7FFFFFFF)
            j = j + 1 if (leaf or passed) else lk // 4
            tests += 1
    return set(cands), tests


def literal_hits(rec, o, d, want_t=False):
    """HK:308-318 for every sphere, fp32, the oracle's operation order."""
    o = o.astype(f32); d = d.astype(f32)
    c = rec[:, 0:3]
    r2 = (rec[:, 7] * rec[:, 7]).astype(f32)
    oc = (o[None, :] - c).astype(f32)
    a = f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))
    dot = ((d[0] * oc[:, 0]).astype(f32) + (d[1] * oc[:, 1]).astype(f32)).astype(f32)
    dot = (dot + (d[2] * oc[:, 2]).astype(f32)).astype(f32)
    b = (f32(2.0) * dot).astype(f32)
    cc = ((oc[:, 0] * oc[:, 0]).astype(f32) + (oc[:, 1] * oc[:, 1]).astype(f32)).astype(f32)
    cc = ((cc + (oc[:, 2] * oc[:, 2]).astype(f32)).astype(f32) - r2).astype(f32)
    disc = ((b * b).astype(f32) - (f32(f32(4.0) * a) * cc).astype(f32)).astype(f32)
    with np.errstate(invalid="ignore", divide="ignore"):
        t = ((-b - np.sqrt(disc).astype(f32)).astype(f32) / f32(f32(2.0) * a)).astype(f32)
        hit = (disc > 0) & (t > f32(0.001)) & (t < f32(9999.0))
    if want_t:
        return hit, t
    return set(np.nonzero(hit)[0].tolist())


def rays_for(rec, cam, light, rng, count):
    c = rec[:, 0:3].astype(np.float64); r = np.abs(rec[:, 7].astype(np.float64))
    rays = []
    for k in range(count):
        kind = k % 4
        if kind == 0: o = np.array(cam, float)
        elif kind == 1: o = np.array(light, float)
        else:                                   # a point on a sphere, as after a bounce
            s = int(rng.integers(0, len(r)))
            u = rng.normal(size=3); u /= np.linalg.norm(u)
            o = c[s] + u * r[s]
        s = int(rng.integers(0, len(r)))        # aim at the limb of a sphere (grazing) or anywhere
        to = c[s] - o
        dist = np.linalg.norm(to)
        if dist > 1e-9 and k % 3:
            w = rng.normal(size=3); w -= w @ to / dist ** 2 * to
            w /= max(np.linalg.norm(w), 1e-30)
            d = to + w * r[s] * rng.choice([0.0, 0.9, 0.999, 1.0, 1.001, 1.1])
        else:
            d = rng.normal(size=3)
        d = (d / np.linalg.norm(d)).astype(f32)
        d = (d / f32(np.sqrt(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2])))).astype(f32)
        rays.append((o.astype(f32), d))
    return rays


def run_case(spheres, cam, light, seed, count=600):
    rec = np.ascontiguousarray(rt.SceneRaytracing().createScene(spheres).pack_spheres(), dtype=f32).reshape(-1, 8)
    out, link, m = hierarchy(rec)
    rng = np.random.default_rng(seed)
    bound = max(np.linalg.norm(rec[:, 0:3].astype(np.float64), axis=1) + np.abs(rec[:, 7]))
    reach = max(bound, np.linalg.norm(cam), np.linalg.norm(light))
    sgn = 2.0 * reach * 7.3e-7 < 5.0e-4          # rt_api.hip: the sign-aware form only for compact scenes
    total_hits = total_tests = 0
    for o, d in rays_for(rec, cam, light, rng, count):
        hits = literal_hits(rec, o, d)
        for mode in ([True, False] if sgn else [False]):
            cands, tests = walk(out, link, m, o, d, mode)
            assert hits <= cands, ("lost", sorted(hits - cands), o, d, mode)
        total_hits += len(hits); total_tests += tests
    return total_hits, total_tests / count, rec.shape[0]


def test_baseline_scene_rays():
    hits, tests, n = run_case(synthetic_spheres(300, 5), [0.0593, 2.692, 3.293], [0, 5, 0], seed=1)
    assert hits > 300                      # the rays do hit things
    assert tests < 0.45 * n                # and the walk culls


@pytest.mark.parametrize("seed", range(8))
def test_random_scales_and_offsets(seed):
    rng = np.random.default_rng(100 + seed)
    scale = float(10 ** rng.uniform(-2, 3))
    off = float(rng.choice([0.0, 10.0, 300.0, 3000.0, 1e5])) * np.array([1.0, 0.5, -0.25])
    ratio = float(10 ** rng.uniform(0, 2.5))
    n = int(rng.choice([17, 64, 200]))
    spheres = [rt.Sphere(off + rng.normal(size=3) * scale, scale * 0.25 / ratio * float(10 ** rng.uniform(0, np.log10(ratio))),
                         [1, 1, 1]) for _ in range(n)]
    if seed % 2:
        R = scale * float(10 ** rng.uniform(1, 2))
        spheres.append(rt.Sphere(off + np.array([0, -R - scale, 0]), R, [1, 1, 1]))
    cam = off + np.array([0.0, 0.5 * scale, 3.0 * scale])
    light = off + np.array([0.3 * scale, 2.5 * scale, 0.5 * scale])
    hits, _, _ = run_case(spheres, cam, light, seed, count=300)
    assert hits > 0


def test_small_spheres_far_from_every_origin():
    rng = np.random.default_rng(9)
    for dist, radius in [(300.0, 0.05), (3000.0, 2.0), (120.0, 0.01)]:
        pos = np.stack([rng.uniform(-1, 1, 200) * dist * 0.2, rng.uniform(-0.6, 0.6, 200) * dist * 0.2,
                        -dist * rng.uniform(0.9, 1.1, 200)], axis=1)
        spheres = [rt.Sphere(p, radius * float(rng.uniform(0.5, 2.0)), [1, 1, 1]) for p in pos]
        hits, _, _ = run_case(spheres, [0.0, 0.0, 0.0], [0.1 * dist, 0.8 * dist, -0.2 * dist], seed=3, count=300)
        assert hits > 0


# ---- shadow rays: the walk runs backwards from just behind the shaded point (rt_bvh.hip: reversed_shadow_walk) ----------
REV_DELTA, REV_DELTA_REL, REV_SLACK, REV_SLACK_ABS = 0.0051, 2.0 ** -17, 2.0 ** -12, 2.0 ** -21


def length32(v):
    return f32(np.sqrt(f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))))


def reversed_walk_ray(L, P, s):
    """The kernel's statements, fp32: the walk's origin, direction and additive slack for the shadow ray (L, s) towards P."""
    dl = (P - L).astype(f32)
    l1 = f32(f32(abs(dl[0]) + abs(dl[1])) + abs(dl[2]))
    la = f32(l1 + f32(f32(abs(L[0]) + abs(L[1])) + abs(L[2])))
    delta = fma(la, f32(REV_DELTA_REL), f32(REV_DELTA))
    lb = f32(l1 + delta)
    madd = fma(f32(lb * lb), f32(REV_SLACK), f32(f32(la * la) * f32(REV_SLACK_ABS)))
    wo = np.array([fma(delta, s[k], P[k]) for k in range(3)], f32)
    return wo, (-s).astype(f32), madd


def lit(hit, t, L, s, P):
    """RK:155-159 in fp32: is the nearest of `hit` within 0.005 of P?"""
    if not hit.any():
        return False
    tm = t[hit].min()
    dv = ((L + (tm * s).astype(f32)).astype(f32) - P).astype(f32)
    return bool(length32(dv) < f32(0.005))


def shaded_points(rec, L, rng, count):
    """Points a shadow ray is cast towards: on sphere surfaces (lit and far sides), and along rays from the light that graze
    a sphere, at depths around its entry and exit points -- where the reversed walk's origin lands on, in and just outside
    spheres."""
    c = rec[:, 0:3].astype(np.float64); r = np.abs(rec[:, 7].astype(np.float64))
    L = np.array(L, float)
    for k in range(count):
        s = int(rng.integers(0, len(r)))
        if k % 3 == 0:
            v = rng.normal(size=3); v /= np.linalg.norm(v)
            P = c[s] + v * r[s] * (1.0 + rng.choice([0.0, 1e-7, -1e-7, 1e-4]))
        else:
            to = c[s] - L; dist = np.linalg.norm(to)
            if dist < 1e-9:
                continue
            w = rng.normal(size=3); w -= w @ to / dist ** 2 * to; w /= max(np.linalg.norm(w), 1e-30)
            dirn = to + w * r[s] * rng.choice([0.0, 0.5, 0.9, 0.999, 0.9999, 1.0, 1.0001, 1.001])
            dirn /= np.linalg.norm(dirn)
            tc = dirn @ to
            half = np.sqrt(max(0.0, r[s] ** 2 - max(0.0, to @ to - tc * tc)))
            depth = tc + rng.choice([-half, 0.0, half, -half - 0.004, -half + 0.004, half + 0.004, -half - 0.0055, r[s], -r[s], 3 * r[s]]) \
                + rng.choice([0.0, 1e-4, -1e-4, 1e-3])
            P = L + max(depth, 1e-3) * dirn
        yield P.astype(f32)


def run_shadow_case(spheres, light, seed, count):
    rec = np.ascontiguousarray(rt.SceneRaytracing().createScene(spheres).pack_spheres(), dtype=f32).reshape(-1, 8)
    out, link, m = hierarchy(rec)
    L = np.array(light, f32)
    n_lit = fwd = rev = 0
    for P in shaded_points(rec, L, np.random.default_rng(seed), count):
        dl = (P - L).astype(f32)
        s = (dl / length32(dl)).astype(f32)                                  # RK:147
        if not np.isfinite(s).all():
            continue
        hit, t = literal_hits(rec, L, s, want_t=True)
        wo, wd, madd = reversed_walk_ray(L, P, s)
        cands, tests = walk(out, link, m, wo, wd, True, madd)
        keep = np.zeros_like(hit); keep[list(cands)] = True
        want = lit(hit, t, L, s, P)
        assert lit(hit & keep, t, L, s, P) == want, ("shadow result changed", P, L)
        # the stronger statement the header proves (claim B): nothing accepted before l + dA is lost
        ell = np.linalg.norm((P - L).astype(np.float64))
        d_a = 0.005001 + 10 * 2.0 ** -24 * (ell + np.linalg.norm(L.astype(np.float64)))
        near = set(np.nonzero(hit & (t.astype(np.float64) < ell + d_a))[0].tolist())
        assert near <= cands, ("lost", sorted(near - cands), P, L)
        n_lit += want; rev += tests; fwd += walk(out, link, m, L, s, True)[1]
    return n_lit, fwd, rev


def test_reversed_shadow_walk_baseline_scene():
    n_lit, fwd, rev = run_shadow_case(synthetic_spheres(300, 5), [0, 5, 0], seed=1, count=900)
    assert n_lit > 50                     # both results occur
    assert rev < 0.9 * fwd                # and the reversed walk is the shorter one


@pytest.mark.parametrize("seed", range(6))
def test_reversed_shadow_walk_scales_and_offsets(seed):
    """Only scenes the host plans the sign-aware test for (reach < 342) walk shadow rays backwards."""
    rng = np.random.default_rng(200 + seed)
    scale = float(10 ** rng.uniform(-2, 1.7))
    off = float(rng.choice([0.0, 10.0, 100.0])) * np.array([1.0, 0.5, -0.25])
    ratio = float(10 ** rng.uniform(0, 2.5))
    n = int(rng.choice([17, 64, 200]))
    spheres = [rt.Sphere(off + rng.normal(size=3) * scale, scale * 0.25 / ratio * float(10 ** rng.uniform(0, np.log10(ratio))),
                         [1, 1, 1]) for _ in range(n)]
    if seed % 2:
        R = scale * float(10 ** rng.uniform(0.5, 1.0))
        spheres.append(rt.Sphere(off + np.array([0, -R - scale, 0]), R, [1, 1, 1]))
    light = off + np.array([0.3 * scale, 2.5 * scale, 0.5 * scale])
    if seed == 4:
        light = np.array(spheres[3].center, float)            # the light inside a sphere
    run_shadow_case(spheres, light, seed, count=400)
