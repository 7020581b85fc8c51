"""The brute-force filter's candidate mask must survive a FRACTIONAL discriminant.

Round 1 built the 16-sphere candidate mask in the FMA pipe: `clamp` turned a positive 2^80-scaled
filter discriminant into 1.0 and `code = 2*code + dd` shifted it in.  A discriminant in (0, 1) --
a ray that grazes the inflated sphere to within 2^-40 -- left a fraction in that sum, and its
carries could clear the bit of ANOTHER sphere of the batch (rt_filter.h; the round-1 comment
called it "a double coincidence of measure zero").  It is constructible, and this file constructs
it: the mask now comes from sign bits (v_alignbit_b32), which no value can disturb.

Scene (camera at the origin, every column of the frame casts the same ray because right = 0):
  * sphere 13: centre (-X, 0, 0), radius r with fl(X*X) == fl(fl(r*r) * (1 + 2^-16)) -- the
    hoisted camera record's 4th component (|oc|^2 - r^2 (1+kappa)) cancels to exactly 0 --, so the
    filter value of a primary ray is b*b with b = h.x * X * 2^40;
  * forwards.x = -0.95 / (X * 2^40): b*b is 0.90 (a fraction); at weight 2^(15-13) it adds 3.6;
  * sphere 15: a sphere every ray really hits, whose bit (weight 1) the old sum carried away:
    3.6 + 1 = 4.6 -> mask 0b100 -> spheres 14 and 15 dropped, sphere 13 kept.
"""
import numpy as np
import pytest

import compute_raytracer_amd as rt
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA

F = np.float32
KAPPA = F(2.0 ** -16)
RADIUS = F(1.5000001192092896)
X = F(1.5000115633010864)
W, H, BOUNCES = 64, 8, 2


def fraction_scene():
    scene = rt.synthetic_scene(16, 1)
    sph = []
    for i in range(16):
        sph.append(rt.Sphere([100.0 + 10.0 * i, 100.0, 0.0], 1.0, [0.5, 0.5, 0.5]))   # far off every ray
    sph[13] = rt.Sphere([-float(X), 0.0, 0.0], float(RADIUS), [1.0, 0.0, 0.0])
    sph[15] = rt.Sphere([0.0, 0.0, -5.0], 1.0, [0.0, 1.0, 0.0])
    scene.spheres = sph
    p = scene.pack_params(BOUNCES).copy()
    p[0:3] = 0.0                                              # camera at the origin
    p[4:7] = [-0.95 / (float(X) * 2.0 ** 40), 0.0, -1.0]      # forwards
    p[8:11] = 0.0                                             # right: every column casts the same ray
    p[12:15] = [0.0, 1.0, 0.0]                                # up
    return scene, p.astype(np.float32)


def test_construction_yields_a_fractional_filter_value():
    """numpy restatement of prep_spheres' camera record and of the hoisted filter's FMA chain
    (rt_kernels.hip: prep_spheres; rt_filter.h: filter_one<false>): the record's 4th component is
    exactly 0 and b*b lies in [0.75, 1) for every row of the frame."""
    scene, p = fraction_scene()
    c = np.array(scene.spheres[13].center, dtype=np.float32)
    co = (p[0:3] - c).astype(np.float32)
    cc = F(F(F(co[0] * co[0]) + F(co[1] * co[1])) + F(co[2] * co[2]))
    r2 = F(RADIUS * RADIUS)
    r2f = F(r2 * (F(1) + KAPPA))
    assert cc == r2f and cc - r2f == 0.0
    for y in range(H):
        vc = F(F(F(H) / F(2) - F(y)) / F(W)) * F(2)
        d = (p[4:7] + vc * p[12:15]).astype(np.float32)
        d = (d / F(np.sqrt(np.float64(np.dot(d.astype(np.float64), d.astype(np.float64)))))).astype(np.float32)
        a = np.dot(d.astype(np.float64), d.astype(np.float64))
        hx = np.float64(d[0]) / np.sqrt(a) * (1.0 + 2.0 ** -16)
        b = hx * float(co[0]) * 2.0 ** 40
        assert b < 0.0 and 0.75 <= b * b < 1.0, (y, b)


def _render(L, ctx, abi, p, s, sky, strict, variant):
    import ctypes
    fp = ctypes.POINTER(ctypes.c_float)
    abi.check(L.rt_set_mode(ctx, 1 if strict else 0), ctx)
    abi.check(L.rt_set_variant(ctx, variant), ctx)
    abi.check(L.rt_write_params(ctx, p.ctypes.data_as(fp)), ctx)
    abi.check(L.rt_write_spheres(ctx, s.ctypes.data_as(fp), s.shape[0]), ctx)
    for f in range(6):
        face = np.ascontiguousarray(sky.faces[f])
        abi.check(L.rt_write_cubemap_face(ctx, f, face.shape[1], face.shape[0], face.ctypes.data), ctx)
    abi.check(L.rt_render(ctx), ctx)
    img = np.zeros((H, W, 4), np.uint8)
    abi.check(L.rt_read_pixels(ctx, img.ctypes.data, img.nbytes), ctx)
    st = abi.RtStats()
    abi.check(L.rt_get_stats(ctx, ctypes.byref(st)), ctx)
    return img, st.rays


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 3, 5])
def test_fractional_discriminant_does_not_lose_a_neighbouring_sphere(oracle, variant):
    import ctypes
    from compute_raytracer_amd import abi
    scene, p = fraction_scene()
    s = scene.pack_spheres()
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    ref, _, rays = oracle.render(p, s, sky.faces, W, H)
    assert (ref[..., 1] > ref[..., 0]).all(), "every pixel of the oracle's frame shows the green sphere 15"
    L = abi.load()
    ctx = ctypes.c_void_p()
    abi.check(L.rt_create(0, ctypes.byref(ctx)))
    try:
        abi.check(L.rt_resize(ctx, W, H), ctx)
        strict_img, strict_rays = _render(L, ctx, abi, p, s, sky, True, 0)
        fast_img, fast_rays = _render(L, ctx, abi, p, s, sky, False, variant)
    finally:
        L.rt_destroy(ctx)
    assert np.array_equal(strict_img, ref) and strict_rays == rays
    assert np.array_equal(fast_img, ref) and fast_rays == rays
