"""Cube-map sampling on the GPU against the oracle (a8): seamless filtering across the 12 edges and
8 corners of a WebGPU cube texture (six equal squares), the clamp-to-edge fallback for image sets
that are not one, and the one-fetch form of a one-colour 1x1 sky.

Frames of pure sky: no sphere is in view, the camera sits at the origin and looks along each of the
six axes with `right` / `up` scaled by 2 (the params block is raw floats, RR:157-165), so that every
frame covers a whole face, its four edges and its four corners.  Every kernel that samples the sky is
driven: the literal kernel (strict), the brute-force kernel (fast, few spheres), the hierarchy
kernel (variant 4) and -- in tests/test_triangles_gpu.py -- the triangle kernel.
"""
import ctypes

import numpy as np
import pytest

import compute_raytracer_amd as rt
from compute_raytracer_amd import abi

pytestmark = pytest.mark.gpu

W, H = 96, 96
VIEWS = [  # forwards, right, up
    ((1, 0, 0), (0, 0, -1), (0, 1, 0)), ((-1, 0, 0), (0, 0, 1), (0, 1, 0)),
    ((0, 1, 0), (1, 0, 0), (0, 0, 1)), ((0, -1, 0), (1, 0, 0), (0, 0, -1)),
    ((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (-1, 0, 0), (0, 1, 0)),
    ((1, 1, 1), (1, -1, 0), (-1, -1, 2)), ((-1, 1, -1), (1, 0, -1), (1, 2, 1)),       # straight at two corners
]


def cube(seed, n, m=None):
    rng = np.random.default_rng(seed)
    sky = rt.CubemapMaterial()
    sky.faces = [rng.integers(0, 256, (m or n, n, 4), dtype=np.uint8) for _ in range(6)]
    return sky


def frames(oracle, sky, spheres, strict, variant):
    L = abi.load()
    fp = ctypes.POINTER(ctypes.c_float)
    ctx = ctypes.c_void_p()
    abi.check(L.rt_create(0, ctypes.byref(ctx)))
    try:
        abi.check(L.rt_resize(ctx, W, H), ctx)
        abi.check(L.rt_set_mode(ctx, 1 if strict else 0), ctx)
        abi.check(L.rt_set_variant(ctx, variant), ctx)
        abi.check(L.rt_write_spheres(ctx, spheres.ctypes.data_as(fp), spheres.shape[0]), ctx)
        for f in range(6):
            face = np.ascontiguousarray(sky.faces[f])
            abi.check(L.rt_write_cubemap_face(ctx, f, face.shape[1], face.shape[0], face.ctypes.data), ctx)
        for fwd, right, up in VIEWS:
            p = np.zeros(24, np.float32)
            p[4:7] = fwd
            p[8:11] = 2.0 * np.asarray(right, np.float32)
            p[12:15] = 2.0 * np.asarray(up, np.float32)
            p[16:19] = (0, 5, 0)
            p[19], p[20], p[21] = 3.0, 0.7, 2
            abi.check(L.rt_write_params(ctx, p.ctypes.data_as(fp)), ctx)
            abi.check(L.rt_render(ctx), ctx)
            img = np.zeros((H, W, 4), np.uint8)
            abi.check(L.rt_read_pixels(ctx, img.ctypes.data, img.nbytes), ctx)
            ref, _, _ = oracle.render(p, spheres, sky.faces, W, H)
            assert np.array_equal(img, ref), (fwd, int((img != ref).any(-1).sum()))
    finally:
        L.rt_destroy(ctx)


def far_spheres(n):
    s = np.zeros((n, 8), np.float32)
    s[:, 0] = 5000.0 + 10.0 * np.arange(n)      # a row of small spheres no ray of these views reaches
    s[:, 1] = 9000.0
    s[:, 2] = 7000.0
    s[:, 4:7] = 0.5
    s[:, 7] = 1.0
    return s


KERNELS = [("literal", True, 0, 3), ("brute", False, 0, 3), ("hierarchy", False, 4, 200)]


@pytest.mark.parametrize("kernel,strict,variant,nspheres", KERNELS, ids=[k[0] for k in KERNELS])
@pytest.mark.parametrize("n", [1, 2, 3, 8, 64])
def test_seamless_cube_all_faces_edges_corners(oracle, n, kernel, strict, variant, nspheres):
    frames(oracle, cube(10 + n, n), far_spheres(nspheres), strict, variant)


@pytest.mark.parametrize("kernel,strict,variant,nspheres", KERNELS, ids=[k[0] for k in KERNELS])
def test_non_cube_image_sets_clamp_inside_the_face(oracle, kernel, strict, variant, nspheres):
    frames(oracle, cube(3, 6, 5), far_spheres(nspheres), strict, variant)        # 6 x 5 faces
    mixed = cube(4, 4)
    mixed.faces[3] = np.random.default_rng(9).integers(0, 256, (2, 2, 4), dtype=np.uint8)
    frames(oracle, mixed, far_spheres(nspheres), strict, variant)                # squares of different sizes


@pytest.mark.parametrize("kernel,strict,variant,nspheres", KERNELS, ids=[k[0] for k in KERNELS])
def test_one_colour_sky_takes_the_one_fetch_form_with_the_same_values(oracle, kernel, strict, variant, nspheres):
    sky = rt.CubemapMaterial.constant((13, 200, 77, 255))
    frames(oracle, sky, far_spheres(nspheres), strict, variant)
