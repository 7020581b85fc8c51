"""Host-side mirror classes against the reference's TypeScript semantics (no GPU)."""
import math

import numpy as np
import pytest

import compute_raytracer_amd as rt

F = np.float32


def test_camera_default_matches_gl_matrix_semantics():
    # scene-raytracing.ts:39: new Camera([0.0593, 2.692, 3.293], 106, 270); camera.ts:42-53
    cam = rt.Camera([0.0593, 2.692, 3.293], 106, 270)
    th, ph = math.radians(106), math.radians(270)
    fw = np.array([math.cos(ph) * math.sin(th), math.cos(th), math.sin(ph) * math.sin(th)]).astype(F)
    assert np.array_equal(cam.forwards, fw) and cam.forwards.dtype == F
    # right = normalize(cross(forwards, [0,1,0])) with f32 stores (vec3.create -> Float32Array)
    c = np.array([-float(fw[2]), 0.0, float(fw[0])]).astype(F)
    ln = 1 / math.sqrt(float(c[0]) ** 2 + float(c[1]) ** 2 + float(c[2]) ** 2)
    assert np.array_equal(cam.right, (c.astype(np.float64) * ln).astype(F))
    assert abs(float(np.dot(cam.up, cam.forwards))) < 1e-6 and abs(float(np.linalg.norm(cam.up)) - 1) < 1e-6
    assert cam.position == [0.0593, 2.692, 3.293]          # stays f64 until packed (RR:159)


def test_camera_clamps_theta_and_wraps_phi():
    cam = rt.Camera([0, 0, 0], 500, 725)                   # camera.ts:15: phi % 360, clamp(theta, 1, 180)
    assert float(cam.eulers[0]) == 5.0 and float(cam.eulers[1]) == 180.0
    cam.spin(10, -500)
    assert float(cam.eulers[0]) == 15.0 and float(cam.eulers[1]) == 1.0


def test_camera_move():
    cam = rt.Camera([0, 0, 0], 90, 0)
    cam.move(2.0, 0.0)
    assert cam.position == pytest.approx([2 * float(cam.forwards[i]) for i in range(3)])


def test_params_block_layout():
    # RR:157-165: 24 floats, vec3s at 0/4/8/12/16, lightIntensity 19, minIntensity 20, maxBounces 21
    scene = rt.synthetic_scene(2, 9)
    p = scene.pack_params(8)
    assert p.dtype == F and p.shape == (24,)
    assert np.array_equal(p[0:3], np.array(scene.camera.position).astype(F))
    assert np.array_equal(p[4:7], scene.camera.forwards) and np.array_equal(p[8:11], scene.camera.right)
    assert np.array_equal(p[12:15], scene.camera.up) and np.array_equal(p[16:19], np.array([0, 5, 0], F))
    assert p[19] == F(3.0) and p[20] == F(0.3) and p[21] == F(8.0)
    assert not p[[3, 7, 11, 15, 22, 23]].any()


def test_sphere_record_layout():
    # commented `struct Sphere` RK:13-17: center @0, color @16 B, radius @28 B
    scene = rt.synthetic_scene(1, 1)
    scene.spheres = [rt.Sphere([1, 2, 3], 0.25, [0.1, 0.2, 0.3])]
    s = scene.pack_spheres()
    assert s.shape == (1, 8)
    assert np.array_equal(s[0], np.array([1, 2, 3, 0, 0.1, 0.2, 0.3, 0.25], F))


def test_generator_is_deterministic_and_in_range():
    a = rt.synthetic_scene(64, 357).pack_spheres()
    b = rt.synthetic_scene(64, 357).pack_spheres()
    c = rt.synthetic_scene(64, 358).pack_spheres()
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.array_equal(a[0], np.array([0, -100, 0, 0, 0.8, 0.8, 0.8, 100], F))     # ground sphere
    r = a[1:]
    assert (r[:, 0] >= -12).all() and (r[:, 0] <= 12).all() and (r[:, 2] >= -26).all() and (r[:, 2] <= -3).all()
    assert (r[:, 7] >= 0.04).all() and (r[:, 7] <= 1.5).all() and (r[:, 1] >= r[:, 7] - 1e-6).all()
    assert (r[:, 4:7] >= 0.2).all() and (r[:, 4:7] <= 1.0).all()


def test_splitmix64_known_answers():
    # first outputs of splitmix64 seeded with 0 (published test vector of the algorithm)
    from compute_raytracer_amd.scene_raytracing import SplitMix64
    g = SplitMix64(0)
    assert [g.next_u64() for _ in range(3)] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]
    u = SplitMix64(355).uniform()
    assert 0.0 <= u < 1.0 and float(F(u)) == u        # 24-bit mantissa: exact in f32


def test_cubemap_cross_face_order():
    # cubemap-material.ts:40-47: Right(col2,row1) Left(0,1) Top(1,0) Bottom(1,2) Front(1,1) Back(3,1)
    img = np.zeros((6, 8, 4), np.uint8)            # sw = 2, sh = 2
    for r in range(3):
        for c in range(4):
            img[2 * r:2 * r + 2, 2 * c:2 * c + 2, 0] = 10 * r + c
    m = rt.CubemapMaterial.from_cross(img)
    assert [int(f[0, 0, 0]) for f in m.faces] == [12, 10, 1, 21, 11, 13]
    assert all(f.shape == (2, 2, 4) for f in m.faces)
    with pytest.raises(ValueError):
        rt.CubemapMaterial.from_cross(np.zeros((5, 8, 4), np.uint8))
