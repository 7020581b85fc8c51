"""The hierarchy kernel under a TEXTURED sky: bvh_pixels leaves end-of-path records and sky_resolve samples the
cube map and composes the pixel (compute_raytracer_amd/csrc/rt_bvh.hip).  Every frame must still
be the oracle's, bit for bit -- with more frames in flight than there are record buffers, on caller streams, for
zero bounces (no ray is cast: the pixel is fog over white), for ragged sizes and partitions, and across a
resize that reallocates the records."""
import numpy as np
import pytest

import compute_raytracer_amd as rt
from compute_raytracer_amd import abi
from compute_raytracer_amd.scene_raytracing import synthetic_spheres
from helpers import diff_stats, gpu_render, oracle_render

BVH = 4

pytestmark = pytest.mark.gpu


def textured(seed, n=16):
    rng = np.random.default_rng(seed)
    sky = rt.CubemapMaterial()
    sky.faces = [rng.integers(0, 256, (n, n, 4), dtype=np.uint8) for _ in range(6)]
    return sky


def move(scene, f):
    scene.camera.position = [0.06 + 0.4 * f, 2.7 + 0.1 * f, 3.3 - 0.3 * f]
    scene.camera.eulers = np.array([7.0 * f, 90.0 - 3.0 * f], dtype=np.float32)     # absolute: the frames are replayed below
    scene.camera.update()
    scene.light.position = [0.5 * f, 5.0, -0.5 * f]


def test_more_frames_in_flight_than_record_buffers(oracle):
    """Eleven frames on four caller streams, then the plain API: a frame reuses the record buffer of the frame four
    slots back and must wait for it -- and for nothing else."""
    import torch
    sky = textured(3)
    scene = rt.SceneRaytracing().createScene(synthetic_spheres(300, 17))
    W, H, B, F = 256, 160, 5, 11
    r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky)
    r.set_variant(BVH)
    streams = [torch.cuda.Stream() for _ in range(4)]
    bufs = [torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda") for _ in range(F)]
    torch.cuda.synchronize()
    refs = []
    for f in range(F):
        move(scene, f)
        refs.append(oracle_render(oracle, scene, W, H, B, skybox=sky))
        r.render_to(bufs[f].data_ptr(), bufs[f].numel(), streams[(3 * f) % 4].cuda_stream)
    r.wait()
    torch.cuda.synchronize()
    for f in range(F):
        img = bufs[f].cpu().numpy().reshape(H, W, 4)
        assert np.array_equal(img, refs[f][0]), (f, diff_stats(img, refs[f][0]))
    assert r.stats()["rays"] == refs[F - 1][2]
    for last in (1, 4, 5, 9):                      # batches that end on different buffer sets
        for f in range(last):
            move(scene, f)
            r.recalculateScene()
            r.enqueue()
        assert np.array_equal(r.read_pixels(), refs[last - 1][0]), last
    r.close()


@pytest.mark.parametrize("bounces", [0, 1, 2, 16])
def test_bounce_limits_under_a_textured_sky(oracle, bounces):
    """0 bounces: no ray, white under fog = the fog colour at distance 0 ... i.e. pure colour; 1 bounce: every miss is
    the bounce-0 miss that reuses the fog sample; 16: misses after many halvings of `affect`."""
    sky = textured(5, 8)
    scene = rt.SceneRaytracing().createScene(synthetic_spheres(200, 9))
    W, H = 203, 117
    ref, _, rays = oracle_render(oracle, scene, W, H, bounces, skybox=sky)
    img, st = gpu_render(scene, W, H, bounces, strict=False, skybox=sky, variant=BVH)
    assert np.array_equal(img, ref), diff_stats(img, ref)
    assert st["rays"] == rays


def test_partitions_and_resize_under_a_textured_sky(oracle):
    sky = textured(7, 4)
    scene = rt.SceneRaytracing().createScene(synthetic_spheres(260, 21))
    B, world = 4, 3
    for W, H in ((200, 123), (96, 41)):
        ref, _, rays = oracle_render(oracle, scene, W, H, B, skybox=sky)
        total = 0
        frame = np.zeros((H, W, 4), np.uint8)
        for rank in range(world):
            img, st = gpu_render(scene, W, H, B, strict=False, skybox=sky, variant=BVH, rank=rank, world=world)
            rows = [y for t in range(rank, (H + 7) // 8, world) for y in range(t * 8, min(t * 8 + 8, H))]
            frame[rows] = img.reshape(-1, W, 4)[: len(rows)]
            total += st["rays"]
        assert np.array_equal(frame, ref) and total == rays
    # one context: small frame, larger frame (the records are reallocated), small again
    r = rt.RendererRaytracing(64, 48, scene, maxBounces=B).initialize(sky)
    r.set_variant(BVH)
    for W, H in ((64, 48), (333, 207), (64, 48)):
        abi.check(abi.load().rt_resize(r._ctx, W, H), r._ctx)
        r.width, r.height = W, H
        r.render()
        ref = oracle_render(oracle, scene, W, H, B, skybox=sky)[0]
        assert np.array_equal(r.read_pixels(), ref), (W, H)
    r.close()


def test_gather_path_under_a_textured_sky(oracle):
    """rt_render_gather (communicator of one): render + sky_resolve + exchange + de-interleave on the rotating
    streams, six frames in flight with different cameras; the assembled frame is the oracle's."""
    sky = textured(11, 8)
    scene = rt.SceneRaytracing().createScene(synthetic_spheres(220, 5))
    W, H, B = 176, 99, 4
    r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky)
    r.set_variant(BVH)
    r.comm_init(rt.RendererRaytracing.comm_unique_id(), 0, 1)
    try:
        for root in (0, -1):
            for f in range(6):
                move(scene, f)
                r.recalculateScene()
                r.render_gather(root)
            ref = oracle_render(oracle, scene, W, H, B, skybox=sky)[0]
            assert np.array_equal(r.read_frame(), ref), root
    finally:
        r.close()
