"""Awaited frames and batches of frames in flight, alternating, every frame copied out by the streaming read-back and compared
with the oracle.  What rt_render relies on since awaited frames stay on one stream (rt_api.hip: rt_render): the stream a frame
runs on (stream_rot), the colour buffer it renders into (frames_rendered % 4), the end-of-path record set of a textured sky and
the instance version of a triangle scene (event slot % 4) are three independent indices; a frame must be ordered behind whatever
used ITS buffer, record set and version before, on whatever stream that was, and behind a copy still reading its buffer."""
import numpy as np
import pytest

import compute_raytracer_amd as rt
from compute_raytracer_amd import abi
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
from helpers import diff_stats, triangle_scene, tri_buffers

pytestmark = pytest.mark.gpu

BATCHES = [1, 5, 1, 9, 2, 1, 1, 6, 1]


def random_sky(seed, n=8):
    rng = np.random.default_rng(seed)
    m = rt.CubemapMaterial()
    m.faces = [rng.integers(0, 256, (n, n, 4), dtype=np.uint8) for _ in range(6)]
    return m


def run_batches(r, W, H, advance, reference, batches=BATCHES):
    """advance(): moves the scene one step (host state only); reference() -> the oracle's frame of the current state."""
    host = r.host_frames(max(batches))
    frame_no = 0
    for b in batches:
        want = []
        if b == 1:                                   # the reference's loop: recalculateScene, render, await (RR:435-469)
            advance()
            r.render()
            r.read_pixels_async(0, host[0])
            want.append(reference())
            r.read_pixels_wait()
        else:                                        # b frames enqueued back to back, each copied out behind its kernels
            for i in range(b):
                advance()
                r.recalculateScene()
                r.enqueue()
                r.read_pixels_async(0, host[i])
                want.append(reference())
            r.wait()
            r.read_pixels_wait()
        for i, w in enumerate(want):
            got = host[i].reshape(H, W, 4)
            assert np.array_equal(got, w), ("frame %d (batch of %d, #%d)" % (frame_no + i, b, i), diff_stats(got, w))
        frame_no += b
    return frame_no


def test_sphere_hierarchy_under_a_textured_sky(oracle):
    """300 spheres: the hierarchy kernel; an 8x8-texel sky: end-of-path records + sky_resolve (four record sets, by event slot)."""
    W, H, B = 336, 200, 5
    scene = rt.synthetic_scene(300, 4242)
    sky = random_sky(5)
    r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky)
    try:
        step = [0]
        def advance():
            step[0] += 1
            scene.camera.move(0.05 * ((step[0] % 3) - 1), 0.04)
        def reference():
            return oracle.render(scene.pack_params(B), scene.pack_spheres(), sky.faces, W, H)[0]
        n = run_batches(r, W, H, advance, reference)
        assert n == sum(BATCHES)
        assert abi.KERNEL_IDS[r.stats()["kernel_id"]].startswith("hierarchy")
    finally:
        r.close()


@pytest.mark.parametrize("textured", [False, True])
def test_animated_triangle_scene(oracle, textured):
    """8,320 tiles: awaited frames use the work list of the previous frame on their stream, frames in flight none; the models
    spin, so every frame carries new instance data (four versions, by event slot) and renders in an order made for another picture.
    Awaited frames take the five-wave form, frames in flight the six-wave form of the kernel (rt_triangles.hip)."""
    W, H, B = 1024, 516, 3
    scene, mat = triangle_scene(seed=21, n_models=3)
    sky = random_sky(9) if textured else rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
    try:
        def advance():
            scene.update(0.2)
            scene.camera.move(0.06, -0.02)
        def reference():
            return oracle.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)[0]
        n = run_batches(r, W, H, advance, reference, batches=[1, 5, 1, 6, 1, 1])
        assert n == 15
        assert abi.KERNEL_IDS[r.stats()["kernel_id"]] in ("triangles", "triangles_roles")
        assert r.stats()["instance_uploads"] >= 15
    finally:
        r.close()
