"""GPU parity of the triangle/BVH path: the HIP kernels through the
C ABI against the CPU oracle on procedural scenes (meshes from OBJ text -> SAH BLAS per mesh ->
TLAS over rotated instances -> the reference's buffer layouts).  Tolerance: zero (RGBA8 frame
and ray count bit-identical), as for the sphere path."""
import ctypes

import numpy as np
import pytest

import compute_raytracer_amd as rt
from compute_raytracer_amd import abi
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
from helpers import diff_stats, gpu_render_tri, tri_buffers, triangle_scene

pytestmark = pytest.mark.gpu


def random_sky(seed, w=8, h=8):
    rng = np.random.default_rng(seed)
    m = rt.CubemapMaterial()
    m.faces = [rng.integers(0, 256, (h, w, 4), dtype=np.uint8) for _ in range(6)]
    return m


# the kernels of the triangle path: 0 = the library's choice, one workgroup per tile over the relinked pair records
# (rt_triangles.hip, PAIRS), 6 = the same kernel over the reference's node buffer
KERNELS = {0: "triangles", 6: "triangles"}


@pytest.mark.parametrize("variant", [0, 6])
@pytest.mark.parametrize("seed,W,H,B", [(1, 320, 200, 4), (2, 333, 207, 2), (3, 64, 64, 8), (4, 8, 8, 1), (5, 200, 120, 0)])
def test_triangle_scene_bit_exact(oracle, seed, W, H, B, variant):
    scene, mat = triangle_scene(seed=seed, n_models=3)
    sky = random_sky(seed)
    ref, _, rays = oracle.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)
    img, st = gpu_render_tri(scene, mat, W, H, B, skybox=sky, variant=variant)
    assert np.array_equal(img, ref), diff_stats(img, ref)
    assert st["rays"] == rays and abi.KERNEL_IDS[st["kernel_id"]] == KERNELS[variant]


@pytest.mark.parametrize("variant", [0, 6])
def test_finer_meshes_deeper_trees(oracle, variant):
    scene, mat = triangle_scene(seed=7, n_models=5, rings=24, sectors=32)
    assert scene.triangleCount > 3000
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    ref, _, rays = oracle.render_tri(scene.pack_params(4), tri_buffers(scene, mat), sky.faces, 480, 270)
    img, st = gpu_render_tri(scene, mat, 480, 270, 4, skybox=sky, variant=variant)
    assert np.array_equal(img, ref), diff_stats(img, ref)
    assert st["rays"] == rays and abi.KERNEL_IDS[st["kernel_id"]] == KERNELS[variant]


@pytest.mark.parametrize("variant", [0, 6])
def test_animation_loop_rebuilds_tlas_each_frame(oracle, variant):
    """src/app.ts:117-128: scene.update(dt) (models spin, TLAS + BLAS matrices rebuilt, SR:138-143),
    camera.move, renderer.render: per frame only params, BLAS records, BLAS lookup and TLAS nodes
    are re-uploaded (RR:157-192)."""
    scene, mat = triangle_scene(seed=9, n_models=3)
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    r = rt.RendererRaytracing(240, 136, scene, maxBounces=3).initialize(sky, mat)
    r.set_variant(variant)
    try:
        for frame in range(3):
            scene.update(0.25)
            scene.camera.move(0.1, 0.05)
            r.render()
            img = r.read_pixels()
            ref, _, rays = oracle.render_tri(scene.pack_params(3), tri_buffers(scene, mat), sky.faces, 240, 136)
            assert np.array_equal(img, ref), (frame, diff_stats(img, ref))
            assert r.stats()["rays"] == rays
    finally:
        r.close()


def test_heatmap_kernel(oracle):
    scene, mat = triangle_scene(seed=11, n_models=4, rings=12, sectors=16)
    ref, steps = oracle.heatmap_tri(scene.pack_params(4), tri_buffers(scene, mat), 320, 180)
    img, st = gpu_render_tri(scene, mat, 320, 180, 4, heatmap=True)
    assert np.array_equal(img, ref)
    assert steps.max() > 20


def test_switching_between_sphere_and_triangle_scenes(oracle):
    """One context, both primitive types: the type written last is the one rendered."""
    tri_scene, mat = triangle_scene(seed=13, n_models=2)
    sph_scene = rt.synthetic_scene(32, 99)
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    r = rt.RendererRaytracing(160, 96, tri_scene, maxBounces=2).initialize(sky, mat)
    try:
        r.render()
        a = r.read_pixels()
        ref_t, _, _ = oracle.render_tri(tri_scene.pack_params(2), tri_buffers(tri_scene, mat), sky.faces, 160, 96)
        assert np.array_equal(a, ref_t)
        r.scene = sph_scene; r.loaded = False
        r.render()
        b = r.read_pixels()
        ref_s, _, _ = oracle.render(sph_scene.pack_params(2), sph_scene.pack_spheres(), sky.faces, 160, 96)
        assert np.array_equal(b, ref_s)
        L = abi.load()
        r.showHeatmap()
        assert L.rt_render(r._ctx) == abi.RT_ERR_UNSUPPORTED            # no BVH to count in a sphere scene
        r.showRaytracer()
        r.scene = tri_scene; r.loaded = False
        r.render()
        assert np.array_equal(r.read_pixels(), ref_t)
    finally:
        r.close()


def test_incomplete_triangle_scene_is_a_state_error():
    L = abi.load()
    ctx = ctypes.c_void_p()
    abi.check(L.rt_create(0, ctypes.byref(ctx)))
    try:
        abi.check(L.rt_resize(ctx, 16, 16), ctx)
        p = np.zeros(24, np.float32)
        fp = ctypes.POINTER(ctypes.c_float)
        abi.check(L.rt_write_params(ctx, p.ctypes.data_as(fp)), ctx)
        for i in range(6):
            abi.check(L.rt_write_cubemap_face(ctx, i, 1, 1, p.ctypes.data), ctx)
        t = np.zeros(40, np.float32)
        abi.check(L.rt_write_triangles(ctx, t.ctypes.data_as(fp), 1), ctx)
        assert L.rt_render(ctx) == abi.RT_ERR_STATE and b"rt_write_triangles" in L.rt_last_error(ctx)
        assert L.rt_write_nodes(ctx, 8, t.ctypes.data_as(fp), 1) == abi.RT_ERR_INVALID_ARG
    finally:
        L.rt_destroy(ctx)


@pytest.mark.parametrize("n_streams", [2, 3])
def test_instance_updates_travel_with_the_frame(oracle, n_streams):
    """RR:169-192 rewrites BLAS records, BLAS lookup and TLAS nodes before every frame.  Those writes no longer
    drain: frames are enqueued back to back (rt_render_to, nothing waited for in between) while the host keeps
    rewriting the instance data; every frame must show the state it was enqueued with.  (Three host streams: a rotation
    whose period does not divide the four versions of the instance buffers -- the ordering must not lean on it)"""
    import torch
    scene, mat = triangle_scene(seed=21, n_models=3)
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    W, H, B, N = 200, 120, 3, 7                                       # 7 frames: the four buffer versions wrap
    r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
    try:
        r.render()                                                    # static part uploaded, first state applied
        base = r.stats()["instance_uploads"]
        bufs = [torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda") for _ in range(N)]
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        torch.cuda.synchronize()
        refs = []
        for f in range(N):
            scene.update(0.21)
            scene.camera.move(0.05, -0.02)
            refs.append(oracle.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)[0])
            r.render_to(bufs[f].data_ptr(), bufs[f].numel(), streams[f % n_streams].cuda_stream)      # no wait
        r.wait()
        torch.cuda.synchronize()
        st = r.stats()
        assert st["batch_frames"] == N and st["instance_uploads"] == base + N
        for f in range(N):
            img = bufs[f].cpu().numpy().reshape(H, W, 4)
            assert np.array_equal(img, refs[f]), (f, diff_stats(img, refs[f]))
        # an unchanged scene: each version is brought up to date once, then nothing travels any more
        for _ in range(6):
            r.enqueue()
        r.wait()
        assert r.stats()["instance_uploads"] <= base + N + 3
        assert np.array_equal(r.read_pixels(), refs[-1])
    finally:
        r.close()


def test_large_instance_sets_and_whole_buffer_node_writes(oracle):
    """More than 16 instances take the synchronous path; a host that writes the WHOLE node buffer in one call
    (head included) and later only the head must see both; stats name the kernel."""
    scene, mat = triangle_scene(seed=23, n_models=19, rings=4, sectors=5)
    assert len(scene.instances) == 20
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    W, H, B = 160, 96, 2
    r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
    try:
        for _ in range(2):
            scene.update(0.3)
            r.render()
            ref, _, rays = oracle.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)
            assert np.array_equal(r.read_pixels(), ref) and r.stats()["rays"] == rays
        assert r.stats()["kernel_id"] == 8 and abi.KERNEL_IDS[8] == "triangles"
    finally:
        r.close()
    small, mat = triangle_scene(seed=24, n_models=2)
    L = abi.load()
    fp = ctypes.POINTER(ctypes.c_float)
    r = rt.RendererRaytracing(W, H, small, maxBounces=B).initialize(sky, mat)
    try:
        r.render()                                                    # reference order of writes
        whole = np.ascontiguousarray(tri_buffers(small, mat)["nodes"])
        abi.check(L.rt_write_nodes(r._ctx, 0, whole.ctypes.data_as(fp), whole.shape[0]), r._ctx)   # head + body in one write
        small.update(0.4)
        r.render()                                                    # then the per-frame head write again
        ref, _, _ = oracle.render_tri(small.pack_params(B), tri_buffers(small, mat), sky.faces, W, H)
        assert np.array_equal(r.read_pixels(), ref)
    finally:
        r.close()


@pytest.mark.parametrize("variant", [0, 6])
def test_tile_order_does_not_change_the_picture(oracle, variant):
    """From 4096 tiles on the triangle kernel starts a frame's tiles longest-first, in the order the previous frame on the
    same stream suggests (rt_triangles.hip: order_tiles).  1024 x 516 = 8320 tiles, ragged last row; the camera walks and
    the models spin, so every frame is rendered in an order made for another picture: ten frames one at a time (each of
    the four streams comes round at least twice), then six in flight, each against the oracle."""
    W, H, B = 1024, 516, 3
    scene, mat = triangle_scene(seed=21, n_models=3)
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
    r.set_variant(variant)
    try:
        for frame in range(10):
            scene.update(0.2)
            scene.camera.move(0.08, -0.03)
            r.render()
            img = r.read_pixels()
            ref, _, rays = oracle.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)
            assert np.array_equal(img, ref), (frame, diff_stats(img, ref))
            assert r.stats()["rays"] == rays
        want, host = [], r.host_frames(6)
        for frame in range(6):
            scene.update(0.2)
            scene.camera.move(-0.05, 0.04)
            r.recalculateScene()
            r.enqueue()
            if frame >= 2:
                r.read_pixels_async(2, host[frame - 2])
            want.append(oracle.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)[0])
        r.read_pixels_async(1, host[4])
        r.read_pixels_async(0, host[5])
        r.wait()
        r.read_pixels_wait()
        for frame in range(6):
            assert np.array_equal(host[frame].reshape(H, W, 4), want[frame].reshape(H, W, 4)), frame
    finally:
        r.close()


@pytest.mark.parametrize("heatmap", [False, True])
def test_scenes_beyond_the_packed_stack_take_the_index_stack(oracle, heatmap):
    """The BLAS traversal keeps (count, left) of a pushed child in a 32-bit stack entry while every count, node index and
    lookup slot fits 16 bits (rt_api.hip: packed_ok); a lookup table of more than 65,536 entries -- here: the scene's own
    plus unused padding -- takes the form that pushes indices.  Both render the reference's frame."""
    scene, mat = triangle_scene(seed=5, n_models=2, rings=7, sectors=9)
    scene.static["tri_lookup"] = np.concatenate([np.asarray(scene.static["tri_lookup"], np.float32), np.zeros(70000, np.float32)])
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    W, H, B = 200, 120, 3
    b = tri_buffers(scene, mat)
    if heatmap:
        ref, _ = oracle.heatmap_tri(scene.pack_params(B), b, W, H)
        img, _ = gpu_render_tri(scene, mat, W, H, B, skybox=sky, heatmap=True)
    else:
        ref, _, rays = oracle.render_tri(scene.pack_params(B), b, sky.faces, W, H)
        img, st = gpu_render_tri(scene, mat, W, H, B, skybox=sky)
        assert st["rays"] == rays
    assert np.array_equal(img, ref), diff_stats(img, ref)


def spine_scene(depth):
    """A BLAS no builder would make: a spine of `depth` inner nodes, each with a leaf as its FARTHER child, so that the
    walk pushes one entry per level -- beyond the eight slots the persistent kernel keeps in LDS, and beyond the twenty the
    reference's stack has at all (RK:71; RK:303-306 pushes without a guard: the index clamps to the last slot, and the pops
    that follow read that slot again and again).  One triangle per leaf, each covering its own part of the view."""
    base = 1                                                    # tlasNodesMax of one instance
    nodes = np.zeros((1 + 2 * depth, 8), np.float32)            # S_0, then the pairs (A_k, S_{k+1}); the last "S" is a leaf
    tris = np.zeros((depth + 1, 40), np.float32)
    def box(i, lo, hi, left, count):
        nodes[i, 0:3] = lo; nodes[i, 3] = left; nodes[i, 4:7] = hi; nodes[i, 7] = count
    box(0, [-9, -9, -1.0], [9, 9, 5.0], base + 1, 0)
    for k in range(depth):
        a, s = 1 + 2 * k, 2 + 2 * k
        box(a, [-9, -9, -3.0], [9, 9, -2.0], k, 1)                                   # leaf A_k: lookup slot k
        if k + 1 < depth: box(s, [-9, -9, -1.0], [9, 9, 5.0], base + s + 1, 0)       # S_{k+1}
        else: box(s, [-9, -9, -1.0], [9, 9, 5.0], depth, 1)                          # the bottom: a leaf inside the near box
    for k in range(depth + 1):
        z = -2.05 - 0.9 * k / depth if k < depth else -0.5
        x0 = -8.0 + 16.0 * ((k * 7) % (depth + 1)) / (depth + 1)
        w = 3.0 if k < depth else 40.0
        # front face towards +z (RK:359 culls det < 1e-5)
        for c, (x, y) in enumerate([(x0, -8.0), (x0 + w, -8.0), (x0 + w / 2, 9.0)]):
            tris[k, 12 * c:12 * c + 3] = [x, y, z]
            tris[k, 12 * c + 4:12 * c + 7] = [0, 0, 1]
            tris[k, 12 * c + 8:12 * c + 10] = [c / 2.0, c % 2]
        tris[k, 36:40] = [0.2 + 0.8 * ((k * 5) % 7) / 7.0, 0.3 + 0.7 * ((k * 3) % 5) / 5.0, 0.9 - 0.6 * (k % 4) / 4.0, 1.0 if k % 3 else 0.5]
    d = dict(triangles=tris, blas_nodes=nodes, tri_lookup=np.arange(depth + 1, dtype=np.float32),
             mesh_root=np.array([base]), mesh_box_lo=np.array([[-9.0, -9.0, -3.0]]), mesh_box_hi=np.array([[9.0, 9.0, 5.0]]),
             inst_mesh=np.array([0]), inst_position=np.array([[0.0, 0.0, 0.0]]), inst_eulers=np.array([[0.0, 0.0, 0.0]]),
             inst_speed=np.array([[0.0, 0.0, 0.0]]), camera_position=np.array([0.0593, 2.692, 3.293]),
             camera_eulers=np.array([0.0, 106.0, 270.0], np.float32), light=np.array([0.0, 5.0, 6.0, 3.0, 0.3]))
    return rt.SceneRaytracing.from_packed(d)


@pytest.mark.parametrize("variant", [0, 6])
@pytest.mark.parametrize("depth", [7, 12, 19, 20, 21, 33])
def test_stack_depth_beyond_the_lds_slots_and_beyond_the_reference_stack(oracle, depth, variant):
    scene = spine_scene(depth)
    mat = rt.Material(np.random.default_rng(depth).integers(0, 256, (8, 8, 4), dtype=np.uint8))
    sky = random_sky(depth)
    W, H, B = 160, 96, 3
    ref, _, rays = oracle.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)
    assert len(np.unique(ref.reshape(-1, 4), axis=0)) > 20           # the spine is in view
    img, st = gpu_render_tri(scene, mat, W, H, B, skybox=sky, variant=variant)
    assert np.array_equal(img, ref), diff_stats(img, ref)
    assert st["rays"] == rays and abi.KERNEL_IDS[st["kernel_id"]] == KERNELS[variant]


@pytest.mark.parametrize("per_leaf", [3, 4, 7, -3])
def test_leaves_of_more_triangles_than_a_two_byte_stack_entry_counts(oracle, per_leaf):
    """The five-waves-per-SIMD form of the tile kernel keeps (count << 14 | x) in two bytes per stack entry: leaves of at most
    three triangles (the reference's builder stops at two).  A hand-made tree with three per leaf still takes it, four and seven
    per leaf take the four-byte entries (rt_api.hip: p16_ok); same frame either way."""
    wild = per_leaf < 0                                          # -3: three per leaf, and one leaf whose first slot lies far beyond the lookup
    per_leaf = abs(per_leaf)                                     # table (the oracle clamps it to the last slot; 14 bits would wrap it)
    depth = 9                                                    # spine_scene's tree with `per_leaf` triangles in every leaf, side by side
    nodes = np.zeros((1 + 2 * depth, 8), np.float32)
    tris = np.zeros(((depth + 1) * per_leaf, 40), np.float32)
    def box(i, lo, hi, left, count):
        nodes[i, 0:3] = lo; nodes[i, 3] = left; nodes[i, 4:7] = hi; nodes[i, 7] = count
    box(0, [-9, -9, -1.0], [9, 9, 5.0], 2, 0)
    for k in range(depth):
        a, sidx = 1 + 2 * k, 2 + 2 * k
        box(a, [-9, -9, -3.0], [9, 9, -2.0], k * per_leaf, per_leaf)
        if k + 1 < depth: box(sidx, [-9, -9, -1.0], [9, 9, 5.0], 1 + sidx + 1, 0)
        else: box(sidx, [-9, -9, -1.0], [9, 9, 5.0], depth * per_leaf, per_leaf)
    for k in range(depth + 1):
        for j in range(per_leaf):
            z = (-2.05 - 0.9 * k / depth if k < depth else -0.5) - 0.01 * j
            x0 = -8.0 + 16.0 * ((k * 7) % (depth + 1)) / (depth + 1) + 0.7 * j
            w = 2.0 if k < depth else 30.0
            t = k * per_leaf + j
            for c, (x, y) in enumerate([(x0, -8.0), (x0 + w, -8.0), (x0 + w / 2, 9.0)]):
                tris[t, 12 * c:12 * c + 3] = [x, y, z]
                tris[t, 12 * c + 4:12 * c + 7] = [0, 0, 1]
                tris[t, 12 * c + 8:12 * c + 10] = [c / 2.0, c % 2]
            tris[t, 36:40] = [0.2 + 0.8 * ((t * 5) % 7) / 7.0, 0.3 + 0.7 * ((t * 3) % 5) / 5.0, 0.9 - 0.6 * (t % 4) / 4.0, 1.0 if t % 3 else 0.5]
    if wild:
        nodes[5, 3] = 30000.0
    dd = dict(triangles=tris, blas_nodes=nodes, tri_lookup=np.arange(tris.shape[0], dtype=np.float32),
              mesh_root=np.array([1]), mesh_box_lo=np.array([[-9.0, -9.0, -3.0]]), mesh_box_hi=np.array([[9.0, 9.0, 5.0]]),
              inst_mesh=np.array([0]), inst_position=np.array([[0.0, 0.0, 0.0]]), inst_eulers=np.array([[0.0, 0.0, 0.0]]),
              inst_speed=np.array([[0.0, 0.0, 0.0]]), camera_position=np.array([0.0593, 2.692, 3.293]),
              camera_eulers=np.array([0.0, 106.0, 270.0], np.float32), light=np.array([0.0, 5.0, 6.0, 3.0, 0.3]))
    scene = rt.SceneRaytracing.from_packed(dd)
    mat = rt.Material(np.random.default_rng(per_leaf).integers(0, 256, (8, 8, 4), dtype=np.uint8))
    sky = random_sky(per_leaf)
    W, H, B = 160, 96, 3
    ref, _, rays = oracle.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)
    assert len(np.unique(ref.reshape(-1, 4), axis=0)) > 20
    for variant in (0, 6):
        img, st = gpu_render_tri(scene, mat, W, H, B, skybox=sky, variant=variant)
        assert np.array_equal(img, ref), (variant, diff_stats(img, ref))
        assert st["rays"] == rays


@pytest.mark.parametrize("variant", [0, 6])
def test_an_instance_that_changes_its_mesh_rebuilds_the_relinked_copy(oracle, variant):
    """The library's relinked copy of the BLAS trees is built from the roots the instance records name.  The reference rewrites
    those records before every frame (RR:169-174) and nothing in the interface says a root may not change: a frame that names a
    root the copy does not know must rebuild it (rt_api.hip: rt_flow_covers) -- here model 0 switches from the coarse sphere to
    the fine one, which no instance had referenced before, and back.  The rebuild keeps the roots it knew: alternating between
    two root sets costs ONE rebuild, not one per frame (rt_stats.pair_rebuilds)."""
    scene, mat = triangle_scene(seed=33, n_models=1)              # models: [sphere mesh 0, floor (mesh 2)]; mesh 1 unreferenced
    assert sorted(set(int(k) for k in scene.instances.mesh_index)) == [0, 2]
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    W, H, B = 200, 120, 3
    r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
    r.set_variant(variant)
    try:
        for mesh in (0, 1, 0, 1):
            scene.instances.mesh_index[0] = mesh
            scene.update(0.1)
            r.render()
            ref, _, rays = oracle.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)
            assert np.array_equal(r.read_pixels(), ref), (mesh, diff_stats(r.read_pixels(), ref))
            assert r.stats()["rays"] == rays
            assert abi.KERNEL_IDS[r.stats()["kernel_id"]] == KERNELS[variant]
        assert r.stats()["pair_rebuilds"] == (2 if variant == 0 else 0)
    finally:
        r.close()


def deepen_top_level(scene, levels):
    """The frame's top-level tree under `levels` extra inner nodes: each new node has the old tree (one level down) as its first
    child and a leaf far away from everything (instance 0 again: never entered) as its second.  Same picture, a deeper walk."""
    t = np.asarray(scene.frame["tlas_nodes"], np.float32).reshape(-1, 8)
    n_old, extra = t.shape[0], 2 * levels
    assert n_old + extra <= scene.tlasNodesMax
    out = np.zeros((n_old + extra, 8), np.float32)
    for k in range(levels):                                  # node 0 and the chain nodes at 1, 3, 5, ...: children at (2k+1, 2k+2)
        i = 0 if k == 0 else 2 * k - 1
        out[i] = [-1e4, -1e4, -1e4, 2 * k + 1, 1e4, 1e4, 1e4, 0]
        out[2 * k + 2] = [9e3, 9e3, 9e3, 0, 9.1e3, 9.1e3, 9.1e3, 1]      # the far leaf
    base = 2 * levels - 1                                    # where the old root goes; the rest of the old tree behind the chain
    remap = lambda i: base if i == 0 else extra + i
    for i in range(n_old):
        row = t[i].copy()
        if row[7] == 0:
            row[3] = remap(int(row[3]))                      # old children sit side by side at left, left + 1 (left >= 1)
        out[remap(i)] = row
    scene.frame["tlas_nodes"] = out


def expected_form(scene, mat):
    """rt_tlas_fit.h restated: the stack form the library must pick for the frame's top-level tree (tiny 2, small 1, neither 0)."""
    nodes = tri_buffers(scene, mat)["nodes"]
    n = len(nodes)
    def u32f(f):
        f = float(f)
        return 0 if not f > 0.0 else (4294967295 if f >= 4294967040.0 else int(f))
    def fits(max_depth, max_nodes):
        todo = [(0, 0)]
        while todo:
            i, d = todo.pop()
            i = min(i, n - 1)
            if i >= max_nodes: return False
            if u32f(nodes[i, 7]) != 0: continue
            if d >= max_depth: return False
            left = u32f(nodes[i, 3])
            todo += [(left, d + 1), ((left + 1) & 0xFFFFFFFF, d + 1)]
        return True
    if len(scene.instances) > 16: return 0
    if len(scene.instances) > 12: return 4 if fits(8, 32) else 0       # 13-16 instances: the form that stages sixteen records
    if len(scene.instances) <= 4 and fits(3, 8): return 2
    if fits(4, 16): return 1
    if fits(8, 24): return 3
    return 4 if fits(8, 32) else 0


@pytest.mark.parametrize("n_models,deepen", [(1, 0), (3, 0), (3, 2), (4, 2), (11, 0), (11, 3), (11, 7), (14, 0), (15, 1)])
def test_every_stack_form_of_the_kernel(oracle, n_models, deepen):
    """The host walks every frame's top-level tree (rt_tlas_fit.h) and picks the kernel's stack form (rt_stats.tri_form): up to 4
    instances in a tree of depth <= 3 -- three TLAS slots, six waves per SIMD for frames in flight, five (the four-slot form) for
    awaited ones --, a tree of depth <= 4 within 16 nodes -- four slots, five waves --, depth <= 8 within 24 nodes -- eight slots,
    nine of the eleven carried values parked --, 13-16 instances (or 32 nodes) the same stack with sixteen staged records and six
    parked values --, anything else the reference's twenty.
    n_models + the floor instances, some under extra levels; each awaited and in flight, under a textured sky, against the oracle."""
    scene, mat = triangle_scene(seed=40 + n_models, n_models=n_models, rings=5, sectors=7)
    sky = random_sky(n_models)
    W, H, B = 200, 120, 3
    r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
    seen = set()
    def advance():
        scene.update(0.3)
        levels = min(deepen, (scene.tlasNodesMax - len(scene.frame["tlas_nodes"])) // 2)
        if levels:
            deepen_top_level(scene, levels)
    try:
        for frame in range(2):
            advance()
            r.render()
            ref, _, rays = oracle.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)
            assert np.array_equal(r.read_pixels(), ref), (frame, diff_stats(r.read_pixels(), ref))
            assert r.stats()["rays"] == rays
            want = expected_form(scene, mat)
            assert r.stats()["tri_form"] == (1 if want == 2 else want)                  # awaited: never the six-wave form
            seen.add(r.stats()["tri_form"])
        for batch in range(2):                    # the second batch: the library has seen frames in flight (pipelined_hint)
            host, want = r.host_frames(4), []
            for f in range(4):
                advance()
                r.recalculateScene(); r.enqueue()
                r.read_pixels_async(0, host[f])
                want.append(oracle.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)[0])
            r.wait(); r.read_pixels_wait()
            for f in range(4):
                assert np.array_equal(host[f].reshape(H, W, 4), want[f]), (batch, f)
        assert r.stats()["tri_form"] == expected_form(scene, mat)
        seen.add(r.stats()["tri_form"])
        if (n_models, deepen) == (1, 0): assert seen == {1, 2}
        if (n_models, deepen) == (11, 3): assert 3 in seen
        if (n_models, deepen) == (11, 7): assert 0 in seen
        if n_models >= 14: assert 4 in seen
    finally:
        r.close()


def test_a_lookup_table_longer_than_the_instance_list(oracle):
    """The small forms of the kernel stage one BLAS-lookup entry per instance record (rt_tri_types.h: RtTriInst) and read no other
    per-frame buffer.  A host whose lookup table has MORE entries than it has instances -- here every top-level leaf names its
    instances through the table's second half, a copy of the first -- must get the form that reads the table itself."""
    scene, mat = triangle_scene(seed=61, n_models=2, rings=5, sectors=7)
    sky = random_sky(61)
    W, H, B = 200, 120, 3
    r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
    try:
        for frame in range(3):
            scene.update(0.3)
            look = np.asarray(scene.frame["blas_lookup"], np.float32)
            m = len(look)
            scene.frame["blas_lookup"] = np.concatenate([look, look])
            t = np.asarray(scene.frame["tlas_nodes"], np.float32).copy()
            leaf = t[:, 7] > 0
            t[leaf, 3] += m                                   # the leaves' first lookup slot: into the copy
            scene.frame["tlas_nodes"] = t
            r.render()
            ref, _, rays = oracle.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)
            assert np.array_equal(r.read_pixels(), ref), (frame, diff_stats(r.read_pixels(), ref))
            assert r.stats()["rays"] == rays and r.stats()["tri_form"] == 0
    finally:
        r.close()


def test_a_host_that_rewrites_its_trees_every_frame_is_not_made_to_rebuild_the_copy_every_frame(oracle):
    """The relinked copy of the BLAS trees is invalidated by any write that reaches the nodes it was built from.  A host that
    rewrites those nodes before EVERY frame (nothing in the interface forbids it) must not pay a drain, a rebuild and an upload per
    frame: after four such frames in a row the library walks the reference's node buffer for a while (rt_api.hip) -- same pixels."""
    scene, mat = triangle_scene(seed=71, n_models=2, rings=5, sectors=7)
    sky = random_sky(71)
    W, H, B = 200, 120, 3
    r = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky, mat)
    L = abi.load()
    fp = ctypes.POINTER(ctypes.c_float)
    try:
        r.render()
        base = r.stats()["pair_rebuilds"]
        assert base == 1 and r.stats()["tri_form"] == 1
        nodes = np.ascontiguousarray(scene.pack_blas_nodes(), np.float32)
        ref, _, rays = oracle.render_tri(scene.pack_params(B), tri_buffers(scene, mat), sky.faces, W, H)
        forms = []
        for frame in range(10):
            abi.check(L.rt_write_nodes(r._ctx, 32 * scene.tlasNodesMax, nodes.ctypes.data_as(fp), nodes.shape[0]), r._ctx)   # the same trees, written again
            r.render()
            assert np.array_equal(r.read_pixels(), ref), (frame, diff_stats(r.read_pixels(), ref))
            assert r.stats()["rays"] == rays
            forms.append(r.stats()["tri_form"])
        assert r.stats()["pair_rebuilds"] == base + 3           # four frames in a row rebuilt (the first frame's build included), then the node walk
        assert forms[:3] == [1, 1, 1] and set(forms[3:]) == {0}
        for frame in range(3):                                  # the host stops rewriting: the copy stays stale for the cool-down, pixels right
            r.render()
            assert np.array_equal(r.read_pixels(), ref)
    finally:
        r.close()
