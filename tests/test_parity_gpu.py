"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.

Tolerance (stated here, as BASELINE.json's north_star asks): ZERO for both arithmetic modes.
  * RT_MODE_STRICT evaluates the reference arithmetic literally (no FMA contraction, IEEE
    division and square root) -> RGBA8 frame, and ray count, bit-identical to the oracle.
  * RT_MODE_FAST (the default, the one bench.py measures) puts a conservative fused-arithmetic
    filter in front of the same literal evaluation; the filter never decides a pixel, so the
    result is bit-identical as well.  Any difference is a bug, not rounding.
Floating point enters only before the rgba8unorm store; the frame itself is bytes.

Sizes: the oracle is run where it finishes in seconds; BASELINE's full-size configs (C2, C3) are
checked against the committed golden hashes of the oracle's frames and through size-independent
properties (fast == strict, determinism, tile-partition invariance).
"""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

import compute_raytracer_amd as rt
from compute_raytracer_amd import abi, tiles
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
from helpers import config_inputs, diff_stats, gpu_render, oracle_render

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def random_sky(seed, w=6, h=6):
    """six equal squares by default: a WebGPU cube texture, filtered seamlessly across its edges"""
    rng = np.random.default_rng(seed)
    m = rt.CubemapMaterial()
    m.faces = [rng.integers(0, 256, (h, w, 4), dtype=np.uint8) for _ in range(6)]
    return m


# ---- BASELINE configs against the oracle ---------------------------------------------------------
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fast"])
@pytest.mark.parametrize("name", ["C1", "C2"])
def test_baseline_configs_bit_exact(oracle, name, strict):
    cfg, scene = config_inputs(name)
    W, H, B = cfg["width"], cfg["height"], cfg["bounces"]
    ref, _, rays = oracle_render(oracle, scene, W, H, B)
    img, st = gpu_render(scene, W, H, B, strict=strict)
    assert diff_stats(img, ref)["max"] == 0
    assert np.array_equal(img, ref)
    assert st["rays"] == rays


@pytest.mark.parametrize("name", ["C1", "C2", "C3"])
def test_full_size_frames_match_golden_hashes(name):
    """C3 is the headline 3840x2160 / 1024 spheres / 8 bounces frame: the GPU frame must hash to
    the oracle's frame (tests/golden/frames.json) -- full-size parity without re-running the
    oracle on the GPU box."""
    fr = json.load(open(os.path.join(G, "frames.json")))[name]
    cfg, scene = config_inputs(name)
    for strict in (False, True):
        img, st = gpu_render(scene, cfg["width"], cfg["height"], cfg["bounces"], strict=strict)
        assert hashlib.sha256(img.tobytes()).hexdigest() == fr["sha256"], (name, strict)
        assert st["rays"] == fr["rays"]
    if name == "C1":
        from PIL import Image
        want = np.array(Image.open(os.path.join(G, "c1_frame.png")).convert("RGBA"), dtype=np.uint8)
        assert np.array_equal(img, want)


@pytest.mark.parametrize("name", ["C2", "C3"])
def test_sparse_golden_pixels(name):
    g = json.load(open(os.path.join(G, "sparse_%s.json" % name)))
    cfg, scene = config_inputs(name)
    img, _ = gpu_render(scene, cfg["width"], cfg["height"], cfg["bounces"], strict=False)
    for px in g["pixels"]:
        assert list(img[px["y"], px["x"]]) == px["rgba8"], px


# ---- edge cases -------------------------------------------------------------------------------------
EDGE = [
    # W, H, N, B, sky
    (333, 207, 20, 4, "const"),     # W, H not multiples of 8: threads outside the texture store nothing (RR:445)
    (8, 8, 64, 8, "random"),        # a single wave
    (1, 1, 9, 3, "random"),         # a single pixel
    (257, 9, 0, 4, "random"),       # empty scene: sky * minIntensity everywhere
    (64, 40, 1, 2, "const"),        # only the ground sphere
    (100, 60, 5, 16, "random"),     # N not a multiple of the 8-sphere filter batch
    (96, 64, 130, 2, "random"),
    (64, 64, 64, 0, "const"),       # maxBounces 0: white (RK:103,113)
    (64, 64, 64, 1, "const"),
    (72, 56, 1000, 3, "random"),
    (96, 64, 20, 3, "noncube"),     # 6 x 5 images: not a WebGPU cube, taps clamp inside the face
]


@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fast"])
@pytest.mark.parametrize("W,H,N,B,sky", EDGE)
def test_edge_cases_bit_exact(oracle, W, H, N, B, sky, strict):
    scene = rt.synthetic_scene(N, 9000 + N) if N else rt.synthetic_scene(1, 1)
    if N == 0:
        scene.spheres = []
    skybox = (rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA) if sky == "const" else
              random_sky(W * 31 + N, 6, 5) if sky == "noncube" else random_sky(W * 31 + N))
    ref, _, rays = oracle_render(oracle, scene, W, H, B, skybox=skybox)
    img, st = gpu_render(scene, W, H, B, strict=strict, skybox=skybox)
    assert img.shape == ref.shape
    assert np.array_equal(img, ref), diff_stats(img, ref)
    assert st["rays"] == rays


def test_spheres_touching_camera_and_light(oracle):
    """Origins inside / on spheres: the near-root-only rule (HK:317) and the c <= 0 branch of
    the filter (a sphere whose c is not positive always goes to the literal evaluation)."""
    scene = rt.synthetic_scene(24, 77)
    cam = scene.camera.position
    scene.spheres += [rt.Sphere(cam, 0.5, [1, 0, 0]),                       # camera at the centre
                      rt.Sphere([cam[0], cam[1], cam[2] - 0.6], 0.6, [0, 1, 0]),   # camera on the surface
                      rt.Sphere([0, 5, 0], 0.7, [0, 0, 1]),                 # light inside
                      rt.Sphere([0.3, 4.2, -0.2], 0.5, [1, 1, 0])]          # next to the light
    for strict in (True, False):
        ref, _, rays = oracle_render(oracle, scene, 160, 96, 6)
        img, st = gpu_render(scene, 160, 96, 6, strict=strict)
        assert np.array_equal(img, ref) and st["rays"] == rays


def test_lds_stress_4096_spheres_with_textured_skybox(oracle):
    """BASELINE C5's shape at a size the oracle finishes quickly: 4096 spheres (128 KiB of filter
    records in LDS, camera records from global memory), 16 bounces, bilinear cube map."""
    cfg, scene = config_inputs("C5")
    sky = random_sky(5, w=64, h=64)
    W, H, B = 320, 180, 16
    ref, _, rays = oracle_render(oracle, scene, W, H, B, skybox=sky)
    for strict in (True, False):
        img, st = gpu_render(scene, W, H, B, strict=strict, skybox=sky)
        assert np.array_equal(img, ref), diff_stats(img, ref)
        assert st["rays"] == rays and st["spheres"] == 4096


def test_more_spheres_than_fit_in_lds(oracle):
    """6000 spheres exceed a CU's 160 KiB of LDS (two 16-B records per sphere): the records are
    then read from global memory instead of being staged."""
    scene = rt.synthetic_scene(6000, 4242)
    W, H, B = 160, 96, 3
    ref, _, rays = oracle_render(oracle, scene, W, H, B)
    for strict in (True, False):
        img, st = gpu_render(scene, W, H, B, strict=strict)
        assert np.array_equal(img, ref), diff_stats(img, ref)
        assert st["rays"] == rays and st["spheres"] == 6000


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5])
def test_kernel_variants_agree(oracle, variant):
    cfg, scene = config_inputs("C2", width=480, height=272)
    ref, _, rays = oracle_render(oracle, scene, 480, 272, 4)
    for strict in (True, False):
        img, st = gpu_render(scene, 480, 272, 4, strict=strict, variant=variant)
        assert np.array_equal(img, ref) and st["rays"] == rays


# ---- size-independent properties at BASELINE's full sizes -------------------------------------------
def test_c3_fast_equals_strict_and_is_deterministic():
    cfg, scene = config_inputs("C3")
    W, H, B = cfg["width"], cfg["height"], cfg["bounces"]
    a, sa = gpu_render(scene, W, H, B, strict=False)
    b, sb = gpu_render(scene, W, H, B, strict=True)
    c, sc = gpu_render(scene, W, H, B, strict=False)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    assert sa["rays"] == sb["rays"] == sc["rays"]
    assert np.all(a[..., 3] == 255)


def test_c3_sampled_tiles_against_oracle(oracle):
    """Every 27th 8-row tile of the full-size C3 frame, oracle vs GPU (10 of 270 tiles)."""
    cfg, scene = config_inputs("C3")
    W, H, B = cfg["width"], cfg["height"], cfg["bounces"]
    img, _ = gpu_render(scene, W, H, B, strict=False)
    ref, _, _ = oracle_render(oracle, scene, W, H, B, tile_first=13, tile_step=27)
    rows = [y for y in range(H) if (y // 8) >= 13 and ((y // 8) - 13) % 27 == 0]
    assert len(rows) == 80
    assert np.array_equal(img[rows], ref[rows])


@pytest.mark.parametrize("world", [2, 8])
def test_row_tile_partition_reassembles_the_frame(world):
    """What C4 does across 8 GPUs, emulated rank by rank on one: each rank renders its
    interleaved tiles into a compact device buffer (rt_render_to on torch's stream), the buffers
    are laid out as the all-gather would, rt_assemble_frame de-interleaves; the result must be
    byte-identical to the 1-GPU frame."""
    import torch
    cfg, scene = config_inputs("C2", width=1920, height=1076)   # 135 tiles, last one partial
    W, H, B = cfg["width"], cfg["height"], cfg["bounces"]
    full, st_full = gpu_render(scene, W, H, B, strict=False)
    msg = tiles.message_bytes(W, H, world)
    gathered = torch.zeros(world * msg, dtype=torch.uint8, device="cuda")
    frame = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    rays = 0
    last = None
    for r in range(world):
        ren = rt.RendererRaytracing(W, H, scene, maxBounces=B, rank=r, world=world).initialize()
        part = gathered[r * msg:(r + 1) * msg]
        ren.render_to(part.data_ptr(), part.numel(), stream)
        ren.wait()
        rays += ren.stats()["rays"]
        # the same tiles through the context's own buffer and rt_read_pixels
        ren.render()
        own = ren.read_pixels()
        rows = [y for y in range(H) if (y // 8) % world == r]
        assert own.shape[0] == len(rows) and np.array_equal(own, full[rows])
        if last is not None:
            last.close()
        last = ren
    last.assemble_frame(gathered.data_ptr(), frame.data_ptr(), world, stream)
    torch.cuda.synchronize()
    got = frame.cpu().numpy().reshape(H, W, 4)
    assert np.array_equal(got, full)
    assert np.array_equal(tiles.assemble_numpy(gathered.cpu().numpy(), W, H, world), full)
    assert rays == st_full["rays"]
    last.close()


# ---- the boundary's error behaviour ----------------------------------------------------------------
def test_call_order_and_capacity_errors():
    L = abi.load()
    ctx = ctypes.c_void_p()
    abi.check(L.rt_create(0, ctypes.byref(ctx)))
    try:
        assert L.rt_render(ctx) == abi.RT_ERR_STATE and b"rt_resize" in L.rt_last_error(ctx)
        assert L.rt_resize(ctx, 0, 5) == abi.RT_ERR_INVALID_ARG
        assert L.rt_resize(ctx, 65536, 65536) == abi.RT_ERR_INVALID_ARG and b"2^31" in L.rt_last_error(ctx)
        abi.check(L.rt_resize(ctx, 16, 16), ctx)
        assert L.rt_render(ctx) == abi.RT_ERR_STATE and b"rt_write_params" in L.rt_last_error(ctx)
        p = np.zeros(24, np.float32)
        abi.check(L.rt_write_params(ctx, p.ctypes.data_as(ctypes.POINTER(ctypes.c_float))), ctx)
        assert L.rt_render(ctx) == abi.RT_ERR_STATE and b"rt_write_spheres" in L.rt_last_error(ctx)
        abi.check(L.rt_write_spheres(ctx, None, 0), ctx)
        assert L.rt_render(ctx) == abi.RT_ERR_STATE and b"cube map" in L.rt_last_error(ctx)
        assert L.rt_write_cubemap_face(ctx, 6, 1, 1, p.ctypes.data) == abi.RT_ERR_INVALID_ARG
        assert L.rt_select_kernel(ctx, abi.RT_KERNEL_HEATMAP) == abi.RT_OK
        assert L.rt_select_kernel(ctx, abi.RT_KERNEL_RAYTRACER) == abi.RT_OK
        assert L.rt_select_kernel(ctx, 7) == abi.RT_ERR_INVALID_ARG
        assert L.rt_set_partition(ctx, 2, 2) == abi.RT_ERR_INVALID_ARG
        assert L.rt_set_mode(ctx, 5) == abi.RT_ERR_INVALID_ARG
        buf = np.zeros(16, np.uint8)
        assert L.rt_read_pixels(ctx, buf.ctypes.data, buf.nbytes) == abi.RT_ERR_CAPACITY
        assert L.rt_create(99, ctypes.byref(ctypes.c_void_p())) == abi.RT_ERR_NO_DEVICE
    finally:
        L.rt_destroy(ctx)


def test_renderer_reuse_across_frames_and_camera_moves(oracle):
    """The per-frame call sequence of src/app.ts:117-128: scene.update, camera.move, render --
    params are re-uploaded every frame, spheres only once (RR:194-195)."""
    scene = rt.synthetic_scene(32, 11)
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    r = rt.RendererRaytracing(200, 120, scene, maxBounces=4).initialize(sky)
    try:
        for step in range(3):
            scene.update(0.016)
            scene.camera.move(0.25, -0.1)
            if step == 2:
                scene.camera.spin(7.0, -3.0)
            r.render()
            img = r.read_pixels()
            ref, _, rays = oracle.render(scene.pack_params(4), scene.pack_spheres(), sky.faces, 200, 120)
            assert np.array_equal(img, ref) and r.stats()["rays"] == rays
            assert r.render_time_ms is not None and r.stats()["frames"] == step + 1
    finally:
        r.close()


def test_resize_and_repartition_between_frames(oracle):
    """rt_resize / rt_set_partition are re-callable (the reference sizes its colour buffer once,
    RR:102-109; a host may not): buffers and the path queue are regrown, frames stay exact."""
    scene = rt.synthetic_scene(200, 31)
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    L = abi.load()
    r = rt.RendererRaytracing(96, 64, scene, maxBounces=3).initialize(sky)
    try:
        for (w, h, rank, world) in [(96, 64, 0, 1), (320, 200, 0, 1), (64, 40, 1, 2), (64, 40, 0, 1), (400, 300, 2, 3)]:
            abi.check(L.rt_set_partition(r._ctx, rank, world), r._ctx)
            abi.check(L.rt_resize(r._ctx, w, h), r._ctx)
            r.width, r.height, r.rank, r.world = w, h, rank, world
            r.render()
            img = r.read_pixels()
            ref, _, _ = oracle.render(scene.pack_params(3), scene.pack_spheres(), sky.faces, w, h)
            rows = [y for y in range(h) if (y // 8) % world == rank]
            assert np.array_equal(img, ref[rows]), (w, h, rank, world)
    finally:
        r.close()


def test_two_contexts_side_by_side(oracle):
    a_scene, b_scene = rt.synthetic_scene(50, 1), rt.synthetic_scene(300, 2)
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    ra = rt.RendererRaytracing(128, 72, a_scene, maxBounces=2).initialize(sky)
    rb = rt.RendererRaytracing(160, 96, b_scene, maxBounces=5).initialize(sky)
    try:
        ra.recalculateScene(); rb.recalculateScene()
        ra.enqueue(); rb.enqueue(); ra.enqueue()
        ra.wait(); rb.wait()
        ia, ib = ra.read_pixels(), rb.read_pixels()
        assert np.array_equal(ia, oracle.render(a_scene.pack_params(2), a_scene.pack_spheres(), sky.faces, 128, 72)[0])
        assert np.array_equal(ib, oracle.render(b_scene.pack_params(5), b_scene.pack_spheres(), sky.faces, 160, 96)[0])
        assert ra.stats()["frames"] == 2 and rb.stats()["frames"] == 1
    finally:
        ra.close(); rb.close()


@pytest.mark.parametrize("n", [40, 300])
def test_nan_and_inf_records_in_the_middle_of_the_scene(oracle, n):
    """a NaN / inf sphere record that is NOT the last one used to drop out of the scene
    bound, so fast mode kept its filter and hierarchy outside their proven range.  Now any such record
    sends the frame to the literal kernel: fast == strict == oracle."""
    scene = rt.synthetic_scene(n, 600 + n)
    s = scene.pack_spheres().copy()
    s[1, 0] = 3.0e6                  # far sphere first ...
    s[n // 3, 2] = np.nan            # ... a NaN record in the middle ...
    s[n // 2, 7] = np.inf            # ... an infinite radius, then ordinary spheres
    s[n // 2 + 1, 1] = -np.inf
    p = scene.pack_params(4)
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    W, H = 160, 96
    ref, _, rays = oracle.render(p, s, sky.faces, W, H)
    L = abi.load()
    fp = ctypes.POINTER(ctypes.c_float)
    for strict in (True, False):
        ctx = ctypes.c_void_p()
        abi.check(L.rt_create(0, ctypes.byref(ctx)))
        try:
            abi.check(L.rt_resize(ctx, W, H), ctx)
            abi.check(L.rt_set_mode(ctx, 1 if strict else 0), ctx)
            abi.check(L.rt_write_params(ctx, p.ctypes.data_as(fp)), ctx)
            abi.check(L.rt_write_spheres(ctx, s.ctypes.data_as(fp), n), ctx)
            for f in range(6):
                face = np.ascontiguousarray(sky.faces[f])
                abi.check(L.rt_write_cubemap_face(ctx, f, face.shape[1], face.shape[0], face.ctypes.data), ctx)
            abi.check(L.rt_render(ctx), ctx)
            img = np.zeros((H, W, 4), np.uint8)
            abi.check(L.rt_read_pixels(ctx, img.ctypes.data, img.nbytes), ctx)
            st = abi.RtStats()
            abi.check(L.rt_get_stats(ctx, ctypes.byref(st)), ctx)
        finally:
            L.rt_destroy(ctx)
        assert np.array_equal(img, ref), (strict, diff_stats(img, ref))
        assert st.rays == rays


# ---- C4: the headline frame row-tiled over 2 / 4 / 8 ranks ------------------------------------------
@pytest.mark.parametrize("world", [2, 4, 8])
def test_c4_emulated_ranks_reassemble_the_golden_c3_frame(world):
    """BASELINE config C4 = the C3 frame (3840x2160, 1024 spheres, 8 bounces) split into 8-row tiles,
    tile t rendered by rank t % world.  One GPU plays every rank in turn (rt_set_partition +
    rt_render_to into that rank's slot of the all-gather layout), rt_assemble_frame de-interleaves:
    the frame must hash to the ORACLE's C3 frame (tests/golden/frames.json) and the ranks' ray
    counts must add up to the oracle's."""
    import torch
    fr = json.load(open(os.path.join(G, "frames.json")))["C3"]
    cfg, scene = config_inputs("C3")
    W, H, B = cfg["width"], cfg["height"], cfg["bounces"]
    assert (W, H, len(scene.spheres), B) == (3840, 2160, 1024, 8)
    msg = tiles.message_bytes(W, H, world)
    gathered = torch.zeros(world * msg, dtype=torch.uint8, device="cuda")
    frame = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ren = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize()
    L = abi.load()
    rays = 0
    try:
        for r in range(world):
            abi.check(L.rt_set_partition(ren._ctx, r, world), ren._ctx)
            ren.rank, ren.world = r, world
            part = gathered[r * msg:(r + 1) * msg]
            ren.render_to(part.data_ptr(), part.numel(), stream)
            ren.wait()
            st = ren.stats()
            assert st["local_tiles"] == tiles.tiles_of_rank(H, r, world)
            rays += st["rays"]
        ren.assemble_frame(gathered.data_ptr(), frame.data_ptr(), world, stream)
        torch.cuda.synchronize()
    finally:
        ren.close()
    got = frame.cpu().numpy()
    assert hashlib.sha256(got.tobytes()).hexdigest() == fr["sha256"]
    assert rays == fr["rays"]


# ---- C5 at full size: 7680x4320, 4096 spheres, 16 bounces, 6 x 512^2 textured cube ----------------------
def _c5():
    sys_path_golden = os.path.join(G, "make_golden.py")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", sys_path_golden)
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = json.load(open(os.path.join(G, "c5_tiles.json")))
    cfg, scene = config_inputs("C5")
    sky = mg.c5_sky()
    sky_sha = hashlib.sha256(b"".join(np.ascontiguousarray(f).tobytes() for f in sky.faces)).hexdigest()
    assert sky_sha == g["sky_sha256"], "the procedural C5 sky is not the one the golden tiles were rendered under"
    assert hashlib.sha256(scene.pack_spheres().tobytes()).hexdigest() == g["scene_sha256"]
    assert (cfg["width"], cfg["height"], cfg["spheres"], cfg["bounces"]) == (7680, 4320, 4096, 16)
    return g, cfg, scene, sky


def test_c5_full_size_against_golden_tiles_and_strict(oracle):
    """BASELINE config C5 at its full size.  The oracle's rows for 10 of the 540 tiles and 256 sparse
    pixels (64 of them next to cube-face edges) are committed (tests/golden/c5_tiles.json, generated
    by make_golden.py --c5); the fast frame must reproduce them, equal the strict (literal) frame
    bit for bit, and -- if the oracle's full-frame hash is on file -- hash to it."""
    g, cfg, scene, sky = _c5()
    W, H, B = cfg["width"], cfg["height"], cfg["bounces"]
    fast, st_fast = gpu_render(scene, W, H, B, strict=False, skybox=sky)
    for t in g["tiles"]:
        rows = fast[8 * t["tile"]:8 * t["tile"] + 8]
        if hashlib.sha256(rows.tobytes()).hexdigest() != t["sha256"]:
            ref, _, _ = oracle_render(oracle, scene, W, H, B, skybox=sky, tile_first=t["tile"], tile_step=(H + 7) // 8)
            bad = np.argwhere((rows != ref[8 * t["tile"]:8 * t["tile"] + 8]).any(-1))
            raise AssertionError("C5 tile %d differs from the oracle at %d pixels, first (row, x) %s"
                                 % (t["tile"], len(bad), bad[:4].tolist()))
    for px in g["pixels"]:
        assert list(fast[px["y"], px["x"]]) == px["rgba8"], px
    strict, st_strict = gpu_render(scene, W, H, B, strict=True, skybox=sky)
    assert np.array_equal(fast, strict)
    assert st_fast["rays"] == st_strict["rays"]
    assert np.all(fast[..., 3] == 255)
    if "frame_sha256" in g:
        assert hashlib.sha256(fast.tobytes()).hexdigest() == g["frame_sha256"]
        assert st_fast["rays"] == g["frame_rays"]


def test_c5_full_size_under_the_reference_skybox(oracle):
    """BASELINE.md names the sky of C5: the reference's src/assets/images/daylight-skybox.png, cut into six 512x512
    faces as cubemap-material.ts:35-58 does.  Those faces are a committed fixture (tests/golden/ref_sky.png, the sky
    the oracle is pinned under by the reference's screenshot, tests/test_ref_pin.py); the oracle's rows for ten tiles
    and 256 sparse pixels of the full-size C5 frame under them are in tests/golden/c5_ref_sky.json."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = json.load(open(os.path.join(G, "c5_ref_sky.json")))
    cfg, scene = config_inputs("C5")
    sky = mg.ref_sky()
    assert hashlib.sha256(b"".join(np.ascontiguousarray(f).tobytes() for f in sky.faces)).hexdigest() == g["sky_sha256"]
    assert hashlib.sha256(scene.pack_spheres().tobytes()).hexdigest() == g["scene_sha256"]
    W, H, B = cfg["width"], cfg["height"], cfg["bounces"]
    fast, st = gpu_render(scene, W, H, B, strict=False, skybox=sky)
    for t in g["tiles"]:
        rows = fast[8 * t["tile"]:8 * t["tile"] + 8]
        if hashlib.sha256(rows.tobytes()).hexdigest() != t["sha256"]:
            ref, _, _ = oracle_render(oracle, scene, W, H, B, skybox=sky, tile_first=t["tile"], tile_step=(H + 7) // 8)
            bad = np.argwhere((rows != ref[8 * t["tile"]:8 * t["tile"] + 8]).any(-1))
            raise AssertionError("C5 (reference sky) tile %d differs from the oracle at %d pixels, first (row, x) %s"
                                 % (t["tile"], len(bad), bad[:4].tolist()))
    for px in g["pixels"]:
        assert list(fast[px["y"], px["x"]]) == px["rgba8"], px
    assert np.all(fast[..., 3] == 255) and st["kernel_id"] == 6            # the 16-wave hierarchy form


def test_c5_eight_emulated_ranks_reassemble_the_one_rank_frame():
    """C5 is specified on 8 GPUs: 540 tiles -> 68 / 67 per rank.  One GPU plays the 8 ranks in turn;
    the de-interleaved frame must equal the frame one rank renders alone, ray counts must add up."""
    import torch
    g, cfg, scene, sky = _c5()
    W, H, B = cfg["width"], cfg["height"], cfg["bounces"]
    world = 8
    full, st_full = gpu_render(scene, W, H, B, strict=False, skybox=sky)
    msg = tiles.message_bytes(W, H, world)
    gathered = torch.zeros(world * msg, dtype=torch.uint8, device="cuda")
    frame = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ren = rt.RendererRaytracing(W, H, scene, maxBounces=B).initialize(sky)
    L = abi.load()
    rays, counts = 0, []
    try:
        for r in range(world):
            abi.check(L.rt_set_partition(ren._ctx, r, world), ren._ctx)
            ren.rank, ren.world = r, world
            part = gathered[r * msg:(r + 1) * msg]
            ren.render_to(part.data_ptr(), part.numel(), stream)
            ren.wait()
            st = ren.stats()
            counts.append(st["local_tiles"])
            rays += st["rays"]
        ren.assemble_frame(gathered.data_ptr(), frame.data_ptr(), world, stream)
        torch.cuda.synchronize()
    finally:
        ren.close()
    assert counts == [68, 68, 68, 68, 67, 67, 67, 67]
    assert np.array_equal(frame.cpu().numpy().reshape(H, W, 4), full)
    assert rays == st_full["rays"]
