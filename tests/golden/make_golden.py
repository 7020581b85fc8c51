#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the CPU oracle (oracle/rt_oracle.c).

The reference repository holds no tests, golden images or known-answer vectors for this path
 and cannot be executed offline, so these vectors pin the ORACLE (and through
it the HIP kernels) against regressions; they are not outputs of the reference (the one output the reference holds --
its screenshot -- pins the oracle itself: make_ref_scene.py, tests/test_ref_pin.py).

    python tests/golden/make_golden.py            # C1, C2 (seconds)
    python tests/golden/make_golden.py --c3       # also the full C3 frame hash (about a minute of CPU)
    python tests/golden/make_golden.py --c5-full  # adds the sha256 / ray count of the WHOLE C5 frame (ten minutes)
    python tests/golden/make_golden.py --c5       # only c5_tiles.json: 10 of the 540 8-row tiles of the 7680x4320 /
                                                  # 4096-sphere / 16-bounce frame under a 6 x 512^2 textured cube

Files written:
    c1_frame.png          full C1 frame (256x256 RGBA8, lossless)
    frames.json           per config: sha256 of the RGBA8 frame, total rays
    sparse_<cfg>.json     per config: 256 pixels {x, y, rgb as f32 bit patterns, rgba8, rays}
    scene_<cfg>.json      first/last sphere records + sha256 of the packed scene (generator KAT)
    c5_tiles.json         C5: sha256 of the six sky faces, per sampled tile sha256 of its 8 rows + ray count,
                          256 sparse pixels (64 of them chosen within one texel of a cube-face edge)
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import compute_raytracer_amd as rt  # noqa: E402
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA  # noqa: E402
from oracle import rt_oracle_py as orc  # noqa: E402


def sparse_pixels(W, H, n, seed):
    rng = np.random.default_rng(seed)
    pts = {(W // 2, H // 2), (0, 0), (W - 1, 0), (0, H - 1), (W - 1, H - 1)}
    while len(pts) < n:
        pts.add((int(rng.integers(0, W)), int(rng.integers(0, H))))
    return sorted(pts, key=lambda p: (p[1], p[0]))


C5_TILE_FIRST, C5_TILE_STEP = 27, 54          # tiles 27, 81, ... 513: 10 of 540


def c5_sky():
    """The deterministic 6 x 512 x 512 textured cube C5 is tested and benchmarked with (the reference's
    daylight-skybox.png does not travel with this repository; cubemap-material.ts:40-47 face order)."""
    return rt.CubemapMaterial.synthetic_daylight()


def c5_edge_pixels(p, W, H, n_face, count, seed):
    """Pixels whose PRIMARY ray lands within one texel of a cube-face edge (RK:78-86 in float64 is
    enough to choose them)."""
    xs, ys = np.meshgrid(np.arange(0, W, 3), np.arange(0, H, 3))
    hc = (xs - W / 2) / W * 2
    vc = (H / 2 - ys) / W * 2
    d = p[4:7][None, None, :] + hc[..., None] * p[8:11] + vc[..., None] * p[12:15]
    a = np.abs(d)
    ma = a.max(-1)
    second = np.sort(a, -1)[..., 1]
    near = (second / ma) > 1.0 - 2.0 / n_face
    cand = np.stack([xs[near], ys[near]], 1)
    rng = np.random.default_rng(seed)
    pick = cand[rng.choice(len(cand), size=min(count, len(cand)), replace=False)]
    return [(int(x), int(y)) for x, y in pick]


def make_c5():
    cfg = rt.BASELINE_CONFIGS["C5"]
    W, H, N, B = cfg["width"], cfg["height"], cfg["spheres"], cfg["bounces"]
    scene = rt.synthetic_scene(N, cfg["seed"])
    p, s = scene.pack_params(B), scene.pack_spheres()
    sky = c5_sky()
    out = {"config": "C5", "width": W, "height": H, "spheres": N, "bounces": B, "seed": cfg["seed"],
           "sky_sha256": hashlib.sha256(b"".join(np.ascontiguousarray(f).tobytes() for f in sky.faces)).hexdigest(),
           "scene_sha256": hashlib.sha256(s.tobytes()).hexdigest(),
           "tile_first": C5_TILE_FIRST, "tile_step": C5_TILE_STEP, "tiles": []}
    ntiles = (H + 7) // 8
    for t in range(C5_TILE_FIRST, ntiles, C5_TILE_STEP):
        img, _, rays = orc.render(p, s, sky.faces, W, H, tile_first=t, tile_step=ntiles)
        rows = img[8 * t:8 * t + 8]
        out["tiles"].append({"tile": t, "sha256": hashlib.sha256(rows.tobytes()).hexdigest(), "rays": rays})
        print("C5 tile", t, out["tiles"][-1], flush=True)
    pts = sorted(set(c5_edge_pixels(p, W, H, sky.faces[0].shape[0], 64, 5) + sparse_pixels(W, H, 192, 1234 + N)),
                 key=lambda q: (q[1], q[0]))
    for (x, y) in pts:
        rgb, rays = orc.pixel(p, s, sky.faces, W, H, x, y)
        out.setdefault("pixels", []).append({"x": x, "y": y, "rgba8": [orc.unorm8(rgb[0]), orc.unorm8(rgb[1]), orc.unorm8(rgb[2]), 255],
                                             "rays": rays})
    json.dump(out, open(os.path.join(HERE, "c5_tiles.json"), "w"))


def ref_sky():
    """The six faces of the reference's daylight-skybox.png as committed by make_ref_scene.py (tests/golden/ref_sky.png)."""
    from PIL import Image
    strip = np.array(Image.open(os.path.join(HERE, "ref_sky.png")).convert("RGBA"), dtype=np.uint8)
    m = rt.CubemapMaterial()
    m.faces = [np.ascontiguousarray(strip[:, k * strip.shape[0]:(k + 1) * strip.shape[0]]) for k in range(6)]
    return m


def make_c5_ref_sky():
    """C5 under the sky BASELINE.md names for it -- the reference's src/assets/images/daylight-skybox.png, cut as
    cubemap-material.ts:35-58 cuts it -- instead of the procedural stand-in: the same ten tiles and sparse pixels."""
    cfg = rt.BASELINE_CONFIGS["C5"]
    W, H, N, B = cfg["width"], cfg["height"], cfg["spheres"], cfg["bounces"]
    scene = rt.synthetic_scene(N, cfg["seed"])
    p, s = scene.pack_params(B), scene.pack_spheres()
    sky = ref_sky()
    out = {"config": "C5", "sky": "tests/golden/ref_sky.png (daylight-skybox.png, six 512x512 faces)", "width": W, "height": H,
           "spheres": N, "bounces": B, "seed": cfg["seed"],
           "sky_sha256": hashlib.sha256(b"".join(np.ascontiguousarray(f).tobytes() for f in sky.faces)).hexdigest(),
           "scene_sha256": hashlib.sha256(s.tobytes()).hexdigest(), "tiles": [], "pixels": []}
    ntiles = (H + 7) // 8
    for t in range(C5_TILE_FIRST, ntiles, C5_TILE_STEP):
        img, _, rays = orc.render(p, s, sky.faces, W, H, tile_first=t, tile_step=ntiles)
        out["tiles"].append({"tile": t, "sha256": hashlib.sha256(img[8 * t:8 * t + 8].tobytes()).hexdigest(), "rays": rays})
        print("C5 (reference sky) tile", t, out["tiles"][-1], flush=True)
    pts = sorted(set(c5_edge_pixels(p, W, H, sky.faces[0].shape[0], 64, 5) + sparse_pixels(W, H, 192, 1234 + N)), key=lambda q: (q[1], q[0]))
    for (x, y) in pts:
        rgb, rays = orc.pixel(p, s, sky.faces, W, H, x, y)
        out["pixels"].append({"x": x, "y": y, "rgba8": [orc.unorm8(rgb[0]), orc.unorm8(rgb[1]), orc.unorm8(rgb[2]), 255], "rays": rays})
    json.dump(out, open(os.path.join(HERE, "c5_ref_sky.json"), "w"))


def make_c5_full():
    """The whole 7680x4320 C5 frame on the CPU (about ten minutes on 8 cores): adds its sha256 and ray
    count to c5_tiles.json and cross-checks the sampled tiles against the full frame."""
    path = os.path.join(HERE, "c5_tiles.json")
    g = json.load(open(path))
    cfg = rt.BASELINE_CONFIGS["C5"]
    W, H, N, B = cfg["width"], cfg["height"], cfg["spheres"], cfg["bounces"]
    scene = rt.synthetic_scene(N, cfg["seed"])
    p, s = scene.pack_params(B), scene.pack_spheres()
    sky = c5_sky()
    img, _, rays = orc.render(p, s, sky.faces, W, H)
    for t in g["tiles"]:
        assert hashlib.sha256(img[8 * t["tile"]:8 * t["tile"] + 8].tobytes()).hexdigest() == t["sha256"]
    g["frame_sha256"] = hashlib.sha256(img.tobytes()).hexdigest()
    g["frame_rays"] = rays
    json.dump(g, open(path, "w"))
    print("C5 full frame", g["frame_sha256"], rays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c3", action="store_true")
    ap.add_argument("--c5", action="store_true")
    ap.add_argument("--c5-full", action="store_true")
    ap.add_argument("--c5-ref-sky", action="store_true")
    a = ap.parse_args()
    if a.c5_ref_sky:
        make_c5_ref_sky()
        return
    if a.c5:
        make_c5()
        return
    if a.c5_full:
        make_c5_full()
        return
    sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    frames_path = os.path.join(HERE, "frames.json")
    frames = json.load(open(frames_path)) if os.path.exists(frames_path) else {}
    for name in ["C1", "C2", "C3"]:
        cfg = rt.BASELINE_CONFIGS[name]
        W, H, N, B = cfg["width"], cfg["height"], cfg["spheres"], cfg["bounces"]
        scene = rt.synthetic_scene(N, cfg["seed"])
        p, s = scene.pack_params(B), scene.pack_spheres()
        json.dump({
            "config": name, "n": N, "seed": cfg["seed"],
            "params_bits": [int(v) for v in p.view(np.uint32)],
            "first": [int(v) for v in s[:3].reshape(-1).view(np.uint32)],
            "last": [int(v) for v in s[-1].view(np.uint32)],
            "sha256": hashlib.sha256(s.tobytes()).hexdigest(),
        }, open(os.path.join(HERE, "scene_%s.json" % name), "w"), indent=1)
        pts = sparse_pixels(W, H, 256, 1234 + N)
        rows = []
        for (x, y) in pts:
            rgb, rays = orc.pixel(p, s, sky.faces, W, H, x, y)
            rows.append({"x": x, "y": y, "rgb_bits": [int(v) for v in rgb.view(np.uint32)],
                         "rgba8": [orc.unorm8(rgb[0]), orc.unorm8(rgb[1]), orc.unorm8(rgb[2]), 255], "rays": rays})
        json.dump({"config": name, "pixels": rows}, open(os.path.join(HERE, "sparse_%s.json" % name), "w"))
        if name == "C3" and not a.c3:
            continue
        img, _, rays = orc.render(p, s, sky.faces, W, H)
        frames[name] = {"sha256": hashlib.sha256(img.tobytes()).hexdigest(), "rays": rays,
                        "width": W, "height": H, "spheres": N, "bounces": B}
        if name == "C1":
            from PIL import Image
            Image.fromarray(img, "RGBA").save(os.path.join(HERE, "c1_frame.png"), optimize=True)
        print(name, frames[name])
    json.dump(frames, open(frames_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
