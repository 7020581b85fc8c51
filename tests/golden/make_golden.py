#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the CPU oracle (oracle/rt_oracle.c).

The reference repository holds no tests, golden images or known-answer vectors for this path
 and cannot be executed offline, so these vectors pin the ORACLE (and through
it the HIP kernels) against regressions; they are not outputs of the reference.  PARITY UNPINNED.

    python tests/golden/make_golden.py            # C1, C2 (seconds)
    python tests/golden/make_golden.py --c3       # also the full C3 frame hash (about a minute of CPU)

Files written:
    c1_frame.png          full C1 frame (256x256 RGBA8, lossless)
    frames.json           per config: sha256 of the RGBA8 frame, total rays
    sparse_<cfg>.json     per config: 256 pixels {x, y, rgb as f32 bit patterns, rgba8, rays}
    scene_<cfg>.json      first/last sphere records + sha256 of the packed scene (generator KAT)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c3", action="store_true")
    a = ap.parse_args()
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
