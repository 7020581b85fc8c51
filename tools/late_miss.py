"""How many pixels of a BASELINE sphere configuration end their path on a miss AFTER bounce 0 -- the pixels whose end-of-path
record needs its second half (direction + bounce count) for sky_resolve (rt_bvh.hip).  From the oracle's per-pixel ray
counts (a hit costs two rays, a miss one): odd and > 1 = late miss.  usage: python tools/late_miss.py C5 [width height]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import compute_raytracer_amd as rt
from oracle import rt_oracle_py as orc
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
cfg = rt.BASELINE_CONFIGS[name]
W = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["width"] // 8
H = int(sys.argv[3]) if len(sys.argv) > 3 else cfg["height"] // 8
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)      # which pixels miss does not depend on the sky's colours
_, c, rays = orc.render_ray_counts(scene.pack_params(cfg["bounces"]), scene.pack_spheres(), sky.faces, W, H)
c = c.reshape(-1).astype(int)
print("%s scene at %dx%d: %.2f rays per pixel; primary miss %.3f, late miss %.3f, paths that use every bounce %.3f"
      % (name, W, H, c.mean(), (c == 1).mean(), ((c % 2 == 1) & (c > 1)).mean(), (c % 2 == 0).mean()))
print("end-of-path records: 16 B per pixel + 16 B per late miss = %.1f B per pixel at least (one 32-byte sector per pixel today)" % (16 + 16 * ((c % 2 == 1) & (c > 1)).mean()))
