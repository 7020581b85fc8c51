"""Wave-occupancy statistics of the pixel-per-lane mapping, from the oracle's per-pixel ray
counts (development tool): what fraction of lane-slots do useful ray-sphere work when an 8x8
tile is one wave64 and every wave runs until its longest path ends."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import compute_raytracer_amd as rt
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
from oracle import rt_oracle_py as orc

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = rt.BASELINE_CONFIGS[name]
W, H, B = cfg["width"] // scale, cfg["height"] // scale, cfg["bounces"]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
img, cnt, rays = orc.render_ray_counts(scene.pack_params(B), scene.pack_spheres(), sky.faces, W, H)
cnt = cnt.astype(np.int64)
# per pixel: traces (primary/reflection) T and shadow rays S
T = np.where(cnt % 2 == 1, (cnt + 1) // 2, cnt // 2)
S = cnt - T
print("rays", rays, "per px %.2f" % (rays / (W * H)), " primary", W * H, " reflection", int(T.sum()) - W * H, " shadow", int(S.sum()))
Hp, Wp = (H // 8) * 8, (W // 8) * 8
for (th, tw) in [(8, 8), (4, 16), (2, 32), (1, 64)]:
    Hq, Wq = (H // th) * th, (W // tw) * tw
    Tt = T[:Hq, :Wq].reshape(Hq // th, th, Wq // tw, tw).transpose(0, 2, 1, 3).reshape(-1, th * tw)
    St = S[:Hq, :Wq].reshape(Hq // th, th, Wq // tw, tw).transpose(0, 2, 1, 3).reshape(-1, th * tw)
    # wave executes max over lanes of T trace loops and of S shadow loops (each bounce in lockstep)
    useful_full = (Tt - 1).clip(min=0).sum(); slots_full = ((Tt.max(axis=1) - 1).clip(min=0) * 64).sum()
    useful_sh = St.sum(); slots_sh = (St.max(axis=1) * 64).sum()
    # cost weights: primary 6, reflection 12, shadow 6 VALU/test
    useful = 6 * Tt.shape[0] * 64 + 12 * useful_full + 6 * useful_sh
    slots = 6 * Tt.shape[0] * 64 + 12 * slots_full + 6 * slots_sh
    print("tile %dx%d: reflection lanes %.3f  shadow lanes %.3f  weighted %.3f" % (tw, th, useful_full / max(slots_full, 1), useful_sh / max(slots_sh, 1), useful / slots))
