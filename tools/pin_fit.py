#!/usr/bin/env python3
"""How the scene state behind the reference's screenshot (info/sample_settings.png) was found -- the numbers in
tests/golden/make_ref_scene.py:STATE.  Build container only: reads /root/reference, renders with the CPU oracle.

Eleven unknowns: camera position (3) and angles (2) -- the user had walked and turned --, mousey's spin angle, and
what the GUI shows only rounded: light x / z, minIntensity, mousey x / z.  Objective: mean over the canvas of
sum over channels of |oracle frame - screenshot|, the pixels that depend on the missing mousey texture (found by
rendering with a white and a black texture) at 0.15 weight, the DOM overlay excluded.

    stage 1  400 random states inside generous bounds, frames at quarter size, both images blurred (sigma 3):
             the blur widens the basin; the six best go through Nelder-Mead at sigma 3, then sigma 1
    stage 2  mousey's angle scanned in 3-degree steps (its silhouette is nearly symmetric front / back, which
             stage 1's white-texture frames cannot tell apart; its shadow can)
    stage 3  light x / z (the cat's shadow), mousey yaw / x / z, camera, minIntensity in turn at half size,
             then all eleven together at full size

    python tools/pin_fit.py            stage 3 only, from the committed STATE (a few minutes; prints the polished state)
    python tools/pin_fit.py --search   all stages from scratch (about half an hour on 8 cores)

The result is reproducible to the digits that matter: different stage-1 seeds end within 0.005 units / 0.05 degrees of
each other, and the agreement figures of ref_pin.json do not move."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import compute_raytracer_amd as rt                      # noqa: E402
from compute_raytracer_amd.camera import Camera        # noqa: E402
from compute_raytracer_amd.procedural import tri_buffers   # noqa: E402
from oracle import rt_oracle_py as orc                  # noqa: E402
import make_ref_scene as M                              # noqa: E402

NAMES = ["cam x", "cam y", "cam z", "phi", "theta", "mousey yaw", "mousey x", "mousey z", "light x", "light z", "minIntensity"]


def load():
    from PIL import Image
    shot = np.array(Image.open(M.REF + "/info/sample_settings.png").convert("RGB"))
    c = M.CANVAS
    canvas = shot[c["y"]:c["y"] + c["height"], c["x"]:c["x"] + c["width"]].copy()
    sky = rt.CubemapMaterial.from_png(M.REF + "/src/assets/images/daylight-skybox.png")
    meshes = rt.SceneRaytracing().createReferenceScene(M.REF + "/src/assets/models").meshes      # OBJ -> soup -> tree, once
    return canvas, sky, meshes


class Fit:
    def __init__(self):
        self.canvas, self.sky, self.meshes = load()
        self.H, self.W = self.canvas.shape[:2]
        self.black = np.zeros((1, 1, 4), np.uint8); self.black[..., 3] = 255

    def scene(self, P):
        s = rt.SceneRaytracing().createScene([])
        s.createTriangleScene(self.meshes, [dict(meshIndex=0, position=[-2.5, 0, 0], eulers=[180, 0, 0]),
                                            dict(meshIndex=1, position=[P[6], 0, P[7]], eulers=[180, P[5], 0]),
                                            dict(meshIndex=2, position=[0, 0, 0], eulers=[0, 0, 0])])
        s.camera = Camera([P[0], P[1], P[2]], P[4], P[3])
        s.light.position = [P[8], 5.0, P[9]]
        s.light.minIntensity = P[10]
        return s

    def render(self, P, S, tex=None):
        s = self.scene(P)
        b = tri_buffers(s, rt.Material.white() if tex is None else rt.Material(tex))
        return orc.render_tri(s.pack_params(M.BOUNCES), b, self.sky.faces, self.W // S, self.H // S)[0][..., :3]

    def cost_fn(self, idx, S, base, sigma=0.0):
        from scipy.ndimage import gaussian_filter
        blur = (lambda a: np.stack([gaussian_filter(a[..., c], sigma) for c in range(3)], -1)) if sigma > 0 else (lambda a: a)
        tgt = blur(self.canvas[::S, ::S][:self.H // S, :self.W // S].astype(np.float32))
        seen = np.ones(tgt.shape[:2], bool)
        seen[:M.OVERLAY["height"] // S + 1, :M.OVERLAY["width"] // S + 1] = False
        free = (self.render(base, S) == self.render(base, S, self.black)).all(-1) & seen

        def cost(q):
            P = base.copy(); P[idx] = q
            d = np.abs(blur(self.render(P, S).astype(np.float32)) - tgt).sum(-1)
            return float(d[free].mean() + 0.15 * d[seen & ~free].mean())
        return cost

    def descend(self, P, idx, steps, S, iters, sigma=0.0):
        from scipy.optimize import minimize
        idx = np.array(idx)
        cost = self.cost_fn(idx, S, P, sigma)
        q0 = P[idx]
        simplex = np.vstack([q0] + [q0 + np.eye(len(idx))[k] * steps[k] for k in range(len(idx))])
        r = minimize(cost, q0, method="Nelder-Mead", options=dict(xatol=1e-4, fatol=1e-4, maxiter=iters, initial_simplex=simplex))
        P = P.copy(); P[idx] = r.x
        return P, r.fun


def main():
    t0 = time.time()
    f = Fit()
    st = M.STATE
    P = np.array(st["camera_position"] + [st["camera_phi"], st["camera_theta"], st["mousey_yaw"], st["mousey_x"], st["mousey_z"],
                                           st["light"][0], st["light"][2], st["minIntensity"]])
    if "--search" in sys.argv:
        rng = np.random.default_rng(1)
        cands = []
        for _ in range(400):
            p = np.array([rng.uniform(-9, -3), rng.uniform(1.8, 4.5), rng.uniform(-1.5, 5), rng.uniform(-50, 30), rng.uniform(98, 118),
                          rng.uniform(0, 360), 0.0, 0.0, -2.0, 2.0, 0.3])
            cands.append((f.cost_fn(np.arange(0), 4, p, 3.0)(np.zeros(0)), p))
        cands.sort(key=lambda t: t[0])
        best = None
        for c, p in cands[:6]:
            p, _ = f.descend(p, range(6), [0.5, 0.3, 0.5, 5, 2, 20], 4, 400, 3.0)
            p, v = f.descend(p, range(6), [0.1, 0.1, 0.1, 1, 0.5, 5], 4, 400, 1.0)
            print("stage 1: %.2f -> %.3f" % (c, v), np.round(p[:6], 3), "%.0f s" % (time.time() - t0), flush=True)
            if best is None or v < best[0]:
                best = (v, p)
        P = best[1]
        scan = sorted((f.cost_fn(np.array([5]), 2, P)(np.array([float(y)])), y) for y in range(0, 360, 3))
        print("stage 2: mousey yaw", scan[:4], flush=True)
        P[5] = scan[0][1]
    for idx, steps, S, iters in [([8, 9], [0.2, 0.2], 2, 80), ([5, 6, 7], [3, 0.1, 0.2], 2, 120), ([8, 9], [0.05, 0.05], 2, 60),
                                 ([0, 1, 2, 3, 4], [0.01, 0.01, 0.01, 0.1, 0.05], 2, 200), ([5, 6, 7, 8, 9], [1, 0.03, 0.05, 0.03, 0.03], 2, 200),
                                 ([10], [0.0005], 1, 25),
                                 (list(range(11)), [0.004, 0.004, 0.004, 0.04, 0.02, 0.5, 0.01, 0.02, 0.01, 0.01, 0.0002], 1, 500)]:
        P, v = f.descend(P, idx, steps, S, iters)
        print("stage 3:", [NAMES[i] for i in idx], "1/%d size: %.4f" % (S, v), "%.0f s" % (time.time() - t0), flush=True)
    for n, v in zip(NAMES, P):
        print("%-13s %.9g" % (n, v))


if __name__ == "__main__":
    main()
