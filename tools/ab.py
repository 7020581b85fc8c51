"""A/B timing of kernel variants and arithmetic modes on one GPU (development tool).
usage: python tools/ab.py [C2|C3|C5 ...] [--variants 1,2,3] [--frames 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import compute_raytracer_amd as rt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["C2", "C3"])
    ap.add_argument("--variants", default="1,2,3,4")
    ap.add_argument("--modes", default="fast,strict")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--spheres", type=int, default=0, help="override the sphere count of the config")
    a = ap.parse_args()
    for name in a.configs:
        cfg = dict(rt.BASELINE_CONFIGS[name])
        if a.spheres:
            cfg["spheres"] = a.spheres
        scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
        for mode in a.modes.split(","):
            for v in [int(x) for x in a.variants.split(",")]:
                r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"]).initialize()
                r.set_mode(mode == "strict")
                r.set_variant(v)
                ms = []
                for _ in range(a.frames + 1):
                    r.render()
                    ms.append(r.stats()["kernel_ms"])
                st = r.stats()
                best = min(ms[1:])
                img = r.read_pixels()
                r.close()
                print("%s N=%d B=%d mode=%-6s variant=%d  kernel %.3f ms (min of %d; all %s)  rays %d  %.1f Mrays/s  %.1f fps  csum %d"
                      % (name, cfg["spheres"], cfg["bounces"], mode, v, best, a.frames, ["%.2f" % m for m in ms],
                         st["rays"], st["rays"] / best / 1e3, 1e3 / best, int(img.astype(np.uint64).sum())), flush=True)


if __name__ == "__main__":
    main()
