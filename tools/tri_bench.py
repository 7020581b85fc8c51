"""Timing of the triangle/BVH kernels on a procedural scene of the reference scene's size
(~12.6k triangles, 3 BLAS + floor), development tool.
usage: python tools/tri_bench.py [--width 1344 --height 846 --bounces 4 --frames 5]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import compute_raytracer_amd as rt
from helpers import triangle_scene


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1344)
    ap.add_argument("--height", type=int, default=846)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--rings", type=int, default=48)
    ap.add_argument("--sectors", type=int, default=64)
    a = ap.parse_args()
    t0 = time.time()
    scene, mat = triangle_scene(seed=21, n_models=2, rings=a.rings, sectors=a.sectors)
    print("scene: %d triangles, %d nodes, built in %.1f s" % (scene.triangleCount, scene.node_buffer_length(), time.time() - t0), flush=True)
    for heat in (False, True):
        r = rt.RendererRaytracing(a.width, a.height, scene, maxBounces=a.bounces).initialize(None, mat)
        if heat:
            r.showHeatmap()
        ms = []
        for _ in range(a.frames + 1):
            scene.update(0.016)
            r.render()
            ms.append(r.stats()["kernel_ms"])
        st = r.stats()
        img = r.read_pixels()
        r.close()
        best = min(ms[1:])
        print("%s %dx%d B=%d: kernel %.3f ms (min of %d; all %s) rays %d  %.1f Mrays/s  %.1f fps  csum %d" % (
            "heatmap " if heat else "raytrace", a.width, a.height, a.bounces, best, a.frames, ["%.2f" % m for m in ms],
            st["rays"], st["rays"] / best / 1e3, 1e3 / best, int(img.astype(np.uint64).sum())), flush=True)


if __name__ == "__main__":
    main()
