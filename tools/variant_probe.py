"""Development probe: C3 (or another config) through chosen kernel variants: golden hash, one-frame-at-a-time
kernel time, frames-in-flight wall time per frame, and the same for rank 0 of 8.
usage: python tools/variant_probe.py [C3] [variants, e.g. 4,7,6]"""
import sys, os, hashlib, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import compute_raytracer_amd as rt

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "7,4").split(",")]
cfg = rt.BASELINE_CONFIGS[name]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
sky = rt.CubemapMaterial.synthetic_daylight() if (cfg["skybox"] or os.environ.get("RT355_PROBE_SKY")) else None
fr = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "frames.json"))).get(name)
for world in (1, 8):
    for v in variants:
        r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"], rank=0, world=world).initialize(sky)
        r.set_variant(v)
        ms = []
        for _ in range(10):
            r.render(); ms.append(r.stats()["kernel_ms"])
        st = r.stats()
        good = None
        if world == 1 and fr:
            good = hashlib.sha256(r.read_pixels().tobytes()).hexdigest() == fr["sha256"] and st["rays"] == fr["rays"]
        best = 1e9
        for _ in range(3):
            r.wait(); t0 = time.perf_counter()
            for _ in range(24):
                r.enqueue()
            r.wait(); best = min(best, (time.perf_counter() - t0) / 24 * 1e3)
        t0 = time.perf_counter()
        for _ in range(10):
            r.render()
        wall = (time.perf_counter() - t0) / 10 * 1e3
        print("%s world %d variant %d: golden %s  serial kernel ms min %.3f median %.3f (render+wait wall %.3f, prep %.3f)  pipelined %.3f"
              % (name, world, v, good, min(ms), sorted(ms)[len(ms) // 2], wall, st["prep_ms"], best), flush=True)
        r.close()
