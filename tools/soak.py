"""Soak test (development): many frames, every read-back hashed against the golden C3 frame; with `sky` as second
argument the synthetic 6 x 512^2 sky (bvh_pixels records + sky_resolve), hashed against the first frame.
usage: python tools/soak.py [batches=150] [sky]"""
import os, sys, json, hashlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compute_raytracer_amd as rt
meta = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "frames.json")))["C3"]
cfg = rt.BASELINE_CONFIGS["C3"]
spheres = int(os.environ.get("SOAK_SPHERES", cfg["spheres"]))       # other counts (other kernel forms): hashed against the first frame
scene = rt.synthetic_scene(spheres, cfg["seed"])
textured = len(sys.argv) > 2 and sys.argv[2] == "sky"
r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"]).initialize(
    rt.CubemapMaterial.synthetic_daylight() if textured else None)
r.recalculateScene()
if textured or spheres != cfg["spheres"]:
    r.render()
    meta = {"sha256": hashlib.sha256(r.read_pixels().tobytes()).hexdigest(), "rays": r.stats()["rays"]}
bad = 0; t0 = time.time(); frames = 0
for batch in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    n = 1 + (batch * 7) % 23
    for _ in range(n):
        r.enqueue()
    img = r.read_pixels(); frames += n
    ok = hashlib.sha256(img.tobytes()).hexdigest() == meta["sha256"] and r.stats()["rays"] == meta["rays"]
    bad += not ok
    if not ok: print("MISMATCH in batch", batch, n, flush=True)
print("soak: %d frames in %d batches, %.1f s, mismatches %d" % (frames, batch + 1, time.time() - t0, bad))
r.close()
