"""Development probe: what rt_render_gather adds to a frame (one rank, world = 1 communicator): frames in flight
through rt_render (enqueue) and through rt_render_gather, at the full C3 frame and at an eighth of its rows
(the work of one rank of eight).  usage: python tools/dist_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import compute_raytracer_amd as rt
cfg = rt.BASELINE_CONFIGS["C3"]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
for h in (2160, 272):
    for root in (None, 0, -1):
        r = rt.RendererRaytracing(cfg["width"], h, scene, maxBounces=cfg["bounces"]).initialize()
        if root is not None:
            r.comm_init(rt.RendererRaytracing.comm_unique_id(), 0, 1)
        r.recalculateScene()
        step = r.enqueue if root is None else (lambda: r.render_gather(root))
        for _ in range(8): step()
        r.wait()
        best, cpu = 1e9, 1e9
        for _ in range(4):
            r.wait(); t0 = time.perf_counter()
            for _ in range(32): step()
            t1 = time.perf_counter(); r.wait()
            best = min(best, (time.perf_counter() - t0) / 32 * 1e3); cpu = min(cpu, (t1 - t0) / 32 * 1e3)
        st = r.stats()
        print("3840x%-4d %-22s ms/frame %.3f  (host time per call %.3f)  kernel_ms %.3f gather_ms %.3f"
              % (h, "rt_render" if root is None else "rt_render_gather(%d)" % root, best, cpu, st["kernel_ms"], st.get("gather_ms", 0.0)), flush=True)
        r.close()
