"""Development probe: rt_render_gather frames in flight (world = 1) with the pieces bench.py has around it added one at
a time: torch's CUDA context, a gloo process group.  usage: python tools/dist_probe2.py [torch] [gloo]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
use_torch, use_gloo = "torch" in sys.argv, "gloo" in sys.argv
if use_torch or use_gloo:
    import torch
    torch.cuda.set_device(0)
if use_gloo:
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29578", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo")
import compute_raytracer_amd as rt
cfg = rt.BASELINE_CONFIGS["C3"]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
r = rt.RendererRaytracing(cfg["width"], cfg["height"], scene, maxBounces=cfg["bounces"]).initialize()
r.comm_init(rt.RendererRaytracing.comm_unique_id(), 0, 1)
r.recalculateScene()
for _ in range(3): r.render_gather(0)
def fence():
    r.wait()
    if use_torch or use_gloo: torch.cuda.synchronize()
    if use_gloo: dist.barrier()
    r.wait()
for steps in (20, 20, 200):
    fence(); t0 = time.perf_counter()
    done = 0
    while done < steps:
        chunk = min(steps - done, 64)
        for _ in range(chunk): r.render_gather(0)
        done += chunk
        if done < steps: r.wait()
    fence()
    print("%s: %d steps %.3f ms/frame" % (" ".join(sys.argv[1:]) or "bare", steps, (time.perf_counter() - t0) / steps * 1e3), flush=True)
r.close()
