"""What the ROOT of eight pays per C3 frame, emulated on one GPU: rank 0's share of the frame (34 of 270
row tiles) rendered into its slot of a gather buffer, four frames in flight on four streams -- (a) alone, (b) with the de-interleave
of the WHOLE 33 MB frame (assemble_frame over the gather buffer: what rt_render_gather enqueued behind every exchange until round
5) on the same stream behind it, (c) with the frame copied out of the gather buffer by one hipMemcpy2DAsync per rank instead
(rt_read_frame's way since round 5: the copy engine de-interleaves, no kernel).  The other ranks' tiles are whatever the buffer
holds: the cost of moving them does not depend on their contents.  The period to beat: 1.475 ms / 6 = 0.245 ms.
usage: python tools/root_probe.py [C3|C5] [world=8]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import compute_raytracer_amd as rt
from compute_raytracer_amd import abi, tiles

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = rt.BASELINE_CONFIGS[name]
W, H = cfg["width"], cfg["height"]
scene = rt.synthetic_scene(cfg["spheres"], cfg["seed"])
sky = None
if cfg["skybox"]:
    import numpy as np
    from PIL import Image
    strip = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "ref_sky.png")).convert("RGBA"), dtype=np.uint8)
    sky = rt.CubemapMaterial(); sky.faces = [np.ascontiguousarray(strip[:, k * strip.shape[0]:(k + 1) * strip.shape[0]]) for k in range(6)]
torch.cuda.set_device(0)
r = rt.RendererRaytracing(W, H, scene, maxBounces=cfg["bounces"], rank=0, world=world).initialize(sky)
r.recalculateScene()
L = abi.load()
hip = ctypes.CDLL("libamdhip64.so")
msg = tiles.padded_tiles(H, world) * 8 * W * 4
gather = [torch.zeros(world * msg, dtype=torch.uint8, device="cuda") for _ in range(4)]
frame = [torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda") for _ in range(4)]
host = [torch.zeros(H * W * 4, dtype=torch.uint8).pin_memory() for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
copy_stream = torch.cuda.Stream()
tile_bytes = 8 * W * 4
T = (H + 7) // 8

VP = ctypes.c_void_p
def render_to(k, s):          # the C ABI directly: the Python renderer's render_to re-packs the scene parameters per call, 0.1 ms of host time
    abi.check(L.rt_render_to(r._ctx, VP(gather[k].data_ptr()), msg, VP(s)), r._ctx)
def assemble(k, s):
    abi.check(L.rt_assemble_frame(r._ctx, VP(gather[k].data_ptr()), VP(frame[k].data_ptr()), world, VP(s)), r._ctx)

def frame_of(k, mode):
    s = streams[k].cuda_stream
    render_to(k, s)
    if mode == "assemble":
        assemble(k, s)
    elif mode == "copy2d":
        # rt_read_frame's copies, here enqueued behind the frame on its own stream (a host that reads every frame)
        for q in range(world):
            n = tiles.tiles_of_rank(H, q, world)
            full = n - 1 if (q == (T - 1) % world and H % 8) else n
            if full:
                rc = hip.hipMemcpy2DAsync(ctypes.c_void_p(host[k].data_ptr() + q * tile_bytes), ctypes.c_size_t(world * tile_bytes),
                                          ctypes.c_void_p(gather[k].data_ptr() + q * msg), ctypes.c_size_t(tile_bytes),
                                          ctypes.c_size_t(tile_bytes), ctypes.c_size_t(full), 2, ctypes.c_void_p(s))
                assert rc == 0, rc
    elif mode == "assemble+copy":
        assemble(k, s)
        rc = hip.hipMemcpyAsync(ctypes.c_void_p(host[k].data_ptr()), ctypes.c_void_p(frame[k].data_ptr()), ctypes.c_size_t(H * W * 4), 2, ctypes.c_void_p(s))
        assert rc == 0, rc

def run(mode, n):
    torch.cuda.synchronize(); r.wait()
    t0 = time.perf_counter()
    for i in range(n):
        if i and i % 48 == 0: r.wait()
        frame_of(i % 4, mode)
    r.wait(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print("%s as rank 0 of %d: message %.2f MB per rank, frame %.1f MB" % (name, world, msg / 1e6, H * W * 4 / 1e6), flush=True)
for mode in ("render", "assemble", "copy2d", "assemble+copy"):
    run(mode, 16)
    res = sorted(run(mode, 96) for _ in range(5))
    print("%-14s in flight: min %.3f  median %.3f ms per frame" % (mode, res[0], res[2]), flush=True)
r.close()
