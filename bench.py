#!/usr/bin/env python3
"""bench.py -- the headline measurement: Mrays/s (and frames/s) of the ray-trace hot path on
BASELINE.json's configuration C3 (3840x2160, 1024 spheres, 8 bounces), on N GPUs of one node.

    python bench.py                                  (= --gpus 1 --steps 100 --warmup 5)
    python bench.py --gpus N --steps K --warmup W    (N > 1 without a launcher: starts the line below as a CHILD process --
                                                      before anything here has touched torch or HIP --, relays rank 0's JSON
                                                      line and the exit code; --dry-launch prints the child's command instead)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one frame: per-frame scene preparation + the ray-trace kernel over this rank's row
tiles and -- for N > 1 -- the RCCL exchange of the tiles and their de-interleave into the full
frame, all enqueued by ONE C-ABI call (rt_render_gather, include/rt355.h: the library owns the
communicator).  torch.distributed is used for the control plane only (gloo: hands rank 0's RCCL
unique id to the other ranks, barriers, the max-over-ranks of the timings).  Scene, cube map and
parameters are resident in HBM before the timed region.  "ray" = one scene traversal
(primary/reflection RK:114 + shadow RK:153); the per-frame count is the kernel's own exact counter
(tests check it against the oracle).

Prints ONE JSON line on rank 0.
  * value / ms_per_step: K frames enqueued back to back (up to four overlap on the device);
    serial_ms_per_step: the same frames one at a time (render + wait, the reference's
    `await onSubmittedWorkDone`, RR:467), timed separately after the main region.
  * roofline: the binding roof of this path is FP32 VALU issue, not HBM.  `achieved`
    prices what the dominant kernel EXECUTES: VALU wave-instructions per launch (PMC pass committed
    under profiles/, summarised by tools/pmc_summary.py into profiles/traffic.json -- `from_profile`
    says which) x 128 fp32 flop slots (a wave64 VALU instruction holds a SIMD-32 for 2 cycles, 32
    lanes x 2 flop each: the unit the 157.3 TFLOP/s vector peak is made of) / the frame time.  frac <= 1.
    The ALGORITHMIC rate of the reference's test-every-sphere loop (25 flop x N x rays / time) is kept
    apart as `algorithmic_speedup_vs_bruteforce`.  The HBM fraction the north_star asks for is `hbm`.
  * frame_check: sha256 of the last timed frame against tests/golden/frames.json (the oracle's frame).
  * exchange (N > 1): "rccl" -- the product's path -- or "host": rt_comm_init was refused on some rank (exchange_error says why)
    and every rank fell back together to its rows read back and all-gathered over gloo, every frame awaited.  A record with
    its cause in it, not a figure of the product's.
  * cpu_baseline: the CPU oracle (the repository's scalar restatement of the shader: kind "port") on a
    bounded sample of the same frame.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Four frames are kept in flight on four streams, next to RCCL's kernels and the library's copies: with
# HIP's default of 4 hardware queues several of them share one queue and serialise (measured: 3.67
# instead of 2.43 ms per frame through the N > 1 path).  Must be set before the HIP runtime loads.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver: dmabuf IPC only (RCCL peer access between ranks)
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")          # one node: RCCL's bootstrap needs no other interface

FLOP_PER_TEST = 25          # WGSL-literal count of HK:308-311
PEAK_FP32_TFLOPS = 157.3    # MI355X_MICROARCH.md "Peak FP32 (vector)" = 1024 SIMDs x 64 flop/clk x 2.4 GHz
FLOP_SLOTS_PER_VALU = 128   # one wave64 VALU instruction = 2 cycles of a SIMD-32 = 2 x 32 lanes x 2 flop
PEAK_HBM_GBPS = 8000.0      # MI355X_MICROARCH.md "HBM3E peak BW" (spec)
PEAK_L2_GBPS = 34500.0      # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate
# The reference's live scene type (triangles behind a two-level BVH, RK:168-410) on a procedural scene of the
# reference scene's size (12.8 k triangles, 3 instances + floor): the window of the reference's screenshot and 4K
TRI_CONFIGS = {"TRI": dict(width=1344, height=846, bounces=4), "TRI4K": dict(width=3840, height=2160, bounces=4),
               # the reference's OWN scene in the state of its screenshot (cat 428 + mousey 12,174 + floor 2 triangles, the
               # daylight sky box, 1344x846, 4 bounces: what its overlay reports "6 ms" for), from the
               # fixtures tests/golden/make_ref_scene.py committed; mousey's texture is missing upstream: white stand-in
               "REF": dict(width=1344, height=846, bounces=4, fixture=True)}
FLOP_PER_NODE_TEST = 16     # the walk's conservative sphere test: 8 FMAs (rt_bvh.hip)
FLIGHT = 4                  # frames the library keeps concurrent (rt_ctx rotates over 4 streams / buffer sets)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=None, help="BASELINE config (default C3; C4 when --gpus > 1)")
    ap.add_argument("--mode", default="fast", choices=["fast", "strict"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--serial", action="store_true",
                    help="the MAIN timed region runs one frame at a time (render + wait, RR:467); default: frames "
                         "are enqueued back to back and up to four overlap on the device")
    ap.add_argument("--serial-steps", type=int, default=-1,
                    help="frames of the separate one-at-a-time measurement (default min(steps, 20); 0: skip)")
    ap.add_argument("--gather", default="root", choices=["root", "all"],
                    help="N > 1: tiles go to rank 0 only (grouped ncclSend/ncclRecv) or to every rank (ncclAllGather)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle sample")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region of K steps is run this many times in the one invocation: ms_per_step / value are the FIRST "
                         "region's (the contract's K steps), ms_per_step_median / _min are taken over all of them")
    ap.add_argument("--no-node", action="store_true", help="skip the measurement through the Node host (node/bench-frames.js)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="--gpus N > 1 without a launcher: print the torch.distributed.run command the ranks would be started with, and exit")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the N>1 code path (rt_comm_init + rt_render_gather) even with one rank; used by "
                         "tests/test_bench_gpu.py to exercise that path on a 1-GPU box")
    return ap.parse_args()


def cpu_baseline(cfg, scene, sky, target_s):
    """Times the CPU oracle on every `step`-th 8-row tile of the same frame (bounded sample)."""
    from oracle import rt_oracle_py as orc
    W, H, B = cfg["width"], cfg["height"], cfg["bounces"]
    p, s = scene.pack_params(B), scene.pack_spheres()
    # the GPU box's CPU share for one GPU is 16 hardware threads; never use more than that
    threads = max(1, min(orc.max_threads(), len(os.sched_getaffinity(0)), 16))
    ntiles = (H + 7) // 8
    # calibrate on a thin sample, then size the timed sample for ~target_s seconds
    step0 = max(1, ntiles // 4)
    t0 = time.perf_counter()
    _, _, rays0 = orc.render(p, s, sky.faces, W, H, tile_first=step0 // 2, tile_step=step0, threads=threads)
    dt0 = time.perf_counter() - t0
    n0 = len(range(step0 // 2, ntiles, step0))
    per_tile = dt0 / max(n0, 1)
    want = max(1, min(ntiles, int(target_s / max(per_tile, 1e-9))))
    step = max(1, ntiles // want)
    t0 = time.perf_counter()
    _, _, rays = orc.render(p, s, sky.faces, W, H, tile_first=step // 2, tile_step=step, threads=threads)
    dt = time.perf_counter() - t0
    n = len(range(step // 2, ntiles, step))
    return {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": "oracle/rt_oracle.c (scalar fp32 C, OpenMP over rows) on %d of %d 8-row tiles (every %d-th, "
                  "from tile %d) of the same %dx%d/%d-sphere/%d-bounce frame: %d rays in %.1f s"
                  % (n, ntiles, step, step // 2, W, H, cfg["spheres"], B, rays, dt),
        "fps_equiv": (n / ntiles) / dt,
    }


def cpu_baseline_tri(cfg, scene, mat, sky, target_s, gpu_frame):
    """The triangle path of the oracle on every `step`-th tile; also the work counters (node / triangle /
    instance gathers per ray) the GPU kernel's gather roofline is priced with, and a check of the GPU frame's
    sampled rows against the oracle's."""
    import numpy as np
    from oracle import rt_oracle_py as orc
    from compute_raytracer_amd.procedural import tri_buffers
    W, H, B = cfg["width"], cfg["height"], cfg["bounces"]
    p, b = scene.pack_params(B), tri_buffers(scene, mat)
    threads = max(1, min(orc.max_threads(), len(os.sched_getaffinity(0)), 16))
    ntiles = (H + 7) // 8
    step0 = max(1, ntiles // 4)
    t0 = time.perf_counter()
    orc.render_tri(p, b, sky.faces, W, H, tile_first=step0 // 2, tile_step=step0, threads=threads)
    per_tile = (time.perf_counter() - t0) / max(len(range(step0 // 2, ntiles, step0)), 1)
    want = max(1, min(ntiles, int(target_s / max(per_tile, 1e-9))))
    step = max(1, ntiles // want)
    orc.tri_counters()
    t0 = time.perf_counter()
    img, _, rays = orc.render_tri(p, b, sky.faces, W, H, tile_first=step // 2, tile_step=step, threads=threads)
    dt = time.perf_counter() - t0
    nodes, tests, blas = orc.tri_counters()
    rows = [y for y in range(H) if (y // 8) >= step // 2 and ((y // 8) - step // 2) % step == 0]
    n = len(range(step // 2, ntiles, step))
    return {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": "oracle/rt_oracle.c triangle path (scalar fp32 C, OpenMP over rows) on %d of %d 8-row tiles (every %d-th) "
                  "of the same %dx%d / %d-triangle / %d-bounce frame: %d rays in %.1f s" % (n, ntiles, step, W, H, scene.triangleCount, B, rays, dt),
        "fps_equiv": (n / ntiles) / dt,
        "gather_bytes_per_ray": (32.0 * nodes + 48.0 * tests + 80.0 * blas) / max(rays, 1),
        "per_ray": {"node_loads_32B": nodes / max(rays, 1), "triangle_tests_48B": tests / max(rays, 1), "instance_records_80B": blas / max(rays, 1)},
        "gpu_rows_match": bool(np.array_equal(gpu_frame[rows], img[rows])) if gpu_frame is not None else None,
    }


def a_no_children():
    return os.environ.get("RT355_BENCH_NO_CHILDREN") == "1"


def gpu_clocks():
    """The clock state of the GPU the line was measured on, asked of amd-smi (rocm-smi if that fails) in a child
    process: current / maximum shader and memory clocks and the performance level, whatever the tool reports of them."""
    import subprocess
    # under rocprofv3 the profiler's preloaded library initialises the GPU in every process it is loaded into, and the tools are
    # scripts that exec their interpreter: that exec is refused on this pool -- no clocks in a profiled run
    if any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR")) or a_no_children():
        return None
    for cmd in (["amd-smi", "metric", "--clock", "--json"], ["rocm-smi", "--showclocks", "--showperflevel", "--json"]):
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=30)
            if out.returncode != 0 or not out.stdout.strip():
                continue
            starts = [i for i in (out.stdout.find("{"), out.stdout.find("[")) if i >= 0]
            doc = json.loads(out.stdout[min(starts):])
            found = {}
            def walk(o, path):
                if isinstance(o, dict):
                    for k, v in o.items():
                        walk(v, path + [str(k)])
                elif isinstance(o, list):
                    for i, v in enumerate(o[:1]):          # the first GPU: the one rank 0 measures on
                        walk(v, path)
                else:
                    key = "/".join(path).lower()
                    if any(w in key for w in ("gfx_0", "gfx/", "sclk", "mem_0", "mclk", "perf", "fclk")) and len(found) < 24:
                        found["/".join(path)] = o
            walk(doc, [])
            return {"tool": " ".join(cmd), "values": found}
        except Exception:
            continue
    return None


def node_bench(name, scene, mat, sky, cfg, frames):
    """The same workload through the Node host (node/bench-frames.js: Node-12 CommonJS scene layer -> N-API addon -> C ABI), in a
    fresh child process -- never an exec; the parent's context is idle meanwhile.  C3: the scene comes from the Node layer's own
    generator; REF: the packed upload buffers are handed over as files.  -> the child's JSON line, or {"skipped": why}."""
    import shutil
    import subprocess
    import tempfile
    node = shutil.which("node")
    addon = os.path.join(ROOT, "node", "rt355.node")
    if not node or not os.path.exists(addon):
        return {"skipped": "no node binary" if not node else "node/rt355.node not built"}
    tmp = None
    try:
        if name in ("C3", "C2", "C1"):
            arg = name
        else:
            import numpy as np
            tmp = tempfile.mkdtemp(prefix="rt355_node_")
            f = lambda a: [float(v) for v in np.asarray(a).reshape(-1)]
            sky_file = os.path.join(tmp, "sky.rgba")
            with open(sky_file, "wb") as fh:
                for face in sky.faces:
                    fh.write(np.ascontiguousarray(face, dtype=np.uint8).tobytes())
            doc = {"width": cfg["width"], "height": cfg["height"], "bounces": cfg["bounces"], "tlasNodesMax": scene.tlasNodesMax,
                   "camera": {"position": f(scene.camera.position), "forwards": f(scene.camera.forwards),
                              "right": f(scene.camera.right), "up": f(scene.camera.up)},
                   "light": {"position": f(scene.light.position), "lightIntensity": scene.light.lightIntensity,
                             "minIntensity": scene.light.minIntensity},
                   "packed": {"triangleData": f(scene.pack_triangles()), "nodeDataB": f(scene.pack_blas_nodes()),
                              "triangleIndexData": f(scene.pack_tri_lookup())},
                   "frame": {"blasData": f(scene.pack_blas()), "blasIndexData": f(scene.pack_blas_lookup()),
                             "nodeDataA": f(scene.pack_tlas_nodes())},
                   "meshTexture": {"width": int(mat.image.shape[1]), "height": int(mat.image.shape[0]), "data": mat.image.reshape(-1).tolist()}}
            if sky.faces[0].shape[0] > 1:
                doc["sky"] = {"size": int(sky.faces[0].shape[0]), "file": sky_file}
            arg = os.path.join(tmp, "scene.json")
            json.dump(doc, open(arg, "w"))
        env = dict(os.environ)
        env.pop("GPU_MAX_HW_QUEUES", None)       # the library's own default is what a Node host gets (rt_create)
        out = subprocess.run([node, os.path.join(ROOT, "node", "bench-frames.js"), arg, str(frames)], capture_output=True, text=True,
                             timeout=600, env=env, cwd=ROOT)
        if out.returncode != 0:
            return {"skipped": "node/bench-frames.js failed: " + out.stderr.strip()[-300:]}
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:                       # the Node figure is an extra: its absence must not cost the line
        return {"skipped": "%s: %s" % (type(e).__name__, e)}
    finally:
        if tmp:
            shutil.rmtree(tmp, ignore_errors=True)


def reduce_over_ranks(torch, dist, elapsed, rays_local, kernel_ms, gather_ms, serial_ms, region_ms, serial_regions):
    """The contract's "take the MAX over ranks": every time is the slowest rank's, the rays of the frame are the ranks' sum.  Every
    rank must bring lists of the same lengths (the same --repeats, the same decision whether a serial measurement was made).
    -> (elapsed, rays_frame, kernel_ms, gather_ms, serial_ms, region_ms, serial_regions); tests/test_tiles_gloo.py runs it over
    gloo with two and three ranks on the CPU."""
    t = torch.tensor([elapsed, float(rays_local), kernel_ms, gather_ms, serial_ms or 0.0] + list(region_ms) + list(serial_regions or []),
                     dtype=torch.float64)
    tmax = t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tsum = t.clone()
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    n = len(region_ms)
    return (float(tmax[0]), int(round(float(tsum[1]))), float(tmax[2]), float(tmax[3]),
            float(tmax[4]) if serial_ms is not None else None,
            [float(v) for v in tmax[5:5 + n]],                                   # every region: the slowest rank's time
            [float(v) for v in tmax[5 + n:]] if serial_regions else serial_regions)


def agree_on_exchange(dist, world, error):
    """Every rank brings its rt_comm_init verdict (None, or the library's message); all of them leave with the same answer:
    ("rccl", None), or ("host", the first rank's message) if ANY rank was refused.  tests/test_tiles_gloo.py runs it over gloo."""
    errs = [None] * world
    dist.all_gather_object(errs, error)
    first = next((e for e in errs if e), None)
    return ("host", first) if first else ("rccl", None)


def kernel_label(kernel_id):
    """What the library says it launched (rt_stats.kernel_id -- no copy of its dispatch rules here)."""
    from compute_raytracer_amd import abi
    name = abi.load().rt_kernel_name(int(kernel_id)).decode()
    return name, abi.KERNEL_IDS.get(int(kernel_id), "?").startswith("hierarchy"), abi.KERNEL_IDS.get(int(kernel_id)) == "brute_pipeline"


def golden_frame(name):
    """sha256 / rays of the oracle's frame for a BASELINE config, if committed (C4 is the C3 frame)."""
    try:
        fr = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))
        return fr.get("C3" if name == "C4" else name)
    except Exception:
        return None


def launch_ranks(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment): the ranks are started
    as a child `python -m torch.distributed.run` of THIS process, which has not imported torch nor made any HIP call and makes
    none afterwards -- it relays the child's stdout (rank 0's one JSON line) and stderr, and exits with the child's code.
    (Never an exec: replacing a process is fine only before the GPU has been touched, and a child is fine always.)"""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:                  # a free port on the loopback interface
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    argv = [x for x in sys.argv[1:] if x != "--dry-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + argv
    if a.dry_launch:
        print(json.dumps({"launch": cmd}), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")           # torchrun would set it, with a warning on stderr
    return subprocess.run(cmd, env=env).returncode  # stdout / stderr are inherited: the JSON line arrives as the child prints it


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    # stdout carries ONE line, the JSON.  Libraries print banners there from native code (gloo: "[Gloo] Rank 0 is
    # connected ...", RCCL: its version block at the first communicator), so file descriptor 1 points at stderr
    # until the line is ready.
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RT355_BENCH_ONE_DEVICE") == "1":     # testing aid: every rank on device 0 (a one-GPU box rehearsing N > 1, if RCCL permits it)
        local_rank = 0
    if world != a.gpus:
        a.gpus = world                    # a launcher's WORLD_SIZE is what there is

    import torch  # torch.distributed (control plane) and torch.cuda.synchronize only; no tensor touches the data path
    import compute_raytracer_amd as rt
    from compute_raytracer_amd import tiles
    from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA

    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    multi = world > 1 or a.force_dist
    dist = None
    if multi:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:          # --force-dist without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"),
                              RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dist.init_process_group("gloo")              # control plane; the pixels travel over RCCL inside librt355.so

    name = a.config or ("C3" if world == 1 else "C4")
    tri = name in TRI_CONFIGS
    if not tri and name not in rt.BASELINE_CONFIGS:
        sys.exit("bench.py: unknown config %s" % name)
    mat = None
    ref_sky = None
    if tri and TRI_CONFIGS[name].get("fixture"):
        import numpy as np
        from PIL import Image
        g = os.path.join(ROOT, "tests", "golden")
        d = np.load(os.path.join(g, "ref_scene.npz"))
        scene = rt.SceneRaytracing.from_packed(d)
        mat = rt.Material.white()
        strip = np.array(Image.open(os.path.join(g, "ref_sky.png")).convert("RGBA"), dtype=np.uint8)
        ref_sky = rt.CubemapMaterial()
        ref_sky.faces = [np.ascontiguousarray(strip[:, k * strip.shape[0]:(k + 1) * strip.shape[0]]) for k in range(6)]
        cfg = dict(TRI_CONFIGS[name], spheres=0, seed=0, skybox=None)
    elif tri:
        from compute_raytracer_amd.procedural import triangle_scene
        cfg = dict(TRI_CONFIGS[name], spheres=0, seed=21, skybox=None)
        scene, mat = triangle_scene(seed=21, n_models=2, rings=48, sectors=64)     # 12,846 triangles, ~10 s of host build
    else:
        cfg = rt.BASELINE_CONFIGS[name]
    W, H, N, B = cfg["width"], cfg["height"], cfg["spheres"], cfg["bounces"]
    if not tri:
        scene = rt.synthetic_scene(N, cfg["seed"])
    sky_name = "constant sky"
    if cfg["skybox"]:
        # C5: BASELINE.md names the reference's daylight-skybox.png; its six 512x512 faces (cubemap-material.ts:40-47) are
        # the committed fixture tests/golden/ref_sky.png
        from PIL import Image
        import numpy as np
        strip = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "ref_sky.png")).convert("RGBA"), dtype=np.uint8)
        sky = rt.CubemapMaterial()
        sky.faces = [np.ascontiguousarray(strip[:, k * strip.shape[0]:(k + 1) * strip.shape[0]]) for k in range(6)]
        sky_name = "the reference's daylight sky box (6 x 512^2)"
    else:
        sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
    if ref_sky is not None:
        sky = ref_sky

    r = rt.RendererRaytracing(W, H, scene, device=local_rank, maxBounces=B)
    r.initialize(sky, mat)
    r.set_mode(a.mode == "strict")
    r.set_variant(a.variant)
    root = 0 if a.gather == "root" else -1
    # How the ranks' rows reach the frame: RCCL inside the library.  If rt_comm_init is REFUSED on any rank (two ranks rehearsing on
    # one device: RCCL's duplicate-device rule; a node whose RCCL cannot initialise), every rank falls back TOGETHER to its rows read
    # back and all-gathered over the control plane (gloo), every frame awaited -- slow, and said so in the line (`exchange`,
    # config.parallelism): a labelled curve instead of no record.  The rendering is the library's either way.
    exchange = "rccl" if multi else None
    exchange_error = None
    host_frame = [None]
    if multi:
        ids = [rt.RendererRaytracing.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        try:
            r.comm_init(ids[0], rank, world)          # collective: ncclCommInitRank on this rank's GPU
        except rt.abi.RtError as e:
            exchange_error = "rank %d: %s" % (rank, e)
        exchange, exchange_error = agree_on_exchange(dist, world, exchange_error)
        if exchange == "host":
            r.close()
            r = rt.RendererRaytracing(W, H, scene, device=local_rank, maxBounces=B, rank=rank, world=world)   # the partition without the communicator
            r.initialize(sky, mat)
            r.set_mode(a.mode == "strict")
            r.set_variant(a.variant)
            import numpy as np
            host_rows = np.zeros((tiles.padded_tiles(H, world) * 8, W, 4), dtype=np.uint8)
    r.recalculateScene()   # uploads: scene resident in HBM before anything is timed

    def step(serial):
        if exchange == "host":
            r.enqueue(); r.wait()
            pix = r.read_pixels()                     # this rank's tiles, in order (only the frame's last tile can be short)
            host_rows[:pix.shape[0]] = pix
            g = tiles.all_gather_frame(torch.from_numpy(host_rows), W, H)
            if rank == 0 or root < 0:
                host_frame[0] = tiles.assemble_torch(g, W, H, world)
        elif multi:
            r.enqueue_gather(root)        # this rank's tiles + RCCL exchange + de-interleave: one C-ABI call (rt_render_gather)
        else:
            r.enqueue()                   # prep + ray-trace kernel; the library rotates its streams
        if serial:
            r.wait()

    def fence():
        r.wait()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            r.wait()

    def timed(steps, serial):
        """-> (elapsed s, kernel ms sum, gather ms sum, frames with event times)"""
        kms = gms = 0.0
        frames = 0
        fence()
        t0 = time.perf_counter()
        done = 0
        while done < steps:
            chunk = min(steps - done, 64)     # stay inside the library's event ring (RT355_MAX_IN_FLIGHT)
            for _ in range(chunk):
                step(serial)
            done += chunk
            if done < steps:
                r.wait()
                st = r.stats()
                kms += st["batch_kernel_ms"]; gms += st["batch_gather_ms"]; frames += st["batch_frames"]
        fence()
        elapsed = time.perf_counter() - t0
        st = r.stats()
        kms += st["batch_kernel_ms"]; gms += st["batch_gather_ms"]; frames += st["batch_frames"]
        return elapsed, kms, gms, frames      # (serial: every rt_wait is a batch of one; the caller samples instead)

    # Prime every stream and buffer set the library rotates over (the first frame on a stream pays for its
    # hardware queue, the first gather on a buffer set for RCCL's and the runtime's lazy set-up: 7 ms that a
    # warm-up shorter than the rotation would leave inside the timed region), then the W warm-up steps.
    # ... and for at least 40 ms of frames: the device's clocks ramp over tens of milliseconds, and thirteen frames of a 0.2-0.5 ms
    # configuration are over before they have (the first timed region of the triangle configurations read 4-6 % above the later
    # ones; C3's thirteen frames are 20 ms and its first region was within 0.5 %).
    # ... in batches as long as the timed region's (the library's event ring holds 64 frames; an event's first use allocates its
    # signal, ~10 us each: a first region of 20 frames behind batches of 8 paid for twelve untouched slots, 0.25 ms whatever the
    # configuration -- 6 % of REF's 4.4 ms region, 0.7 % of C3's).
    t_prime = time.perf_counter()
    primed = 0
    batch = max(2 * FLIGHT, min(a.steps, 64))
    while primed < 2 * FLIGHT or (time.perf_counter() - t_prime < 0.040 and primed < 4096):
        for _ in range(batch):
            step(a.serial)            # (--serial: one at a time here too, so that a profiler sees launches of one kind only)
        primed += batch
        r.wait()
    fence()
    for _ in range(a.warmup):
        step(a.serial)
    regions = [timed(a.steps, a.serial) for _ in range(max(1, a.repeats))]      # every region: exactly K steps between two fences
    elapsed, kms, gms, kframes = regions[0]
    region_ms = [e / a.steps * 1e3 for (e, _, _, _) in regions]
    rays_local = r.stats()["rays"]
    kid_main = r.stats()["kernel_id"]                 # the form the timed frames ran as
    if a.serial:                      # every rt_wait reports its own batch of one: sample the per-frame times afterwards
        kms = gms = 0.0
        kframes = 0
        for _ in range(min(a.steps, 8)):
            step(True)
            st = r.stats()
            kms += st["kernel_ms"]; gms += st["gather_ms"]; kframes += 1

    # the last timed frame, hashed against the oracle's frame of this config
    check = None
    gold = None if tri else golden_frame(name)
    tri_frame = r.read_pixels() if tri and not multi and rank == 0 else None
    if gold is not None and (not multi or rank == 0 or root < 0):
        frame = host_frame[0].numpy() if exchange == "host" else (r.read_frame() if multi else r.read_pixels())
        check = {"sha256_matches_oracle_frame": hashlib.sha256(frame.tobytes()).hexdigest() == gold["sha256"],
                 "golden": "tests/golden/frames.json[%s]" % ("C3" if name == "C4" else name)}

    # one frame at a time, separately timed
    ssteps = min(a.steps, 20) if a.serial_steps < 0 else a.serial_steps
    serial_ms = None
    serial_regions = None
    if a.serial:
        serial_ms = elapsed / a.steps * 1e3
        serial_regions = region_ms
    elif ssteps > 0:
        for _ in range(8):            # untimed: the library sizes a frame's grid by how the caller has been enqueuing, and the
            step(True)                # triangle kernel's work list is made from the previous awaited frame on the same stream
        serial_regions = [timed(ssteps, True)[0] / ssteps * 1e3 for _ in range(max(1, a.repeats))]
        serial_ms = serial_regions[0]
    kid_serial = r.stats()["kernel_id"]

    # what a host that READS every frame gets (render, wait, copy the frame to pageable host memory: the PCIe-inclusive
    # rate -- never `value`), and, for a triangle scene, what the reference's animation loop costs (scene.update on the
    # host, the per-frame writes of RR:169-192, render, wait: src/app.ts:117-128)
    readback_ms = streamed_ms = animated_ms = animated_host_ms = loop_ms = None
    if not multi and ssteps > 0:
        r.enqueue(); r.wait(); r.read_pixels()
        t0 = time.perf_counter()
        for _ in range(ssteps):
            r.enqueue(); r.wait(); r.read_pixels()
        readback_ms = (time.perf_counter() - t0) / ssteps * 1e3
        # ... and frames in flight WITH every frame copied out: the streaming read-back (rt_read_pixels_async) copies frame
        # i - 2 to pinned host memory while frames i - 1 and i render
        host = r.host_frames(4)
        def streamed(n):
            fence()
            t0 = time.perf_counter()
            for i in range(n):
                if i and i % 48 == 0:
                    r.wait()                              # the library's event ring holds 64 frames
                r.enqueue()
                if i >= 2:
                    r.read_pixels_async(2, host[i % 4])
            r.read_pixels_async(1, host[(n + 2) % 4]); r.read_pixels_async(0, host[(n + 3) % 4])
            r.wait(); r.read_pixels_wait()
            return (time.perf_counter() - t0) / n * 1e3
        streamed(8)
        streamed_ms = streamed(max(ssteps, 24))
        # The loop a drop-in runs (src/app.ts:117-128, RR:435-469), one frame at a time: scene.update / camera.move on the host,
        # recalculateScene (rt_write_params; for a triangle scene also the three per-frame instance writes), rt_render, rt_wait.
        # The Python scene layer's own update (numpy: matrices, inverses, the top-level rebuild) is kept out of it -- the K
        # states are prepared beforehand, the loop installs one per step -- and reported by itself below (animated_host_scene_update_ms;
        # animated_ms_per_step is the loop WITH it).
        if tri:
            pose = scene.instances.eulers.copy()
            states = []
            for _ in range(ssteps):
                scene.update(0.016)
                states.append(scene.frame)
            r.render()
            t0 = time.perf_counter()
            for st_ in states:
                scene.frame = st_
                r.render()                                # recalculateScene + rt_render + rt_wait
            loop_ms = (time.perf_counter() - t0) / ssteps * 1e3
            scene.instances.eulers = pose
            scene.buildTopLevel()
        else:
            r.render()
            t0 = time.perf_counter()
            for _ in range(ssteps):
                scene.camera.move(0.0, 0.0)               # no key pressed: the picture stays the timed one
                r.render()
            loop_ms = (time.perf_counter() - t0) / ssteps * 1e3
        if tri:
            pose = scene.instances.eulers.copy()
            t0 = time.perf_counter()
            for _ in range(ssteps):
                scene.update(0.016)
                r.recalculateScene()
                r.enqueue(); r.wait()
            animated_ms = (time.perf_counter() - t0) / ssteps * 1e3
            t0 = time.perf_counter()
            for _ in range(ssteps):
                scene.update(0.016)
            animated_host_ms = (time.perf_counter() - t0) / ssteps * 1e3
            scene.instances.eulers = pose              # back to the timed frame's state (the oracle sample below renders it)
            scene.buildTopLevel()

    if multi:
        elapsed, rays_frame, kernel_ms, gather_ms, serial_ms, region_ms, serial_regions = reduce_over_ranks(
            torch, dist, elapsed, rays_local, kms / max(kframes, 1), gms / max(kframes, 1), serial_ms, region_ms, serial_regions)
        rays_kernel = rays_frame / world      # average rays per launch
    else:
        rays_frame = rays_local
        kernel_ms = kms / max(kframes, 1)
        gather_ms = 0.0
        rays_kernel = rays_frame

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = rays_frame * a.steps / elapsed / 1e6
        label, hierarchy, queue_pipeline = kernel_label(kid_main)
        # launches of consecutive frames overlap on the device (each on a share of the chip): the chip-level
        # rate is work per launch / frame period; one frame at a time: / the launch duration.  The two-kernel
        # brute-force pipeline shares a path queue: its frames are serialised by the library.
        overlapping = not a.serial and not queue_pipeline
        roof_ms = ms_per_step if overlapping else kernel_ms
        flops_alg = FLOP_PER_TEST * N * rays_kernel
        local_rows = tiles.tiles_of_rank(H, 0, world) * 8
        hbm_bytes = 4 * W * min(local_rows, H) + 32 * N + 96
        if tri:     # image store + the scene read once: 160-B triangles, 32-B nodes, 80-B instance records, two f32 lookups
            hbm_bytes = 4 * W * min(local_rows, H) + 164 * scene.triangleCount + 32 * scene.node_buffer_length() + 84 * len(scene.instances) + 96
        hbm_gbps = hbm_bytes / (roof_ms * 1e-3) / 1e9

        # executed work of the dominant kernel, from the committed PMC summary (tools/pmc_summary.py)
        prof, prof_key, estimated = None, None, False
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                prof_key = "%s/%s/v%d/n%d" % (name, a.mode, a.variant, world)
                prof = tj.get(prof_key)
                if prof is None and world > 1:      # per-rank launches of the row-tiled frame: 1/world of the one-GPU launch
                    prof_key = "%s/%s/v%d/n1" % ("C3" if name == "C4" else name, a.mode, a.variant)
                    prof = tj.get(prof_key)
                    estimated = prof is not None
            except Exception:
                prof = None
        # a profile is evidence for the build it was taken with: another build's instruction counts are not reported
        from compute_raytracer_amd import abi as _abi
        build_id = _abi.load().rt_build_id().decode()
        stale = prof is not None and prof.get("build_id") != build_id
        stale_note = None
        if stale:
            stale_note = "profiles/traffic.json[%s] was taken with build %s, this library is build %s: re-run tools/collect_profiles.sh" % (
                prof_key, prof.get("build_id"), build_id)
            prof = None
        executed = (prof or {}).get("executed", {})
        counts = (prof or {}).get("counts")
        valu_serial = executed.get("valu_wave_insts_per_launch")               # PMC pass of one-frame-at-a-time launches
        valu = executed.get("valu_wave_insts_per_launch_in_flight", valu_serial) if overlapping else valu_serial
        if valu is not None and estimated:
            valu = valu / world
            valu_serial = valu_serial / world
        roof = {"bound": "valu-issue (fp32 vector pipe)", "kernel": label, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "kernel_ms_avg": kernel_ms, "time_ms": roof_ms,
                "launches_in_flight": FLIGHT if overlapping else 1}
        if valu is not None:
            ach = valu * FLOP_SLOTS_PER_VALU / (roof_ms * 1e-3) / 1e12
            roof.update({
                "achieved": ach, "frac": ach / PEAK_FP32_TFLOPS,
                "valu_wave_insts_per_launch": valu,
                "from_profile": {"key": prof_key, "files": prof.get("source_files"), "kernel": prof.get("kernel"),
                                 "per_rank_estimate_insts_over_world": estimated},
                "traffic": None if estimated else prof.get("hbm_bytes_per_launch"),
                "basis": "achieved = VALU wave-instructions the kernel executes per launch (PMC SQ_INSTS_VALU, from_profile) x 128 "
                         "fp32 flop slots (2 SIMD-32 cycles x 32 lanes x 2) / time_ms; peak = 1024 SIMDs x 64 flop/clk x 2.4 GHz. "
                         "time_ms = frame period with launches_in_flight frames overlapping, else the launch duration.",
            })
            if serial_ms is not None and not queue_pipeline:
                s_ach = valu_serial * FLOP_SLOTS_PER_VALU / (serial_ms * 1e-3) / 1e12
                roof["serial"] = {"time_ms": serial_ms, "achieved": s_ach, "frac": s_ach / PEAK_FP32_TFLOPS}
            if counts:
                # what of the issue slots is the algorithm's own arithmetic: the FMAs of the node / leaf tests of the walk
                # (8 per test) and the flops of the literal evaluations (25 each, HK:308-311), counted by the counting
                # builds of the same kernel (tools/count_probe.py); lanes_busy = lane-steps with a live ray / all lane-steps
                useful_flop = FLOP_PER_NODE_TEST * counts["node_and_leaf_tests"] + FLOP_PER_TEST * counts["literal_tests"]
                if estimated:
                    useful_flop /= world
                u = useful_flop / (roof_ms * 1e-3) / 1e12
                roof["useful"] = {"achieved": u, "frac": u / PEAK_FP32_TFLOPS, "flop_per_launch": useful_flop,
                                  "node_and_leaf_tests_per_ray": counts["node_and_leaf_tests"] / max(counts["rays"], 1),
                                  "literal_tests_per_ray": counts["literal_tests"] / max(counts["rays"], 1),
                                  "lanes_busy": counts.get("lanes_busy")}
        else:
            roof.update({"achieved": None, "frac": None, "traffic": None,
                         "basis": stale_note or "no PMC pass for %s under profiles/: executed-instruction roofline not available" % prof_key})
        roof["hbm"] = {"achieved": hbm_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": hbm_gbps / PEAK_HBM_GBPS,
                       "bytes_per_launch": hbm_bytes}
        alg_tf = flops_alg / (roof_ms * 1e-3) / 1e12
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline_tri(cfg, scene, mat, sky, a.cpu_seconds, tri_frame) if tri else cpu_baseline(cfg, scene, sky, a.cpu_seconds)
        if tri:
            # The triangle kernel is a pointer chase: a node pair per step, ~17 dependent steps per ray.  Its memory side is
            # MEASURED (TCP / TCC passes of tools/collect_profiles.sh, profiles/traffic.json[...]["cache"]): `achieved` = L1 -> L2
            # read requests x 64 B per launch / time against the aggregate L2 bandwidth -- a fraction of a per cent: no
            # bandwidth binds this kernel, the latency of the dependent loads does (`l2.wait_share`: the part of the
            # wave-cycles spent in s_waitcnt).  What the lanes REQUEST (counted by the oracle on the sampled tiles: 32-B
            # nodes, 48-B corner triples, 80-B instance records) is kept as `requested`: the L1 absorbs most of it.
            valu_view = {k: roof.get(k) for k in ("achieved", "frac", "valu_wave_insts_per_launch", "from_profile", "serial") if k in roof}
            cache = (prof or {}).get("cache") or {}
            roof = {"bound": "latency of dependent loads (L1 / L2 gathers)", "kernel": label, "peak": PEAK_L2_GBPS, "unit": "GB/s", "kernel_ms_avg": kernel_ms,
                    "time_ms": roof_ms, "launches_in_flight": FLIGHT if overlapping else 1, "traffic": (prof or {}).get("hbm_bytes_per_launch"),
                    "valu": valu_view, "hbm": roof["hbm"], "achieved": None, "frac": None}
            if cache.get("l2_read_bytes_per_launch"):
                ach = cache["l2_read_bytes_per_launch"] / (roof_ms * 1e-3) / 1e9
                roof.update({"achieved": ach, "frac": ach / PEAK_L2_GBPS,
                             "l2": {"read_bytes_per_launch": cache["l2_read_bytes_per_launch"], "hit_rate": cache.get("l2_hit_rate"),
                                    "mean_read_latency_cycles": cache.get("mean_l2_read_latency_cycles"),
                                    "wait_share": (cache["wait_any_per_launch"] / executed["wave_cycles_per_launch"])
                                    if cache.get("wait_any_per_launch") and executed.get("wave_cycles_per_launch") else None},
                             "basis": "achieved = TCP_TCC_READ_REQ_sum x 64 B per launch (PMC pass of one-frame-at-a-time launches, from_profile) / time_ms; "
                                      "peak = aggregate L2 bandwidth (34.5 TB/s).  The kernel is latency-bound: see l2.wait_share and `valu`"})
            else:
                roof["basis"] = stale_note or "no TCP / TCC pass for %s under profiles/" % prof_key
            if cpu is not None:
                gbytes = cpu["gather_bytes_per_ray"] * rays_kernel
                roof["requested"] = {"achieved": gbytes / (roof_ms * 1e-3) / 1e9, "frac": gbytes / (roof_ms * 1e-3) / 1e9 / PEAK_L2_GBPS,
                                     "gather_bytes_per_launch": gbytes, "gathers_per_ray": cpu["per_ray"],
                                     "note": "bytes the lanes ask for, lane by lane (an upper bound of any cache level's traffic)"}
            check = {"sampled_tiles_match_oracle": cpu["gpu_rows_match"]} if cpu is not None else None
            if cfg.get("fixture") and tri_frame is not None:
                pin = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_pin.json")))
                check = dict(check or {}, sha256_matches_oracle_frame=hashlib.sha256(tri_frame.tobytes()).hexdigest() == pin["oracle_frame_sha256_white_texture"],
                             golden="tests/golden/ref_pin.json (the oracle's frame of this scene; it agrees with the reference's screenshot within one "
                                    "level of 255 on %.2f %% of the pixels that do not depend on the missing texture)"
                                    % (100 * pin["agreement_levels_of_255_max_over_channels"]["texture_free"]["within1"]))
        out = {
            "metric": ("Mrays/s at %dx%d, %d triangles, %d bounces" % (W, H, scene.triangleCount, B)) if tri else
                      ("Mrays/s at %dx%d, %d spheres, %d bounces" % (W, H, N, B)),
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "frames_per_s": 1e3 / ms_per_step,
            # the same K steps timed `repeats` times in this invocation (the first is ms_per_step): how far one region is from another
            "repeats": len(region_ms), "ms_per_step_median": sorted(region_ms)[len(region_ms) // 2], "ms_per_step_min": min(region_ms),
            "ms_per_step_all": region_ms,
            "serial_ms_per_step_median": sorted(serial_regions)[len(serial_regions) // 2] if serial_regions else None,
            "serial_ms_per_step_min": min(serial_regions) if serial_regions else None,
            # what a caller gets, by how it calls: the reference's loop awaits every frame (RR:467) = serial; a host that also
            # copies every frame out pays the PCIe read-back; `value` / ms_per_step are frames enqueued back to back
            "serial_ms_per_step": serial_ms,
            "serial_value": (rays_frame / (serial_ms * 1e-3) / 1e6) if serial_ms else None,
            "loop_ms_per_step": loop_ms,          # the reference's own loop through this library, one frame at a time (see above)
            "readback_ms_per_step": readback_ms,
            "streamed_readback_ms_per_step": streamed_ms,
            "animated_ms_per_step": animated_ms,
            "animated_host_scene_update_ms": animated_host_ms if animated_ms is not None else None,
            "kernel": {"id": int(kid_main), "name": label, "serial_id": int(kid_serial), "build_id": build_id},
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("REF: %dx%d, the reference's own scene in the state of its screenshot info/sample_settings.png (cat + mousey + floor: "
                                    "%d triangles, %d instances, %d nodes), %d bounces, its daylight sky box, white stand-in for the missing mousey texture "
                                    "(tests/golden/ref_scene.npz); the reference's overlay reports 6 ms per frame for it on an unnamed GPU"
                                    % (W, H, scene.triangleCount, len(scene.instances), scene.node_buffer_length(), B)) if (tri and cfg.get("fixture")) else
                                   ("%s: %dx%d, procedural triangle scene of the reference scene's size (%d triangles, %d instances, "
                                    "%d nodes; seed %d), %d bounces, constant sky, reference default camera/light"
                                    % (name, W, H, scene.triangleCount, len(scene.instances), scene.node_buffer_length(), cfg["seed"], B)) if tri else
                                   "%s: %dx%d, %d spheres (seed %d), %d bounces, %s, reference default camera/light"
                                   % (name, W, H, N, cfg["seed"], B, sky_name),
                       "mode": a.mode, "variant": a.variant, "rays_per_frame": rays_frame,
                       "frames_in_flight": 1 if a.serial else FLIGHT,
                       "parallelism": "row-tiles x%d%s" % (world, "" if not multi else
                                                           " + rows read back and all-gathered over gloo, every frame awaited (FALLBACK: rt_comm_init was refused)"
                                                           if exchange == "host" else
                                                           " + RCCL %s inside librt355 (rt_render_gather)" %
                                                           ("gather to rank 0" if root == 0 else "all-gather"))},
            "roofline": roof,
            "algorithmic_speedup_vs_bruteforce": None if tri else {
                "value": alg_tf / PEAK_FP32_TFLOPS, "algorithmic_tflops": alg_tf, "flop_per_launch": flops_alg,
                "note": "25 flop x N spheres x rays per launch / time_ms against the fp32 vector peak: how much faster the frame "
                        "is produced than a kernel that really examined every (ray, sphere) pair could at that peak"},
            "frame_check": check,
        }
        if multi:
            out["gather_ms_avg"] = gather_ms
            out["exchange"] = exchange
            if exchange_error:
                out["exchange_error"] = exchange_error
        out["scaling_note"] = ("ms_per_step / value: frames enqueued back to back (up to %d in flight); serial_ms_per_step / serial_value: every frame "
                               "awaited, the reference's loop (src/app.ts:124-127).  BASELINE.json's >= 6x at 8 GPUs is claimed for the frames-in-flight "
                               "figure; DESIGN.md 6 gives the emulated per-rank times for both." % FLIGHT)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        out["clocks"] = gpu_clocks()
        # the drop-in in the reference's host language, same workload (C3 and the reference's own scene)
        if not multi and not a.no_node and not a_no_children() and a.mode == "fast" and a.variant == 0 and name in ("C3", "REF"):
            nb = node_bench(name, scene, mat, sky, cfg, min(a.steps, 100))
            out["node"] = nb
            out["node_loop_ms_per_step"] = nb.get("awaitedMsPerFrame")            # `await renderer.render()` per frame (src/app.ts:124-127)
            out["node_inflight_ms_per_step"] = nb.get("inflightMsPerFrame")       # rt.render back to back, one await per batch
            out["node_streamed_readback_ms_per_step"] = nb.get("streamedMsPerFrame")
            if nb.get("sha256") and check and "sha256_matches_oracle_frame" in check:
                gold_sha = (golden_frame(name) or {}).get("sha256") if not tri else json.load(open(os.path.join(ROOT, "tests", "golden", "ref_pin.json")))["oracle_frame_sha256_white_texture"]
                out["node"]["frame_matches_oracle"] = nb["sha256"] == gold_sha
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)                     # whatever the teardown prints goes to stderr again

    r.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
