#!/usr/bin/env python3
"""bench.py -- the headline measurement: Mrays/s (and frames/s) of the ray-trace hot path on
BASELINE.json's configuration C3 (3840x2160, 1024 spheres, 8 bounces), on N GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one frame: per-frame scene preparation + the ray-trace kernel over this rank's row
tiles (+ for N > 1 the RCCL all-gather of the tiles and the de-interleave into the full frame
on every rank).  Scene, cube map and parameters are resident in HBM before the timed region.
"ray" = one scene traversal (primary/reflection RK:114 + shadow RK:153); the per-frame count is
the kernel's own exact counter (tests check it against the oracle).

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel (the ray-trace kernel)
against the FP32 vector peak, which is the roof that binds this path; the HBM
fraction the north_star asks for is reported inside it as `hbm`.  `cpu_baseline` times the CPU
oracle (the repository's own scalar restatement of the shader: kind "port") on a bounded sample
of the same frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Four frames are kept in flight on four streams, next to RCCL's stream and the library's own: with
# HIP's default of 4 hardware queues several of them share one queue and serialise (measured: 3.67
# instead of 2.43 ms per frame through the N > 1 path).  Must be set before the HIP runtime loads.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

FLOP_PER_TEST = 25          # WGSL-literal count of HK:308-311
PEAK_FP32_TFLOPS = 157.3    # MI355X_MICROARCH.md "Peak FP32 (vector)"
PEAK_HBM_GBPS = 8000.0      # MI355X_MICROARCH.md "HBM3E peak BW" (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=None, help="BASELINE config (default C3; C4 when --gpus > 1)")
    ap.add_argument("--mode", default="fast", choices=["fast", "strict"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--serial", action="store_true",
                    help="one frame at a time (render + wait, as the reference does, RR:467); default: frames are "
                         "enqueued back to back and up to four overlap on the device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle sample")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the N>1 code path (render_to + all-gather + assemble) even with one rank; "
                         "used by tests/test_bench_gpu.py to exercise that path on a 1-GPU box")
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


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py: --gpus %d needs torch.distributed.run with --nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = world

    import torch  # device memory, streams and torch.distributed only
    import compute_raytracer_amd as rt
    from compute_raytracer_amd import tiles
    from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA

    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    multi = world > 1 or a.force_dist
    if multi:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:          # --force-dist without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"),
                              RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    name = a.config or ("C3" if world == 1 else "C4")
    if name not in rt.BASELINE_CONFIGS:
        sys.exit("bench.py: unknown config %s" % name)
    cfg = rt.BASELINE_CONFIGS[name]
    W, H, N, B = cfg["width"], cfg["height"], cfg["spheres"], cfg["bounces"]
    scene = rt.synthetic_scene(N, cfg["seed"])
    if cfg["skybox"]:
        png = os.path.join(ROOT, "assets", "daylight-skybox.png")    # the reference's asset, if the user supplies it
        sky = rt.CubemapMaterial.from_png(png) if os.path.exists(png) else rt.CubemapMaterial.synthetic_daylight()
    else:
        sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)

    r = rt.RendererRaytracing(W, H, scene, device=local_rank, maxBounces=B, rank=rank, world=world)
    r.initialize(sky)
    r.set_mode(a.mode == "strict")
    r.set_variant(a.variant)
    r.recalculateScene()   # uploads: scene resident in HBM before anything is timed

    FLIGHT = 4                          # frames kept concurrent (the library rotates rt_render over 4 streams itself)
    if multi:
        msg = tiles.message_bytes(W, H, world)
        # frame k runs on stream k % 4: render of this rank's tiles, all-gather (RCCL's stream, ordered
        # after the render), de-interleave; four frames are in flight, each with its own buffers
        streams = [torch.cuda.Stream() for _ in range(FLIGHT)]
        local = [torch.zeros(msg, dtype=torch.uint8, device="cuda") for _ in range(FLIGHT)]
        gathered = [torch.empty(world * msg, dtype=torch.uint8, device="cuda") for _ in range(FLIGHT)]
        frames = [torch.empty(H * W * 4, dtype=torch.uint8, device="cuda") for _ in range(FLIGHT)]
        frame = frames[0]
        torch.cuda.synchronize()
    pending = []       # [(work, buffer index)] gathers whose frame is not assembled yet
    counter = [0]

    def finish_oldest():
        work, k = pending.pop(0)
        with torch.cuda.stream(streams[k]):
            work.wait()                                               # stream k waits for the gather
            r.assemble_frame(gathered[k].data_ptr(), frames[k].data_ptr(), world, streams[k].cuda_stream)

    def step():
        if not multi:
            r.enqueue()                       # prep + ray-trace kernel; the library rotates its streams
            if a.serial:
                r.wait()
        else:
            k = counter[0] % FLIGHT
            counter[0] += 1
            while len(pending) >= FLIGHT:     # buffer set k is free once its previous frame is assembled
                finish_oldest()
            with torch.cuda.stream(streams[k]):
                r.render_to(local[k].data_ptr(), local[k].numel(), streams[k].cuda_stream)   # this rank's tiles
                work = dist.all_gather_into_tensor(gathered[k], local[k], async_op=True)      # RCCL over xGMI
            pending.append((work, k))
            if a.serial:
                while pending:
                    finish_oldest()
                torch.cuda.synchronize()

    def fence():
        if multi:
            while pending:                    # every step's frame is assembled inside the timed region
                finish_oldest()
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    r.wait()
    kernel_ms_sum, kernel_frames = 0.0, 0
    t0 = time.perf_counter()
    done = 0
    while done < a.steps:
        chunk = min(a.steps - done, 64)       # stay inside the library's event ring (RT355_MAX_IN_FLIGHT)
        for _ in range(chunk):
            step()
        done += chunk
        if done < a.steps:
            r.wait()
            st = r.stats()
            kernel_ms_sum += st["batch_kernel_ms"]
            kernel_frames += st["batch_frames"]
    fence()
    elapsed = time.perf_counter() - t0
    r.wait()
    st = r.stats()
    kernel_ms_sum += st["batch_kernel_ms"]
    kernel_frames += st["batch_frames"]
    rays_local = st["rays"]

    if multi:
        t = torch.tensor([elapsed, float(rays_local), kernel_ms_sum / max(kernel_frames, 1)],
                         dtype=torch.float64, device="cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays_frame = int(round(float(tsum[1])))
        kernel_ms = float(tmax[2])            # slowest rank's average kernel time
        rays_kernel = rays_frame / world      # average rays per launch
    else:
        rays_frame = rays_local
        kernel_ms = kernel_ms_sum / max(kernel_frames, 1)
        rays_kernel = rays_frame

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = rays_frame * a.steps / elapsed / 1e6
        # which kernels rendered the frame (the library's rule, rt_api.hip: enqueue)
        hierarchy = a.mode == "fast" and (a.variant == 4 or (a.variant == 0 and N >= 128))
        flops_launch = FLOP_PER_TEST * N * rays_kernel
        # launches of consecutive frames overlap on the device (each on a share of the chip), so the
        # chip-level rate is flops per launch / frame period; one frame at a time: / the launch duration
        # the two-kernel brute-force pipeline shares a path queue: its frames are serialised (rt_api.hip)
        queue_pipeline = (a.mode == "fast" and not hierarchy and
                          (a.variant in (2, 3) or (a.variant in (0, 4, 5) and N >= 320)))
        overlapping = not a.serial and not queue_pipeline
        in_flight = FLIGHT if overlapping else 1
        roof_ms = ms_per_step if overlapping else kernel_ms
        achieved_tf = flops_launch / (roof_ms * 1e-3) / 1e12
        local_rows = tiles.tiles_of_rank(H, 0, world) * 8
        hbm_bytes = 4 * W * min(local_rows, H) + 32 * N + 96
        hbm_gbps = hbm_bytes / (roof_ms * 1e-3) / 1e9
        if a.mode == "strict":
            kernel_label = "trace_pixels<FILTER=false> (literal loop)"
        elif hierarchy:
            kernel_label = "bvh_pixels (bounding-sphere hierarchy, one persistent kernel per frame)"
        elif N >= 320 and a.variant in (0, 5):
            kernel_label = "first_bounce + trace_paths (brute force, one frame's ray-trace launches)"
        else:
            kernel_label = "trace_pixels (brute force, single kernel)"
        roof_note = ("achieved = 25 flop x N spheres x rays per launch / time: the ALGORITHMIC work of the "
                     "reference's test-every-sphere loop (SURVEY.md 8(d)); time = the launch duration (kernel_ms_avg, "
                     "HIP events) when frames run one at a time (--serial), the frame period when launches of "
                     "consecutive frames overlap (launches_in_flight = 4, each on a quarter of the chip: kernel_ms_avg is "
                     "then ~4 frame periods)."
                     + (" The hierarchy evaluates ~5 % of those tests, so frac > 1 means 'faster than brute force could "
                        "run at the FP32 roof'; the executed-instruction view is in `executed`." if hierarchy else ""))
        executed = None
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                key = "%s/%s/v%d/n%d" % (name, a.mode, a.variant, world)
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
                executed = tj.get(key, {}).get("executed")      # PMC-derived, from the committed profile
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s at %dx%d, %d spheres, %d bounces" % (W, H, N, B),
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "frames_per_s": 1e3 / ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %dx%d, %d spheres (seed %d), %d bounces, constant sky, reference default "
                                   "camera/light" % (name, W, H, N, cfg["seed"], B),
                       "mode": a.mode, "variant": a.variant, "rays_per_frame": rays_frame,
                       "parallelism": "row-tiles x%d%s" % (world, "" if world == 1 else " + RCCL all-gather")},
            "roofline": {
                "bound": "valu-fp32", "kernel": kernel_label,
                "achieved": achieved_tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tf / PEAK_FP32_TFLOPS,
                "traffic": traffic,
                "kernel_ms_avg": kernel_ms, "flop_per_launch": flops_launch, "launches_in_flight": in_flight,
                "note": roof_note,
                "hbm": {"achieved": hbm_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": hbm_gbps / PEAK_HBM_GBPS, "bytes_per_launch": hbm_bytes},
            },
        }
        if executed is not None:
            out["roofline"]["executed"] = executed
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, scene, sky, a.cpu_seconds)
        print(json.dumps(out), flush=True)

    if multi and rank == 0 and os.environ.get("RT355_BENCH_CHECK_FRAME"):
        # test hook: hash of the assembled frame, to compare with the single-kernel path
        import hashlib
        torch.cuda.synchronize()
        print("frame_sha256 " + hashlib.sha256(frame.cpu().numpy().tobytes()).hexdigest(), file=sys.stderr, flush=True)
    r.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
