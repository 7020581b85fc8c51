"""The N>1 host path (row-tile partition -> all-gather -> de-interleave) with world_size 2 and 3
over gloo on the CPU.  The per-rank pixel data comes from the CPU oracle standing in for the
HIP renderer (test infrastructure); what is under test is compute_raytracer_amd.tiles, the
logic bench.py and the GPU path share.  The gathered frame must be byte-identical to the
1-rank frame."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import compute_raytracer_amd as rt
from compute_raytracer_amd import tiles
from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, W, H, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import rt_oracle_py as orc
        scene = rt.synthetic_scene(12, 4242)
        sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
        p, s = scene.pack_params(3), scene.pack_spheres()
        # this rank's tiles, as the renderer would leave them: compact [padded_tiles][8][W][4]
        part, _, rays = orc.render(p, s, sky.faces, W, H, tile_first=rank, tile_step=world, threads=1)
        pt = tiles.padded_tiles(H, world)
        local = np.zeros((pt, 8, W, 4), np.uint8)
        for j in range(tiles.tiles_of_rank(H, rank, world)):
            y0 = (rank + j * world) * 8
            rows = min(8, H - y0)
            local[j, :rows] = part[y0:y0 + rows]
        t = torch.from_numpy(local.reshape(-1))
        assert t.numel() == tiles.message_bytes(W, H, world)
        gathered = tiles.all_gather_frame(t, W, H)
        frame = tiles.assemble_torch(gathered, W, H, world).numpy()
        frame2 = tiles.assemble_numpy(gathered.numpy(), W, H, world)
        full, _, full_rays = orc.render(p, s, sky.faces, W, H, threads=1)
        rays_t = torch.tensor([rays], dtype=torch.int64)
        dist.all_reduce(rays_t)
        ok = np.array_equal(frame, full) and np.array_equal(frame2, full) and int(rays_t[0]) == full_rays
        open(os.path.join(result_dir, "rank%d" % rank), "w").write("ok" if ok else "MISMATCH")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H", [(2, 40, 64), (2, 24, 53), (3, 16, 100)])
def test_gather_is_byte_identical_to_single_rank(tmp_path, world, W, H):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, W, H, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d" % r)).read() == "ok"


def test_assemble_round_trip_single_process():
    rng = np.random.default_rng(3)
    for (W, H, world) in [(8, 8, 1), (5, 53, 4), (12, 2160 // 8, 8), (3, 17, 2)]:
        frame = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
        pt = tiles.padded_tiles(H, world)
        g = np.zeros((world, pt, 8, W, 4), np.uint8)
        for y in range(H):
            r, j = tiles.owner_of_row(y, world)
            g[r, j, y & 7] = frame[y]
        assert np.array_equal(tiles.assemble_numpy(g, W, H, world), frame)
        assert np.array_equal(tiles.assemble_torch(torch.from_numpy(g.reshape(-1)), W, H, world).numpy(), frame)


def _control_plane_worker(rank, world, port, W, H, result_dir):
    """What bench.py does around the C-ABI calls when N > 1, with gloo standing in for RCCL: rank 0's
    128-byte communicator id reaches every rank (rt_comm_unique_id -> rt_comm_init), the tiles go to the
    ROOT only (rt_render_gather(root 0): grouped ncclSend / ncclRecv = a gather), the root de-interleaves,
    and the timings are reduced with MAX / the ray counts with SUM."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import rt_oracle_py as orc
        ids = [bytes(range(128)) if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ok = ids[0] == bytes(range(128)) and len(ids[0]) == 128
        scene = rt.synthetic_scene(9, 99)
        sky = rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
        p, s = scene.pack_params(2), scene.pack_spheres()
        part, _, rays = orc.render(p, s, sky.faces, W, H, tile_first=rank, tile_step=world, threads=1)
        pt = tiles.padded_tiles(H, world)
        local = np.zeros((pt, 8, W, 4), np.uint8)
        for j in range(tiles.tiles_of_rank(H, rank, world)):
            y0 = (rank + j * world) * 8
            rows = min(8, H - y0)
            local[j, :rows] = part[y0:y0 + rows]
        mine = torch.from_numpy(local.reshape(-1))
        slots = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, slots, dst=0)                       # rank r's message lands in slot r of the root's buffer
        if rank == 0:
            frame = tiles.assemble_numpy(torch.cat(slots).numpy(), W, H, world)
            full, _, full_rays = orc.render(p, s, sky.faces, W, H, threads=1)
            ok = ok and np.array_equal(frame, full)
        t = torch.tensor([0.5 + rank, float(rays)], dtype=torch.float64)
        tmax, tsum = t.clone(), t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        ok = ok and float(tmax[0]) == 0.5 + (world - 1)
        if rank == 0:
            ok = ok and int(round(float(tsum[1]))) == full_rays
        dist.barrier()
        open(os.path.join(result_dir, "rank%d" % rank), "w").write("ok" if ok else "MISMATCH")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H", [(2, 32, 53), (3, 24, 40)])
def test_bench_control_plane_and_gather_to_root(tmp_path, world, W, H):
    port = _free_port()
    mp.spawn(_control_plane_worker, args=(world, port, W, H, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d" % r)).read() == "ok"


def _reduce_worker(rank, world, port, result_dir):
    import importlib.util
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        # rank r: elapsed 1 + r, rays 100 (r + 1), five regions, the serial pair present on every rank
        got = bench.reduce_over_ranks(torch, dist, 1.0 + rank, 100 * (rank + 1), 0.5 + 0.1 * rank, 0.05 * (world - rank), 2.0 - 0.1 * rank,
                                      [1.0 + 0.01 * rank * k for k in range(5)], [2.0 + 0.02 * (world - rank) * k for k in range(5)])
        want = (1.0 + (world - 1), 100 * world * (world + 1) // 2, 0.5 + 0.1 * (world - 1), 0.05 * world, 2.0,
                [1.0 + 0.01 * (world - 1) * k for k in range(5)], [2.0 + 0.02 * world * k for k in range(5)])
        ok = all(np.allclose(g, w) for g, w in zip(got, want)) and isinstance(got[1], int)
        # ... and without a serial measurement (bench.py --serial-steps 0): None stays None, the lists keep their lengths
        got2 = bench.reduce_over_ranks(torch, dist, 3.0, 7, 0.1, 0.0, None, [0.5 + rank], None)
        ok = ok and got2[4] is None and got2[6] is None and np.allclose(got2[5], [0.5 + world - 1]) and got2[1] == 7 * world
        open(os.path.join(result_dir, "rank%d" % rank), "w").write("ok" if ok else "MISMATCH %r %r" % (got, got2))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_reduction_over_ranks(tmp_path, world):
    """bench.py's max-over-ranks / sum-of-rays step (the one piece of its N > 1 path that runs without a GPU), over gloo."""
    port = _free_port()
    mp.spawn(_reduce_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / ("rank%d" % r)).read() == "ok"


def _exchange_worker(rank, world, port, W, H, result_dir):
    import importlib.util
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        # nobody refused: the product's path; the LAST rank refused: every rank answers "host" with that rank's message
        a = bench.agree_on_exchange(dist, world, None)
        b = bench.agree_on_exchange(dist, world, "rank %d: refused" % rank if rank == world - 1 else None)
        ok = a == ("rccl", None) and b == ("host", "rank %d: refused" % (world - 1))
        # the labelled fallback's data path, as bench.py's step() does it: this rank's rows (tiles rank, rank + world, ...: only the
        # frame's last tile can be short) into the zero-padded message, all-gather over gloo, de-interleave
        frame = np.random.default_rng(5).integers(0, 256, (H, W, 4), dtype=np.uint8)       # the same picture on every rank
        mine = np.concatenate([frame[8 * t:8 * t + 8] for t in range(rank, tiles.total_tiles(H), world)] or [np.zeros((0, W, 4), np.uint8)])
        host_rows = np.zeros((tiles.padded_tiles(H, world) * 8, W, 4), dtype=np.uint8)
        host_rows[:mine.shape[0]] = mine
        g = tiles.all_gather_frame(torch.from_numpy(host_rows), W, H)
        ok = ok and np.array_equal(tiles.assemble_torch(g, W, H, world).numpy(), frame)
        open(os.path.join(result_dir, "rank%d" % rank), "w").write("ok" if ok else "MISMATCH %r %r" % (a, b))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(2, 1080), (3, 1077), (2, 20)])
def test_bench_exchange_verdict_and_host_fallback(tmp_path, world, H):
    """bench.py with N > 1: the ranks agree on how the rows travel (one refused rt_comm_init sends ALL of them to the labelled host
    exchange), and that exchange's buffer layout -- ragged tile counts, a short last tile -- rebuilds the frame."""
    port = _free_port()
    mp.spawn(_exchange_worker, args=(world, port, 64, H, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / ("rank%d" % r)).read() == "ok"
