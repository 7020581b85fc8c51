"""Row-tile partition of a frame over ranks and the gather that rebuilds it.

A tile is 8 image rows (one row of the reference's 8x8 workgroups, RK:73 / RR:445).  Tile t is
rendered by rank t % world -- interleaved, so that sky rows (paths end at bounce 0) and ground
rows (all bounces + shadow rays) are spread evenly.  Each rank renders into a compact buffer
[padded_tiles][8][W][4]; the buffers are exchanged with ONE all-gather (RCCL over xGMI on GPUs,
gloo in the CPU tests) and de-interleaved into the row-major frame.

The arithmetic here (who owns what, how many, where it lands) is host logic and is shared by
the GPU path and the gloo tests; the pixel data itself only ever comes from the HIP kernels.
"""
import numpy as np


def total_tiles(height):
    return (height + 7) // 8


def tiles_of_rank(height, rank, world):
    """Tiles rank `rank` owns: rank, rank+world, ...  (mirrors rt_tiles_of_rank)."""
    t = total_tiles(height)
    return (t - rank + world - 1) // world if t > rank else 0


def padded_tiles(height, world):
    """Per-rank tile count padded to the max over ranks (mirrors rt_padded_tiles): the
    all-gather needs equal message sizes."""
    return (total_tiles(height) + world - 1) // world


def message_bytes(width, height, world):
    return padded_tiles(height, world) * 8 * width * 4


def owner_of_row(y, world):
    tile = y // 8
    return tile % world, tile // world  # (rank, local tile index)


def assemble_numpy(gathered, width, height, world):
    """Reference de-interleave on the host (used by the gloo tests and to check
    rt_assemble_frame): gathered is [world][padded][8][W][4] uint8."""
    pt = padded_tiles(height, world)
    g = np.asarray(gathered, dtype=np.uint8).reshape(world, pt, 8, width, 4)
    frame = np.empty((height, width, 4), dtype=np.uint8)
    for y in range(height):
        r, j = owner_of_row(y, world)
        frame[y] = g[r, j, y & 7]
    return frame


def all_gather_frame(local_tiles_tensor, width, height, group=None):
    """local_tiles_tensor: this rank's compact buffer, a uint8 torch tensor of
    message_bytes(width, height, world) bytes (device tensor on GPUs, CPU tensor under gloo).
    Returns the gathered [world * message_bytes] tensor (still interleaved)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    flat = local_tiles_tensor.reshape(-1)
    if flat.numel() != message_bytes(width, height, world):
        raise ValueError("all_gather_frame: local buffer must be padded_tiles*8*W*4 bytes")
    out = torch.empty(world * flat.numel(), dtype=torch.uint8, device=flat.device)
    dist.all_gather_into_tensor(out, flat, group=group)
    return out


def assemble_torch(gathered, width, height, world):
    """De-interleave with tensor ops (CPU/gloo path; on the GPU rt_assemble_frame does it)."""
    pt = padded_tiles(height, world)
    g = gathered.reshape(world, pt, 8, width, 4)
    # frame tile index = local*world + rank  ->  [pt][world][8][W][4]
    frame = g.permute(1, 0, 2, 3, 4).reshape(pt * world * 8, width, 4)
    return frame[:height].contiguous()
