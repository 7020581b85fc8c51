"""The multi-GPU exchange without a second GPU: what rt_comm.hip decides per rank and frame -- the
buffer a rank renders into, its RCCL operations (peer, offset, byte count, order), the wait-on relation between the
exchanges of consecutive frames, the stream / buffer set of a frame -- lives in the host-only header
compute_raytracer_amd/csrc/rt_exchange_plan.h, which rt_comm.hip executes.  tests/c/exchange_plan_test.cpp builds the
plans of every rank of a group (world 1, 2, 3, 4, 8; root -1, 0, last; ragged heights; 1-6 frames in flight) and checks
them against each other; here the same arithmetic is also held against the Python layer (tiles.py) and the C ABI."""
import os
import subprocess

import pytest

from compute_raytracer_amd import tiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plans_of_every_rank_pair_up(tmp_path):
    exe = str(tmp_path / "exchange_plan_test")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Werror", os.path.join(ROOT, "tests", "c", "exchange_plan_test.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "exchange plan ok" in r.stdout, r.stdout + r.stderr


def test_python_layer_and_c_abi_agree_with_the_plan():
    import ctypes
    from compute_raytracer_amd import abi
    L = abi.load()
    L.rt_tiles_of_rank.restype = ctypes.c_uint32
    L.rt_padded_tiles.restype = ctypes.c_uint32
    for H in (7, 8, 9, 846, 2160, 4320):
        for world in (1, 2, 3, 4, 8):
            assert L.rt_padded_tiles(H, world) == tiles.padded_tiles(H, world) == -(-((H + 7) // 8) // world)
            for rank in range(world):
                assert L.rt_tiles_of_rank(H, rank, world) == tiles.tiles_of_rank(H, rank, world)
    assert tiles.padded_tiles(2160, 8) == 34 and tiles.padded_tiles(4320, 8) == 68
