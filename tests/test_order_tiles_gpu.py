"""The two kernels that make an awaited triangle frame's work list (rt_triangles.hip: order_hist -- per-workgroup LDS histograms
into 128 global bins, the last workgroup to finish turns them into split counts and first positions -- and order_scatter -- one
global atomic per class and workgroup, then every tile placed), run on cost arrays of their own through the diagnostic entry point
rt_order_tiles: the list must be a permutation, classes must not increase along it, and the two header words must be what the
rule says (restated here in Python: quarter-octave classes, half / twice the throughput time, the caps)."""
import ctypes

import numpy as np
import pytest

from compute_raytracer_amd import abi

pytestmark = pytest.mark.gpu
U32 = ctypes.POINTER(ctypes.c_uint32)


def cost_class(c):
    c = int(c)
    if c < 4:
        return c
    e = c.bit_length() - 1
    return 4 * e + ((c >> (e - 2)) & 3) - 4


def model(cost, wave_slots, mult4=1, mult16=4, cap16=64):
    n = len(cost)
    cls = np.array([cost_class(c) for c in cost])
    total = int(np.asarray(cost, np.uint64).sum())
    thr = total // (2 * max(wave_slots, 1))
    cls_of = lambda v: cost_class(min(v, 0xFFFFFFFF))
    above = lambda k: int((cls > k).sum())
    split, split16 = above(cls_of(mult4 * thr)), above(cls_of(mult16 * thr))
    if above(cls_of(4 * thr)) == 0:
        split = split16 = 0
    most, most16 = min(n // 16, 1024), min(n // 64, cap16)
    split16 = min(split16, most16)
    split = max(min(split, most), split16)
    return split - split16, split16, cls


def run(cost, wave_slots=5120):
    L = abi.load()
    ctx = ctypes.c_void_p()
    abi.check(L.rt_create(0, ctypes.byref(ctx)))
    try:
        cost = np.ascontiguousarray(cost, np.uint32)
        order = np.full(len(cost) + 2, 0xFFFFFFFF, np.uint32)
        abi.check(L.rt_order_tiles(ctx, cost.ctypes.data_as(U32), len(cost), wave_slots, order.ctypes.data_as(U32), order.size), ctx)
        return order
    finally:
        L.rt_destroy(ctx)


@pytest.mark.parametrize("n", [1, 63, 1024, 1025, 4096, 17808, 129600, 300001])
def test_list_is_a_permutation_sorted_by_class_with_the_rule_s_split_counts(n):
    rng = np.random.default_rng(n)
    # mostly short tiles, a heavy tail (what a frame's tile times look like), a few zeros (tiles that recorded nothing)
    cost = (rng.lognormal(7.5, 1.0, n)).astype(np.uint64)
    cost[rng.integers(0, n, max(1, n // 50))] = 0
    cost[rng.integers(0, n, max(1, n // 200))] *= 40
    cost = np.minimum(cost, 0xFFFFFFFF).astype(np.uint32)
    order = run(cost)
    q4, q16, cls = model(cost, 5120)
    assert (int(order[0]), int(order[1])) == (q4, q16)
    perm = order[2:]
    assert np.array_equal(np.sort(perm), np.arange(n, dtype=np.uint32)), "not a permutation"
    along = cls[perm.astype(np.int64)]
    assert np.all(along[:-1] >= along[1:]), "classes increase along the list"


@pytest.mark.parametrize("kind", ["equal", "zeros", "huge", "one long tile"])
def test_degenerate_cost_arrays(kind):
    n = 20000
    cost = {"equal": np.full(n, 2000, np.uint32), "zeros": np.zeros(n, np.uint32),
            "huge": np.full(n, 0xFFFFFFFF, np.uint32), "one long tile": np.full(n, 1500, np.uint32)}[kind]
    if kind == "one long tile":
        cost[12345] = 40_000_000
    order = run(cost)
    q4, q16, cls = model(cost, 5120)
    assert (int(order[0]), int(order[1])) == (q4, q16)
    assert np.array_equal(np.sort(order[2:]), np.arange(n, dtype=np.uint32))
    if kind == "one long tile":
        assert order[2] == 12345 and q16 == 1            # the head of the list, as sixteenths


def test_consecutive_runs_leave_the_scan_space_clean():
    """The kernels zero their own bins and ticket for the next frame; run twice on one context's buffers via the renderer is what
    the picture tests do -- here: the diagnostic twice in a row with different sizes gives each its own right answer."""
    for n in (5000, 4097, 9999):
        cost = np.random.default_rng(n).integers(0, 100000, n).astype(np.uint32)
        order = run(cost)
        assert np.array_equal(np.sort(order[2:]), np.arange(n, dtype=np.uint32))
        assert (int(order[0]), int(order[1])) == model(cost, 5120)[:2]
