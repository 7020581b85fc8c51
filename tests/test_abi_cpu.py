"""The C-ABI library on a machine without a GPU: it loads, exports every symbol that
include/rt355.h declares, refuses to work without a device (no CPU fallback) and its pure
helper functions agree with the Python host logic."""
import ctypes
import os
import re

import pytest

import compute_raytracer_amd as rt
from compute_raytracer_amd import abi, tiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "rt355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = abi.load()
    names = header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "librt355.so does not export %s" % n
    assert sorted(abi.SYMBOLS) == names, "abi.SYMBOLS is out of sync with include/rt355.h"
    assert lib.rt_abi_version() == 4
    assert len(lib.rt_build_id()) == 16 and lib.rt_kernel_name(4) == b"bvh_pixels<8>"


def test_kernel_ids_of_the_header_the_python_table_and_the_library_agree():
    """rt_stats.kernel_id says which kernel rendered a frame: the enum of include/rt355.h, abi.KERNEL_IDS and rt_kernel_name must
    list the same ids (10 = trace_roles, the awaited triangle frame whose work list splits tiles)."""
    text = open(os.path.join(ROOT, "include", "rt355.h")).read()
    ids = {name: int(v) for name, v in re.findall(r"\b(RT_KID_[A-Z_0-9]+)\s*=\s*(\d+)", text)}
    assert sorted(ids.values()) == sorted(abi.KERNEL_IDS) == list(range(len(ids)))
    assert ids["RT_KID_TRIANGLES"] == 8 and ids["RT_KID_TRIANGLES_ROLES"] == 10 and abi.KERNEL_IDS[10] == "triangles_roles"
    lib = abi.load()
    names = [lib.rt_kernel_name(i) for i in sorted(ids.values())]
    assert all(names) and len(set(names)) == len(names) and names[10] == b"trace_roles"


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = abi.load()
    ctx = ctypes.c_void_p()
    rc = lib.rt_create(0, ctypes.byref(ctx))
    assert rc == abi.RT_ERR_NO_DEVICE and not ctx.value
    assert b"no CPU path" in lib.rt_last_error(None)
    r = rt.RendererRaytracing(64, 64, rt.synthetic_scene(3, 1))
    with pytest.raises(abi.RtError) as e:
        r.initialize()
    assert e.value.code == abi.RT_ERR_NO_DEVICE


def test_null_arguments_are_rejected():
    lib = abi.load()
    assert lib.rt_create(0, None) == abi.RT_ERR_INVALID_ARG
    assert lib.rt_resize(None, 8, 8) == abi.RT_ERR_INVALID_ARG
    assert lib.rt_render(None) == abi.RT_ERR_INVALID_ARG
    assert lib.rt_wait(None) == abi.RT_ERR_INVALID_ARG
    assert lib.rt_destroy(None) == abi.RT_OK
    assert b"NULL" in lib.rt_last_error(None)


def test_triangle_entry_points_check_arguments():
    lib = abi.load()
    assert lib.rt_write_triangles(None, None, 0) == abi.RT_ERR_INVALID_ARG
    assert lib.rt_write_nodes(None, 0, None, 0) == abi.RT_ERR_INVALID_ARG
    assert lib.rt_write_blas(None, None, 0) == abi.RT_ERR_INVALID_ARG
    assert lib.rt_write_tri_lookup(None, None, 0) == abi.RT_ERR_INVALID_ARG
    assert lib.rt_write_blas_lookup(None, None, 0) == abi.RT_ERR_INVALID_ARG
    assert lib.rt_write_mesh_texture(None, 0, 0, None) == abi.RT_ERR_INVALID_ARG
    assert b"NULL" in lib.rt_last_error(None)


@pytest.mark.parametrize("H", [1, 7, 8, 9, 256, 1080, 2160, 4320, 53])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_tile_arithmetic_matches_host(H, world):
    lib = abi.load()
    total = 0
    for r in range(world):
        n = lib.rt_tiles_of_rank(H, r, world)
        assert n == tiles.tiles_of_rank(H, r, world)
        assert n == len([t for t in range(tiles.total_tiles(H)) if t % world == r])
        total += n
    assert total == tiles.total_tiles(H)
    assert lib.rt_padded_tiles(H, world) == tiles.padded_tiles(H, world) == max(
        tiles.tiles_of_rank(H, r, world) for r in range(world))
    assert lib.rt_tiles_of_rank(H, world, world) == 0 and lib.rt_padded_tiles(H, 0) == 0


def _plan(records, params):
    import numpy as np
    lib = abi.load()
    fp = ctypes.POINTER(ctypes.c_float)
    ok, sgn = ctypes.c_int(-1), ctypes.c_int(-1)
    r = np.ascontiguousarray(records, dtype=np.float32)
    abi.check(lib.rt_filter_plan(r.ctypes.data_as(fp), r.shape[0], params.ctypes.data_as(fp),
                                 ctypes.byref(ok), ctypes.byref(sgn)))
    return ok.value, sgn.value


def test_filter_plan_nan_and_inf_are_sticky_wherever_the_record_sits():
    """`if (!(len <= bound)) bound = len` let the sphere after a NaN record replace the
    NaN bound.  Any NaN / inf record -- first, middle or last -- must switch the filter forms off."""
    import numpy as np
    scene = rt.synthetic_scene(12, 5)
    p = scene.pack_params(4)
    base = scene.pack_spheres()
    assert _plan(base, p) == (1, 1)
    for bad in (np.nan, np.inf, -np.inf):
        for pos in (0, 5, 11):
            for field in (0, 1, 2, 7):
                s = base.copy()
                s[pos, field] = bad
                assert _plan(s, p) == (0, 0), (bad, pos, field)
    # [far sphere, NaN sphere, small spheres]: the far sphere's bound must not be forgotten either
    s = base.copy()
    s[0, 0] = 3.0e6
    s[1, 2] = np.nan
    assert _plan(s, p) == (0, 0)
    s[1, 2] = 0.0
    assert _plan(s, p) == (0, 0)                    # reach >= 2^20
    s[0, 0] = 400.0
    assert _plan(s, p) == (1, 0)                    # 342 <= reach < 2^20: unsigned filter only
    q = p.copy(); q[0] = np.nan
    assert _plan(base, q) == (0, 0)                 # NaN camera
    q = p.copy(); q[17] = np.inf
    assert _plan(base, q) == (0, 0)                 # light at infinity
    for k in (4, 9, 14):                            # the camera's basis: forwards.x, right.y, up.z
        for bad in (np.nan, np.inf, 3.0e30, -2.0e6):
            q = p.copy(); q[k] = bad
            assert _plan(base, q) == (0, 0), (k, bad)   # a NaN or absurd basis: the literal kernel (normalize() may return 0)
        q = p.copy(); q[k] = 500.0
        assert _plan(base, q) == (1, 0)             # a long basis vector counts like a far camera
    assert _plan(base[:0], p) == (1, 1)             # empty scene
    s = base.copy(); s[3, 7] = 0.0
    assert _plan(s, p) == (1, 1)                    # a zero radius is fine (nothing to rescale) ...
    s[4, 7] = 1e-10
    assert _plan(s, p) == (1, 0)                    # ... a radius in (0, 2^-30) keeps the filter but not its sign-aware, rescaled form
