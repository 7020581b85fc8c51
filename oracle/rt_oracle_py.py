"""ctypes wrapper around oracle/librt_oracle.so.  TEST INFRASTRUCTURE ONLY (see rt_oracle.h):
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from
the product package.  Pinned by the reference's one held output, its screenshot (rt_oracle.h; tests/test_ref_pin.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Face(ctypes.Structure):
    _fields_ = [("w", ctypes.c_uint32), ("h", ctypes.c_uint32), ("rgba", ctypes.c_void_p)]


class _TriScene(ctypes.Structure):
    _fields_ = [("triangles", ctypes.c_void_p), ("n_triangles", ctypes.c_uint32),
                ("nodes", ctypes.c_void_p), ("n_nodes", ctypes.c_uint32),
                ("blas", ctypes.c_void_p), ("n_blas", ctypes.c_uint32),
                ("tri_lookup", ctypes.c_void_p), ("n_tri_lookup", ctypes.c_uint32),
                ("blas_lookup", ctypes.c_void_p), ("n_blas_lookup", ctypes.c_uint32),
                ("mesh_tex", _Face)]


class _Hit(ctypes.Structure):
    _fields_ = [("t", ctypes.c_float), ("normal", ctypes.c_float * 3), ("hit", ctypes.c_int)]


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "librt_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        L.rt_oracle_render.restype = ctypes.c_int
        L.rt_oracle_render.argtypes = [fp, fp, ctypes.c_uint32, ctypes.POINTER(_Face), ctypes.c_uint32,
                                       ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
        L.rt_oracle_render_ex.restype = ctypes.c_int
        L.rt_oracle_render_ex.argtypes = [fp, fp, ctypes.c_uint32, ctypes.POINTER(_Face), ctypes.c_uint32,
                                          ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64),
                                          ctypes.c_int]
        L.rt_oracle_hit_sphere.restype = _Hit
        L.rt_oracle_hit_sphere.argtypes = [fp, fp, fp, ctypes.c_float, ctypes.c_float]
        L.rt_oracle_ray_dir.restype = None
        L.rt_oracle_ray_dir.argtypes = [fp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, fp]
        L.rt_oracle_cube_sample.restype = None
        L.rt_oracle_cube_sample.argtypes = [ctypes.POINTER(_Face), fp, fp]
        L.rt_oracle_unorm8.restype = ctypes.c_uint8
        L.rt_oracle_unorm8.argtypes = [ctypes.c_float]
        L.rt_oracle_ray_color.restype = None
        L.rt_oracle_ray_color.argtypes = [fp, fp, ctypes.c_uint32, ctypes.POINTER(_Face), fp, fp, fp,
                                          ctypes.POINTER(ctypes.c_uint64)]
        L.rt_oracle_pixel.restype = None
        L.rt_oracle_pixel.argtypes = [fp, fp, ctypes.c_uint32, ctypes.POINTER(_Face), ctypes.c_uint32, ctypes.c_uint32,
                                      ctypes.c_uint32, ctypes.c_uint32, fp, ctypes.POINTER(ctypes.c_uint64)]
        L.rt_oracle_render_tri.restype = ctypes.c_int
        L.rt_oracle_render_tri.argtypes = [fp, ctypes.POINTER(_TriScene), ctypes.POINTER(_Face), ctypes.c_uint32,
                                           ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
        L.rt_oracle_heatmap_tri.restype = ctypes.c_int
        L.rt_oracle_heatmap_tri.argtypes = [fp, ctypes.POINTER(_TriScene), ctypes.c_uint32, ctypes.c_uint32,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.rt_oracle_max_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _faces(faces):
    keep = [np.ascontiguousarray(f, dtype=np.uint8) for f in faces]
    arr = (_Face * 6)()
    for i, f in enumerate(keep):
        arr[i].w, arr[i].h, arr[i].rgba = f.shape[1], f.shape[0], f.ctypes.data
    return arr, keep


def render(params, spheres, faces, W, H, tile_first=0, tile_step=1, want_float=False, threads=0):
    """Returns (rgba8 (H,W,4) uint8, rgb (H,W,3) float32 or None, rays)."""
    params = np.ascontiguousarray(params, dtype=np.float32)
    spheres = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 8)
    arr, keep = _faces(faces)
    out = np.zeros((H, W, 4), dtype=np.uint8)
    outf = np.zeros((H, W, 3), dtype=np.float32) if want_float else None
    rays = ctypes.c_uint64(0)
    rc = lib().rt_oracle_render(_fp(params), _fp(spheres), spheres.shape[0], arr, W, H, tile_first, tile_step,
                                out.ctypes.data, outf.ctypes.data if want_float else None,
                                ctypes.byref(rays), threads)
    if rc != 0:
        raise RuntimeError("rt_oracle_render failed: %d" % rc)
    return out, outf, rays.value


def render_ray_counts(params, spheres, faces, W, H, threads=0):
    """Per-pixel scene-traversal counts (H,W) uint16 and the RGBA8 frame."""
    params = np.ascontiguousarray(params, dtype=np.float32)
    spheres = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 8)
    arr, keep = _faces(faces)
    out = np.zeros((H, W, 4), dtype=np.uint8)
    cnt = np.zeros((H, W), dtype=np.uint16)
    rays = ctypes.c_uint64(0)
    rc = lib().rt_oracle_render_ex(_fp(params), _fp(spheres), spheres.shape[0], arr, W, H, 0, 1,
                                   out.ctypes.data, None, cnt.ctypes.data, ctypes.byref(rays), threads)
    if rc != 0:
        raise RuntimeError("rt_oracle_render_ex failed: %d" % rc)
    return out, cnt, rays.value


def hit_sphere(origin, direction, sphere, t_min, t_max):
    o = np.ascontiguousarray(origin, dtype=np.float32)
    d = np.ascontiguousarray(direction, dtype=np.float32)
    s = np.ascontiguousarray(sphere, dtype=np.float32)
    h = lib().rt_oracle_hit_sphere(_fp(o), _fp(d), _fp(s), np.float32(t_min), np.float32(t_max))
    return bool(h.hit), np.float32(h.t), np.array(list(h.normal), dtype=np.float32)


def ray_dir(params, W, H, x, y):
    params = np.ascontiguousarray(params, dtype=np.float32)
    d = np.zeros(3, np.float32)
    lib().rt_oracle_ray_dir(_fp(params), W, H, x, y, _fp(d))
    return d


def cube_sample(faces, direction):
    arr, keep = _faces(faces)
    d = np.ascontiguousarray(direction, dtype=np.float32)
    out = np.zeros(3, np.float32)
    lib().rt_oracle_cube_sample(arr, _fp(d), _fp(out))
    return out


def unorm8(c):
    return int(lib().rt_oracle_unorm8(np.float32(c)))


def ray_color(params, spheres, faces, origin, direction):
    params = np.ascontiguousarray(params, dtype=np.float32)
    spheres = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 8)
    arr, keep = _faces(faces)
    o = np.ascontiguousarray(origin, dtype=np.float32)
    d = np.ascontiguousarray(direction, dtype=np.float32)
    out = np.zeros(4, np.float32)
    rays = ctypes.c_uint64(0)
    lib().rt_oracle_ray_color(_fp(params), _fp(spheres), spheres.shape[0], arr, _fp(o), _fp(d), _fp(out),
                              ctypes.byref(rays))
    return out, rays.value


def pixel(params, spheres, faces, W, H, x, y):
    """(rgb float32[3] before quantisation, scene traversals) of one pixel."""
    params = np.ascontiguousarray(params, dtype=np.float32)
    spheres = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 8)
    arr, keep = _faces(faces)
    out = np.zeros(3, np.float32)
    rays = ctypes.c_uint64(0)
    lib().rt_oracle_pixel(_fp(params), _fp(spheres), spheres.shape[0], arr, W, H, x, y, _fp(out), ctypes.byref(rays))
    return out, rays.value


def _tri_scene(buffers):
    """buffers: dict with 'triangles' (n,40), 'nodes' (n,8), 'blas' (n,20), 'tri_lookup', 'blas_lookup'
    float32 arrays in the layouts of RR:169-229 and 'mesh_tex' (h,w,4) uint8."""
    keep = {k: np.ascontiguousarray(buffers[k], dtype=np.float32) for k in
            ("triangles", "nodes", "blas", "tri_lookup", "blas_lookup")}
    tex = np.ascontiguousarray(buffers["mesh_tex"], dtype=np.uint8)
    keep["mesh_tex"] = tex
    t = _TriScene()
    t.triangles, t.n_triangles = keep["triangles"].ctypes.data, keep["triangles"].size // 40
    t.nodes, t.n_nodes = keep["nodes"].ctypes.data, keep["nodes"].size // 8
    t.blas, t.n_blas = keep["blas"].ctypes.data, keep["blas"].size // 20
    t.tri_lookup, t.n_tri_lookup = keep["tri_lookup"].ctypes.data, keep["tri_lookup"].size
    t.blas_lookup, t.n_blas_lookup = keep["blas_lookup"].ctypes.data, keep["blas_lookup"].size
    t.mesh_tex.w, t.mesh_tex.h, t.mesh_tex.rgba = tex.shape[1], tex.shape[0], tex.ctypes.data
    return t, keep


def render_tri(params, buffers, faces, W, H, tile_first=0, tile_step=1, want_float=False, threads=0):
    """RK:73-166 over a triangle scene.  Returns (rgba8, rgb float or None, rays)."""
    params = np.ascontiguousarray(params, dtype=np.float32)
    t, keep = _tri_scene(buffers)
    arr, keepf = _faces(faces)
    out = np.zeros((H, W, 4), dtype=np.uint8)
    outf = np.zeros((H, W, 3), dtype=np.float32) if want_float else None
    rays = ctypes.c_uint64(0)
    rc = lib().rt_oracle_render_tri(_fp(params), ctypes.byref(t), arr, W, H, tile_first, tile_step, out.ctypes.data,
                                    outf.ctypes.data if want_float else None, None, ctypes.byref(rays), threads)
    if rc != 0:
        raise RuntimeError("rt_oracle_render_tri failed: %d" % rc)
    return out, outf, rays.value


def heatmap_tri(params, buffers, W, H, threads=0):
    """HK:63-83.  Returns (rgba8 (H,W,4), raw traversal counts (H,W) uint32)."""
    params = np.ascontiguousarray(params, dtype=np.float32)
    t, keep = _tri_scene(buffers)
    out = np.zeros((H, W, 4), dtype=np.uint8)
    steps = np.zeros((H, W), dtype=np.uint32)
    rc = lib().rt_oracle_heatmap_tri(_fp(params), ctypes.byref(t), W, H, out.ctypes.data, steps.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError("rt_oracle_heatmap_tri failed: %d" % rc)
    return out, steps


def tri_counters():
    """(node loads, triangle tests, instance records) of the triangle path since the last call; clears."""
    out = (ctypes.c_uint64 * 3)()
    L = lib()
    L.rt_oracle_tri_counters.restype = None
    L.rt_oracle_tri_counters.argtypes = [ctypes.c_void_p]
    L.rt_oracle_tri_counters(out)
    return int(out[0]), int(out[1]), int(out[2])


def trace_tri_rays(buffers, origins, dirs, want_tri=False):
    """Nearest-hit t (or -1) of arbitrary rays against a triangle scene (RK:168-244); want_tri: also the index of
    the triangle hit (-1: none)."""
    t, keep = _tri_scene(buffers)
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
    out = np.zeros(o.shape[0], dtype=np.float32)
    L = lib()
    L.rt_oracle_trace_tri_rays.restype = ctypes.c_int
    L.rt_oracle_trace_tri_rays.argtypes = [ctypes.POINTER(_TriScene), ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p]
    tri = np.full(o.shape[0], -1, dtype=np.int32) if want_tri else None
    rc = L.rt_oracle_trace_tri_rays(ctypes.byref(t), o.shape[0], o.ctypes.data, d.ctypes.data, out.ctypes.data,
                                    tri.ctypes.data if want_tri else None)
    if rc != 0:
        raise RuntimeError("rt_oracle_trace_tri_rays failed")
    return (out, tri) if want_tri else out


def max_threads():
    return lib().rt_oracle_max_threads()


def tri_work_px(params, buffers, faces, W, H):
    """Measurement only: (H, W, 4) uint32 per pixel {scene traversals, BLAS inner-node visits, triangle tests,
    instance records read} of the path over a triangle scene (single-threaded)."""
    params = np.ascontiguousarray(params, dtype=np.float32)
    t, keep = _tri_scene(buffers)
    arr, keepf = _faces(faces)
    out = np.zeros((H, W, 4), dtype=np.uint32)
    L = lib()
    L.rt_oracle_tri_work_px.restype = ctypes.c_int
    L.rt_oracle_tri_work_px.argtypes = [ctypes.c_void_p, ctypes.POINTER(_TriScene), ctypes.POINTER(_Face),
                                        ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    rc = L.rt_oracle_tri_work_px(_fp(params), ctypes.byref(t), arr, W, H, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("rt_oracle_tri_work_px failed: %d" % rc)
    return out


def tri_trace_px(params, buffers, faces, W, H):
    """Measurement only: (codes uint8 array, offsets uint64 array of W*H+1) -- the step sequence of every pixel's path."""
    params = np.ascontiguousarray(params, dtype=np.float32)
    t, keep = _tri_scene(buffers)
    arr, keepf = _faces(faces)
    L = lib()
    L.rt_oracle_tri_trace_px.restype = ctypes.c_uint64
    L.rt_oracle_tri_trace_px.argtypes = [ctypes.c_void_p, ctypes.POINTER(_TriScene), ctypes.POINTER(_Face), ctypes.c_uint32,
                                         ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    offs = np.zeros(W * H + 1, dtype=np.uint64)
    need = L.rt_oracle_tri_trace_px(_fp(params), ctypes.byref(t), arr, W, H, None, 0, offs.ctypes.data)
    codes = np.zeros(int(need), dtype=np.uint8)
    L.rt_oracle_tri_trace_px(_fp(params), ctypes.byref(t), arr, W, H, codes.ctypes.data, need, offs.ctypes.data)
    return codes, offs
