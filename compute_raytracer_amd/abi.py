"""ctypes binding of include/rt355.h (librt355.so).  This is the only way the Python host
reaches the device; there is deliberately no CPU fallback: if the HIP library is missing or
no gfx950 device is present, the calls raise."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT355_LIB") or os.path.join(_HERE, "librt355.so")   # RT355_LIB: A/B builds (tools/)

RT_OK = 0
RT_ERR_INVALID_ARG = -1
RT_ERR_NO_DEVICE = -2
RT_ERR_HIP = -3
RT_ERR_UNSUPPORTED = -4
RT_ERR_STATE = -5
RT_ERR_CAPACITY = -6
RT_ERR_COMM = -7
RT355_COMM_ID_BYTES = 128

RT_KERNEL_RAYTRACER = 0
RT_KERNEL_HEATMAP = 1
RT_MODE_FAST = 0
RT_MODE_STRICT = 1

# every symbol include/rt355.h declares (tests check the library exports each of them)
SYMBOLS = [
    "rt_create", "rt_destroy", "rt_last_error", "rt_abi_version", "rt_resize", "rt_write_params",
    "rt_write_spheres", "rt_write_cubemap_face", "rt_write_triangles", "rt_write_nodes", "rt_write_blas",
    "rt_write_tri_lookup", "rt_write_blas_lookup", "rt_write_mesh_texture", "rt_select_kernel", "rt_set_mode",
    "rt_set_variant", "rt_set_partition", "rt_tiles_of_rank", "rt_padded_tiles", "rt_render", "rt_wait",
    "rt_read_pixels", "rt_get_stats", "rt_render_to", "rt_assemble_frame", "rt_device_pixels",
    "rt_build_hierarchy", "rt_filter_plan", "rt_build_flow", "rt_order_tiles",
    "rt_comm_unique_id", "rt_comm_init", "rt_comm_destroy", "rt_render_gather", "rt_frame_pixels", "rt_read_frame",
    "rt_group_create", "rt_group_destroy", "rt_group_size", "rt_group_ctx", "rt_group_render", "rt_group_wait",
    "rt_build_id", "rt_kernel_name", "rt_set_comm_timeout",
    "rt_read_pixels_async", "rt_read_pixels_wait", "rt_host_alloc", "rt_host_free",
]

# rt_kernel_id (include/rt355.h): which kernel form rendered a frame
KERNEL_IDS = {0: "none", 1: "literal", 2: "brute_single", 3: "brute_pipeline", 4: "hierarchy_8", 5: "hierarchy_12",
              6: "hierarchy_16", 7: "hierarchy_global", 8: "triangles", 9: "heatmap", 10: "triangles_roles"}


class RtStats(ctypes.Structure):
    _fields_ = [
        ("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("local_tiles", ctypes.c_uint32),
        ("spheres", ctypes.c_uint32), ("rays", ctypes.c_uint64), ("kernel_ms", ctypes.c_float),
        ("prep_ms", ctypes.c_float), ("frames", ctypes.c_uint32), ("mode", ctypes.c_int),
        ("batch_frames", ctypes.c_uint32), ("batch_kernel_ms", ctypes.c_float),
        ("gather_ms", ctypes.c_float), ("batch_gather_ms", ctypes.c_float),
        ("kernel_id", ctypes.c_uint32), ("grid_share", ctypes.c_uint32), ("instance_uploads", ctypes.c_uint32),
        ("tri_form", ctypes.c_uint32), ("pair_rebuilds", ctypes.c_uint32),
    ]


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rt355 error %d: %s" % (code, message))
        self.code = code


_lib = None


def _one_hip_runtime():
    """A PyTorch-ROCm wheel bundles its own libamdhip64.so.7; librt355.so is linked against the
    system one (/opt/rocm).  Both have the same SONAME, so whichever is loaded first serves the
    whole process -- and torch does not find its GPUs on the system copy.  When torch is installed
    but not imported yet, load ITS runtime first (without importing torch), so that a later
    `import torch` (bench.py, the multi-GPU path) and this library share one HIP runtime."""
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass


def load():
    """Loads librt355.so (built by `make lib` / __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(RT_ERR_NO_DEVICE, "librt355.so is not built (%s); run `make lib`. There is no CPU path." % LIB_PATH)
    # frames in flight run on four streams; HIP's default of 4 hardware queues makes streams share a
    # queue as soon as the host has a few of its own (effective only if HIP is not initialised yet)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    _one_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    vp, u32, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t
    fp = ctypes.POINTER(ctypes.c_float)
    sig = {
        "rt_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(vp)]),
        "rt_destroy": (ctypes.c_int, [vp]),
        "rt_last_error": (ctypes.c_char_p, [vp]),
        "rt_abi_version": (ctypes.c_int, []),
        "rt_resize": (ctypes.c_int, [vp, u32, u32]),
        "rt_write_params": (ctypes.c_int, [vp, fp]),
        "rt_write_spheres": (ctypes.c_int, [vp, fp, u32]),
        "rt_write_cubemap_face": (ctypes.c_int, [vp, ctypes.c_int, u32, u32, vp]),
        "rt_write_triangles": (ctypes.c_int, [vp, fp, u32]),
        "rt_write_nodes": (ctypes.c_int, [vp, sz, fp, u32]),
        "rt_write_blas": (ctypes.c_int, [vp, fp, u32]),
        "rt_write_tri_lookup": (ctypes.c_int, [vp, fp, u32]),
        "rt_write_blas_lookup": (ctypes.c_int, [vp, fp, u32]),
        "rt_write_mesh_texture": (ctypes.c_int, [vp, u32, u32, vp]),
        "rt_select_kernel": (ctypes.c_int, [vp, ctypes.c_int]),
        "rt_set_mode": (ctypes.c_int, [vp, ctypes.c_int]),
        "rt_set_variant": (ctypes.c_int, [vp, ctypes.c_int]),
        "rt_set_partition": (ctypes.c_int, [vp, u32, u32]),
        "rt_tiles_of_rank": (u32, [u32, u32, u32]),
        "rt_padded_tiles": (u32, [u32, u32]),
        "rt_render": (ctypes.c_int, [vp]),
        "rt_wait": (ctypes.c_int, [vp]),
        "rt_read_pixels": (ctypes.c_int, [vp, vp, sz]),
        "rt_get_stats": (ctypes.c_int, [vp, ctypes.POINTER(RtStats)]),
        "rt_render_to": (ctypes.c_int, [vp, vp, sz, vp]),
        "rt_assemble_frame": (ctypes.c_int, [vp, vp, vp, u32, vp]),
        "rt_device_pixels": (ctypes.c_int, [vp, ctypes.POINTER(vp), ctypes.POINTER(sz)]),
        "rt_build_hierarchy": (ctypes.c_int, [fp, u32, fp, ctypes.POINTER(u32), u32, ctypes.POINTER(u32)]),
        "rt_filter_plan": (ctypes.c_int, [fp, u32, fp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
        "rt_order_tiles": (ctypes.c_int, [vp, ctypes.POINTER(u32), u32, u32, ctypes.POINTER(u32), sz]),
        "rt_comm_unique_id": (ctypes.c_int, [vp]),
        "rt_comm_init": (ctypes.c_int, [vp, vp, u32, u32]),
        "rt_comm_destroy": (ctypes.c_int, [vp]),
        "rt_render_gather": (ctypes.c_int, [vp, ctypes.c_int]),
        "rt_frame_pixels": (ctypes.c_int, [vp, ctypes.POINTER(vp), ctypes.POINTER(sz)]),
        "rt_read_frame": (ctypes.c_int, [vp, vp, sz]),
        "rt_group_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(vp)]),
        "rt_group_destroy": (ctypes.c_int, [vp]),
        "rt_group_size": (ctypes.c_int, [vp]),
        "rt_group_ctx": (vp, [vp, ctypes.c_int]),
        "rt_group_render": (ctypes.c_int, [vp, ctypes.c_int]),
        "rt_group_wait": (ctypes.c_int, [vp]),
        "rt_build_id": (ctypes.c_char_p, []),
        "rt_kernel_name": (ctypes.c_char_p, [ctypes.c_int]),
        "rt_set_comm_timeout": (ctypes.c_int, [vp, u32]),
        "rt_read_pixels_async": (ctypes.c_int, [vp, u32, vp, sz]),
        "rt_read_pixels_wait": (ctypes.c_int, [vp]),
        "rt_host_alloc": (ctypes.c_int, [sz, ctypes.POINTER(vp)]),
        "rt_host_free": (ctypes.c_int, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc, ctx=None):
    if rc != RT_OK:
        msg = load().rt_last_error(ctx)
        raise RtError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc
