"""RendererRaytracing -- host-side mirror of src/rendering-raycast/renderer-raytracing.ts.

Same surface as the reference class (RR:53-76, RR:434): construct with (width, height, scene)
-- the canvas argument is dropped, there is no canvas in a headless renderer --, `initialize()`,
`render()`, `showRaytracer()`, `showHeatmap()`.  Every WebGPU call of the reference is replaced
by the C-ABI call include/rt355.h lists next to it.  The frame is the rgba8unorm colour buffer
itself (RR:102-109); the blit to the canvas (RR:449-463) has no counterpart.

No CPU fallback: without librt355.so and a gfx950 device, `initialize()` raises.
"""
import ctypes

import numpy as np

from . import abi
from .cubemap_material import CubemapMaterial
from .scene_raytracing import CONSTANT_SKY_RGBA


class RendererRaytracing:
    def __init__(self, width, height, scene, device=0, maxBounces=4, rank=0, world=1):
        self.scene = scene                     # RR:54
        self.width = int(width)                # RR:56-57
        self.height = int(height)
        self.device = int(device)
        # RR:157 hard-codes maxBounces = 4; it already travels in the uniform (RK:10), so the
        # BASELINE configs simply set it.
        self.maxBounces = maxBounces
        self.rank, self.world = int(rank), int(world)
        self.skyboxMaterial = None             # RR:33
        self.meshMaterial = None               # RR:32 mouseyMaterial (the mesh texture, binding 8)
        self.loaded = False                    # RR:51
        self.render_time_ms = None             # the 'render-time' label of RR:468-469
        self._ctx = None
        self._lib = None
        self._pinned = []

    # ---- RR:62-68 -------------------------------------------------------------------------
    def initialize(self, skybox=None, meshMaterial=None):
        self._lib = abi.load()
        ctx = ctypes.c_void_p()
        abi.check(self._lib.rt_create(self.device, ctypes.byref(ctx)))           # RR:78-97 setupDevice
        self._ctx = ctx
        self.meshMaterial = meshMaterial
        self._create_assets(skybox)                                              # RR:99-153
        self.showRaytracer()                                                     # RR:356-365
        return self

    def _create_assets(self, skybox):
        L, c = self._lib, self._ctx
        self.skyboxMaterial = skybox if skybox is not None else CubemapMaterial.constant(CONSTANT_SKY_RGBA)
        for i, face in enumerate(self.skyboxMaterial.faces):                     # CM:73-77
            f = np.ascontiguousarray(face, dtype=np.uint8)
            abi.check(L.rt_write_cubemap_face(c, i, f.shape[1], f.shape[0], f.ctypes.data), c)
        abi.check(L.rt_set_partition(c, self.rank, self.world), c)
        abi.check(L.rt_resize(c, self.width, self.height), c)                    # RR:102-109 colorBuffer

    def showRaytracer(self):                                                     # RR:70-72
        abi.check(self._lib.rt_select_kernel(self._ctx, abi.RT_KERNEL_RAYTRACER), self._ctx)

    def showHeatmap(self):                                                       # RR:74-76
        abi.check(self._lib.rt_select_kernel(self._ctx, abi.RT_KERNEL_HEATMAP), self._ctx)

    def set_mode(self, strict):
        abi.check(self._lib.rt_set_mode(self._ctx, abi.RT_MODE_STRICT if strict else abi.RT_MODE_FAST), self._ctx)

    def set_variant(self, variant):
        abi.check(self._lib.rt_set_variant(self._ctx, int(variant)), self._ctx)

    # ---- RR:155-230 -----------------------------------------------------------------------
    def recalculateScene(self):
        L, c = self._lib, self._ctx
        p = self.scene.pack_params(self.maxBounces)                              # RR:157-165
        abi.check(L.rt_write_params(c, p.ctypes.data_as(ctypes.POINTER(ctypes.c_float))), c)
        fp = ctypes.POINTER(ctypes.c_float)
        if self.scene.hasTriangles:                                                 # the reference's live scene type
            b = np.ascontiguousarray(self.scene.pack_blas(), dtype=np.float32)           # RR:169-174
            abi.check(L.rt_write_blas(c, b.ctypes.data_as(fp), b.shape[0]), c)
            bl = np.ascontiguousarray(self.scene.pack_blas_lookup(), dtype=np.float32)   # RR:177-181
            abi.check(L.rt_write_blas_lookup(c, bl.ctypes.data_as(fp), bl.shape[0]), c)
            na = np.ascontiguousarray(self.scene.pack_tlas_nodes(), dtype=np.float32)    # RR:184-192
            abi.check(L.rt_write_nodes(c, 0, na.ctypes.data_as(fp), na.shape[0]), c)
        if self.loaded:                                                          # RR:194-195
            return
        self.loaded = True
        if self.scene.hasTriangles:
            t = np.ascontiguousarray(self.scene.pack_triangles(), dtype=np.float32)      # RR:198-209
            abi.check(L.rt_write_triangles(c, t.ctypes.data_as(fp), t.shape[0]), c)
            nb = np.ascontiguousarray(self.scene.pack_blas_nodes(), dtype=np.float32)    # RR:212-223
            abi.check(L.rt_write_nodes(c, 32 * self.scene.tlasNodesMax, nb.ctypes.data_as(fp), nb.shape[0]), c)
            tl = np.ascontiguousarray(self.scene.pack_tri_lookup(), dtype=np.float32)    # RR:225-229
            abi.check(L.rt_write_tri_lookup(c, tl.ctypes.data_as(fp), tl.shape[0]), c)
            if self.meshMaterial is not None:                                            # RR:113-114 mouseyMaterial
                img = np.ascontiguousarray(self.meshMaterial.image, dtype=np.uint8)
                abi.check(L.rt_write_mesh_texture(c, img.shape[1], img.shape[0], img.ctypes.data), c)
            return
        s = np.ascontiguousarray(self.scene.pack_spheres(), dtype=np.float32)    # in place of RR:198-229
        abi.check(L.rt_write_spheres(c, s.ctypes.data_as(fp), s.shape[0]), c)

    # ---- RR:434-470 -----------------------------------------------------------------------
    def render(self):
        import time
        t0 = time.perf_counter()                                                 # RR:435
        self.recalculateScene()                                                  # RR:437
        abi.check(self._lib.rt_render(self._ctx), self._ctx)                     # RR:442-446, 465
        abi.check(self._lib.rt_wait(self._ctx), self._ctx)                       # RR:467
        self.render_time_ms = (time.perf_counter() - t0) * 1e3                   # RR:468-469

    def enqueue(self):
        """render() without the wait (the reference never does this; bench.py uses it to time
        K frames back to back)."""
        abi.check(self._lib.rt_render(self._ctx), self._ctx)

    def wait(self):
        abi.check(self._lib.rt_wait(self._ctx), self._ctx)

    # ---- results ----------------------------------------------------------------------------
    def local_rows(self):
        rows = 0
        for j in range(abi.load().rt_tiles_of_rank(self.height, self.rank, self.world)):
            y0 = (self.rank + j * self.world) * 8
            rows += min(8, self.height - y0)
        return rows

    def read_pixels(self):
        """The rows this rank rendered, (rows, W, 4) uint8; world == 1: the whole frame."""
        rows = self.local_rows()
        out = np.empty((rows, self.width, 4), dtype=np.uint8)
        abi.check(self._lib.rt_read_pixels(self._ctx, out.ctypes.data, out.nbytes), self._ctx)
        return out

    # ---- streaming read-back: frames in flight AND copied out (rt_read_pixels_async) -----------
    def host_frames(self, n):
        """n pinned (H, W, 4) uint8 frames for read_pixels_async (freed by close())."""
        import weakref
        nbytes = self.height * self.width * 4
        out = []
        for _ in range(n):
            p = ctypes.c_void_p()
            abi.check(self._lib.rt_host_alloc(nbytes, ctypes.byref(p)))
            buf = (ctypes.c_uint8 * nbytes).from_address(p.value)
            # the pinned memory lives as long as anything refers to it -- the arrays handed out (numpy keeps `buf` as their
            # base) or this renderer -- and is freed when the last reference goes, not at close(): a view that outlives the
            # renderer stays valid
            weakref.finalize(buf, self._lib.rt_host_free, ctypes.c_void_p(p.value))
            self._pinned.append(buf)
            out.append(np.frombuffer(buf, dtype=np.uint8).reshape(self.height, self.width, 4))
        return out

    def read_pixels_async(self, frames_back, dst):
        abi.check(self._lib.rt_read_pixels_async(self._ctx, int(frames_back), dst.ctypes.data, dst.nbytes), self._ctx)

    def read_pixels_wait(self):
        abi.check(self._lib.rt_read_pixels_wait(self._ctx), self._ctx)

    def stats(self):
        st = abi.RtStats()
        abi.check(self._lib.rt_get_stats(self._ctx, ctypes.byref(st)), self._ctx)
        return {k: getattr(st, k) for k, _ in abi.RtStats._fields_}

    # ---- device-pointer interop for the process-per-GPU path --------------------------------
    def render_to(self, device_ptr, nbytes, stream_ptr=None):
        self.recalculateScene()
        abi.check(self._lib.rt_render_to(self._ctx, ctypes.c_void_p(device_ptr), nbytes,
                                         ctypes.c_void_p(stream_ptr) if stream_ptr else None), self._ctx)

    def assemble_frame(self, gathered_ptr, frame_ptr, world, stream_ptr=None):
        abi.check(self._lib.rt_assemble_frame(self._ctx, ctypes.c_void_p(gathered_ptr), ctypes.c_void_p(frame_ptr),
                                              world, ctypes.c_void_p(stream_ptr) if stream_ptr else None), self._ctx)

    # ---- multi-GPU behind the C ABI: render + RCCL gather + de-interleave in one call ---------------
    @staticmethod
    def comm_unique_id():
        """rank 0: the 128 bytes every rank passes to comm_init (hand them over by any side channel)."""
        buf = ctypes.create_string_buffer(abi.RT355_COMM_ID_BYTES)
        abi.check(abi.load().rt_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        """Collective over the group (ncclCommInitRank on this context's device); fixes the partition."""
        if len(unique_id) != abi.RT355_COMM_ID_BYTES:
            raise ValueError("comm_init: the unique id is %d bytes" % abi.RT355_COMM_ID_BYTES)
        buf = ctypes.create_string_buffer(bytes(unique_id), abi.RT355_COMM_ID_BYTES)
        abi.check(self._lib.rt_comm_init(self._ctx, buf, rank, world), self._ctx)
        self.rank, self.world = int(rank), int(world)

    def render_gather(self, root=0, wait=False):
        """RR:434-470 across the group: this rank's tiles, the RCCL exchange and the de-interleave are
        enqueued by ONE library call.  root = -1: every rank receives the frame."""
        self.recalculateScene()
        abi.check(self._lib.rt_render_gather(self._ctx, int(root)), self._ctx)
        if wait:
            abi.check(self._lib.rt_wait(self._ctx), self._ctx)

    def enqueue_gather(self, root=0):
        """render_gather() without recalculateScene and without the wait -- the counterpart of enqueue() for a group (bench.py
        times K frames of a scene that is resident before the timed region, with one rank or with many)."""
        abi.check(self._lib.rt_render_gather(self._ctx, int(root)), self._ctx)

    def read_frame(self):
        """The whole W x H frame of the latest render_gather (on a rank that received it)."""
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        abi.check(self._lib.rt_read_frame(self._ctx, out.ctypes.data, out.nbytes), self._ctx)
        return out

    def close(self):
        """Destroys the context.  The pinned frames of host_frames() stay valid for as long as an array refers to them."""
        if self._ctx is not None:
            self._lib.rt_destroy(self._ctx)      # waits for every copy that was begun
            self._ctx = None
            self._pinned = []                    # the renderer's own references; the memory goes with the last view

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
