/*
 * rt355.h -- C ABI of librt355.so, the MI355X (gfx950) stand-in for the WebGPU
 * calls the reference's RendererRaytracing makes.
 *
 * Every entry point names the reference interface it replaces; citations are
 * relative to the reference repository:
 *   RR = src/rendering-raycast/renderer-raytracing.ts
 *   RK = src/rendering-raycast/shaders/raytracer-kernel.wgsl
 *   CM = src/material/cubemap-material.ts
 *   MT = src/material/material.ts
 *
 * Conventions
 *   - plain C: pointers and sizes only; no C++/torch/HIP types in any signature
 *     (a HIP stream crosses as void*).
 *   - every function returns int: RT_OK (0) or a negative rt_status.  The message
 *     for the most recent failure on the calling thread is rt_last_error().
 *   - every rt_write_* copies out of caller memory before it returns, as
 *     GPUQueue.writeBuffer does (RR:165); the caller may reuse/free at once.
 *   - byte layouts are exactly the ones RR writes (f32 indices and counts,
 *     vec3 padded to 16 B), so buffers captured from a browser run replay unchanged.
 *   - a context is bound to ONE device and is not thread-safe (single JS thread in
 *     the reference).  Multi-GPU: the frame is split into 8-row tiles (one WGSL
 *     workgroup row, RK:73), tile t is rendered by rank t % world, and the LIBRARY
 *     moves the compact per-rank tile buffers over RCCL (xGMI) and de-interleaves
 *     them: rt_comm_init + rt_render_gather (one process per GPU) or rt_group_create +
 *     rt_group_render (one process, all GPUs).  rt_set_partition / rt_render_to /
 *     rt_assemble_frame remain for hosts that bring their own transport.
 *   - there is no CPU fallback anywhere in this library.
 */
#ifndef RT355_H
#define RT355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT355_ABI_VERSION 4

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = -1,  /* null pointer, bad size, bad enum                      */
    RT_ERR_NO_DEVICE = -2,    /* no gfx950 device / ordinal out of range                */
    RT_ERR_HIP = -3,          /* a HIP runtime call failed (message has the hipError)   */
    RT_ERR_UNSUPPORTED = -4,  /* combination not supported (e.g. heatmap of a sphere scene) */
    RT_ERR_STATE = -5,        /* call order: e.g. render before resize / write_params   */
    RT_ERR_CAPACITY = -6,     /* destination buffer too small                           */
    RT_ERR_COMM = -7          /* an RCCL call failed, a peer reported an asynchronous error, or the
                                 exchange did not complete within rt_set_comm_timeout: the communicator was
                                 aborted and stays unusable until rt_comm_destroy / rt_group_destroy        */
} rt_status;

/* Which kernel form rendered a frame (rt_stats.kernel_id): the library chooses by scene type, sphere
 * count, LDS footprint and mode; callers that price a frame (bench.py) read it from here. */
typedef enum rt_kernel_id {
    RT_KID_NONE = 0,
    RT_KID_LITERAL = 1,          /* trace_pixels, the reference's loop as written (strict mode, or a scene outside the filter's range) */
    RT_KID_BRUTE_SINGLE = 2,     /* trace_pixels with the FMA filter: one kernel, every sphere tested */
    RT_KID_BRUTE_PIPELINE = 3,   /* first_bounce + trace_paths over a path queue */
    RT_KID_HIERARCHY_8 = 4,      /* bvh_pixels, 8-wave workgroups, three per CU, nodes in LDS */
    RT_KID_HIERARCHY_12 = 5,     /* 12-wave workgroups, two per CU */
    RT_KID_HIERARCHY_16 = 6,     /* 16-wave workgroups, one per CU */
    RT_KID_HIERARCHY_GLOBAL = 7, /* nodes read from global memory (scenes beyond a CU's LDS) */
    RT_KID_TRIANGLES = 8,        /* trace_triangles (TLAS / BLAS traversal) */
    RT_KID_HEATMAP = 9,          /* heatmap_triangles */
    RT_KID_TRIANGLES_ROLES = 10  /* trace_roles: an awaited frame whose work list splits tiles -- the idle lanes of a part walk the next reflection ray while its pixels' lanes walk the shadow ray */
    /* (10, 11: the persistent triangle kernels of ABI 3 -- measured slower on every configuration, removed in ABI 4;
       docs/experiments.md keeps the account) */
} rt_kernel_id;

typedef enum rt_kernel {
    RT_KERNEL_RAYTRACER = 0,  /* RR:70-72 showRaytracer()  */
    RT_KERNEL_HEATMAP = 1     /* RR:74-76 showHeatmap(): traversal-cost visualiser, triangle scenes only */
} rt_kernel;

/* Arithmetic mode of the ray-trace kernel.
 * RT_MODE_STRICT: no FMA contraction, IEEE div/sqrt, the oracle's operation order ->
 *                 expected bit-identical to oracle/rt_oracle.c.
 * RT_MODE_FAST  : FMA contraction allowed (default); parity within the tolerance stated
 *                 in tests/test_parity_gpu.py. */
typedef enum rt_mode { RT_MODE_FAST = 0, RT_MODE_STRICT = 1 } rt_mode;

typedef struct rt_ctx rt_ctx;

typedef struct rt_stats {
    uint32_t width, height;       /* current target                                        */
    uint32_t local_tiles;         /* 8-row tiles this context renders                      */
    uint32_t spheres;             /* primitives in the scene                               */
    uint64_t rays;                /* scene traversals of the last completed render:
                                     primary/reflection (RK:114) + shadow (RK:153) rays    */
    float kernel_ms;              /* hipEvent time of the last ray-trace kernel launch     */
    float prep_ms;                /* hipEvent time of the last per-frame scene preparation  */
    uint32_t frames;              /* completed renders since rt_create                     */
    int mode;                     /* rt_mode in effect                                     */
    uint32_t batch_frames;        /* renders completed by the last rt_wait                  */
    float batch_kernel_ms;        /* sum of their ray-trace kernel times (hipEvents on the
                                     stream each kernel was launched on)                    */
    float gather_ms;              /* rt_render_gather / rt_group_render: RCCL exchange + de-interleave of the
                                     last frame (hipEvents, same stream); kernel_ms then is the render alone */
    float batch_gather_ms;        /* ... summed over the frames the last rt_wait completed   */
    uint32_t kernel_id;           /* rt_kernel_id of the latest frame enqueued                */
    uint32_t grid_share;          /* that frame's share of the chip: 1 = all resident slots, k = 1/k of them
                                     (frames on k distinct streams in flight)                 */
    uint32_t instance_uploads;    /* frames whose per-frame instance data (rt_write_blas / _blas_lookup /
                                     _nodes at offset 0) travelled with the frame, without a drain */
    uint32_t tri_form;            /* triangle scenes: the stack form of the latest frame's kernel -- 0: twenty top-level slots,
                                     1: four (a top-level tree of depth <= 4 within 16 nodes), five waves per SIMD, 2: three
                                     (depth <= 3, 8 nodes, <= 4 instances), six waves per SIMD, 3: eight (depth <= 8, 24 nodes),
                                     five waves, 4: eight with sixteen staged instances (13-16 instances or 32 nodes), five waves
                                      */
    uint32_t pair_rebuilds;       /* times the library rebuilt its relinked copy of the BLAS trees (a drain + an upload:
                                     a node write reached the trees, or a frame named a root the copy did not know) */
} rt_stats;

/* ---- lifetime ---------------------------------------------------------------------- */

/* Replaces navigator.gpu.requestAdapter()/requestDevice() (RR:82-86).  `device` is the HIP
 * ordinal; its streams are created on it.  Fails (RT_ERR_NO_DEVICE) when no GPU is present. */
int rt_create(int device, rt_ctx** out);
int rt_destroy(rt_ctx* ctx);

/* Thread-local message of the last failure ("" if none).  `ctx` may be NULL. */
const char* rt_last_error(rt_ctx* ctx);

int rt_abi_version(void);

/* sha256 (hex, first 16 digits) of the library's sources at build time: profiles taken with one build are
 * not evidence for another (bench.py matches it against profiles/traffic.json). */
const char* rt_build_id(void);
/* Name of a kernel form, e.g. "bvh_pixels<8>" (static storage). */
const char* rt_kernel_name(int kernel_id);

/* ---- resources ----------------------------------------------------------------------- */

/* Replaces createTexture({size:{width,height}, format:'rgba8unorm'}) (RR:102-109): the
 * W x H x 4 B colour buffer, row-major, row 0 = top.  Re-callable.  1 <= W, H <= 65536 and
 * fewer than 2^31 pixels (padded to 8x8 tiles). */
int rt_resize(rt_ctx* ctx, uint32_t width, uint32_t height);

/* Replaces queue.writeBuffer(sceneParameters, 0, Float32Array(24)) (RR:157-165).
 * p[0..2] cameraPos, [4..6] forwards, [8..10] right, [12..14] up, [16..18] lightPosition,
 * [19] lightIntensity, [20] minIntensity, [21] maxBounces (f32, truncated by u32(), RK:110). */
int rt_write_params(rt_ctx* ctx, const float params[24]);

/* Sphere primitives: `n` records of 8 f32 {cx,cy,cz,_, r,g,b, radius} = the WGSL layout of
 * the reference's commented `struct Sphere` (RK:13-17; model/sphere.ts:1-10).  Takes the
 * place of the triangle/BVH uploads (RR:198-229) for sphere scenes.  n may be 0. */
int rt_write_spheres(rt_ctx* ctx, const float* records, uint32_t n);

/* Replaces copyExternalImageToTexture(face i) (CM:73-77).  face order as CM:40-47:
 * 0 +X, 1 -X, 2 +Y, 3 -Y, 4 +Z, 5 -Z; rgba8unorm, w*h*4 bytes, row 0 = top.  Six equal squares
 * are a WebGPU cube texture and filter seamlessly across edges; other image sets clamp per image.
 * Memory: a sphere scene rendered through the hierarchy kernel under a sky that is not one colour
 * keeps 32 bytes per local pixel of end-of-path records in each of four frame slots (the sky is
 * sampled by a second kernel): 1.06 GB per slot at 7680x4320, allocated when a
 * frame first needs them. */
int rt_write_cubemap_face(rt_ctx* ctx, int face, uint32_t w, uint32_t h, const uint8_t* rgba);

/* The reference's live scene type: triangles behind a two-level BVH (RK:168-410).  Same byte
 * layouts as RR writes: 160-B triangles (RR:198-209), 32-B nodes {min.xyz, leftChildIndex,
 * max.xyz, primitiveCount} (indices and counts as f32), 80-B BLAS records {inverseModel
 * column-major, rootNodeIndex, pad}, f32 lookup tables.  rt_write_nodes takes a byte offset like
 * queue.writeBuffer(nodeBuffer, offset, ...): the TLAS nodes are rewritten every frame at offset
 * 0 (RR:184-192), the BLAS nodes once at 32*tlasNodesMax (RR:212-223).  Writing triangles or
 * nodes switches the context to the triangle scene; rt_write_spheres switches back. */
/* Per-frame instance data.  The reference rewrites the BLAS records, the BLAS lookup and the TLAS nodes
 * before EVERY frame (RR:169-192; scene-raytracing.ts:138-143).  Those three writes -- rt_write_blas and
 * rt_write_blas_lookup of up to 16 instances, rt_write_nodes inside the first 31 nodes -- do not wait for
 * the frames in flight: the library keeps their current contents on the host, and the next frame carries
 * them to the device itself (in the kernarg block of a one-workgroup kernel, in front of the ray-trace
 * kernel on the frame's stream, into one of four versions of the three buffers -- a frame in flight keeps
 * reading the version it was enqueued with).  Larger instance sets, and every other rt_write_*, drain. */
int rt_write_triangles(rt_ctx* ctx, const float* data, uint32_t n_triangles);        /* RR:198-209 */
int rt_write_nodes(rt_ctx* ctx, size_t byte_offset, const float* data, uint32_t n);  /* RR:184-192, 212-223 */
int rt_write_blas(rt_ctx* ctx, const float* data, uint32_t n_blas);                  /* RR:169-174 */
int rt_write_tri_lookup(rt_ctx* ctx, const float* data, uint32_t n);                 /* RR:225-229 */
int rt_write_blas_lookup(rt_ctx* ctx, const float* data, uint32_t n);                /* RR:177-181 */
int rt_write_mesh_texture(rt_ctx* ctx, uint32_t w, uint32_t h, const uint8_t* rgba); /* MT:61-65 */

/* Replaces selecting one of the two compute pipelines (RR:70-76, RR:356-374). */
int rt_select_kernel(rt_ctx* ctx, int kernel);

/* Arithmetic mode (no reference counterpart: WGSL leaves fp contraction to the driver). */
int rt_set_mode(rt_ctx* ctx, int mode);

/* Kernel variant for A/B measurements (0 = library default).  Fast-mode sphere scenes:
 * 0 = bounding-sphere hierarchy from 128 spheres on (from 72 on once the caller keeps frames in flight), single
 * brute-force kernel below;
 * 4 = hierarchy for any sphere count; 5 = brute force (two-kernel pipeline from 320 spheres on);
 * 1, 2, 3 = individual brute-force forms.  Triangle scenes: 0 = one workgroup per tile (rt_triangles.hip), reading the BLAS
 * trees from the library's relinked pair records where the scene fits them (up to 12 instances, node buffer and lookup table
 * within 16-bit indices); 6 = the same kernel on the reference's node buffer only.  Every variant produces the same pixels.
 * See DESIGN.md. */
int rt_set_variant(rt_ctx* ctx, int variant);

/* ---- multi-GPU partition --------------------------------------------------------------- */

/* This context renders the 8-row tiles t with t % world == rank, into a compact buffer of
 * rt_local_tiles() * 8 rows.  Default rank 0, world 1 = the whole frame. */
int rt_set_partition(rt_ctx* ctx, uint32_t rank, uint32_t world);

/* Tiles owned by `rank`, and the padded per-rank tile count (max over ranks) that sizes the
 * all-gather message: bytes = rt_padded_tiles * 8 * width * 4. */
uint32_t rt_tiles_of_rank(uint32_t height, uint32_t rank, uint32_t world);
uint32_t rt_padded_tiles(uint32_t height, uint32_t world);

/* ---- frame ------------------------------------------------------------------------------- */

/* Replaces beginComputePass/setPipeline/setBindGroup/dispatchWorkgroups(ceil(W/8),
 * ceil(H/8),1)/submit (RR:442-446, RR:465): enqueues scene preparation + the ray-trace
 * kernel and returns without waiting.  The reference keeps one frame in flight (RR:467) --
 * rt_render + rt_wait per frame does the same.  A caller that enqueues frames back to back
 * gets up to four of them running concurrently (the library rotates over four streams
 * and four colour buffers, each concurrent frame taking a share of the chip): the
 * dependent-ray tail of one frame then runs beside the bulk of the next.  Every frame uses
 * the parameters and scene written before its rt_render call; rt_read_pixels returns the
 * frame of the latest rt_render.  Up to RT355_MAX_IN_FLIGHT frames may be enqueued between
 * two rt_wait; beyond that the library drains by itself.  Scene-setup calls (rt_write_spheres,
 * rt_write_cubemap_face, rt_write_triangles ..., rt_resize, rt_set_partition) wait for the
 * frames in flight first; rt_write_params does not need to. */
#define RT355_MAX_IN_FLIGHT 64
int rt_render(rt_ctx* ctx);

/* Replaces `await queue.onSubmittedWorkDone()` (RR:467). */
int rt_wait(rt_ctx* ctx);

/* Copies this context's tiles (world 1: the W*H*4 frame) to host memory.
 * Needs cap >= local_tiles*8*W*4 clipped to the frame. */
int rt_read_pixels(rt_ctx* ctx, uint8_t* dst, size_t cap);

/* Streaming read-back: frames kept in flight AND copied out.  rt_read_pixels returns the latest frame only and waits;
 * a host that wants every frame of a pipelined sequence begins an asynchronous copy per frame instead: the frame
 * `frames_back` rt_render calls ago (0 = the latest, at most 3 -- the library rotates over four colour buffers) is copied
 * to `dst` on a copy stream of the library's own, behind that frame's kernels and beside the rendering of the frames
 * after it; a later frame that would overwrite the colour buffer waits for the copy, nothing else does.  `dst` should be
 * pinned memory (rt_host_alloc) -- with pageable memory the copy is staged and the call may block.  rt_read_pixels_wait
 * returns when every copy begun so far has landed.  Whole-frame contexts only (no partition, no rt_render_gather). */
int rt_read_pixels_async(rt_ctx* ctx, uint32_t frames_back, uint8_t* dst, size_t cap);
int rt_read_pixels_wait(rt_ctx* ctx);
int rt_host_alloc(size_t bytes, void** out);   /* pinned host memory (hipHostMalloc) */
int rt_host_free(void* p);

int rt_get_stats(rt_ctx* ctx, rt_stats* out);

/* ---- device-pointer interop (process-per-GPU hosts: torch.distributed / RCCL) --------- */

/* As rt_render, but the kernel runs on `hip_stream` (a hipStream_t; NULL = the context's
 * stream) and writes the compact tile buffer to `device_dst` (device memory of this
 * context's GPU, >= rt_padded_tiles*8*W*4 bytes).  Nothing is copied to the host.  Frames
 * enqueued on different streams (into different buffers) may run concurrently, as with
 * rt_render; frames on one stream execute in order. */
int rt_render_to(rt_ctx* ctx, void* device_dst, size_t cap, void* hip_stream);

/* De-interleaves an all-gathered buffer [world][padded_tiles][8][W][4] into the row-major
 * frame [H][W][4] (both device memory) on `hip_stream`. */
int rt_assemble_frame(rt_ctx* ctx, const void* gathered, void* frame, uint32_t world,
                      void* hip_stream);

/* Device address of the colour buffer of the latest rt_render (valid until the next
 * rt_render / rt_resize / rt_destroy). */
int rt_device_pixels(rt_ctx* ctx, void** out_ptr, size_t* out_bytes);

/* ---- multi-GPU: render + RCCL gather behind one call (RR:434-470 across a group of GPUs) ------ */

/* One process per GPU.  Rank 0 calls rt_comm_unique_id and hands the bytes to the other ranks by
 * any side channel (a TCP store, MPI, a file); then EVERY rank calls rt_comm_init on its context
 * (collective: ncclCommInitRank on the context's device).  It fixes the context's partition to
 * (rank, world) as rt_set_partition does. */
#define RT355_COMM_ID_BYTES 128
int rt_comm_unique_id(uint8_t id[RT355_COMM_ID_BYTES]);
int rt_comm_init(rt_ctx* ctx, const uint8_t id[RT355_COMM_ID_BYTES], uint32_t rank, uint32_t world);
int rt_comm_destroy(rt_ctx* ctx);      /* back to a single-GPU context (rank 0 of 1) */

/* Failure of a peer.  rt_wait on a context with a communicator polls its frames' events and, between
 * polls, ncclCommGetAsyncError; when RCCL reports an error, or when `ms` > 0 and the frames have not
 * completed within `ms` milliseconds of the rt_wait call, the communicator is aborted (ncclCommAbort:
 * the exchange kernels leave the device), rt_wait returns RT_ERR_COMM, and every later rt_render_gather /
 * rt_group_render on it returns RT_ERR_COMM at once.  The context stays valid for rt_comm_destroy,
 * rt_destroy and single-GPU rendering.  A host that wants to retry forms a new group (in a fresh child
 * process if the GPU itself is gone).  ms = 0 (default): no deadline, errors only.  The deadline bounds the whole rt_wait --
 * the render kernels of the frames in flight AND their exchanges --, so it must be chosen above the slowest batch of frames the
 * host enqueues (a full-size C5 frame renders for ~20 ms on one GPU): it is a liveness bound, not a latency target. */
int rt_set_comm_timeout(rt_ctx* ctx, uint32_t ms);

/* Collective; replaces RendererRaytracing.render()'s submit (RR:442-446, 465) for the whole group:
 * this rank's tiles are rendered, exchanged over RCCL on the same stream and de-interleaved into the
 * row-major W x H frame.  root >= 0: only that rank receives (grouped ncclSend / ncclRecv -- each
 * rank's tiles travel once, over its direct xGMI link to the root); root = -1: every rank receives
 * (ncclAllGather).  Returns after enqueueing; rt_wait completes it.  Frames enqueued back to back
 * overlap on the device as with rt_render (four streams / buffer sets).  Every rank of the group must
 * make the same sequence of rt_render_gather calls with the same root. */
int rt_render_gather(rt_ctx* ctx, int root);

/* The frame of the latest rt_render_gather on a rank that received it: device address (valid until
 * four more frames are enqueued / rt_resize / rt_destroy; after rt_resize or rt_set_partition the call
 * fails with RT_ERR_STATE until the next rt_render_gather), or a copy to host memory (waits first;
 * cap >= W*H*4).  RT_ERR_STATE on a rank that did not receive.  rt_read_pixels / rt_device_pixels
 * keep returning this rank's own tiles. */
int rt_frame_pixels(rt_ctx* ctx, void** out_ptr, size_t* out_bytes);
int rt_read_frame(rt_ctx* ctx, uint8_t* dst, size_t cap);

/* One process, all GPUs -- the shape of the reference's host, ONE JavaScript thread (src/app.ts):
 * a context per device (n_devices = 0: every visible device) joined by ncclCommInitAll.  Scene and
 * parameters are written per member: for (i < rt_group_size(g)) rt_write_*(rt_group_ctx(g, i), ...).
 * rt_group_render enqueues the render on every device, the exchange (one RCCL group over all
 * devices) and the de-interleave; rt_group_wait completes it; the frame is read from the root's
 * context with rt_read_frame / rt_frame_pixels (root = -1: from any member). */
typedef struct rt_group rt_group;
int rt_group_create(int n_devices, rt_group** out);
int rt_group_destroy(rt_group* g);
int rt_group_size(const rt_group* g);
rt_ctx* rt_group_ctx(rt_group* g, int i);
int rt_group_render(rt_group* g, int root);
int rt_group_wait(rt_group* g);

/* ---- diagnostics -------------------------------------------------------------------------- */

/* Runs the HOST side of the sphere path on its own -- the build of the bounding-sphere
 * hierarchy the fast mode walks -- without a device or a context, so that it can
 * be checked on a machine without a GPU.  `records` as for rt_write_spheres.  Writes n_nodes + 1
 * node records (4 floats each: centre * 2^40, (|C|^2 (1-2^-17) - R^2 (1+2^-16)) * 2^80; leaf records
 * are zero here, the device fills them) and links (inner node: 4 * index of the first node after
 * its subtree; leaf: 0x80000000 | sphere index; the last entry is the sentinel).  RT_ERR_CAPACITY
 * when cap_nodes < n_nodes + 1 (*n_nodes is set either way; at most 2 n + 64 nodes). */
int rt_build_hierarchy(const float* records, uint32_t n, float* rec4, uint32_t* link, uint32_t cap_nodes,
                       uint32_t* n_nodes);

/* Runs the HOST side of the triangle kernel's pair-record forms on its own (no device, no context): the relinked copy of the BLAS
 * trees they walk.  `nodes`: the node buffer as rt_write_nodes receives it (8 f32 per node); `roots`: the
 * rootNodeIndex of every instance.  Writes *n_pairs records of 16 words {c1.min.xyz, meta1, c1.max.xyz, 0, c2.min.xyz, meta2,
 * c2.max.xyz, 0} -- the two children of an inner node, meta = primitiveCount << 16 | x with x = the leaf's first lookup
 * slot or the inner child's own record number -- ordered most-visited first (by the surface area of the parent's box), and
 * per root its meta.  RT_ERR_UNSUPPORTED for a node buffer beyond 65,536 entries or a primitiveCount beyond 65,535 (such scenes
 * are rendered by the tile-per-wave kernel), RT_ERR_CAPACITY when cap_pairs < *n_pairs (*n_pairs is set either way). */
int rt_build_flow(const float* nodes, uint32_t n_nodes, const uint32_t* roots, uint32_t n_roots, float* pairs, uint32_t cap_pairs,
                  uint32_t* n_pairs, uint32_t* root_meta);

/* Diagnostic (needs the context's device): the two kernels that turn the per-tile times of an awaited triangle frame into the
 * next frame's work list (rt_triangles.hip: order_hist, order_scatter), run on `cost[n]` (10 ns ticks) for a
 * device of `wave_slots` resident waves.  Writes order[0] = tiles to be rendered as four quarters, order[1] = as sixteen 2x2
 * blocks, order[2 .. 2 + n) = the tiles, longest class first (quarter-octave classes of the cost; within a class any order).
 * cap: entries of `order`, >= n + 2.  The kernels run twice on the same device buffers (they must leave their scan space and the
 * costs zero for the next frame); the second pass is returned.  The library calls the same kernels behind every awaited frame of
 * >= 4096 tiles. */
int rt_order_tiles(rt_ctx* ctx, const uint32_t* cost, uint32_t n, uint32_t wave_slots, uint32_t* order, size_t cap);

/* Which filter forms a frame of this scene may use (no device needed): *filter_ok = 0 when
 * max(|center| + |radius| over the spheres, |cameraPos|, |lightPosition|) is NaN, infinite or
 * >= 2^20 -- fast mode then renders the frame with the literal kernel --, *signed_filter = 1 when
 * that reach is below 342 and no sphere has a radius in (0, 2^-30) (the sign-aware filter and hierarchy
 * walk; the walk's rescaled node test needs the second condition).  A NaN in ANY record, whatever its
 * position, switches both off. */
int rt_filter_plan(const float* records, uint32_t n, const float params[24], int* filter_ok, int* signed_filter);

#ifdef __cplusplus
}
#endif
#endif /* RT355_H */
