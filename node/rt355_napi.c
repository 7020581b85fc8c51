/*
 * rt355_napi.c -- thin Node N-API shim over the C ABI of librt355.so (include/rt355.h).
 *
 * It takes the place of the WebGPU object model the reference's RendererRaytracing talks to
 * (src/rendering-raycast/renderer-raytracing.ts): every exported function is one C-ABI call;
 * the mapping to the reference's WebGPU calls is documented in include/rt355.h.
 * Plain C against /usr/include/node/node_api.h (N-API v8, Node >= 12.22), no node-gyp.
 *
 *   const rt = require('./rt355.node');
 *   const ctx = rt.create(0);  rt.resize(ctx, w, h);  rt.writeParams(ctx, Float32Array(24)); ...
 *   rt.render(ctx);            // enqueue (RR:442-446, 465)
 *   await rt.wait(ctx);        // promise resolved from a worker thread (RR:467)
 *
 * Errors: a non-zero rt_status becomes a thrown JS Error (synchronous calls) or a rejected
 * promise (wait), with rt_last_error() as the message and `.code` = the status.
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rt355.h"

#define NAPI_OK(call)                                                     \
    do {                                                                  \
        if ((call) != napi_ok) {                                          \
            napi_throw_error(env, NULL, "rt355: N-API call failed: " #call); \
            return NULL;                                                  \
        }                                                                 \
    } while (0)

static napi_value throw_status(napi_env env, int rc, rt_ctx* ctx) {
    char code[16];
    snprintf(code, sizeof code, "%d", rc);
    const char* msg = rt_last_error(ctx);
    napi_throw_error(env, code, (msg && *msg) ? msg : "rt355 error");
    return NULL;
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value* argv) {
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        napi_throw_type_error(env, NULL, "rt355: missing arguments");
        return 0;
    }
    return 1;
}

/* What a JS-side "context" or "group" is: an external pointing at one of these.  The wrapper outlives
 * rt_destroy (the pointer inside is nulled, so a second destroy or any later use throws instead of
 * touching freed memory) and carries the "an asynchronous wait is outstanding" flag: rt_ctx is not
 * thread-safe, and while rt_wait runs on a libuv worker no other call may enter the same context. */
typedef struct handle {
    rt_ctx* ctx;             /* NULL once destroyed (or for a group handle) */
    rt_group* group;         /* group handle: the group; member handle: NULL */
    struct handle* parent;   /* member of a group: the group's handle (shares its busy flag; not destroyable alone) */
    int busy;                /* a wait() promise is pending */
    int members;             /* group handle: member handles alive (they point back at this wrapper) */
    int collected;           /* group handle: its external is gone; freed with the last member */
} handle;

static void handle_finalize(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    handle* h = (handle*)data;
    if (!h) return;
    if (h->parent) {                      /* a member: its context belongs to the group */
        handle* g = h->parent;
        if (--g->members == 0 && g->collected) free(g);
        free(h);
        return;
    }
    if (!h->busy) {                       /* never free a context a worker thread is still waiting on */
        if (h->group) { rt_group_destroy(h->group); h->group = NULL; }
        else if (h->ctx) rt_destroy(h->ctx);
    }
    if (h->members > 0) { h->collected = 1; return; }   /* members still look at `group` through this wrapper */
    free(h);
}

static handle* get_handle(napi_env env, napi_value v, int want_group) {
    void* p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "rt355: first argument must be a context from create() / a group from createGroup()");
        return NULL;
    }
    handle* h = (handle*)p;
    if (!want_group && h->parent && !h->parent->group) {
        napi_throw_error(env, "-5", "rt355: this context belonged to a group that has been destroyed");
        return NULL;
    }
    if (want_group ? !h->group : !h->ctx) {
        napi_throw_error(env, "-5", want_group ? "rt355: not a live group (destroyed, or a context was passed)"
                                               : "rt355: not a live context (destroyed, or a group was passed)");
        return NULL;
    }
    if (h->busy || (h->parent && h->parent->busy)) {
        napi_throw_error(env, "-5", "rt355: an asynchronous wait() on this context is still pending; await it first");
        return NULL;
    }
    return h;
}

static rt_ctx* get_ctx(napi_env env, napi_value v) {
    handle* h = get_handle(env, v, 0);
    return h ? h->ctx : NULL;
}

static napi_value wrap_handle(napi_env env, rt_ctx* ctx, rt_group* group, handle* parent) {
    handle* h = (handle*)calloc(1, sizeof *h);
    if (!h) { napi_throw_error(env, NULL, "rt355: out of memory"); return NULL; }
    h->ctx = ctx; h->group = group; h->parent = parent;
    napi_value ext;
    if (napi_create_external(env, h, handle_finalize, NULL, &ext) != napi_ok) {
        free(h);
        napi_throw_error(env, NULL, "rt355: cannot create external");
        return NULL;
    }
    return ext;
}

static int get_u32(napi_env env, napi_value v, uint32_t* out) {
    if (napi_get_value_uint32(env, v, out) != napi_ok) {
        napi_throw_type_error(env, NULL, "rt355: expected an unsigned integer");
        return 0;
    }
    return 1;
}

static int get_i32(napi_env env, napi_value v, int32_t* out) {
    if (napi_get_value_int32(env, v, out) != napi_ok) {
        napi_throw_type_error(env, NULL, "rt355: expected an integer");
        return 0;
    }
    return 1;
}

/* typed array -> pointer + element count, checking the element type */
static int get_typed(napi_env env, napi_value v, napi_typedarray_type want, void** data, size_t* len) {
    napi_typedarray_type t;
    napi_value ab;
    size_t off;
    bool is = false;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is ||
        napi_get_typedarray_info(env, v, &t, len, data, &ab, &off) != napi_ok || t != want) {
        napi_throw_type_error(env, NULL, want == napi_float32_array ? "rt355: expected a Float32Array"
                                                                     : "rt355: expected a Uint8Array");
        return 0;
    }
    return 1;
}

static napi_value undefined(napi_env env) {
    napi_value u;
    napi_get_undefined(env, &u);
    return u;
}

/* create(device) -> external */
static napi_value Create(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    int32_t dev = 0;
    if (!get_args(env, info, 1, argv) || !get_i32(env, argv[0], &dev)) return NULL;
    rt_ctx* ctx = NULL;
    int rc = rt_create(dev, &ctx);
    if (rc != RT_OK) return throw_status(env, rc, NULL);
    napi_value ext = wrap_handle(env, ctx, NULL, NULL);
    if (!ext) rt_destroy(ctx);
    return ext;
}

static void release_pending(napi_env env, rt_ctx* ctx);
static napi_value Destroy(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    handle* h = get_handle(env, argv[0], 0);
    if (!h) return NULL;
    if (h->parent) { napi_throw_error(env, "-5", "rt355: a group member is destroyed with its group (destroyGroup)"); return NULL; }
    rt_destroy(h->ctx);                   /* waits for every copy that was begun */
    release_pending(env, h->ctx);
    h->ctx = NULL;                        /* a second destroy, or any later use, throws */
    return undefined(env);
}

static napi_value Resize(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    uint32_t w, h;
    if (!get_args(env, info, 3, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_u32(env, argv[1], &w) || !get_u32(env, argv[2], &h)) return NULL;
    int rc = rt_resize(ctx, w, h);
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);
}

static napi_value WriteParams(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    void* data; size_t len;
    if (!get_args(env, info, 2, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_typed(env, argv[1], napi_float32_array, &data, &len)) return NULL;
    if (len != 24) { napi_throw_range_error(env, NULL, "rt355: params must be a Float32Array(24) (RR:158)"); return NULL; }
    int rc = rt_write_params(ctx, (const float*)data);
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);
}

static napi_value WriteSpheres(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    void* data; size_t len;
    if (!get_args(env, info, 2, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_typed(env, argv[1], napi_float32_array, &data, &len)) return NULL;
    if (len % 8) { napi_throw_range_error(env, NULL, "rt355: sphere records are 8 floats each"); return NULL; }
    int rc = rt_write_spheres(ctx, (const float*)data, (uint32_t)(len / 8));
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);
}

static napi_value WriteCubemapFace(napi_env env, napi_callback_info info) {
    napi_value argv[5];
    int32_t face; uint32_t w, h;
    void* data; size_t len;
    if (!get_args(env, info, 5, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_i32(env, argv[1], &face) || !get_u32(env, argv[2], &w) || !get_u32(env, argv[3], &h) ||
        !get_typed(env, argv[4], napi_uint8_array, &data, &len))
        return NULL;
    if (len != (size_t)w * h * 4) { napi_throw_range_error(env, NULL, "rt355: face needs w*h*4 bytes"); return NULL; }
    int rc = rt_write_cubemap_face(ctx, face, w, h, (const uint8_t*)data);
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);
}

/* the reference's triangle scene, same byte layouts as RR:169-229 */
#define F32_UPLOAD(NAME, CALL, STRIDE, WHAT)                                                     \
    static napi_value NAME(napi_env env, napi_callback_info info) {                              \
        napi_value argv[2];                                                                      \
        void* data; size_t len;                                                                  \
        if (!get_args(env, info, 2, argv)) return NULL;                                          \
        rt_ctx* ctx = get_ctx(env, argv[0]);                                                     \
        if (!ctx || !get_typed(env, argv[1], napi_float32_array, &data, &len)) return NULL;      \
        if (len % (STRIDE)) { napi_throw_range_error(env, NULL, "rt355: " WHAT); return NULL; }  \
        int rc = CALL(ctx, (const float*)data, (uint32_t)(len / (STRIDE)));                      \
        return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);                        \
    }
F32_UPLOAD(WriteTriangles, rt_write_triangles, 40, "triangles are 40 floats each (RR:198-209)")
F32_UPLOAD(WriteBlas, rt_write_blas, 20, "BLAS records are 20 floats each (RR:169-174)")
F32_UPLOAD(WriteTriLookup, rt_write_tri_lookup, 1, "")
F32_UPLOAD(WriteBlasLookup, rt_write_blas_lookup, 1, "")

static napi_value WriteNodes(napi_env env, napi_callback_info info) {   /* (ctx, byteOffset, Float32Array) */
    napi_value argv[3];
    uint32_t off;
    void* data; size_t len;
    if (!get_args(env, info, 3, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_u32(env, argv[1], &off) || !get_typed(env, argv[2], napi_float32_array, &data, &len)) return NULL;
    if (len % 8) { napi_throw_range_error(env, NULL, "rt355: nodes are 8 floats each (RR:184-192)"); return NULL; }
    int rc = rt_write_nodes(ctx, off, (const float*)data, (uint32_t)(len / 8));
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);
}

static napi_value WriteMeshTexture(napi_env env, napi_callback_info info) {   /* (ctx, w, h, Uint8Array) */
    napi_value argv[4];
    uint32_t w, h;
    void* data; size_t len;
    if (!get_args(env, info, 4, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_u32(env, argv[1], &w) || !get_u32(env, argv[2], &h) ||
        !get_typed(env, argv[3], napi_uint8_array, &data, &len))
        return NULL;
    if (len != (size_t)w * h * 4) { napi_throw_range_error(env, NULL, "rt355: texture needs w*h*4 bytes"); return NULL; }
    int rc = rt_write_mesh_texture(ctx, w, h, (const uint8_t*)data);
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);
}

#define INT_SETTER(NAME, CALL)                                                 \
    static napi_value NAME(napi_env env, napi_callback_info info) {            \
        napi_value argv[2];                                                    \
        int32_t v;                                                             \
        if (!get_args(env, info, 2, argv)) return NULL;                        \
        rt_ctx* ctx = get_ctx(env, argv[0]);                                   \
        if (!ctx || !get_i32(env, argv[1], &v)) return NULL;                   \
        int rc = CALL(ctx, v);                                                 \
        return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);      \
    }
INT_SETTER(SelectKernel, rt_select_kernel)
INT_SETTER(SetMode, rt_set_mode)
INT_SETTER(SetVariant, rt_set_variant)

static napi_value SetPartition(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    uint32_t rank, world;
    if (!get_args(env, info, 3, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_u32(env, argv[1], &rank) || !get_u32(env, argv[2], &world)) return NULL;
    int rc = rt_set_partition(ctx, rank, world);
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);
}

static napi_value Render(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int rc = rt_render(ctx);
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);
}

/* wait(ctx) -> Promise<void>: rt_wait on a libuv worker so the JS loop is not blocked */
typedef struct {
    handle* h;               /* busy while the job is queued or running */
    int rc;
    char msg[256];
    napi_deferred deferred;
    napi_async_work work;
    napi_ref keep;           /* strong reference to the context's external: it cannot be collected (and `h` freed)
                                while the worker thread is inside rt_wait */
} wait_job;

static void wait_execute(napi_env env, void* data) {
    (void)env;
    wait_job* j = (wait_job*)data;
    j->rc = j->h->group ? rt_group_wait(j->h->group) : rt_wait(j->h->ctx);
    if (j->rc != RT_OK) {   /* rt_last_error is thread-local: fetch it on this thread */
        strncpy(j->msg, rt_last_error(j->h->ctx), sizeof j->msg - 1);
        j->msg[sizeof j->msg - 1] = 0;
    }
}

static void wait_complete(napi_env env, napi_status status, void* data) {
    wait_job* j = (wait_job*)data;
    j->h->busy = 0;
    if (status == napi_ok && j->rc == RT_OK) {
        napi_resolve_deferred(env, j->deferred, undefined(env));
    } else {
        napi_value msg, err;
        napi_create_string_utf8(env, j->rc != RT_OK ? j->msg : "rt355: wait was cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &err);
        napi_reject_deferred(env, j->deferred, err);
    }
    napi_delete_async_work(env, j->work);
    if (j->keep) napi_delete_reference(env, j->keep);
    free(j);
}

/* wait(ctx) / groupWait(group) -> Promise<void>.  Until it settles every other call on the same context
 * (or on the group and its members) throws: the C context is single-threaded. */
static napi_value queue_wait(napi_env env, handle* h, napi_value external) {
    napi_value promise, name;
    wait_job* j = (wait_job*)calloc(1, sizeof *j);
    if (!j) { napi_throw_error(env, NULL, "rt355: out of memory"); return NULL; }
    j->h = h;
    if (napi_create_reference(env, external, 1, &j->keep) != napi_ok) {
        free(j);
        napi_throw_error(env, NULL, "rt355: cannot reference the context");
        return NULL;
    }
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok ||
        napi_create_string_utf8(env, "rt355.wait", NAPI_AUTO_LENGTH, &name) != napi_ok ||
        napi_create_async_work(env, NULL, name, wait_execute, wait_complete, j, &j->work) != napi_ok) {
        napi_delete_reference(env, j->keep);
        free(j);
        napi_throw_error(env, NULL, "rt355: cannot create the wait job");
        return NULL;
    }
    h->busy = 1;
    if (napi_queue_async_work(env, j->work) != napi_ok) {
        h->busy = 0;
        napi_delete_async_work(env, j->work);
        napi_delete_reference(env, j->keep);
        free(j);
        napi_throw_error(env, NULL, "rt355: cannot queue the wait");
        return NULL;
    }
    return promise;
}

static napi_value Wait(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    handle* h = get_handle(env, argv[0], 0);
    if (!h) return NULL;
    if (h->parent) { napi_throw_error(env, "-5", "rt355: wait for a group member through groupWait()"); return NULL; }
    return queue_wait(env, h, argv[0]);
}

static napi_value WaitSync(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int rc = rt_wait(ctx);
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);
}

static napi_value ReadPixels(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    void* data; size_t len;
    if (!get_args(env, info, 2, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_typed(env, argv[1], napi_uint8_array, &data, &len)) return NULL;
    int rc = rt_read_pixels(ctx, (uint8_t*)data, len);
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);
}

static void set_num(napi_env env, napi_value obj, const char* key, double v) {
    napi_value n;
    napi_create_double(env, v, &n);
    napi_set_named_property(env, obj, key, n);
}

static napi_value Stats(napi_env env, napi_callback_info info) {
    napi_value argv[1], obj;
    if (!get_args(env, info, 1, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    rt_stats st;
    int rc = rt_get_stats(ctx, &st);
    if (rc != RT_OK) return throw_status(env, rc, ctx);
    NAPI_OK(napi_create_object(env, &obj));
    set_num(env, obj, "width", st.width);
    set_num(env, obj, "height", st.height);
    set_num(env, obj, "localTiles", st.local_tiles);
    set_num(env, obj, "spheres", st.spheres);
    set_num(env, obj, "rays", (double)st.rays);
    set_num(env, obj, "kernelMs", st.kernel_ms);
    set_num(env, obj, "prepMs", st.prep_ms);
    set_num(env, obj, "frames", st.frames);
    set_num(env, obj, "mode", st.mode);
    set_num(env, obj, "batchFrames", st.batch_frames);
    set_num(env, obj, "batchKernelMs", st.batch_kernel_ms);
    set_num(env, obj, "gatherMs", st.gather_ms);
    set_num(env, obj, "batchGatherMs", st.batch_gather_ms);
    set_num(env, obj, "kernelId", st.kernel_id);
    set_num(env, obj, "gridShare", st.grid_share);
    set_num(env, obj, "instanceUploads", st.instance_uploads);
    set_num(env, obj, "triForm", st.tri_form);
    set_num(env, obj, "pairRebuilds", st.pair_rebuilds);
    return obj;
}

/* ---- multi-GPU: render + RCCL gather behind one call (include/rt355.h) ------------------------------ */

/* readFrame(ctx, Uint8Array(W*H*4)): the whole frame of the latest renderGather / groupRender */
static napi_value ReadFrame(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    void* data; size_t len;
    if (!get_args(env, info, 2, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_typed(env, argv[1], napi_uint8_array, &data, &len)) return NULL;
    int rc = rt_read_frame(ctx, (uint8_t*)data, len);
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);
}

/* one process, all GPUs: createGroup(nDevices (0 = all)) -> group */
static napi_value CreateGroup(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    int32_t n = 0;
    if (!get_args(env, info, 1, argv) || !get_i32(env, argv[0], &n)) return NULL;
    rt_group* g = NULL;
    int rc = rt_group_create(n, &g);
    if (rc != RT_OK) return throw_status(env, rc, NULL);
    napi_value ext = wrap_handle(env, NULL, g, NULL);
    if (!ext) rt_group_destroy(g);
    return ext;
}

static napi_value DestroyGroup(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    handle* h = get_handle(env, argv[0], 1);
    if (!h) return NULL;
    /* the members' pending read-backs end with the group: rt_group_destroy waits for the copies, the arrays are released here */
    rt_ctx* members[64];
    int n = h->group ? rt_group_size(h->group) : 0;
    if (n > 64) n = 64;
    for (int i = 0; i < n; ++i) members[i] = rt_group_ctx(h->group, i);
    rt_group_destroy(h->group);
    for (int i = 0; i < n; ++i) if (members[i]) release_pending(env, members[i]);
    h->group = NULL;      /* member handles check the context pointer the group hands out: see GroupCtx */
    return undefined(env);
}

static napi_value GroupSize(napi_env env, napi_callback_info info) {
    napi_value argv[1], n;
    if (!get_args(env, info, 1, argv)) return NULL;
    handle* h = get_handle(env, argv[0], 1);
    if (!h) return NULL;
    napi_create_int32(env, rt_group_size(h->group), &n);
    return n;
}

/* groupCtx(group, i) -> the member context, for the rt_write_* / resize / setMode calls.  Valid only while
 * the group lives; the JS side must not keep it past destroyGroup (RendererRaytracing drops them there). */
static napi_value GroupCtx(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    int32_t i;
    if (!get_args(env, info, 2, argv)) return NULL;
    handle* h = get_handle(env, argv[0], 1);
    if (!h || !get_i32(env, argv[1], &i)) return NULL;
    rt_ctx* c = rt_group_ctx(h->group, i);
    if (!c) return throw_status(env, RT_ERR_INVALID_ARG, NULL);
    napi_value ext = wrap_handle(env, c, NULL, h);
    if (ext) ++h->members;
    return ext;
}

static napi_value GroupRender(napi_env env, napi_callback_info info) {   /* (group, root (-1: every member)) */
    napi_value argv[2];
    int32_t root;
    if (!get_args(env, info, 2, argv)) return NULL;
    handle* h = get_handle(env, argv[0], 1);
    if (!h || !get_i32(env, argv[1], &root)) return NULL;
    int rc = rt_group_render(h->group, root);
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, NULL);
}

static napi_value GroupWait(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    handle* h = get_handle(env, argv[0], 1);
    return h ? queue_wait(env, h, argv[0]) : NULL;
}

static napi_value GroupWaitSync(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    handle* h = get_handle(env, argv[0], 1);
    if (!h) return NULL;
    int rc = rt_group_wait(h->group);
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, NULL);
}

/* one process per GPU: commUniqueId() -> Uint8Array(128); commInit(ctx, id, rank, world); renderGather(ctx, root) */
static napi_value CommUniqueId(napi_env env, napi_callback_info info) {
    (void)info;
    uint8_t id[RT355_COMM_ID_BYTES];
    int rc = rt_comm_unique_id(id);
    if (rc != RT_OK) return throw_status(env, rc, NULL);
    napi_value ab, ta;
    void* p = NULL;
    NAPI_OK(napi_create_arraybuffer(env, RT355_COMM_ID_BYTES, &p, &ab));
    memcpy(p, id, RT355_COMM_ID_BYTES);
    NAPI_OK(napi_create_typedarray(env, napi_uint8_array, RT355_COMM_ID_BYTES, ab, 0, &ta));
    return ta;
}

static napi_value CommInit(napi_env env, napi_callback_info info) {
    napi_value argv[4];
    void* data; size_t len;
    uint32_t rank, world;
    if (!get_args(env, info, 4, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_typed(env, argv[1], napi_uint8_array, &data, &len) || !get_u32(env, argv[2], &rank) ||
        !get_u32(env, argv[3], &world))
        return NULL;
    if (len != RT355_COMM_ID_BYTES) { napi_throw_range_error(env, NULL, "rt355: the unique id is 128 bytes"); return NULL; }
    int rc = rt_comm_init(ctx, (const uint8_t*)data, rank, world);
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);
}

static napi_value CommDestroy(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int rc = rt_comm_destroy(ctx);
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);
}

INT_SETTER(RenderGather, rt_render_gather)

/* ---- streaming read-back (rt_read_pixels_async): pinned Uint8Arrays owned by the addon ------------------------------ */
static void pinned_finalize(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    if (data) rt_host_free(data);
}

/* hostAlloc(bytes) -> Uint8Array over pinned host memory (hipHostMalloc), freed when the array is collected */
static napi_value HostAlloc(napi_env env, napi_callback_info info) {
    napi_value argv[1], ab, u8;
    uint32_t bytes = 0;
    void* p = NULL;
    if (!get_args(env, info, 1, argv) || !get_u32(env, argv[0], &bytes)) return NULL;
    int rc = rt_host_alloc(bytes, &p);
    if (rc != RT_OK) return throw_status(env, rc, NULL);
    if (napi_create_external_arraybuffer(env, p, bytes, pinned_finalize, NULL, &ab) != napi_ok ||
        napi_create_typedarray(env, napi_uint8_array, bytes, ab, 0, &u8) != napi_ok) {
        rt_host_free(p);
        napi_throw_error(env, NULL, "rt355: cannot wrap the pinned buffer");
        return NULL;
    }
    return u8;
}

/* Destinations of copies that have been begun and not yet awaited: the addon holds a reference to each array, so that a
 * script that drops its own cannot have the (pinned) memory freed under a running DMA; readPixelsWait releases them. */
/* The table belongs to the environment that loaded the addon (napi_set_instance_data in Init): a reference is deleted with
 * the napi_env it was created with, also when worker threads load the addon a second time. */
#define PENDING_MAX 64
typedef struct pending_table { struct { napi_ref ref; rt_ctx* ctx; } e[PENDING_MAX]; int n; } pending_table;
static pending_table* pending_of(napi_env env) {
    void* p = NULL;
    return napi_get_instance_data(env, &p) == napi_ok ? (pending_table*)p : NULL;
}
static void free_pending_table(napi_env env, void* data, void* hint) { (void)env; (void)hint; free(data); }
static void release_pending(napi_env env, rt_ctx* ctx) {
    pending_table* t = pending_of(env);
    if (!t) return;
    int k = 0;
    for (int i = 0; i < t->n; ++i) {
        if (t->e[i].ctx == ctx) napi_delete_reference(env, t->e[i].ref);
        else t->e[k++] = t->e[i];
    }
    t->n = k;
}

/* readPixelsAsync(ctx, framesBack, Uint8Array): the copy runs beside the rendering of the next frames; the array stays
 * referenced by the addon until readPixelsWait(ctx) has returned */
static napi_value ReadPixelsAsync(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    void* data; size_t len;
    uint32_t back = 0;
    if (!get_args(env, info, 3, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_u32(env, argv[1], &back) || !get_typed(env, argv[2], napi_uint8_array, &data, &len)) return NULL;
    pending_table* t = pending_of(env);
    if (!t) { napi_throw_error(env, NULL, "rt355: no instance data"); return NULL; }
    if (t->n == PENDING_MAX) {                 /* more copies begun than anybody awaits: complete this context's, then go on */
        int rcw = rt_read_pixels_wait(ctx);
        if (rcw != RT_OK) return throw_status(env, rcw, ctx);
        release_pending(env, ctx);
        if (t->n == PENDING_MAX) { napi_throw_error(env, NULL, "rt355: too many read-backs pending on other contexts"); return NULL; }
    }
    napi_ref ref;
    if (napi_create_reference(env, argv[2], 1, &ref) != napi_ok) { napi_throw_error(env, NULL, "rt355: cannot reference the destination"); return NULL; }
    int rc = rt_read_pixels_async(ctx, back, (uint8_t*)data, len);
    if (rc != RT_OK) { napi_delete_reference(env, ref); return throw_status(env, rc, ctx); }
    t->e[t->n].ref = ref; t->e[t->n].ctx = ctx; ++t->n;
    return undefined(env);
}

static napi_value ReadPixelsWait(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int rc = rt_read_pixels_wait(ctx);
    release_pending(env, ctx);
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);
}

/* setCommTimeout(ctx, ms): deadline of wait() on a context with a communicator (0: RCCL errors only) */
static napi_value SetCommTimeout(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    uint32_t ms = 0;
    if (!get_args(env, info, 2, argv)) return NULL;
    rt_ctx* ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_u32(env, argv[1], &ms)) return NULL;
    int rc = rt_set_comm_timeout(ctx, ms);
    return rc == RT_OK ? undefined(env) : throw_status(env, rc, ctx);
}

static napi_value BuildId(napi_env env, napi_callback_info info) {
    (void)info;
    napi_value s;
    napi_create_string_utf8(env, rt_build_id(), NAPI_AUTO_LENGTH, &s);
    return s;
}

static napi_value KernelName(napi_env env, napi_callback_info info) {
    napi_value argv[1], s;
    uint32_t id = 0;
    if (!get_args(env, info, 1, argv) || !get_u32(env, argv[0], &id)) return NULL;
    napi_create_string_utf8(env, rt_kernel_name((int)id), NAPI_AUTO_LENGTH, &s);
    return s;
}

static napi_value AbiVersion(napi_env env, napi_callback_info info) {
    (void)info;
    napi_value n;
    napi_create_int32(env, rt_abi_version(), &n);
    return n;
}

static napi_value Init(napi_env env, napi_value exports) {
    pending_table* table = (pending_table*)calloc(1, sizeof(pending_table));
    if (!table || napi_set_instance_data(env, table, free_pending_table, NULL) != napi_ok) {
        free(table);
        napi_throw_error(env, NULL, "rt355: cannot set instance data");
        return NULL;
    }
    static const struct { const char* name; napi_callback fn; } fns[] = {
        {"create", Create}, {"destroy", Destroy}, {"resize", Resize}, {"writeParams", WriteParams},
        {"writeSpheres", WriteSpheres}, {"writeCubemapFace", WriteCubemapFace}, {"selectKernel", SelectKernel},
        {"writeTriangles", WriteTriangles}, {"writeNodes", WriteNodes}, {"writeBlas", WriteBlas},
        {"writeTriLookup", WriteTriLookup}, {"writeBlasLookup", WriteBlasLookup}, {"writeMeshTexture", WriteMeshTexture},
        {"setMode", SetMode}, {"setVariant", SetVariant}, {"setPartition", SetPartition}, {"render", Render},
        {"wait", Wait}, {"waitSync", WaitSync}, {"readPixels", ReadPixels}, {"stats", Stats},
        {"abiVersion", AbiVersion}, {"buildId", BuildId}, {"kernelName", KernelName},
        {"hostAlloc", HostAlloc}, {"readPixelsAsync", ReadPixelsAsync}, {"readPixelsWait", ReadPixelsWait}, {"setCommTimeout", SetCommTimeout},
        {"readFrame", ReadFrame}, {"createGroup", CreateGroup}, {"destroyGroup", DestroyGroup}, {"groupSize", GroupSize},
        {"groupCtx", GroupCtx}, {"groupRender", GroupRender}, {"groupWait", GroupWait}, {"groupWaitSync", GroupWaitSync},
        {"commUniqueId", CommUniqueId}, {"commInit", CommInit}, {"commDestroy", CommDestroy}, {"renderGather", RenderGather},
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; ++i) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok ||
            napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) {
            napi_throw_error(env, NULL, "rt355: cannot export function");
            return NULL;
        }
    }
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
