// rt_api.hip -- host side of librt355.so: the C ABI declared in include/rt355.h.
// One context = one GPU + four streams (frames in flight); see the header for what each entry point replaces in
// the reference's renderer-raytracing.ts.  No CPU fallback: without a gfx950 device
// rt_create fails and nothing else can be called.
#include "rt_ctx.h"

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rt_bvh_build.h"
#include "rt_tlas_fit.h"

thread_local std::string g_rt_err;
thread_local int g_rt_kernel_id = 0;
thread_local int g_rt_tri_form = 0;

#ifndef RT355_BUILD_ID
#define RT355_BUILD_ID "unknown"
#endif

// children per inner node of the sphere hierarchy (rt_bvh_build.h); development builds read it from the environment
static uint32_t rt_bvh_arity() {
#ifdef RT355_DEV_EXPORTS
    if (const char* e = getenv("RT355_BVH_ARITY")) return (uint32_t)atoi(e);
#endif
    return 4u;
}

// Rebuilds the hierarchy's topology for moved spheres off the caller's thread.  rt_write_spheres with an
// unchanged sphere count posts the new records here; frames meanwhile use the OLD topology with node
// bounds refitted on the device (rt_bvh.hip: bvh_refit) -- always valid, only looser as spheres drift --
// and the next frame that follows a sphere write after the worker has finished takes the new one.
struct rt_rebuild {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    bool stop = false, pending = false, ready = false;
    std::vector<float> in_records;            // job (latest posted wins)
    uint32_t in_n = 0;
    std::vector<float> out_rec;               // result
    std::vector<uint32_t> out_link;
    uint32_t out_n = 0, out_nodes = 0;

    void run() {
        std::vector<float> records, rec;
        std::vector<uint32_t> link;
        for (;;) {
            uint32_t n;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || pending; });
                if (stop) return;
                records.swap(in_records);
                n = in_n;
                pending = false;
            }
            const uint32_t nodes = rt_bvh_build(records.data(), n, rec, link, rt_bvh_arity());
            std::lock_guard<std::mutex> lk(m);
            out_rec.swap(rec);
            out_link.swap(link);
            out_n = n;
            out_nodes = nodes;
            ready = true;
        }
    }
    void post(const float* records, uint32_t n) {
        {
            std::lock_guard<std::mutex> lk(m);
            in_records.assign(records, records + 8u * (size_t)n);
            in_n = n;
            pending = true;
        }
        if (!th.joinable()) th = std::thread([this] { run(); });
        cv.notify_one();
    }
    ~rt_rebuild() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_one();
        if (th.joinable()) th.join();
    }
};

namespace {

uint32_t tiles_total(uint32_t H) { return (H + 7u) / 8u; }

// max(scene bound, |camera|, |light|) with a sticky NaN: any NaN operand makes the reach NaN, which
// switches both filter forms off (enqueue)
double rt_reach(double bound, double cam, double lgt) {
    if (bound != bound || cam != cam || lgt != lgt) return NAN;
    return std::max(bound, std::max(cam, lgt));
}

// max over spheres of |center| + |radius|.  A NaN anywhere is sticky -- the bound of a scene with a
// NaN record is NaN whatever the record's position --, an infinite record gives +inf.
double rt_scene_bound(const float* records, uint32_t n) {
    double bound = 0.0;
    for (uint32_t i = 0; i < n; ++i) {
        const float* r = records + 8u * (size_t)i;
        const double len = std::sqrt((double)r[0] * r[0] + (double)r[1] * r[1] + (double)r[2] * r[2]) + std::fabs((double)r[7]);
        if (len != len || bound != bound) bound = NAN;
        else if (len > bound) bound = len;
    }
    return bound;
}

// Which filter forms a frame may use.  reach: bound on |ray origin| and on |center| + radius.  The
// sign-aware filter is valid only while the rounding of h.oc (<= 7.3e-7 |oc|, |oc| <= 2*reach) stays
// below half of the 0.001 a valid hit needs (rt_filter.h: filter_one); beyond 2^20 (or NaN / inf) the
// 2^40-scaled filter arithmetic could overflow, and the frame is rendered by the literal kernel.
// smallest non-zero |radius| (+inf if there is none): the sign-aware forms rescale the test by 2^-124, which
// must not push the quantities that decide it into the denormals; with every radius 0 or >= 2^-30 the
// filter's margins (>= 2^-16 r^2) stay 40 binary orders above the smallest normal number
double rt_scene_min_radius(const float* records, uint32_t n) {
    double mn = INFINITY;
    for (uint32_t i = 0; i < n; ++i) {
        const double r = std::fabs((double)records[8u * (size_t)i + 7u]);
        if (r > 0.0 && r < mn) mn = r;
    }
    return mn;
}

void rt_plan(double scene_bound, double min_radius, const float* p, bool& filter_ok, uint32_t& signed_filter) {
    const double cam = std::sqrt((double)p[0] * p[0] + (double)p[1] * p[1] + (double)p[2] * p[2]);
    const double lgt = std::sqrt((double)p[16] * p[16] + (double)p[17] * p[17] + (double)p[18] * p[18]);
    // the camera's basis vectors (forwards, right, up: p[4..6], p[8..10], p[12..14]) bound the primary directions the same
    // way: below 2^20 the squared length of fw + hc rt + vc up cannot overflow, so normalize() never returns the zero
    // vector (rt_bvh.hip: flat_sky relies on it); a NaN or absurd basis sends the frame to the literal kernel
    double basis = 0.0;
    for (int k = 4; k < 15; ++k)
        if (k % 4 != 3) { const double a = std::fabs((double)p[k]); basis = (a != a || basis != basis) ? NAN : std::max(basis, a); }
    const double reach = rt_reach(scene_bound, cam, basis != basis ? (double)NAN : std::max(lgt, basis));   // std::max keeps a NaN first operand
    filter_ok = reach == reach && reach < 1048576.0;
    signed_filter = (filter_ok && 2.0 * reach * 7.3e-7 < 5.0e-4 && min_radius >= 9.313225746154785e-10) ? 1u : 0u;   // 2^-30
}

}  // namespace

extern "C" {

int rt_abi_version(void) { return RT355_ABI_VERSION; }

const char* rt_build_id(void) { return RT355_BUILD_ID; }

const char* rt_kernel_name(int id) {
    switch (id) {
        case RT_KID_LITERAL: return "trace_pixels<literal>";
        case RT_KID_BRUTE_SINGLE: return "trace_pixels<filter>";
        case RT_KID_BRUTE_PIPELINE: return "first_bounce+trace_paths";
        case RT_KID_HIERARCHY_8: return "bvh_pixels<8>";
        case RT_KID_HIERARCHY_12: return "bvh_pixels<12>";
        case RT_KID_HIERARCHY_16: return "bvh_pixels<16>";
        case RT_KID_HIERARCHY_GLOBAL: return "bvh_pixels<global>";
        case RT_KID_TRIANGLES: return "trace_triangles";
        case RT_KID_HEATMAP: return "heatmap_triangles";
        case RT_KID_TRIANGLES_ROLES: return "trace_roles";
        default: return "none";
    }
}

const char* rt_last_error(rt_ctx*) { return g_rt_err.c_str(); }

// A context keeps four frames in flight on four streams, next to its copy stream and RCCL's: with the HIP runtime's default of
// four hardware queues per device several of them share a queue and serialise (C3 through the N > 1 path: 3.67 instead of
// 2.43 ms per frame, round 2).  The runtime reads GPU_MAX_HW_QUEUES once, when it initialises -- at the process's first HIP
// call --, so the library asks for eight before ITS first HIP call, unless the host has set the variable itself
// (RT355_KEEP_HW_QUEUES=1: leave the runtime's default alone).  A host that has initialised HIP before it creates its first
// context keeps what its runtime read then.
void rt_default_hw_queues(void) {
    static bool done = false;
    if (done) return;
    done = true;
    if (!getenv("RT355_KEEP_HW_QUEUES")) (void)setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

int rt_create(int device, rt_ctx** out) {
    if (!out) return fail(RT_ERR_INVALID_ARG, "rt_create: out is NULL");
    *out = nullptr;
    rt_default_hw_queues();
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(RT_ERR_NO_DEVICE, "rt_create: no HIP device visible (this library has no CPU path)");
    }
    if (device < 0 || device >= count) return fail(RT_ERR_NO_DEVICE, "rt_create: device ordinal out of range");
    hipDeviceProp_t prop;
    RT_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        char buf[200];
        std::snprintf(buf, sizeof buf, "rt_create: device %d is %s; librt355 carries gfx950 code only", device,
                      prop.gcnArchName);
        return fail(RT_ERR_NO_DEVICE, buf);
    }
    RT_HIP(hipSetDevice(device));
    rt_ctx* c = new (std::nothrow) rt_ctx();
    if (!c) return fail(RT_ERR_HIP, "rt_create: out of host memory");
    c->device = device;
    c->n_cus = (uint32_t)prop.multiProcessorCount;
    c->wave_slots = (uint32_t)prop.multiProcessorCount * 20u;      // 4 SIMDs x 5 waves: the form of the triangle kernel that awaited frames -- the ones with a work list -- run (rt_launch_triangles)
    hipError_t err;
    err = hipSuccess;
    for (int k = 0; k < kStreams && err == hipSuccess; ++k) err = hipStreamCreateWithFlags(&c->streams[k], hipStreamNonBlocking);
    c->stream = c->streams[0];
    if (err == hipSuccess) err = hipEventCreateWithFlags(&c->ev_scene, hipEventDisableTiming);
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    for (int k = 0; k < kStreams && err == hipSuccess; ++k) err = hipEventCreateWithFlags(&c->ev_copy[k], hipEventDisableTiming);
    for (int i = 0; i < RT355_MAX_IN_FLIGHT && err == hipSuccess; ++i) {
        if ((err = hipEventCreate(&c->ev_prep0[i])) != hipSuccess) break;
        if ((err = hipEventCreate(&c->ev_k0[i])) != hipSuccess) break;
        if ((err = hipEventCreate(&c->ev_k1[i])) != hipSuccess) break;
        err = hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming);
    }
    if (err != hipSuccess ||
        (err = hipMalloc(reinterpret_cast<void**>(&c->d_rays), kCtrlBytes * RT355_MAX_IN_FLIGHT)) != hipSuccess ||
        (err = hipMemset(c->d_rays, 0, kCtrlBytes * RT355_MAX_IN_FLIGHT)) != hipSuccess ||
        (err = hipHostMalloc(reinterpret_cast<void**>(&c->h_rays), 16u * RT355_MAX_IN_FLIGHT + 8u * kStreams,
                             hipHostMallocDefault)) != hipSuccess) {
        rt_destroy(c);
        return fail_hip(err, "rt_create: stream/event/counter setup");
    }
    std::memset(c->h_rays, 0, 16u * RT355_MAX_IN_FLIGHT + 8u * kStreams);
    c->h_split = c->h_rays + 2u * RT355_MAX_IN_FLIGHT;
    // An event's FIRST record allocates its signal (~10 us each, measured through bench.py's first timed region: 0.25 ms for twelve
    // untouched slots): every event of the ring is recorded once here, so that a host's first frames do not pay for it one by one.
    for (int i = 0; i < RT355_MAX_IN_FLIGHT && err == hipSuccess; ++i) {
        if ((err = hipEventRecord(c->ev_prep0[i], c->stream)) != hipSuccess) break;
        if ((err = hipEventRecord(c->ev_k0[i], c->stream)) != hipSuccess) break;
        if ((err = hipEventRecord(c->ev_k1[i], c->stream)) != hipSuccess) break;
        err = hipEventRecord(c->ev_done[i], c->stream);
    }
    if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
    if (err != hipSuccess) {
        rt_destroy(c);
        return fail_hip(err, "rt_create: event warm-up");
    }
    *out = c;
    return RT_OK;
}

int rt_destroy(rt_ctx* c) {
    if (!c) return RT_OK;
    (void)hipSetDevice(c->device);
    for (uint32_t i = 0; i < c->in_flight; ++i) (void)hipEventSynchronize(c->ev_done[i]);
    for (int k = 0; k < kStreams; ++k)
        if (c->streams[k]) (void)hipStreamSynchronize(c->streams[k]);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (int k = 0; k < kStreams; ++k)
        if (c->ev_copy[k]) (void)hipEventDestroy(c->ev_copy[k]);
    rt_comm_release(c);
    delete c->rebuild;
    c->rebuild = nullptr;
    (void)hipFree(c->d_records);
    (void)hipFree(c->d_scene);
    for (int i = 0; i < 6; ++i) (void)hipFree(c->d_face[i]);
    for (int k = 0; k < kStreams; ++k) (void)hipFree(c->d_outs[k]);
    for (int k = 0; k < kStreams; ++k) (void)hipFree(c->d_fin[k]);
    (void)hipFree(c->d_queue);
    (void)hipFree(c->d_bvh_rec);
    (void)hipFree(c->d_bvh_link);
    for (rt_ctx::DevBuf* b : {&c->d_tri, &c->d_tri_lookup, &c->d_tex, &c->d_corners, &c->d_flow}) (void)hipFree(b->p);
    for (int k = 0; k < kStreams; ++k) { (void)hipFree(c->d_tile_cost[k].p); (void)hipFree(c->d_tile_order[k].p); }
    for (int v = 0; v < kVersions; ++v)
        for (rt_ctx::DevBuf* b : {&c->d_nodes[v], &c->d_blas[v], &c->d_blas_lookup[v]}) (void)hipFree(b->p);
    (void)hipFree(c->d_rays);
    if (c->h_rays) (void)hipHostFree(c->h_rays);
    for (int i = 0; i < RT355_MAX_IN_FLIGHT; ++i) {
        if (c->ev_prep0[i]) (void)hipEventDestroy(c->ev_prep0[i]);
        if (c->ev_k0[i]) (void)hipEventDestroy(c->ev_k0[i]);
        if (c->ev_k1[i]) (void)hipEventDestroy(c->ev_k1[i]);
        if (c->ev_done[i]) (void)hipEventDestroy(c->ev_done[i]);
    }
    if (c->ev_scene) (void)hipEventDestroy(c->ev_scene);
    for (int v = 0; v < kVersions; ++v)
        if (c->ev_ver[v]) (void)hipEventDestroy(c->ev_ver[v]);
    for (int k = 0; k < kStreams; ++k)
        if (c->streams[k]) (void)hipStreamDestroy(c->streams[k]);
    delete c;
    return RT_OK;
}

int rt_wait(rt_ctx* c);
int rt_read_pixels_wait(rt_ctx* c);
uint32_t rt_local_tiles(const rt_ctx* c) { return rt_tiles_of_rank(c->H, c->rank, c->world); }
static uint32_t local_tiles(const rt_ctx* c) { return rt_local_tiles(c); }

uint32_t rt_tiles_of_rank(uint32_t height, uint32_t rank, uint32_t world) {
    if (world == 0 || rank >= world) return 0;
    const uint32_t t = tiles_total(height);
    return t > rank ? (t - rank + world - 1u) / world : 0u;
}

uint32_t rt_padded_tiles(uint32_t height, uint32_t world) {
    if (world == 0) return 0;
    return (tiles_total(height) + world - 1u) / world;
}

// Scene-setup calls change device state that frames in flight may read: they drain first.
int rt_drain(rt_ctx* c) {
    bool copies = false;
    for (int k = 0; k < kStreams; ++k) copies = copies || c->copy_pending[k];
    if (copies) { int rc = rt_read_pixels_wait(c); if (rc != RT_OK) return rc; }
    return c->in_flight ? rt_wait(c) : RT_OK;
}
static int drain(rt_ctx* c) { return rt_drain(c); }

static int ensure_out(rt_ctx* c) {
    // sized for the padded tile count so that the buffer can be an all-gather operand; one
    // buffer per stream rt_render rotates over (consecutive frames may overlap on the device)
    const size_t need = (size_t)rt_padded_tiles(c->H, c->world) * 8u * c->W * 4u;
    if (need > c->out_bytes || !c->d_outs[0]) {
        RT_HIP(hipStreamSynchronize(c->stream));
        for (int k = 0; k < kStreams; ++k) { (void)hipFree(c->d_outs[k]); c->d_outs[k] = nullptr; }
        c->d_out = nullptr;
        c->out_bytes = 0;
        if (need) {
            for (int k = 0; k < kStreams; ++k) {
                RT_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_outs[k]), need));
                RT_HIP(hipMemsetAsync(c->d_outs[k], 0, need, c->stream));
            }
            RT_HIP(hipStreamSynchronize(c->stream));
            c->d_out = c->d_outs[0];
            c->out_bytes = need;
        }
    }
    return RT_OK;
}

// The path queue of the two-kernel brute-force pipeline (48 B per local pixel in the worst case:
// every primary ray hits) is allocated when a frame first takes that pipeline -- the default
// hierarchy path never reads it (1.6 GB at 8K).
static int ensure_queue(rt_ctx* c) {
    const size_t entries = (size_t)rt_padded_tiles(c->H, c->world) * 8u * c->W;
    if (entries > c->queue_cap) {
        { int rc = drain(c); if (rc != RT_OK) return rc; }
        (void)hipFree(c->d_queue);
        c->d_queue = nullptr;
        c->queue_cap = 0;
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_queue), entries * 3u * sizeof(float4)));
        c->queue_cap = entries;
    }
    return RT_OK;
}

// End-of-path records for the hierarchy kernel under a textured sky (32 B per local pixel slot, one buffer per
// frame that may be in flight on its own stream): allocated when a frame first needs them.
static int ensure_fin(rt_ctx* c) {
    const size_t slots = (size_t)rt_padded_tiles(c->H, c->world) * 8u * c->W;
    if (slots > c->fin_slots) {
        { int rc = drain(c); if (rc != RT_OK) return rc; }
        for (int k = 0; k < kStreams; ++k) { (void)hipFree(c->d_fin[k]); c->d_fin[k] = nullptr; }
        c->fin_slots = 0;
        for (int k = 0; k < kStreams; ++k) RT_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_fin[k]), slots * 2u * sizeof(float4)));
        c->fin_slots = slots;
    }
    return RT_OK;
}

int rt_resize(rt_ctx* c, uint32_t width, uint32_t height) {
    if (!c) return fail(RT_ERR_INVALID_ARG, "rt_resize: ctx is NULL");
    if (width == 0 || height == 0 || width > 65536u || height > 65536u)
        return fail(RT_ERR_INVALID_ARG, "rt_resize: width/height must be in 1..65536");
    // pixel indices and the padded tile space are 32-bit in the kernels
    if ((uint64_t)((width + 7u) / 8u) * ((height + 7u) / 8u) * 64u >= (1ull << 31))
        return fail(RT_ERR_INVALID_ARG, "rt_resize: more than 2^31 pixel slots (width x height, padded to 8x8 tiles)");
    RT_HIP(hipSetDevice(c->device));
    { int rc = drain(c); if (rc != RT_OK) return rc; }
    c->W = width;
    c->H = height;
    return ensure_out(c);
}

int rt_set_partition(rt_ctx* c, uint32_t rank, uint32_t world) {
    if (!c) return fail(RT_ERR_INVALID_ARG, "rt_set_partition: ctx is NULL");
    if (world == 0 || rank >= world) return fail(RT_ERR_INVALID_ARG, "rt_set_partition: need rank < world");
    if (c->comm && (rank != c->rank || world != c->world))
        return fail(RT_ERR_STATE, "rt_set_partition: the partition of a context with a communicator is its rank in the group");
    RT_HIP(hipSetDevice(c->device));
    { int rc = drain(c); if (rc != RT_OK) return rc; }
    c->rank = rank;
    c->world = world;
    if (c->W && c->H) return ensure_out(c);
    return RT_OK;
}

int rt_write_params(rt_ctx* c, const float params[24]) {
    if (!c || !params) return fail(RT_ERR_INVALID_ARG, "rt_write_params: NULL argument");
    std::memcpy(c->params, params, sizeof c->params);   // travels in the kernarg segment
    c->have_params = true;
    c->prep_params_valid = false;
    return RT_OK;
}

int rt_write_spheres(rt_ctx* c, const float* records, uint32_t n) {
    if (!c || (n && !records)) return fail(RT_ERR_INVALID_ARG, "rt_write_spheres: NULL argument");
    if (n > (1u << 24)) return fail(RT_ERR_INVALID_ARG, "rt_write_spheres: more than 2^24 spheres");
    RT_HIP(hipSetDevice(c->device));
    { int rc = drain(c); if (rc != RT_OK) return rc; }
    if (n > c->cap_n) {
        RT_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_records); (void)hipFree(c->d_scene);
        c->d_records = nullptr; c->d_scene = nullptr;
        c->cap_n = 0;
        const size_t cap16 = ((size_t)n + 15u) & ~(size_t)15u;
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_records), (size_t)n * 32u));
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_scene), cap16 * 8u * sizeof(float4)));
        c->cap_n = n;
    }
    if (n) {
        RT_HIP(hipMemcpyAsync(c->d_records, records, (size_t)n * 32u, hipMemcpyHostToDevice, c->stream));
        RT_HIP(hipStreamSynchronize(c->stream));   // caller may free `records` now (writeBuffer semantics)
    }
    c->n = n;
    c->scene_kind = 0;
    c->n16 = (n + 15u) & ~15u;
    c->h_records.assign(records, records + 8u * (size_t)n);   // the hierarchy is built lazily from this copy
    c->bvh_valid = false;
    // Moved spheres, same count: the next frame refits the node bounds on the device and keeps the topology;
    // a new topology for these positions is built on the worker thread meanwhile (enqueue takes it when done).
    if (n != 0 && n == c->bvh_topo_n) {
        if (!c->rebuild) c->rebuild = new (std::nothrow) rt_rebuild();
        if (c->rebuild) c->rebuild->post(records, n);
    }
    c->scene_bound = (float)rt_scene_bound(records, n);   // scene extent, for the filter's validity range (enqueue)
    c->scene_min_radius = rt_scene_min_radius(records, n);
    c->have_spheres = true;
    c->prep_spheres_valid = false;
    return RT_OK;
}

int rt_write_cubemap_face(rt_ctx* c, int face, uint32_t w, uint32_t h, const uint8_t* rgba) {
    if (!c || !rgba) return fail(RT_ERR_INVALID_ARG, "rt_write_cubemap_face: NULL argument");
    if (face < 0 || face > 5) return fail(RT_ERR_INVALID_ARG, "rt_write_cubemap_face: face must be 0..5");
    if (w == 0 || h == 0 || w > 16384u || h > 16384u)
        return fail(RT_ERR_INVALID_ARG, "rt_write_cubemap_face: face size must be in 1..16384");
    RT_HIP(hipSetDevice(c->device));
    { int rc = drain(c); if (rc != RT_OK) return rc; }
    const size_t bytes = (size_t)w * h * 4u;
    if (c->fw[face] != w || c->fh[face] != h || !c->d_face[face]) {
        RT_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_face[face]);
        c->d_face[face] = nullptr;
        c->fw[face] = c->fh[face] = 0;
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_face[face]), bytes < 16 ? 16 : bytes));
        c->fw[face] = w;
        c->fh[face] = h;
    }
    RT_HIP(hipMemcpyAsync(c->d_face[face], rgba, bytes, hipMemcpyHostToDevice, c->stream));
    RT_HIP(hipStreamSynchronize(c->stream));
    c->face_texel0[face] = (uint32_t)rgba[0] | ((uint32_t)rgba[1] << 8) | ((uint32_t)rgba[2] << 16);
    return RT_OK;
}

// capacity of a device buffer, contents kept; drains when it has to reallocate
static int grow_buf(rt_ctx* c, rt_ctx::DevBuf& b, size_t need) {
    if (need <= b.cap) return RT_OK;
    { int rc = drain(c); if (rc != RT_OK) return rc; }
    RT_HIP(hipStreamSynchronize(c->stream));
    void* np = nullptr;
    const size_t cap = need < 256 ? 256 : need;
    RT_HIP(hipMalloc(&np, cap));
    RT_HIP(hipMemsetAsync(np, 0, cap, c->stream));
    if (b.p && b.used) RT_HIP(hipMemcpyAsync(np, b.p, b.used, hipMemcpyDeviceToDevice, c->stream));
    RT_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(b.p);
    b.p = np;
    b.cap = cap;
    return RT_OK;
}

// writeBuffer(buffer, byte_offset, data): grows the device buffer when needed (keeping what is
// already there), copies, and returns once the caller's memory is no longer needed
static int write_buf(rt_ctx* c, rt_ctx::DevBuf& b, size_t byte_offset, const void* data, size_t bytes, const char* who) {
    RT_HIP(hipSetDevice(c->device));
    { int rc = drain(c); if (rc != RT_OK) return rc; }
    const size_t need = byte_offset + bytes;
    { int rc = grow_buf(c, b, need); if (rc != RT_OK) return rc; }
    if (bytes) {
        if (!data) return fail(RT_ERR_INVALID_ARG, who);
        RT_HIP(hipMemcpyAsync(static_cast<char*>(b.p) + byte_offset, data, bytes, hipMemcpyHostToDevice, c->stream));
        RT_HIP(hipStreamSynchronize(c->stream));
    }
    if (need > b.used) b.used = need;
    return RT_OK;
}

int rt_write_triangles(rt_ctx* c, const float* data, uint32_t n) {                    // RR:198-209
    if (!c || (n && !data)) return fail(RT_ERR_INVALID_ARG, "rt_write_triangles: NULL argument");
    c->d_tri.used = 0;
    c->corners_valid = false;
    int rc = write_buf(c, c->d_tri, 0, data, (size_t)n * 160u, "rt_write_triangles: NULL data");
    if (rc == RT_OK) c->scene_kind = 1;
    return rc;
}
// The same write into every version of a per-frame buffer (the static part of the scene, or an instance set too large
// to travel with a frame): drains, like every scene-setup call.
static int write_versions(rt_ctx* c, rt_ctx::DevBuf (&b)[kVersions], size_t byte_offset, const void* data, size_t bytes, const char* who) {
    for (int v = 0; v < kVersions; ++v) c->ver_gen[v] = 0;      // the next frame on each version re-applies the per-frame state
    for (int v = 0; v < kVersions; ++v) {
        int rc = write_buf(c, b[v], byte_offset, data, bytes, who);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}
// capacity for a per-frame write in every version, contents kept; drains only when something must grow
static int reserve_versions(rt_ctx* c, rt_ctx::DevBuf (&b)[kVersions], size_t bytes) {
    for (int v = 0; v < kVersions; ++v) {
        if (b[v].cap < bytes) c->ver_gen[v] = 0;
        int rc = grow_buf(c, b[v], bytes);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}

int rt_write_nodes(rt_ctx* c, size_t byte_offset, const float* data, uint32_t n) {    // RR:184-192, 212-223
    if (!c || (n && !data)) return fail(RT_ERR_INVALID_ARG, "rt_write_nodes: NULL argument");
    if (byte_offset % 32u) return fail(RT_ERR_INVALID_ARG, "rt_write_nodes: byte_offset must be a multiple of the 32-byte node");
    const size_t bytes = (size_t)n * 32u, end = byte_offset + bytes;
    const size_t head_bytes = (size_t)kHeadNodes * 32u;
    for (uint32_t i = 0; i < n; ++i) {              // u32(primitiveCount) as the kernel forms it (NaN and negatives: 0)
        const float f = data[8u * (size_t)i + 7u];
        const uint32_t cnt = !(f > 0.0f) ? 0u : (f >= 4294967040.0f ? 4294967295u : (uint32_t)f);
        c->node_count_max = std::max(c->node_count_max, cnt);
    }
    if (n) {                                        // the mirror the relinked copy of the BLAS trees is built from (rt_flow_build.h)
        if (c->h_nodes.size() < end / 4u) c->h_nodes.resize(end / 4u, 0.0f);
        std::memcpy(reinterpret_cast<char*>(c->h_nodes.data()) + byte_offset, data, bytes);
        if (!c->flow_dirty && end / 32u > c->flow.min_node) c->flow_dirty = true;      // the write reaches nodes the copy was built from
        if (end / 32u > c->flow.n_nodes) c->flow_dirty = true;                          // the buffer grew: the clamp-to-last-node rule moves
    }
    // the part of the write that falls into the head region updates the host's copy of it; frames carry that copy
    if (byte_offset < head_bytes && n) {
        const size_t hi = std::min(end, head_bytes);
        if (c->inst.head.size() < (size_t)kHeadNodes * 8u) c->inst.head.resize((size_t)kHeadNodes * 8u, 0.0f);
        std::memcpy(reinterpret_cast<char*>(c->inst.head.data()) + byte_offset, data, hi - byte_offset);
        c->inst.head_nodes = std::max(c->inst.head_nodes, (uint32_t)(hi / 32u));
        ++c->inst.gen;
    }
    int rc = RT_OK;
    if (end <= head_bytes) {                        // the per-frame TLAS write (RR:184-192): no drain, the next frame carries it
        RT_HIP(hipSetDevice(c->device));
        rc = reserve_versions(c, c->d_nodes, head_bytes);
    } else {
        rc = write_versions(c, c->d_nodes, byte_offset, data, bytes, "rt_write_nodes: NULL data");
    }
    if (rc == RT_OK) { c->scene_kind = 1; c->nodes_used = std::max(c->nodes_used, end); }
    return rc;
}
int rt_write_blas(rt_ctx* c, const float* data, uint32_t n) {                         // RR:169-174
    if (!c || (n && !data)) return fail(RT_ERR_INVALID_ARG, "rt_write_blas: NULL argument");
    if (n <= kInstMax) {
        RT_HIP(hipSetDevice(c->device));
        int rc = reserve_versions(c, c->d_blas, (size_t)kInstMax * 80u);
        if (rc != RT_OK) return rc;
        c->inst.blas.assign(data, data + 20u * (size_t)n);
        c->inst.blas_on = true;
        ++c->inst.gen;
        return RT_OK;
    }
    c->inst.blas_on = false;
    for (int v = 0; v < kVersions; ++v) c->d_blas[v].used = 0;
    return write_versions(c, c->d_blas, 0, data, (size_t)n * 80u, "rt_write_blas: NULL data");
}
int rt_write_tri_lookup(rt_ctx* c, const float* data, uint32_t n) {                   // RR:225-229
    if (!c || (n && !data)) return fail(RT_ERR_INVALID_ARG, "rt_write_tri_lookup: NULL argument");
    c->d_tri_lookup.used = 0;
    c->corners_valid = false;
    return write_buf(c, c->d_tri_lookup, 0, data, (size_t)n * 4u, "rt_write_tri_lookup: NULL data");
}
int rt_write_blas_lookup(rt_ctx* c, const float* data, uint32_t n) {                  // RR:177-181
    if (!c || (n && !data)) return fail(RT_ERR_INVALID_ARG, "rt_write_blas_lookup: NULL argument");
    if (n <= kInstMax) {
        RT_HIP(hipSetDevice(c->device));
        int rc = reserve_versions(c, c->d_blas_lookup, (size_t)kInstMax * 4u);
        if (rc != RT_OK) return rc;
        c->inst.lookup.assign(data, data + (size_t)n);
        c->inst.lookup_on = true;
        ++c->inst.gen;
        return RT_OK;
    }
    c->inst.lookup_on = false;
    for (int v = 0; v < kVersions; ++v) c->d_blas_lookup[v].used = 0;
    return write_versions(c, c->d_blas_lookup, 0, data, (size_t)n * 4u, "rt_write_blas_lookup: NULL data");
}
int rt_write_mesh_texture(rt_ctx* c, uint32_t w, uint32_t h, const uint8_t* rgba) {   // material.ts:61-65
    if (!c || !rgba) return fail(RT_ERR_INVALID_ARG, "rt_write_mesh_texture: NULL argument");
    if (w == 0 || h == 0 || w > 16384u || h > 16384u)
        return fail(RT_ERR_INVALID_ARG, "rt_write_mesh_texture: size must be in 1..16384");
    c->d_tex.used = 0;
    int rc = write_buf(c, c->d_tex, 0, rgba, (size_t)w * h * 4u, "rt_write_mesh_texture: NULL data");
    if (rc == RT_OK) { c->tex_w = w; c->tex_h = h; }
    return rc;
}

int rt_select_kernel(rt_ctx* c, int kernel) {
    if (!c) return fail(RT_ERR_INVALID_ARG, "rt_select_kernel: ctx is NULL");
    if (kernel == RT_KERNEL_RAYTRACER || kernel == RT_KERNEL_HEATMAP) { c->kernel = kernel; return RT_OK; }
    return fail(RT_ERR_INVALID_ARG, "rt_select_kernel: unknown kernel");
}

int rt_set_mode(rt_ctx* c, int mode) {
    if (!c) return fail(RT_ERR_INVALID_ARG, "rt_set_mode: ctx is NULL");
    if (mode != RT_MODE_FAST && mode != RT_MODE_STRICT) return fail(RT_ERR_INVALID_ARG, "rt_set_mode: unknown mode");
    c->mode = mode;
    return RT_OK;
}

int rt_set_variant(rt_ctx* c, int variant) {
    if (!c) return fail(RT_ERR_INVALID_ARG, "rt_set_variant: ctx is NULL");
    if (variant < 0 || variant > 6) return fail(RT_ERR_INVALID_ARG, "rt_set_variant: unknown variant");
    c->variant = variant;
    return RT_OK;
}

int rt_enqueue(rt_ctx* c, uint8_t* dst, hipStream_t s) {
    if (!c->W || !c->H) return fail(RT_ERR_STATE, "rt_render: rt_resize has not been called");
    if (!c->have_params) return fail(RT_ERR_STATE, "rt_render: rt_write_params has not been called");
    const bool tri = c->scene_kind == 1;
    if (!tri && !c->have_spheres) return fail(RT_ERR_STATE, "rt_render: rt_write_spheres has not been called");
    if (tri) {
        const bool have_blas = c->inst.blas_on ? !c->inst.blas.empty() : c->d_blas[0].used != 0;
        const bool have_lookup = c->inst.lookup_on ? !c->inst.lookup.empty() : c->d_blas_lookup[0].used != 0;
        if (!c->d_tri.used || !c->nodes_used || !have_blas || !c->d_tri_lookup.used || !have_lookup)
            return fail(RT_ERR_STATE, "rt_render: a triangle scene needs rt_write_triangles, _nodes, _blas, _tri_lookup and _blas_lookup");
        if (!c->d_tex.used) {   // meshTex is mandatory in the reference (RR:113-114); default: 1x1 white
            const uint8_t white[4] = {255, 255, 255, 255};
            int rc = rt_write_mesh_texture(c, 1, 1, white);
            if (rc != RT_OK) return rc;
        }
    } else if (c->kernel == RT_KERNEL_HEATMAP) {
        return fail(RT_ERR_UNSUPPORTED, "rt_render: the heatmap kernel counts BVH traversal steps; sphere scenes have no BVH");
    }
    for (int i = 0; i < 6; ++i)
        if (!c->d_face[i]) return fail(RT_ERR_STATE, "rt_render: all six cube map faces must be written first");
    RT_HIP(hipSetDevice(c->device));

    // ---- which kernels render this frame ----
    bool filter_ok = false;
    uint32_t signed_filter = 0;
    rt_plan(c->scene_bound, c->scene_min_radius, c->params, filter_ok, signed_filter);
    // Fast mode renders through the bounding-sphere hierarchy (rt_bvh.hip) from 128 spheres on (default,
    // variant 0) -- from 72 on for a caller that keeps frames in flight, where the hierarchy kernel's
    // end-of-frame tail is hidden: at 3840x2160 / 8 bounces the two forms cross at ~110 spheres one frame at a
    // time and at ~60 in flight (96 spheres: 1.33 vs 1.00 ms and 0.76 vs 0.98 ms; profiles/r02/count_sweep.log) --
    // or whenever variant 4 asks for it; variant 5 is the brute-force default, 1-3 its forms.
    bool own_stream = false;                       // one of the streams rt_render / rt_render_gather rotate over
    for (int k = 0; k < kStreams; ++k) own_stream = own_stream || s == c->streams[k];
    const bool hint = own_stream && c->pipelined_hint;
    const uint32_t bvh_from = hint ? 72u : 128u;
    const bool use_bvh = !tri && c->mode == RT_MODE_FAST && filter_ok && c->n > 0 &&
                         (c->variant == 4 || (c->variant == 0 && c->n >= bvh_from));
    // Frames in flight: the hierarchy kernel and the triangle kernels write nothing but their
    // own control block and `dst`, so consecutive frames may overlap on the device (the next
    // frame's workgroups start while the last paths of this one finish), and so may the
    // single-kernel brute-force and literal forms; the two-kernel brute-force pipeline shares one
    // path queue and is serialised behind the frames in flight.
    // (rt_kernels.hip: launch_fast -- variants 2 and 3 force the pipeline, 0 / 5 take it from 320 spheres on)
    const bool queue_pipeline = !tri && !use_bvh && c->mode == RT_MODE_FAST && filter_ok &&
                                (c->variant == 2 || c->variant == 3 ||
                                 ((c->variant == 0 || c->variant == 4 || c->variant == 5) && c->n >= 320u));
    const bool overlap_ok = !queue_pipeline;
    const bool need_prep = !tri && c->n && (!c->prep_spheres_valid || (!use_bvh && !c->prep_params_valid));
    const bool need_bvh = use_bvh && !c->bvh_valid;

    if (queue_pipeline) { int rc = ensure_queue(c); if (rc != RT_OK) return rc; }
    bool sky_flat = true;       // six 1x1 faces of one colour
    for (int i = 0; i < 6; ++i)
        if (c->fw[i] != 1u || c->fh[i] != 1u || c->face_texel0[i] != c->face_texel0[0]) sky_flat = false;
    const bool resolve_pass = use_bvh && !sky_flat;      // rt_bvh.hip: sky_resolve
    if (resolve_pass) { int rc = ensure_fin(c); if (rc != RT_OK) return rc; }
    // The hierarchy after rt_write_spheres.  A changed sphere count (or the first frame): host build, here and
    // now.  The same count: the topology on the device stays, its node bounds are refitted there (below);
    // a topology the worker thread has finished meanwhile is taken first.  Either way the device arrays may be
    // rewritten or reallocated: nothing may be in flight (rt_write_spheres has drained already).
    const bool refit = need_bvh && c->bvh_topo_n == c->n && c->bvh_topo_n != 0u;
    bool upload_topology = false;
    if (need_bvh) {
        int rc = drain(c);
        if (rc != RT_OK) return rc;
        uint32_t nodes = c->bvh_nodes;
        if (!refit) {
            nodes = rt_bvh_build(c->h_records.data(), c->n, c->h_bvh_rec, c->h_bvh_link, rt_bvh_arity());
            upload_topology = true;
        } else if (c->rebuild) {
            std::lock_guard<std::mutex> lk(c->rebuild->m);
            if (c->rebuild->ready && c->rebuild->out_n == c->n) {
                c->h_bvh_rec.swap(c->rebuild->out_rec);
                c->h_bvh_link.swap(c->rebuild->out_link);
                nodes = c->rebuild->out_nodes;
                upload_topology = true;
            }
            c->rebuild->ready = false;
        }
        if (nodes + 1u > c->bvh_cap) {
            (void)hipFree(c->d_bvh_rec); (void)hipFree(c->d_bvh_link);
            c->d_bvh_rec = nullptr; c->d_bvh_link = nullptr; c->bvh_cap = 0;
            RT_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_bvh_rec), ((size_t)nodes + 1u) * sizeof(float4)));
            RT_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_bvh_link), ((size_t)nodes + 1u) * sizeof(uint32_t)));
            c->bvh_cap = nodes + 1u;
        }
        c->bvh_nodes = nodes;
    }
    // ---- what the triangle kernels read beside the reference's buffers: built here, BEFORE a slot of the event ring is
    // taken (both may have to drain the frames in flight, which empties the ring) ----
    if (tri && !c->corners_valid) {
        // the compact corner array follows the triangles and the lookup table
        { int rc = drain(c); if (rc != RT_OK) return rc; }
        const uint32_t n_slots = (uint32_t)(c->d_tri_lookup.used / 4u);
        { int rc = grow_buf(c, c->d_corners, (size_t)n_slots * 48u); if (rc != RT_OK) return rc; }
        if (c->scene_stream && c->scene_stream != s) RT_HIP(hipStreamWaitEvent(s, c->ev_scene, 0));
        RT_HIP(rt_launch_tri_corners(static_cast<float4*>(c->d_corners.p), static_cast<const float*>(c->d_tri.p),
                                     static_cast<const float*>(c->d_tri_lookup.p), n_slots, (uint32_t)(c->d_tri.used / 160u), s));
        RT_HIP(hipEventRecord(c->ev_scene, s));          // frames on other streams wait for it (the scene-update event)
        c->scene_stream = s;
        c->corners_valid = true;
    }
    // The relinked pair records of the BLAS trees (rt_flow_build.h), for scenes whose instance data travels with the frame (up
    // to 16 instances) and whose indices fit 16 bits: rebuilt when a write has reached the nodes the copy was made from or a
    // frame names a root it does not know -- from the union of the roots it knows and the new ones, so that a host that
    // alternates root sets between frames pays for each root once.
    uint32_t root_meta[kInstMax] = {0};
    bool have_pairs = false;       // the relinked copy is current and covers this frame's roots
    if (tri && c->kernel != RT_KERNEL_HEATMAP && c->variant != 6) {
        const uint32_t n_nodes = (uint32_t)(c->nodes_used / 32u);
        const uint32_t n_inst = (uint32_t)(c->inst.blas.size() / 20u);
        const bool fits = c->inst.blas_on && c->inst.lookup_on && n_inst >= 1u && n_inst <= kInstMax && !c->inst.lookup.empty() &&
                          c->inst.lookup.size() <= kInstMax && n_nodes >= 1u && n_nodes <= 65536u && c->d_tri_lookup.used / 4u <= 65536u &&
                          c->node_count_max <= 65535u && c->h_nodes.size() / 8u >= n_nodes;
        if (fits) {
            uint32_t roots[kInstMax];
            for (uint32_t i = 0; i < n_inst; ++i) roots[i] = rt_flow_u32f(c->inst.blas[20u * i + 16u]);
            // (the per-frame head of the node buffer lives in inst.head until a frame carries it: the mirror has it already)
            const bool stale = c->flow_dirty || c->flow.n_nodes != n_nodes;
            const bool need = stale || !rt_flow_covers(c->flow, roots, n_inst);
            ++c->flow_frames_since;
            // A host that invalidates the copy with every frame (it rewrites BLAS nodes per frame: nothing says it may not) would
            // pay a drain, a host-side rebuild of every tree and an upload per frame for records that live one frame.  Four frames
            // in a row that each had to rebuild: the node walk (which reads the reference's buffer as written) for the next sixty,
            // then another try.
            if (need && c->flow_cooldown > 0u) {
                --c->flow_cooldown;
            } else if (need) {
                c->flow_streak = c->flow_frames_since <= 1u ? c->flow_streak + 1u : 1u;
                c->flow_frames_since = 0u;
                if (c->flow_streak >= 4u) { c->flow_cooldown = 60u; c->flow_streak = 0u; }
                { int rc = drain(c); if (rc != RT_OK) return rc; }
                std::vector<uint32_t> all(roots, roots + n_inst);
                if (!stale) all.insert(all.end(), c->flow.roots.begin(), c->flow.roots.end());   // roots already known stay known
                rt_flow_build(c->h_nodes.data(), n_nodes, all.data(), (uint32_t)all.size(), c->flow);
                c->flow_dirty = false;
                ++c->stats.pair_rebuilds;
                if (c->flow.ok && c->flow.n_pairs) {
                    int rc = write_buf(c, c->d_flow, 0, c->flow.pairs.data(), (size_t)c->flow.n_pairs * 64u, "flow pairs");
                    if (rc != RT_OK) return rc;
                }
            }
            // usable: current (no write has reached it, the same node count) and knowing every root of this frame
            have_pairs = !c->flow_dirty && c->flow.ok && c->flow.n_pairs != 0u && c->flow.n_nodes == n_nodes && rt_flow_covers(c->flow, roots, n_inst);
            if (have_pairs) {
                const uint32_t last = n_nodes - 1u;
                for (uint32_t i = 0; i < n_inst; ++i)
                    root_meta[i] = rt_flow_meta(c->h_nodes.data(), n_nodes, roots[i] < last ? roots[i] : last, c->flow.pair_of);
            }
        }
    }
    if (c->in_flight == RT355_MAX_IN_FLIGHT) {   // event ring full: drain
        int rc = rt_wait(c);
        if (rc != RT_OK) return rc;
    }
    const uint32_t slot = c->in_flight;

    // ---- ordering against the frames in flight and against the last scene update ----
    if (need_prep || need_bvh || !overlap_ok) {
        for (uint32_t i = 0; i < c->in_flight; ++i) RT_HIP(hipStreamWaitEvent(s, c->ev_k1[i], 0));
    }
    if (c->scene_stream && c->scene_stream != s) RT_HIP(hipStreamWaitEvent(s, c->ev_scene, 0));

    RtFrameArgs fa;
    std::memcpy(fa.p, c->params, sizeof fa.p);
    fa.W = c->W; fa.H = c->H; fa.N = c->n; fa.N16 = c->n16;
    fa.tile_first = c->rank; fa.tile_step = c->world; fa.n_local_tiles = local_tiles(c);
    fa.signed_filter = signed_filter;
    {
        const float4* b = c->d_scene;
        const uint32_t m = c->n16;
        fa.geo = b; fa.lgt = b + m; fa.cam = b + 2u * m; fa.col = b + 3u * m;
        fa.geo_f = b + 4u * m; fa.lgt_f = b + 5u * m; fa.cam_f = b + 6u * m;
        const float* w = reinterpret_cast<const float*>(b + 7u * m);
        fa.geo_w = w; fa.lgt_w = w + m; fa.cam_w = w + 2u * m;
    }

    RT_HIP(hipEventRecord(c->ev_prep0[slot], s));
    if (need_prep) {
        RtPrepArgs pa;
        std::memcpy(pa.p, c->params, sizeof pa.p);
        pa.N = c->n; pa.N16 = c->n16;
        pa.records = c->d_records;
        float4* b = c->d_scene;
        const uint32_t m = c->n16;
        pa.geo = b; pa.lgt = b + m; pa.cam = b + 2u * m; pa.col = b + 3u * m;
        pa.geo_f = b + 4u * m; pa.lgt_f = b + 5u * m; pa.cam_f = b + 6u * m;
        float* w = reinterpret_cast<float*>(b + 7u * m);
        pa.geo_w = w; pa.lgt_w = w + m; pa.cam_w = w + 2u * m;
        RT_HIP(rt_launch_prep(pa, s));
        c->prep_spheres_valid = true;
        c->prep_params_valid = true;
    }
    if (need_bvh) {
        if (upload_topology) {
            const size_t nn = (size_t)c->bvh_nodes + 1u;
            RT_HIP(hipMemcpyAsync(c->d_bvh_rec, c->h_bvh_rec.data(), nn * sizeof(float4), hipMemcpyHostToDevice, s));
            RT_HIP(hipMemcpyAsync(c->d_bvh_link, c->h_bvh_link.data(), nn * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            RT_HIP(hipStreamSynchronize(s));   // pageable sources: the vectors may be rebuilt later
            c->bvh_topo_n = c->n;
        }
        // bounds of the inner nodes for the current positions (a topology from the worker was built for older ones)
        if (refit) RT_HIP(rt_launch_bvh_refit(c->d_bvh_rec, c->d_bvh_link, c->bvh_nodes, c->d_records, s));
        RT_HIP(rt_launch_bvh_fill(c->d_bvh_rec, c->d_bvh_link, c->bvh_nodes, fa.geo_f, s));
        c->bvh_valid = true;
    }
    if (need_prep || need_bvh) {
        RT_HIP(hipEventRecord(c->ev_scene, s));
        c->scene_stream = s;
    }
    fa.bvh_rec = use_bvh ? c->d_bvh_rec : nullptr;
    fa.bvh_link = use_bvh ? c->d_bvh_link : nullptr;
    fa.bvh_nodes = use_bvh ? c->bvh_nodes : 0u;
    // frames in flight on DIFFERENT streams run concurrently and share the chip: this frame's grid is
    // 1 / (number of distinct streams among it and the kStreams-1 frames enqueued before it)
    c->slot_stream[slot] = s;
    fa.grid_share = 1u;
    if (overlap_ok) {
        hipStream_t seen[kStreams] = {s};
        uint32_t distinct = 1;
        for (uint32_t back = 1; back < (uint32_t)kStreams && back <= slot; ++back) {
            hipStream_t o = c->slot_stream[slot - back];
            bool known = false;
            for (uint32_t j = 0; j < distinct; ++j) known = known || seen[j] == o;
            if (!known) seen[distinct++] = o;
        }
        fa.grid_share = distinct;
        // a caller that has been enqueuing frames back to back through rt_render / rt_render_gather (the library's
        // own rotation over kStreams streams) gets equal shares from the first frame of a batch on (rt_bvh.hip:
        // launch_bvh_as); one that waits after every frame keeps the whole chip, and frames a host enqueues on its
        // own stream(s) through rt_render_to share the chip by the streams actually in use, whatever the history
        if (hint) fa.grid_share = (uint32_t)kStreams;
    }

    // this frame's partial ray counters and its 32-byte control block: one of RT355_MAX_IN_FLIGHT
    // sets, all zeroed by rt_create and again by every frame's epilogue kernel (no memset or read-back by the host)
    unsigned long long* counters = c->d_rays + (kCtrlBytes / 8u) * slot;
    unsigned long long* ctrl = counters + kCounterBytes / 8u;
    for (int i = 0; i < 6; ++i) { fa.face[i] = c->d_face[i]; fa.fw[i] = c->fw[i]; fa.fh[i] = c->fh[i]; }
    fa.sky_flat = sky_flat ? 1u : 0u;
    fa.sky_seamless = 1u;   // six equal squares: what WebGPU accepts as a cube texture (CM:35-79: 512 x 512 x 6)
    for (int i = 0; i < 6; ++i)
        if (c->fw[i] != c->fw[0] || c->fh[i] != c->fw[0]) fa.sky_seamless = 0u;
    fa.out = dst;
    // one record buffer per frame that may be running: the frame kStreams slots back must be through with this one
    fa.fin = resolve_pass ? c->d_fin[slot % (uint32_t)kStreams] : nullptr;
    if (resolve_pass && slot >= (uint32_t)kStreams) RT_HIP(hipStreamWaitEvent(s, c->ev_k1[slot - (uint32_t)kStreams], 0));
    fa.rays = counters;
    fa.queue = queue_pipeline ? c->d_queue : nullptr;   // rt_kernels.hip takes the pipeline only with a queue
    fa.qctrl = reinterpret_cast<uint32_t*>(ctrl) + 2;
    fa.queue_cap = queue_pipeline ? (uint32_t)c->queue_cap : 0u;
    RtLaunchCfg cfg;
    cfg.mode = filter_ok ? c->mode : (int)RT_MODE_STRICT;
    cfg.variant = (c->variant == 4 || c->variant == 5) ? 0 : c->variant;   // rt_kernels.hip numbers its default 0

    const uint32_t v = slot % (uint32_t)kVersions;
    RtTriScene ts;
    std::memset(&ts, 0, sizeof ts);
    if (tri) {
        ts.nodes = static_cast<const float4*>(c->d_nodes[v].p);
        ts.blas = static_cast<const float*>(c->d_blas[v].p);
        ts.tri = static_cast<const float*>(c->d_tri.p);
        ts.corners = static_cast<const float4*>(c->d_corners.p);
        ts.tri_lookup = static_cast<const float*>(c->d_tri_lookup.p);
        ts.blas_lookup = static_cast<const float*>(c->d_blas_lookup[v].p);
        ts.tex = static_cast<const uint8_t*>(c->d_tex.p);
        ts.n_nodes = (uint32_t)(c->nodes_used / 32u);
        ts.n_blas = c->inst.blas_on ? (uint32_t)(c->inst.blas.size() / 20u) : (uint32_t)(c->d_blas[v].used / 80u);
        ts.n_tri = (uint32_t)(c->d_tri.used / 160u);
        ts.n_tri_lookup = (uint32_t)(c->d_tri_lookup.used / 4u);
        ts.n_blas_lookup = c->inst.lookup_on ? (uint32_t)c->inst.lookup.size() : (uint32_t)(c->d_blas_lookup[v].used / 4u);
        ts.tex_w = c->tex_w; ts.tex_h = c->tex_h;
        ts.packed_ok = (c->node_count_max <= 65535u && ts.n_nodes <= 65536u && ts.n_tri_lookup <= 65536u) ? 1u : 0u;
        ts.tile_order = nullptr; ts.tile_cost = nullptr; ts.xcd_rows = 0u; ts.prio = 0u; ts.in_flight = hint ? 1u : 0u; ts.dbg = nullptr;
        // the top-level tree this frame walks (the mirror holds every node write, per-frame heads included): small enough for the
        // four-slot TLAS stack?  (rt_tlas_fit.h; the same constants as the kernel's: rt_tri_device.h kSmallStack / kSmallNodes)
        ts.tlas_small = 0u;
        if (c->h_nodes.size() / 8u >= ts.n_nodes) {
            if (ts.n_blas <= 4u && rt_tlas_fits(c->h_nodes.data(), ts.n_nodes, 3u, 8u)) ts.tlas_small = 2u;
            else if (rt_tlas_fits(c->h_nodes.data(), ts.n_nodes, 4u, 16u)) ts.tlas_small = 1u;
            else if (rt_tlas_fits(c->h_nodes.data(), ts.n_nodes, 8u, 24u)) ts.tlas_small = 3u;
            else if (rt_tlas_fits(c->h_nodes.data(), ts.n_nodes, 8u, 32u)) ts.tlas_small = 4u;
        }
        // two-byte stack entries (count << 14 | x): every meta of the records and of this frame's roots must fit them
        ts.p16_ok = (have_pairs && c->flow.max_count <= 3u && c->flow.max_x <= 16383u) ? 1u : 0u;
        for (uint32_t i = 0; i < kInstMax && ts.p16_ok; ++i)
            if ((root_meta[i] >> 16) > 3u || (root_meta[i] & 0xFFFFu) > 16383u) ts.p16_ok = 0u;
        // (13-16 instances: only the form that stages sixteen records can walk the pair records -- the others find a root's meta in
        // one of TWELVE staged records --, i.e. only a frame whose tree passed a walk above)
        ts.pairs = (have_pairs && (ts.n_blas <= 12u || ts.tlas_small != 0u)) ? static_cast<const float4*>(c->d_flow.p) : nullptr;
        for (uint32_t i = 0; i < kInstMax; ++i) ts.root_meta[i] = root_meta[i];
        // which form of the kernel renders the frame (rt_triangles.hip); the small forms take the frame's instance data in their
        // own arguments, in the layout they stage it in
        ts.form = (uint32_t)rt_tri_stack_form(ts, c->kernel == RT_KERNEL_HEATMAP);
        if (ts.form != 0u) {
            const uint32_t nn = std::min(ts.n_nodes, kInstHeadNodes), nb = std::min(ts.n_blas, kInstBlas);
            std::memcpy(ts.inst.head, c->h_nodes.data(), (size_t)nn * 32u);
            std::memcpy(ts.inst.blas, c->inst.blas.data(), (size_t)nb * 80u);
            const uint32_t nl = std::min(ts.n_blas_lookup, nb);
            for (uint32_t k = 0; k < nb; ++k) {
                if (k < nl) ts.inst.blas[20u * k + 19u] = c->inst.lookup[k];
                std::memcpy(&ts.inst.blas[20u * k + 17u], &root_meta[k], 4);
            }
            if (c->inst_gen_carried != c->inst.gen) { c->inst_gen_carried = c->inst.gen; ++c->stats.instance_uploads; }
        }
    }
    // Per-frame instance data (RR:169-192) travels with the frame: version `v` of the three buffers is brought to the
    // host's current state by a one-workgroup kernel whose kernarg block holds the values, in front of the ray-trace
    // kernel on the frame's stream.  The frame kVersions slots back read the same version: it must be through.
    // Ordering does not lean on which stream a slot happens to use (a host may rotate rt_render_to over any number of its own
    // streams): before a version is rewritten, EVERY frame in flight that reads it must be through; and a frame that reads a
    // version another stream brought up to date waits for that update.
    // (The small forms of the triangle kernel -- the reference's scene, any scene of a few instances -- read none of the three
    // buffers: their instance data is in the kernel's own arguments, ts.inst above.  No kernel, no event, no wait; the versions
    // simply fall behind, and a later frame of another form brings its own up to date.)
    if (tri && ts.form == 0u && c->ver_gen[v] != c->inst.gen) {
        for (uint32_t i = v; i < slot; i += (uint32_t)kVersions) RT_HIP(hipStreamWaitEvent(s, c->ev_k1[i], 0));
        RtInstanceArgs ia;
        ia.nodes = static_cast<float*>(c->d_nodes[v].p);
        ia.blas = static_cast<float*>(c->d_blas[v].p);
        ia.lookup = static_cast<float*>(c->d_blas_lookup[v].p);
        ia.n_head_f = ia.nodes ? c->inst.head_nodes * 8u : 0u;
        ia.n_blas_f = (c->inst.blas_on && ia.blas) ? (uint32_t)c->inst.blas.size() : 0u;
        ia.n_lookup_f = (c->inst.lookup_on && ia.lookup) ? (uint32_t)c->inst.lookup.size() : 0u;
        if (ia.n_head_f) std::memcpy(ia.data, c->inst.head.data(), ia.n_head_f * sizeof(float));
        if (ia.n_blas_f) std::memcpy(ia.data + 31 * 8, c->inst.blas.data(), ia.n_blas_f * sizeof(float));
        if (ia.n_lookup_f) std::memcpy(ia.data + 31 * 8 + 16 * 20, c->inst.lookup.data(), ia.n_lookup_f * sizeof(float));
        RT_HIP(rt_launch_apply_instances(ia, s));
        if (!c->ev_ver[v]) RT_HIP(hipEventCreateWithFlags(&c->ev_ver[v], hipEventDisableTiming));
        RT_HIP(hipEventRecord(c->ev_ver[v], s));
        c->ver_stream[v] = s;
        c->ver_gen[v] = c->inst.gen;
        ++c->stats.instance_uploads;
    } else if (tri && ts.form == 0u && c->ev_ver[v] && c->ver_stream[v] != s) {
        RT_HIP(hipStreamWaitEvent(s, c->ev_ver[v], 0));
    }
    // tile order of the triangle kernel: only for frames on the library's own streams (each has its set of buffers)
    int order_set = -1;
    const uint32_t order_n = ((c->W + 7u) / 8u) * fa.n_local_tiles;
    // ... and only for a caller that waits after each frame (the reference's loop): with frames in flight the tiles of
    // the next frame fill the slots a long tile leaves idle anyway, and row-major order keeps neighbours in one L2
    if (tri && c->kernel != RT_KERNEL_HEATMAP && order_n >= kOrderMinTiles && !hint) {
        for (int k = 0; k < kStreams; ++k) if (s == c->streams[k]) order_set = k;
        if (order_set >= 0 && c->d_tile_cost[order_set].cap < (size_t)order_n * 4u) {
            // only frames on this stream use the set: wait for them, not for the batch (no slot bookkeeping involved)
            RT_HIP(hipStreamSynchronize(s));
            const size_t cap = ((size_t)order_n * 4u + 7u) & ~(size_t)7u, scan_bytes = (size_t)rt_order_scan_words() * 4u;
            for (rt_ctx::DevBuf* b : {&c->d_tile_cost[order_set], &c->d_tile_order[order_set]}) {
                (void)hipFree(b->p);
                b->p = nullptr; b->cap = 0;
                RT_HIP(hipMalloc(&b->p, cap + scan_bytes));                // the list carries two header words; behind the costs: order_tiles' bins
                b->cap = cap;
            }
            RT_HIP(hipMemsetAsync(c->d_tile_cost[order_set].p, 0, cap + scan_bytes, s));   // tiles leave their times by atomicMax
            c->order_tiles[order_set] = 0;
        }
    }
    RT_HIP(hipEventRecord(c->ev_k0[slot], s));
    g_rt_kernel_id = RT_KID_NONE;
    if (tri) {
#ifdef RT355_DEV_EXPORTS
        if (getenv("RT355_TRI_TIMELINE")) {
            const size_t words = 3u * ((size_t)order_n + 3u * 8192u + 15u * 256u + 64u + 8u * ((c->W + 7u) / 8u));
            if (c->d_tri_dbg.cap < words * 8u) { (void)hipFree(c->d_tri_dbg.p); c->d_tri_dbg.p = nullptr; RT_HIP(hipMalloc(&c->d_tri_dbg.p, words * 8u)); c->d_tri_dbg.cap = words * 8u; }
            RT_HIP(hipMemsetAsync(c->d_tri_dbg.p, 0, c->d_tri_dbg.cap, s));
            ts.dbg = static_cast<unsigned long long*>(c->d_tri_dbg.p);
        }
#endif
#ifdef RT355_DEV_EXPORTS
        if (order_set >= 0 && getenv("RT355_KEEP_TILE_COST")) RT_HIP(hipMemsetAsync(c->d_tile_cost[order_set].p, 0, (size_t)order_n * 4u, s));
#endif
        if (order_set >= 0) {
            ts.tile_cost = static_cast<uint32_t*>(c->d_tile_cost[order_set].p);
            if (c->order_tiles[order_set] == order_n) {
                ts.tile_order = static_cast<const uint32_t*>(c->d_tile_order[order_set].p);
                // does that list split tiles?  Then the kernel whose parts' idle lanes trace ahead (the word may be a frame old)
                ts.roles = __atomic_load_n(&c->h_split[order_set], __ATOMIC_RELAXED) != 0ull ? 1u : 0u;
            }
#ifdef RT355_DEV_EXPORTS
            if (getenv("RT355_TRI_NOLIST")) ts.tile_order = nullptr;
#endif
        }
        RT_HIP(rt_launch_triangles(fa, ts, c->kernel == RT_KERNEL_HEATMAP, s));
    } else {
        if (use_bvh) RT_HIP(rt_launch_bvh(fa, s));
        else RT_HIP(rt_launch_trace(fa, cfg, s));
    }
    RT_HIP(hipEventRecord(c->ev_k1[slot], s));
#ifdef RT355_DEV_EXPORTS
    if (order_set >= 0 && getenv("RT355_KEEP_TILE_COST")) {      // tools/tile_cost_probe.py reads the times of whole tiles in index order
        c->order_tiles[order_set] = 0;
        order_set = -1;
    }
#endif
    // The end of the frame: its counters summed into host-visible memory and zeroed for the slot's next frame (rt_assemble.hip:
    // frame_epilogue), then the event rt_wait waits for -- as early as can be: an awaited frame's successor is launched by a host
    // that has to wake up first, and its turnaround (15-30 us) is what the work-list kernels behind the event hide in.  (Round 5
    // ran the epilogue inside order_hist's first workgroup to save a link of the chain: the host woke 11 us later and the frame
    // period GREW by as much -- TRI 0.276 -> 0.298 ms; profiles/r05/ref_awaited_trace.log.)
    RT_HIP(rt_launch_frame_epilogue(counters, c->h_rays + 2u * slot, (uint32_t)(kCtrlBytes / 8u), s));
    RT_HIP(hipEventRecord(c->ev_done[slot], s));
    if (order_set >= 0) {
        // after the frame's event: rt_wait does not wait for it, the stream's next frame does
        uint32_t* const cost = static_cast<uint32_t*>(c->d_tile_cost[order_set].p);
        uint32_t* const scan = reinterpret_cast<uint32_t*>(static_cast<char*>(c->d_tile_cost[order_set].p) + c->d_tile_cost[order_set].cap);
        uint32_t* const list = static_cast<uint32_t*>(c->d_tile_order[order_set].p);
        RT_HIP(rt_launch_order_hist(cost, scan, list, order_n, c->wave_slots, nullptr, nullptr, 0u, c->h_split + order_set, s));
        RT_HIP(rt_launch_order_scatter(cost, scan, list, order_n, s));
        c->order_tiles[order_set] = order_n;
    }
    c->stats.kernel_id = (uint32_t)g_rt_kernel_id;
    if (tri) c->stats.tri_form = (uint32_t)g_rt_tri_form;
    c->stats.grid_share = fa.grid_share;
    c->in_flight = slot + 1;
    return RT_OK;
}

#ifdef RT355_DEV_EXPORTS
// development builds only (tools/tri_timeline.py): {start, end, tile << 8 | part} of every workgroup of the last triangle frame
__attribute__((visibility("default"))) long long rt_debug_tri_timeline(rt_ctx* c, unsigned long long* dst, size_t cap_words) {
    if (!c || !c->d_tri_dbg.p) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    const size_t n = std::min(cap_words, c->d_tri_dbg.cap / 8u);
    if (hipMemcpy(dst, c->d_tri_dbg.p, n * 8u, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (long long)n;
}
// development builds only (tools/tile_cost_probe.py): the per-tile times of the last frame rendered on stream `set`
__attribute__((visibility("default"))) int rt_debug_tile_cost(rt_ctx* c, int set, uint32_t* dst, uint32_t cap) {
    if (!c || set < 0 || set >= kStreams || !c->d_tile_cost[set].p) return -1;
    const uint32_t n = std::min(cap, (uint32_t)(c->d_tile_cost[set].cap / 4u));
    if (hipStreamSynchronize(c->streams[set]) != hipSuccess) return -1;
    if (hipMemcpy(dst, c->d_tile_cost[set].p, (size_t)n * 4u, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)n;
}
#endif

// Colour buffer k is about to be written by a frame on stream `st`.  A frame of the current batch that renders into the same
// buffer on ANOTHER stream must be through first (a streaming read-back of it orders itself: ev_copy).  Streams and buffers
// rotate independently since awaited frames stay on one stream: within a run of rt_render calls their offset is constant and
// frame i + 4 follows frame i on its stream anyway, but an awaited frame in between shifts the offset, and rt_render_gather /
// rt_group_render pair stream k with buffer k whatever rt_render did before.
int rt_order_colour_buffer(rt_ctx* c, uint32_t k, hipStream_t st) {
    const int before = c->buf_slot[k];
    if (before >= 0 && (uint32_t)before < c->in_flight && c->slot_stream[before] != st) RT_HIP(hipStreamWaitEvent(st, c->ev_k1[before], 0));
    return RT_OK;
}

int rt_render(rt_ctx* c) {
    if (!c) return fail(RT_ERR_INVALID_ARG, "rt_render: ctx is NULL");
    // consecutive frames rotate over kStreams streams and colour buffers, so that frames enqueued
    // back to back overlap on the device; rt_read_pixels returns the latest one
    const uint32_t k = c->frames_rendered % (uint32_t)kStreams;
    uint8_t* dst = c->d_outs[k];
    // The stream: the next of the four while frames are in flight (they overlap); the SAME one for a frame that follows an
    // awaited frame -- the reference's loop.  The work list of an awaited triangle frame is made from the previous frame of
    // its stream: rotating regardless, that was the frame four steps back, and with the instances turning (the reference's
    // scene spins one of its meshes) a list that old fitted the picture badly: 0.56 ms per frame against 0.38 with nothing
    // moving (profiles/r04/loop_breakdown.log).
    if (c->in_flight != 0u) c->stream_rot = (c->stream_rot + 1u) % (uint32_t)kStreams;
    hipStream_t st = c->streams[c->stream_rot];
    // a streaming read-back (rt_read_pixels_async) may still be copying the frame this buffer holds
    if (c->copy_pending[k]) RT_HIP(hipStreamWaitEvent(st, c->ev_copy[k], 0));
    { int rc = rt_order_colour_buffer(c, k, st); if (rc != RT_OK) return rc; }
    int rc = rt_enqueue(c, dst, st);
#ifdef RT355_DEV_EXPORTS
    // development builds (tools/root_probe.py): what the ROOT of a group of RT355_DEV_ROOT_WORLD ranks runs behind its share of the
    // frame -- the de-interleave of the whole gathered frame, on the frame's own stream, moving its end event as rt_render_gather does
    if (rc == RT_OK)
        if (const char* e = getenv("RT355_DEV_ROOT_WORLD")) {
            static uint8_t* dev_gather[kStreams] = {nullptr};
            static uint8_t* dev_frame[kStreams] = {nullptr};
            const uint32_t world = (uint32_t)atoi(e);
            const size_t gb = (size_t)world * rt_padded_tiles(c->H, world) * 8u * c->W * 4u, fb = (size_t)c->H * c->W * 4u;
            if (!dev_gather[k]) { RT_HIP(hipMalloc(reinterpret_cast<void**>(&dev_gather[k]), gb)); RT_HIP(hipMalloc(reinterpret_cast<void**>(&dev_frame[k]), fb)); }
            RT_HIP(rt_launch_assemble(dev_gather[k], dev_frame[k], c->W, c->H, world, rt_padded_tiles(c->H, world), st));
            RT_HIP(hipEventRecord(c->ev_k1[c->in_flight - 1u], st));
        }
#endif
    if (rc == RT_OK) { c->d_out = dst; c->buf_slot[k] = (int)c->in_flight - 1; ++c->frames_rendered; }
    return rc;
}

int rt_render_to(rt_ctx* c, void* device_dst, size_t cap, void* hip_stream) {
    if (!c || !device_dst) return fail(RT_ERR_INVALID_ARG, "rt_render_to: NULL argument");
    const size_t need = (size_t)local_tiles(c) * 8u * c->W * 4u;
    if (cap < need) return fail(RT_ERR_CAPACITY, "rt_render_to: destination smaller than local_tiles*8*W*4");
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    return rt_enqueue(c, static_cast<uint8_t*>(device_dst), s);
}

// forget the frames in flight (their work has left the device: a communicator was aborted): counters zeroed, ring empty
void rt_abandon_in_flight(rt_ctx* c) {
    for (int k = 0; k < kStreams; ++k)
        if (c->streams[k]) (void)hipStreamSynchronize(c->streams[k]);
    if (c->in_flight) {
        (void)hipMemsetAsync(c->d_rays, 0, kCtrlBytes * c->in_flight, c->stream);
        (void)hipStreamSynchronize(c->stream);
    }
    (void)hipGetLastError();
    c->in_flight = 0;
    c->stats.batch_frames = 0;
    for (int v = 0; v < kVersions; ++v) c->ver_gen[v] = 0;
}

int rt_wait(rt_ctx* c) {
    if (!c) return fail(RT_ERR_INVALID_ARG, "rt_wait: ctx is NULL");
    RT_HIP(hipSetDevice(c->device));
    if (c->comm && c->in_flight) {
        // frames that end in an RCCL exchange: a poll that notices a failed peer or a missed deadline (rt_comm.hip)
        int rc = rt_comm_wait_frames(c);
        if (rc != RT_OK) return rc;
    }
    // one synchronisation per frame, on the event behind its epilogue kernel: ray count and fault word are in host memory
    // by then, the counter set is zero again (rt_assemble.hip: frame_epilogue)
    for (uint32_t i = 0; i < c->in_flight; ++i) RT_HIP(hipEventSynchronize(c->ev_done[i]));
    if (c->in_flight) {
        c->stats.frames += c->in_flight;
        c->stats.rays = c->h_rays[2u * (c->in_flight - 1u)];       // of the latest frame
        c->stats.batch_frames = c->in_flight;
        // Frames in flight?  Only the library's own rotation counts: a host that enqueues several frames on ONE stream
        // of its own (rt_render_to) has them serialised by that stream, and must not be given a quarter of the chip.
        {
            uint32_t distinct = 0;
            for (int k = 0; k < kStreams; ++k) {
                bool used = false;
                for (uint32_t i = 0; i < c->in_flight; ++i) used = used || c->slot_stream[i] == c->streams[k];
                distinct += used ? 1u : 0u;
            }
            c->pipelined_hint = distinct > 1u;
        }
        c->stats.batch_kernel_ms = 0.0f;
        for (uint32_t i = 0; i < c->in_flight; ++i) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, c->ev_k0[i], c->ev_k1[i]) == hipSuccess) {
                c->stats.kernel_ms = ms;
                c->stats.batch_kernel_ms += ms;
            }
            if (hipEventElapsedTime(&ms, c->ev_prep0[i], c->ev_k0[i]) == hipSuccess) c->stats.prep_ms = ms;
        }
        (void)hipGetLastError();
        c->in_flight = 0;
        for (int k = 0; k < kStreams; ++k) c->buf_slot[k] = -1;      // every frame of the batch is complete
        // a kernel that could not run as planned says so in the word behind its first ray counter (rt_device.h: report_fault)
        unsigned long long fault = 0ull;
        for (uint32_t i = 0; i < c->stats.batch_frames; ++i) fault = std::max(fault, c->h_rays[2u * i + 1u]);
        if (fault != 0ull) {
            char buf[160];
            std::snprintf(buf, sizeof buf, "rt_wait: the ray-trace kernel reported fault %llu (1: dynamic LDS not at address 0); the frame is incomplete", fault);
            g_rt_err = buf;
            return RT_ERR_HIP;
        }
        return rt_comm_after_wait(c);
    }
    return RT_OK;
}

int rt_read_pixels(rt_ctx* c, uint8_t* dst, size_t cap) {
    if (!c || !dst) return fail(RT_ERR_INVALID_ARG, "rt_read_pixels: NULL argument");
    if (!c->d_out) return fail(RT_ERR_STATE, "rt_read_pixels: no colour buffer (rt_resize first)");
    int rc = rt_wait(c);
    if (rc != RT_OK) return rc;
    // rows owned by this rank, clipped to the frame (the last tile may be partial)
    const uint32_t lt = local_tiles(c);
    size_t rows = 0;
    for (uint32_t j = 0; j < lt; ++j) {
        const uint32_t y0 = (c->rank + j * c->world) * 8u;
        rows += (c->H - y0 >= 8u) ? 8u : (c->H - y0);
    }
    // the compact buffer keeps 8 rows per tile; only the final tile can be short, so the
    // first `rows` rows are contiguous valid data
    const size_t need = rows * c->W * 4u;
    if (cap < need) return fail(RT_ERR_CAPACITY, "rt_read_pixels: destination smaller than the local rows * W * 4");
    RT_HIP(hipMemcpyAsync(dst, c->d_out, need, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_read_pixels_async(rt_ctx* c, uint32_t frames_back, uint8_t* dst, size_t cap) {
    if (!c || !dst) return fail(RT_ERR_INVALID_ARG, "rt_read_pixels_async: NULL argument");
    if (frames_back >= (uint32_t)kStreams || frames_back >= c->frames_rendered)
        return fail(RT_ERR_INVALID_ARG, "rt_read_pixels_async: frames_back must name one of the last four rt_render calls");
    if (c->world != 1u || c->comm) return fail(RT_ERR_STATE, "rt_read_pixels_async: whole-frame contexts only");
    if (!c->d_outs[0]) return fail(RT_ERR_STATE, "rt_read_pixels_async: no colour buffer (rt_resize first)");
    const size_t need = (size_t)c->H * c->W * 4u;
    if (cap < need) return fail(RT_ERR_CAPACITY, "rt_read_pixels_async: destination smaller than H * W * 4");
    RT_HIP(hipSetDevice(c->device));
    const uint32_t k = (c->frames_rendered - 1u - frames_back) % (uint32_t)kStreams;
    if (c->buf_slot[k] >= 0) RT_HIP(hipStreamWaitEvent(c->copy_stream, c->ev_k1[c->buf_slot[k]], 0));   // behind that frame's kernels
    RT_HIP(hipMemcpyAsync(dst, c->d_outs[k], need, hipMemcpyDeviceToHost, c->copy_stream));
    RT_HIP(hipEventRecord(c->ev_copy[k], c->copy_stream));
    c->copy_pending[k] = true;
    return RT_OK;
}

int rt_read_pixels_wait(rt_ctx* c) {
    if (!c) return fail(RT_ERR_INVALID_ARG, "rt_read_pixels_wait: ctx is NULL");
    RT_HIP(hipSetDevice(c->device));
    RT_HIP(hipStreamSynchronize(c->copy_stream));
    for (int k = 0; k < kStreams; ++k) c->copy_pending[k] = false;
    return RT_OK;
}

int rt_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return fail(RT_ERR_INVALID_ARG, "rt_host_alloc: NULL / zero size");
    *out = nullptr;
    RT_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return RT_OK;
}

int rt_host_free(void* p) {
    if (p) RT_HIP(hipHostFree(p));
    return RT_OK;
}

int rt_get_stats(rt_ctx* c, rt_stats* out) {
    if (!c || !out) return fail(RT_ERR_INVALID_ARG, "rt_get_stats: NULL argument");
    c->stats.width = c->W;
    c->stats.height = c->H;
    c->stats.local_tiles = local_tiles(c);
    c->stats.spheres = c->n;
    c->stats.mode = c->mode;
    *out = c->stats;
    return RT_OK;
}

int rt_assemble_frame(rt_ctx* c, const void* gathered, void* frame, uint32_t world, void* hip_stream) {
    if (!c || !gathered || !frame) return fail(RT_ERR_INVALID_ARG, "rt_assemble_frame: NULL argument");
    if (world == 0) return fail(RT_ERR_INVALID_ARG, "rt_assemble_frame: world must be >= 1");
    if (!c->W || !c->H) return fail(RT_ERR_STATE, "rt_assemble_frame: rt_resize has not been called");
    RT_HIP(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    RT_HIP(rt_launch_assemble(static_cast<const uint8_t*>(gathered), static_cast<uint8_t*>(frame), c->W, c->H, world,
                              rt_padded_tiles(c->H, world), s));
    return RT_OK;
}

int rt_build_hierarchy(const float* records, uint32_t n, float* rec4, uint32_t* link, uint32_t cap_nodes,
                       uint32_t* n_nodes) {
    if ((n && !records) || !n_nodes) return fail(RT_ERR_INVALID_ARG, "rt_build_hierarchy: NULL argument");
    std::vector<float> r;
    std::vector<uint32_t> l;
    const uint32_t nodes = rt_bvh_build(records, n, r, l, rt_bvh_arity());
    *n_nodes = nodes;
    if (l.empty()) return RT_OK;                       // n == 0: nothing to write
    if (cap_nodes < nodes + 1u || !rec4 || !link) return fail(RT_ERR_CAPACITY, "rt_build_hierarchy: need n_nodes + 1 entries");
    std::memcpy(rec4, r.data(), r.size() * sizeof(float));
    std::memcpy(link, l.data(), l.size() * sizeof(uint32_t));
    return RT_OK;
}

int rt_build_flow(const float* nodes, uint32_t n_nodes, const uint32_t* roots, uint32_t n_roots, float* pairs, uint32_t cap_pairs,
                  uint32_t* n_pairs, uint32_t* root_meta) {
    if (!nodes || (n_roots && !roots) || !n_pairs) return fail(RT_ERR_INVALID_ARG, "rt_build_flow: NULL argument");
    RtFlow f;
    rt_flow_build(nodes, n_nodes, roots, n_roots, f);
    *n_pairs = f.n_pairs;
    if (!f.ok) return fail(RT_ERR_UNSUPPORTED, "rt_build_flow: node buffer beyond 65,536 entries or a count beyond 16 bits");
    if (f.n_pairs && (cap_pairs < f.n_pairs || !pairs)) return fail(RT_ERR_CAPACITY, "rt_build_flow: need room for n_pairs records");
    if (f.n_pairs) std::memcpy(pairs, f.pairs.data(), (size_t)f.n_pairs * 64u);
    if (root_meta)
        for (uint32_t r = 0; r < n_roots; ++r)
            root_meta[r] = rt_flow_meta(nodes, n_nodes, roots[r] < n_nodes - 1u ? roots[r] : n_nodes - 1u, f.pair_of);
    return RT_OK;
}

int rt_order_tiles(rt_ctx* c, const uint32_t* cost, uint32_t n, uint32_t wave_slots, uint32_t* order, size_t cap) {
    if (!c || (n && !cost) || !order) return fail(RT_ERR_INVALID_ARG, "rt_order_tiles: NULL argument");
    if (cap < (size_t)n + 2u) return fail(RT_ERR_CAPACITY, "rt_order_tiles: order needs n + 2 entries");
    if (n > (1u << 26)) return fail(RT_ERR_INVALID_ARG, "rt_order_tiles: more than 2^26 tiles");
    order[0] = order[1] = 0u;
    if (n == 0u) return RT_OK;
    RT_HIP(hipSetDevice(c->device));
    const size_t cost_bytes = ((size_t)n * 4u + 7u) & ~(size_t)7u, scan_bytes = (size_t)rt_order_scan_words() * 4u;
    void *d_cost = nullptr, *d_order = nullptr;
    hipError_t e = hipMalloc(&d_cost, cost_bytes + scan_bytes);
    if (e == hipSuccess) e = hipMalloc(&d_order, ((size_t)n + 2u) * 4u);
    if (e == hipSuccess) e = hipMemsetAsync(d_cost, 0, cost_bytes + scan_bytes, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_cost, cost, (size_t)n * 4u, hipMemcpyHostToDevice, c->stream);
    uint32_t* const scan = reinterpret_cast<uint32_t*>(static_cast<char*>(d_cost) + cost_bytes);
    // twice on the same buffers, as consecutive frames of a stream do: the first pass must leave the scan words (and the costs) zero
    for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {
        if (pass) e = hipMemcpyAsync(d_cost, cost, (size_t)n * 4u, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = rt_launch_order_hist(static_cast<uint32_t*>(d_cost), scan, static_cast<uint32_t*>(d_order), n, wave_slots, nullptr, nullptr, 0u, nullptr, c->stream);
        if (e == hipSuccess) e = rt_launch_order_scatter(static_cast<uint32_t*>(d_cost), scan, static_cast<uint32_t*>(d_order), n, c->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(order, d_order, ((size_t)n + 2u) * 4u, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_cost);
    (void)hipFree(d_order);
    if (e != hipSuccess) return fail_hip(e, "rt_order_tiles");
    return RT_OK;
}

int rt_filter_plan(const float* records, uint32_t n, const float params[24], int* filter_ok, int* signed_filter) {
    if ((n && !records) || !params || !filter_ok || !signed_filter) return fail(RT_ERR_INVALID_ARG, "rt_filter_plan: NULL argument");
    bool ok = false;
    uint32_t sgn = 0;
    rt_plan((double)(float)rt_scene_bound(records, n), rt_scene_min_radius(records, n), params, ok, sgn);
    *filter_ok = ok ? 1 : 0;
    *signed_filter = (int)sgn;
    return RT_OK;
}

int rt_device_pixels(rt_ctx* c, void** out_ptr, size_t* out_bytes) {
    if (!c || !out_ptr || !out_bytes) return fail(RT_ERR_INVALID_ARG, "rt_device_pixels: NULL argument");
    if (!c->d_out) return fail(RT_ERR_STATE, "rt_device_pixels: no colour buffer (rt_resize first)");
    *out_ptr = c->d_out;
    *out_bytes = c->out_bytes;
    return RT_OK;
}

}  // extern "C"
