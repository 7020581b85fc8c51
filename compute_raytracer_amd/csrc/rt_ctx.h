// rt_ctx.h -- the context behind the C ABI, shared by rt_api.hip (single-GPU entry points) and
// rt_comm.hip (the RCCL group entry points).  Private to the library.
#pragma once
#include "../../include/rt355.h"
#include "rt_types.h"
#include "rt_tri_types.h"
#include "rt_flow_build.h"

#include <cstdio>
#include <string>
#include <vector>

extern thread_local std::string g_rt_err;      // rt_last_error()

inline int fail(int code, const char* what) {
    g_rt_err = what;
    return code;
}
inline int fail_hip(hipError_t e, const char* where) {
    char buf[256];
    std::snprintf(buf, sizeof buf, "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
    g_rt_err = buf;
    return RT_ERR_HIP;
}
#define RT_HIP(call)                                                   \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) return fail_hip(e_, #call);              \
    } while (0)

// Frames the library itself keeps concurrent (rt_render rotates over this many streams and colour
// buffers).  Why: at the end of a frame every lane of the persistent hierarchy kernel still carries
// a path of up to 2*bounces dependent rays; that tail is latency, not work (0.45 ms of a 2.9 ms C3
// frame, 0.45 of 0.8 ms when 8 ranks share the frame).  Frames in flight each take a share of the
// chip (RtFrameArgs::grid_share), so one frame's tail runs beside the others' bulk.
constexpr int kStreams = 4;
constexpr int kVersions = 4;         // versions of the per-frame instance buffers (>= kStreams)
constexpr uint32_t kHeadNodes = 31u; // TLAS nodes (2 M - 1 for M <= 16 instances) that travel with a frame
constexpr uint32_t kInstMax = 16u;   // instances whose records travel with a frame
constexpr uint32_t kOrderMinTiles = 4096u;   // frames of fewer tiles than the chip has wave slots start all of them at once: no order to choose
constexpr size_t kCounterBytes = (size_t)RT_RAY_COUNTERS * RT_RAY_COUNTER_STRIDE;   // partial ray counters of one frame
constexpr size_t kCtrlBytes = kCounterBytes + 32u;                                   // + the 32-byte control block

struct rt_comm_state;   // rt_comm.hip

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;        // uploads, read-back, and frames 0, 3, 6 ... of rt_render (= streams[0])
    hipStream_t streams[kStreams] = {nullptr};   // rt_render rotates: consecutive frames may overlap on the device
    hipEvent_t ev_prep0[RT355_MAX_IN_FLIGHT] = {nullptr}, ev_k0[RT355_MAX_IN_FLIGHT] = {nullptr},
               ev_k1[RT355_MAX_IN_FLIGHT] = {nullptr};
    hipEvent_t ev_done[RT355_MAX_IN_FLIGHT] = {nullptr};          // behind the frame's epilogue kernel: what rt_wait waits for
    hipEvent_t ev_scene = nullptr;       // the scene arrays / hierarchy a frame reads are complete ...
    hipStream_t scene_stream = nullptr;  // ... recorded on this stream
    uint32_t in_flight = 0;              // frames enqueued since the last rt_wait
    hipStream_t slot_stream[RT355_MAX_IN_FLIGHT] = {nullptr};   // the stream each of them was enqueued on
    uint32_t frames_rendered = 0;        // rt_render calls: selects the colour buffer
    uint32_t stream_rot = 0;             // the stream rt_render enqueues on: advances only while frames are in flight
    bool pipelined_hint = false;         // the last rt_wait completed frames on MORE THAN ONE of the library's own streams
                                         // (rt_render / rt_render_gather back to back): the caller keeps frames in flight
    uint32_t W = 0, H = 0;
    uint32_t rank = 0, world = 1;
    float params[24] = {0};
    bool have_params = false;
    bool prep_spheres_valid = false;     // prep_spheres ran since the last rt_write_spheres
    bool prep_params_valid = false;      // ... and since the last rt_write_params (camera / light records)
    float* d_records = nullptr;
    uint32_t n = 0, cap_n = 0;
    bool have_spheres = false;
    float4* d_scene = nullptr;           // 8 float4 arrays of n16: geo lgt cam col geo_f lgt_f cam_f + {geo_w,lgt_w,cam_w,-}
    uint32_t n16 = 0;                    // n rounded up to a multiple of 16
    float scene_bound = 0.0f;            // max over spheres of |center| + radius (host side)
    double scene_min_radius = 0.0;       // smallest non-zero |radius| (rt_plan: the rescaled sign-aware test)
    // bounding-sphere hierarchy (rt_bvh.hip): host copy of the records it is built from, the
    // build result and its device copy
    std::vector<float> h_records;
    std::vector<float> h_bvh_rec;
    std::vector<uint32_t> h_bvh_link;
    float4* d_bvh_rec = nullptr;
    uint32_t* d_bvh_link = nullptr;
    uint32_t bvh_cap = 0, bvh_nodes = 0;
    bool bvh_valid = false;              // the device records bound the current spheres
    uint32_t bvh_topo_n = 0;             // sphere count the device topology (links, leaf ids) was built for; 0: none
    struct rt_rebuild* rebuild = nullptr;   // worker thread that rebuilds the topology for moved spheres (rt_api.hip)
    uint8_t* d_face[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t fw[6] = {0, 0, 0, 0, 0, 0}, fh[6] = {0, 0, 0, 0, 0, 0};
    uint32_t face_texel0[6] = {0, 0, 0, 0, 0, 0};   // first texel (rgb) of each face: a one-texel sky of one colour is "flat"
    uint8_t* d_out = nullptr;              // colour buffer of the LATEST rt_render (one of d_outs)
    uint8_t* d_outs[kStreams] = {nullptr};
    // streaming read-back (rt_read_pixels_async): a copy stream, per colour buffer the event of the last copy out of it and the
    // event slot of the frame of the current batch that renders into it (-1: none in flight)
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copy[kStreams] = {nullptr};
    bool copy_pending[kStreams] = {false};
    int buf_slot[kStreams] = {-1, -1, -1, -1};
    float4* d_fin[kStreams] = {nullptr};         // end-of-path records of frames with a textured sky (rt_bvh.hip: sky_resolve), lazily
    size_t fin_slots = 0;                        // pixel slots each of them holds
    size_t out_bytes = 0;
    // per frame in flight: RT_RAY_COUNTERS partial ray counters (kCtrlBytes - 32 bytes), then a 32-byte
    // control block: unused u64, queue count, queue head, pixel / tile-pair cursor
    unsigned long long* d_rays = nullptr;
    float4* d_queue = nullptr;             // path queue of the two-kernel pipeline
    size_t queue_cap = 0;                  // entries
    unsigned long long* h_split = nullptr; // behind h_rays, per stream: the number of tiles the stream's current work list renders in parts (order_hist writes it)
    unsigned long long* h_rays = nullptr;  // pinned, device-visible: per frame in flight {rays, fault word}, written by the frame's epilogue kernel
    // the reference's triangle scene (RR:169-229), device copies in the reference's byte layouts
    struct DevBuf { void* p = nullptr; size_t cap = 0; size_t used = 0; };
    DevBuf d_tri, d_tri_lookup, d_tex;
    DevBuf d_corners;                            // 48 B per lookup slot: the corners hitTriangle reads, in lookup order (tri_corners)
    bool corners_valid = false;                  // ... built since the last rt_write_triangles / rt_write_tri_lookup
    // Tile order of the triangle kernel (rt_triangles.hip: order_tiles), one set per stream of rt_render: the time every
    // tile of the stream's last frame took, and the longest-first permutation made of it for the stream's next frame.
    DevBuf d_tile_cost[kStreams], d_tile_order[kStreams];
    uint32_t wave_slots = 4096;                      // waves of the triangle kernel the device holds at once
    uint32_t order_tiles[kStreams] = {0, 0, 0, 0};   // tile count d_tile_order[k] is a permutation of (0: none yet)
    // The buffers the reference rewrites before every frame (RR:169-192: BLAS records, BLAS lookup, the TLAS nodes at
    // the head of the node buffer) exist in kVersions versions: a frame in flight keeps reading the version it was
    // enqueued with while the host already writes the next state (rt_api.hip: apply_instances).
    DevBuf d_nodes[kVersions], d_blas[kVersions], d_blas_lookup[kVersions];
    struct {
        std::vector<float> head, blas, lookup;   // current contents: first head_nodes nodes, BLAS records, BLAS lookup
        uint32_t head_nodes = 0;                 // extent of the node buffer's head that per-frame writes have touched
        bool blas_on = false, lookup_on = false; // the last write of that buffer was a per-frame (small) one
        uint64_t gen = 0;                        // bumped by every per-frame write
    } inst;
    uint64_t inst_gen_carried = 0;               // inst.gen the last frame of a small form took along in its arguments (rt_stats.instance_uploads)
    uint64_t ver_gen[kVersions] = {0, 0, 0, 0};  // inst.gen each device version holds
    hipEvent_t ev_ver[kVersions] = {nullptr};    // behind the apply_instances kernel that last brought the version up to date ...
    hipStream_t ver_stream[kVersions] = {nullptr};   // ... on this stream
    size_t nodes_used = 0;                       // bytes of the node buffer written so far (any version)
    uint32_t node_count_max = 0;                 // largest u32(primitiveCount) of any node written so far (packed BLAS stack: <= 65535)
    // The triangle kernel reads the BLAS trees from the library's relinked copy (rt_flow_build.h: pair records) where the scene
    // fits it: built on the host from a mirror of the node buffer, when a frame first needs it or a write has touched what it
    // was built from.
    std::vector<float> h_nodes;                  // mirror of the node buffer (every rt_write_nodes lands here too)
    RtFlow flow;
    bool flow_dirty = true;
    uint32_t flow_frames_since = 0, flow_streak = 0, flow_cooldown = 0;   // frames since the last rebuild; consecutive frames that rebuilt; frames left on the node walk
    DevBuf d_flow;                               // the pair records
    DevBuf d_tri_dbg;                            // development builds: the triangle kernel's per-workgroup timeline
    uint32_t n_cus = 256;
    uint32_t tex_w = 0, tex_h = 0;
    int scene_kind = 0;                    // 0 spheres, 1 triangles: the primitive type written last
    int mode = RT_MODE_FAST;
    int variant = 0;
    int kernel = RT_KERNEL_RAYTRACER;
    rt_stats stats = {};
    rt_comm_state* comm = nullptr;         // RCCL communicator + gather buffers (rt_comm_init / rt_group_create)
};

// rt_api.hip (defined inside its extern "C" block; not exported through include/rt355.h)
#define RT_INTERNAL __attribute__((visibility("hidden")))
extern "C" {
RT_INTERNAL int rt_enqueue(rt_ctx* c, uint8_t* dst, hipStream_t s);   // prep + ray-trace launches of one frame into `dst` on `s`
RT_INTERNAL int rt_order_colour_buffer(rt_ctx* c, uint32_t k, hipStream_t st);   // before a frame on `st` writes colour buffer k
RT_INTERNAL void rt_default_hw_queues(void);                          // GPU_MAX_HW_QUEUES=8 unless the host set it: before the first HIP call
RT_INTERNAL int rt_drain(rt_ctx* c);                                  // waits for the frames in flight
RT_INTERNAL uint32_t rt_local_tiles(const rt_ctx* c);
RT_INTERNAL void rt_abandon_in_flight(rt_ctx* c);                     // after a communicator abort: forget the frames in flight
}
// rt_comm.hip
RT_INTERNAL void rt_comm_release(rt_ctx* c);                          // called by rt_destroy
RT_INTERNAL int rt_comm_after_wait(rt_ctx* c);                        // called by rt_wait once the frames in flight are complete
RT_INTERNAL int rt_comm_wait_frames(rt_ctx* c);                       // called by rt_wait: polls the frames' events and the communicator


