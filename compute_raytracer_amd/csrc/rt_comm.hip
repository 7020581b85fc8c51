// rt_comm.hip -- multi-GPU behind the C ABI: the frame is split into 8-row tiles over the GPUs of one
// node (tile t -> rank t % world), each GPU renders its tiles, and the library itself moves them over
// RCCL (xGMI) and de-interleaves them into the row-major frame.  One call -- rt_render_gather, or
// rt_group_render for a single-process host -- is the whole of RendererRaytracing.render()'s GPU work
// (RR:434-470) across the group.  SURVEY.md 8(b)/(e).
//
// Two ways to form the group:
//   * one process per GPU (torchrun, MPI, ...): rank 0 calls rt_comm_unique_id, the host hands the
//     128 bytes to the other ranks by any side channel, every rank calls rt_comm_init on its context;
//   * one process, all GPUs (the reference's host is ONE Node thread): rt_group_create builds a context
//     per device and an ncclCommInitAll communicator.
// The exchange: gather to a root (grouped ncclSend / ncclRecv: each rank's padded tile buffer
// travels once, over its direct xGMI link to the root) or ncclAllGather in place (every rank ends up
// with the frame).  The root renders straight into its slot of the gather buffer, so nothing is copied
// before the exchange; `assemble_frame` (rt_assemble.hip) de-interleaves on the same stream.
// Frames in flight rotate over the context's four streams exactly as rt_render does; RCCL orders the
// exchanges of one communicator among themselves.
#include "rt_ctx.h"

#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <new>
#include <thread>

#include "rt_wait_poll.h"
#include "rt_exchange_plan.h"

static_assert(kExchangeStreams == (uint32_t)kStreams, "rt_exchange_plan.h and rt_ctx.h must agree on the number of streams");

struct rt_comm_state {
    ncclComm_t comm = nullptr;
    bool owns_comm = true;                 // false: the communicator belongs to an rt_group
    uint8_t* d_gather[kStreams] = {nullptr};   // [world][padded_tiles][8][W][4], on ranks that receive
    uint8_t* d_frame[kStreams] = {nullptr};    // [H][W][4], on ranks that receive
    size_t gather_bytes = 0, frame_bytes = 0;
    int latest = -1;                       // buffer set of the latest rt_render_gather (-1: none yet)
    bool latest_has_frame = false;         // this rank received that frame
    hipEvent_t ev_r1[RT355_MAX_IN_FLIGHT] = {nullptr};   // end of the render, before the exchange
    hipEvent_t ev_x1[RT355_MAX_IN_FLIGHT] = {nullptr};   // end of the exchange, before the de-interleave
    bool slot_gathered[RT355_MAX_IN_FLIGHT] = {false};
    uint32_t latest_w = 0, latest_h = 0;   // target size the latest frame was assembled for
    // failure handling: a call that failed half way, an asynchronous RCCL error or a missed deadline leaves the
    // communicator unusable; every collective entry point then returns RT_ERR_COMM until it is destroyed
    bool poisoned = false;
    uint32_t timeout_ms = 0;               // rt_set_comm_timeout
    ncclComm_t* group_slot = nullptr;      // rt_group: where the group keeps this communicator (cleared on abort)
};

static int fail_nccl(ncclResult_t r, const char* where) {
    char buf[256];
    std::snprintf(buf, sizeof buf, "%s: %s (%d)", where, ncclGetErrorString(r), (int)r);
    g_rt_err = buf;
    return RT_ERR_COMM;
}
#define RT_NCCL(call)                                                  \
    do {                                                               \
        ncclResult_t r_ = (call);                                      \
        if (r_ != ncclSuccess) return fail_nccl(r_, #call);            \
    } while (0)

static size_t message_bytes(const rt_ctx* c) { return rt_plan_message_bytes(c->W, c->H, c->world); }

static void free_buffers(rt_ctx* c) {
    rt_comm_state* s = c->comm;
    // rt_read_pixels / rt_device_pixels may be pointing into a gather buffer (the root renders straight into its slot)
    for (int k = 0; k < kStreams; ++k)
        if (s->d_gather[k] && c->d_out >= s->d_gather[k] && c->d_out < s->d_gather[k] + s->gather_bytes) c->d_out = c->d_outs[0];
    for (int k = 0; k < kStreams; ++k) {
        (void)hipFree(s->d_gather[k]); s->d_gather[k] = nullptr;
        (void)hipFree(s->d_frame[k]); s->d_frame[k] = nullptr;
    }
    s->gather_bytes = s->frame_bytes = 0;
    s->latest = -1;
    s->latest_has_frame = false;
}

void rt_comm_release(rt_ctx* c) {
    rt_comm_state* s = c->comm;
    if (!s) return;
    free_buffers(c);
    for (int i = 0; i < RT355_MAX_IN_FLIGHT; ++i)
        if (s->ev_r1[i]) (void)hipEventDestroy(s->ev_r1[i]);
    for (int i = 0; i < RT355_MAX_IN_FLIGHT; ++i)
        if (s->ev_x1[i]) (void)hipEventDestroy(s->ev_x1[i]);
    if (s->comm && s->owns_comm) (void)ncclCommDestroy(s->comm);
    delete s;
    c->comm = nullptr;
}

// render time and exchange time of the frames the last rt_wait completed
int rt_comm_after_wait(rt_ctx* c) {
    rt_comm_state* s = c->comm;
    c->stats.gather_ms = 0.0f;
    c->stats.batch_gather_ms = 0.0f;
    if (!s) return RT_OK;
    float kernel_sum = 0.0f;
    bool any = false;
    for (uint32_t i = 0; i < c->stats.batch_frames && i < RT355_MAX_IN_FLIGHT; ++i) {
        if (!s->slot_gathered[i]) continue;
        float render = 0.0f, gather = 0.0f;
        if (hipEventElapsedTime(&render, c->ev_k0[i], s->ev_r1[i]) == hipSuccess &&
            hipEventElapsedTime(&gather, s->ev_r1[i], c->ev_k1[i]) == hipSuccess) {
            c->stats.kernel_ms = render;
            c->stats.gather_ms = gather;
            c->stats.batch_gather_ms += gather;
            kernel_sum += render;
            any = true;
        }
        s->slot_gathered[i] = false;
    }
    if (any) c->stats.batch_kernel_ms = kernel_sum;
    (void)hipGetLastError();
    return RT_OK;
}

// Aborts the communicator (the exchange kernels leave the device), waits for what is left on the streams and
// forgets the frames in flight.  The context stays valid for rt_comm_destroy / rt_destroy / single-GPU rendering.
static void abort_comm(rt_ctx* c) {
    rt_comm_state* s = c->comm;
    s->poisoned = true;
    if (s->comm) {
        (void)ncclCommAbort(s->comm);
        if (s->group_slot) *s->group_slot = nullptr;
        s->comm = nullptr;
    }
    for (int i = 0; i < RT355_MAX_IN_FLIGHT; ++i) s->slot_gathered[i] = false;
    s->latest = -1;
    s->latest_has_frame = false;
    rt_abandon_in_flight(c);
}

int rt_comm_wait_frames(rt_ctx* c) {
    rt_comm_state* s = c->comm;
    hipError_t hip_err = hipSuccess;
    ncclResult_t async = ncclSuccess;
    const RtPollVerdict v = rt_poll_until(
        c->in_flight, s->timeout_ms,
        [&](uint32_t i) {
            const hipError_t e = hipEventQuery(c->ev_k1[i]);
            if (e == hipSuccess) return true;
            if (e != hipErrorNotReady) { hip_err = e; return true; }      // reported below
            (void)hipGetLastError();
            return false;
        },
        [&]() {
            if (!s->comm || s->poisoned) return s->poisoned;
            ncclResult_t r = ncclSuccess;
            if (ncclCommGetAsyncError(s->comm, &r) != ncclSuccess) { async = ncclSystemError; return true; }
            if (r != ncclSuccess && r != ncclInProgress) { async = r; return true; }
            return false;
        },
        []() { return (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); },
        []() { std::this_thread::yield(); });
    if (hip_err != hipSuccess) return fail_hip(hip_err, "rt_wait: hipEventQuery");
    if (v == RtPollVerdict::Done) return RT_OK;
    const bool earlier = s->poisoned && async == ncclSuccess;      // a call of this batch failed half way: no RCCL error to name
    abort_comm(c);
    if (earlier) return fail(RT_ERR_COMM, "rt_wait: an exchange of this batch failed while it was being enqueued (see that call's error); communicator aborted");
    if (v == RtPollVerdict::Timeout) return fail(RT_ERR_COMM, "rt_wait: the frames did not complete within the communicator's deadline (rt_set_comm_timeout); communicator aborted");
    return fail_nccl(async, "rt_wait: asynchronous RCCL error; communicator aborted");
}

static int attach(rt_ctx* c, ncclComm_t comm, bool owns, uint32_t rank, uint32_t world) {
    rt_comm_state* s = new (std::nothrow) rt_comm_state();
    if (!s) return fail(RT_ERR_HIP, "rt_comm_init: out of host memory");
    s->comm = comm;
    s->owns_comm = owns;
    for (int i = 0; i < RT355_MAX_IN_FLIGHT; ++i) {
        hipError_t e = hipEventCreate(&s->ev_r1[i]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_x1[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            c->comm = s;
            rt_comm_release(c);
            return fail_hip(e, "rt_comm_init: hipEventCreate");
        }
    }
    int rc = rt_set_partition(c, rank, world);     // before the communicator is attached: afterwards the partition is fixed
    c->comm = s;
    if (rc != RT_OK) rt_comm_release(c);
    return rc;
}

// buffers of the ranks that receive: sized for the current target and partition
static int ensure_buffers(rt_ctx* c, bool receives) {
    rt_comm_state* s = c->comm;
    if (!receives) return RT_OK;
    const size_t gb = message_bytes(c) * c->world, fb = (size_t)c->H * c->W * 4u;
    if (gb <= s->gather_bytes && fb <= s->frame_bytes && s->d_gather[0] && s->d_frame[0]) return RT_OK;
    { int rc = rt_drain(c); if (rc != RT_OK) return rc; }
    free_buffers(c);
    for (int k = 0; k < kStreams; ++k) {
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_gather[k]), gb));
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_frame[k]), fb));
        RT_HIP(hipMemsetAsync(s->d_gather[k], 0, gb, c->stream));
    }
    RT_HIP(hipStreamSynchronize(c->stream));
    s->gather_bytes = gb;
    s->frame_bytes = fb;
    return RT_OK;
}

// The render of one frame and, on the same stream, its exchange and de-interleave.  The NCCL calls
// are issued WITHOUT group markers: the callers bracket them (one context: its own group; an
// rt_group: one group over all its devices, as a single thread driving several GPUs must).
static int render_part(rt_ctx* c, int root, uint32_t k, uint8_t** part) {
    rt_comm_state* s = c->comm;
    const RtXPlan plan = rt_exchange_plan(c->W, c->H, c->rank, c->world, root);
    { int rc = ensure_buffers(c, plan.receives); if (rc != RT_OK) return rc; }
    *part = plan.part_in_gather ? s->d_gather[k] + plan.part_offset : c->d_outs[k];
    if (!*part) return fail(RT_ERR_STATE, "rt_render_gather: rt_resize has not been called");
    if (!plan.part_in_gather) { int rc = rt_order_colour_buffer(c, k, c->streams[k]); if (rc != RT_OK) return rc; }
    int rc = rt_enqueue(c, *part, c->streams[k]);
    if (rc != RT_OK) return rc;
    const uint32_t slot = c->in_flight - 1u;
    if (!plan.part_in_gather) c->buf_slot[k] = (int)slot;
    RT_HIP(hipEventRecord(s->ev_r1[slot], c->streams[k]));
    s->slot_gathered[slot] = true;
    return RT_OK;
}

// Frames in flight sit on different streams, and every rank must run the exchanges of one communicator in
// the same order: the order of the calls.  That order is made explicit on the device -- the exchange of a
// frame waits for the exchange of the frame enqueued before it (not for its de-interleave, and the renders
// stay free to overlap) -- instead of being left to how RCCL treats one communicator on several streams.
static int order_exchange(rt_ctx* c, uint32_t k) {
    const int before = rt_exchange_waits_on(c->in_flight - 1u);
    if (before >= 0) RT_HIP(hipStreamWaitEvent(c->streams[k], c->comm->ev_x1[before], 0));
    return RT_OK;
}

// the frame's RCCL operations, as rt_exchange_plan.h lists them for this rank
static int exchange_part(rt_ctx* c, int root, uint32_t k, uint8_t* part) {
    rt_comm_state* s = c->comm;
    hipStream_t st = c->streams[k];
    const RtXPlan plan = rt_exchange_plan(c->W, c->H, c->rank, c->world, root);
    for (const RtXOp& op : plan.ops) {
        if (op.kind == RtXKind::AllGather)
            RT_NCCL(ncclAllGather(s->d_gather[k] + op.gather_offset, s->d_gather[k], op.bytes, ncclUint8, s->comm, st));   // in place: part = recv + rank * msg
        else if (op.kind == RtXKind::Recv)
            RT_NCCL(ncclRecv(s->d_gather[k] + op.gather_offset, op.bytes, ncclUint8, (int)op.peer, s->comm, st));
        else
            RT_NCCL(ncclSend(part, op.bytes, ncclUint8, (int)op.peer, s->comm, st));
    }
    return RT_OK;
}

static int finish_part(rt_ctx* c, int root, uint32_t k, uint8_t* part) {
    rt_comm_state* s = c->comm;
    const bool receives = root < 0 || (uint32_t)root == c->rank;
    hipStream_t st = c->streams[k];
    RT_HIP(hipEventRecord(s->ev_x1[c->in_flight - 1u], st));
    // The de-interleave of the whole frame on the GPU that also renders its share: priced as the root of eight on one GPU
    // (tools/root_probe.py, profiles/r05/root_probe.log) -- 66 MB of HBM traffic per C3 frame, behind the exchange on the frame's
    // own stream while three other frames render: +0.004 ms on the frame period.  Taking it off the device (one 2-D copy per rank
    // in rt_read_frame) was built and measured too: the copies are PCIe-bound either way, and eight strided ones are slower than
    // this kernel + one linear copy (0.667 against 0.589 ms per frame read).  So it stays here.
    if (receives)
        RT_HIP(rt_launch_assemble(s->d_gather[k], s->d_frame[k], c->W, c->H, c->world, rt_padded_tiles(c->H, c->world), st));
    // the frame is complete when the exchange and the de-interleave are: move the slot's end event
    RT_HIP(hipEventRecord(c->ev_k1[c->in_flight - 1u], st));
    s->latest = (int)k;
    s->latest_has_frame = receives;
    s->latest_w = c->W; s->latest_h = c->H;
    c->d_out = part;       // rt_read_pixels / rt_device_pixels keep returning THIS rank's tiles
    ++c->frames_rendered;
    return RT_OK;
}

static int check_gather_args(rt_ctx* c, int root, const char* who) {
    if (!c) return fail(RT_ERR_INVALID_ARG, who);
    if (!c->comm) return fail(RT_ERR_STATE, "rt_render_gather: no communicator (rt_comm_init / rt_group_create first)");
    if (c->comm->poisoned) return fail(RT_ERR_COMM, "rt_render_gather: the communicator failed earlier (an RCCL error, a missed deadline or a call that failed half way); destroy it");
    if (root < -1 || root >= (int)c->world) return fail(RT_ERR_INVALID_ARG, "rt_render_gather: root must be -1 (all ranks) or a rank");
    if (!c->W || !c->H) return fail(RT_ERR_STATE, "rt_render_gather: rt_resize has not been called");
    return RT_OK;
}

struct rt_group {
    std::vector<rt_ctx*> ctx;
    std::vector<ncclComm_t> comms;
};

extern "C" {

int rt_comm_unique_id(uint8_t id[RT355_COMM_ID_BYTES]) {
    if (!id) return fail(RT_ERR_INVALID_ARG, "rt_comm_unique_id: id is NULL");
    static_assert(RT355_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "RT355_COMM_ID_BYTES must be RCCL's NCCL_UNIQUE_ID_BYTES");
    ncclUniqueId u;
    RT_NCCL(ncclGetUniqueId(&u));
    std::memcpy(id, u.internal, RT355_COMM_ID_BYTES);
    return RT_OK;
}

int rt_comm_init(rt_ctx* c, const uint8_t id[RT355_COMM_ID_BYTES], uint32_t rank, uint32_t world) {
    if (!c || !id) return fail(RT_ERR_INVALID_ARG, "rt_comm_init: NULL argument");
    if (world == 0 || rank >= world) return fail(RT_ERR_INVALID_ARG, "rt_comm_init: need rank < world");
    if (c->comm) return fail(RT_ERR_STATE, "rt_comm_init: the context already has a communicator");
    RT_HIP(hipSetDevice(c->device));
    { int rc = rt_drain(c); if (rc != RT_OK) return rc; }
    ncclUniqueId u;
    std::memcpy(u.internal, id, RT355_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    RT_NCCL(ncclCommInitRank(&comm, (int)world, u, (int)rank));
    int rc = attach(c, comm, true, rank, world);
    if (rc != RT_OK && !c->comm) (void)ncclCommDestroy(comm);
    return rc;
}

int rt_comm_destroy(rt_ctx* c) {
    if (!c) return fail(RT_ERR_INVALID_ARG, "rt_comm_destroy: ctx is NULL");
    if (!c->comm) return RT_OK;
    if (!c->comm->owns_comm) return fail(RT_ERR_STATE, "rt_comm_destroy: the communicator belongs to an rt_group");
    RT_HIP(hipSetDevice(c->device));
    { int rc = rt_drain(c); if (rc != RT_OK) return rc; }
    c->d_out = c->d_outs[0];
    c->out_bytes = c->d_outs[0] ? c->out_bytes : 0;
    rt_comm_release(c);
    return rt_set_partition(c, 0, 1);
}

// Everything that can be refused BEFORE anything is enqueued: arguments, state, and the buffers of a receiving rank.
static int prepare_gather(rt_ctx* c, int root, const char* who) {
    { int rc = check_gather_args(c, root, who); if (rc != RT_OK) return rc; }
    RT_HIP(hipSetDevice(c->device));
    return ensure_buffers(c, root < 0 || (uint32_t)root == c->rank);
}

// From the render's enqueue to the end event every step must happen on every rank, or the ranks' collectives no
// longer pair up: a failure in between poisons the communicator (the peers find out through their deadline).
static int poison(rt_ctx* c, int rc) {
    if (rc != RT_OK && c->comm) c->comm->poisoned = true;
    return rc;
}

int rt_render_gather(rt_ctx* c, int root) {
    { int rc = prepare_gather(c, root, "rt_render_gather: ctx is NULL"); if (rc != RT_OK) return rc; }
    if (!c->comm->owns_comm) return fail(RT_ERR_STATE, "rt_render_gather: this context belongs to an rt_group; call rt_group_render");
    const uint32_t k = rt_exchange_set(c->frames_rendered);
    uint8_t* part = nullptr;
    {
        const uint32_t before = c->in_flight;          // a render that failed before it enqueued anything leaves the group intact
        int rc = render_part(c, root, k, &part);
        if (rc != RT_OK) return c->in_flight != before ? poison(c, rc) : rc;
    }
    { int rc = order_exchange(c, k); if (rc != RT_OK) return poison(c, rc); }
    { ncclResult_t r = ncclGroupStart(); if (r != ncclSuccess) return poison(c, fail_nccl(r, "ncclGroupStart")); }
    int rc = exchange_part(c, root, k, part);
    ncclResult_t ge = ncclGroupEnd();
    if (rc != RT_OK) return poison(c, rc);
    if (ge != ncclSuccess) return poison(c, fail_nccl(ge, "ncclGroupEnd"));
    return poison(c, finish_part(c, root, k, part));
}

int rt_set_comm_timeout(rt_ctx* c, uint32_t ms) {
    if (!c) return fail(RT_ERR_INVALID_ARG, "rt_set_comm_timeout: ctx is NULL");
    if (!c->comm) return fail(RT_ERR_STATE, "rt_set_comm_timeout: no communicator (rt_comm_init / rt_group_create first)");
    c->comm->timeout_ms = ms;
    return RT_OK;
}

int rt_frame_pixels(rt_ctx* c, void** out_ptr, size_t* out_bytes) {
    if (!c || !out_ptr || !out_bytes) return fail(RT_ERR_INVALID_ARG, "rt_frame_pixels: NULL argument");
    if (!c->comm || c->comm->latest < 0) return fail(RT_ERR_STATE, "rt_frame_pixels: no rt_render_gather yet");
    if (!c->comm->latest_has_frame) return fail(RT_ERR_STATE, "rt_frame_pixels: this rank did not receive the frame (it is not the root)");
    if (c->comm->latest_w != c->W || c->comm->latest_h != c->H || (size_t)c->H * c->W * 4u > c->comm->frame_bytes)
        return fail(RT_ERR_STATE, "rt_frame_pixels: the target was resized after the latest rt_render_gather");
    *out_ptr = c->comm->d_frame[c->comm->latest];
    *out_bytes = (size_t)c->H * c->W * 4u;
    return RT_OK;
}

int rt_read_frame(rt_ctx* c, uint8_t* dst, size_t cap) {
    if (!c || !dst) return fail(RT_ERR_INVALID_ARG, "rt_read_frame: NULL argument");
    void* p = nullptr;
    size_t bytes = 0;
    { int rc = rt_frame_pixels(c, &p, &bytes); if (rc != RT_OK) return rc; }
    if (cap < bytes) return fail(RT_ERR_CAPACITY, "rt_read_frame: destination smaller than W*H*4");
    { int rc = rt_wait(c); if (rc != RT_OK) return rc; }
    RT_HIP(hipMemcpyAsync(dst, p, bytes, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(hipStreamSynchronize(c->stream));
    return RT_OK;
}

// ---- one process, all GPUs -------------------------------------------------------------------------------

int rt_group_create(int n_devices, rt_group** out) {
    if (!out) return fail(RT_ERR_INVALID_ARG, "rt_group_create: out is NULL");
    *out = nullptr;
    rt_default_hw_queues();
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(RT_ERR_NO_DEVICE, "rt_group_create: no HIP device visible (this library has no CPU path)");
    }
    if (n_devices == 0) n_devices = count;
    if (n_devices < 0 || n_devices > count) return fail(RT_ERR_NO_DEVICE, "rt_group_create: more devices requested than visible");
    rt_group* g = new (std::nothrow) rt_group();
    if (!g) return fail(RT_ERR_HIP, "rt_group_create: out of host memory");
    for (int d = 0; d < n_devices; ++d) {
        rt_ctx* c = nullptr;
        int rc = rt_create(d, &c);
        if (rc != RT_OK) { rt_group_destroy(g); return rc; }
        g->ctx.push_back(c);
    }
    g->comms.assign((size_t)n_devices, nullptr);
    std::vector<int> devs((size_t)n_devices);
    for (int d = 0; d < n_devices; ++d) devs[(size_t)d] = d;
    ncclResult_t r = ncclCommInitAll(g->comms.data(), n_devices, devs.data());
    if (r != ncclSuccess) { rt_group_destroy(g); return fail_nccl(r, "ncclCommInitAll"); }   // whatever it did create is destroyed
    for (int d = 0; d < n_devices; ++d) {
        (void)hipSetDevice(d);
        int rc = attach(g->ctx[(size_t)d], g->comms[(size_t)d], false, (uint32_t)d, (uint32_t)n_devices);
        if (rc != RT_OK) { rt_group_destroy(g); return rc; }
        g->ctx[(size_t)d]->comm->group_slot = &g->comms[(size_t)d];
    }
    *out = g;
    return RT_OK;
}

int rt_group_destroy(rt_group* g) {
    if (!g) return RT_OK;
    for (rt_ctx* c : g->ctx) (void)rt_destroy(c);          // drains, releases the gather buffers
    for (ncclComm_t comm : g->comms)
        if (comm) (void)ncclCommDestroy(comm);
    delete g;
    return RT_OK;
}

int rt_group_size(const rt_group* g) { return g ? (int)g->ctx.size() : 0; }

rt_ctx* rt_group_ctx(rt_group* g, int i) {
    if (!g || i < 0 || i >= (int)g->ctx.size()) { (void)fail(RT_ERR_INVALID_ARG, "rt_group_ctx: index out of range"); return nullptr; }
    return g->ctx[(size_t)i];
}

int rt_group_render(rt_group* g, int root) {
    if (!g || g->ctx.empty()) return fail(RT_ERR_INVALID_ARG, "rt_group_render: group is NULL");
    const size_t n = g->ctx.size();
    std::vector<uint8_t*> part(n, nullptr);
    std::vector<uint32_t> k(n, 0u);
    // every member is checked (arguments, state, buffers) before anything is enqueued on any
    for (size_t d = 0; d < n; ++d) { int rc = prepare_gather(g->ctx[d], root, "rt_group_render: ctx is NULL"); if (rc != RT_OK) return rc; }
    // from here on a failure leaves the members at different points of the collective sequence: the whole group is poisoned
    auto poison_all = [&](int rc) { if (rc != RT_OK) for (rt_ctx* m : g->ctx) if (m->comm) m->comm->poisoned = true; return rc; };
    for (size_t d = 0; d < n; ++d) {
        rt_ctx* c = g->ctx[d];
        { hipError_t e = hipSetDevice(c->device); if (e != hipSuccess) return poison_all(fail_hip(e, "hipSetDevice")); }
        k[d] = rt_exchange_set(c->frames_rendered);
        {
            const uint32_t before = c->in_flight;
            int rc = render_part(c, root, k[d], &part[d]);
            if (rc != RT_OK) return (d == 0 && c->in_flight == before) ? rc : poison_all(rc);
        }
        { int rc = order_exchange(c, k[d]); if (rc != RT_OK) return poison_all(rc); }
    }
    { ncclResult_t r = ncclGroupStart(); if (r != ncclSuccess) return poison_all(fail_nccl(r, "ncclGroupStart")); }
    int rc = RT_OK;
    for (size_t d = 0; d < n && rc == RT_OK; ++d) {
        (void)hipSetDevice(g->ctx[d]->device);
        rc = exchange_part(g->ctx[d], root, k[d], part[d]);
    }
    ncclResult_t ge = ncclGroupEnd();
    if (rc != RT_OK) return poison_all(rc);
    if (ge != ncclSuccess) return poison_all(fail_nccl(ge, "ncclGroupEnd"));
    for (size_t d = 0; d < n; ++d) {
        { hipError_t e = hipSetDevice(g->ctx[d]->device); if (e != hipSuccess) return poison_all(fail_hip(e, "hipSetDevice")); }
        { int rc2 = finish_part(g->ctx[d], root, k[d], part[d]); if (rc2 != RT_OK) return poison_all(rc2); }
    }
    return RT_OK;
}

int rt_group_wait(rt_group* g) {
    if (!g) return fail(RT_ERR_INVALID_ARG, "rt_group_wait: group is NULL");
    int first = RT_OK;
    std::string msg;
    for (rt_ctx* c : g->ctx) {       // every member is waited for (or aborted) even when one fails
        int rc = rt_wait(c);
        if (rc != RT_OK && first == RT_OK) { first = rc; msg = g_rt_err; }
    }
    if (first != RT_OK) g_rt_err = msg;
    return first;
}

}  // extern "C"
