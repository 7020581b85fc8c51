// rt_exchange_plan.h -- what the multi-GPU exchange of one frame consists of, decided on the host without a device:
// which buffer a rank renders into, which RCCL operations it issues in which order (peer, offset into the gather buffer,
// byte count), which earlier frame's exchange the frame's exchange waits for, and which stream / buffer set the frame
// uses.  rt_comm.hip executes exactly this plan; tests/c/exchange_plan_test.cpp checks it for every rank of a group
// against every other rank's (sends pair with receives, receives tile the gather buffer, every rank lists the
// communicator's operations in the same order) -- the part of SURVEY.md 8(e) that can be verified without a second GPU.
// Host-only, no HIP, no RCCL.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

constexpr uint32_t kExchangeStreams = 4;       // = rt_ctx.h kStreams: frames rotate over this many streams and buffer sets

inline uint32_t rt_plan_tiles_total(uint32_t H) { return (H + 7u) / 8u; }
// tiles t with t % world == rank (include/rt355.h: rt_tiles_of_rank)
inline uint32_t rt_plan_tiles_of_rank(uint32_t H, uint32_t rank, uint32_t world) {
    const uint32_t n = rt_plan_tiles_total(H);
    return rank < n ? (n - rank + world - 1u) / world : 0u;
}
// max over ranks: the size every rank's message is padded to (rt_padded_tiles)
inline uint32_t rt_plan_padded_tiles(uint32_t H, uint32_t world) { return (rt_plan_tiles_total(H) + world - 1u) / world; }
inline size_t rt_plan_message_bytes(uint32_t W, uint32_t H, uint32_t world) { return (size_t)rt_plan_padded_tiles(H, world) * 8u * W * 4u; }

enum class RtXKind : uint8_t { AllGather, Send, Recv };
struct RtXOp {
    RtXKind kind;
    uint32_t peer;            // Send: the root; Recv: the sender; AllGather: unused
    size_t gather_offset;     // Recv: where the peer's tiles land in this rank's gather buffer; AllGather: offset of THIS rank's part (in place)
    size_t bytes;             // the padded message, the same for every rank
};
struct RtXPlan {
    size_t message = 0;               // bytes per rank
    bool receives = false;            // this rank ends up with the frame (and de-interleaves it)
    bool part_in_gather = false;      // the rank renders straight into its slot of the gather buffer ...
    size_t part_offset = 0;           // ... at this offset; otherwise into its own colour buffer of the frame's set
    std::vector<RtXOp> ops;           // in issue order (one RCCL group)
};

// root >= 0: gather to that rank (grouped send / recv); root = -1: every rank receives (all-gather in place)
inline RtXPlan rt_exchange_plan(uint32_t W, uint32_t H, uint32_t rank, uint32_t world, int root) {
    RtXPlan p;
    p.message = rt_plan_message_bytes(W, H, world);
    p.receives = root < 0 || (uint32_t)root == rank;
    p.part_in_gather = p.receives;
    p.part_offset = p.receives ? p.message * rank : 0u;
    if (root < 0) {
        p.ops.push_back(RtXOp{RtXKind::AllGather, 0u, p.message * rank, p.message});
    } else if ((uint32_t)root == rank) {
        for (uint32_t r = 0; r < world; ++r)
            if (r != rank) p.ops.push_back(RtXOp{RtXKind::Recv, r, p.message * r, p.message});
    } else {
        p.ops.push_back(RtXOp{RtXKind::Send, (uint32_t)root, 0u, p.message});
    }
    return p;
}

// Frame number f of a context (rt_render_gather calls so far) uses stream and buffer set f % kExchangeStreams.
inline uint32_t rt_exchange_set(uint32_t frames_rendered) { return frames_rendered % kExchangeStreams; }
// The exchange of the frame in event slot `slot` (frames enqueued since the last rt_wait) is ordered behind the exchange
// of the frame in this slot: every rank then runs the exchanges of one communicator in the order of the calls, whatever
// streams they sit on.  -1: none.
inline int rt_exchange_waits_on(uint32_t slot) { return slot > 0u ? (int)slot - 1 : -1; }

// Where tile t of the frame lies after the exchange: in the gather buffer [world][padded][8][W][4] at this byte offset
// (assemble_frame reads it from there); tile t belongs to rank t % world and is that rank's tile number t / world.
inline size_t rt_gathered_tile_offset(uint32_t W, uint32_t H, uint32_t world, uint32_t tile) {
    return rt_plan_message_bytes(W, H, world) * (tile % world) + (size_t)(tile / world) * 8u * W * 4u;
}
