// The multi-GPU exchange as rt_comm.hip executes it (compute_raytracer_amd/csrc/rt_exchange_plan.h), checked for whole groups
// without a device: every rank's plan against every other rank's.  Built and run by tests/test_exchange_plan_cpu.py.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "../../compute_raytracer_amd/csrc/rt_exchange_plan.h"

#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) { std::printf("FAILED line %d: %s (world %u root %d H %u)\n", __LINE__, #cond, g_world, g_root, g_H); std::exit(1); } \
    } while (0)
static uint32_t g_world, g_H; static int g_root;

// one frame of a group: plans of all ranks
static void check_frame(uint32_t W, uint32_t H, uint32_t world, int root) {
    g_world = world; g_root = root; g_H = H;
    std::vector<RtXPlan> plan;
    for (uint32_t r = 0; r < world; ++r) plan.push_back(rt_exchange_plan(W, H, r, world, root));
    const size_t msg = rt_plan_message_bytes(W, H, world);
    const uint32_t padded = rt_plan_padded_tiles(H, world);
    CHECK(msg == (size_t)padded * 8u * W * 4u);
    uint32_t tiles = 0, most = 0;
    for (uint32_t r = 0; r < world; ++r) { const uint32_t t = rt_plan_tiles_of_rank(H, r, world); tiles += t; most = std::max(most, t); }
    CHECK(tiles == rt_plan_tiles_total(H) && most == padded);              // every tile has one owner; the padding is the largest share
    for (uint32_t r = 0; r < world; ++r) {
        const RtXPlan& p = plan[r];
        CHECK(p.message == msg);
        CHECK(p.receives == (root < 0 || (uint32_t)root == r));
        CHECK(p.part_in_gather == p.receives);                              // a receiving rank renders straight into its slot
        if (p.receives) CHECK(p.part_offset == msg * r);
        CHECK((size_t)rt_plan_tiles_of_rank(H, r, world) * 8u * W * 4u <= msg);   // what the rank renders fits its message
    }
    if (root < 0) {
        // all-gather: one operation per rank, in place (own part at rank * msg), equal counts
        for (uint32_t r = 0; r < world; ++r) {
            CHECK(plan[r].ops.size() == 1 && plan[r].ops[0].kind == RtXKind::AllGather);
            CHECK(plan[r].ops[0].bytes == msg && plan[r].ops[0].gather_offset == msg * r);
        }
    } else {
        // gather to root: every send has exactly one matching receive of equal size; the receives and the root's own part tile
        // the gather buffer [world][msg] without gap or overlap; nobody else talks
        std::map<uint32_t, size_t> recv_from;                               // sender -> offset
        for (const RtXOp& op : plan[root].ops) {
            CHECK(op.kind == RtXKind::Recv && op.bytes == msg && op.peer != (uint32_t)root && op.peer < world);
            CHECK(recv_from.insert({op.peer, op.gather_offset}).second);    // one receive per sender
            CHECK(op.gather_offset == msg * op.peer);                       // lands where assemble_frame expects rank op.peer's tiles
        }
        CHECK(recv_from.size() == world - 1u);
        std::vector<size_t> starts;
        for (auto& kv : recv_from) starts.push_back(kv.second);
        starts.push_back(plan[root].part_offset);
        std::sort(starts.begin(), starts.end());
        for (uint32_t i = 0; i < world; ++i) CHECK(starts[i] == msg * i);
        for (uint32_t r = 0; r < world; ++r) {
            if (r == (uint32_t)root) continue;
            CHECK(plan[r].ops.size() == 1 && plan[r].ops[0].kind == RtXKind::Send);
            CHECK(plan[r].ops[0].peer == (uint32_t)root && plan[r].ops[0].bytes == msg);
            CHECK(recv_from.count(r) == 1);
        }
    }
    // every tile of the frame is found where the de-interleave looks for it: inside its owner's message, at the owner's
    // local tile number, and no two tiles share a place
    std::vector<size_t> where;
    for (uint32_t t = 0; t < rt_plan_tiles_total(H); ++t) {
        const size_t off = rt_gathered_tile_offset(W, H, world, t);
        const uint32_t owner = t % world;
        CHECK(off >= msg * owner && off + (size_t)8u * W * 4u <= msg * (owner + 1u));
        CHECK((off - msg * owner) / ((size_t)8u * W * 4u) == t / world && t / world < rt_plan_tiles_of_rank(H, owner, world));
        where.push_back(off);
    }
    std::sort(where.begin(), where.end());
    CHECK(std::adjacent_find(where.begin(), where.end()) == where.end());
}

// several frames in flight: the order of the communicator's operations, as each rank's streams will execute them
static void check_in_flight(uint32_t world, int root, uint32_t frames) {
    g_world = world; g_root = root;
    // per rank: the global sequence of (frame, op kind, peer) in the order the device is made to run them.  Frame f sits on
    // stream f % 4; its exchange waits for the exchange of the frame in the slot before it, so exchanges run in call order.
    std::vector<std::vector<std::pair<uint32_t, RtXOp>>> seq(world);
    for (uint32_t r = 0; r < world; ++r) {
        std::vector<int> done_before(frames, -1);
        for (uint32_t f = 0; f < frames; ++f) {
            const uint32_t slot = f;                                         // nothing waited for in between
            CHECK(rt_exchange_set(f) == f % kExchangeStreams && rt_exchange_set(f) < kExchangeStreams);
            const int w = rt_exchange_waits_on(slot);
            CHECK(w == (int)slot - 1);                                       // a chain: slot 0 free, slot s behind slot s - 1
            for (const RtXOp& op : rt_exchange_plan(1024, 2160, r, world, root).ops) seq[r].push_back({f, op});
        }
        // frames on the same stream are ordered by the stream; frames on different streams by the chain: in all cases by f
        for (size_t i = 1; i < seq[r].size(); ++i) CHECK(seq[r][i - 1].first <= seq[r][i].first);
    }
    // pairing across ranks, frame by frame: what rank a sends to b in frame f, b receives from a in frame f, and both ranks have
    // completed the same number of exchanges with each other before it
    for (uint32_t a = 0; a < world; ++a)
        for (uint32_t b = 0; b < world; ++b) {
            if (a == b) continue;
            std::vector<uint32_t> sends, recvs;
            for (auto& e : seq[a]) if (e.second.kind == RtXKind::Send && e.second.peer == b) sends.push_back(e.first);
            for (auto& e : seq[b]) if (e.second.kind == RtXKind::Recv && e.second.peer == a) recvs.push_back(e.first);
            CHECK(sends == recvs);
        }
    if (root < 0)
        for (uint32_t r = 1; r < world; ++r) CHECK(seq[r].size() == seq[0].size());     // the same number of collectives everywhere
}

int main() {
    const uint32_t heights[] = {8, 9, 64, 846, 1080, 2160, 4320, 7};
    for (uint32_t world : {1u, 2u, 3u, 4u, 8u})
        for (int root : {-1, 0, (int)world - 1})
            for (uint32_t H : heights)
                for (uint32_t W : {8u, 1344u, 3840u}) check_frame(W, H, world, root);
    // BASELINE C4 / C5: 270 tiles over 8 ranks = 33 or 34 each, padded to 34; 540 -> 67 / 68
    CHECK(rt_plan_padded_tiles(2160, 8) == 34 && rt_plan_tiles_of_rank(2160, 0, 8) == 34 && rt_plan_tiles_of_rank(2160, 7, 8) == 33);
    CHECK(rt_plan_padded_tiles(4320, 8) == 68 && rt_plan_tiles_of_rank(4320, 3, 8) == 68 && rt_plan_tiles_of_rank(4320, 4, 8) == 67);
    CHECK(rt_plan_message_bytes(3840, 2160, 8) == (size_t)34 * 8 * 3840 * 4);
    for (uint32_t world : {2u, 3u, 4u, 8u})
        for (int root : {-1, 0, (int)world - 1})
            for (uint32_t frames = 1; frames <= 6; ++frames) check_in_flight(world, root, frames);
    std::printf("exchange plan ok\n");
    return 0;
}
