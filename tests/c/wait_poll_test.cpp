// The decision logic of rt_wait for frames that end in an RCCL exchange (compute_raytracer_amd/csrc/rt_wait_poll.h),
// driven by stub environments: no GPU, no RCCL.  Built and run by tests/test_wait_poll_cpu.py.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../compute_raytracer_amd/csrc/rt_wait_poll.h"

#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) { std::printf("FAILED line %d: %s\n", __LINE__, #cond); std::exit(1); } \
    } while (0)

struct Env {
    std::vector<uint64_t> done_at;    // frame i completes at this clock value
    uint64_t fail_at = ~0ull;         // the communicator reports an error from this clock value on
    uint64_t clock = 0;
    uint32_t queries = 0, comm_queries = 0, idles = 0;
    std::vector<uint32_t> query_log;
    RtPollVerdict run(uint32_t timeout_ms) {
        return rt_poll_until((uint32_t)done_at.size(), timeout_ms,
                             [&](uint32_t i) { ++queries; query_log.push_back(i); return clock >= done_at[i]; },
                             [&]() { ++comm_queries; return clock >= fail_at; },
                             [&]() { return clock; },
                             [&]() { ++idles; ++clock; });
    }
};

int main() {
    {   // nothing in flight: done at once, nothing is queried
        Env e;
        CHECK(e.run(0) == RtPollVerdict::Done && e.queries == 0 && e.comm_queries == 0 && e.idles == 0);
    }
    {   // frames complete in order; no deadline; a completed frame is never queried again
        Env e; e.done_at = {3, 3, 7};
        CHECK(e.run(0) == RtPollVerdict::Done);
        CHECK(e.clock == 7);
        uint32_t zero = 0, one = 0;
        for (uint32_t q : e.query_log) { zero += q == 0; one += q == 1; }
        CHECK(zero == 4 && one == 1);            // frame 0: clock 0,1,2,3; frame 1: once, at clock 3
        CHECK(e.comm_queries == 7);              // one per round that did not finish
    }
    {   // frames complete out of order (different streams): the wait still ends when all are done
        Env e; e.done_at = {9, 2, 5};
        CHECK(e.run(0) == RtPollVerdict::Done && e.clock == 9);
    }
    {   // a peer dies: the communicator's asynchronous error ends the wait although the frames never complete
        Env e; e.done_at = {~0ull, ~0ull}; e.fail_at = 4;
        CHECK(e.run(0) == RtPollVerdict::CommError && e.clock == 4);
    }
    {   // ... also with a deadline that is further away
        Env e; e.done_at = {~0ull}; e.fail_at = 4;
        CHECK(e.run(100) == RtPollVerdict::CommError && e.clock == 4);
    }
    {   // deadline: no error is ever reported, the frames never complete
        Env e; e.done_at = {1, ~0ull};
        CHECK(e.run(10) == RtPollVerdict::Timeout && e.clock == 10);
    }
    {   // completion at the very round the error shows: the frames are complete, the caller gets them
        Env e; e.done_at = {5}; e.fail_at = 5;
        CHECK(e.run(0) == RtPollVerdict::Done);
    }
    {   // the deadline counts from the call, not from the epoch
        Env e; e.done_at = {1005}; e.clock = 1000;
        CHECK(e.run(10) == RtPollVerdict::Done && e.clock == 1005);
        Env f; f.done_at = {1020}; f.clock = 1000;
        CHECK(f.run(10) == RtPollVerdict::Timeout && f.clock == 1010);
    }
    std::printf("wait poll ok\n");
    return 0;
}
