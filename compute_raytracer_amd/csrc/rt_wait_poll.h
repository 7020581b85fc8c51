// rt_wait_poll.h -- how rt_wait waits for frames that end in an RCCL exchange (rt_comm.hip).
//
// hipEventSynchronize cannot be told that a peer died: a rank whose partner never enters the
// collective would sit in it forever.  So the wait is a poll: the frames' end events are queried, and
// between queries the communicator is asked for an asynchronous error (ncclCommGetAsyncError) and the
// caller's deadline is checked.  The loop is a template over its four environment calls so that the
// decision logic is unit-tested on a machine without a GPU (tests/c/wait_poll_test.cpp).
#pragma once
#include <cstdint>

enum class RtPollVerdict { Done, CommError, Timeout };

// done(i): has frame i completed?   comm_failed(): has the communicator reported an asynchronous error?
// now_ms(): monotonic clock.         idle(): yield between two rounds of queries.
// timeout_ms = 0: no deadline.  Frames are waited for in order; a completed frame is never queried again.
template <class Done, class CommFailed, class Now, class Idle>
RtPollVerdict rt_poll_until(uint32_t n_frames, uint32_t timeout_ms, Done&& done, CommFailed&& comm_failed, Now&& now_ms,
                            Idle&& idle) {
    const uint64_t t0 = now_ms();
    uint32_t next = 0;
    for (;;) {
        while (next < n_frames && done(next)) ++next;
        if (next == n_frames) return RtPollVerdict::Done;        // completion wins over an error that arrives with it
        if (comm_failed()) return RtPollVerdict::CommError;
        if (timeout_ms != 0u && now_ms() - t0 >= (uint64_t)timeout_ms) return RtPollVerdict::Timeout;
        idle();
    }
}
