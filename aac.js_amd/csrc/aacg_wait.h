/*
 * aacg_wait.h — every place the library's host code waits for the GPU waits HERE: bounded, and light on the caller's core.
 *
 * Round 5 left two things open: the back-pressure of aacg_decode_pipelined was a bare
 * `while (hipEventQuery(ev) == hipErrorNotReady) {}` — a core at 100 % in the caller's thread — and aacg_synchronize / aacg_wait /
 * aacg_get_overlap / aacg_pipeline_decode waited in hipStreamSynchronize / hipDeviceSynchronize / hipEventSynchronize without bound,
 * so the one unexplained ten-minute stall of the GPU suite was a HANG for whoever sits behind `readChunk()`
 * (reference src/decoder.js:125-216 returns or throws; it never blocks).  Now a wait polls (hipEventQuery / hipStreamQuery):
 * it spins for the first microseconds (the common case: the GPU is a launch or two behind), then backs off (yield or sleep, by
 * policy), and gives up after the engine's wait limit with hipErrorNotReady — which the callers turn into AACG_ERR_TIMEOUT plus a
 * dump of what was in flight (aacg_last_error).
 *
 * Host code only.
 */
#ifndef AACG_WAIT_H
#define AACG_WAIT_H

#include <hip/hip_runtime.h>
#include <sched.h>
#include <time.h>

#include <chrono>

enum {
    AACG_WAIT_SPIN  = 0,   /* poll without pause (round 5's behaviour; measurement only) */
    AACG_WAIT_YIELD = 1,   /* spin_us of polling, then sched_yield() between polls */
    AACG_WAIT_SLEEP = 2,   /* spin_us of polling, yield_us of polls with sched_yield() between them, then nanosleep between polls
                              (5 us doubling to 50 us): the default.  Measured on the headline route (tools/micro/pipe_drive
                              --wait-mode, profiles/r06_wait_modes.txt): the back-pressure waits of a caller that enqueues as fast as
                              the GPU decodes last about 50 us; sleeping through them costs 0.1 us per launch (a sleep of 5 us is one
                              of 55: the kernel's timer slack), yielding nothing — so a wait only sleeps once it is long        */
    AACG_WAIT_BLOCK = 3    /* spin_us of polling, then hipEventSynchronize (UNBOUNDED: measurement only, events only) */
};

struct aacg_wait_policy {
    double limit_s = 30.0;                  /* give up after this long: AACG_ERR_TIMEOUT (aacg_set_wait_limit_ms) */
    int    mode = AACG_WAIT_SLEEP;
    double spin_us = 20.0;
    double yield_us = 200.0;                /* AACG_WAIT_SLEEP: polls with sched_yield() between them before the first sleep */
};

namespace aacg_wait_detail {
template <class Query>
inline hipError_t poll(Query query, const aacg_wait_policy& w, hipEvent_t blockable)
{
    hipError_t st = query();
    if (st != hipErrorNotReady) return st;
    const auto t0 = std::chrono::steady_clock::now();
    const auto spin_end = t0 + std::chrono::nanoseconds((long long)(w.spin_us * 1e3));
    const auto deadline = t0 + std::chrono::nanoseconds((long long)(w.limit_s * 1e9));
    const auto yield_end = spin_end + std::chrono::nanoseconds((long long)(w.yield_us * 1e3));
    long pause_ns = 5000;
    for (;;) {
        st = query();
        if (st != hipErrorNotReady) break;
        const auto now = std::chrono::steady_clock::now();
        if (now >= deadline) break;
        if (now < spin_end || w.mode == AACG_WAIT_SPIN) continue;
        if (w.mode == AACG_WAIT_YIELD || (w.mode == AACG_WAIT_SLEEP && now < yield_end)) { sched_yield(); continue; }
        if (w.mode == AACG_WAIT_BLOCK && blockable) { st = hipEventSynchronize(blockable); break; }
        const timespec ts = {0, pause_ns};
        nanosleep(&ts, nullptr);
        if (pause_ns < 50000) pause_ns *= 2;
    }
    (void)hipGetLastError();                /* hipErrorNotReady is a status here, not an error to be found by the next call */
    return st;
}
}  // namespace aacg_wait_detail

/* hipSuccess: complete; hipErrorNotReady: the limit has passed; anything else: that HIP error */
inline hipError_t aacg_wait_event(hipEvent_t ev, const aacg_wait_policy& w)
{
    return aacg_wait_detail::poll([ev]() { return hipEventQuery(ev); }, w, ev);
}
/* everything enqueued on `s` so far (null: the legacy default stream) */
inline hipError_t aacg_wait_stream(hipStream_t s, const aacg_wait_policy& w)
{
    return aacg_wait_detail::poll([s]() { return hipStreamQuery(s); }, w, nullptr);
}

#endif /* AACG_WAIT_H */
