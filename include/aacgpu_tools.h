/*
 * aacgpu_tools.h — measurement and diagnostic entry points of libaacgpu.so.  Nothing a host of the decode path needs
 * (include/aacgpu.h is that interface); bench.py, tools/ and the parity tests use these.  Same library, same C ABI rules.
 */
#ifndef AACGPU_TOOLS_H
#define AACGPU_TOOLS_H

#include "aacgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Calibration for the bench: a float4 copy of `bytes` (multiple of 16) device to device with the run kernels' launch
 * shape, enqueued on hip_stream.  Gives the copy rate of THIS box for a launch of that size, next to the run kernel. */
int aacg_calib_copy(void* d_dst, const void* d_src, size_t bytes, void* hip_stream);
/* Timing marks for the bench: HIP events created with hipEventDisableSystemFence — HIP's flag for events that only measure
 * time.  A default event performs a system-scope fence when it is recorded (cache write-back and invalidation), which a
 * region of twenty 12-us launches between two events pays for (the launch behind a mark starts on cold caches); these do
 * not.  They order nothing for the host: the caller synchronises the stream before asking for the elapsed time.          */
int aacg_timer_create(void** mark);
int aacg_timer_record(void* mark, void* hip_stream);
int aacg_timer_elapsed_ms(void* first, void* second, float* ms);
void aacg_timer_destroy(void* mark);
/* aacg_decode_pipelined with a timing mark bound to the launch's completion (its time stamp is the end of that dispatch; no marker
 * packet enters the queue): bench.py brackets its timed regions with the marks of the launches at their ends.  The mark also stands
 * for the engine's own completion event of that launch (one event per dispatch), so it must stay alive until the pipeline has
 * been joined (aacg_pipeline_join / aacg_synchronize) or a new sequence has begun. */
int aacg_decode_pipelined_timed(aacg_engine* e, aacg_plan* p, const void* d_coeffs, const aacg_band_meta* d_meta, void* d_pcm, void* stop_mark);
/* How many launches of aacg_decode_pipelined continued the launch before them through the cross-launch cells (and so were
 * allowed to overlap it) since the engine was made: tests assert that the route they mean to exercise was taken. */
uint64_t aacg_pipeline_chained(const aacg_engine* e);
/* 1 if the engine's internal streams were seen to run side by side, each pair of them, when the pipeline was set up (HIP multiplexes
 * streams onto a few hardware queues; two streams on one queue serialise): 0 = pipelined launches are correct but do not overlap. */
int aacg_pipeline_concurrent(const aacg_engine* e);
/* how many of the engine's streams the current pipelined sequence takes in turn (aacg_pipeline_streams, aacg_routes.h); 0 before the first */
int aacg_pipeline_streams_used(const aacg_engine* e);
/* How the host waits once a wait has lasted spin_us microseconds of polling (aacg_wait.h): 0 keep polling (round 5), 1 sched_yield
 * between polls, 2 sleep between polls (the default), 3 hipEventSynchronize on a blocking-sync event (UNBOUNDED: measurement only).
 * Before the first pipelined launch; spin_us < 0 keeps the default (20).  tools/micro/pipe_drive --wait-mode. */
int aacg_debug_set_wait_mode(aacg_engine* e, int mode, double spin_us);
/* What is in flight on the engine right now, as text (what an AACG_ERR_TIMEOUT's aacg_last_error carries): launch counts, every
 * stream's and completion event's state, the cross-launch rendezvous cells' state words.  Returns the text's length. */
int aacg_debug_in_flight(aacg_engine* e, char* dst, size_t n);
/* Tests of the bounded waits: one of the engine's streams (0: its own, 1..3: the pipeline's) busy for `ms` milliseconds (<= 5000)
 * with a one-wave kernel that polls a word nobody sets.  Nothing on the decode path calls this. */
int aacg_debug_stall(aacg_engine* e, int which, uint32_t ms);

/* Diagnostic: the IMDCT stage of the kernels on its own, for known-answer tests against the reference's MDCT.process
 * (mdct.js:62-115) and FFT.process (fft.js:105-192) vectors.  One spectrum in (1024 floats: one long window, or eight
 * short ones), windows forced to 1.  Long: out[0..2047] = the 2048 IMDCT outputs.  Short: out[128 w + i] =
 * y_(w-1)[128 + i] + y_w[i] (i < 128), so a window whose neighbours are zero shows its 256 outputs.  identity_rotation:
 * the pre / post rotations (mdct.js:73-76, 82-87) are replaced by the identity, which leaves the N/4-point complex inverse
 * FFT of z[k] = X[N/2-1-2k] + i X[2k] in the output order of mdct.js:90-114.  is_short: bit 0 = eight short windows; bit 1 =
 * the int16 seam's variant of the stage (mirror-lane exchanges as DPP moves, long columns dealt out by long_col).  Nothing on the decode path calls this. */
int aacg_debug_transform(int device_ordinal, int sample_index, int is_short, int identity_rotation, const float* in, float* out);

/* Diagnostic: route choices a parity test wants to make by hand; 0 (the default) = the engine's own choice.  Nothing on the
 * decode path calls this.  AACG_DEBUG_ROUTE_UNFUSED_COUPLING: independent coupling (cce.js:121-128) as the separate pass over
 * the interleaved PCM (aacg_couple_pcm, what plans with double-duty runs take) even where the engine would apply it in the
 * targets' epilogues (aacg_imdct_run_*_cpl): the two routes must produce the same bits.
 * AACG_DEBUG_ROUTE_RECOMPUTE: chains longer than a run the old way — every later run recomputes
 * the frame before it (aacg_imdct_run_*_dd) — instead of the run-to-run rendezvous (aacg_imdct_run_*_rv) the engine takes for plain
 * batches; both must produce the same bits.  Set before the plan is made. */
#define AACG_DEBUG_ROUTE_UNFUSED_COUPLING 1
#define AACG_DEBUG_ROUTE_RECOMPUTE        8
int aacg_debug_set_route(aacg_engine* e, int flags);


/* The route decision on its own (aacg_pick_route, aacg_routes.cpp: the one function launch_run executes and
 * aacg_plan_kernels prints), for tests without a device: input / output kind and debug flags of a hypothetical engine,
 * AACG_ROUTE_PLAN_* flags of a hypothetical planned batch, pipelined != 0: launched through aacg_decode_pipelined.
 * Writes the launches by kernel name; AACG_ERR_UNSUPPORTED if the route names a run kernel that is not registered. */
#define AACG_ROUTE_PLAN_TNS              0x001   /* TNS records (AACG_TNS_SPEC engine) */
#define AACG_ROUTE_PLAN_PNS              0x002   /* noise bands (AACG_PNS_SPEC engine) */
#define AACG_ROUTE_PLAN_LONG_CHAINS      0x004   /* some chain is longer than a run */
#define AACG_ROUTE_PLAN_FULL_LATER_RUNS  0x008   /* ... with a later run of 16 frames (double duty on the recompute route) */
#define AACG_ROUTE_PLAN_WIDE_FRAMES      0x010   /* at least half of the units belong to frames of more than two channels */
#define AACG_ROUTE_PLAN_CCE_INDEPENDENT  0x020   /* independently switched coupling elements */
#define AACG_ROUTE_PLAN_CCE_DEPENDENT    0x040   /* coupling in the spectral domain */
#define AACG_ROUTE_PLAN_NO_RUNS          0x080   /* no element with a filterbank pass of its own in the main runs */
int aacg_debug_route(int input_kind, int output_kind, int debug_route, int plan_flags, int pipelined, char* dst, size_t n);
/* The registered run kernels: the index-th symbol into dst, returns its switches (the key launch_run looks it up by), < 0 past the end. */
int aacg_debug_run_kernel(int index, char* dst, size_t n);
/* The ordering rule of pipelined launches (aacg_pipeline_order, aacg_routes.cpp) for a sequence on `streams` streams: stream of launch n; the round (one launch per
 * stream) whose completion events the host waits for before enqueuing it (-1: none); whether its own completion gets an event;
 * the launch up to which everything is known complete when it is enqueued (-1: nothing).  Returns the number of rotating
 * overlap buffers the rule must cover. */
int aacg_debug_pipeline_order(unsigned long long n, int streams, int* stream, long long* sync_round, int* marked, long long* complete_upto);

#ifdef __cplusplus
}
#endif
#endif /* AACGPU_TOOLS_H */
