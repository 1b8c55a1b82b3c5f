/*
 * aacg_routes.h — the run kernels as a registry, and the ONE decision which of them a plan takes.
 *
 * imdct_run_body (aacg_kernels.h) is instantiated per variant in translation units of their own (each its own code object, so
 * that adding to one never moves another's code); every such unit exports a table of {switches, symbol, host stub}.  The
 * engine turns a plan into a route — the staging launches in front, the run kernel by its switches, the launches behind —
 * in one function (aacg_pick_route, aacg_engine.hip): launch_run executes that descriptor, aacg_plan_kernels prints it.
 * Reference for what every route computes: src/decoder.js:218-248 + src/filter_bank.js:88-204.
 */
#ifndef AACG_ROUTES_H
#define AACG_ROUTES_H

#include "aacg_device.h"

/* template switches of imdct_run_body = key of a run kernel */
enum {
    AACG_RK_QUANT = 1,      /* KIND = AACG_INPUT_QUANT_I16 (else f32 spectra) */
    AACG_RK_I16   = 2,      /* OUT = AACG_OUTPUT_I16 */
    AACG_RK_DD    = 4,      /* double duty: the first wave of a full later run recomputes the frame before it */
    AACG_RK_EX    = 8,      /* the optional stages (AACG_TNS_SPEC, AACG_PNS_SPEC) inside the run */
    AACG_RK_CPL   = 16,     /* independent coupling applied where the target's PCM is formed */
    AACG_RK_RV    = 32,     /* rendezvous cells between the runs of a chain (and, pipelined, between launches); takes aacg_rv_args */
    AACG_RK_NT    = 64      /* non-temporal loads of the spectra: batches of multichannel frames */
};

struct aacg_run_kernel {
    unsigned    key;        /* AACG_RK_* */
    const char* name;       /* the symbol a rocprofv3 kernel trace shows */
    const void* fn;         /* host stub, for hipLaunchKernel */
    bool preloaded = false; /* AACG_RUN_KERNEL_PRE signature: six leading pointer arguments (run table, tables, links, units, spectra,
                               band words) that arrive in SGPRs with the wave (-amdgpu-kernarg-preload-count), then the two
                               argument records */
};
/* The signature of a run kernel whose early pointers are preloaded: by-value struct arguments are not preloaded, so the pointers
 * a wave needs for its first loads travel once more as leading scalar arguments — the table loads and the run record's batch go
 * out with the wave's first instructions, one dependent round trip earlier (0.2 us per launch on the headline route). */
#define AACG_RUN_KERNEL_PRE(name, ...) \
    extern "C" __global__ __launch_bounds__(AACG_WG_THREADS) \
    void name(const aacg_run* runs, const aacg_tables* tab, const aacg_rv_link* links, const aacg_dev_unit* units, const void* coeffs, \
              const aacg_band_meta* meta, const aacg_kparams P, const aacg_rv_args V) \
    { imdct_run_body<__VA_ARGS__, true>(P, &V, runs, tab, links, units, coeffs, meta); }

/* one table per translation unit */
extern const aacg_run_kernel aacg_run_kernels_plain[];  extern const int aacg_run_kernels_plain_n;    /* aacg_engine.hip */
extern const aacg_run_kernel aacg_run_kernels_rv[];     extern const int aacg_run_kernels_rv_n;       /* aacg_engine_rv.hip */
extern const aacg_run_kernel aacg_run_kernels_nt[];     extern const int aacg_run_kernels_nt_n;       /* aacg_engine_nt.hip */
extern const aacg_run_kernel aacg_run_kernels_ext[];    extern const int aacg_run_kernels_ext_n;      /* aacg_engine_ext.hip */
extern const aacg_run_kernel aacg_run_kernels_i16[];    extern const int aacg_run_kernels_i16_n;      /* aacg_engine_i16.hip */
extern const aacg_run_kernel aacg_run_kernels_exrun[];  extern const int aacg_run_kernels_exrun_n;    /* aacg_engine_exrun.hip */
extern const aacg_run_kernel aacg_run_kernels_couple[]; extern const int aacg_run_kernels_couple_n;   /* aacg_engine_couple.hip */

/* what launch_run does for a plan: the single statement of the route */
enum { AACG_STAGE_NONE = 0, AACG_STAGE_SPECTRAL_EX = 1, AACG_STAGE_DEPENDENT_COUPLING = 2 };
struct aacg_route {
    int      stage;                   /* launches in front of the run kernel that leave f32 spectra in HBM (AACG_STAGE_*) */
    bool     stage_quant;             /* ... from quantised input */
    bool     has_run;                 /* the plan has main runs */
    unsigned run_key;                 /* ... launched with this kernel (AACG_RK_*) */
    bool     rv;                      /* it walks the rendezvous cut of the chains (runs_rv / links_rv) */
    bool     has_side;                /* independently switched coupling elements: their own filterbank pass into the side buffer */
    unsigned side_key;
    bool     side_first;              /* ... in front of the run kernel (fused coupling) instead of behind it */
    bool     couple_pcm;              /* aacg_couple_pcm over the finished PCM behind everything (coupling not fused) */
    bool     overlappable;            /* consecutive launches of the plan may overlap (aacg_decode_pipelined): rendezvous kernels, no staging */
};

#ifdef AACG_HOST_H
/* aacg_routes.cpp (plain C++, shared with the lane emulator of tests/emu).
 * aacg_pick_route: flags of the engine and of the planned batch -> the route.  `pipelined`: the launch comes through
 * aacg_decode_pipelined; `tracing`: a -DAACG_PROFILE build with per-wave time stamps (keeps the plain kernels). */
aacg_route aacg_pick_route(int input_kind, int output_kind, int debug_route, bool tracing, const aacg_plan_host& h, bool pipelined);
/* the symbol of the run kernel with these switches: "aacg_imdct_run_" quant|f32 [_ex][_dd][_cpl][_rv][_i16][_nt] */
std::string aacg_run_kernel_name(unsigned key);
/* the launches of a route by kernel name, " + " between them: what a rocprofv3 kernel trace of the batch shows */
std::string aacg_route_names(const aacg_route& r, bool any_tns);

/* How the n-th launch of a pipelined sequence (aacg_decode_pipelined) is ordered against the ones before it: the ONE place the
 * rule lives — the engine issues by it, the lane emulator schedules workgroups by it, tests/test_routes.py walks it and checks
 * that it keeps every launch behind all launches up to n - AACG_OV_BUFFERS + 1 (aacg_device.h).
 * Launches go to `streams` streams in turn (a ROUND = one launch per stream; aacg_pipeline_streams picks 2 or AACG_PIPE_STREAMS for a
 * sequence).  The launches of every AACG_PIPE_MARK-th
 * round carry a completion event; before the host enqueues the first launch of such a round it WAITS (host side, no packet in
 * any GPU queue) for the events of the round AACG_PIPE_DEPTH(streams) rounds back — a stream's launch complete means its earlier ones are,
 * so everything up to that round's end is complete before anything of this round and the next AACG_PIPE_MARK - 1 exists. */
#define AACG_PIPE_MARK  2
/* rounds between the one the host waits for and the one it is about to enqueue: as many as the buffers allow (the deeper, the
 * longer the host may be away before a queue runs dry) — the last launch enqueued before the next wait is
 * S (check + AACG_PIPE_MARK) - 1, everything up to S (check - depth) + S - 1 is known complete, and the distance must stay
 * below AACG_OV_BUFFERS: S (AACG_PIPE_MARK + depth) - S <= AACG_OV_BUFFERS - 1; even, so that the round is a marked one */
#define AACG_PIPE_DEPTH(S) ((((AACG_OV_BUFFERS - 1 + (S)) / (S) - AACG_PIPE_MARK) / AACG_PIPE_MARK) * AACG_PIPE_MARK)
struct aacg_pipe_order {
    int     stream;                   /* which of the internal streams: in-order behind its earlier launches */
    int64_t sync_round;               /* the host waits for this round's events before enqueuing launch n, or -1 */
    bool    marked;                   /* its completion gets an event (slot: round / AACG_PIPE_MARK mod AACG_PIPE_RING, stream) */
    int64_t complete_upto;            /* every launch up to this one is KNOWN complete when launch n is enqueued, or -1 */
};
#define AACG_PIPE_RING  8             /* marked rounds whose events are kept: > AACG_PIPE_DEPTH(1) / AACG_PIPE_MARK */
aacg_pipe_order aacg_pipeline_order(uint64_t n, int streams);
/* How many streams a pipelined sequence of this plan's launches takes in turn — measured, route by route (same box, interleaved,
 * tools/ab.sh; us per launch with two / three streams):
 *   plain kernels, f32 PCM, a launch = one round of workgroups   config 2: 11.6-11.8 / 11.3;  config 4 shape: 11.6 / 11.4;  f32 seam: - / 12.2
 *   the same with int16 PCM                                       10.0 / 10.9
 *   optional stages inside the run kernel (TNS on config 3)       int16 seam 23.7-23.9 / 24.0;  f32 seam 25.5 / 25.0-25.2
 *   plain kernels on frames of eight short windows only             11.4 / 12.0  (frames of long windows: 11.5 / 11.1; config 3's mix, one in four: 11.4 / 11.4)
 *   a launch of several rounds of workgroups (config 5 shape)     58.8-59.3 / 63.4-64.2 (a third launch in flight scatters a stream's
 *                                                                 elements over more of the L2s)
 * Three where a CU that is done with launch n + 1's workgroup would otherwise find nothing of launch n + 2 to run; two where the
 * launches in flight already contend for the write path more than that wait costs. */
int aacg_pipeline_streams(const aacg_plan_host& h, unsigned run_key);
#endif

#endif
