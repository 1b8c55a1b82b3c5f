/*
 * aacg_engine.hip — the C ABI of include/aacgpu.h on top of the gfx950 kernels.
 *
 * Replaces AACDecoder.prototype.process + the interleave of readChunk (reference
 * src/decoder.js:201-215, 218-334) for batches of frames from many streams.  There is no
 * CPU fallback: every entry point either runs the HIP kernels or returns an error.
 */
#include <hip/hip_runtime.h>

#include <chrono>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <algorithm>
#include <vector>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <sys/mman.h>

#include "aacg_kernels.h"
#include "aacg_host.h"
#include "aacg_routes.h"
#include "aacg_wait.h"

/* ---- kernels ----------------------------------------------------------------------- */
/* 1024 threads = 16 waves, one workgroup per CU: 4 waves per SIMD -> 128 VGPRs per lane */
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant(const aacg_kparams P) { imdct_run_body<AACG_INPUT_QUANT_I16>(P); }

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32(const aacg_kparams P) { imdct_run_body<AACG_INPUT_SPEC_F32>(P); }

/* Every other variant of the run kernel lives in a translation unit of its own (aacg_engine_{rv,nt,ext,i16,exrun,couple}.hip:
 * their own code objects, so that adding to them never moves the two kernels above); each exports a table of its kernels
 * (aacg_routes.h). */
const aacg_run_kernel aacg_run_kernels_plain[] = {
    {AACG_RK_QUANT, "aacg_imdct_run_quant", (const void*)aacg_imdct_run_quant},
    {0, "aacg_imdct_run_f32", (const void*)aacg_imdct_run_f32}
};
const int aacg_run_kernels_plain_n = 2;

/* aacg_engine_refresh.hip: a kept plan's unit records from the device parser's output */
void aacg_refresh_launch(aacg_dev_unit* units, const aacg_unit_desc* parsed, aacg_parse_result* results, const aacg_refresh_map* map, uint32_t n_units,
                         uint32_t max_units, int refuse_pns, uint32_t* refused, hipStream_t s);
/* aacg_engine_spectral.hip: the optional stages (AACG_PNS_SPEC noise bands, AACG_TNS_SPEC filters) -> f32 spectra */
int aacg_spectral_ex_set_lds_limits(void);
void aacg_spectral_ex_launch(bool quant, int n_units, hipStream_t s, const aacg_kparams& P);
void aacg_tns_matrices_launch(const aacg_dev_tns* d_recs, double* d_m, uint32_t n_records, hipStream_t s);
/* aacg_engine_couple.hip: AACG_CCE_SPEC */
void aacg_couple_launch(bool pcm, hipStream_t s, const aacg_couple_params& Q);
struct cce_bufs { const aacg_run* runs; const aacg_couple_job* jobs; const float* gains; float* side; };
struct rv_bufs { const aacg_run* runs; const aacg_rv_link* links; unsigned long long* state; float* data; };

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_spectral(const aacg_kparams P, int n_units) { spectral_body(P, n_units); }

/* Do two HIP streams run concurrently?  HIP multiplexes its streams onto a few hardware queues, and two streams that share one
 * serialise whatever the program meant (tools/micro/queue_map.hip).  One wave waits — bounded — for a word the other stream's
 * kernel sets: it sees it only if that kernel runs while this one is still running. */
extern "C" __global__ void aacg_probe_wait(const unsigned* flag, unsigned* seen, long long ticks)
{
    const long long t0 = wall_clock64();
    unsigned ok = 0;
    while (!ok && wall_clock64() - t0 < ticks) ok = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *seen = ok;
}
extern "C" __global__ void aacg_probe_set(unsigned* flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#define AACG_LDS_BYTES_SPECTRAL ((AACG_TAB_QUANT_FLOATS + AACG_WG_WAVES * 512) * 4)

/* ---- engine ------------------------------------------------------------------------ */
struct aacg_engine {
    aacg_config cfg;
    hipStream_t stream = nullptr;
    aacg_tables* d_tab = nullptr;
    aacg_pns_tables* d_pns = nullptr;       /* AACG_PNS_SPEC */
    unsigned long long rv_epoch = 0;        /* rendezvous epoch: one per launch of the _rv kernels, never 0 */
    float* d_overlap = nullptr;             /* [max_streams][max_channels][AACG_OV_BUFFERS][1024] */
    std::vector<uint8_t> parity;            /* live buffer (0..AACG_OV_BUFFERS-1) per (stream, channel) */
    uint64_t epoch = 0;                     /* bumped whenever `parity` changes: lets a relaunched plan skip its check */
    aacg_tables h_tab;
    aacg_host_windows h_win;
    /* pipelined launches (aacg_decode_pipelined): AACG_PIPE_STREAMS internal streams taken in turn, so that a launch starts on the
     * CUs the launches before it have left while those are still finishing, and a CU that is done with launch n + 1's workgroup
     * finds one of launch n + 2 waiting.  The chains of consecutive launches meet in cross-launch cells (aacg_xl_cell); how far
     * a launch may run ahead of the ones whose overlap buffers and cells it reuses is bounded by the HOST: it does not enqueue
     * a round of launches before the round AACG_PIPE_DEPTH(streams) back is complete (aacg_pipeline_order, aacg_routes.cpp, is the rule in
     * one place; aacg_device.h has the arithmetic). */
    struct pipe_t {
        hipStream_t stream[AACG_PIPE_STREAMS] = {};
        hipEvent_t mark[AACG_PIPE_RING][AACG_PIPE_STREAMS] = {};   /* completion events of the marked rounds' launches (aacg_pipeline_order) */
        hipEvent_t seen[AACG_PIPE_RING][AACG_PIPE_STREAMS] = {};   /* the event that stands for each of them: the engine's own, or a caller's timing
                                                                       mark bound to that launch (aacg_decode_pipelined_timed) — one event per dispatch */
        hipEvent_t tail[AACG_PIPE_STREAMS] = {};   /* joins: everything on stream k so far */
        hipEvent_t fork = nullptr;
        uint64_t n = 0;                     /* overlappable launches of the current sequence so far (aacg_pipeline_order(n, streams)) */
        int streams = AACG_PIPE_STREAMS;    /* of the current sequence (aacg_pipeline_streams) */
        uint64_t issued = 0;                /* every launch through the pipeline */
        hipStream_t joined_stream = nullptr; uint64_t joined_n = ~0ull;   /* the stream most recently put behind the pipeline, and at which launch count */
        bool open = false;                  /* launches issued since the last join that nobody outside is ordered behind yet */
        bool concurrent = false;            /* the streams were seen to run side by side, each pair of them (pipe_setup's probe) */
        bool serial = false;                /* the most recent launch was one that cannot overlap: on stream[0], behind its predecessor */
        aacg_plan* plan = nullptr;          /* the plan of launch n - 1 */
        unsigned long long epoch = 0;       /* rv epoch of launch n - 1 */
        uint64_t chained = 0;               /* launches that continued their predecessor through the cross-launch cells (introspection) */
    } pipe;
    aacg_xl_cell* d_xl_cells = nullptr;     /* [max_streams][max_channels][AACG_OV_BUFFERS] */
    float* d_xl_head = nullptr;             /* [max_streams][max_channels][AACG_OV_BUFFERS][1024] */
    /* host-buffer path: two pipeline slots (stream + device buffers grown on demand) */
    struct slot_t {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr, kernel_done = nullptr;
        bool busy = false;
        void* d_units = nullptr;  size_t units_cap = 0;
        void* d_runs = nullptr;   size_t runs_cap = 0;
        void* d_coeffs = nullptr; size_t coeffs_cap = 0;
        void* d_meta = nullptr;   size_t meta_cap = 0;
        void* d_tns = nullptr;    size_t tns_cap = 0;
        void* d_scratch = nullptr; size_t scratch_cap = 0;
        void* d_spec = nullptr;   size_t spec_cap = 0;       /* PNS route: f32 spectra between the two kernels */
        void* d_cce[4] = {nullptr, nullptr, nullptr, nullptr}; size_t cce_cap[4] = {0, 0, 0, 0};   /* AACG_CCE_SPEC: runs, jobs, gains, side PCM */
        void* d_rv[4] = {nullptr, nullptr, nullptr, nullptr}; size_t rv_cap[4] = {0, 0, 0, 0};     /* _rv kernels: runs, links, rendezvous state, payload */
        void* d_pcm = nullptr;    size_t pcm_cap = 0;
        /* page-locked staging for callers that pass ordinary (pageable) memory */
        void* h_in = nullptr;     size_t h_in_cap = 0;
        void* h_pcm = nullptr;    size_t h_pcm_cap = 0;
        void* user_pcm = nullptr; size_t user_pcm_bytes = 0;   /* copy-back target at aacg_wait, or null */
        aacg_plan_host h;
    } slot[2];
    /* plans: device buffers come from a free list and are filled by asynchronous copies on the engine's own stream, so that
     * creating or destroying a plan does not wait for kernels that are running (hipMalloc / hipMemcpy / hipFree would).
     * No stream of its own for that: one more stream shifted HIP's stream -> hardware-queue assignment, and a caller's two
     * streams ended up on one queue (two batches in flight: 12.0 -> 13.5 us per step) */
    std::vector<std::pair<void*, size_t>> pool;
    uint64_t submitted = 0;
    hipEvent_t last_kernel = nullptr;       /* completion of the most recently submitted batch's kernel */
    void* d_trace = nullptr;                /* profiling: per-wave phase timestamps when (ablate & 16) */
    int debug_route = 0;                    /* aacg_debug_set_route: diagnostic route choices for parity tests (0 in production) */
    int ablate = 0;                         /* -DAACG_PROFILE builds: env AACG_ABLATE (aacg_kernels.h); always 0 in the shipped library */
    /* host waits (aacg_wait.h): bounded by wait.limit_s, AACG_ERR_TIMEOUT + a dump of what was in flight when it passes */
    aacg_wait_policy wait;
    std::vector<hipStream_t> foreign;       /* callers' streams this engine has enqueued work on (aacg_decode_device & co.): what "everything of this engine" must cover */
    float* h_ov = nullptr;                  /* page-locked 4 KB: aacg_get_overlap / aacg_set_overlap staging (an asynchronous copy the wait can bound) */
    bool wedged = false;                    /* a wait has timed out: destruction leaks instead of waiting again */
    hipStream_t dump_stream = nullptr;      /* in_flight(): the cells are read by a copy on a stream of its own, into page-locked memory made at aacg_create */
    aacg_xl_cell* h_dump = nullptr;         /* (nothing is allocated or freed while a device may be hung: hipHostFree / hipStreamDestroy wait for it) */
    std::string pipe_note;                  /* what pipe_setup found (one hardware queue, ...): told once through aacg_last_error */
    std::string err;
};

#define AACG_PLAN_BUFFERS 13
struct aacg_plan {
    aacg_engine* e;
    aacg_plan_host h;
    uint32_t n_units = 0;
    aacg_dev_unit* d_units = nullptr;
    aacg_run* d_runs = nullptr;
    aacg_dev_tns* d_tns = nullptr;
    float* d_scratch = nullptr;             /* parked predecessor tails of double-duty runs */
    float* d_spec = nullptr;                /* PNS route: f32 spectra between the two kernels */
    void*  d_cce[4] = {nullptr, nullptr, nullptr, nullptr};   /* AACG_CCE_SPEC: coupling elements' runs, jobs, gains, side PCM */
    void*  d_rv[4] = {nullptr, nullptr, nullptr, nullptr};    /* _rv kernels: run table, link records, rendezvous state words and payload (two sets: overlapping launches) */
    size_t bytes[AACG_PLAN_BUFFERS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  /* sizes of the buffers above, for the engine's free list */
    hipEvent_t uploaded = nullptr;          /* the tables are on the device */
    hipEvent_t last_use = nullptr;          /* recorded at destruction on last_stream: everything launched with this plan */
    hipStream_t last_stream = nullptr;      /* stream of the most recent launch (no per-launch event: it costs 3 us per step) */
    uint32_t unit_sets = 1, cur_set = 0;    /* aacg_plan_set_unit_sets: d_units holds unit_sets copies of the records; launches read cur_set */
    const aacg_dev_unit* units_now() const { return d_units + (size_t)cur_set * n_units; }
    bool used = false;
    bool last_pipelined = false;            /* its most recent launch went through aacg_decode_pipelined */
    uint64_t seen_epoch = ~0ull;            /* engine epoch right after this plan's last launch */
    uint32_t launches = 0;
};

namespace {

bool hip_ok(aacg_engine* e, hipError_t rc, const char* what)
{
    if (rc == hipSuccess) return true;
    if (e) e->err = std::string(what) + ": " + hipGetErrorString(rc);
    return false;
}

#define HIP_TRY(e, call, code) do { if (!hip_ok((e), (call), #call)) return (code); } while (0)

size_t coef_elem_size(const aacg_engine* e) { return e->cfg.input_kind == AACG_INPUT_QUANT_I16 ? 2 : 4; }
size_t pcm_elem_size(const aacg_engine* e) { return e->cfg.output_kind == AACG_OUTPUT_I16 ? 2 : 4; }

/* smallest free block that fits without wasting more than half of it, else a new allocation */
void* pool_take(aacg_engine* e, size_t bytes, size_t* got)
{
    size_t best = e->pool.size();
    for (size_t i = 0; i < e->pool.size(); i++)
        if (e->pool[i].second >= bytes && e->pool[i].second <= 2 * bytes + 4096 && (best == e->pool.size() || e->pool[i].second < e->pool[best].second)) best = i;
    if (best != e->pool.size()) {
        void* p = e->pool[best].first;
        *got = e->pool[best].second;
        e->pool.erase(e->pool.begin() + (long)best);
        return p;
    }
    void* p = nullptr;
    if (!hip_ok(e, hipMalloc(&p, bytes), "hipMalloc (plan)")) return nullptr;
    *got = bytes;
    return p;
}

void pool_give(aacg_engine* e, void* p, size_t bytes)
{
    if (!p) return;
    if (e->pool.size() >= 48) { (void)hipFree(e->pool.front().first); e->pool.erase(e->pool.begin()); }
    e->pool.emplace_back(p, bytes);
}

int grow(aacg_engine* e, void** p, size_t* cap, size_t need)
{
    if (need <= *cap) return AACG_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    HIP_TRY(e, hipMalloc(p, need), AACG_ERR_OUT_OF_MEMORY);
    *cap = need;
    return AACG_OK;
}

int grow_host(aacg_engine* e, void** p, size_t* cap, size_t need)
{
    if (need <= *cap) return AACG_OK;
    if (*p) { (void)hipHostFree(*p); *p = nullptr; *cap = 0; }
    HIP_TRY(e, hipHostMalloc(p, need, hipHostMallocDefault), AACG_ERR_OUT_OF_MEMORY);
    *cap = need;
    return AACG_OK;
}

/* true if the runtime knows this host pointer as page-locked (hipHostMalloc / hipHostRegister) */
bool is_pinned(const void* p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

/* ---- bounded host waits (aacg_wait.h) ------------------------------------------------------------------------------ */
#define AACG_DUMP_CELLS 4096u
const char* event_state(hipEvent_t ev)
{
    if (!ev) return "-";
    const hipError_t st = hipEventQuery(ev);
    (void)hipGetLastError();
    return st == hipSuccess ? "done" : st == hipErrorNotReady ? "PENDING" : "error";
}
const char* stream_state(hipStream_t s)
{
    const hipError_t st = hipStreamQuery(s);
    (void)hipGetLastError();
    return st == hipSuccess ? "idle" : st == hipErrorNotReady ? "BUSY" : "error";
}

/* What was in flight, for the text of an AACG_ERR_TIMEOUT: the pipeline's counters, every stream's and completion event's state,
 * and the rendezvous cells' state words (read by a copy on a stream of its own, itself bounded: a hung kernel does not stop the
 * copy engines). */
std::string in_flight(aacg_engine* e)
{
    char b[512];
    std::string o;
    const aacg_engine::pipe_t& pp = e->pipe;
    std::snprintf(b, sizeof b, "pipeline: %llu launches issued, %llu in the current sequence on %d streams, %llu chained, open %d serial %d concurrent %d, rv epoch %llu; ",
                  (unsigned long long)pp.issued, (unsigned long long)pp.n, pp.streams, (unsigned long long)pp.chained, (int)pp.open, (int)pp.serial, (int)pp.concurrent, e->rv_epoch);
    o += b;
    o += std::string("engine stream ") + stream_state(e->stream);
    for (int k = 0; k < AACG_PIPE_STREAMS; k++) if (pp.stream[k]) { std::snprintf(b, sizeof b, ", pipe stream %d %s", k, stream_state(pp.stream[k])); o += b; }
    for (int k = 0; k < 2; k++) if (e->slot[k].stream) { std::snprintf(b, sizeof b, ", batch slot %d %s (busy %d, kernel %s, copy-back %s)", k, stream_state(e->slot[k].stream), (int)e->slot[k].busy, event_state(e->slot[k].kernel_done), event_state(e->slot[k].done)); o += b; }
    for (size_t k = 0; k < e->foreign.size(); k++) { std::snprintf(b, sizeof b, ", caller stream %p %s", (void*)e->foreign[k], stream_state(e->foreign[k])); o += b; }
    if (pp.stream[0]) {
        o += "; completion events [ring slot: per stream]";
        for (int r = 0; r < AACG_PIPE_RING; r++) {
            std::snprintf(b, sizeof b, " %d:", r); o += b;
            for (int k = 0; k < AACG_PIPE_STREAMS; k++) { o += k ? "/" : ""; o += event_state(pp.seen[r][k] ? pp.seen[r][k] : nullptr); }
        }
    }
    /* the cross-launch cells' state words by kind and epoch */
    const size_t n_cells = (size_t)e->cfg.max_streams * (size_t)e->cfg.max_channels * (size_t)AACG_OV_BUFFERS;
    const size_t take = n_cells < AACG_DUMP_CELLS ? n_cells : AACG_DUMP_CELLS;
    /* the stream is made at the first dump, not with the engine: one more stream at set-up shifts the runtime's stream -> hardware
     * queue assignment under the pipeline's streams (the comment at aacg_engine::pool); it is destroyed with the engine */
    if (!e->dump_stream && hipStreamCreateWithFlags(&e->dump_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); e->dump_stream = nullptr; }
    hipStream_t ds = e->dump_stream;
    aacg_xl_cell* h = e->h_dump;
    if (e->d_xl_cells && ds && h && hipMemcpyAsync(h, e->d_xl_cells, take * sizeof(aacg_xl_cell), hipMemcpyDeviceToHost, ds) == hipSuccess) {
        aacg_wait_policy quick; quick.limit_s = 0.25;
        if (aacg_wait_stream(ds, quick) == hipSuccess) {
            unsigned long long n_tail = 0, n_head = 0, lo = ~0ull, hi = 0;
            std::string heads;
            for (size_t i = 0; i < take; i++) {
                const unsigned long long w = h[i].state, ep = w >> 2;
                if (!w) continue;
                if ((w & 3) == AACG_RV_TAIL) n_tail++;
                if ((w & 3) == AACG_RV_HEAD) { n_head++; if (n_head <= 4) { std::snprintf(b, sizeof b, " cell %zu epoch %llu", i, ep); heads += b; } }
                lo = ep < lo ? ep : lo; hi = ep > hi ? ep : hi;
            }
            std::snprintf(b, sizeof b, "; cross-launch cells (first %zu of %zu): %llu TAIL, %llu HEAD (a HEAD nobody finished = a consumer whose producer never came:%s), epochs %llu..%llu",
                          take, n_cells, n_tail, n_head, heads.empty() ? " none" : heads.c_str(), lo == ~0ull ? 0ull : lo, hi);
            o += b;
        } else o += "; cross-launch cells unreadable: a device-to-host copy on a fresh stream did not complete in 250 ms either";
    } else { (void)hipGetLastError(); o += "; cross-launch cells not read"; }
    return o;
}

int timed_out(aacg_engine* e, const char* what)
{
    char b[160];
    std::snprintf(b, sizeof b, "%s: the GPU did not answer within %.1f s (aacg_set_wait_limit_ms). In flight: ", what, e->wait.limit_s);
    e->err = std::string(b) + in_flight(e);
    e->wedged = true;
    return AACG_ERR_TIMEOUT;
}
int wait_stream(aacg_engine* e, hipStream_t s, const char* what)
{
    const hipError_t st = aacg_wait_stream(s, e->wait);
    if (st == hipSuccess) return AACG_OK;
    if (st == hipErrorNotReady) return timed_out(e, what);
    e->err = std::string(what) + ": " + hipGetErrorString(st);
    return AACG_ERR_NO_DEVICE;
}
int wait_event(aacg_engine* e, hipEvent_t ev, const char* what)
{
    const hipError_t st = aacg_wait_event(ev, e->wait);
    if (st == hipSuccess) return AACG_OK;
    if (st == hipErrorNotReady) return timed_out(e, what);
    e->err = std::string(what) + ": " + hipGetErrorString(st);
    return AACG_ERR_NO_DEVICE;
}
/* a caller's stream the engine has put work on */
void note_stream(aacg_engine* e, hipStream_t s)
{
    if (!s || s == e->stream) return;
    for (hipStream_t k : e->pipe.stream) if (k == s) return;
    for (hipStream_t k : e->foreign) if (k == s) return;
    if (e->foreign.size() >= 32) e->foreign.erase(e->foreign.begin());
    e->foreign.push_back(s);
}

}  // namespace

/* ---- the route: ONE decision (aacg_pick_route, aacg_routes.cpp), executed by launch_run, printed by aacg_plan_kernels ---- */
/* the registered kernel with these switches; its symbol is what aacg_run_kernel_name composes (checked at aacg_create) */
const aacg_run_kernel* aacg_find_run_kernel(unsigned key)
{
    const aacg_run_kernel* const tabs[] = {aacg_run_kernels_plain, aacg_run_kernels_rv, aacg_run_kernels_nt, aacg_run_kernels_ext,
                                           aacg_run_kernels_i16, aacg_run_kernels_exrun, aacg_run_kernels_couple};
    const int counts[] = {aacg_run_kernels_plain_n, aacg_run_kernels_rv_n, aacg_run_kernels_nt_n, aacg_run_kernels_ext_n,
                          aacg_run_kernels_i16_n, aacg_run_kernels_exrun_n, aacg_run_kernels_couple_n};
    for (size_t t = 0; t < sizeof tabs / sizeof tabs[0]; t++)
        for (int i = 0; i < counts[t]; i++)
            if (tabs[t][i].key == key) return &tabs[t][i];
    return nullptr;
}

namespace {

aacg_route route_of(const aacg_engine* e, const aacg_plan_host& h, bool pipelined)
{
    return aacg_pick_route(e->cfg.input_kind, e->cfg.output_kind, e->debug_route, e->d_trace != nullptr, h, pipelined);
}

/* does the route stage f32 spectra in HBM for this batch? */
bool needs_spec_buffer(const aacg_engine* e, const aacg_plan_host& h) { return route_of(e, h, false).stage != AACG_STAGE_NONE; }

/* stop: an event bound to this dispatch's completion (hipExtLaunchKernel): it orders and times like an event recorded behind
 * the launch, without a marker packet in the queue between this launch and the next (tools/micro/stop_event.hip) */
int launch_kernel(aacg_engine* e, const aacg_run_kernel* k, unsigned blocks, hipStream_t s, const aacg_kparams& P, const aacg_rv_args* V, hipEvent_t stop = nullptr)
{
    if (!k) { e->err = "no run kernel for this route"; return AACG_ERR_UNSUPPORTED; }
    aacg_kparams p = P;
    aacg_rv_args v;
    void* args[8] = {&p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (k->key & AACG_RK_RV) { v = *V; args[1] = &v; }
    const void *a_runs = p.runs, *a_tab = p.tab, *a_links = V ? V->links : nullptr, *a_units = p.units, *a_coeffs = p.coeffs, *a_meta = p.meta;
    if (k->preloaded) { args[0] = &a_runs; args[1] = &a_tab; args[2] = &a_links; args[3] = &a_units; args[4] = &a_coeffs; args[5] = &a_meta; args[6] = &p; args[7] = &v; }   /* AACG_RUN_KERNEL_PRE (aacg_routes.h) */
    const dim3 block(AACG_WG_THREADS);
    if (stop) HIP_TRY(e, hipExtLaunchKernel(k->fn, dim3(blocks), block, args, 0, s, nullptr, stop, 0), AACG_ERR_NO_DEVICE);
    else      HIP_TRY(e, hipLaunchKernel(k->fn, dim3(blocks), block, args, 0, s), AACG_ERR_NO_DEVICE);
    return AACG_OK;
}

/* A plan's TNS records and, behind them in the same buffer, their transition matrices (tns_matrix_row, aacg_kernels.h: made on
 * the device by aacg_tns_matrices when the records are uploaded, read by every launch that runs filters). */
static size_t tns_record_bytes(size_t n) { return (sizeof(aacg_dev_tns) * n + 255u) & ~(size_t)255u; }
static size_t tns_buffer_bytes(size_t n) { return n ? tns_record_bytes(n) + sizeof(double) * AACG_TNS_M_DOUBLES * n : 0; }
static double* tns_matrices_of(const aacg_dev_tns* d_tns, size_t n) { return (double*)((char*)const_cast<aacg_dev_tns*>(d_tns) + tns_record_bytes(n)); }

/* what a pipelined launch adds to the rendezvous arguments: the cross-launch cells, the epoch its input state carries, and
 * which of the plan's two sets of in-launch cells it uses (overlapping launches must not share one) */
struct xl_args { bool on; unsigned long long epoch_in; int set; int trace_part = 0; };   /* trace_part: which quarter of the profiling buffer this launch stamps */

/* enqueue the launches of route R for a planned batch (device pointers) */
int launch_run(aacg_engine* e, const aacg_route& R, const aacg_dev_unit* d_units, const aacg_run* d_runs, const aacg_dev_tns* d_tns,
               float* d_scratch, float* d_spec, const cce_bufs& cb, const rv_bufs& rvb, const aacg_plan_host& h, const void* d_coeffs, const aacg_band_meta* d_meta,
               void* d_pcm, int flip, hipStream_t s, const xl_args& xl, unsigned long long* epoch_out, hipEvent_t stop = nullptr)
{
    if (h.zero_fill)
        HIP_TRY(e, hipMemsetAsync(d_pcm, 0, h.pcm_floats * pcm_elem_size(e), s), AACG_ERR_NO_DEVICE);
    aacg_kparams P;
    std::memset(&P, 0, sizeof P);
    P.units = d_units; P.coeffs = d_coeffs; P.meta = d_meta; P.pcm = (float*)d_pcm;
    P.overlap = e->d_overlap; P.tab = e->d_tab; P.flip = flip;
    if (R.rv) {
        const size_t n_runs = h.runs_rv.size(), n_links = h.n_links_rv;
        P.runs = rvb.runs; P.n_runs = (int32_t)n_runs;
        aacg_rv_args V;
        std::memset(&V, 0, sizeof V);
        V.links = rvb.links;
        V.state = rvb.state ? rvb.state + (size_t)xl.set * AACG_RV_STATE_WORDS * n_links : nullptr;
        V.data = rvb.data ? rvb.data + (size_t)xl.set * AACG_RV_DATA_FLOATS * n_links : nullptr;
        V.epoch = ++e->rv_epoch;
        if (xl.on) { V.xl_cells = e->d_xl_cells; V.xl_head = e->d_xl_head; V.epoch_in = xl.epoch_in; }
        if (R.run_key & AACG_RK_EX) {                    /* optional stages inside the run kernel */
            P.tns = h.any_tns ? d_tns : nullptr; P.pns = e->d_pns;
            if (h.any_tns) aacg_set_tns_m(&P, tns_matrices_of(d_tns, h.tns.size()));
        }
        if (epoch_out) *epoch_out = V.epoch;
        P.ablate = e->d_trace ? e->ablate : (e->ablate & ~16);                                       /* profiling builds only (0 in the library that ships) */
        if (e->d_trace) P.spec_out = (float*)e->d_trace + (size_t)xl.trace_part * (1u << 18);       /* ... the last four launches keep their stamps */
        return launch_kernel(e, aacg_find_run_kernel(R.run_key), (unsigned)n_runs, s, P, &V, stop);
    }
    P.runs = d_runs; P.n_runs = (int32_t)h.runs.size();
    P.tns = h.any_tns ? d_tns : nullptr;
    P.scratch = h.needs_scratch ? d_scratch : nullptr;
    P.ablate = e->d_trace ? e->ablate : (e->ablate & ~16);
    if (e->d_trace) P.spec_out = (float*)e->d_trace;
    auto couple = [&](int point) {                      /* the coupling launches of one coupling point, round by round */
        for (uint32_t r = 0; r < h.couple_rounds; r++) {
            const uint32_t first = h.couple_first[(size_t)point * h.couple_rounds + r], last = h.couple_first[(size_t)point * h.couple_rounds + r + 1];
            if (last <= first) continue;
            aacg_couple_params Q;
            Q.jobs = cb.jobs + first; Q.n_jobs = (int32_t)(last - first); Q.units = d_units; Q.meta = d_meta; Q.tab = e->d_tab;
            Q.gains = cb.gains; Q.spec = d_spec; Q.side = cb.side; Q.pcm = (float*)d_pcm; Q.reserved = 0;
            aacg_couple_launch(point == AACG_CCE_AFTER_IMDCT, s, Q);
        }
    };
    if (R.stage == AACG_STAGE_DEPENDENT_COUPLING) {
        /* AACG_CCE_SPEC with coupling in the spectral domain: every unit's spectrum (the coupling elements' too) as f32,
         * then decoder.js:258-266 / 304-316 in stages: coupling before TNS, the TNS filters, coupling after TNS — each its
         * own small launch, in place.  (Independent coupling alone leaves the spectral route as it is.) */
        float* trace_or_null = P.spec_out;
        P.spec_out = d_spec; P.pns = e->d_pns; P.tns = nullptr;
        if (R.stage_quant) aacg_spectral_ex_launch(true, (int)h.units.size(), s, P);
        else HIP_TRY(e, hipMemcpyAsync(d_spec, d_coeffs, (size_t)h.coef_blocks * 4096u, hipMemcpyDeviceToDevice, s), AACG_ERR_NO_DEVICE);
        couple(AACG_CCE_BEFORE_TNS);
        if (h.any_tns) {
            P.coeffs = d_spec; P.meta = nullptr; P.tns = d_tns;
            { aacg_kparams Q = P; aacg_set_tns_m(&Q, tns_matrices_of(d_tns, h.tns.size())); aacg_spectral_ex_launch(false, (int)h.units.size(), s, Q); }
        }
        couple(AACG_CCE_AFTER_TNS);
        P.spec_out = trace_or_null; P.coeffs = d_spec; P.meta = nullptr; P.tns = nullptr;
    } else if (R.stage == AACG_STAGE_SPECTRAL_EX) {
        /* the optional stages first, as a launch of their own that leaves f32 spectra, which the f32 run kernel takes from there */
        float* trace_or_null = P.spec_out;
        P.spec_out = d_spec; P.pns = e->d_pns;
        { aacg_kparams Q = P; if (h.any_tns) aacg_set_tns_m(&Q, tns_matrices_of(d_tns, h.tns.size())); aacg_spectral_ex_launch(R.stage_quant, (int)h.units.size(), s, Q); }
        P.spec_out = trace_or_null; P.coeffs = d_spec; P.meta = nullptr; P.tns = nullptr;
    } else if (R.has_run && (R.run_key & AACG_RK_EX)) {
        P.pns = e->d_pns;                                /* optional stages (noise bands, TNS filters) inside the run kernel: one launch */
        if (h.any_tns) aacg_set_tns_m(&P, tns_matrices_of(d_tns, h.tns.size()));
    }
    int rc = AACG_OK;
    auto side_pass = [&]() {                            /* the independently switched coupling elements' own filterbank pass */
        aacg_kparams C = P;
        C.runs = cb.runs; C.n_runs = (int32_t)h.cce_runs.size(); C.pcm = cb.side; C.scratch = nullptr;
        return launch_kernel(e, aacg_find_run_kernel(R.side_key), (unsigned)h.cce_runs.size(), s, C, nullptr);
    };
    if (R.has_side && R.side_first && (rc = side_pass())) return rc;
    if (R.has_run) {
        /* independent coupling in the targets' epilogues: the run kernel adds gain * side where it forms the PCM — no
         * read-modify-write pass over the interleaved PCM */
        if (R.run_key & AACG_RK_CPL) aacg_set_cpl(&P, cb.jobs + h.fused_first, cb.gains, cb.side);
        if ((rc = launch_kernel(e, aacg_find_run_kernel(R.run_key), (unsigned)h.runs.size(), s, P, nullptr))) return rc;
    }
    if (R.has_side && !R.side_first && (rc = side_pass())) return rc;
    if (R.couple_pcm) couple(AACG_CCE_AFTER_IMDCT);
    HIP_TRY(e, hipGetLastError(), AACG_ERR_NO_DEVICE);
    return AACG_OK;
}

/* Everything the pipeline has in flight in front of work on stream s (device-side wait; s null: the host waits). */
int pipe_join(aacg_engine* e, hipStream_t s)
{
    aacg_engine::pipe_t& pp = e->pipe;
    if (!pp.open) return AACG_OK;
    if (s && s == pp.joined_stream && pp.issued == pp.joined_n) return AACG_OK;   /* s is behind all of it already */
    for (int k = 0; k < AACG_PIPE_STREAMS; k++) {
        if (s == pp.stream[k]) continue;                /* its own launches are in front of it anyway */
        if (s) {
            HIP_TRY(e, hipEventRecord(pp.tail[k], pp.stream[k]), AACG_ERR_NO_DEVICE);
            HIP_TRY(e, hipStreamWaitEvent(s, pp.tail[k], 0), AACG_ERR_NO_DEVICE);
        } else { const int wrc = wait_stream(e, pp.stream[k], "waiting for the pipelined launches"); if (wrc) return wrc; }
    }
    if (!s) pp.open = false;                           /* the host has seen the streams drained: nothing in flight any more */
    pp.joined_stream = s; pp.joined_n = pp.issued;
    return AACG_OK;
}

/* Everything this engine has enqueued anywhere is complete: its own stream, the pipeline's, the host-buffer path's two, and
 * every caller's stream it was asked to launch on (what hipDeviceSynchronize stood for until round 5 — without waiting for
 * streams the engine never touched, and bounded). */
int quiesce(aacg_engine* e, const char* what)
{
    int rc = wait_stream(e, e->stream, what);
    if (rc) return rc;
    if (e->pipe.stream[0]) {
        for (hipStream_t ps : e->pipe.stream) if (ps && (rc = wait_stream(e, ps, what))) return rc;
        e->pipe.open = false;
        e->pipe.joined_stream = nullptr; e->pipe.joined_n = e->pipe.issued;
    }
    for (auto& sl : e->slot) if (sl.stream && (rc = wait_stream(e, sl.stream, what))) return rc;
    for (size_t k = 0; k < e->foreign.size();) {
        const hipError_t st = aacg_wait_stream(e->foreign[k], e->wait);
        if (st == hipErrorNotReady) return timed_out(e, what);
        if (st != hipSuccess) { (void)hipGetLastError(); e->foreign.erase(e->foreign.begin() + (long)k); continue; }   /* the caller has destroyed it: nothing left on it */
        k++;
    }
    e->wedged = false;                                  /* everything has answered */
    return AACG_OK;
}

}  // namespace

extern "C" {

int aacg_abi_version(void) { return AACG_ABI_VERSION; }

/* the dominant kernel of the headline route: BASELINE config 2 through aacg_decode_pipelined */
const char* aacg_kernel_name(void) { return "aacg_imdct_run_quant_rv"; }

int aacg_debug_set_route(aacg_engine* e, int flags)
{
    if (!e || (flags & ~(AACG_DEBUG_ROUTE_UNFUSED_COUPLING | AACG_DEBUG_ROUTE_RECOMPUTE))) return AACG_ERR_INVALID_ARG;
    e->debug_route = flags;
    return AACG_OK;
}

int aacg_plan_kernels(aacg_engine* e, const aacg_plan* p, char* dst, size_t n)
{
    return aacg_plan_kernels_ex(e, p, 0, dst, n);
}

int aacg_plan_kernels_ex(aacg_engine* e, const aacg_plan* p, int pipelined, char* dst, size_t n)
{
    if (!e || !p || p->e != e || !dst || n == 0) return AACG_ERR_INVALID_ARG;
    const std::string r = aacg_route_names(route_of(e, p->h, pipelined != 0), p->h.any_tns);
    if (r.size() + 1 > n) return AACG_ERR_INVALID_ARG;
    std::memcpy(dst, r.c_str(), r.size() + 1);
    return AACG_OK;
}

/* The route decision on its own, for tests without a device: flags of a hypothetical engine and planned batch in, the
 * launches by name out.  plan_flags: AACG_ROUTE_PLAN_* (aacgpu_tools.h). */
int aacg_debug_route(int input_kind, int output_kind, int debug_route, int plan_flags, int pipelined, char* dst, size_t n)
{
    if (!dst || n == 0) return AACG_ERR_INVALID_ARG;
    aacg_plan_host h;
    h.any_tns = (plan_flags & AACG_ROUTE_PLAN_TNS) != 0;
    h.any_pns = (plan_flags & AACG_ROUTE_PLAN_PNS) != 0;
    h.needs_scratch = (plan_flags & AACG_ROUTE_PLAN_FULL_LATER_RUNS) != 0;
    h.long_chains = (plan_flags & (AACG_ROUTE_PLAN_LONG_CHAINS | AACG_ROUTE_PLAN_FULL_LATER_RUNS)) != 0;
    h.wide_frames = (plan_flags & AACG_ROUTE_PLAN_WIDE_FRAMES) != 0;
    h.any_cce = (plan_flags & (AACG_ROUTE_PLAN_CCE_INDEPENDENT | AACG_ROUTE_PLAN_CCE_DEPENDENT)) != 0;
    h.any_cce_dependent = (plan_flags & AACG_ROUTE_PLAN_CCE_DEPENDENT) != 0;
    /* independent coupling is fused into the targets' epilogues unless the plan has double-duty runs (aacg_plan_build) */
    h.fused_independent = (plan_flags & AACG_ROUTE_PLAN_CCE_INDEPENDENT) && !h.needs_scratch;
    if (plan_flags & AACG_ROUTE_PLAN_CCE_INDEPENDENT) h.cce_runs.resize(1);
    if (!(plan_flags & AACG_ROUTE_PLAN_NO_RUNS)) { h.runs.resize(1); h.runs_rv.resize(1); }
    const aacg_route r = aacg_pick_route(input_kind, output_kind, debug_route, false, h, pipelined != 0);
    if ((r.has_run && !aacg_find_run_kernel(r.run_key)) || (r.has_side && !aacg_find_run_kernel(r.side_key))) return AACG_ERR_UNSUPPORTED;   /* a route without a kernel */
    const std::string t = aacg_route_names(r, h.any_tns);
    if (t.size() + 1 > n) return AACG_ERR_INVALID_ARG;
    std::memcpy(dst, t.c_str(), t.size() + 1);
    return AACG_OK;
}

int aacg_debug_pipeline_order(unsigned long long n, int streams, int* stream, long long* sync_round, int* marked, long long* complete_upto)
{
    const aacg_pipe_order o = aacg_pipeline_order(n, streams);
    if (stream) *stream = o.stream;
    if (sync_round) *sync_round = o.sync_round;
    if (marked) *marked = o.marked ? 1 : 0;
    if (complete_upto) *complete_upto = o.complete_upto;
    return AACG_OV_BUFFERS;
}

/* The registered run kernels: `index`-th symbol into dst; returns its switches (AACG_RK_*), or < 0 past the end.  Every
 * symbol must be what aacg_run_kernel_name composes from its switches (tests/test_routes.py; aacg_create checks it too). */
int aacg_debug_run_kernel(int index, char* dst, size_t n)
{
    const aacg_run_kernel* const tabs[] = {aacg_run_kernels_plain, aacg_run_kernels_rv, aacg_run_kernels_nt, aacg_run_kernels_ext,
                                           aacg_run_kernels_i16, aacg_run_kernels_exrun, aacg_run_kernels_couple};
    const int counts[] = {aacg_run_kernels_plain_n, aacg_run_kernels_rv_n, aacg_run_kernels_nt_n, aacg_run_kernels_ext_n,
                          aacg_run_kernels_i16_n, aacg_run_kernels_exrun_n, aacg_run_kernels_couple_n};
    for (size_t t = 0; t < sizeof tabs / sizeof tabs[0]; t++) {
        if (index < counts[t]) {
            if (!dst || std::strlen(tabs[t][index].name) + 1 > n) return AACG_ERR_INVALID_ARG;
            std::strcpy(dst, tabs[t][index].name);
            return (int)tabs[t][index].key;
        }
        index -= counts[t];
    }
    return AACG_ERR_INVALID_ARG;
}

const char* aacg_last_error(const aacg_engine* e) { return e ? e->err.c_str() : "null engine"; }

int aacg_create(const aacg_config* cfg, aacg_engine** out)
{
    if (!cfg || !out) return AACG_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->abi_version != AACG_ABI_VERSION || cfg->max_streams < 1 || cfg->max_channels < 1 ||
        cfg->max_channels > AACG_MAX_CHANNELS || (cfg->tns_mode != AACG_TNS_REFERENCE && cfg->tns_mode != AACG_TNS_SPEC) ||
        (cfg->pns_mode != AACG_PNS_REFERENCE && cfg->pns_mode != AACG_PNS_SPEC) ||
        (cfg->pns_mode == AACG_PNS_SPEC && cfg->input_kind != AACG_INPUT_QUANT_I16) ||
        (cfg->output_kind != AACG_OUTPUT_F32 && cfg->output_kind != AACG_OUTPUT_I16) ||
        (cfg->cce_mode != AACG_CCE_REFERENCE && cfg->cce_mode != AACG_CCE_SPEC) ||
        (cfg->cce_mode == AACG_CCE_SPEC && cfg->output_kind != AACG_OUTPUT_F32) ||
        (cfg->input_kind != AACG_INPUT_SPEC_F32 && cfg->input_kind != AACG_INPUT_QUANT_I16))
        return AACG_ERR_INVALID_ARG;

    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || cfg->device_ordinal < 0 || cfg->device_ordinal >= n_dev)
        return AACG_ERR_NO_DEVICE;

    aacg_engine* e = new (std::nothrow) aacg_engine();
    if (!e) return AACG_ERR_OUT_OF_MEMORY;
    e->cfg = *cfg;
    int rc = aacg_build_tables(cfg->sample_index, &e->h_tab, &e->h_win);
    if (rc) { delete e; return rc; }

    const size_t n_cells = (size_t)cfg->max_streams * (size_t)cfg->max_channels * (size_t)AACG_OV_BUFFERS;
    const size_t ov_bytes = n_cells * 1024u * sizeof(float);
    if (!hip_ok(e, hipSetDevice(cfg->device_ordinal), "hipSetDevice") ||
        !hip_ok(e, hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking), "hipStreamCreate") ||
        !hip_ok(e, hipMalloc((void**)&e->d_tab, sizeof(aacg_tables)), "hipMalloc tables") ||
        !hip_ok(e, hipMalloc((void**)&e->d_overlap, ov_bytes), "hipMalloc overlap") ||
        !hip_ok(e, hipMemcpy(e->d_tab, &e->h_tab, sizeof(aacg_tables), hipMemcpyHostToDevice), "upload tables") ||
        !hip_ok(e, hipMemset(e->d_overlap, 0, ov_bytes), "zero overlap") ||
        /* pipelined launches: the cross-launch cells (state words count only with an epoch in them: they start from zero) and the pool of windowed first halves */
        !hip_ok(e, hipMalloc((void**)&e->d_xl_cells, n_cells * sizeof(aacg_xl_cell)), "hipMalloc cells") ||
        !hip_ok(e, hipMemset(e->d_xl_cells, 0, n_cells * sizeof(aacg_xl_cell)), "zero cells") ||
        !hip_ok(e, hipMalloc((void**)&e->d_xl_head, ov_bytes), "hipMalloc heads") ||
        /* (the run kernels' ~152 KiB of LDS per workgroup are static allocations: dp_lds_fixed) */
        aacg_spectral_ex_set_lds_limits() != 0 ||
        !hip_ok(e, hipFuncSetAttribute((const void*)aacg_spectral, hipFuncAttributeMaxDynamicSharedMemorySize, AACG_LDS_BYTES_SPECTRAL), "LDS attr")) {
        std::fprintf(stderr, "aacgpu: %s\n", e->err.c_str());
        aacg_destroy(e);
        return AACG_ERR_NO_DEVICE;
    }
    e->parity.assign((size_t)cfg->max_streams * (size_t)cfg->max_channels, 0);
    /* every registered run kernel carries the symbol its switches compose: what aacg_plan_kernels prints is what launches */
    for (int i = 0;; i++) {
        char name[96];
        const int key = aacg_debug_run_kernel(i, name, sizeof name);
        if (key < 0) break;
        if (aacg_run_kernel_name((unsigned)key) != name) { std::fprintf(stderr, "aacgpu: run kernel table: %s registered with switches %d\n", name, key); aacg_destroy(e); return AACG_ERR_INVALID_ARG; }
    }
    if (cfg->pns_mode == AACG_PNS_SPEC) {
        aacg_pns_tables* pt = new (std::nothrow) aacg_pns_tables;
        const bool ok = pt && aacg_build_pns_tables(cfg->sample_index, pt) == AACG_OK &&
                        hip_ok(e, hipMalloc((void**)&e->d_pns, sizeof *pt), "hipMalloc pns tables") &&
                        hip_ok(e, hipMemcpy(e->d_pns, pt, sizeof *pt, hipMemcpyHostToDevice), "upload pns tables");
        delete pt;
        if (!ok) { aacg_destroy(e); return AACG_ERR_OUT_OF_MEMORY; }
    }
#ifdef AACG_PROFILE                              /* profiling builds only (make profile); the shipped library has no such switch */
    if (const char* a = std::getenv("AACG_ABLATE")) e->ablate = std::atoi(a);
    if ((e->ablate & 16) && hipMalloc(&e->d_trace, 1u << 22) == hipSuccess) (void)hipMemset(e->d_trace, 0, 1u << 22);
#endif
    /* the uploads and memsets above went to the null stream; the engine's streams are non-blocking and would not wait for it */
    if (!hip_ok(e, hipHostMalloc((void**)&e->h_ov, 4096, hipHostMallocDefault), "hipHostMalloc (overlap staging)") ||
        !hip_ok(e, hipHostMalloc((void**)&e->h_dump, AACG_DUMP_CELLS * sizeof(aacg_xl_cell), hipHostMallocDefault), "hipHostMalloc (dump staging)")) { aacg_destroy(e); return AACG_ERR_OUT_OF_MEMORY; }
    { const int wrc = wait_stream(e, nullptr, "aacg_create: waiting for the table uploads"); if (wrc) { std::fprintf(stderr, "aacgpu: %s\n", e->err.c_str()); aacg_destroy(e); return wrc; } }
    *out = e;
    return AACG_OK;
}

int aacg_set_wait_limit_ms(aacg_engine* e, uint32_t ms)
{
    if (!e || !ms) return AACG_ERR_INVALID_ARG;
    e->wait.limit_s = ms * 1e-3;
    return AACG_OK;
}
int aacg_debug_in_flight(aacg_engine* e, char* dst, size_t n)
{
    if (!e || !dst || !n) return AACG_ERR_INVALID_ARG;
    (void)hipSetDevice(e->cfg.device_ordinal);
    const std::string t = in_flight(e);
    std::snprintf(dst, n, "%s", t.c_str());
    return (int)t.size();
}
static int pipe_setup(aacg_engine* e);
/* tests only (aacgpu_tools.h): keeps one of the engine's streams busy for `ms` milliseconds (a single wave polling a word nobody
 * sets, bounded by the device clock), so that the bounded waits and their dump can be exercised without a broken device */
int aacg_debug_stall(aacg_engine* e, int which, uint32_t ms)
{
    if (!e || which < 0 || which > AACG_PIPE_STREAMS || ms > 5000) return AACG_ERR_INVALID_ARG;
    HIP_TRY(e, hipSetDevice(e->cfg.device_ordinal), AACG_ERR_NO_DEVICE);
    if (which) { const int rc = pipe_setup(e); if (rc) return rc; }
    hipStream_t s = which ? e->pipe.stream[which - 1] : e->stream;
    /* the cross-launch cells' pool is zeroed memory nobody writes a 1 into at word 2 of a cell (the PCM address of a HEAD is 8-byte aligned and never 1) */
    static unsigned* d_flag = nullptr;
    if (!d_flag) { HIP_TRY(e, hipMalloc((void**)&d_flag, 8), AACG_ERR_OUT_OF_MEMORY); HIP_TRY(e, hipMemset(d_flag, 0, 8), AACG_ERR_NO_DEVICE); }
    hipLaunchKernelGGL(aacg_probe_wait, dim3(1), dim3(1), 0, s, d_flag, d_flag + 1, (long long)ms * 100000LL);      /* 100 MHz clock */
    HIP_TRY(e, hipGetLastError(), AACG_ERR_NO_DEVICE);
    if (which) { e->pipe.open = true; e->pipe.issued++; }     /* something of the pipeline is in flight */
    return AACG_OK;
}
/* measurement only (aacgpu_tools.h): how the host waits once a wait has lasted spin_us */
int aacg_debug_set_wait_mode(aacg_engine* e, int mode, double spin_us)
{
    if (!e || mode < AACG_WAIT_SPIN || mode > AACG_WAIT_BLOCK || e->pipe.stream[0]) return AACG_ERR_INVALID_ARG;   /* before the first pipelined launch: the events' flags depend on it */
    e->wait.mode = mode;
    if (spin_us >= 0) e->wait.spin_us = spin_us;
    return AACG_OK;
}

void aacg_destroy(aacg_engine* e)
{
    if (!e) return;
    (void)hipSetDevice(e->cfg.device_ordinal);
    /* a device that has stopped answering is not waited for again, and nothing of it is freed (hipFree waits for the device):
     * the memory stays with the process, the caller gets its thread back */
    if (e->wedged) { e->wait.limit_s = e->wait.limit_s < 0.25 ? e->wait.limit_s : 0.25; e->wedged = false; }   /* it did not answer once: one short look whether it does now */
    if (quiesce(e, "aacg_destroy") != AACG_OK) {
        std::fprintf(stderr, "aacgpu: aacg_destroy: %s — device memory of this engine is left allocated\n", e->err.c_str());
        delete e;
        return;
    }
    if (e->h_ov) (void)hipHostFree(e->h_ov);
    if (e->h_dump) (void)hipHostFree(e->h_dump);
    if (e->dump_stream) (void)hipStreamDestroy(e->dump_stream);
    if (e->d_tab) (void)hipFree(e->d_tab);
    if (e->d_pns) (void)hipFree(e->d_pns);
    if (e->d_xl_cells) (void)hipFree(e->d_xl_cells);
    if (e->d_xl_head) (void)hipFree(e->d_xl_head);
    for (auto& round : e->pipe.mark) for (hipEvent_t ev : round) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->pipe.tail) if (ev) (void)hipEventDestroy(ev);
    if (e->pipe.fork) (void)hipEventDestroy(e->pipe.fork);
    for (hipStream_t st : e->pipe.stream) if (st) (void)hipStreamDestroy(st);
    if (e->d_overlap) (void)hipFree(e->d_overlap);
    for (auto& sl : e->slot) {
        for (void* p : {sl.d_units, sl.d_runs, sl.d_coeffs, sl.d_meta, sl.d_tns, sl.d_scratch, sl.d_spec, sl.d_pcm}) if (p) (void)hipFree(p);
        for (void* p : sl.d_cce) if (p) (void)hipFree(p);
        for (void* p : sl.d_rv) if (p) (void)hipFree(p);
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_pcm) (void)hipHostFree(sl.h_pcm);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.kernel_done) (void)hipEventDestroy(sl.kernel_done);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
    }
    if (e->d_trace) (void)hipFree(e->d_trace);
    for (auto& b : e->pool) (void)hipFree(b.first);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

/* ---- overlap state ----------------------------------------------------------------- */
static int ov_check(aacg_engine* e, uint32_t stream, uint32_t channel)
{
    if (!e) return AACG_ERR_INVALID_ARG;
    if ((int)stream >= e->cfg.max_streams || (int)channel >= e->cfg.max_channels) {
        e->err = "stream/channel out of range";
        return AACG_ERR_INVALID_ARG;
    }
    HIP_TRY(e, hipSetDevice(e->cfg.device_ordinal), AACG_ERR_NO_DEVICE);
    const int rc = quiesce(e, "overlap state: waiting for the launches that advance it");
    if (rc) return rc;
    e->pipe.open = false;                               /* nothing in flight any more: the next pipelined launch starts from complete state */
    return AACG_OK;
}

static float* ov_ptr(aacg_engine* e, uint32_t stream, uint32_t channel)
{
    const int p = e->parity[(size_t)stream * (size_t)e->cfg.max_channels + channel];
    return e->d_overlap + aacg_ov_offset(e->cfg.max_channels, stream, channel, p);
}

int aacg_reset_stream(aacg_engine* e, uint32_t stream)
{
    int rc = ov_check(e, stream, 0);
    if (rc) return rc;
    for (int c = 0; c < e->cfg.max_channels; c++)
        HIP_TRY(e, hipMemsetAsync(ov_ptr(e, stream, (uint32_t)c), 0, 4096, e->stream), AACG_ERR_NO_DEVICE);
    return wait_stream(e, e->stream, "aacg_reset_stream");      /* complete before any later launch, whatever stream it takes */
}

int aacg_get_overlap(aacg_engine* e, uint32_t stream, uint32_t channel, float* dst)
{
    int rc = ov_check(e, stream, channel);
    if (rc) return rc;
    if (!dst) return AACG_ERR_INVALID_ARG;
    HIP_TRY(e, hipMemcpyAsync(e->h_ov, ov_ptr(e, stream, channel), 4096, hipMemcpyDeviceToHost, e->stream), AACG_ERR_NO_DEVICE);
    if ((rc = wait_stream(e, e->stream, "aacg_get_overlap"))) return rc;
    std::memcpy(dst, e->h_ov, 4096);
    /* the pool holds the state PCM-scaled (AACG_PCM_SCALE in the windows); the ABI speaks the reference's scale
     * (FilterBank.overlaps, filter_bank.js:38-41): a power of two, exact both ways */
    for (int i = 0; i < 1024; i++) dst[i] *= 32768.0f;
    return AACG_OK;
}

int aacg_set_overlap(aacg_engine* e, uint32_t stream, uint32_t channel, const float* src)
{
    int rc = ov_check(e, stream, channel);
    if (rc) return rc;
    if (!src) return AACG_ERR_INVALID_ARG;
    for (int i = 0; i < 1024; i++) e->h_ov[i] = src[i] * AACG_PCM_SCALE;
    HIP_TRY(e, hipMemcpyAsync(ov_ptr(e, stream, channel), e->h_ov, 4096, hipMemcpyHostToDevice, e->stream), AACG_ERR_NO_DEVICE);
    return wait_stream(e, e->stream, "aacg_set_overlap");
}

int aacg_get_table(aacg_engine* e, int which, float* dst, size_t n)
{
    if (!e || !dst) return AACG_ERR_INVALID_ARG;
    const float* src; size_t cnt;
    switch (which) {
    case 0: src = e->h_tab.iq;           cnt = 8191; break;
    case 1: src = e->h_tab.sf;           cnt = 428;  break;
    case 2: src = e->h_win.sine_long;    cnt = 1024; break;
    case 3: src = e->h_win.kbd_long;     cnt = 1024; break;
    case 4: src = e->h_win.sine_short;   cnt = 128;  break;
    case 5: src = e->h_win.kbd_short;    cnt = 128;  break;
    case 100:                                  /* profiling: raw phase-timestamp buffer of the last launch */
        if (!e->d_trace) return AACG_ERR_INVALID_ARG;
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(dst, e->d_trace, sizeof(float) * (n < (1u << 20) ? n : (1u << 20)), hipMemcpyDeviceToHost);
        return (int)(1u << 20);
    default: return AACG_ERR_INVALID_ARG;
    }
    std::memcpy(dst, src, sizeof(float) * (n < cnt ? n : cnt));
    return (int)cnt;
}

/* ---- plans ------------------------------------------------------------------------- */
int aacg_plan_create(aacg_engine* e, const aacg_unit_desc* units, uint32_t n_units, aacg_plan** out)
{
    return aacg_plan_create_tns(e, units, n_units, nullptr, 0, out);
}

int aacg_plan_create_tns(aacg_engine* e, const aacg_unit_desc* units, uint32_t n_units,
                         const aacg_tns_info* tns, uint32_t n_tns, aacg_plan** out)
{
    return aacg_plan_create_ex(e, units, n_units, tns, n_tns, nullptr, 0, out);
}

int aacg_plan_create_ex(aacg_engine* e, const aacg_unit_desc* units, uint32_t n_units,
                        const aacg_tns_info* tns, uint32_t n_tns, const aacg_cce_info* cce, uint32_t n_cce, aacg_plan** out)
{
    if (!e || !units || !n_units || !out) return AACG_ERR_INVALID_ARG;
    if (e->cfg.tns_mode != AACG_TNS_SPEC) { tns = nullptr; n_tns = 0; }   /* REFERENCE mode: TNS is the identity */
    if (e->cfg.cce_mode != AACG_CCE_SPEC) { cce = nullptr; n_cce = 0; }   /* REFERENCE mode: a coupling element in the batch is refused */
    *out = nullptr;
    aacg_plan* p = new (std::nothrow) aacg_plan();
    if (!p) return AACG_ERR_OUT_OF_MEMORY;
    p->e = e;
    p->n_units = n_units;
    int rc = aacg_plan_build(units, n_units, e->cfg.sample_index, e->cfg.max_streams, e->cfg.max_channels,
                             e->parity.data(), &p->h, &e->err, tns, n_tns, cce, n_cce);
    if (rc) { delete p; return rc; }
    if (p->h.any_pns && e->cfg.pns_mode != AACG_PNS_SPEC) {
        e->err = "a unit carries AACG_UNIT_HAS_PNS: NOISE_BT bands are not decodable by the reference either (AACG_PNS_SPEC engines fill them)";
        delete p;
        return AACG_ERR_UNSUPPORTED;
    }
    const size_t ub = sizeof(aacg_dev_unit) * n_units, rb = sizeof(aacg_run) * p->h.runs.size();
    const size_t xb = needs_spec_buffer(e, p->h) ? (size_t)p->h.coef_blocks * 1024u * sizeof(float) : 0;
    const size_t tb = tns_buffer_bytes(p->h.tns.size());
    const size_t sb = p->h.needs_scratch ? p->h.runs.size() * AACG_SLOT_FLOATS * sizeof(float) : 0;
    const size_t cb[4] = {sizeof(aacg_run) * p->h.cce_runs.size(), sizeof(aacg_couple_job) * p->h.couple_jobs.size(),
                          sizeof(float) * p->h.gains.size(), (size_t)p->h.side_blocks * 4096u};
    bool ok = hip_ok(e, hipSetDevice(e->cfg.device_ordinal), "hipSetDevice") &&
              hip_ok(e, hipEventCreateWithFlags(&p->uploaded, hipEventDisableTiming), "hipEventCreate") &&
              hip_ok(e, hipEventCreateWithFlags(&p->last_use, hipEventDisableTiming), "hipEventCreate");
    /* the rendezvous cut of the chains, for plans that can take it (serially: long chains; through the pipeline: every plain
     * batch): run table, links, and a set of in-launch cells per pipeline stream — overlapping launches of the plan must not share one */
    const bool rvp = route_of(e, p->h, true).rv;
    const size_t rvs[4] = {rvp ? sizeof(aacg_run) * p->h.runs_rv.size() : 0, rvp ? sizeof(aacg_rv_link) * p->h.links_rv.size() : 0,
                           rvp ? (size_t)AACG_PIPE_STREAMS * sizeof(unsigned long long) * AACG_RV_STATE_WORDS * (size_t)p->h.n_links_rv : 0,
                           rvp ? (size_t)AACG_PIPE_STREAMS * sizeof(float) * AACG_RV_DATA_FLOATS * (size_t)p->h.n_links_rv : 0};
    const size_t want[AACG_PLAN_BUFFERS] = {ub, rb, tb, sb, xb, cb[0], cb[1], cb[2], cb[3], rvs[0], rvs[1], rvs[2], rvs[3]};
    void** const slot[AACG_PLAN_BUFFERS] = {(void**)&p->d_units, (void**)&p->d_runs, (void**)&p->d_tns, (void**)&p->d_scratch, (void**)&p->d_spec,
                             &p->d_cce[0], &p->d_cce[1], &p->d_cce[2], &p->d_cce[3], &p->d_rv[0], &p->d_rv[1], &p->d_rv[2], &p->d_rv[3]};
    const void* const src[AACG_PLAN_BUFFERS] = {p->h.units.data(), p->h.runs.data(), p->h.tns.data(), nullptr, nullptr,
                                 p->h.cce_runs.data(), p->h.couple_jobs.data(), p->h.gains.data(), nullptr,
                                 p->h.runs_rv.data(), p->h.links_rv.data(), nullptr, nullptr};
    for (int i = 0; i < AACG_PLAN_BUFFERS && ok; i++) {
        if (!want[i]) continue;
        *slot[i] = pool_take(e, want[i], &p->bytes[i]);
        ok = *slot[i] != nullptr &&
             (!src[i] || hip_ok(e, hipMemcpyAsync(*slot[i], src[i], i == 2 ? sizeof(aacg_dev_tns) * p->h.tns.size() : want[i], hipMemcpyHostToDevice, e->stream), "upload plan tables"));
        if (ok && i == 2) aacg_tns_matrices_launch(p->d_tns, tns_matrices_of(p->d_tns, p->h.tns.size()), (uint32_t)p->h.tns.size(), e->stream);   /* behind the records, once */
        /* rendezvous state words count only with a launch's epoch in them; a recycled or fresh buffer starts from zero all the same */
        if (ok && i == 11) ok = hip_ok(e, hipMemsetAsync(*slot[i], 0, want[i], e->stream), "zero rendezvous state");
    }
    ok = ok && hip_ok(e, hipEventRecord(p->uploaded, e->stream), "hipEventRecord");
    if (!ok) {
        aacg_plan_destroy(p);
        return AACG_ERR_OUT_OF_MEMORY;
    }
    *out = p;
    return AACG_OK;
}

void aacg_plan_destroy(aacg_plan* p)
{
    if (!p) return;
    aacg_engine* e = p->e;
    (void)hipSetDevice(e->cfg.device_ordinal);
    /* the buffers go back to the free list: wait for the copies into them and for the last kernel that reads them
     * (two events, not the whole device) */
    bool idle = !e->wedged;
    if (p->uploaded) { idle = idle && wait_event(e, p->uploaded, "aacg_plan_destroy: the plan's uploads") == AACG_OK; (void)hipEventDestroy(p->uploaded); }
    if (p->last_pipelined) idle = idle && pipe_join(e, nullptr) == AACG_OK;       /* its launches on the internal streams */
    if (p->last_use) {
        if (p->used && idle) {
            if (hipEventRecord(p->last_use, p->last_stream) == hipSuccess) idle = wait_event(e, p->last_use, "aacg_plan_destroy: the plan's last launch") == AACG_OK;
            else { (void)hipGetLastError(); idle = quiesce(e, "aacg_plan_destroy") == AACG_OK; }        /* the stream is gone: wait for everything of the engine */
        }
        (void)hipEventDestroy(p->last_use);
    }
    if (!idle) {                                        /* a launch that may still be running reads these buffers: they are not recycled */
        if (e->pipe.plan == p) e->pipe.plan = nullptr;
        delete p;
        return;
    }
    void* const ptr[AACG_PLAN_BUFFERS] = {p->d_units, p->d_runs, p->d_tns, p->d_scratch, p->d_spec, p->d_cce[0], p->d_cce[1], p->d_cce[2], p->d_cce[3],
                           p->d_rv[0], p->d_rv[1], p->d_rv[2], p->d_rv[3]};
    for (int i = 0; i < AACG_PLAN_BUFFERS; i++) pool_give(e, ptr[i], p->bytes[i]);
    if (e->pipe.plan == p) e->pipe.plan = nullptr;
    delete p;
}

static int plan_check_parity(aacg_engine* e, const aacg_plan* p)
{
    const int flip = (int)(p->launches % AACG_OV_BUFFERS);
    for (const aacg_chain& c : p->h.chains)
        for (int k = 0; k < c.n_ch; k++)
            if (e->parity[(size_t)c.stream * (size_t)e->cfg.max_channels + c.channel + k] != (c.parity[k] + flip) % AACG_OV_BUFFERS) {
                e->err = "plan is stale: another plan advanced one of its streams";
                return AACG_ERR_STALE_PLAN;
            }
    return AACG_OK;
}

/* every chain of the plan has moved on by one overlap buffer */
static void plan_advance(aacg_engine* e, aacg_plan* p)
{
    for (const aacg_chain& c : p->h.chains)
        for (int k = 0; k < c.n_ch; k++) {
            uint8_t& b = e->parity[(size_t)c.stream * (size_t)e->cfg.max_channels + c.channel + k];
            b = (uint8_t)((b + 1) % AACG_OV_BUFFERS);
        }
    p->launches++;
    p->seen_epoch = ++e->epoch;
}

/* the device buffers a route needs are the ones the plan was made with (aacg_debug_set_route may have changed since) */
static int plan_check_route(aacg_engine* e, const aacg_plan* p, const aacg_route& R)
{
    if ((R.rv && !p->d_rv[0]) || (!R.rv && R.has_run && !p->d_runs) || (R.stage != AACG_STAGE_NONE && !p->d_spec)) {
        e->err = "the plan was made for another route (aacg_debug_set_route changed since)";
        return AACG_ERR_STALE_PLAN;
    }
    return AACG_OK;
}

int aacg_decode_device(aacg_engine* e, aacg_plan* p, const void* d_coeffs, const aacg_band_meta* d_meta,
                       void* d_pcm, void* hip_stream)
{
    if (!e || !p || p->e != e || !d_coeffs || !d_pcm) return AACG_ERR_INVALID_ARG;
    const bool quant = e->cfg.input_kind == AACG_INPUT_QUANT_I16;
    if (quant && !d_meta) { e->err = "QUANT_I16 engine needs band meta"; return AACG_ERR_INVALID_ARG; }
    /* relaunched back to back (nothing else advanced any stream since): the per-chain check is known to pass */
    int rc = p->seen_epoch == e->epoch ? AACG_OK : plan_check_parity(e, p);
    if (rc) return rc;
    const aacg_route R = route_of(e, p->h, false);
    if ((rc = plan_check_route(e, p, R))) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    note_stream(e, s);
    if (!p->used) HIP_TRY(e, hipStreamWaitEvent(s, p->uploaded, 0), AACG_ERR_NO_DEVICE);
    /* pipelined launches in flight (of this plan or of another: the caller orders PLANS, the engine its own streams) first */
    if ((rc = pipe_join(e, s))) return rc;
    /* the plan's previous launch ran on ANOTHER stream: this one continues its overlap state, reuses its scratch areas and
     * rendezvous cells, so it is ordered behind it on the device (an event only when the stream changes: per launch it
     * would cost 3 us) */
    if (p->used && p->last_stream != s && !p->last_pipelined) {
        HIP_TRY(e, hipEventRecord(p->last_use, p->last_stream), AACG_ERR_NO_DEVICE);
        HIP_TRY(e, hipStreamWaitEvent(s, p->last_use, 0), AACG_ERR_NO_DEVICE);
    }
    const cce_bufs cb = {(const aacg_run*)p->d_cce[0], (const aacg_couple_job*)p->d_cce[1], (const float*)p->d_cce[2], (float*)p->d_cce[3]};
    const rv_bufs rvb = {(const aacg_run*)p->d_rv[0], (const aacg_rv_link*)p->d_rv[1], (unsigned long long*)p->d_rv[2], (float*)p->d_rv[3]};
    const xl_args serial = {false, 0ull, 0};
    rc = launch_run(e, R, p->units_now(), p->d_runs, p->d_tns, p->d_scratch, p->d_spec, cb, rvb, p->h, d_coeffs, d_meta, d_pcm,
                    (int)(p->launches % AACG_OV_BUFFERS), s, serial, nullptr);
    if (rc) return rc;
    p->last_stream = s;
    p->used = true;
    p->last_pipelined = false;
    plan_advance(e, p);
    return AACG_OK;
}

/* ---- pipelined launches ---------------------------------------------------------------- */
/* true if a kernel on b runs while one on a is running (both idle before) */
static bool streams_overlap(aacg_engine* e, hipStream_t a, hipStream_t b, unsigned* d_probe, unsigned* h_seen)
{
    aacg_wait_policy quick = e->wait; quick.limit_s = quick.limit_s < 2.0 ? quick.limit_s : 2.0;
    if (hipMemsetAsync(d_probe, 0, 8, a) != hipSuccess || aacg_wait_stream(a, quick) != hipSuccess) { (void)hipGetLastError(); return false; }
    hipLaunchKernelGGL(aacg_probe_wait, dim3(1), dim3(1), 0, a, d_probe, d_probe + 1, 50000LL);      /* at most 0.5 ms (100 MHz clock) */
    hipLaunchKernelGGL(aacg_probe_set, dim3(1), dim3(1), 0, b, d_probe);
    *h_seen = 0;
    if (aacg_wait_stream(b, quick) != hipSuccess || hipMemcpyAsync(h_seen, d_probe + 1, 4, hipMemcpyDeviceToHost, a) != hipSuccess ||
        aacg_wait_stream(a, quick) != hipSuccess) { (void)hipGetLastError(); return false; }
    return *h_seen != 0;
}

/* The pipeline's streams and events, made in locals and committed to the engine only when ALL of them exist: a failure half-way
 * destroys what was made and leaves e->pipe untouched, so that the next call tries again (the first stream used
 * to be the "ready" flag, and a later failure left null streams and events behind it). */
static int pipe_setup(aacg_engine* e)
{
    aacg_engine::pipe_t& pp = e->pipe;
    if (pp.stream[0]) return AACG_OK;
    HIP_TRY(e, hipSetDevice(e->cfg.device_ordinal), AACG_ERR_NO_DEVICE);     /* the streams belong to the engine's device, whatever the caller's current one is */
    /* Streams that really run side by side.  Streams of the highest priority are dealt their hardware queues apart from the
     * crowd of ordinary streams a host process may have made (PyTorch: 32 at once); every candidate is PROBED against the
     * streams already chosen, and further candidates are tried if it shares a queue with one of them after all.  Streams that
     * never overlap still decode correctly — serially.  (Highest priority also means: ahead of the host process's own
     * ordinary streams when both have work; the pipeline's launches are 11 us each, nothing starves.) */
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    struct made_t {
        hipStream_t stream[AACG_PIPE_STREAMS] = {};
        hipStream_t spare[8] = {}; int n_spare = 0;
        hipEvent_t mark[AACG_PIPE_RING][AACG_PIPE_STREAMS] = {}, tail[AACG_PIPE_STREAMS] = {}, fork = nullptr;
        unsigned* d_probe = nullptr; unsigned* h_seen = nullptr;
        void release(bool keep) {
            for (int i = 0; i < n_spare; i++) if (spare[i]) (void)hipStreamDestroy(spare[i]);
            if (d_probe) (void)hipFree(d_probe);
            if (h_seen) (void)hipHostFree(h_seen);
            if (keep) return;
            for (auto& round : mark) for (hipEvent_t ev : round) if (ev) (void)hipEventDestroy(ev);
            for (hipEvent_t ev : tail) if (ev) (void)hipEventDestroy(ev);
            if (fork) (void)hipEventDestroy(fork);
            for (hipStream_t st : stream) if (st) (void)hipStreamDestroy(st);
        }
    } m;
#define PIPE_TRY(call, code) do { if (!hip_ok(e, (call), #call)) { m.release(false); return (code); } } while (0)
    PIPE_TRY(hipMalloc((void**)&m.d_probe, 8), AACG_ERR_OUT_OF_MEMORY);
    PIPE_TRY(hipHostMalloc((void**)&m.h_seen, 4, hipHostMallocDefault), AACG_ERR_OUT_OF_MEMORY);
    PIPE_TRY(hipStreamCreateWithPriority(&m.stream[0], hipStreamNonBlocking, greatest), AACG_ERR_NO_DEVICE);
    int have = 1;
    for (int attempt = 0; attempt < 8 && have < AACG_PIPE_STREAMS; attempt++) {
        hipStream_t cand = nullptr;
        PIPE_TRY(hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, attempt < 6 ? greatest : 0), AACG_ERR_NO_DEVICE);
        bool apart = true;
        for (int k = 0; k < have && apart; k++) apart = streams_overlap(e, m.stream[k], cand, m.d_probe, m.h_seen);
        if (apart) m.stream[have++] = cand; else m.spare[m.n_spare++] = cand;
    }
    const bool concurrent = have == AACG_PIPE_STREAMS;
    while (have < AACG_PIPE_STREAMS) { m.stream[have++] = m.spare[--m.n_spare]; m.spare[m.n_spare] = nullptr; }
    /* events that order and nothing else: no time stamps, no system-scope fence at the record (AACG_WAIT_BLOCK, a measurement
     * mode: interrupt-driven waits) */
    const unsigned flags = hipEventDisableTiming | hipEventDisableSystemFence | (e->wait.mode == AACG_WAIT_BLOCK ? hipEventBlockingSync : 0u);
    for (auto& round : m.mark) for (hipEvent_t& ev : round) PIPE_TRY(hipEventCreateWithFlags(&ev, flags), AACG_ERR_NO_DEVICE);
    for (hipEvent_t& ev : m.tail) PIPE_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence), AACG_ERR_NO_DEVICE);
    PIPE_TRY(hipEventCreateWithFlags(&m.fork, hipEventDisableTiming | hipEventDisableSystemFence), AACG_ERR_NO_DEVICE);
#undef PIPE_TRY
    m.release(true);
    std::memcpy(pp.mark, m.mark, sizeof pp.mark);
    std::memcpy(pp.tail, m.tail, sizeof pp.tail);
    pp.fork = m.fork;
    pp.concurrent = concurrent;
    std::memcpy(pp.stream, m.stream, sizeof pp.stream);      /* last: stream[0] is what says "set up" */
    if (!concurrent)
        e->pipe_note = "aacg_decode_pipelined: the engine's internal HIP streams were NOT seen to run side by side when the pipeline was set up "
                       "(the runtime put them on one hardware queue): pipelined launches are correct and run one behind the other (aacg_pipeline_info)";
    return AACG_OK;
}

int aacg_decode_pipelined(aacg_engine* e, aacg_plan* p, const void* d_coeffs, const aacg_band_meta* d_meta, void* d_pcm)
{
    return aacg_decode_pipelined_timed(e, p, d_coeffs, d_meta, d_pcm, nullptr);
}

/* the same with a timing mark (aacg_timer_create) bound to the launch's completion: measurement only (aacgpu_tools.h) */
int aacg_decode_pipelined_timed(aacg_engine* e, aacg_plan* p, const void* d_coeffs, const aacg_band_meta* d_meta, void* d_pcm, void* stop_mark)
{
    if (!e || !p || p->e != e || !d_coeffs || !d_pcm) return AACG_ERR_INVALID_ARG;
    const bool quant = e->cfg.input_kind == AACG_INPUT_QUANT_I16;
    if (quant && !d_meta) { e->err = "QUANT_I16 engine needs band meta"; return AACG_ERR_INVALID_ARG; }
    int rc = p->seen_epoch == e->epoch ? AACG_OK : plan_check_parity(e, p);
    if (rc) return rc;
    const aacg_route R = route_of(e, p->h, true);
    if ((rc = plan_check_route(e, p, R)) || (rc = pipe_setup(e))) return rc;
    aacg_engine::pipe_t& pp = e->pipe;
    /* a route whose launches cannot overlap (optional stages, coupling, int16 PCM: no rendezvous build) runs on the pipeline's
     * first stream, launch behind launch: the stream orders them, no event is needed between two of them */
    const bool follows = !R.overlappable && pp.open && pp.serial && pp.plan == p && p->last_pipelined && p->seen_epoch == e->epoch;
    /* Does this launch continue the one before it — same plan, nothing in between, a route whose chains meet in cells?  Then
     * the two may overlap: its input state arrives through the cross-launch cells, tagged with that launch's epoch.  Its
     * stream puts it behind launch n - AACG_PIPE_STREAMS; the event waits of aacg_pipeline_order (aacg_routes.cpp) do the rest:
     * together every launch is behind every launch up to n - AACG_OV_BUFFERS + 1, whose buffers and cells it reuses
     * (aacg_device.h).  Otherwise it starts behind everything in flight, from complete state. */
    const bool continues = R.overlappable && pp.open && !pp.serial && pp.plan == p && p->last_pipelined && p->seen_epoch == e->epoch;
    if (!continues) { pp.n = 0; pp.streams = aacg_pipeline_streams(p->h, R.run_key); std::memset(pp.seen, 0, sizeof pp.seen); }      /* a new sequence */
    const aacg_pipe_order ord = aacg_pipeline_order(pp.n, pp.streams);
    hipStream_t s = R.overlappable ? pp.stream[ord.stream] : pp.stream[0];
    if (continues) {
        /* back-pressure instead of cross-stream waits: the HOST waits until the round AACG_PIPE_DEPTH(streams) back is complete — no
         * barrier packet enters a GPU queue (one per two launches cost the three-stream pipeline all it had gained), and the
         * queues still hold a round of launches when the host comes back */
        if (ord.sync_round >= 0) {
            /* (bounded: if an event has not completed after two seconds — a stalled device, a driver that lost a signal — the
             * ordering moves to the GPU for this round, which is always correct, instead of leaving the caller in a wait) */
            aacg_wait_policy bp = e->wait; bp.limit_s = bp.limit_s < 2.0 ? bp.limit_s : 2.0;
            for (hipEvent_t ev : pp.seen[((uint64_t)ord.sync_round / AACG_PIPE_MARK) % AACG_PIPE_RING]) {
                if (!ev) continue;
                /* aacg_wait.h: a few microseconds of polling (the GPU is seldom more than a launch behind), then the caller's core
                 * is given back between polls — round 5 spun here without pause */
                const hipError_t st = aacg_wait_event(ev, bp);
                if (st == hipErrorNotReady) {
                    for (hipStream_t ps : pp.stream) HIP_TRY(e, hipStreamWaitEvent(ps, ev, 0), AACG_ERR_NO_DEVICE);
                } else HIP_TRY(e, st, AACG_ERR_NO_DEVICE);
            }
        }
    } else if (!follows) {
        if ((rc = pipe_join(e, s))) return rc;
        if (p->used && !p->last_pipelined) {
            HIP_TRY(e, hipEventRecord(p->last_use, p->last_stream), AACG_ERR_NO_DEVICE);
            HIP_TRY(e, hipStreamWaitEvent(s, p->last_use, 0), AACG_ERR_NO_DEVICE);
        }
        if (e->last_kernel) HIP_TRY(e, hipStreamWaitEvent(s, e->last_kernel, 0), AACG_ERR_NO_DEVICE);   /* the host-buffer path's batches */
    }
    if (!p->used) HIP_TRY(e, hipStreamWaitEvent(s, p->uploaded, 0), AACG_ERR_NO_DEVICE);
    if (!continues && R.overlappable) {
        /* a new sequence: its next launches go to the other streams and are not behind this one — but they must be behind
         * everything this one is behind (an earlier sequence's launches on a third stream may share overlap buffers with them:
         * the back-pressure bookkeeping starts anew with the sequence) */
        HIP_TRY(e, hipEventRecord(pp.fork, s), AACG_ERR_NO_DEVICE);
        for (hipStream_t ps : pp.stream) if (ps != s) HIP_TRY(e, hipStreamWaitEvent(ps, pp.fork, 0), AACG_ERR_NO_DEVICE);
    }
    const cce_bufs cb = {(const aacg_run*)p->d_cce[0], (const aacg_couple_job*)p->d_cce[1], (const float*)p->d_cce[2], (float*)p->d_cce[3]};
    const rv_bufs rvb = {(const aacg_run*)p->d_rv[0], (const aacg_rv_link*)p->d_rv[1], (unsigned long long*)p->d_rv[2], (float*)p->d_rv[3]};
    const xl_args xl = {R.overlappable, continues ? pp.epoch : 0ull, ord.stream, (int)(pp.n & 3u)};
    unsigned long long epoch = 0;
    /* the host will wait for this one: its event rides on the dispatch itself where the route is a single launch
     * (no marker packet between this launch and the next of its stream), else it is recorded behind the route's last launch */
    const bool ordered = R.overlappable && ord.marked;
    const size_t slot = (size_t)((pp.n / (uint64_t)pp.streams / AACG_PIPE_MARK) % AACG_PIPE_RING);
    /* a caller's timing mark bound to this launch stands for the engine's own event (it must stay alive until the pipeline
     * has been joined: aacgpu_tools.h) — a second event would be a marker packet in the queue behind the dispatch */
    hipEvent_t const mine = (stop_mark && R.rv) ? (hipEvent_t)stop_mark : pp.mark[slot][ord.stream];
    if (ordered) pp.seen[slot][ord.stream] = mine;
    hipEvent_t bound = R.rv ? (stop_mark ? (hipEvent_t)stop_mark : (ordered ? mine : nullptr)) : nullptr;
    rc = launch_run(e, R, p->units_now(), p->d_runs, p->d_tns, p->d_scratch, p->d_spec, cb, rvb, p->h, d_coeffs, d_meta, d_pcm,
                    (int)(p->launches % AACG_OV_BUFFERS), s, xl, &epoch, bound);
    if (rc) return rc;
    if (ordered && bound != mine) HIP_TRY(e, hipEventRecord(mine, s), AACG_ERR_NO_DEVICE);
    if (stop_mark && bound != (hipEvent_t)stop_mark) HIP_TRY(e, hipEventRecord((hipEvent_t)stop_mark, s), AACG_ERR_NO_DEVICE);
    if (continues) pp.chained++;
    if (R.overlappable) pp.n++;
    pp.issued++;
    pp.serial = !R.overlappable;
    pp.open = true;
    pp.plan = p;
    pp.epoch = epoch;
    p->last_stream = s;
    p->used = true;
    p->last_pipelined = true;
    plan_advance(e, p);
    if (!e->pipe_note.empty()) { e->err = e->pipe_note; e->pipe_note.clear(); }   /* once: aacg_last_error after a successful call */
    return AACG_OK;
}

int aacg_pipeline_fork(aacg_engine* e, void* hip_stream)
{
    if (!e || !hip_stream) return AACG_ERR_INVALID_ARG;
    int rc = pipe_setup(e);
    if (rc) return rc;
    HIP_TRY(e, hipEventRecord(e->pipe.fork, (hipStream_t)hip_stream), AACG_ERR_NO_DEVICE);
    for (hipStream_t st : e->pipe.stream) HIP_TRY(e, hipStreamWaitEvent(st, e->pipe.fork, 0), AACG_ERR_NO_DEVICE);
    return AACG_OK;
}

int aacg_pipeline_join(aacg_engine* e, void* hip_stream)
{
    if (!e) return AACG_ERR_INVALID_ARG;
    return pipe_join(e, (hipStream_t)hip_stream);
}

uint64_t aacg_pipeline_chained(const aacg_engine* e) { return e ? e->pipe.chained : 0; }
int aacg_pipeline_concurrent(const aacg_engine* e) { return e && e->pipe.concurrent ? 1 : 0; }
int aacg_pipeline_streams_used(const aacg_engine* e) { return e && e->pipe.stream[0] ? e->pipe.streams : 0; }
int aacg_pipeline_info(const aacg_engine* e, int* streams_used, int* concurrent)
{
    if (!e) return AACG_ERR_INVALID_ARG;
    if (streams_used) *streams_used = e->pipe.stream[0] ? e->pipe.streams : 0;
    if (concurrent) *concurrent = e->pipe.stream[0] ? (e->pipe.concurrent ? 1 : 0) : -1;
    return AACG_OK;
}

/* The plan's device unit records take what the parser found (device to device); the run tables stay. */
int aacg_plan_refresh_from_parse(aacg_engine* e, aacg_plan* p, const aacg_unit_desc* d_parsed_units,
                                 const aacg_parse_result* d_results, uint32_t max_units, uint32_t* d_refused, void* hip_stream)
{
    /* (without a map the kernel only reads the results) */
    return aacg_plan_refresh_from_parse_ex(e, p, d_parsed_units, const_cast<aacg_parse_result*>(d_results), max_units, nullptr, p ? p->cur_set : 0, d_refused, hip_stream);
}

/* More sets of unit records for a plan that is refreshed while its earlier launches are in flight: the buffer is made anew
 * (n_sets copies of the planner's records), before the plan's first launch. */
int aacg_plan_set_unit_sets(aacg_engine* e, aacg_plan* p, uint32_t n_sets)
{
    if (!e || !p || p->e != e || n_sets < 1 || n_sets > 8) return AACG_ERR_INVALID_ARG;
    if (p->used || p->launches) { e->err = "aacg_plan_set_unit_sets: before the plan's first launch"; return AACG_ERR_INVALID_ARG; }
    if (n_sets == p->unit_sets) return AACG_OK;
    HIP_TRY(e, hipSetDevice(e->cfg.device_ordinal), AACG_ERR_NO_DEVICE);
    const size_t ub = sizeof(aacg_dev_unit) * p->n_units;
    size_t got = 0;
    void* fresh = pool_take(e, ub * n_sets, &got);
    if (!fresh) return AACG_ERR_OUT_OF_MEMORY;
    for (uint32_t k = 0; k < n_sets; k++)
        if (!hip_ok(e, hipMemcpyAsync((char*)fresh + ub * k, p->h.units.data(), ub, hipMemcpyHostToDevice, e->stream), "upload plan unit sets")) { pool_give(e, fresh, got); return AACG_ERR_NO_DEVICE; }
    pool_give(e, p->d_units, p->bytes[0]);               /* its upload is in front of whatever takes it next on the same stream */
    p->d_units = (aacg_dev_unit*)fresh; p->bytes[0] = got;
    p->unit_sets = n_sets; p->cur_set = 0;
    HIP_TRY(e, hipEventRecord(p->uploaded, e->stream), AACG_ERR_NO_DEVICE);
    return AACG_OK;
}

int aacg_plan_refresh_from_parse_ex(aacg_engine* e, aacg_plan* p, const aacg_unit_desc* d_parsed_units, aacg_parse_result* d_results,
                                    uint32_t max_units, const aacg_refresh_map* d_map, uint32_t set, uint32_t* d_refused, void* hip_stream)
{
    if (!e || !p || p->e != e || !d_parsed_units || !d_results || !max_units || !d_refused || set >= p->unit_sets) return AACG_ERR_INVALID_ARG;
    if (e->cfg.input_kind != AACG_INPUT_QUANT_I16) { e->err = "aacg_plan_refresh_from_parse needs a QUANT_I16 engine"; return AACG_ERR_INVALID_ARG; }
    if (p->h.any_tns || p->h.any_pns || e->cfg.tns_mode == AACG_TNS_SPEC) {
        e->err = "aacg_plan_refresh_from_parse: TNS records / noise tables are prepared on the host, such plans are rebuilt per batch";
        return AACG_ERR_UNSUPPORTED;
    }
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    note_stream(e, s);
    HIP_TRY(e, hipStreamWaitEvent(s, p->uploaded, 0), AACG_ERR_NO_DEVICE);
    if (p->unit_sets > 1) {
        /* a set of its own: nothing in flight reads it (the caller's word), so neither the pipeline is joined nor the plan's
         * sequence of overlapped launches ended; the next launch reads this set */
        aacg_refresh_launch(p->d_units + (size_t)set * p->n_units, d_parsed_units, d_results, d_map, p->n_units, max_units, 1, d_refused, s);
        HIP_TRY(e, hipGetLastError(), AACG_ERR_NO_DEVICE);
        p->cur_set = set;
        return AACG_OK;
    }
    if (p->last_pipelined) { int jrc = pipe_join(e, s); if (jrc) return jrc; }
    if (p->used && p->last_stream != s && !p->last_pipelined) {              /* the records' readers on the plan's previous stream first */
        HIP_TRY(e, hipEventRecord(p->last_use, p->last_stream), AACG_ERR_NO_DEVICE);
        HIP_TRY(e, hipStreamWaitEvent(s, p->last_use, 0), AACG_ERR_NO_DEVICE);
    }
    aacg_refresh_launch(p->d_units, d_parsed_units, d_results, d_map, p->n_units, max_units, 1, d_refused, s);
    HIP_TRY(e, hipGetLastError(), AACG_ERR_NO_DEVICE);
    p->last_stream = s;
    p->used = true;
    p->last_pipelined = false;
    return AACG_OK;
}

/* The plan's unit records rewritten from the host's next batch of the same structure; the run tables stay. */
int aacg_plan_refresh_units(aacg_engine* e, aacg_plan* p, const aacg_unit_desc* units, uint32_t n_units, void* hip_stream)
{
    if (!e || !p || p->e != e || !units) return AACG_ERR_INVALID_ARG;
    int rc = aacg_plan_refresh_host(&p->h, units, n_units, e->cfg.sample_index, e->cfg.tns_mode == AACG_TNS_SPEC, &e->err);
    if (rc) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    note_stream(e, s);
    HIP_TRY(e, hipStreamWaitEvent(s, p->uploaded, 0), AACG_ERR_NO_DEVICE);
    /* the plan's previous launch on ANOTHER stream may still be reading the records this copy overwrites: order behind it */
    if (p->last_pipelined) { int jrc = pipe_join(e, s); if (jrc) return jrc; }
    if (p->used && p->last_stream != s && !p->last_pipelined) {
        HIP_TRY(e, hipEventRecord(p->last_use, p->last_stream), AACG_ERR_NO_DEVICE);
        HIP_TRY(e, hipStreamWaitEvent(s, p->last_use, 0), AACG_ERR_NO_DEVICE);
    }
    /* pageable source: the runtime stages it before returning, so the host copy may change again right away; in stream
     * order behind the launches that read the previous records */
    HIP_TRY(e, hipMemcpyAsync(p->d_units, p->h.units.data(), sizeof(aacg_dev_unit) * p->h.units.size(), hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    if (p->h.runs_moved) {                                  /* the batch's blocks lie elsewhere: the run tables' copies of the offsets with them */
        if (p->d_runs) HIP_TRY(e, hipMemcpyAsync(p->d_runs, p->h.runs.data(), sizeof(aacg_run) * p->h.runs.size(), hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
        if (p->d_rv[0]) HIP_TRY(e, hipMemcpyAsync(p->d_rv[0], p->h.runs_rv.data(), sizeof(aacg_run) * p->h.runs_rv.size(), hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    }
    p->last_stream = s;
    p->used = true;
    p->last_pipelined = false;
    return AACG_OK;
}

int aacg_spectral_device(aacg_engine* e, aacg_plan* p, const void* d_coeffs, const aacg_band_meta* d_meta,
                         float* d_spec_out, void* hip_stream)
{
    if (!e || !p || p->e != e || !d_coeffs || !d_meta || !d_spec_out) return AACG_ERR_INVALID_ARG;
    if (e->cfg.input_kind != AACG_INPUT_QUANT_I16) { e->err = "spectral stage needs a QUANT_I16 engine"; return AACG_ERR_INVALID_ARG; }
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    note_stream(e, s);
    aacg_kparams P;
    std::memset(&P, 0, sizeof P);
    P.units = p->units_now(); P.coeffs = d_coeffs; P.meta = d_meta; P.spec_out = d_spec_out; P.tab = e->d_tab;
    HIP_TRY(e, hipStreamWaitEvent(s, p->uploaded, 0), AACG_ERR_NO_DEVICE);
    hipLaunchKernelGGL(aacg_spectral, dim3((p->n_units + AACG_WG_WAVES - 1) / AACG_WG_WAVES), dim3(AACG_WG_THREADS),
                       AACG_LDS_BYTES_SPECTRAL, s, P, (int)p->n_units);
    HIP_TRY(e, hipGetLastError(), AACG_ERR_NO_DEVICE);
    p->last_stream = s;
    p->used = true;
    return AACG_OK;
}

int aacg_synchronize(aacg_engine* e, void* hip_stream)
{
    if (!e) return AACG_ERR_INVALID_ARG;
    HIP_TRY(e, hipSetDevice(e->cfg.device_ordinal), AACG_ERR_NO_DEVICE);
    const int rc = wait_stream(e, hip_stream ? (hipStream_t)hip_stream : e->stream, "aacg_synchronize");
    if (rc) return rc;
    return pipe_join(e, nullptr);                       /* and whatever aacg_decode_pipelined has in flight */
}

/* ---- host-buffer path: process(elements) + interleave for a batch ------------------- */
/* Page-locked memory for a caller's buffers.  Small blocks: hipHostMalloc.  Large ones (a batch's PCM: 32 MiB and more, FRESH for
 * every flush of a host that keeps what readChunk() returned) are what round 6 measured (tools/micro/pinned_alloc.hip,
 * profiles/r06_pinned_alloc.txt): hipHostMalloc of 32 MiB 3.5-6 ms, two threads at it together 11 ms for both — of which the
 * runtime's own part is small: the kernel zeroing and mapping 8192 fresh pages one fault at a time is 3.1 ms.  The same bytes as
 * sixteen 2 MiB pages (MADV_HUGEPAGE), faulted in by four threads at once, 0.5 ms; hipHostRegister of pages that are there 0.07 ms;
 * a copy down into it runs at the rate of hipHostMalloc memory (0.60 ms per 32 MiB) and hipPointerGetAttributes calls it host
 * memory like the other. */
namespace {
struct host_block { void* base; size_t mapped; };
std::mutex g_host_lock;
std::unordered_map<void*, host_block> g_host_blocks;
const size_t HOST_HUGE = (size_t)2u << 20, HOST_LARGE = (size_t)8u << 20;
}
void* aacg_host_alloc(size_t bytes)
{
    if (bytes >= HOST_LARGE) {
        const size_t len = (bytes + HOST_HUGE - 1) & ~(HOST_HUGE - 1);
        void* base = mmap(nullptr, len + HOST_HUGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (base != MAP_FAILED) {
            char* q = (char*)(((uintptr_t)base + HOST_HUGE - 1) & ~(uintptr_t)(HOST_HUGE - 1));
            (void)madvise(q, len, MADV_HUGEPAGE);
            const unsigned hw = std::thread::hardware_concurrency(), T = hw >= 8 ? 4u : hw >= 2 ? 2u : 1u;
            auto touch = [q, len, T](unsigned k) {
                const size_t pieces = len / HOST_HUGE, a = pieces * k / T * HOST_HUGE, b = pieces * (k + 1) / T * HOST_HUGE;
                for (size_t i = a; i < b; i += 4096) ((volatile char*)q)[i] = 0;          /* (4 KiB steps: where huge pages are off, every page still gets its fault here) */
            };
            std::vector<std::thread> helpers;
            bool spawned = true;
            try { for (unsigned k = 1; k < T; k++) helpers.emplace_back(touch, k); } catch (...) { spawned = false; }
            touch(0);
            for (auto& h : helpers) h.join();
            if (!spawned) for (unsigned k = (unsigned)helpers.size() + 1; k < T; k++) touch(k);
            if (hipHostRegister(q, len, hipHostRegisterDefault) == hipSuccess) {
                std::lock_guard<std::mutex> g(g_host_lock);
                g_host_blocks[q] = host_block{base, len + HOST_HUGE};
                return q;
            }
            (void)hipGetLastError();
            (void)munmap(base, len + HOST_HUGE);
        }
    }
    void* p = nullptr;
    return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
void aacg_host_free(void* p)
{
    if (!p) return;
    host_block b{nullptr, 0};
    {
        std::lock_guard<std::mutex> g(g_host_lock);
        auto it = g_host_blocks.find(p);
        if (it != g_host_blocks.end()) { b = it->second; g_host_blocks.erase(it); }
    }
    if (b.base) { (void)hipHostUnregister(p); (void)munmap(b.base, b.mapped); }
    else (void)hipHostFree(p);
}

int aacg_wait(aacg_engine* e, uint64_t ticket)
{
    if (!e || ticket == 0 || ticket > e->submitted) return AACG_ERR_INVALID_ARG;
    aacg_engine::slot_t& sl = e->slot[(ticket - 1) & 1];
    if (sl.busy) {
        const int rc = wait_event(e, sl.done, "aacg_wait");
        if (rc) return rc;
        sl.busy = false;
        if (sl.user_pcm) { std::memcpy(sl.user_pcm, sl.h_pcm, sl.user_pcm_bytes); sl.user_pcm = nullptr; }
    }
    return AACG_OK;
}

int aacg_submit(aacg_engine* e, const aacg_unit_desc* units, uint32_t n_units,
                const void* coeffs, uint32_t n_coef_blocks,
                const aacg_band_meta* meta, uint32_t n_meta,
                void* pcm_out, size_t n_pcm_floats, uint64_t* ticket)
{
    return aacg_submit_tns(e, units, n_units, coeffs, n_coef_blocks, meta, n_meta, nullptr, 0, pcm_out, n_pcm_floats, ticket);
}

int aacg_submit_tns(aacg_engine* e, const aacg_unit_desc* units, uint32_t n_units,
                    const void* coeffs, uint32_t n_coef_blocks,
                    const aacg_band_meta* meta, uint32_t n_meta,
                    const aacg_tns_info* tns, uint32_t n_tns,
                    void* pcm_out, size_t n_pcm_floats, uint64_t* ticket)
{
    const aacg_batch b = {units, n_units, coeffs, n_coef_blocks, meta, n_meta, tns, n_tns, nullptr, 0, pcm_out, n_pcm_floats};
    return aacg_submit_ex(e, &b, ticket);
}

int aacg_submit_ex(aacg_engine* e, const aacg_batch* batch, uint64_t* ticket)
{
    if (!e || !batch) return AACG_ERR_INVALID_ARG;
    const aacg_unit_desc* units = batch->units; const uint32_t n_units = batch->n_units;
    const void* coeffs = batch->coeffs; const uint32_t n_coef_blocks = batch->n_coef_blocks;
    const aacg_band_meta* meta = batch->meta; const uint32_t n_meta = batch->n_meta;
    const aacg_tns_info* tns = batch->tns; uint32_t n_tns = batch->n_tns;
    const aacg_cce_info* cce = batch->cce; uint32_t n_cce = batch->n_cce;
    void* pcm_out = batch->pcm_out; const size_t n_pcm_floats = batch->n_pcm_floats;
    if (!units || !n_units || !coeffs || !pcm_out || !ticket) return AACG_ERR_INVALID_ARG;
    if (e->cfg.tns_mode != AACG_TNS_SPEC) { tns = nullptr; n_tns = 0; }   /* REFERENCE mode: TNS is the identity */
    if (e->cfg.cce_mode != AACG_CCE_SPEC) { cce = nullptr; n_cce = 0; }   /* REFERENCE mode: a coupling element in the batch is refused */
    const bool quant = e->cfg.input_kind == AACG_INPUT_QUANT_I16;
    if (quant && !meta) { e->err = "QUANT_I16 engine needs band meta"; return AACG_ERR_INVALID_ARG; }
    if (e->cfg.max_batch_units > 0 && (int)n_units > e->cfg.max_batch_units) {
        e->err = "batch exceeds max_batch_units";
        return AACG_ERR_CAPACITY;
    }
    HIP_TRY(e, hipSetDevice(e->cfg.device_ordinal), AACG_ERR_NO_DEVICE);
    aacg_engine::slot_t& sl = e->slot[e->submitted & 1];
    if (sl.busy) {                                     /* the batch two submissions ago was never waited for */
        int wrc = aacg_wait(e, e->submitted - 1);
        if (wrc) return wrc;
    }
    if (!sl.stream) {
        HIP_TRY(e, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking), AACG_ERR_NO_DEVICE);
        HIP_TRY(e, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming), AACG_ERR_NO_DEVICE);
        HIP_TRY(e, hipEventCreateWithFlags(&sl.kernel_done, hipEventDisableTiming), AACG_ERR_NO_DEVICE);
    }

    int rc = aacg_plan_build(units, n_units, e->cfg.sample_index, e->cfg.max_streams, e->cfg.max_channels,
                             e->parity.data(), &sl.h, &e->err, tns, n_tns, cce, n_cce);
    if (rc) return rc;
    const aacg_plan_host& h = sl.h;
    if (h.coef_blocks > n_coef_blocks || (quant && h.meta_blocks > n_meta) || h.pcm_floats > n_pcm_floats) {
        e->err = "a unit points outside the coefficient / meta / pcm buffers";
        return AACG_ERR_INVALID_ARG;
    }
    if (quant) {
        /* PNS bands: aac.js's generator degenerates to NaN output (ics.js:234,239);
         * like the reference's other unsupported tools this is refused, not silently altered. */
        for (uint32_t i = 0; i < n_units; i++)
            for (int c = 0; c < units[i].n_ch; c++) {
                const aacg_chan_info& ci = units[i].ch[c];
                const aacg_band_meta& m = meta[units[i].meta_offset + (uint32_t)c];
                for (int b = 0; b < ci.group_count * ci.max_sfb; b++)
                    if ((m.band[b] >> AACG_META_BT_SHIFT) == AACG_NOISE_BT) {
                        if (e->cfg.pns_mode != AACG_PNS_SPEC) {
                            e->err = "NOISE_BT (PNS) band: not decodable by the reference either";
                            return AACG_ERR_UNSUPPORTED;
                        }
                        if (!(units[i].flags & AACG_UNIT_HAS_PNS)) {
                            e->err = "a unit has a NOISE_BT band but not AACG_UNIT_HAS_PNS";
                            return AACG_ERR_INVALID_ARG;
                        }
                    }
            }
        if (h.any_pns && e->cfg.pns_mode != AACG_PNS_SPEC) {
            e->err = "a unit carries AACG_UNIT_HAS_PNS: NOISE_BT bands are not decodable by the reference either";
            return AACG_ERR_UNSUPPORTED;
        }
    }

    const size_t ub = sizeof(aacg_dev_unit) * h.units.size(), rb = sizeof(aacg_run) * h.runs.size();
    const size_t tb = tns_buffer_bytes(h.tns.size());
    const size_t sb = h.needs_scratch ? h.runs.size() * AACG_SLOT_FLOATS * sizeof(float) : 0;
    const size_t xb = needs_spec_buffer(e, h) ? (size_t)n_coef_blocks * 1024u * sizeof(float) : 0;
    const size_t ccb[4] = {sizeof(aacg_run) * h.cce_runs.size(), sizeof(aacg_couple_job) * h.couple_jobs.size(),
                           sizeof(float) * h.gains.size(), (size_t)h.side_blocks * 4096u};
    const void* const cce_src[4] = {h.cce_runs.data(), h.couple_jobs.data(), h.gains.data(), nullptr};
    for (int i = 0; i < 4; i++) if (ccb[i] && (rc = grow(e, &sl.d_cce[i], &sl.cce_cap[i], ccb[i]))) return rc;
    const aacg_route R = route_of(e, h, false);
    const bool rvp = R.rv;
    const size_t rvs[4] = {rvp ? sizeof(aacg_run) * h.runs_rv.size() : 0, rvp ? sizeof(aacg_rv_link) * h.links_rv.size() : 0,
                           rvp ? sizeof(unsigned long long) * AACG_RV_STATE_WORDS * (size_t)h.n_links_rv : 0,
                           rvp ? sizeof(float) * AACG_RV_DATA_FLOATS * (size_t)h.n_links_rv : 0};
    for (int i = 0; i < 4; i++) {
        const size_t had = sl.rv_cap[i];
        if (rvs[i] && (rc = grow(e, &sl.d_rv[i], &sl.rv_cap[i], rvs[i]))) return rc;
        /* a new state buffer starts from zero — on the slot's own stream, in front of the launch that reads it (a plain
         * hipMemset is ordered on the null stream, which a non-blocking stream does not wait for) */
        if (i == 2 && sl.rv_cap[i] != had) HIP_TRY(e, hipMemsetAsync(sl.d_rv[i], 0, sl.rv_cap[i], sl.stream), AACG_ERR_NO_DEVICE);
    }
    const size_t cb = (size_t)n_coef_blocks * 1024u * coef_elem_size(e);
    const size_t mb = quant ? (size_t)n_meta * sizeof(aacg_band_meta) : 0;
    const size_t pb = h.pcm_floats * pcm_elem_size(e);
    if ((rc = grow(e, &sl.d_units, &sl.units_cap, ub)) || (rb && (rc = grow(e, &sl.d_runs, &sl.runs_cap, rb))) ||
        (rc = grow(e, &sl.d_coeffs, &sl.coeffs_cap, cb)) || (quant && (rc = grow(e, &sl.d_meta, &sl.meta_cap, mb))) ||
        (tb && (rc = grow(e, &sl.d_tns, &sl.tns_cap, tb))) || (sb && (rc = grow(e, &sl.d_scratch, &sl.scratch_cap, sb))) || (xb && (rc = grow(e, &sl.d_spec, &sl.spec_cap, xb))) || (rc = grow(e, &sl.d_pcm, &sl.pcm_cap, pb)))
        return rc;

    /* Ordinary (pageable) caller memory goes through the slot's page-locked staging buffers (one host
     * memcpy each way, then truly asynchronous DMA); page-locked caller memory is used in place. */
    hipStream_t s = sl.stream;
    const void* src_coeffs = coeffs;
    const void* src_meta = meta;
    if (!is_pinned(coeffs) || (quant && !is_pinned(meta))) {
        const size_t mb_al = (cb + 255) & ~(size_t)255;
        if ((rc = grow_host(e, &sl.h_in, &sl.h_in_cap, mb_al + mb))) return rc;
        std::memcpy(sl.h_in, coeffs, cb);
        if (quant) std::memcpy((char*)sl.h_in + mb_al, meta, mb);
        src_coeffs = sl.h_in;
        src_meta = (char*)sl.h_in + mb_al;
    }
    void* dst_pcm = pcm_out;
    sl.user_pcm = nullptr;
    if (!is_pinned(pcm_out)) {
        if ((rc = grow_host(e, &sl.h_pcm, &sl.h_pcm_cap, pb))) return rc;
        dst_pcm = sl.h_pcm;
        sl.user_pcm = pcm_out;
        sl.user_pcm_bytes = pb;
    }
    HIP_TRY(e, hipMemcpyAsync(sl.d_units, h.units.data(), ub, hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    if (rb) HIP_TRY(e, hipMemcpyAsync(sl.d_runs, h.runs.data(), rb, hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    if (rvs[0]) HIP_TRY(e, hipMemcpyAsync(sl.d_rv[0], h.runs_rv.data(), rvs[0], hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    if (rvs[1]) HIP_TRY(e, hipMemcpyAsync(sl.d_rv[1], h.links_rv.data(), rvs[1], hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    if (tb) {
        HIP_TRY(e, hipMemcpyAsync(sl.d_tns, h.tns.data(), sizeof(aacg_dev_tns) * h.tns.size(), hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
        aacg_tns_matrices_launch((const aacg_dev_tns*)sl.d_tns, tns_matrices_of((const aacg_dev_tns*)sl.d_tns, h.tns.size()), (uint32_t)h.tns.size(), s);
    }
    for (int i = 0; i < 3; i++) if (ccb[i]) HIP_TRY(e, hipMemcpyAsync(sl.d_cce[i], cce_src[i], ccb[i], hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    HIP_TRY(e, hipMemcpyAsync(sl.d_coeffs, src_coeffs, cb, hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    if (quant) HIP_TRY(e, hipMemcpyAsync(sl.d_meta, src_meta, mb, hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    /* kernels chain through the overlap state: this one starts after the previous batch's kernel,
     * while its uploads above overlapped it */
    if (e->last_kernel) HIP_TRY(e, hipStreamWaitEvent(s, e->last_kernel, 0), AACG_ERR_NO_DEVICE);
    if ((rc = pipe_join(e, s))) return rc;              /* ... and after whatever aacg_decode_pipelined has in flight */
    const xl_args serial = {false, 0ull, 0};
    rc = launch_run(e, R, (const aacg_dev_unit*)sl.d_units, (const aacg_run*)sl.d_runs, (const aacg_dev_tns*)sl.d_tns,
                    (float*)sl.d_scratch, (float*)sl.d_spec, cce_bufs{(const aacg_run*)sl.d_cce[0], (const aacg_couple_job*)sl.d_cce[1], (const float*)sl.d_cce[2], (float*)sl.d_cce[3]},
                    rv_bufs{(const aacg_run*)sl.d_rv[0], (const aacg_rv_link*)sl.d_rv[1], (unsigned long long*)sl.d_rv[2], (float*)sl.d_rv[3]}, h, sl.d_coeffs,
                    (const aacg_band_meta*)sl.d_meta, sl.d_pcm, 0, s, serial, nullptr);
    if (rc) return rc;
    HIP_TRY(e, hipEventRecord(sl.kernel_done, s), AACG_ERR_NO_DEVICE);
    e->last_kernel = sl.kernel_done;
    HIP_TRY(e, hipMemcpyAsync(dst_pcm, sl.d_pcm, pb, hipMemcpyDeviceToHost, s), AACG_ERR_NO_DEVICE);
    HIP_TRY(e, hipEventRecord(sl.done, s), AACG_ERR_NO_DEVICE);
    sl.busy = true;

    for (const aacg_chain& c : h.chains)
        for (int k = 0; k < c.n_ch; k++)
            { uint8_t& b = e->parity[(size_t)c.stream * (size_t)e->cfg.max_channels + c.channel + k]; b = (uint8_t)((b + 1) % AACG_OV_BUFFERS); }
    e->epoch++;
    *ticket = ++e->submitted;
    return AACG_OK;
}

/* Synchronous form: process(elements) + interleave for a whole batch. */
int aacg_decode_batch(aacg_engine* e, const aacg_unit_desc* units, uint32_t n_units,
                      const void* coeffs, uint32_t n_coef_blocks,
                      const aacg_band_meta* meta, uint32_t n_meta,
                      void* pcm_out, size_t n_pcm_floats)
{
    return aacg_decode_batch_tns(e, units, n_units, coeffs, n_coef_blocks, meta, n_meta, nullptr, 0, pcm_out, n_pcm_floats);
}

int aacg_decode_batch_ex(aacg_engine* e, const aacg_batch* b)
{
    uint64_t t = 0;
    int rc = aacg_submit_ex(e, b, &t);
    if (rc) return rc;
    return aacg_wait(e, t);
}

int aacg_decode_batch_tns(aacg_engine* e, const aacg_unit_desc* units, uint32_t n_units,
                          const void* coeffs, uint32_t n_coef_blocks,
                          const aacg_band_meta* meta, uint32_t n_meta,
                          const aacg_tns_info* tns, uint32_t n_tns,
                          void* pcm_out, size_t n_pcm_floats)
{
    uint64_t t = 0;
    int rc = aacg_submit_tns(e, units, n_units, coeffs, n_coef_blocks, meta, n_meta, tns, n_tns, pcm_out, n_pcm_floats, &t);
    if (rc) return rc;
    return aacg_wait(e, t);
}

}  // extern "C"
