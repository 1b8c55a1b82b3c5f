/*
 * aacg_engine.hip — the C ABI of include/aacgpu.h on top of the gfx950 kernels.
 *
 * Replaces AACDecoder.prototype.process + the interleave of readChunk (reference
 * src/decoder.js:201-215, 218-334) for batches of frames from many streams.  There is no
 * CPU fallback: every entry point either runs the HIP kernels or returns an error.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <algorithm>
#include <vector>

#include "aacg_kernels.h"
#include "aacg_host.h"

/* ---- kernels ----------------------------------------------------------------------- */
/* 1024 threads = 16 waves, one workgroup per CU: 4 waves per SIMD -> 128 VGPRs per lane */
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant(const aacg_kparams P) { imdct_run_body<AACG_INPUT_QUANT_I16>(P); }

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32(const aacg_kparams P) { imdct_run_body<AACG_INPUT_SPEC_F32>(P); }

/* The variant for plans with full later runs (_dd: their first wave does double duty) lives in aacg_engine_ext.hip,
 * the optional TNS / PNS stages in aacg_engine_spectral.hip: their own code objects, so that adding to them never
 * moves the two kernels above. */
void aacg_ext_launch(bool quant, dim3 grid, dim3 block, hipStream_t s, const aacg_kparams& P);
/* aacg_engine_spectral.hip: the optional stages (AACG_PNS_SPEC noise bands, AACG_TNS_SPEC filters) -> f32 spectra */
void aacg_refresh_launch(aacg_dev_unit* units, const aacg_unit_desc* parsed, const aacg_parse_result* results, uint32_t n_units,
                         uint32_t max_units, int refuse_pns, uint32_t* refused, hipStream_t s);
/* aacg_engine_i16.hip: the run kernels with int16 PCM stores (AACG_OUTPUT_I16 engines) */
void aacg_i16_launch(bool quant, bool dd, bool wide, dim3 grid, dim3 block, hipStream_t s, const aacg_kparams& P);
/* aacg_engine_exrun.hip: the run kernels with the optional stages inside (one launch for TNS / PNS batches) */
void aacg_exrun_launch(bool quant, dim3 grid, dim3 block, hipStream_t s, const aacg_kparams& P);
int aacg_spectral_ex_set_lds_limits(void);
void aacg_spectral_ex_launch(bool quant, int n_units, hipStream_t s, const aacg_kparams& P);
/* aacg_engine_couple.hip: AACG_CCE_SPEC */
void aacg_couple_launch(bool pcm, hipStream_t s, const aacg_couple_params& Q);
void aacg_couple_run_launch(bool quant, bool wide, dim3 grid, dim3 block, hipStream_t s, const aacg_kparams& P);
struct cce_bufs { const aacg_run* runs; const aacg_couple_job* jobs; const float* gains; float* side; };
/* aacg_engine8.hip: the one-channel-per-wave run kernels (two workgroups per CU) */
void aacg_run8_launch(bool quant, dim3 grid, dim3 block, hipStream_t s, const aacg_kparams8& P);
struct run8_bufs { const aacg_run8* runs; unsigned long long* rv_state; float* rv_data; };
/* aacg_engine_nt.hip: the plain run kernels for multichannel batches (non-temporal loads of the spectra) */
void aacg_nt_launch(bool quant, dim3 grid, dim3 block, hipStream_t s, const aacg_kparams& P);
/* aacg_engine_rv.hip: the 16-wave kernels for chains longer than a run, with a run-to-run rendezvous instead of a recomputed frame */
void aacg_rv_launch(bool quant, bool wide, dim3 grid, dim3 block, hipStream_t s, const aacg_kparams& P, const aacg_rv_args& V);
struct rv_bufs { const aacg_run* runs; const aacg_rv_link* links; unsigned long long* state; float* data; };

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_spectral(const aacg_kparams P, int n_units) { spectral_body(P, n_units); }

#define AACG_LDS_BYTES_SPECTRAL ((AACG_TAB_QUANT_FLOATS + AACG_WG_WAVES * 512) * 4)

/* ---- engine ------------------------------------------------------------------------ */
struct aacg_engine {
    aacg_config cfg;
    hipStream_t stream = nullptr;
    aacg_tables* d_tab = nullptr;
    aacg_pns_tables* d_pns = nullptr;       /* AACG_PNS_SPEC */
    aacg_win8* d_win8 = nullptr;            /* the windows as the 8-wave kernels read them */
    bool run8 = false;                      /* plain batches (f32 PCM, no optional stage) on the one-channel-per-wave kernels: opt-in
                                               (AACG_RUN8=1 / AACG_DEBUG_ROUTE_NARROW_KERNELS) — built to parity and measured slower than the
                                               16-wave kernels on every BASELINE configuration */
    unsigned long long rv_epoch = 0;        /* rendezvous epoch: one per launch of those kernels, never 0 */
    bool rv = true;                         /* chains longer than a run: rendezvous between their runs (the _rv kernels) instead of a recomputed frame (_dd) */
    float* d_overlap = nullptr;             /* [max_streams][max_channels][2][1024] */
    std::vector<uint8_t> parity;            /* live buffer per (stream, channel) */
    uint64_t epoch = 0;                     /* bumped whenever `parity` changes: lets a relaunched plan skip its check */
    aacg_tables h_tab;
    aacg_host_windows h_win;
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
        void* d_run8[3] = {nullptr, nullptr, nullptr}; size_t run8_cap[3] = {0, 0, 0};             /* 8-wave kernels: runs, rendezvous state, payload */
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
    std::string err;
};

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
    void*  d_run8[3] = {nullptr, nullptr, nullptr};           /* 8-wave kernels: their run table, rendezvous state words and payload */
    void*  d_rv[4] = {nullptr, nullptr, nullptr, nullptr};    /* _rv kernels: run table, link records, rendezvous state words and payload */
    size_t bytes[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  /* sizes of the sixteen buffers above, for the engine's free list */
    hipEvent_t uploaded = nullptr;          /* the tables are on the device */
    hipEvent_t last_use = nullptr;          /* recorded at destruction on last_stream: everything launched with this plan */
    hipStream_t last_stream = nullptr;      /* stream of the most recent launch (no per-launch event: it costs 3 us per step) */
    bool used = false;
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

/* does launch_run stage f32 spectra in HBM for this batch (the optional stages as a launch of their own, or coupling
 * in the spectral domain)?  Batches whose optional stages run inside the run kernel need no such buffer. */
bool needs_spec_buffer(const aacg_engine* e, const aacg_plan_host& h)
{
    const bool quant = e->cfg.input_kind == AACG_INPUT_QUANT_I16, i16 = e->cfg.output_kind == AACG_OUTPUT_I16;
    const bool stages = h.any_tns || (quant && h.any_pns);
    return h.any_cce_dependent || (stages && (i16 || h.any_cce || h.needs_scratch));
}

/* Plain batches — f32 PCM, no optional stage, no coupling element — run on the one-channel-per-wave kernels (aacg_kernels8.h). */
bool takes_run8(const aacg_engine* e, const aacg_plan_host& h)
{
    const bool quant = e->cfg.input_kind == AACG_INPUT_QUANT_I16;
    const bool on = (e->run8 || (e->debug_route & AACG_DEBUG_ROUTE_NARROW_KERNELS)) && !(e->debug_route & AACG_DEBUG_ROUTE_WIDE_KERNELS);
    return on && e->cfg.output_kind == AACG_OUTPUT_F32 &&
           !h.any_cce && !h.any_tns && !(quant && h.any_pns) && !h.runs8.empty();
}

/* Plain batches with a chain longer than a run: the 16-wave kernels with a run-to-run rendezvous (no recomputed frame). */
bool takes_rv(const aacg_engine* e, const aacg_plan_host& h)
{
    const bool quant = e->cfg.input_kind == AACG_INPUT_QUANT_I16;
    return e->rv && !takes_run8(e, h) && e->cfg.output_kind == AACG_OUTPUT_F32 && !h.any_cce && !h.any_tns && !(quant && h.any_pns) && !h.runs_rv.empty();
}

/* The launches launch_run makes for a plan, by kernel name: what a rocprofv3 kernel trace of the batch shows. */
std::string route_names(const aacg_engine* e, const aacg_plan_host& h)
{
    const bool i16 = e->cfg.output_kind == AACG_OUTPUT_I16;
    bool quant = e->cfg.input_kind == AACG_INPUT_QUANT_I16, ex = false;
    const bool stages = h.any_tns || (quant && h.any_pns);
    std::string r;
    auto add = [&](const std::string& k) { if (!r.empty()) r += " + "; r += k; };
    if (h.any_cce_dependent) {
        add(quant ? "aacg_spectral_ex_quant" : "copy");
        add("aacg_couple_spec");
        if (h.any_tns) add("aacg_spectral_ex_f32");
        quant = false;
    } else if (stages && !i16 && !h.any_cce && !h.needs_scratch) {
        ex = true;
    } else if (stages) {
        add(quant ? "aacg_spectral_ex_quant" : "aacg_spectral_ex_f32");
        quant = false;
    }
    if (takes_run8(e, h)) return std::string("aacg_imdct_run8_") + (quant ? "quant" : "f32");
    if (takes_rv(e, h)) return std::string("aacg_imdct_run_") + (quant ? "quant" : "f32") + "_rv" + (h.wide_frames ? "_nt" : "");
    const std::string run = std::string("aacg_imdct_run_") + (quant ? "quant" : "f32");
    const bool fused = h.fused_independent && !ex && !i16 && !(e->debug_route & AACG_DEBUG_ROUTE_UNFUSED_COUPLING);
    if (fused) {
        if (!h.cce_runs.empty()) add(run + " (coupling elements)");
        if (!h.runs.empty()) add(run + "_cpl" + (h.wide_frames ? "_nt" : ""));
        return r;
    }
    const bool nt = !ex && !h.needs_scratch && h.wide_frames && !e->d_trace;
    if (!h.runs.empty()) add(run + (ex ? "_ex" : "") + (!ex && h.needs_scratch ? "_dd" : "") + (!ex && i16 ? "_i16" : "") + (nt ? "_nt" : ""));
    if (h.any_cce) {
        if (!h.cce_runs.empty()) add(run + " (coupling elements)");
        add("aacg_couple_pcm");
    }
    return r;
}

/* enqueue the run kernel for a planned batch (device pointers) */
int launch_run(aacg_engine* e, const aacg_dev_unit* d_units, const aacg_run* d_runs, const aacg_dev_tns* d_tns,
               float* d_scratch, float* d_spec, const cce_bufs& cb, const run8_bufs& r8, const rv_bufs& rvb, const aacg_plan_host& h, const void* d_coeffs, const aacg_band_meta* d_meta,
               void* d_pcm, int flip, hipStream_t s)
{
    const bool i16 = e->cfg.output_kind == AACG_OUTPUT_I16;
    bool quant = e->cfg.input_kind == AACG_INPUT_QUANT_I16, ex = false;
    if (h.zero_fill)
        HIP_TRY(e, hipMemsetAsync(d_pcm, 0, h.pcm_floats * pcm_elem_size(e), s), AACG_ERR_NO_DEVICE);
    if (takes_run8(e, h)) {
        aacg_kparams8 P8;
        P8.units = d_units; P8.runs = r8.runs; P8.coeffs = d_coeffs; P8.meta = d_meta; P8.pcm = (float*)d_pcm; P8.overlap = e->d_overlap;
        P8.tab = e->d_tab; P8.win = e->d_win8; P8.rv_state = r8.rv_state; P8.rv_data = r8.rv_data; P8.epoch = ++e->rv_epoch;
        P8.flip = flip; P8.n_runs = (int32_t)h.runs8.size();
        P8.trace = (e->d_trace && (e->ablate & 16)) ? (unsigned long long*)e->d_trace : nullptr;
        aacg_run8_launch(quant, dim3((unsigned)h.runs8.size()), dim3(AACG_WG_THREADS), s, P8);
        HIP_TRY(e, hipGetLastError(), AACG_ERR_NO_DEVICE);
        return AACG_OK;
    }
    if (takes_rv(e, h)) {
        aacg_kparams P;
        std::memset(&P, 0, sizeof P);
        P.units = d_units; P.runs = rvb.runs; P.coeffs = d_coeffs; P.meta = d_meta; P.pcm = (float*)d_pcm;
        P.overlap = e->d_overlap; P.tab = e->d_tab; P.flip = flip; P.n_runs = (int32_t)h.runs_rv.size();
        aacg_rv_args V;
        V.links = rvb.links; V.state = rvb.state; V.data = rvb.data; V.epoch = ++e->rv_epoch;
        aacg_rv_launch(quant, h.wide_frames, dim3((unsigned)h.runs_rv.size()), dim3(AACG_WG_THREADS), s, P, V);
        HIP_TRY(e, hipGetLastError(), AACG_ERR_NO_DEVICE);
        return AACG_OK;
    }
    aacg_kparams P;
    P.units = d_units; P.runs = d_runs; P.coeffs = d_coeffs; P.meta = d_meta; P.pcm = (float*)d_pcm;
    P.tns = h.any_tns ? d_tns : nullptr;
    P.scratch = h.needs_scratch ? d_scratch : nullptr;
    P.pns = nullptr;
    P.overlap = e->d_overlap; P.spec_out = nullptr; P.tab = e->d_tab;
    P.flip = flip; P.n_runs = (int32_t)h.runs.size();
    P.ablate = e->d_trace ? e->ablate : (e->ablate & ~16);
    if (e->d_trace) P.spec_out = (float*)e->d_trace;
    const dim3 grid((unsigned)h.runs.size()), block(AACG_WG_THREADS);
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
    if (h.any_cce_dependent) {
        /* AACG_CCE_SPEC with coupling in the spectral domain: every unit's spectrum (the coupling elements' too) as f32,
         * then decoder.js:258-266 / 304-316 in stages: coupling before TNS, the TNS filters, coupling after TNS — each its
         * own small launch, in place.  (Independent coupling alone leaves the spectral route as it is.) */
        float* trace_or_null = P.spec_out;
        P.spec_out = d_spec; P.pns = e->d_pns; P.tns = nullptr;
        if (quant) aacg_spectral_ex_launch(true, (int)h.units.size(), s, P);
        else HIP_TRY(e, hipMemcpyAsync(d_spec, d_coeffs, (size_t)h.coef_blocks * 4096u, hipMemcpyDeviceToDevice, s), AACG_ERR_NO_DEVICE);
        couple(AACG_CCE_BEFORE_TNS);
        if (h.any_tns) {
            P.coeffs = d_spec; P.meta = nullptr; P.tns = d_tns;
            aacg_spectral_ex_launch(false, (int)h.units.size(), s, P);
        }
        couple(AACG_CCE_AFTER_TNS);
        P.spec_out = trace_or_null; P.coeffs = d_spec; P.meta = nullptr; P.tns = nullptr;
        quant = false;
    } else if ((h.any_tns || (quant && h.any_pns)) && !i16 && !h.any_cce && !h.needs_scratch) {
        /* optional stages (noise bands, TNS filters) inside the run kernel: one launch */
        P.pns = e->d_pns;
        ex = true;
    } else if (h.any_tns || (quant && h.any_pns)) {
        /* int16 PCM, coupling elements or double-duty runs: the optional stages first, as a launch of their own that leaves f32 spectra,
         * which the f32 run kernel takes from there */
        float* trace_or_null = P.spec_out;
        P.spec_out = d_spec; P.pns = e->d_pns;
        aacg_spectral_ex_launch(quant, (int)h.units.size(), s, P);
        P.spec_out = trace_or_null; P.coeffs = d_spec; P.meta = nullptr; P.tns = nullptr;
        quant = false;
    }
    auto cce_filterbank = [&]() {                       /* the independently switched coupling elements' own filterbank pass */
        aacg_kparams C = P;
        C.runs = cb.runs; C.n_runs = (int32_t)h.cce_runs.size(); C.pcm = cb.side; C.scratch = nullptr;
        if (quant) hipLaunchKernelGGL(aacg_imdct_run_quant, dim3((unsigned)h.cce_runs.size()), block, 0, s, C);
        else       hipLaunchKernelGGL(aacg_imdct_run_f32, dim3((unsigned)h.cce_runs.size()), block, 0, s, C);
    };
    const bool fused = h.fused_independent && !ex && !i16 && !(e->debug_route & AACG_DEBUG_ROUTE_UNFUSED_COUPLING);
    if (fused) {
        /* independent coupling in the targets' epilogues: the coupling elements go first (into the side buffer), then the
         * run kernel that adds gain * side where it forms the PCM — no read-modify-write pass over the interleaved PCM */
        if (!h.cce_runs.empty()) cce_filterbank();
        aacg_set_cpl(&P, cb.jobs + h.fused_first, cb.gains, cb.side);
        if (!h.runs.empty()) aacg_couple_run_launch(quant, h.wide_frames, grid, block, s, P);
    } else if (!h.runs.empty()) {
        if (ex) {
            aacg_exrun_launch(quant, grid, block, s, P);
        } else if (i16) {
            aacg_i16_launch(quant, h.needs_scratch, h.wide_frames && !e->d_trace, grid, block, s, P);
        } else if (h.needs_scratch) {
            aacg_ext_launch(quant, grid, block, s, P);
        } else if (h.wide_frames && !e->d_trace) {
            aacg_nt_launch(quant, grid, block, s, P);
        } else {
            if (quant) hipLaunchKernelGGL(aacg_imdct_run_quant, grid, block, 0, s, P);
            else       hipLaunchKernelGGL(aacg_imdct_run_f32, grid, block, 0, s, P);
        }
    }
    if (h.any_cce && !fused) {
        if (!h.cce_runs.empty()) cce_filterbank();
        couple(AACG_CCE_AFTER_IMDCT);
    }
    HIP_TRY(e, hipGetLastError(), AACG_ERR_NO_DEVICE);
    return AACG_OK;
}

}  // namespace

extern "C" {

int aacg_abi_version(void) { return AACG_ABI_VERSION; }

const char* aacg_kernel_name(void) { return "aacg_imdct_run_quant"; }

int aacg_debug_set_route(aacg_engine* e, int flags)
{
    if (!e || (flags & ~(AACG_DEBUG_ROUTE_UNFUSED_COUPLING | AACG_DEBUG_ROUTE_WIDE_KERNELS | AACG_DEBUG_ROUTE_NARROW_KERNELS | AACG_DEBUG_ROUTE_RECOMPUTE))) return AACG_ERR_INVALID_ARG;
    e->debug_route = flags;
    if (flags & AACG_DEBUG_ROUTE_RECOMPUTE) e->rv = false;
    return AACG_OK;
}

int aacg_plan_kernels(aacg_engine* e, const aacg_plan* p, char* dst, size_t n)
{
    if (!e || !p || p->e != e || !dst || n == 0) return AACG_ERR_INVALID_ARG;
    const std::string r = route_names(e, p->h);
    if (r.size() + 1 > n) return AACG_ERR_INVALID_ARG;
    std::memcpy(dst, r.c_str(), r.size() + 1);
    return AACG_OK;
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

    const size_t ov_bytes = (size_t)cfg->max_streams * (size_t)cfg->max_channels * 2u * 1024u * sizeof(float);
    if (!hip_ok(e, hipSetDevice(cfg->device_ordinal), "hipSetDevice") ||
        !hip_ok(e, hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking), "hipStreamCreate") ||
        !hip_ok(e, hipMalloc((void**)&e->d_tab, sizeof(aacg_tables)), "hipMalloc tables") ||
        !hip_ok(e, hipMalloc((void**)&e->d_overlap, ov_bytes), "hipMalloc overlap") ||
        !hip_ok(e, hipMemcpy(e->d_tab, &e->h_tab, sizeof(aacg_tables), hipMemcpyHostToDevice), "upload tables") ||
        !hip_ok(e, hipMemset(e->d_overlap, 0, ov_bytes), "zero overlap") ||
        /* (the run kernels' ~152 KiB of LDS per workgroup are static allocations: dp_lds_fixed) */
        aacg_spectral_ex_set_lds_limits() != 0 ||
        !hip_ok(e, hipFuncSetAttribute((const void*)aacg_spectral, hipFuncAttributeMaxDynamicSharedMemorySize, AACG_LDS_BYTES_SPECTRAL), "LDS attr")) {
        std::fprintf(stderr, "aacgpu: %s\n", e->err.c_str());
        aacg_destroy(e);
        return AACG_ERR_NO_DEVICE;
    }
    e->parity.assign((size_t)cfg->max_streams * (size_t)cfg->max_channels, 0);
    {
        aacg_win8* w8 = new (std::nothrow) aacg_win8;
        if (w8) aacg_build_win8(&e->h_tab, w8);
        const bool ok = w8 && hip_ok(e, hipMalloc((void**)&e->d_win8, sizeof *w8), "hipMalloc window tables") &&
                        hip_ok(e, hipMemcpy(e->d_win8, w8, sizeof *w8, hipMemcpyHostToDevice), "upload window tables");
        delete w8;
        if (!ok) { aacg_destroy(e); return AACG_ERR_OUT_OF_MEMORY; }
        if (const char* r = std::getenv("AACG_RUN8")) e->run8 = std::atoi(r) != 0;     /* A/B switch for tools/: 1 = plain batches on the one-channel-per-wave kernels */
        if (const char* r = std::getenv("AACG_RV")) e->rv = std::atoi(r) != 0;         /* A/B switch for tools/: 0 = long chains recompute a frame per later run (_dd kernels) */
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
    if (!hip_ok(e, hipDeviceSynchronize(), "synchronize after setup")) { aacg_destroy(e); return AACG_ERR_NO_DEVICE; }
    *out = e;
    return AACG_OK;
}

void aacg_destroy(aacg_engine* e)
{
    if (!e) return;
    (void)hipSetDevice(e->cfg.device_ordinal);
    (void)hipDeviceSynchronize();
    if (e->d_tab) (void)hipFree(e->d_tab);
    if (e->d_pns) (void)hipFree(e->d_pns);
    if (e->d_win8) (void)hipFree(e->d_win8);
    if (e->d_overlap) (void)hipFree(e->d_overlap);
    for (auto& sl : e->slot) {
        for (void* p : {sl.d_units, sl.d_runs, sl.d_coeffs, sl.d_meta, sl.d_tns, sl.d_scratch, sl.d_spec, sl.d_pcm}) if (p) (void)hipFree(p);
        for (void* p : sl.d_cce) if (p) (void)hipFree(p);
        for (void* p : sl.d_run8) if (p) (void)hipFree(p);
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
    HIP_TRY(e, hipDeviceSynchronize(), AACG_ERR_NO_DEVICE);
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
        HIP_TRY(e, hipMemset(ov_ptr(e, stream, (uint32_t)c), 0, 4096), AACG_ERR_NO_DEVICE);
    HIP_TRY(e, hipDeviceSynchronize(), AACG_ERR_NO_DEVICE);       /* null-stream memsets: a launch on a non-blocking stream must not overtake them */
    return AACG_OK;
}

int aacg_get_overlap(aacg_engine* e, uint32_t stream, uint32_t channel, float* dst)
{
    int rc = ov_check(e, stream, channel);
    if (rc) return rc;
    if (!dst) return AACG_ERR_INVALID_ARG;
    HIP_TRY(e, hipMemcpy(dst, ov_ptr(e, stream, channel), 4096, hipMemcpyDeviceToHost), AACG_ERR_NO_DEVICE);
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
    float scaled[1024];
    for (int i = 0; i < 1024; i++) scaled[i] = src[i] * AACG_PCM_SCALE;
    HIP_TRY(e, hipMemcpy(ov_ptr(e, stream, channel), scaled, 4096, hipMemcpyHostToDevice), AACG_ERR_NO_DEVICE);
    return AACG_OK;
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
    const size_t tb = sizeof(aacg_dev_tns) * p->h.tns.size();
    const size_t sb = p->h.needs_scratch ? p->h.runs.size() * AACG_SLOT_FLOATS * sizeof(float) : 0;
    const size_t cb[4] = {sizeof(aacg_run) * p->h.cce_runs.size(), sizeof(aacg_couple_job) * p->h.couple_jobs.size(),
                          sizeof(float) * p->h.gains.size(), (size_t)p->h.side_blocks * 4096u};
    bool ok = hip_ok(e, hipSetDevice(e->cfg.device_ordinal), "hipSetDevice") &&
              hip_ok(e, hipEventCreateWithFlags(&p->uploaded, hipEventDisableTiming), "hipEventCreate") &&
              hip_ok(e, hipEventCreateWithFlags(&p->last_use, hipEventDisableTiming), "hipEventCreate");
    const bool r8 = takes_run8(e, p->h), rvp = takes_rv(e, p->h);
    const size_t r8b[3] = {r8 ? sizeof(aacg_run8) * p->h.runs8.size() : 0, r8 ? sizeof(unsigned long long) * AACG8_RV_STATE_WORDS * (size_t)p->h.n_links : 0,
                           r8 ? sizeof(float) * AACG8_RV_DATA_FLOATS * (size_t)p->h.n_links : 0};
    const size_t rvs[4] = {rvp ? sizeof(aacg_run) * p->h.runs_rv.size() : 0, rvp ? sizeof(aacg_rv_link) * p->h.links_rv.size() : 0,
                           rvp ? sizeof(unsigned long long) * AACG8_RV_STATE_WORDS * (size_t)p->h.n_links_rv : 0,
                           rvp ? sizeof(float) * AACG8_RV_DATA_FLOATS * (size_t)p->h.n_links_rv : 0};
    const bool other = r8 || rvp;                        /* the plan's launches never read the 16-wave run table / scratch */
    const size_t want[16] = {ub, other ? 0 : rb, tb, other ? 0 : sb, xb, cb[0], cb[1], cb[2], cb[3], r8b[0], r8b[1], r8b[2], rvs[0], rvs[1], rvs[2], rvs[3]};
    void** const slot[16] = {(void**)&p->d_units, (void**)&p->d_runs, (void**)&p->d_tns, (void**)&p->d_scratch, (void**)&p->d_spec,
                             &p->d_cce[0], &p->d_cce[1], &p->d_cce[2], &p->d_cce[3], &p->d_run8[0], &p->d_run8[1], &p->d_run8[2],
                             &p->d_rv[0], &p->d_rv[1], &p->d_rv[2], &p->d_rv[3]};
    const void* const src[16] = {p->h.units.data(), p->h.runs.data(), p->h.tns.data(), nullptr, nullptr,
                                 p->h.cce_runs.data(), p->h.couple_jobs.data(), p->h.gains.data(), nullptr, p->h.runs8.data(), nullptr, nullptr,
                                 p->h.runs_rv.data(), p->h.links_rv.data(), nullptr, nullptr};
    for (int i = 0; i < 16 && ok; i++) {
        if (!want[i]) continue;
        *slot[i] = pool_take(e, want[i], &p->bytes[i]);
        ok = *slot[i] != nullptr &&
             (!src[i] || hip_ok(e, hipMemcpyAsync(*slot[i], src[i], want[i], hipMemcpyHostToDevice, e->stream), "upload plan tables"));
        /* rendezvous state words count only with a launch's epoch in them; a recycled or fresh buffer starts from zero all the same */
        if (ok && (i == 10 || i == 14)) ok = hip_ok(e, hipMemsetAsync(*slot[i], 0, want[i], e->stream), "zero rendezvous state");
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
    if (p->uploaded) { (void)hipEventSynchronize(p->uploaded); (void)hipEventDestroy(p->uploaded); }
    if (p->last_use) {
        if (p->used) {
            if (hipEventRecord(p->last_use, p->last_stream) == hipSuccess) (void)hipEventSynchronize(p->last_use);
            else { (void)hipGetLastError(); (void)hipDeviceSynchronize(); }        /* the stream is gone: wait for everything */
        }
        (void)hipEventDestroy(p->last_use);
    }
    void* const ptr[16] = {p->d_units, p->d_runs, p->d_tns, p->d_scratch, p->d_spec, p->d_cce[0], p->d_cce[1], p->d_cce[2], p->d_cce[3],
                           p->d_run8[0], p->d_run8[1], p->d_run8[2], p->d_rv[0], p->d_rv[1], p->d_rv[2], p->d_rv[3]};
    for (int i = 0; i < 16; i++) pool_give(e, ptr[i], p->bytes[i]);
    delete p;
}

static int plan_check_parity(aacg_engine* e, const aacg_plan* p)
{
    const int flip = (int)(p->launches & 1u);
    for (const aacg_chain& c : p->h.chains)
        for (int k = 0; k < c.n_ch; k++)
            if (e->parity[(size_t)c.stream * (size_t)e->cfg.max_channels + c.channel + k] != (c.parity[k] ^ flip)) {
                e->err = "plan is stale: another plan advanced one of its streams";
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
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    if (!p->used) HIP_TRY(e, hipStreamWaitEvent(s, p->uploaded, 0), AACG_ERR_NO_DEVICE);
    /* the plan's previous launch ran on ANOTHER stream: this one continues its overlap state, reuses its scratch areas and
     * rendezvous cells, so it is ordered behind it on the device (an event only when the stream changes: per launch it
     * would cost 3 us) */
    if (p->used && p->last_stream != s) {
        HIP_TRY(e, hipEventRecord(p->last_use, p->last_stream), AACG_ERR_NO_DEVICE);
        HIP_TRY(e, hipStreamWaitEvent(s, p->last_use, 0), AACG_ERR_NO_DEVICE);
    }
    const cce_bufs cb = {(const aacg_run*)p->d_cce[0], (const aacg_couple_job*)p->d_cce[1], (const float*)p->d_cce[2], (float*)p->d_cce[3]};
    const run8_bufs r8 = {(const aacg_run8*)p->d_run8[0], (unsigned long long*)p->d_run8[1], (float*)p->d_run8[2]};
    if (takes_run8(e, p->h) && !p->d_run8[0]) { e->err = "the plan was made for the 16-wave kernels (aacg_debug_set_route changed since)"; return AACG_ERR_STALE_PLAN; }
    const rv_bufs rvb = {(const aacg_run*)p->d_rv[0], (const aacg_rv_link*)p->d_rv[1], (unsigned long long*)p->d_rv[2], (float*)p->d_rv[3]};
    if (takes_rv(e, p->h) && !p->d_rv[0]) { e->err = "the plan was made for another route (aacg_debug_set_route changed since)"; return AACG_ERR_STALE_PLAN; }
    if (!takes_run8(e, p->h) && !takes_rv(e, p->h) && !p->d_runs && !p->h.runs.empty()) { e->err = "the plan was made for another route (aacg_debug_set_route changed since)"; return AACG_ERR_STALE_PLAN; }
    rc = launch_run(e, p->d_units, p->d_runs, p->d_tns, p->d_scratch, p->d_spec, cb, r8, rvb, p->h, d_coeffs, d_meta, d_pcm, (int)(p->launches & 1u), s);
    if (rc) return rc;
    p->last_stream = s;
    p->used = true;

    for (const aacg_chain& c : p->h.chains)
        for (int k = 0; k < c.n_ch; k++)
            e->parity[(size_t)c.stream * (size_t)e->cfg.max_channels + c.channel + k] ^= 1;
    p->launches++;
    p->seen_epoch = ++e->epoch;
    return AACG_OK;
}

/* The plan's device unit records take what the parser found (device to device); the run tables stay. */
int aacg_plan_refresh_from_parse(aacg_engine* e, aacg_plan* p, const aacg_unit_desc* d_parsed_units,
                                 const aacg_parse_result* d_results, uint32_t max_units, uint32_t* d_refused, void* hip_stream)
{
    if (!e || !p || p->e != e || !d_parsed_units || !d_results || !max_units || !d_refused) return AACG_ERR_INVALID_ARG;
    if (e->cfg.input_kind != AACG_INPUT_QUANT_I16) { e->err = "aacg_plan_refresh_from_parse needs a QUANT_I16 engine"; return AACG_ERR_INVALID_ARG; }
    if (p->h.any_tns || p->h.any_pns || e->cfg.tns_mode == AACG_TNS_SPEC) {
        e->err = "aacg_plan_refresh_from_parse: TNS records / noise tables are prepared on the host, such plans are rebuilt per batch";
        return AACG_ERR_UNSUPPORTED;
    }
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    HIP_TRY(e, hipStreamWaitEvent(s, p->uploaded, 0), AACG_ERR_NO_DEVICE);
    if (p->used && p->last_stream != s) {              /* the records' readers on the plan's previous stream first */
        HIP_TRY(e, hipEventRecord(p->last_use, p->last_stream), AACG_ERR_NO_DEVICE);
        HIP_TRY(e, hipStreamWaitEvent(s, p->last_use, 0), AACG_ERR_NO_DEVICE);
    }
    aacg_refresh_launch(p->d_units, d_parsed_units, d_results, p->n_units, max_units, 1, d_refused, s);
    HIP_TRY(e, hipGetLastError(), AACG_ERR_NO_DEVICE);
    p->last_stream = s;
    p->used = true;
    return AACG_OK;
}

/* The plan's unit records rewritten from the host's next batch of the same structure; the run tables stay. */
int aacg_plan_refresh_units(aacg_engine* e, aacg_plan* p, const aacg_unit_desc* units, uint32_t n_units, void* hip_stream)
{
    if (!e || !p || p->e != e || !units) return AACG_ERR_INVALID_ARG;
    int rc = aacg_plan_refresh_host(&p->h, units, n_units, e->cfg.sample_index, e->cfg.tns_mode == AACG_TNS_SPEC, &e->err);
    if (rc) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    HIP_TRY(e, hipStreamWaitEvent(s, p->uploaded, 0), AACG_ERR_NO_DEVICE);
    /* the plan's previous launch on ANOTHER stream may still be reading the records this copy overwrites: order behind it */
    if (p->used && p->last_stream != s) {
        HIP_TRY(e, hipEventRecord(p->last_use, p->last_stream), AACG_ERR_NO_DEVICE);
        HIP_TRY(e, hipStreamWaitEvent(s, p->last_use, 0), AACG_ERR_NO_DEVICE);
    }
    /* pageable source: the runtime stages it before returning, so the host copy may change again right away; in stream
     * order behind the launches that read the previous records */
    HIP_TRY(e, hipMemcpyAsync(p->d_units, p->h.units.data(), sizeof(aacg_dev_unit) * p->h.units.size(), hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    p->last_stream = s;
    p->used = true;
    return AACG_OK;
}

int aacg_spectral_device(aacg_engine* e, aacg_plan* p, const void* d_coeffs, const aacg_band_meta* d_meta,
                         float* d_spec_out, void* hip_stream)
{
    if (!e || !p || p->e != e || !d_coeffs || !d_meta || !d_spec_out) return AACG_ERR_INVALID_ARG;
    if (e->cfg.input_kind != AACG_INPUT_QUANT_I16) { e->err = "spectral stage needs a QUANT_I16 engine"; return AACG_ERR_INVALID_ARG; }
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    aacg_kparams P;
    std::memset(&P, 0, sizeof P);
    P.units = p->d_units; P.coeffs = d_coeffs; P.meta = d_meta; P.spec_out = d_spec_out; P.tab = e->d_tab;
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
    HIP_TRY(e, hipStreamSynchronize(hip_stream ? (hipStream_t)hip_stream : e->stream), AACG_ERR_NO_DEVICE);
    return AACG_OK;
}

/* ---- host-buffer path: process(elements) + interleave for a batch ------------------- */
void* aacg_host_alloc(size_t bytes)
{
    void* p = nullptr;
    return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
void aacg_host_free(void* p) { if (p) (void)hipHostFree(p); }

int aacg_wait(aacg_engine* e, uint64_t ticket)
{
    if (!e || ticket == 0 || ticket > e->submitted) return AACG_ERR_INVALID_ARG;
    aacg_engine::slot_t& sl = e->slot[(ticket - 1) & 1];
    if (sl.busy) {
        HIP_TRY(e, hipEventSynchronize(sl.done), AACG_ERR_NO_DEVICE);
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
    const size_t tb = sizeof(aacg_dev_tns) * h.tns.size();
    const size_t sb = h.needs_scratch ? h.runs.size() * AACG_SLOT_FLOATS * sizeof(float) : 0;
    const size_t xb = needs_spec_buffer(e, h) ? (size_t)n_coef_blocks * 1024u * sizeof(float) : 0;
    const size_t ccb[4] = {sizeof(aacg_run) * h.cce_runs.size(), sizeof(aacg_couple_job) * h.couple_jobs.size(),
                           sizeof(float) * h.gains.size(), (size_t)h.side_blocks * 4096u};
    const void* const cce_src[4] = {h.cce_runs.data(), h.couple_jobs.data(), h.gains.data(), nullptr};
    for (int i = 0; i < 4; i++) if (ccb[i] && (rc = grow(e, &sl.d_cce[i], &sl.cce_cap[i], ccb[i]))) return rc;
    const bool rvp = takes_rv(e, h);
    const size_t rvs[4] = {rvp ? sizeof(aacg_run) * h.runs_rv.size() : 0, rvp ? sizeof(aacg_rv_link) * h.links_rv.size() : 0,
                           rvp ? sizeof(unsigned long long) * AACG8_RV_STATE_WORDS * (size_t)h.n_links_rv : 0,
                           rvp ? sizeof(float) * AACG8_RV_DATA_FLOATS * (size_t)h.n_links_rv : 0};
    for (int i = 0; i < 4; i++) {
        const size_t had = sl.rv_cap[i];
        if (rvs[i] && (rc = grow(e, &sl.d_rv[i], &sl.rv_cap[i], rvs[i]))) return rc;
        /* a new state buffer starts from zero — on the slot's own stream, in front of the launch that reads it (a plain
         * hipMemset is ordered on the null stream, which a non-blocking stream does not wait for) */
        if (i == 2 && sl.rv_cap[i] != had) HIP_TRY(e, hipMemsetAsync(sl.d_rv[i], 0, sl.rv_cap[i], sl.stream), AACG_ERR_NO_DEVICE);
    }
    const bool r8 = takes_run8(e, h);
    const size_t r8b[3] = {r8 ? sizeof(aacg_run8) * h.runs8.size() : 0, r8 ? sizeof(unsigned long long) * AACG8_RV_STATE_WORDS * (size_t)h.n_links : 0,
                           r8 ? sizeof(float) * AACG8_RV_DATA_FLOATS * (size_t)h.n_links : 0};
    for (int i = 0; i < 3; i++) {
        const size_t had = sl.run8_cap[i];
        if (r8b[i] && (rc = grow(e, &sl.d_run8[i], &sl.run8_cap[i], r8b[i]))) return rc;
        if (i == 1 && sl.run8_cap[i] != had) HIP_TRY(e, hipMemsetAsync(sl.d_run8[i], 0, sl.run8_cap[i], sl.stream), AACG_ERR_NO_DEVICE);   /* likewise */
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
    if (r8b[0]) HIP_TRY(e, hipMemcpyAsync(sl.d_run8[0], h.runs8.data(), r8b[0], hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    if (rvs[0]) HIP_TRY(e, hipMemcpyAsync(sl.d_rv[0], h.runs_rv.data(), rvs[0], hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    if (rvs[1]) HIP_TRY(e, hipMemcpyAsync(sl.d_rv[1], h.links_rv.data(), rvs[1], hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    if (tb) HIP_TRY(e, hipMemcpyAsync(sl.d_tns, h.tns.data(), tb, hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    for (int i = 0; i < 3; i++) if (ccb[i]) HIP_TRY(e, hipMemcpyAsync(sl.d_cce[i], cce_src[i], ccb[i], hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    HIP_TRY(e, hipMemcpyAsync(sl.d_coeffs, src_coeffs, cb, hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    if (quant) HIP_TRY(e, hipMemcpyAsync(sl.d_meta, src_meta, mb, hipMemcpyHostToDevice, s), AACG_ERR_NO_DEVICE);
    /* kernels chain through the overlap state: this one starts after the previous batch's kernel,
     * while its uploads above overlapped it */
    if (e->last_kernel) HIP_TRY(e, hipStreamWaitEvent(s, e->last_kernel, 0), AACG_ERR_NO_DEVICE);
    rc = launch_run(e, (const aacg_dev_unit*)sl.d_units, (const aacg_run*)sl.d_runs, (const aacg_dev_tns*)sl.d_tns,
                    (float*)sl.d_scratch, (float*)sl.d_spec, cce_bufs{(const aacg_run*)sl.d_cce[0], (const aacg_couple_job*)sl.d_cce[1], (const float*)sl.d_cce[2], (float*)sl.d_cce[3]},
                    run8_bufs{(const aacg_run8*)sl.d_run8[0], (unsigned long long*)sl.d_run8[1], (float*)sl.d_run8[2]},
                    rv_bufs{(const aacg_run*)sl.d_rv[0], (const aacg_rv_link*)sl.d_rv[1], (unsigned long long*)sl.d_rv[2], (float*)sl.d_rv[3]}, h, sl.d_coeffs,
                    (const aacg_band_meta*)sl.d_meta, sl.d_pcm, 0, s);
    if (rc) return rc;
    HIP_TRY(e, hipEventRecord(sl.kernel_done, s), AACG_ERR_NO_DEVICE);
    e->last_kernel = sl.kernel_done;
    HIP_TRY(e, hipMemcpyAsync(dst_pcm, sl.d_pcm, pb, hipMemcpyDeviceToHost, s), AACG_ERR_NO_DEVICE);
    HIP_TRY(e, hipEventRecord(sl.done, s), AACG_ERR_NO_DEVICE);
    sl.busy = true;

    for (const aacg_chain& c : h.chains)
        for (int k = 0; k < c.n_ch; k++)
            e->parity[(size_t)c.stream * (size_t)e->cfg.max_channels + c.channel + k] ^= 1;
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
