/*
 * aacg_pipeline.hip — bytes in, PCM out: the device front end and the transform behind ONE call per batch, several batches in
 * flight (aacg_pipeline_*, include/aacgpu.h).
 *
 * What a host of the reference does per frame and per stream in `readChunk()` (src/decoder.js:125-216: parse the
 * raw_data_block, process(elements), interleave), for a batch of streams at once and without the host in between: the frames'
 * bytes go up through page-locked staging, aacg_parse_device writes the unit records / spectra / band words in HBM,
 * aacg_plan_refresh_from_parse_ex turns them into a KEPT plan's unit records (the plan's run tables depend on which streams
 * bring how many frames of which elements, not on what the frames hold), aacg_decode_pipelined runs the transform, and the PCM
 * comes down into page-locked memory.  The host only finds the frame boundaries (ADTS frame_length) and says which stream slot
 * each run of frames belongs to.
 *
 * Round 6:
 *   - LANES.  Round 5 ran H2D -> parse -> refresh -> transform -> D2H -> hipStreamSynchronize on one stream, one batch at a
 *     time: 1.14 ms per 4096-frame stereo batch, of which the PCM's way down PCIe alone is 0.6.  Now a batch takes one of
 *     `lanes` sets of device buffers and a HIP stream of its own; aacg_pipeline_submit returns when the batch is enqueued,
 *     aacg_pipeline_collect waits for it (bounded).  Batch k's PCM goes down while batch k + 1's bytes go up and are parsed; the
 *     transform launches of consecutive batches go through aacg_decode_pipelined — the SAME plan continued launch after launch,
 *     its unit records kept in one set per lane (aacg_plan_set_unit_sets) so that refreshing the next batch's records does not
 *     wait for the launch that reads the previous ones — and so meet in the cross-launch rendezvous cells like bench.py's.
 *   - LAYOUTS.  Streams of up to eight channels: a stream's element layout (which SCE / CPE / LFE elements a frame has, in
 *     order — decoder.js:233-247 deals channels out in element order and drops what exceeds chanConfig) is learnt from its
 *     first frame (one small synchronous parse when a stream is new), the kept plan lists exactly those elements, and a
 *     later frame with other elements is refused as a whole (AACG_PARSE_LAYOUT).
 *
 * Host code only (the kernels are the parser's and the engine's); it uses nothing but the public ABI of those two.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/aacgpu.h"
#include "../../include/aacgpu_tools.h"
#include "aacg_wait.h"

#define AACG_PIPELINE_MAX_LANES 8

/* The batch's bytes up (and its few kilobytes of results down) are moved by THIS kernel, not by hipMemcpyAsync: page-locked host
 * memory is mapped into the device's address space, and a few workgroups of 16-byte loads and stores move 1.4 MB in 30 us.  The
 * runtime's copies go through the two SDMA engines whatever their direction: in round 6's first build a lane's bytes waited up
 * to 0.6 ms behind another lane's PCM on its way down (kernel + copy trace, profiles/r06_resident_budget.txt), and the lanes ran
 * in lockstep pairs with the link idle a third of the time — 0.89-0.93 ms per batch.  A kernel on the lane's own stream waits
 * for nothing but its predecessor on that stream.  The PCM itself stays with the SDMA engines, ONE copy per batch: 0.64 ms per
 * batch (52 GB/s).  Measured against it: the PCM through this kernel too — 0.89, its PCIe-bound stores hold up the other
 * kernels' stores (the parse kernel then takes 1.0-1.4 ms instead of 0.55) — and the PCM as two SDMA halves on two streams —
 * 0.84-0.88, the halves of one batch take both engines and the next lane's copy waits behind them. */
extern "C" __global__ __launch_bounds__(256)
void aacg_pipe_copy(const uint4* src, uint4* dst, size_t n16, int clear_last)
{
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u* s = (const v4u*)src;
    v4u* d = (v4u*)dst;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        __builtin_nontemporal_store(__builtin_nontemporal_load(&s[i]), &d[i]);
        /* the results' copy: its last 16 bytes are the batch's refusal count, cleared for the lane's next batch by the lane that has
         * just read them (one launch less on the lane's stream than a memset in front of every batch) */
        if (clear_last && i == n16 - 1) { const v4u zero = {0u, 0u, 0u, 0u}; ((v4u*)src)[i] = zero; }
    }
}

struct aacg_pipeline {
    aacg_pipeline_config cfg;
    aacg_engine* engine = nullptr;
    aacg_parser* parser = nullptr;      /* layouts of new streams (synchronous, host pointers) */
    int n_lanes = 5;
    uint32_t C = 2;                     /* channels of a frame's PCM (chanConfig) */
    uint32_t Cp = 2, U = 1;             /* what the parser is allowed per frame: channels (block stride), elements */
    bool learn = false;                 /* C > 2: layouts are learnt; C <= 2: every frame one SCE / one CPE */
    /* a stream's element layout: channels of every SCE / LFE / CPE of a frame in order; kept = how many of them fit into C channels */
    struct layout_t { uint8_t n = 0, kept = 0; uint8_t nch[8] = {}; };
    std::vector<layout_t> layout;       /* per slot; n = 0: not learnt yet */
    /* kept plans, by batch shape: (frames per stream, the stream slots in order); the layouts are the slots' */
    struct kept { uint32_t frames; std::vector<uint32_t> slots; aacg_plan* plan; void* d_map; uint32_t n_units; uint64_t used; };
    std::vector<kept> plans;
    uint64_t tick = 0;
    struct lane_t {
        /* A parser of its own per lane: a parser's launches are ordered one behind the other (they share its lane-order scratch),
         * and a batch's parse is the longest single step of the route — one GPU lane walks one frame's bits, so 4096 frames keep a
         * quarter of the chip busy for as long as the longest frame takes (0.5-0.8 ms; profiles/r06_resident_budget.txt).  With a
         * parser per lane the parses of consecutive batches run side by side. */
        aacg_parser* parser = nullptr;
        hipStream_t st = nullptr;
        hipEvent_t done = nullptr;
        void *d_bytes = nullptr, *d_frames = nullptr, *d_units = nullptr, *d_q = nullptr, *d_meta = nullptr, *d_res = nullptr, *d_pcm = nullptr, *d_refused = nullptr;
        void *h_in = nullptr, *h_pcm = nullptr, *h_res = nullptr;
        size_t bytes_cap = 0, h_in_cap = 0;
        /* the batch in flight */
        bool busy = false, count_stale = false;
        uint64_t ticket = 0;
        void* user_pcm = nullptr; bool direct = false; size_t pcm_bytes = 0;
        aacg_parse_result* user_results = nullptr; uint32_t* user_refused = nullptr; uint32_t n = 0, F = 0;
        std::vector<uint32_t> unlearnt;   /* streams of the batch (by position) whose layout was not known: nothing of them was decoded */
    } lane[AACG_PIPELINE_MAX_LANES];
    uint64_t submitted = 0;
    size_t res_cap16 = 0;               /* a lane's h_res: results of up to max_streams x max_frames frames, then (16-byte aligned) the refusal count */
    aacg_wait_policy wait;
    std::string err;
};

namespace {

bool ok(aacg_pipeline* p, hipError_t rc, const char* what)
{
    if (rc == hipSuccess) return true;
    p->err = std::string(what) + ": " + hipGetErrorString(rc);
    return false;
}
#define P_TRY(p, call, code) do { if (!ok((p), (call), #call)) return (code); } while (0)

size_t pcm_elem(const aacg_pipeline* p) { return p->cfg.output_kind == AACG_OUTPUT_I16 ? 2 : 4; }

/* true if the runtime knows this host pointer as page-locked (aacg_host_alloc / hipHostMalloc / hipHostRegister) */
bool is_pinned(const void* ptr)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, ptr) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

/* bytes (a multiple of 16, both ends 16-byte aligned) between device memory and mapped page-locked host memory, on stream s */
void pipe_copy(const void* src, void* dst, size_t bytes, hipStream_t s, bool clear_last = false)
{
    const size_t n16 = bytes / 16;
    if (!n16) return;
    const unsigned blocks = (unsigned)((n16 + 255) / 256 < 64 ? (n16 + 255) / 256 : 64);      /* 64 x 256 lanes x 16 bytes in flight: the link, not the chip */
    hipLaunchKernelGGL(aacg_pipe_copy, dim3(blocks), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, n16, clear_last ? 1 : 0);
}

void drop_plan(aacg_pipeline* p, size_t i)
{
    aacg_plan_destroy(p->plans[i].plan);                 /* waits (bounded) for the launches that read it */
    if (p->plans[i].d_map) (void)hipFree(p->plans[i].d_map);
    p->plans.erase(p->plans.begin() + (long)i);
}
void drop_plans(aacg_pipeline* p) { while (!p->plans.empty()) drop_plan(p, p->plans.size() - 1); }

/* The plan for a batch of this shape.  Frame f of stream s is parsed frame i = s * F + f; the parser (max_units U, max_channels
 * Cp) puts its element e at record i * U + e and its running channel c at block i * Cp + c.  The plan lists, frame by frame,
 * the elements of the stream's layout that fit into the C output channels. */
int plan_for(aacg_pipeline* p, const uint32_t* slots, uint32_t S, uint32_t F, aacg_pipeline::kept** out)
{
    for (auto& k : p->plans)
        if (k.frames == F && k.slots.size() == S && std::memcmp(k.slots.data(), slots, S * sizeof(uint32_t)) == 0) { k.used = ++p->tick; *out = &k; return AACG_OK; }
    const uint32_t C = p->C, Cp = p->Cp, U = p->U;
    std::vector<aacg_unit_desc> u;
    std::vector<aacg_refresh_map> map;
    u.reserve((size_t)S * F * (p->learn ? 4 : 1));
    for (uint32_t s = 0; s < S; s++) {
        aacg_pipeline::layout_t lay = p->layout[slots[s]];
        if (!p->learn) { lay.n = lay.kept = 1; lay.nch[0] = (uint8_t)C; }
        for (uint32_t f = 0; f < F; f++) {
            const uint32_t i = s * F + f;
            uint32_t chan = 0;
            for (uint32_t e = 0; e < lay.kept; e++) {
                aacg_unit_desc d;
                std::memset(&d, 0, sizeof d);
                d.stream = slots[s]; d.pcm_offset = i * 1024u * C; d.channel = (uint16_t)chan; d.n_out_ch = (uint16_t)C; d.n_ch = lay.nch[e];
                d.coef_offset = d.meta_offset = i * Cp + chan;
                for (uint32_t c = 0; c < d.n_ch; c++) { d.ch[c].group_count = 1; d.ch[c].group_len[0] = 1; }
                u.push_back(d);
                map.push_back({i * U + e, (uint32_t)lay.n | ((uint32_t)lay.kept << 8)});
                chan += lay.nch[e];
            }
        }
    }
    if (u.empty()) { *out = nullptr; return AACG_OK; }     /* no stream of the batch has a layout yet: nothing to transform, every frame is refused */
    aacg_plan* plan = nullptr;
    int rc = aacg_plan_create(p->engine, u.data(), (uint32_t)u.size(), &plan);
    if (rc == AACG_OK && (rc = aacg_plan_set_unit_sets(p->engine, plan, (uint32_t)p->n_lanes))) { aacg_plan_destroy(plan); plan = nullptr; }
    if (rc) { p->err = std::string("aacg_plan_create: ") + aacg_last_error(p->engine); return rc; }
    void* d_map = nullptr;
    if (hipMalloc(&d_map, map.size() * sizeof(aacg_refresh_map)) != hipSuccess ||
        hipMemcpy(d_map, map.data(), map.size() * sizeof(aacg_refresh_map), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        if (d_map) (void)hipFree(d_map);
        aacg_plan_destroy(plan);
        p->err = "hipMalloc (plan map)";
        return AACG_ERR_OUT_OF_MEMORY;
    }
    if (p->plans.size() >= 8) {                          /* the least recently used shape makes room */
        size_t lru = 0;
        for (size_t i = 1; i < p->plans.size(); i++) if (p->plans[i].used < p->plans[lru].used) lru = i;
        drop_plan(p, lru);
    }
    p->plans.push_back({F, std::vector<uint32_t>(slots, slots + S), plan, d_map, (uint32_t)u.size(), ++p->tick});
    *out = &p->plans.back();
    return AACG_OK;
}

std::string lanes_text(aacg_pipeline* p)
{
    std::string o;
    char b[160];
    for (int k = 0; k < p->n_lanes; k++) {
        const auto& L = p->lane[k];
        hipError_t st = L.st ? hipStreamQuery(L.st) : hipSuccess;
        (void)hipGetLastError();
        std::snprintf(b, sizeof b, "%slane %d %s (ticket %llu, %s)", k ? ", " : "", k, st == hipSuccess ? "idle" : st == hipErrorNotReady ? "BUSY" : "error",
                      (unsigned long long)L.ticket, L.busy ? "not collected" : "collected");
        o += b;
    }
    return o;
}

int timed_out(aacg_pipeline* p, const char* what)
{
    char b[200], dump[3000] = "";
    std::snprintf(b, sizeof b, "%s: the GPU did not answer within %.1f s (aacg_pipeline_set_wait_limit_ms); %llu batches submitted; ", what, p->wait.limit_s, (unsigned long long)p->submitted);
    (void)aacg_debug_in_flight(p->engine, dump, sizeof dump);
    p->err = std::string(b) + lanes_text(p) + "; engine: " + dump;
    return AACG_ERR_TIMEOUT;
}

/* the lane's batch is complete: what it staged goes to the caller */
int finish_lane(aacg_pipeline* p, aacg_pipeline::lane_t& L)
{
    if (!L.busy) return AACG_OK;
    const hipError_t st = aacg_wait_event(L.done, p->wait);
    if (st == hipErrorNotReady) return timed_out(p, "aacg_pipeline_collect");
    P_TRY(p, st, AACG_ERR_NO_DEVICE);
    if (!L.direct) std::memcpy(L.user_pcm, L.h_pcm, L.pcm_bytes);
    /* a stream whose first frame did not parse has no layout yet: every frame of it in this batch came out silent and says so */
    aacg_parse_result* res = (aacg_parse_result*)L.h_res;
    uint32_t* refused = (uint32_t*)((char*)L.h_res + p->res_cap16);
    for (uint32_t s : L.unlearnt)
        for (uint32_t f = 0; f < L.F; f++) { aacg_parse_result& r = res[(size_t)s * L.F + f]; if (r.status == AACG_PARSE_OK) r.status = AACG_PARSE_LAYOUT; (*refused)++; }
    if (L.user_results) std::memcpy(L.user_results, res, (size_t)L.n * sizeof(aacg_parse_result));
    if (L.user_refused) std::memcpy(L.user_refused, refused, 4);
    L.busy = false;
    return AACG_OK;
}

/* the element layouts of streams that are new: their first frames through one small synchronous parse (host pointers) */
int learn_layouts(aacg_pipeline* p, const uint8_t* bytes, size_t n_bytes, const aacg_parse_frame* frames, const uint32_t* slots, uint32_t S, uint32_t F)
{
    std::vector<uint32_t> who;
    for (uint32_t s = 0; s < S; s++) if (!p->layout[slots[s]].n) who.push_back(s);
    if (who.empty()) return AACG_OK;
    /* gather those frames' bytes (they lie anywhere in the caller's buffer) */
    std::vector<aacg_parse_frame> fr(who.size());
    std::vector<uint8_t> buf;
    for (size_t k = 0; k < who.size(); k++) {
        const aacg_parse_frame& f = frames[(size_t)who[k] * F];
        fr[k].byte_offset = (uint32_t)buf.size(); fr[k].byte_length = f.byte_length;
        buf.insert(buf.end(), bytes + f.byte_offset, bytes + f.byte_offset + f.byte_length);
        buf.resize((buf.size() + 15) & ~(size_t)15);
    }
    (void)n_bytes;
    const uint32_t n = (uint32_t)who.size(), U = p->U, Cp = p->Cp;
    std::vector<aacg_unit_desc> units((size_t)n * U);
    std::vector<int16_t> q((size_t)n * Cp * 1024);
    std::vector<aacg_band_meta> meta((size_t)n * Cp);
    std::vector<aacg_parse_result> res(n);
    int rc = aacg_parse_batch(p->parser, buf.data(), buf.size(), fr.data(), n, U, Cp, (uint32_t)p->cfg.parse_options, units.data(), q.data(), meta.data(), nullptr, res.data());
    if (rc) { p->err = std::string("learning the streams' element layouts: ") + aacg_parser_last_error(p->parser); return rc; }
    for (uint32_t k = 0; k < n; k++) {
        if (res[k].status != AACG_PARSE_OK || !res[k].n_units) continue;      /* not learnt: the stream's frames of this batch are refused, the next batch tries again */
        aacg_pipeline::layout_t lay;
        uint32_t chan = 0;
        lay.n = res[k].n_units > 8 ? 8 : res[k].n_units;
        for (uint32_t e = 0; e < lay.n; e++) {
            lay.nch[e] = units[(size_t)k * U + e].n_ch;
            if (chan + lay.nch[e] <= p->C && lay.kept == e) lay.kept = (uint8_t)(e + 1);   /* decoder.js:233: elements while channel < channels; one that would cross the end is not taken either */
            chan += lay.nch[e];
        }
        p->layout[slots[who[k]]] = lay;
        /* plans made while this slot's layout was unknown list nothing of it */
        for (size_t i = p->plans.size(); i-- > 0;) {
            bool has = false;
            for (uint32_t s : p->plans[i].slots) has = has || s == slots[who[k]];
            if (has) drop_plan(p, i);
        }
    }
    return AACG_OK;
}

}  // namespace

extern "C" {

const char* aacg_pipeline_last_error(const aacg_pipeline* p) { return p ? p->err.c_str() : "null pipeline"; }

void aacg_pipeline_destroy(aacg_pipeline* p)
{
    if (!p) return;
    (void)hipSetDevice(p->cfg.device_ordinal);
    bool answered = true;
    for (int k = 0; k < p->n_lanes && answered; k++) if (p->lane[k].st) answered = aacg_wait_stream(p->lane[k].st, p->wait) != hipErrorNotReady;
    (void)hipGetLastError();
    if (!answered) {
        /* a device that does not answer is not waited for again and nothing of it is freed (hipFree waits for the device) */
        std::fprintf(stderr, "aacgpu: aacg_pipeline_destroy: the GPU did not answer within the wait limit — the pipeline's device memory is left allocated\n");
        delete p;
        return;
    }
    drop_plans(p);
    for (auto& L : p->lane) {
        for (void* d : {L.d_bytes, L.d_units, L.d_q, L.d_meta, L.d_res, L.d_pcm}) if (d) (void)hipFree(d);      /* (d_frames lies in d_bytes, d_refused in d_res) */
        for (void* h : {L.h_in, L.h_pcm, L.h_res}) if (h) (void)hipHostFree(h);
        if (L.done) (void)hipEventDestroy(L.done);
        if (L.parser) aacg_parser_destroy(L.parser);
    }
    if (p->parser) aacg_parser_destroy(p->parser);
    if (p->engine) aacg_destroy(p->engine);
    for (auto& L : p->lane) if (L.st) (void)hipStreamDestroy(L.st);
    delete p;
}

int aacg_pipeline_create(const aacg_pipeline_config* cfg, const aacg_code_entry* entries, const uint32_t counts[12], aacg_pipeline** out)
{
    if (!cfg || !out || !entries || !counts) return AACG_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->abi_version != AACG_ABI_VERSION || cfg->max_streams < 1 || cfg->max_frames < 1 || cfg->channels < 1 || cfg->channels > AACG_MAX_CHANNELS ||
        (cfg->output_kind != AACG_OUTPUT_F32 && cfg->output_kind != AACG_OUTPUT_I16) || (uint64_t)cfg->max_streams * (uint64_t)cfg->max_frames > (1u << 22) ||
        cfg->lanes < 0 || cfg->lanes > AACG_PIPELINE_MAX_LANES)
        return AACG_ERR_INVALID_ARG;
    aacg_pipeline* p = new (std::nothrow) aacg_pipeline();
    if (!p) return AACG_ERR_OUT_OF_MEMORY;
    p->cfg = *cfg;
    p->n_lanes = cfg->lanes ? cfg->lanes : 5;      /* four of the lowest priority and one of the level above: int16 PCM 0.37 -> 0.33 ms per batch, the link's rate; six and more share queues again */
    p->C = (uint32_t)cfg->channels;
    p->learn = p->C > 2;
    p->Cp = p->learn ? AACG_MAX_CHANNELS : p->C;
    p->U = p->learn ? 8u : 1u;
    p->layout.resize((size_t)cfg->max_streams);
    aacg_config ec;
    std::memset(&ec, 0, sizeof ec);
    ec.abi_version = AACG_ABI_VERSION; ec.device_ordinal = cfg->device_ordinal; ec.sample_index = cfg->sample_index;
    ec.max_streams = cfg->max_streams; ec.max_channels = cfg->channels; ec.input_kind = AACG_INPUT_QUANT_I16;
    ec.tns_mode = AACG_TNS_REFERENCE; ec.pns_mode = AACG_PNS_REFERENCE; ec.output_kind = cfg->output_kind; ec.cce_mode = AACG_CCE_REFERENCE;
    int rc = aacg_create(&ec, &p->engine);
    if (rc == AACG_OK) rc = aacg_parser_create(cfg->device_ordinal, cfg->sample_index, entries, counts, &p->parser);
    if (rc) { aacg_pipeline_destroy(p); return rc; }
    const size_t n = (size_t)cfg->max_streams * (size_t)cfg->max_frames, C = p->C, Cp = p->Cp, U = p->U;
    p->res_cap16 = (n * sizeof(aacg_parse_result) + 15) & ~(size_t)15;
    bool good = ok(p, hipSetDevice(cfg->device_ordinal), "hipSetDevice");
    /* The lanes' streams take the LOWEST priority: the runtime deals streams of one priority onto a handful of hardware queues
     * of their own, and two streams on one queue run one behind the other — a lane's parse behind another lane's copy down.
     * At the ordinary priority the lanes shared two queues with whatever else the process had made (kernel trace of round 6's first
     * build); at the lowest they are by themselves, and what competes with them for compute units — the transform launches on
     * the engine's highest-priority streams — is what should win. */
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    for (int k = 0; k < p->n_lanes && good; k++) {
        auto& L = p->lane[k];
        if (aacg_parser_create(cfg->device_ordinal, cfg->sample_index, entries, counts, &L.parser) != AACG_OK) { p->err = "aacg_parser_create (lane)"; good = false; break; }
        /* lanes 0..3 at the lowest priority, a level nobody else uses (four hardware queues); further lanes at the level between
         * that and the engine's pipeline streams: other queues again, so that a fifth lane does not wait behind the first */
        const int prio = k < 4 ? least : (least + greatest) / 2;
        good = ok(p, hipStreamCreateWithPriority(&L.st, hipStreamNonBlocking, prio), "hipStreamCreate") &&
               ok(p, hipEventCreateWithFlags(&L.done, hipEventDisableTiming), "hipEventCreate") &&
               ok(p, hipMalloc(&L.d_units, n * U * sizeof(aacg_unit_desc)), "hipMalloc") &&
               ok(p, hipMalloc(&L.d_q, n * Cp * 2048), "hipMalloc") && ok(p, hipMemsetAsync(L.d_q, 0, n * Cp * 2048, L.st), "hipMemset") &&
               ok(p, hipMalloc(&L.d_meta, n * Cp * sizeof(aacg_band_meta)), "hipMalloc") &&
               ok(p, hipMalloc(&L.d_res, p->res_cap16 + 16), "hipMalloc") &&      /* the refusal count lies behind the results: one copy brings both down */
               ok(p, hipMemsetAsync(L.d_res, 0, p->res_cap16 + 16, L.st), "hipMemset") &&
               ok(p, hipMalloc(&L.d_pcm, n * C * 1024 * pcm_elem(p)), "hipMalloc") &&
               ok(p, hipHostMalloc(&L.h_res, p->res_cap16 + 16, hipHostMallocDefault), "hipHostMalloc") &&
               aacg_wait_stream(L.st, p->wait) == hipSuccess;
        if (good) L.d_refused = (char*)L.d_res + p->res_cap16;
    }
    if (!good) { if (p->err.empty()) p->err = "the pipeline's set-up did not complete"; std::fprintf(stderr, "aacgpu: %s\n", p->err.c_str()); aacg_pipeline_destroy(p); return AACG_ERR_OUT_OF_MEMORY; }
    *out = p;
    return AACG_OK;
}

int aacg_pipeline_set_wait_limit_ms(aacg_pipeline* p, uint32_t ms)
{
    if (!p || !ms) return AACG_ERR_INVALID_ARG;
    p->wait.limit_s = ms * 1e-3;
    (void)aacg_set_wait_limit_ms(p->engine, ms);
    (void)aacg_parser_set_wait_limit_ms(p->parser, ms);
    for (int k = 0; k < p->n_lanes; k++) (void)aacg_parser_set_wait_limit_ms(p->lane[k].parser, ms);
    return AACG_OK;
}

int aacg_pipeline_reset_stream(aacg_pipeline* p, uint32_t slot)
{
    if (!p || (int)slot >= p->cfg.max_streams) return AACG_ERR_INVALID_ARG;
    /* the slot's batches in flight belong to the stream that had it: they are finished first (aacg_reset_stream waits for the
     * engine's launches; the PCM on its way down and the caller's buffers are the lanes') */
    for (int k = 0; k < p->n_lanes; k++) { int rc = finish_lane(p, p->lane[k]); if (rc) return rc; }
    int rc = aacg_reset_stream(p->engine, slot);
    if (rc) { p->err = aacg_last_error(p->engine); return rc; }
    if (p->learn && p->layout[slot].n) {                    /* a new stream: its layout is learnt anew; the plans made for the old one go */
        p->layout[slot] = aacg_pipeline::layout_t();
        for (size_t i = p->plans.size(); i-- > 0;) {
            bool has = false;
            for (uint32_t s : p->plans[i].slots) has = has || s == slot;
            if (has) drop_plan(p, i);
        }
    }
    return AACG_OK;
}

int aacg_pipeline_submit(aacg_pipeline* p, const uint8_t* bytes, size_t n_bytes, const aacg_parse_frame* frames,
                         const uint32_t* slots, uint32_t n_streams, uint32_t frames_per_stream,
                         void* pcm_out, aacg_parse_result* results, uint32_t* n_refused, uint64_t* ticket)
{
    if (!p || !bytes || !frames || !slots || !pcm_out || !n_streams || !frames_per_stream || !ticket) return AACG_ERR_INVALID_ARG;
    if ((int)n_streams > p->cfg.max_streams || (int)frames_per_stream > p->cfg.max_frames) { p->err = "batch larger than the pipeline was created for"; return AACG_ERR_CAPACITY; }
    const uint32_t n = n_streams * frames_per_stream, C = p->C, Cp = p->Cp, U = p->U;
    for (uint32_t s = 0; s < n_streams; s++) if ((int)slots[s] >= p->cfg.max_streams) { p->err = "stream slot out of range"; return AACG_ERR_CAPACITY; }
    {   /* a slot is one stream's overlap state: a batch brings each at most once (its frames are consecutive ones of that stream) */
        std::vector<bool> seen((size_t)p->cfg.max_streams, false);
        for (uint32_t s = 0; s < n_streams; s++) { if (seen[slots[s]]) { p->err = "a stream slot is listed twice in one batch"; return AACG_ERR_INVALID_ARG; } seen[slots[s]] = true; }
    }
    for (uint32_t i = 0; i < n; i++)
        if ((size_t)frames[i].byte_offset + frames[i].byte_length > n_bytes) { p->err = "a frame points outside the byte buffer"; return AACG_ERR_INVALID_ARG; }
    P_TRY(p, hipSetDevice(p->cfg.device_ordinal), AACG_ERR_NO_DEVICE);
    aacg_pipeline::lane_t& L = p->lane[p->submitted % (uint64_t)p->n_lanes];
    const uint32_t set = (uint32_t)(p->submitted % (uint64_t)p->n_lanes);
    int rc = finish_lane(p, L);                          /* the batch `lanes` submissions ago, if nobody has collected it */
    if (rc) return rc;
    if (p->learn && (rc = learn_layouts(p, bytes, n_bytes, frames, slots, n_streams, frames_per_stream))) return rc;
    /* staging: the bytes (16-byte aligned, AACG_PARSE_PAD readable bytes behind them) and the frame table in one page-locked block */
    const size_t padded = ((n_bytes + 15) & ~(size_t)15) + 64, table = (size_t)n * sizeof(aacg_parse_frame), up = padded + ((table + 15) & ~(size_t)15);
    if (up > L.h_in_cap) {
        if (L.h_in) (void)hipHostFree(L.h_in);
        L.h_in = nullptr; L.h_in_cap = 0;
        const size_t want = up * 3 / 2 + 4096;
        P_TRY(p, hipHostMalloc(&L.h_in, want, hipHostMallocDefault), AACG_ERR_OUT_OF_MEMORY);
        L.h_in_cap = want;
    }
    if (up > L.bytes_cap) {                              /* bytes and frame table travel as one block: the table lies behind the bytes on the device too */
        if (L.d_bytes) (void)hipFree(L.d_bytes);
        L.d_bytes = nullptr; L.bytes_cap = 0;
        const size_t want = up * 3 / 2 + 4096;
        P_TRY(p, hipMalloc(&L.d_bytes, want), AACG_ERR_OUT_OF_MEMORY);
        L.bytes_cap = want;
    }
    L.d_frames = (char*)L.d_bytes + padded;
    std::memcpy(L.h_in, bytes, n_bytes);
    std::memset((char*)L.h_in + n_bytes, 0, padded - n_bytes);
    std::memcpy((char*)L.h_in + padded, frames, table);
    aacg_pipeline::kept* kp = nullptr;
    if ((rc = plan_for(p, slots, n_streams, frames_per_stream, &kp))) return rc;
    const size_t pcm_bytes = (size_t)n * C * 1024u * pcm_elem(p);
    /* page-locked caller memory (aacg_host_alloc) takes the PCM straight from the device; anything else goes through the
     * lane's own page-locked staging and one host copy at collect */
    const bool direct = is_pinned(pcm_out);
    if (!direct && !L.h_pcm)
        P_TRY(p, hipHostMalloc(&L.h_pcm, (size_t)p->cfg.max_streams * (size_t)p->cfg.max_frames * C * 1024u * pcm_elem(p), hipHostMallocDefault), AACG_ERR_OUT_OF_MEMORY);
    hipStream_t st = L.st;
    L.unlearnt.clear();
    if (p->learn) for (uint32_t s = 0; s < n_streams; s++) if (!p->layout[slots[s]].kept) L.unlearnt.push_back(s);
    if (!L.unlearnt.empty()) P_TRY(p, hipMemsetAsync(L.d_pcm, 0, pcm_bytes, st), AACG_ERR_NO_DEVICE);      /* no unit writes their frames */
    pipe_copy(L.h_in, L.d_bytes, up, st);                 /* aacg_pipe_copy: not the SDMA engines, where it would queue behind other lanes' PCM */
    if (L.count_stale) P_TRY(p, hipMemsetAsync(L.d_refused, 0, 16, st), AACG_ERR_NO_DEVICE);      /* a submission that failed half-way left its count behind */
    L.count_stale = true;
    /* the spectra of a refused frame and the positions outside the coded bands are never read by the transform (a refused frame
     * becomes a silent unit), so the parser need not clear 8 KB per frame first */
    rc = aacg_parse_device(L.parser, L.d_bytes, (const aacg_parse_frame*)L.d_frames, n, U, Cp, (uint32_t)p->cfg.parse_options | AACG_PARSE_SKIP_ZERO_FILL,
                           (aacg_unit_desc*)L.d_units, (int16_t*)L.d_q, (aacg_band_meta*)L.d_meta, nullptr, (aacg_parse_result*)L.d_res, st);
    if (rc) { p->err = std::string("aacg_parse_device: ") + aacg_parser_last_error(L.parser); return rc; }
    for (int attempt = 0; kp; attempt++) {
        /* this lane's set of the plan's unit records: the launch that read it last was this lane's previous batch, whose PCM has
         * come down on this stream since */
        if (attempt) P_TRY(p, hipMemsetAsync(L.d_refused, 0, 16, st), AACG_ERR_NO_DEVICE);      /* the stale plan's refresh has counted this batch's refusals already */
        rc = aacg_plan_refresh_from_parse_ex(p->engine, kp->plan, (const aacg_unit_desc*)L.d_units, (aacg_parse_result*)L.d_res, U,
                                             (const aacg_refresh_map*)kp->d_map, set, (uint32_t*)L.d_refused, st);
        /* the transform: behind this lane's parse and refresh (fork), in front of its copy down (join); consecutive batches of
         * one shape are consecutive launches of one plan and overlap through the rendezvous cells */
        if (rc == AACG_OK) rc = aacg_pipeline_fork(p->engine, st);
        if (rc == AACG_OK) rc = aacg_decode_pipelined(p->engine, kp->plan, L.d_q, (const aacg_band_meta*)L.d_meta, L.d_pcm);
        if (rc != AACG_ERR_STALE_PLAN || attempt) break;
        /* another shape's plan has advanced these streams since this one was used: plans are made from the engine's current state */
        drop_plans(p);
        if ((rc = plan_for(p, slots, n_streams, frames_per_stream, &kp))) return rc;
    }
    if (rc == AACG_OK && kp) rc = aacg_pipeline_join(p->engine, st);
    if (rc) { p->err = std::string("transform: ") + aacg_last_error(p->engine); return rc; }
    {
        char* dst = (char*)(direct ? pcm_out : L.h_pcm);
        P_TRY(p, hipMemcpyAsync(dst, L.d_pcm, pcm_bytes, hipMemcpyDeviceToHost, st), AACG_ERR_NO_DEVICE);      /* one SDMA copy per batch (see aacg_pipe_copy) */
    }
    /* the results and, behind where the largest batch's would end, the refusal count: one small launch (a launch that writes to
     * host memory costs 50 us of the lane's time whatever it carries) */
    L.count_stale = false;
    pipe_copy(L.d_res, L.h_res, p->res_cap16 + 16, st, true);           /* ... and the count is cleared for the lane's next batch (set to zero at create) */
    P_TRY(p, hipGetLastError(), AACG_ERR_NO_DEVICE);
    P_TRY(p, hipEventRecord(L.done, st), AACG_ERR_NO_DEVICE);
    L.busy = true; L.ticket = ++p->submitted; L.user_pcm = pcm_out; L.direct = direct; L.pcm_bytes = pcm_bytes;
    L.user_results = results; L.user_refused = n_refused; L.n = n; L.F = frames_per_stream;
    *ticket = L.ticket;
    return AACG_OK;
}

int aacg_pipeline_collect(aacg_pipeline* p, uint64_t ticket)
{
    if (!p || !ticket || ticket > p->submitted) return AACG_ERR_INVALID_ARG;
    aacg_pipeline::lane_t& L = p->lane[(ticket - 1) % (uint64_t)p->n_lanes];
    if (L.ticket != ticket) return AACG_OK;              /* a later batch has taken the lane: this one was finished then */
    P_TRY(p, hipSetDevice(p->cfg.device_ordinal), AACG_ERR_NO_DEVICE);
    return finish_lane(p, L);
}

int aacg_pipeline_decode(aacg_pipeline* p, const uint8_t* bytes, size_t n_bytes, const aacg_parse_frame* frames,
                         const uint32_t* slots, uint32_t n_streams, uint32_t frames_per_stream,
                         void* pcm_out, aacg_parse_result* results, uint32_t* n_refused)
{
    uint64_t t = 0;
    int rc = aacg_pipeline_submit(p, bytes, n_bytes, frames, slots, n_streams, frames_per_stream, pcm_out, results, n_refused, &t);
    if (rc) return rc;
    return aacg_pipeline_collect(p, t);
}

int aacg_pipeline_stream_layout(aacg_pipeline* p, uint32_t slot, uint8_t element_channels[8], uint32_t* kept)
{
    if (!p || (int)slot >= p->cfg.max_streams) return AACG_ERR_INVALID_ARG;
    aacg_pipeline::layout_t lay = p->layout[slot];
    if (!p->learn) { lay.n = lay.kept = 1; lay.nch[0] = (uint8_t)p->C; }
    if (element_channels) std::memcpy(element_channels, lay.nch, 8);
    if (kept) *kept = lay.kept;
    return (int)lay.n;
}

}  // extern "C"
