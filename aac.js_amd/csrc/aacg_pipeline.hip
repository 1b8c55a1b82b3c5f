/*
 * aacg_pipeline.hip — bytes in, PCM out: the device front end and the transform behind ONE call per batch
 * (aacg_pipeline_*, include/aacgpu.h).
 *
 * What a host of the reference does per frame and per stream in `readChunk()` (src/decoder.js:125-216: parse the
 * raw_data_block, process(elements), interleave), for a batch of streams at once and without the host in between: the frames'
 * bytes go up through page-locked staging, aacg_parse_device writes the unit records / spectra / band words in HBM,
 * aacg_plan_refresh_from_parse turns them into a KEPT plan's unit records (the plan's run tables depend on which streams
 * bring how many frames, not on what the frames hold), aacg_decode_device runs the transform, and the PCM comes down through
 * page-locked staging — three kernels and three copies on one HIP stream, no host work per frame.  The host only finds the
 * frame boundaries (ADTS frame_length) and says which stream slot each run of frames belongs to.
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

struct aacg_pipeline {
    aacg_pipeline_config cfg;
    aacg_engine* engine = nullptr;
    aacg_parser* parser = nullptr;
    hipStream_t stream = nullptr;
    /* kept plans, by batch shape: (frames per stream, the stream slots in order) */
    struct kept { uint32_t frames; std::vector<uint32_t> slots; aacg_plan* plan; uint64_t used; };
    std::vector<kept> plans;
    uint64_t tick = 0;
    /* device buffers sized for max_streams x max_frames frames; page-locked staging both ways */
    void *d_bytes = nullptr, *d_frames = nullptr, *d_units = nullptr, *d_q = nullptr, *d_meta = nullptr, *d_res = nullptr, *d_pcm = nullptr, *d_refused = nullptr;
    void *h_in = nullptr, *h_pcm = nullptr, *h_res = nullptr;
    size_t bytes_cap = 0, h_in_cap = 0;
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

/* the plan for a batch of this shape: every frame one SCE (channels 1) or one CPE (channels 2), frame f of stream s at unit
 * s * F + f, its blocks where aacg_parse_device (max_units 1, max_channels C) puts them */
int plan_for(aacg_pipeline* p, const uint32_t* slots, uint32_t S, uint32_t F, aacg_plan** out)
{
    for (auto& k : p->plans)
        if (k.frames == F && k.slots.size() == S && std::memcmp(k.slots.data(), slots, S * sizeof(uint32_t)) == 0) { k.used = ++p->tick; *out = k.plan; return AACG_OK; }
    const uint32_t C = (uint32_t)p->cfg.channels, n = S * F;
    std::vector<aacg_unit_desc> u(n);
    std::memset(u.data(), 0, n * sizeof(aacg_unit_desc));
    for (uint32_t s = 0; s < S; s++)
        for (uint32_t f = 0; f < F; f++) {
            aacg_unit_desc& d = u[(size_t)s * F + f];
            const uint32_t i = s * F + f;
            d.stream = slots[s]; d.pcm_offset = i * 1024u * C; d.channel = 0; d.n_out_ch = (uint16_t)C; d.n_ch = (uint8_t)C;
            d.coef_offset = d.meta_offset = i * C;
            for (uint32_t c = 0; c < C; c++) { d.ch[c].group_count = 1; d.ch[c].group_len[0] = 1; }
        }
    aacg_plan* plan = nullptr;
    int rc = aacg_plan_create(p->engine, u.data(), n, &plan);
    if (rc) { p->err = std::string("aacg_plan_create: ") + aacg_last_error(p->engine); return rc; }
    if (p->plans.size() >= 8) {                          /* the least recently used shape makes room */
        size_t lru = 0;
        for (size_t i = 1; i < p->plans.size(); i++) if (p->plans[i].used < p->plans[lru].used) lru = i;
        aacg_plan_destroy(p->plans[lru].plan);
        p->plans.erase(p->plans.begin() + (long)lru);
    }
    p->plans.push_back({F, std::vector<uint32_t>(slots, slots + S), plan, ++p->tick});
    *out = plan;
    return AACG_OK;
}

void drop_plans(aacg_pipeline* p)
{
    for (auto& k : p->plans) aacg_plan_destroy(k.plan);
    p->plans.clear();
}

}  // namespace

extern "C" {

const char* aacg_pipeline_last_error(const aacg_pipeline* p) { return p ? p->err.c_str() : "null pipeline"; }

void aacg_pipeline_destroy(aacg_pipeline* p)
{
    if (!p) return;
    (void)hipSetDevice(p->cfg.device_ordinal);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    drop_plans(p);
    for (void* d : {p->d_bytes, p->d_frames, p->d_units, p->d_q, p->d_meta, p->d_res, p->d_pcm, p->d_refused}) if (d) (void)hipFree(d);
    for (void* h : {p->h_in, p->h_pcm, p->h_res}) if (h) (void)hipHostFree(h);
    if (p->parser) aacg_parser_destroy(p->parser);
    if (p->engine) aacg_destroy(p->engine);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

int aacg_pipeline_create(const aacg_pipeline_config* cfg, const aacg_code_entry* entries, const uint32_t counts[12], aacg_pipeline** out)
{
    if (!cfg || !out || !entries || !counts) return AACG_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->abi_version != AACG_ABI_VERSION || cfg->max_streams < 1 || cfg->max_frames < 1 || (cfg->channels != 1 && cfg->channels != 2) ||
        (cfg->output_kind != AACG_OUTPUT_F32 && cfg->output_kind != AACG_OUTPUT_I16) || (uint64_t)cfg->max_streams * (uint64_t)cfg->max_frames > (1u << 22))
        return AACG_ERR_INVALID_ARG;
    aacg_pipeline* p = new (std::nothrow) aacg_pipeline();
    if (!p) return AACG_ERR_OUT_OF_MEMORY;
    p->cfg = *cfg;
    aacg_config ec;
    std::memset(&ec, 0, sizeof ec);
    ec.abi_version = AACG_ABI_VERSION; ec.device_ordinal = cfg->device_ordinal; ec.sample_index = cfg->sample_index;
    ec.max_streams = cfg->max_streams; ec.max_channels = cfg->channels; ec.input_kind = AACG_INPUT_QUANT_I16;
    ec.tns_mode = AACG_TNS_REFERENCE; ec.pns_mode = AACG_PNS_REFERENCE; ec.output_kind = cfg->output_kind; ec.cce_mode = AACG_CCE_REFERENCE;
    int rc = aacg_create(&ec, &p->engine);
    if (rc == AACG_OK) rc = aacg_parser_create(cfg->device_ordinal, cfg->sample_index, entries, counts, &p->parser);
    if (rc) { aacg_pipeline_destroy(p); return rc; }
    const size_t n = (size_t)cfg->max_streams * (size_t)cfg->max_frames, C = (size_t)cfg->channels;
    const bool good =
        ok(p, hipSetDevice(cfg->device_ordinal), "hipSetDevice") &&
        ok(p, hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking), "hipStreamCreate") &&
        ok(p, hipMalloc(&p->d_frames, n * sizeof(aacg_parse_frame)), "hipMalloc") &&
        ok(p, hipMalloc(&p->d_units, n * sizeof(aacg_unit_desc)), "hipMalloc") &&
        ok(p, hipMalloc(&p->d_q, n * C * 2048), "hipMalloc") && ok(p, hipMemset(p->d_q, 0, n * C * 2048), "hipMemset") &&
        ok(p, hipMalloc(&p->d_meta, n * C * sizeof(aacg_band_meta)), "hipMalloc") &&
        ok(p, hipMalloc(&p->d_res, n * sizeof(aacg_parse_result)), "hipMalloc") &&
        ok(p, hipMalloc(&p->d_pcm, n * C * 1024 * pcm_elem(p)), "hipMalloc") &&
        ok(p, hipMalloc(&p->d_refused, 16), "hipMalloc") &&
        ok(p, hipHostMalloc(&p->h_pcm, n * C * 1024 * pcm_elem(p), hipHostMallocDefault), "hipHostMalloc") &&
        ok(p, hipHostMalloc(&p->h_res, n * sizeof(aacg_parse_result) + 16, hipHostMallocDefault), "hipHostMalloc") &&
        ok(p, hipDeviceSynchronize(), "hipDeviceSynchronize");
    if (!good) { std::fprintf(stderr, "aacgpu: %s\n", p->err.c_str()); aacg_pipeline_destroy(p); return AACG_ERR_OUT_OF_MEMORY; }
    *out = p;
    return AACG_OK;
}

int aacg_pipeline_reset_stream(aacg_pipeline* p, uint32_t slot)
{
    if (!p) return AACG_ERR_INVALID_ARG;
    int rc = aacg_reset_stream(p->engine, slot);
    if (rc) p->err = aacg_last_error(p->engine);
    return rc;
}

int aacg_pipeline_decode(aacg_pipeline* p, const uint8_t* bytes, size_t n_bytes, const aacg_parse_frame* frames,
                         const uint32_t* slots, uint32_t n_streams, uint32_t frames_per_stream,
                         void* pcm_out, aacg_parse_result* results, uint32_t* n_refused)
{
    if (!p || !bytes || !frames || !slots || !pcm_out || !n_streams || !frames_per_stream) return AACG_ERR_INVALID_ARG;
    if ((int)n_streams > p->cfg.max_streams || (int)frames_per_stream > p->cfg.max_frames) { p->err = "batch larger than the pipeline was created for"; return AACG_ERR_CAPACITY; }
    const uint32_t n = n_streams * frames_per_stream, C = (uint32_t)p->cfg.channels;
    for (uint32_t s = 0; s < n_streams; s++) if ((int)slots[s] >= p->cfg.max_streams) { p->err = "stream slot out of range"; return AACG_ERR_CAPACITY; }
    for (uint32_t i = 0; i < n; i++)
        if ((size_t)frames[i].byte_offset + frames[i].byte_length > n_bytes) { p->err = "a frame points outside the byte buffer"; return AACG_ERR_INVALID_ARG; }
    P_TRY(p, hipSetDevice(p->cfg.device_ordinal), AACG_ERR_NO_DEVICE);
    /* staging: the bytes (16-byte aligned, AACG_PARSE_PAD readable bytes behind them) and the frame table in one page-locked block */
    const size_t padded = ((n_bytes + 15) & ~(size_t)15) + 64, table = (size_t)n * sizeof(aacg_parse_frame);
    if (padded + table > p->h_in_cap) {
        if (p->h_in) (void)hipHostFree(p->h_in);
        p->h_in = nullptr; p->h_in_cap = 0;
        const size_t want = (padded + table) * 3 / 2 + 4096;
        P_TRY(p, hipHostMalloc(&p->h_in, want, hipHostMallocDefault), AACG_ERR_OUT_OF_MEMORY);
        p->h_in_cap = want;
    }
    if (padded > p->bytes_cap) {
        if (p->d_bytes) (void)hipFree(p->d_bytes);
        p->d_bytes = nullptr; p->bytes_cap = 0;
        const size_t want = padded * 3 / 2 + 4096;
        P_TRY(p, hipMalloc(&p->d_bytes, want), AACG_ERR_OUT_OF_MEMORY);
        p->bytes_cap = want;
    }
    std::memcpy(p->h_in, bytes, n_bytes);
    std::memset((char*)p->h_in + n_bytes, 0, padded - n_bytes);
    std::memcpy((char*)p->h_in + padded, frames, table);
    aacg_plan* plan = nullptr;
    int rc = plan_for(p, slots, n_streams, frames_per_stream, &plan);
    if (rc) return rc;
    hipStream_t st = p->stream;
    P_TRY(p, hipMemcpyAsync(p->d_bytes, p->h_in, padded, hipMemcpyHostToDevice, st), AACG_ERR_NO_DEVICE);
    P_TRY(p, hipMemcpyAsync(p->d_frames, (char*)p->h_in + padded, table, hipMemcpyHostToDevice, st), AACG_ERR_NO_DEVICE);
    P_TRY(p, hipMemsetAsync(p->d_refused, 0, 4, st), AACG_ERR_NO_DEVICE);
    /* the spectra of a refused frame and the positions outside the coded bands are never read by the transform (a refused frame
     * becomes a silent unit), so the parser need not clear 8 KB per frame first */
    rc = aacg_parse_device(p->parser, p->d_bytes, (const aacg_parse_frame*)p->d_frames, n, 1, C, (uint32_t)p->cfg.parse_options | AACG_PARSE_SKIP_ZERO_FILL,
                           (aacg_unit_desc*)p->d_units, (int16_t*)p->d_q, (aacg_band_meta*)p->d_meta, nullptr, (aacg_parse_result*)p->d_res, st);
    if (rc) { p->err = std::string("aacg_parse_device: ") + aacg_parser_last_error(p->parser); return rc; }
    for (int attempt = 0;; attempt++) {
        rc = aacg_plan_refresh_from_parse(p->engine, plan, (const aacg_unit_desc*)p->d_units, (const aacg_parse_result*)p->d_res, 1, (uint32_t*)p->d_refused, st);
        if (rc == AACG_OK) rc = aacg_decode_device(p->engine, plan, p->d_q, (const aacg_band_meta*)p->d_meta, p->d_pcm, st);
        if (rc != AACG_ERR_STALE_PLAN || attempt) break;
        /* another shape's plan has advanced these streams since this one was used: plans are made from the engine's current state */
        drop_plans(p);
        if ((rc = plan_for(p, slots, n_streams, frames_per_stream, &plan))) return rc;
    }
    if (rc) { p->err = std::string("transform: ") + aacg_last_error(p->engine); return rc; }
    const size_t pcm_bytes = (size_t)n * C * 1024u * pcm_elem(p);
    /* page-locked caller memory (aacg_host_alloc) takes the PCM straight from the device; anything else goes through the
     * pipeline's own page-locked staging and one host copy */
    const bool direct = is_pinned(pcm_out);
    P_TRY(p, hipMemcpyAsync(direct ? pcm_out : p->h_pcm, p->d_pcm, pcm_bytes, hipMemcpyDeviceToHost, st), AACG_ERR_NO_DEVICE);
    P_TRY(p, hipMemcpyAsync(p->h_res, p->d_res, (size_t)n * sizeof(aacg_parse_result), hipMemcpyDeviceToHost, st), AACG_ERR_NO_DEVICE);
    P_TRY(p, hipMemcpyAsync((char*)p->h_res + (size_t)n * sizeof(aacg_parse_result), p->d_refused, 4, hipMemcpyDeviceToHost, st), AACG_ERR_NO_DEVICE);
    P_TRY(p, hipStreamSynchronize(st), AACG_ERR_NO_DEVICE);
    if (!direct) std::memcpy(pcm_out, p->h_pcm, pcm_bytes);
    if (results) std::memcpy(results, p->h_res, (size_t)n * sizeof(aacg_parse_result));
    if (n_refused) std::memcpy(n_refused, (char*)p->h_res + (size_t)n * sizeof(aacg_parse_result), 4);
    return AACG_OK;
}

}  // extern "C"
