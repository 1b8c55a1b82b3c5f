/*
 * aacg_parser.hip — the device front end behind include/aacgpu.h's aacg_parser_* / aacg_parse_* entry points:
 * the kernel (one lane per frame, aacg_parse.h) and its host plumbing.  Independent of aacg_engine: it shares
 * record formats with it, no state.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "aacg_parse.h"
#include "aacg_host.h"
#include "aacg_wait.h"

extern "C" DP_KERNEL(AACG_PARSE_WG_LARGE, 1)
void aacg_parse_frames(const aacg_parse_params P) { aacg_parse::parse_body(P); }

/* Lane order: a counting sort of the frames by length, longest first (8-byte buckets).  Three small launches. */
__device__ __forceinline__ uint32_t length_bucket(uint32_t bytes)
{
    const uint32_t k = bytes >> 3;
    return AACG_PARSE_BUCKETS - 1u - (k < AACG_PARSE_BUCKETS ? k : AACG_PARSE_BUCKETS - 1u);
}
/* one frame per thread; the buckets are counted in LDS first: frames cluster in a few hundred buckets, and 64 k global
 * atomics on those took 70 us per pass */
extern "C" __global__ __launch_bounds__(256)
void aacg_parse_order_count(const aacg_parse_frame* frames, uint32_t n, uint32_t* hist)
{
    __shared__ uint32_t cnt[AACG_PARSE_BUCKETS];
    for (uint32_t b = threadIdx.x; b < AACG_PARSE_BUCKETS; b += 256u) cnt[b] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) atomicAdd(&cnt[length_bucket(frames[i].byte_length)], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < AACG_PARSE_BUCKETS; b += 256u) if (cnt[b]) atomicAdd(&hist[b], cnt[b]);
}
extern "C" __global__ __launch_bounds__(AACG_PARSE_BUCKETS)
void aacg_parse_order_scan(uint32_t* hist)
{
    __shared__ uint32_t sum[AACG_PARSE_BUCKETS];
    const uint32_t t = threadIdx.x, mine = hist[t];
    sum[t] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < AACG_PARSE_BUCKETS; d <<= 1) {
        const uint32_t add = t >= d ? sum[t - d] : 0u;
        __syncthreads();
        sum[t] += add;
        __syncthreads();
    }
    hist[t] = sum[t] - mine;                               /* exclusive: where the bucket starts */
}
extern "C" __global__ __launch_bounds__(256)
void aacg_parse_order_fill(const aacg_parse_frame* frames, uint32_t n, uint32_t* next, uint32_t* order, uint32_t n_wg, uint32_t waves_per_wg)
{
    __shared__ uint32_t cnt[AACG_PARSE_BUCKETS];          /* the block's frames per bucket, then where its share starts */
    for (uint32_t b = threadIdx.x; b < AACG_PARSE_BUCKETS; b += 256u) cnt[b] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t bucket = 0, rank = 0;
    if (i < n) { bucket = length_bucket(frames[i].byte_length); rank = atomicAdd(&cnt[bucket], 1u); }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < AACG_PARSE_BUCKETS; b += 256u) if (cnt[b]) cnt[b] = atomicAdd(&next[b], cnt[b]);
    __syncthreads();
    if (i < n) {
        /* sorted position -> lane: piece c of 64 neighbours becomes wave c / n_wg of workgroup c % n_wg */
        const uint32_t pos = cnt[bucket] + rank, piece = pos >> 6;
        order[((piece % n_wg) * waves_per_wg + piece / n_wg) * 64u + (pos & 63u)] = i;
    }
}

/* Small batches (a resident pipeline's 4096 frames): everything in front of the parse kernel in ONE launch.  On a stream that
 * carries a batch through a dozen small steps every step costs its launch gap (5-10 us, and up to 60 when a transform launch of
 * the highest priority holds the dispatcher: profiles/r06_resident_budget.txt), and five of them were memsets, three the
 * counting sort.  Block 0 sorts (count, scan, place: the three kernels above in one workgroup's LDS); the other blocks clear
 * the regions the parser promises zeroed.  Regions are multiples of 16 bytes. */
#define AACG_PARSE_PREPARE_MAX 16384u
struct aacg_clear_regions { void* at[4]; unsigned long long n16[4]; };
extern "C" __global__ __launch_bounds__(1024)
void aacg_parse_prepare(const aacg_clear_regions C, const aacg_parse_frame* frames, uint32_t n, uint32_t* order, uint32_t lanes, uint32_t n_wg, uint32_t waves_per_wg)
{
    const uint32_t t = threadIdx.x;
    if (order && blockIdx.x == 0) {
        __shared__ uint32_t cnt[AACG_PARSE_BUCKETS], sum[AACG_PARSE_BUCKETS];
        for (uint32_t i = t; i < lanes; i += 1024u) order[i] = 0xffffffffu;        /* idle lanes */
        cnt[t] = 0;
        __syncthreads();
        for (uint32_t i = t; i < n; i += 1024u) atomicAdd(&cnt[length_bucket(frames[i].byte_length)], 1u);
        __syncthreads();
        const uint32_t mine = cnt[t];
        sum[t] = mine;
        __syncthreads();
        for (uint32_t d = 1; d < AACG_PARSE_BUCKETS; d <<= 1) {
            const uint32_t add = t >= d ? sum[t - d] : 0u;
            __syncthreads();
            sum[t] += add;
            __syncthreads();
        }
        cnt[t] = sum[t] - mine;                                                      /* where the bucket starts */
        __syncthreads();
        for (uint32_t i = t; i < n; i += 1024u) {
            const uint32_t pos = atomicAdd(&cnt[length_bucket(frames[i].byte_length)], 1u), piece = pos >> 6;
            order[((piece % n_wg) * waves_per_wg + piece / n_wg) * 64u + (pos & 63u)] = i;
        }
        return;
    }
    const uint32_t first = order ? 1u : 0u, blocks = gridDim.x - first;
    if (!blocks) return;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u zero = {0u, 0u, 0u, 0u};
    for (int r = 0; r < 4; r++) {
        v4u* d = (v4u*)C.at[r];
        const size_t n16 = (size_t)C.n16[r];
        for (size_t i = (size_t)(blockIdx.x - first) * 1024u + t; i < n16; i += (size_t)blocks * 1024u) d[i] = zero;
    }
}

struct aacg_parser {
    int device = 0;
    hipStream_t stream = nullptr;
    aacg_parse_tables* d_tab = nullptr;
    size_t lds_bytes = 0;
    uint32_t lut_words = 0;
    int n_cus = 256;
    /* device staging of aacg_parse_batch, grown on demand */
    uint32_t* d_order = nullptr;      /* lane order + the bucket counters behind it */
    size_t order_cap = 0;
    hipEvent_t order_free = nullptr;  /* recorded behind every launch: the scratch above may be rewritten after it */
    hipStream_t last_stream = nullptr;
    void* d_buf[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap[7] = {0, 0, 0, 0, 0, 0, 0};
    aacg_wait_policy wait;            /* aacg_parse_batch's host waits are bounded (aacg_wait.h): AACG_ERR_TIMEOUT */
    std::string err;
};

namespace {

int fail(aacg_parser* p, int rc, const std::string& m) { if (p) p->err = m; return rc; }
int hip_fail(aacg_parser* p, hipError_t e, const char* what) { return fail(p, AACG_ERR_NO_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); }
#define HIPCHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(p, e_, #call); } while (0)

int grow(aacg_parser* p, int i, size_t bytes)
{
    if (bytes <= p->cap[i]) return AACG_OK;
    if (p->d_buf[i]) (void)hipFree(p->d_buf[i]);
    p->d_buf[i] = nullptr; p->cap[i] = 0;
    if (hipMalloc(&p->d_buf[i], bytes) != hipSuccess) return fail(p, AACG_ERR_OUT_OF_MEMORY, "hipMalloc failed");
    p->cap[i] = bytes;
    return AACG_OK;
}

int launch(aacg_parser* p, aacg_parse_params& P, hipStream_t s)
{
    /* up to one small workgroup per CU: stage the frames in LDS; beyond that the rate matters more than the time per
     * frame, and 16 waves per CU reading in place overlap what one wave per SIMD cannot (AACG_PARSE_WG overrides) */
    const char* env = std::getenv("AACG_PARSE_WG");
    P.wg_threads = env ? (uint32_t)std::atoi(env) : (P.n_frames <= (uint32_t)p->n_cus * AACG_PARSE_WG_SMALL ? AACG_PARSE_WG_SMALL : P.n_frames <= (uint32_t)p->n_cus * 512u ? 512u : AACG_PARSE_WG_LARGE);
    if (P.wg_threads != AACG_PARSE_WG_SMALL && P.wg_threads != 512 && P.wg_threads != AACG_PARSE_WG_LARGE) return fail(p, AACG_ERR_INVALID_ARG, "AACG_PARSE_WG must be 256, 512 or 1024");
    while (AACG_PARSE_LDS_FIXED(p->lut_words, P.wg_threads) > p->lds_bytes) P.wg_threads /= 2;      /* very large tables */
    P.arena_bytes = (uint32_t)(p->lds_bytes - AACG_PARSE_LDS_FIXED(p->lut_words, P.wg_threads));
    const unsigned grid = (P.n_frames + P.wg_threads - 1) / P.wg_threads;
    /* Frames of similar length into the same wave, and long and short waves onto every CU alike (AACG_PARSE_SORT=0: table
     * order).  Measured with frame lengths spread 44..1186 bytes: 16 k frames 1.40 -> 0.89 ms, 64 k 1.36 -> 0.94. */
    static const bool sort_enabled = [] { const char* v = std::getenv("AACG_PARSE_SORT"); return !(v && v[0] == '0'); }();
    const bool sorted = sort_enabled && P.n_frames > 64u;
    const size_t lanes = (size_t)grid * P.wg_threads;
    if (sorted) {
        const size_t need = lanes + AACG_PARSE_BUCKETS;
        if (need > p->order_cap) {
            if (p->d_order) (void)hipFree(p->d_order);
            p->d_order = nullptr; p->order_cap = 0;
            if (hipMalloc((void**)&p->d_order, need * sizeof(uint32_t)) != hipSuccess) return fail(p, AACG_ERR_OUT_OF_MEMORY, "hipMalloc failed");
            p->order_cap = need;
        }
    }
    /* what the parser promises zeroed: the spectra (positions outside the coded bands), the band words (only the coded ones are
     * written), TNS records, and the unit records of refused frames and of element slots beyond a frame's count (never stale memory) */
    struct region { void* at; size_t bytes; };
    const region regions[4] = {
        { (P.options & AACG_PARSE_SKIP_ZERO_FILL) ? nullptr : (void*)P.q, (size_t)P.n_frames * P.max_channels * 1024u * sizeof(int16_t) },
        { (void*)P.meta, (size_t)P.n_frames * P.max_channels * sizeof(aacg_band_meta) },
        { (void*)P.tns, (size_t)P.n_frames * P.max_channels * sizeof(aacg_tns_info) },
        { (void*)P.units, (size_t)P.n_frames * P.max_units * sizeof(aacg_unit_desc) } };
    /* The lane-order scratch belongs to the parser, not to the launch: a launch on another stream first waits for the
     * previous launch's kernels (same stream: ordered anyway).  Two streams may therefore alternate on one parser. */
    if (p->last_stream && p->last_stream != s) HIPCHECK(hipStreamWaitEvent(s, p->order_free, 0));
    P.order = sorted ? p->d_order : nullptr;
    bool fused = P.n_frames <= AACG_PARSE_PREPARE_MAX;
    for (const region& r : regions) if (r.at && (((uintptr_t)r.at | r.bytes) & 15u)) fused = false;
    if (fused) {                                         /* one launch in front of the parse kernel (aacg_parse_prepare) */
        aacg_clear_regions C;
        size_t total16 = 0;
        for (int i = 0; i < 4; i++) { C.at[i] = regions[i].at; C.n16[i] = regions[i].at ? regions[i].bytes / 16u : 0u; total16 += (size_t)C.n16[i]; }
        unsigned blocks = (unsigned)((total16 + 4095u) / 4096u);                    /* four 16-byte stores per thread */
        blocks = (blocks < 1u ? 1u : blocks > 512u ? 512u : blocks) + (sorted ? 1u : 0u);
        hipLaunchKernelGGL(aacg_parse_prepare, dim3(blocks), dim3(1024), 0, s, C, P.frames, P.n_frames, sorted ? p->d_order : nullptr, (uint32_t)lanes, grid, P.wg_threads / 64u);
    } else {
        for (const region& r : regions) if (r.at) HIPCHECK(hipMemsetAsync(r.at, 0, r.bytes, s));
        if (sorted) {
            uint32_t* hist = p->d_order + lanes;
            const unsigned blocks = (P.n_frames + 255u) / 256u;
            HIPCHECK(hipMemsetAsync(p->d_order, 0xff, lanes * sizeof(uint32_t), s));
            HIPCHECK(hipMemsetAsync(hist, 0, AACG_PARSE_BUCKETS * sizeof(uint32_t), s));
            hipLaunchKernelGGL(aacg_parse_order_count, dim3(blocks), dim3(256), 0, s, P.frames, P.n_frames, hist);
            hipLaunchKernelGGL(aacg_parse_order_scan, dim3(1), dim3(AACG_PARSE_BUCKETS), 0, s, hist);
            hipLaunchKernelGGL(aacg_parse_order_fill, dim3(blocks), dim3(256), 0, s, P.frames, P.n_frames, hist, p->d_order, grid, P.wg_threads / 64u);
        }
    }
    hipLaunchKernelGGL(aacg_parse_frames, dim3(grid), dim3(P.wg_threads), p->lds_bytes, s, P);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipEventRecord(p->order_free, s));
    p->last_stream = s;
    return AACG_OK;
}

}  // namespace

extern "C" {

const char* aacg_parse_kernel_name(void) { return "aacg_parse_frames"; }
const char* aacg_parser_last_error(const aacg_parser* p) { return p ? p->err.c_str() : "null parser"; }

const char* aacg_parse_status_string(int status)
{
    static const char* const text[] = {
        "ok", "Insufficient data", "Invalid band type: 12", "Too many bands", "Scalefactor out of range",
        "Pulse tool not allowed in eight short sequence.", "Pulse SWB or offset out of range", "TODO: add pulse data",
        "TNS filter out of range", "Prediction not implemented.", "TODO: decode gain control/SSR", "TODO: PCE_ELEMENT",
        "maxSFB out of range", "Reserved ms mask type: 3", "Huffman: escape sequence too long",
        "more elements or channels in the frame than allowed for",
        "the frame's elements are not the ones its stream began with" };
    return status >= 0 && status < (int)(sizeof text / sizeof *text) ? text[status] : "unknown status";
}

int aacg_parser_create(int device_ordinal, int sample_index, const aacg_code_entry* entries, const uint32_t counts[12], aacg_parser** out)
{
    if (!out) return AACG_ERR_INVALID_ARG;
    *out = nullptr;
    aacg_parser* p = new aacg_parser;
    p->device = device_ordinal;
    std::vector<aacg_parse_tables> tab(1);
    int rc = aacg_parse_build_tables(sample_index, entries, counts, tab.data(), &p->err);
    if (rc == AACG_OK && hipSetDevice(device_ordinal) != hipSuccess) rc = fail(p, AACG_ERR_NO_DEVICE, "hipSetDevice failed");
    if (rc == AACG_OK) {
        p->lds_bytes = 160u * 1024u;                 /* one workgroup per CU: what the tables leave is the frames' staging arena */
        p->lut_words = tab[0].lut_words;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0) p->n_cus = prop.multiProcessorCount;
        if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&p->order_free, hipEventDisableTiming) != hipSuccess ||
            hipMalloc((void**)&p->d_tab, sizeof(aacg_parse_tables)) != hipSuccess ||
            hipMemcpy(p->d_tab, tab.data(), sizeof(aacg_parse_tables), hipMemcpyHostToDevice) != hipSuccess ||
            hipFuncSetAttribute((const void*)aacg_parse_frames, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes) != hipSuccess)
            rc = fail(p, AACG_ERR_NO_DEVICE, "HIP setup of the parser failed");
    }
    /* on failure the object is still returned, so that aacg_parser_last_error() can say why */
    *out = p;
    return rc;
}

void aacg_parser_destroy(aacg_parser* p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    /* its own stream and the last caller's stream it launched on, bounded; a device that does not answer keeps the memory */
    if ((p->stream && aacg_wait_stream(p->stream, p->wait) == hipErrorNotReady) ||
        (p->last_stream && p->last_stream != p->stream && aacg_wait_stream(p->last_stream, p->wait) == hipErrorNotReady)) {
        std::fprintf(stderr, "aacgpu: aacg_parser_destroy: the GPU did not answer within the wait limit — device memory of this parser is left allocated\n");
        delete p;
        return;
    }
    (void)hipGetLastError();
    if (p->order_free) (void)hipEventDestroy(p->order_free);
    if (p->d_order) (void)hipFree(p->d_order);
    for (int i = 0; i < 7; i++) if (p->d_buf[i]) (void)hipFree(p->d_buf[i]);
    if (p->d_tab) (void)hipFree(p->d_tab);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

int aacg_parse_device(aacg_parser* p, const void* d_bytes, const aacg_parse_frame* d_frames, uint32_t n_frames,
                      uint32_t max_units, uint32_t max_channels, uint32_t options,
                      aacg_unit_desc* d_units, int16_t* d_q, aacg_band_meta* d_meta, aacg_tns_info* d_tns,
                      aacg_parse_result* d_results, void* hip_stream)
{
    if (!p || !p->d_tab) return AACG_ERR_INVALID_ARG;
    if (!n_frames) return AACG_OK;
    if (!d_bytes || !d_frames || !d_units || !d_q || !d_meta || !d_results || !max_units || !max_channels || ((uintptr_t)d_bytes & 15u))
        return fail(p, AACG_ERR_INVALID_ARG, "aacg_parse_device: null or misaligned argument");
    HIPCHECK(hipSetDevice(p->device));
    aacg_parse_params P;
    P.bytes = (const uint32_t*)d_bytes; P.frames = d_frames; P.tab = p->d_tab; P.units = d_units; P.q = d_q; P.meta = d_meta;
    P.tns = d_tns; P.results = d_results; P.n_frames = n_frames; P.max_units = max_units; P.max_channels = max_channels; P.options = options;
    return launch(p, P, hip_stream ? (hipStream_t)hip_stream : p->stream);
}

int aacg_parse_batch(aacg_parser* p, const uint8_t* bytes, size_t n_bytes, const aacg_parse_frame* frames, uint32_t n_frames,
                     uint32_t max_units, uint32_t max_channels, uint32_t options,
                     aacg_unit_desc* units, int16_t* q, aacg_band_meta* meta, aacg_tns_info* tns, aacg_parse_result* results)
{
    if (!p || !p->d_tab) return AACG_ERR_INVALID_ARG;
    if (!n_frames) return AACG_OK;
    if (!bytes || !frames || !units || !q || !meta || !results || !max_units || !max_channels)
        return fail(p, AACG_ERR_INVALID_ARG, "aacg_parse_batch: null argument");
    for (uint32_t f = 0; f < n_frames; f++)
        if ((size_t)frames[f].byte_offset + frames[f].byte_length > n_bytes) return fail(p, AACG_ERR_INVALID_ARG, "frame " + std::to_string(f) + " lies outside the byte buffer");
    HIPCHECK(hipSetDevice(p->device));
    const size_t padded = (n_bytes + 15u) / 16u * 16u + AACG_PARSE_PAD_BYTES, blocks = (size_t)n_frames * max_channels;
    const size_t sizes[7] = { padded, n_frames * sizeof(aacg_parse_frame), (size_t)n_frames * max_units * sizeof(aacg_unit_desc),
                              blocks * 1024u * sizeof(int16_t), blocks * sizeof(aacg_band_meta), tns ? blocks * sizeof(aacg_tns_info) : 0,
                              n_frames * sizeof(aacg_parse_result) };
    for (int i = 0; i < 7; i++) { int rc = sizes[i] ? grow(p, i, sizes[i]) : AACG_OK; if (rc) return rc; }
    hipStream_t s = p->stream;
    const size_t tail = padded < AACG_PARSE_PAD_BYTES + 16u ? padded : AACG_PARSE_PAD_BYTES + 16u;      /* zeros behind the last byte */
    HIPCHECK(hipMemsetAsync((char*)p->d_buf[0] + padded - tail, 0, tail, s));
    HIPCHECK(hipMemcpyAsync(p->d_buf[0], bytes, n_bytes, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(p->d_buf[1], frames, sizes[1], hipMemcpyHostToDevice, s));
    aacg_parse_params P;
    P.bytes = (const uint32_t*)p->d_buf[0]; P.frames = (const aacg_parse_frame*)p->d_buf[1]; P.tab = p->d_tab;
    P.units = (aacg_unit_desc*)p->d_buf[2]; P.q = (int16_t*)p->d_buf[3]; P.meta = (aacg_band_meta*)p->d_buf[4];
    P.tns = tns ? (aacg_tns_info*)p->d_buf[5] : nullptr; P.results = (aacg_parse_result*)p->d_buf[6];
    P.n_frames = n_frames; P.max_units = max_units; P.max_channels = max_channels; P.options = options & ~AACG_PARSE_SKIP_ZERO_FILL;
    int rc = launch(p, P, s);
    if (rc) return rc;
    /* the kernel first, bounded: the copies into the caller's (pageable) memory below wait inside the runtime, and a device that
     * has just answered will carry them out */
    { const hipError_t st = aacg_wait_stream(s, p->wait);
      if (st == hipErrorNotReady) return fail(p, AACG_ERR_TIMEOUT, "aacg_parse_batch: the parse kernel did not complete within the wait limit (" + std::to_string(n_frames) + " frames, workgroups of " + std::to_string(P.wg_threads) + ")");
      HIPCHECK(st); }
    HIPCHECK(hipMemcpyAsync(units, p->d_buf[2], sizes[2], hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(q, p->d_buf[3], sizes[3], hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(meta, p->d_buf[4], sizes[4], hipMemcpyDeviceToHost, s));
    if (tns) HIPCHECK(hipMemcpyAsync(tns, p->d_buf[5], sizes[5], hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(results, p->d_buf[6], sizes[6], hipMemcpyDeviceToHost, s));
    { const hipError_t st = aacg_wait_stream(s, p->wait);
      if (st == hipErrorNotReady) return fail(p, AACG_ERR_TIMEOUT, "aacg_parse_batch: the copies back did not complete within the wait limit");
      HIPCHECK(st); }
    return AACG_OK;
}

int aacg_parser_set_wait_limit_ms(aacg_parser* p, uint32_t ms)
{
    if (!p || !ms) return AACG_ERR_INVALID_ARG;
    p->wait.limit_s = ms * 1e-3;
    return AACG_OK;
}

}  // extern "C"
