/*
 * pipe_drive.cpp — the headline route (aacg_decode_pipelined on BASELINE config 2) driven from a tight C loop: no Python, no
 * interpreter between two launches, so that a rocprofv3 kernel trace of THIS program reproduces what bench.py's own HIP events
 * measure (under the tracer a Python enqueue loop cannot stay fifteen launches ahead of the GPU and the
 * trace's row spacing came out at 13.1 us against the bench's 11.3-11.5).
 *
 *   tools/micro/pipe_drive [--launches N] [--pre-ms M] [--serial] [--i16] [--layout cpe|51] [--nbuf B] [--repeats R]
 *   rocprofv3 --kernel-trace --stats -d out -- tools/micro/pipe_drive          (the program directly behind `--`)
 *
 * One plan (256 streams x 16 frames, ONLY_LONG, KBD, maxSFB 49, M/S on even bands — aacgpu_workload.make_batch's config 2
 * restated with a generator of its own), B rotating input / output buffer sets (past the Infinity Cache), the same 256 streams
 * continued launch after launch.  What is being handed over between consecutive launches is the reference's overlap state,
 * src/filter_bank.js:105-118.  Prints one JSON line: host-clock and event-clock time per launch.
 *
 * Measurement aid: links the product library through its public C ABI (include/aacgpu.h, aacgpu_tools.h) and nothing else.
 */
#include <hip/hip_runtime.h>
#include <sys/resource.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/aacgpu.h"
#include "../../include/aacgpu_tools.h"

namespace {

struct rng_t {                      /* xoshiro256** — any generator will do, the workload's statistics are what matters */
    uint64_t s[4];
    explicit rng_t(uint64_t seed) { for (auto& w : s) { seed += 0x9E3779B97F4A7C15ull; uint64_t z = seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; w = z ^ (z >> 31); } }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() { const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45); return r; }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    int below(int n) { return (int)(next() % (uint64_t)n); }
};

#define CK(call) do { hipError_t rc_ = (call); if (rc_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(rc_)); std::exit(2); } } while (0)
#define AK(e, call) do { int rc_ = (call); if (rc_) { std::fprintf(stderr, "%s: %d %s\n", #call, rc_, (e) ? aacg_last_error(e) : ""); std::exit(2); } } while (0)

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

int main(int argc, char** argv)
{
    long launches = 30000, repeats = 5;
    double pre_ms = 300.0;
    bool serial = false, i16 = false, wide = false, mark_all = false;
    int nbuf = 8, wait_mode = -1;
    double spin_us = -1.0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--launches") launches = std::atol(val());
        else if (a == "--pre-ms") pre_ms = std::atof(val());
        else if (a == "--repeats") repeats = std::atol(val());
        else if (a == "--nbuf") nbuf = std::atoi(val());
        else if (a == "--serial") serial = true;
        else if (a == "--mark-all") mark_all = true;           /* a completion event bound to EVERY launch: what a kernel trace does to the queue */
        else if (a == "--wait-mode") wait_mode = std::atoi(val());   /* aacg_debug_set_wait_mode: 0 spin, 1 yield, 2 sleep, 3 blocking events */
        else if (a == "--spin-us") spin_us = std::atof(val());
        else if (a == "--i16") i16 = true;
        else if (a == "--layout") wide = std::string(val()) == "51";
        else { std::fprintf(stderr, "usage: pipe_drive [--launches N] [--pre-ms M] [--repeats R] [--serial] [--i16] [--layout cpe|51] [--nbuf B]\n"); return 2; }
    }
    const uint32_t S = 256, T = 16, F = S * T;
    /* element layout of a frame: one CPE (config 2), or CPE CPE CPE SCE = 7 channels (config 5's shape) */
    const std::vector<int> layout = wide ? std::vector<int>{2, 2, 2, 1} : std::vector<int>{2};
    uint32_t C = 0;
    for (int n : layout) C += (uint32_t)n;
    const uint32_t E = (uint32_t)layout.size();

    /* ---- the workload: unit records, quantised spectra, band words ---- */
    std::vector<aacg_unit_desc> units((size_t)F * E);
    std::memset(units.data(), 0, units.size() * sizeof(aacg_unit_desc));
    std::vector<int16_t> q((size_t)F * C * 1024);
    std::vector<aacg_band_meta> meta((size_t)F * C);
    std::memset(meta.data(), 0, meta.size() * sizeof(aacg_band_meta));
    rng_t rng(0xAAC00002ull);
    for (uint32_t f = 0; f < F; f++) {
        uint32_t chan = 0;
        for (uint32_t e = 0; e < E; e++) {
            aacg_unit_desc& u = units[(size_t)f * E + e];
            const int nc = layout[e];
            u.stream = f / T; u.pcm_offset = f * 1024u * C; u.channel = (uint16_t)chan; u.n_out_ch = (uint16_t)C; u.n_ch = (uint8_t)nc;
            u.flags = nc == 2 ? (AACG_UNIT_COMMON_WINDOW | AACG_UNIT_MASK_PRESENT) : 0;
            u.coef_offset = u.meta_offset = f * C + chan;
            for (int c = 0; c < nc; c++) {
                u.ch[c].window_sequence = AACG_ONLY_LONG_SEQUENCE; u.ch[c].window_shape = 1; u.ch[c].max_sfb = 49;
                u.ch[c].group_count = 1; u.ch[c].group_len[0] = 1;
                int16_t* qq = &q[((size_t)f * C + chan + (uint32_t)c) * 1024];
                for (int k = 0; k < 1024; k++) {          /* two-sided geometric magnitudes with scale 24 exp(-k / 180) */
                    const double lam = 24.0 * std::exp(-k / 180.0), mag = std::floor(-std::log(1.0 - rng.uniform()) * lam * 0.5);
                    const double v = (rng.next() & 1) ? mag : -mag;
                    qq[k] = (int16_t)std::max(-8190.0, std::min(8190.0, v));
                }
                aacg_band_meta& m = meta[(size_t)f * C + chan + (uint32_t)c];
                for (int b = 0; b < 49; b++) {            /* spectral codebooks 1..11, scalefactor index 248 +- 8, ms_used on even bands (left) */
                    uint16_t w = (uint16_t)((248 + rng.below(17) - 8) | ((1 + rng.below(11)) << AACG_META_BT_SHIFT));
                    if (nc == 2 && c == 0 && b % 2 == 0) w |= AACG_META_MS_USED;
                    m.band[b] = w;
                }
            }
            chan += (uint32_t)nc;
        }
    }

    /* ---- engine, plan, device buffers ---- */
    aacg_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = AACG_ABI_VERSION; cfg.device_ordinal = 0; cfg.sample_index = 3; cfg.max_streams = (int32_t)S; cfg.max_channels = (int32_t)C;
    cfg.max_batch_units = 0; cfg.input_kind = AACG_INPUT_QUANT_I16; cfg.output_kind = i16 ? AACG_OUTPUT_I16 : AACG_OUTPUT_F32;
    aacg_engine* eng = nullptr;
    AK(eng, aacg_create(&cfg, &eng));
    if (wait_mode >= 0) AK(eng, aacg_debug_set_wait_mode(eng, wait_mode, spin_us));
    aacg_plan* plan = nullptr;
    AK(eng, aacg_plan_create(eng, units.data(), (uint32_t)units.size(), &plan));
    const size_t q_bytes = q.size() * 2, pcm_bytes = (size_t)F * 1024 * C * (i16 ? 2 : 4);
    void* d_meta = nullptr;
    CK(hipMalloc(&d_meta, meta.size() * sizeof(aacg_band_meta)));
    CK(hipMemcpy(d_meta, meta.data(), meta.size() * sizeof(aacg_band_meta), hipMemcpyHostToDevice));
    std::vector<void*> d_q((size_t)nbuf), d_pcm((size_t)nbuf);
    std::vector<int16_t> rolled(q.size());
    for (int b = 0; b < nbuf; b++) {                      /* buffer set b: the same spectra rolled by 131 b blocks, every other set negated */
        const size_t rows = (size_t)F * C, shift = (size_t)(131 * b) % rows;
        for (size_t r = 0; r < rows; r++) {
            const int16_t* src = &q[r * 1024];
            int16_t* dst = &rolled[((r + shift) % rows) * 1024];
            if (b & 1) for (int k = 0; k < 1024; k++) dst[k] = (int16_t)-src[k]; else std::memcpy(dst, src, 2048);
        }
        CK(hipMalloc(&d_q[(size_t)b], q_bytes));
        CK(hipMemcpy(d_q[(size_t)b], rolled.data(), q_bytes, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_pcm[(size_t)b], pcm_bytes));
        CK(hipMemset(d_pcm[(size_t)b], 0, pcm_bytes));
    }
    CK(hipDeviceSynchronize());
    hipStream_t sstream = nullptr;                          /* --serial: aacg_decode_device on one stream */
    CK(hipStreamCreateWithFlags(&sstream, hipStreamNonBlocking));

    long n = 0;
    auto launch = [&](void* mark) {
        const size_t b = (size_t)(n++ % nbuf);
        int rc = serial ? aacg_decode_device(eng, plan, d_q[b], (const aacg_band_meta*)d_meta, d_pcm[b], sstream)
                        : (mark ? aacg_decode_pipelined_timed(eng, plan, d_q[b], (const aacg_band_meta*)d_meta, d_pcm[b], mark)
                                : aacg_decode_pipelined(eng, plan, d_q[b], (const aacg_band_meta*)d_meta, d_pcm[b]));
        if (rc) { std::fprintf(stderr, "launch %ld: %d %s\n", n - 1, rc, aacg_last_error(eng)); std::exit(2); }
    };
    auto drain = [&]() { AK(eng, aacg_synchronize(eng, sstream)); };

    /* steady clocks first (bench.py's preconditioning) */
    const double t_pre = now_s();
    long n_pre = 0;
    while ((now_s() - t_pre) * 1e3 < pre_ms) { for (int i = 0; i < 256; i++) { launch(nullptr); n_pre++; } drain(); }

    /* R timed regions of `launches` launches: host clock from the first enqueue to the drained pipeline, and — overlapped route —
     * the event clock between the completion of the last launch before the region and of the region's last launches (marks bound
     * to the dispatches, as bench.py's) */
    const int tails = 3;
    std::vector<double> host_us, event_us, cpu_frac;
    for (long r = 0; r < repeats; r++) {
        void* open_mark = nullptr;
        void* close_mark[tails] = {nullptr, nullptr, nullptr};
        if (!serial) { AK(eng, aacg_timer_create(&open_mark)); for (auto& m : close_mark) AK(eng, aacg_timer_create(&m)); }
        if (!serial) launch(open_mark);                     /* the launch in front of the region carries the opening mark */
        else { drain(); }
        /* --mark-all: marks of a small ring (a mark may be re-bound once its launch is complete: sixteen launches later it is) */
        static void* ring[64] = {};
        if (mark_all && !serial && !ring[0]) for (auto& m : ring) AK(eng, aacg_timer_create(&m));
        struct rusage ru0; getrusage(RUSAGE_SELF, &ru0);
        const double t0 = now_s();
        for (long i = 0; i < launches; i++) launch(!serial && launches - i <= tails ? close_mark[launches - 1 - i] : (mark_all && !serial ? ring[i & 63] : nullptr));
        const double t_enq = now_s();
        drain();
        const double t1 = now_s();
        struct rusage ru1; getrusage(RUSAGE_SELF, &ru1);
        cpu_frac.push_back(((ru1.ru_utime.tv_sec - ru0.ru_utime.tv_sec) + (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) + 1e-6 * ((ru1.ru_utime.tv_usec - ru0.ru_utime.tv_usec) + (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec))) / (t1 - t0));
        host_us.push_back((t1 - t0) * 1e6 / (double)launches);
        if (!serial) {
            float worst = 0;
            for (auto& m : close_mark) { float ms = 0; AK(eng, aacg_timer_elapsed_ms(open_mark, m, &ms)); worst = std::max(worst, ms); }
            event_us.push_back((double)worst * 1e3 / (double)launches);
            aacg_timer_destroy(open_mark);
            for (auto& m : close_mark) aacg_timer_destroy(m);
        }
        std::fprintf(stderr, "repeat %ld: %.3f us per launch by the host's clock (enqueue loop %.3f)%s\n", r, host_us.back(), (t_enq - t0) * 1e6 / (double)launches,
                     serial ? "" : (", " + std::to_string(event_us.back()) + " by the dispatches' events").c_str());
    }
    auto median = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    char kernels[256] = "";
    (void)aacg_plan_kernels_ex(eng, plan, serial ? 0 : 1, kernels, sizeof kernels);
    /* the last output is finite and not all zero */
    std::vector<float> tail(4096);
    CK(hipMemcpy(tail.data(), d_pcm[(size_t)((n - 1) % nbuf)], tail.size() * (i16 ? 2 : 4), hipMemcpyDeviceToHost));
    bool finite = true, nonzero = false;
    if (!i16) for (float v : tail) { finite = finite && std::isfinite(v); nonzero = nonzero || v != 0.0f; }
    else { const int16_t* w = (const int16_t*)tail.data(); for (size_t i = 0; i < tail.size(); i++) nonzero = nonzero || w[i] != 0; }
    const double abytes = ((2048.0 + 240.0) + (i16 ? 2048.0 : 4096.0) + 8192.0 / T) * F * C;
    const double us = serial ? median(host_us) : median(event_us);
    std::printf("{\"tool\": \"pipe_drive\", \"route\": \"%s\", \"kernel\": \"%s\", \"launches_per_repeat\": %ld, \"repeats\": %ld, \"preconditioning_launches\": %ld, "
                "\"us_per_launch_host_clock\": %.4f, \"us_per_launch_events\": %s, \"frames_per_s\": %.4g, \"algorithmic_bytes_per_launch\": %.0f, "
                "\"achieved_GBs\": %.1f, \"frac_of_8TBs\": %.4f, \"streams_used\": %d, \"concurrent\": %d, \"chained\": %llu, \"wait_mode\": %d, \"event_on_every_launch\": %s, "
                "\"host_cpu_fraction\": %.3f, \"output_ok\": %s}\n",
                serial ? "aacg_decode_device (launch behind launch)" : "aacg_decode_pipelined", kernels, launches, repeats, n_pre,
                median(host_us), serial ? "null" : std::to_string(median(event_us)).c_str(), F / (us * 1e-6), abytes, abytes / (us * 1e-6) / 1e9, abytes / (us * 1e-6) / 8e12,
                aacg_pipeline_streams_used(eng), aacg_pipeline_concurrent(eng), (unsigned long long)aacg_pipeline_chained(eng), wait_mode < 0 ? 2 : wait_mode, mark_all ? "true" : "false",
                median(cpu_frac), (finite && nonzero) ? "true" : "false");
    aacg_plan_destroy(plan);
    aacg_destroy(eng);
    return (finite && nonzero) ? 0 : 1;
}
