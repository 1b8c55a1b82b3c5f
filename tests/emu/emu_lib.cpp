/*
 * emu_lib.cpp — lane-lockstep CPU execution of the HIP kernels' source, for tests only.
 * Exposes: table build, the real host planner, and "launch" of the run / spectral kernels.
 */
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <algorithm>
#include <vector>

#include "../../aac.js_amd/csrc/aacg_kernels.h"
#include "../../aac.js_amd/csrc/aacg_kernels8.h"
#include "../../aac.js_amd/csrc/aacg_parse.h"
#include "../../aac.js_amd/csrc/aacg_host.h"

thread_local emu_lane_ctx g_emu;

namespace {

struct launch_arg {
    emu_lane_ctx ctx;
    const aacg_kparams* P;
    int kind;     /* 0 f32 run, 1 quant run, 2 spectral, 3/4 optional stages, 5/6 f32 / quant run with the optional stages inside, 7 front end */
    int n_units;
    const aacg_parse_params* PP;
    int out_kind;
    const aacg_couple_params* Q;
    const aacg_kparams8* P8;
    const aacg_rv_args* V;
};

void* lane_main(void* p)
{
    launch_arg* a = (launch_arg*)p;
    g_emu = a->ctx;
    /* the same dispatch as the engine's launch_run: double-duty variant / plain; kinds 3, 4: the optional-stage kernel */
    if (a->kind == 7) { aacg_parse::parse_body(*a->PP); return nullptr; }
    if (a->kind == 14) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_F32, false, false, false, true>(*a->P, a->V); return nullptr; }     /* aacg_imdct_run_f32_rv */
    if (a->kind == 15) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_F32, false, false, false, true>(*a->P, a->V); return nullptr; }    /* aacg_imdct_run_quant_rv */
    if (a->kind == 12) { imdct_run8_body<AACG_INPUT_SPEC_F32>(*a->P8); return nullptr; }      /* aacg_imdct_run8_f32 */
    if (a->kind == 13) { imdct_run8_body<AACG_INPUT_QUANT_I16>(*a->P8); return nullptr; }     /* aacg_imdct_run8_quant */
    if (a->kind == 8) { couple_spec_body(*a->Q, 4); return nullptr; }
    if (a->kind == 9) { couple_pcm_body(*a->Q, 4); return nullptr; }
    if (a->kind == 10) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_F32, false, false, true>(*a->P); return nullptr; }   /* aacg_imdct_run_f32_cpl */
    if (a->kind == 11) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_F32, false, false, true>(*a->P); return nullptr; }  /* aacg_imdct_run_quant_cpl */
    const bool dd = a->P->scratch != nullptr;
    if (a->out_kind == AACG_OUTPUT_I16 && a->kind == 0) { if (dd) imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_I16, true>(*a->P); else imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_I16>(*a->P); }
    else if (a->out_kind == AACG_OUTPUT_I16 && a->kind == 1) { if (dd) imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_I16, true>(*a->P); else imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_I16>(*a->P); }
    else if (a->kind == 0) { if (dd) imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_F32, true>(*a->P); else imdct_run_body<AACG_INPUT_SPEC_F32>(*a->P); }
    else if (a->kind == 1) { if (dd) imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_F32, true>(*a->P); else imdct_run_body<AACG_INPUT_QUANT_I16>(*a->P); }
    else if (a->kind == 5) imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_F32, false, true>(*a->P);
    else if (a->kind == 6) imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_F32, false, true>(*a->P);
    else if (a->kind == 3) spectral_ex_body<AACG_INPUT_QUANT_I16>(*a->P, a->n_units);
    else if (a->kind == 4) spectral_ex_body<AACG_INPUT_SPEC_F32>(*a->P, a->n_units);
    else                   spectral_body(*a->P, a->n_units);
    return nullptr;
}

int g_out_kind = AACG_OUTPUT_F32;          /* emu_set_output_kind: the next decodes store int16 PCM */
int g_unfused = 0;                         /* emu_set_unfused: independent coupling as the separate pass over the PCM even where the engine fuses it */
int g_run8 = 0;                            /* emu_set_run8: plain batches on the one-channel-per-wave kernels (the engine's opt-in route) */
int g_rv = 1;                              /* emu_set_rv: chains longer than a run through the run-to-run rendezvous (the engine's route; 2: blocks in reverse); 0: recomputed frames */
int g_staged = 0;                          /* emu_set_staged: optional stages as a launch of their own even where the engine would not */

void launch(const aacg_kparams& P, int kind, int grid, int waves, size_t lds_bytes, int n_units = 0, const aacg_parse_params* PP = nullptr,
            const aacg_couple_params* Q = nullptr, const aacg_kparams8* P8 = nullptr, const aacg_rv_args* V = nullptr)
{
    const int threads = waves * 64;
    std::vector<emu_wave> wv((size_t)waves);
    std::vector<launch_arg> args((size_t)threads);
    std::vector<pthread_t> tid((size_t)threads);
    unsigned char* lds = (unsigned char*)aligned_alloc(64, lds_bytes);
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 256 * 1024);
    for (int b = 0; b < grid; b++) {
        emu_block blk;
        blk.lds = lds;
        blk.lds_bytes = lds_bytes;
        blk.block_id = b;
        std::memset(lds, 0xff, lds_bytes);             /* NaN pattern: reads of unwritten LDS show up */
        pthread_barrier_init(&blk.bar, nullptr, (unsigned)threads);
        for (int w = 0; w < waves; w++) pthread_barrier_init(&wv[(size_t)w].bar, nullptr, 64);
        for (int t = 0; t < threads; t++) {
            args[(size_t)t].ctx = emu_lane_ctx{t & 63, t >> 6, &wv[(size_t)(t >> 6)], &blk};
            args[(size_t)t].P = &P;
            args[(size_t)t].kind = kind;
            args[(size_t)t].n_units = n_units;
            args[(size_t)t].PP = PP;
            args[(size_t)t].out_kind = g_out_kind;
            args[(size_t)t].Q = Q;
            args[(size_t)t].P8 = P8;
            args[(size_t)t].V = V;
            pthread_create(&tid[(size_t)t], &attr, lane_main, &args[(size_t)t]);
        }
        for (int t = 0; t < threads; t++) pthread_join(tid[(size_t)t], nullptr);
        for (int w = 0; w < waves; w++) pthread_barrier_destroy(&wv[(size_t)w].bar);
        pthread_barrier_destroy(&blk.bar);
    }
    pthread_attr_destroy(&attr);
    free(lds);
}

aacg_tables g_tab;
int g_tab_index = -1;
std::string g_err;

}  // namespace

extern "C" {

const char* emu_last_error() { return g_err.c_str(); }
void emu_set_staged(int on) { g_staged = on; }
void emu_set_run8(int on) { g_run8 = on; }
void emu_set_rv(int on) { g_rv = on; }
void emu_set_unfused(int on) { g_unfused = on; }
void emu_set_output_kind(int kind) { g_out_kind = kind; }       /* AACG_OUTPUT_*: the pcm buffer of later decodes is int16 */

int emu_get_windows(int sample_index, float* dst /* 1024+1024+128+128 */)
{
    aacg_tables t; aacg_host_windows w;
    int rc = aacg_build_tables(sample_index, &t, &w);
    if (rc) return rc;
    std::memcpy(dst, w.sine_long, 4096); std::memcpy(dst + 1024, w.kbd_long, 4096);
    std::memcpy(dst + 2048, w.sine_short, 512); std::memcpy(dst + 2176, w.kbd_short, 512);
    return 0;
}

int emu_get_iq_sf(float* iq /* 8192 */, float* sf /* 428 */)
{
    aacg_tables t;
    int rc = aacg_build_tables(3, &t, nullptr);
    if (rc) return rc;
    std::memcpy(iq, t.iq, sizeof t.iq); std::memcpy(sf, t.sf, 428 * 4);
    return 0;
}

/* plan only: returns number of runs (or <0); fills counts for inspection */
int emu_plan(const aacg_unit_desc* units, uint32_t n_units, int sample_index, int max_streams, int max_channels,
             const uint8_t* parity, aacg_run* runs_out, uint32_t runs_cap, int32_t* info /* zero_fill, n_chains, coef_blocks, meta_blocks */)
{
    aacg_plan_host ph;
    int rc = aacg_plan_build(units, n_units, sample_index, max_streams, max_channels, parity, &ph, &g_err);
    if (rc) return rc;
    if (runs_out) for (size_t i = 0; i < ph.runs.size() && i < runs_cap; i++) runs_out[i] = ph.runs[i];
    if (info) { info[0] = ph.zero_fill; info[1] = (int32_t)ph.chains.size(); info[2] = (int32_t)ph.coef_blocks; info[3] = (int32_t)ph.meta_blocks; }
    return (int)ph.runs.size();
}

/* the planner's run table for the one-channel-per-wave kernels: returns the number of runs (or < 0), *n_links = rendezvous cells */
int emu_plan8(const aacg_unit_desc* units, uint32_t n_units, int sample_index, int max_streams, int max_channels,
              aacg_run8* runs_out, uint32_t runs_cap, int32_t* n_links)
{
    aacg_plan_host ph;
    int rc = aacg_plan_build(units, n_units, sample_index, max_streams, max_channels, nullptr, &ph, &g_err);
    if (rc) return rc;
    for (size_t i = 0; i < ph.runs8.size() && i < runs_cap; i++) runs_out[i] = ph.runs8[i];
    if (n_links) *n_links = (int32_t)ph.n_links;
    return (int)ph.runs8.size();
}

/* host planner only: a plan for `first`, then aacg_plan_refresh_host with `next` (tns_spec: the engine's TNS mode) */
int emu_plan_refresh(const aacg_unit_desc* first, const aacg_unit_desc* next, uint32_t n_units, int sample_index, int max_streams,
                     int max_channels, int tns_spec)
{
    aacg_plan_host ph;
    std::vector<uint8_t> parity((size_t)max_streams * (size_t)max_channels, 0);
    int rc = aacg_plan_build(first, n_units, sample_index, max_streams, max_channels, parity.data(), &ph, &g_err);
    if (rc) return rc;
    return aacg_plan_refresh_host(&ph, next, n_units, sample_index, tns_spec != 0, &g_err);
}

/* full path: plan + "launch".  overlap_pool: [max_streams][max_channels][2][1024]; parity: [max_streams*max_channels], updated. */
int emu_decode_tns(int input_kind, int sample_index, int max_streams, int max_channels,
                   const aacg_unit_desc* units, uint32_t n_units, const void* coeffs, const aacg_band_meta* meta,
                   const aacg_tns_info* tns, uint32_t n_tns,
                   float* pcm, size_t n_pcm_floats, float* overlap_pool, uint8_t* parity);
int emu_decode_ex(int input_kind, int sample_index, int max_streams, int max_channels,
                  const aacg_unit_desc* units, uint32_t n_units, const void* coeffs, const aacg_band_meta* meta,
                  const aacg_tns_info* tns, uint32_t n_tns, int pns_mode,
                  float* pcm, size_t n_pcm_floats, float* overlap_pool, uint8_t* parity);
int emu_decode_cce(int input_kind, int sample_index, int max_streams, int max_channels,
                   const aacg_unit_desc* units, uint32_t n_units, const void* coeffs, const aacg_band_meta* meta,
                   const aacg_tns_info* tns, uint32_t n_tns, int pns_mode, const aacg_cce_info* cce, uint32_t n_cce,
                   float* pcm, size_t n_pcm_floats, float* overlap_pool, uint8_t* parity);

int emu_decode(int input_kind, int sample_index, int max_streams, int max_channels,
               const aacg_unit_desc* units, uint32_t n_units, const void* coeffs, const aacg_band_meta* meta,
               float* pcm, size_t n_pcm_floats, float* overlap_pool, uint8_t* parity)
{
    return emu_decode_tns(input_kind, sample_index, max_streams, max_channels, units, n_units, coeffs, meta,
                          nullptr, 0, pcm, n_pcm_floats, overlap_pool, parity);
}

/* tns != NULL: AACG_TNS_SPEC */
int emu_decode_tns(int input_kind, int sample_index, int max_streams, int max_channels,
                   const aacg_unit_desc* units, uint32_t n_units, const void* coeffs, const aacg_band_meta* meta,
                   const aacg_tns_info* tns, uint32_t n_tns,
                   float* pcm, size_t n_pcm_floats, float* overlap_pool, uint8_t* parity)
{
    return emu_decode_ex(input_kind, sample_index, max_streams, max_channels, units, n_units, coeffs, meta, tns, n_tns,
                         AACG_PNS_REFERENCE, pcm, n_pcm_floats, overlap_pool, parity);
}

int emu_decode_ex(int input_kind, int sample_index, int max_streams, int max_channels,
                  const aacg_unit_desc* units, uint32_t n_units, const void* coeffs, const aacg_band_meta* meta,
                  const aacg_tns_info* tns, uint32_t n_tns, int pns_mode,
                  float* pcm, size_t n_pcm_floats, float* overlap_pool, uint8_t* parity)
{
    return emu_decode_cce(input_kind, sample_index, max_streams, max_channels, units, n_units, coeffs, meta, tns, n_tns, pns_mode,
                          nullptr, 0, pcm, n_pcm_floats, overlap_pool, parity);
}

/* pns_mode == AACG_PNS_SPEC: batches with AACG_UNIT_HAS_PNS units take the engine's two-kernel route;
 * cce != NULL: AACG_CCE_SPEC, the engine's staged route (launch_run in aacg_engine.hip) */
int emu_decode_cce(int input_kind, int sample_index, int max_streams, int max_channels,
                   const aacg_unit_desc* units, uint32_t n_units, const void* coeffs, const aacg_band_meta* meta,
                   const aacg_tns_info* tns, uint32_t n_tns, int pns_mode, const aacg_cce_info* cce, uint32_t n_cce,
                   float* pcm, size_t n_pcm_floats, float* overlap_pool, uint8_t* parity)
{
    if (g_tab_index != sample_index) { int rc = aacg_build_tables(sample_index, &g_tab, nullptr); if (rc) return rc; g_tab_index = sample_index; }
    aacg_plan_host ph;
    int rc = aacg_plan_build(units, n_units, sample_index, max_streams, max_channels, parity, &ph, &g_err, tns, n_tns, cce, n_cce);
    if (rc) return rc;
    if (ph.pcm_floats > n_pcm_floats) { g_err = "pcm buffer too small"; return AACG_ERR_CAPACITY; }
    if (ph.zero_fill) std::memset(pcm, 0, n_pcm_floats * (g_out_kind == AACG_OUTPUT_I16 ? 2 : 4));
    aacg_kparams P;
    std::memset(&P, 0, sizeof P);
    P.units = ph.units.data(); P.runs = ph.runs.data(); P.coeffs = coeffs; P.meta = meta; P.pcm = pcm;
    P.overlap = overlap_pool; P.tab = &g_tab; P.flip = 0; P.n_runs = (int32_t)ph.runs.size();
    P.tns = ph.any_tns ? ph.tns.data() : nullptr;
    std::vector<float> scratch(ph.needs_scratch ? ph.runs.size() * AACG_SLOT_FLOATS : 1, 0.0f);
    P.scratch = ph.needs_scratch ? scratch.data() : nullptr;
    std::vector<float> spec;
    bool ex = false;
    static aacg_pns_tables pns_tab;
    if (ph.any_pns && (pns_mode != AACG_PNS_SPEC || input_kind != AACG_INPUT_QUANT_I16)) { g_err = "PNS unit in a batch without AACG_PNS_SPEC"; return AACG_ERR_UNSUPPORTED; }
    std::vector<float> side((size_t)ph.side_blocks * 1024u + 1, 0.0f);
    auto couple = [&](int point) {
        for (uint32_t r = 0; r < ph.couple_rounds; r++) {
            const uint32_t first = ph.couple_first[(size_t)point * ph.couple_rounds + r], last = ph.couple_first[(size_t)point * ph.couple_rounds + r + 1];
            if (last <= first) continue;
            aacg_couple_params Q;
            Q.jobs = ph.couple_jobs.data() + first; Q.n_jobs = (int32_t)(last - first); Q.units = ph.units.data(); Q.meta = meta; Q.tab = &g_tab;
            Q.gains = ph.gains.data(); Q.spec = spec.data(); Q.side = side.data(); Q.pcm = pcm; Q.reserved = 0;
            launch(P, point == AACG_CCE_AFTER_IMDCT ? 9 : 8, (Q.n_jobs + 3) / 4, 4, 64, 0, nullptr, &Q);
        }
    };
    if (ph.any_cce_dependent) {
        const bool quant = input_kind == AACG_INPUT_QUANT_I16;
        aacg_build_pns_tables(sample_index, &pns_tab);
        spec.assign((size_t)ph.coef_blocks * 1024u, 0.0f);
        aacg_kparams Q = P;
        Q.spec_out = spec.data(); Q.pns = &pns_tab; Q.tns = nullptr;
        if (quant) launch(Q, 3, (int)((n_units + AACG_WG_WAVES - 1) / AACG_WG_WAVES), AACG_WG_WAVES, (AACG_SPX_TAB_FLOATS + AACG_WG_WAVES * AACG_SPX_WAVE_FLOATS) * 4, (int)n_units);
        else std::memcpy(spec.data(), coeffs, spec.size() * sizeof(float));
        couple(AACG_CCE_BEFORE_TNS);
        if (ph.any_tns) {
            Q.coeffs = spec.data(); Q.meta = nullptr; Q.tns = ph.tns.data();
            launch(Q, 4, (int)((n_units + AACG_WG_WAVES - 1) / AACG_WG_WAVES), AACG_WG_WAVES, AACG_WG_WAVES * AACG_SPX_WAVE_FLOATS * 4, (int)n_units);
        }
        couple(AACG_CCE_AFTER_TNS);
        P.coeffs = spec.data(); P.meta = nullptr; P.tns = nullptr;
        input_kind = AACG_INPUT_SPEC_F32;
    } else if ((ph.any_pns || ph.any_tns) && g_out_kind == AACG_OUTPUT_F32 && !ph.any_cce && !ph.needs_scratch && !g_staged) {
        aacg_build_pns_tables(sample_index, &pns_tab);  /* the engine's one-launch route: optional stages inside the run kernel */
        P.pns = &pns_tab;
        ex = true;
    } else if (ph.any_pns || ph.any_tns) {              /* the engine's two-kernel route */
        const bool quant = input_kind == AACG_INPUT_QUANT_I16;
        aacg_build_pns_tables(sample_index, &pns_tab);
        spec.assign((size_t)ph.coef_blocks * 1024u, 0.0f);
        aacg_kparams Q = P;
        Q.spec_out = spec.data(); Q.pns = &pns_tab;
        launch(Q, quant ? 3 : 4, (int)((n_units + AACG_WG_WAVES - 1) / AACG_WG_WAVES), AACG_WG_WAVES,
               ((quant ? AACG_SPX_TAB_FLOATS : 0) + AACG_WG_WAVES * AACG_SPX_WAVE_FLOATS) * 4, (int)n_units);
        P.coeffs = spec.data(); P.meta = nullptr; P.tns = nullptr;
        input_kind = AACG_INPUT_SPEC_F32;
    }
    auto cce_filterbank = [&]() {
        aacg_kparams C = P;
        C.runs = ph.cce_runs.data(); C.n_runs = (int32_t)ph.cce_runs.size(); C.pcm = side.data(); C.scratch = nullptr;
        launch(C, input_kind == AACG_INPUT_QUANT_I16 ? 1 : 0, (int)ph.cce_runs.size(), AACG_WG_WAVES,
               input_kind == AACG_INPUT_QUANT_I16 ? AACG_LDS_BYTES_QUANT : AACG_LDS_BYTES_F32);
    };
    const bool fused = ph.fused_independent && !ex && g_out_kind == AACG_OUTPUT_F32 && !g_unfused;     /* the engine's launch_run */
    if (fused) {
        if (!ph.cce_runs.empty()) cce_filterbank();
        aacg_set_cpl(&P, ph.couple_jobs.data() + ph.fused_first, ph.gains.data(), side.data());
        if (!ph.runs.empty())
            launch(P, input_kind == AACG_INPUT_QUANT_I16 ? 11 : 10, (int)ph.runs.size(), AACG_WG_WAVES,
                   input_kind == AACG_INPUT_QUANT_I16 ? AACG_LDS_BYTES_QUANT : AACG_LDS_BYTES_F32);
    } else if (!ph.runs.empty() && ex)
        launch(P, input_kind == AACG_INPUT_QUANT_I16 ? 6 : 5, (int)ph.runs.size(), AACG_WG_WAVES,
               input_kind == AACG_INPUT_QUANT_I16 ? AACG_LDS_BYTES_QUANT_EX : AACG_LDS_BYTES_F32_EX);
    else if (!ph.runs.empty() && g_run8 && g_out_kind == AACG_OUTPUT_F32 && !ph.any_cce && !ph.any_tns && !ph.any_pns) {
        /* the engine's route for plain batches: the one-channel-per-wave kernels (aacg_kernels8.h); the workgroups run one after
         * the other here, in block order or — g_run8 == 2 — in reverse, so that both sides of every rendezvous arrive first once */
        static aacg_win8 win8;
        static unsigned long long epoch = 0;
        aacg_build_win8(&g_tab, &win8);
        std::vector<unsigned long long> rv_state((size_t)ph.n_links * AACG8_RV_STATE_WORDS + 1, 0x5a5a5a5a5a5a5a5aull);
        std::vector<float> rv_data((size_t)ph.n_links * AACG8_RV_DATA_FLOATS + 1, std::numeric_limits<float>::quiet_NaN());
        std::vector<aacg_run8> runs8 = ph.runs8;
        if (g_run8 == 2) std::reverse(runs8.begin(), runs8.end());
        aacg_kparams8 P8;
        std::memset(&P8, 0, sizeof P8);
        P8.units = ph.units.data(); P8.runs = runs8.data(); P8.coeffs = P.coeffs; P8.meta = P.meta; P8.pcm = pcm; P8.overlap = overlap_pool;
        P8.tab = &g_tab; P8.win = &win8; P8.rv_state = rv_state.data(); P8.rv_data = rv_data.data(); P8.epoch = ++epoch; P8.flip = 0;
        P8.n_runs = (int32_t)runs8.size();
        launch(P, input_kind == AACG_INPUT_QUANT_I16 ? 13 : 12, (int)runs8.size(), AACG_WG_WAVES,
               input_kind == AACG_INPUT_QUANT_I16 ? AACG8_LDS_BYTES(AACG8_TAB_QUANT_FLOATS) : AACG8_LDS_BYTES(AACG8_TAB_F32_FLOATS), 0, nullptr, nullptr, &P8);
    } else if (!ph.runs_rv.empty() && g_rv && g_out_kind == AACG_OUTPUT_F32 && !ph.any_cce && !ph.any_tns && !ph.any_pns) {
        /* the engine's route for plain batches with a chain longer than a run: every run 16 frames, a rendezvous between
         * consecutive runs (imdct_run_body<..., RV>); block order forward or — g_rv == 2 — reversed */
        static unsigned long long epoch = 1000;
        std::vector<unsigned long long> rv_state((size_t)ph.n_links_rv * AACG8_RV_STATE_WORDS + 1, 0x5a5a5a5a5a5a5a5aull);
        std::vector<float> rv_data((size_t)ph.n_links_rv * AACG8_RV_DATA_FLOATS + 1, std::numeric_limits<float>::quiet_NaN());
        std::vector<aacg_run> runs = ph.runs_rv;
        std::vector<aacg_rv_link> links = ph.links_rv;
        if (g_rv == 2) { std::reverse(runs.begin(), runs.end()); std::reverse(links.begin(), links.end()); }
        aacg_kparams R = P;
        R.runs = runs.data(); R.n_runs = (int32_t)runs.size(); R.scratch = nullptr;
        aacg_rv_args V;
        V.links = links.data(); V.state = rv_state.data(); V.data = rv_data.data(); V.epoch = ++epoch;
        launch(R, input_kind == AACG_INPUT_QUANT_I16 ? 15 : 14, (int)runs.size(), AACG_WG_WAVES,
               input_kind == AACG_INPUT_QUANT_I16 ? AACG_LDS_BYTES_QUANT : AACG_LDS_BYTES_F32, 0, nullptr, nullptr, nullptr, &V);
    } else if (!ph.runs.empty())
        launch(P, input_kind == AACG_INPUT_QUANT_I16 ? 1 : 0, (int)ph.runs.size(), AACG_WG_WAVES,
               input_kind == AACG_INPUT_QUANT_I16 ? AACG_LDS_BYTES_QUANT : AACG_LDS_BYTES_F32);
    if (ph.any_cce && !fused) {
        if (!ph.cce_runs.empty()) cce_filterbank();
        couple(AACG_CCE_AFTER_IMDCT);
    }
    for (auto& c : ph.chains)
        for (int k = 0; k < c.n_ch; k++) parity[(size_t)c.stream * (size_t)max_channels + c.channel + k] ^= 1;
    return AACG_OK;
}

int emu_spectral(int sample_index, const aacg_unit_desc* units, uint32_t n_units,
                 const void* coeffs, const aacg_band_meta* meta, float* spec_out)
{
    if (g_tab_index != sample_index) { int rc = aacg_build_tables(sample_index, &g_tab, nullptr); if (rc) return rc; g_tab_index = sample_index; }
    aacg_plan_host ph;
    int rc = aacg_plan_build(units, n_units, sample_index, 1 << 16, 8, nullptr, &ph, &g_err);
    if (rc) return rc;
    aacg_kparams P;
    std::memset(&P, 0, sizeof P);
    P.units = ph.units.data(); P.coeffs = coeffs; P.meta = meta; P.spec_out = spec_out; P.tab = &g_tab;
    launch(P, 2, (int)((n_units + AACG_WG_WAVES - 1) / AACG_WG_WAVES), AACG_WG_WAVES,
           (AACG_TAB_QUANT_FLOATS + AACG_WG_WAVES * 512) * 4, (int)n_units);
    return AACG_OK;
}

/* the device front end (aacg_parse.h) on host memory: same arguments as aacg_parse_batch */
int emu_parse(int sample_index, const aacg_code_entry* entries, const uint32_t* counts,
              const uint8_t* bytes, size_t n_bytes, const aacg_parse_frame* frames, uint32_t n_frames,
              uint32_t max_units, uint32_t max_channels, uint32_t options,
              aacg_unit_desc* units, int16_t* q, aacg_band_meta* meta, aacg_tns_info* tns, aacg_parse_result* results)
{
    static aacg_parse_tables tab;
    int rc = aacg_parse_build_tables(sample_index, entries, counts, &tab, &g_err);
    if (rc) return rc;
    std::vector<uint32_t> padded((n_bytes + 15) / 16 * 4 + AACG_PARSE_PAD_BYTES / 4 + 4, 0u);
    std::memcpy(padded.data(), bytes, n_bytes);
    const size_t blocks = (size_t)n_frames * max_channels;
    std::memset(units, 0, (size_t)n_frames * max_units * sizeof *units);
    std::memset(q, 0, blocks * 1024 * sizeof *q);
    std::memset(meta, 0, blocks * sizeof *meta);
    if (tns) std::memset(tns, 0, blocks * sizeof *tns);
    aacg_parse_params PP;
    PP.bytes = padded.data(); PP.frames = frames; PP.tab = &tab; PP.units = units; PP.q = q; PP.meta = meta; PP.tns = tns; PP.results = results;
    PP.n_frames = n_frames; PP.max_units = max_units; PP.max_channels = max_channels; PP.options = options;
    /* AACG_EMU_ARENA: a small staging arena sends most frames down the read-in-place path */
    PP.wg_threads = AACG_PARSE_WG_SMALL;
    /* the launcher's lane order: frames sorted by length (a host sort here, a counting sort on the device), the sorted
     * 64-frame pieces dealt out to the workgroups in turn; idle lanes carry 0xffffffff */
    const uint32_t n_wg = (n_frames + PP.wg_threads - 1) / PP.wg_threads, waves = PP.wg_threads / 64;
    std::vector<uint32_t> sorted(n_frames), order((size_t)n_wg * PP.wg_threads, 0xffffffffu);
    for (uint32_t i = 0; i < n_frames; i++) sorted[i] = i;
    std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t b) { return frames[a].byte_length > frames[b].byte_length; });
    for (uint32_t pos = 0; pos < n_frames; pos++) {
        const uint32_t piece = pos >> 6;
        order[((piece % n_wg) * waves + piece / n_wg) * 64u + (pos & 63u)] = sorted[pos];
    }
    PP.order = n_frames > 64 ? order.data() : nullptr;
    const size_t fixed = AACG_PARSE_LDS_FIXED(tab.lut_words, PP.wg_threads);
    const char* env = std::getenv("AACG_EMU_ARENA");
    PP.arena_bytes = env ? (uint32_t)std::atoi(env) : (uint32_t)(160 * 1024 - fixed);
    aacg_kparams none;
    std::memset(&none, 0, sizeof none);
    launch(none, 7, (int)((n_frames + PP.wg_threads - 1) / PP.wg_threads), (int)PP.wg_threads / 64,
           fixed + PP.arena_bytes, 0, &PP);
    return AACG_OK;
}

}  // extern "C"
