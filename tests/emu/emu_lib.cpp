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
#include "../../aac.js_amd/csrc/aacg_parse.h"
#include "../../aac.js_amd/csrc/aacg_host.h"
#include "../../aac.js_amd/csrc/aacg_routes.h"

thread_local emu_lane_ctx g_emu;

namespace {

struct launch_arg {
    emu_lane_ctx ctx;
    const aacg_kparams* P;
    int kind;     /* 1 run kernel (key = its switches, aacg_routes.h), 2 spectral, 3/4 optional stages (quant / f32), 7 front end, 8/9 coupling passes */
    unsigned key;
    int n_units;
    const aacg_parse_params* PP;
    const aacg_couple_params* Q;
    const aacg_rv_args* V;
};

/* the run kernel with these switches: the same instantiations the engine's translation units make (the non-temporal variants
 * load through the same emulated instruction, so AACG_RK_NT is dropped here) */
void run_kernel(unsigned key, const aacg_kparams& P, const aacg_rv_args* V)
{
    constexpr int Q = AACG_INPUT_QUANT_I16, F = AACG_INPUT_SPEC_F32, O16 = AACG_OUTPUT_I16, O32 = AACG_OUTPUT_F32;
    switch (key & ~(unsigned)AACG_RK_NT) {
    case 0:                                         imdct_run_body<F>(P); break;
    case AACG_RK_QUANT:                             imdct_run_body<Q>(P); break;
    case AACG_RK_DD:                                imdct_run_body<F, O32, true>(P); break;
    case AACG_RK_DD | AACG_RK_QUANT:                imdct_run_body<Q, O32, true>(P); break;
    case AACG_RK_I16:                               imdct_run_body<F, O16>(P); break;
    case AACG_RK_I16 | AACG_RK_QUANT:               imdct_run_body<Q, O16>(P); break;
    case AACG_RK_I16 | AACG_RK_DD:                  imdct_run_body<F, O16, true>(P); break;
    case AACG_RK_I16 | AACG_RK_DD | AACG_RK_QUANT:  imdct_run_body<Q, O16, true>(P); break;
    case AACG_RK_EX:                                imdct_run_body<F, O32, false, true>(P); break;
    case AACG_RK_EX | AACG_RK_QUANT:                imdct_run_body<Q, O32, false, true>(P); break;
    case AACG_RK_CPL:                               imdct_run_body<F, O32, false, false, true>(P); break;
    case AACG_RK_CPL | AACG_RK_QUANT:               imdct_run_body<Q, O32, false, false, true>(P); break;
    case AACG_RK_RV:                                imdct_run_body<F, O32, false, false, false, true>(P, V); break;
    case AACG_RK_RV | AACG_RK_QUANT:                imdct_run_body<Q, O32, false, false, false, true>(P, V); break;
    case AACG_RK_RV | AACG_RK_EX:                   imdct_run_body<F, O32, false, true, false, true>(P, V); break;
    case AACG_RK_RV | AACG_RK_EX | AACG_RK_QUANT:   imdct_run_body<Q, O32, false, true, false, true>(P, V); break;
    case AACG_RK_RV | AACG_RK_I16:                  imdct_run_body<F, O16, false, false, false, true>(P, V); break;
    case AACG_RK_RV | AACG_RK_I16 | AACG_RK_QUANT:  imdct_run_body<Q, O16, false, false, false, true>(P, V); break;
    default: std::abort();                          /* a route without a kernel */
    }
}

void* lane_main(void* p)
{
    launch_arg* a = (launch_arg*)p;
    g_emu = a->ctx;
    if (a->kind == 7) aacg_parse::parse_body(*a->PP);
    else if (a->kind == 1) run_kernel(a->key, *a->P, a->V);
    else if (a->kind == 8) couple_spec_body(*a->Q, 4);
    else if (a->kind == 9) couple_pcm_body(*a->Q, 4);
    else if (a->kind == 3) spectral_ex_body<AACG_INPUT_QUANT_I16>(*a->P, a->n_units);
    else if (a->kind == 4) spectral_ex_body<AACG_INPUT_SPEC_F32>(*a->P, a->n_units);
    else if (a->kind == 10) tns_matrices_body(a->P->tns, (double*)(void*)a->P->scratch, (uint32_t)a->n_units);
    else                   spectral_body(*a->P, a->n_units);
    return nullptr;
}

int g_out_kind = AACG_OUTPUT_F32;          /* emu_set_output_kind: the next decodes store int16 PCM */
int g_unfused = 0;                         /* emu_set_unfused: independent coupling as the separate pass over the PCM even where the engine fuses it */
int g_rv = 1;                              /* emu_set_rv: chains longer than a run through the run-to-run rendezvous (the engine's route; 2: blocks in reverse); 0: recomputed frames */
int g_staged = 0;                          /* emu_set_staged: optional stages as a launch of their own even where the engine would not */

/* one workgroup of a launch, its lanes as threads */
void run_block(const aacg_kparams& P, int kind, unsigned key, int block, int waves, size_t lds_bytes, int n_units = 0, const aacg_parse_params* PP = nullptr,
               const aacg_couple_params* Q = nullptr, const aacg_rv_args* V = nullptr)
{
    const int threads = waves * 64;
    std::vector<emu_wave> wv((size_t)waves);
    std::vector<launch_arg> args((size_t)threads);
    std::vector<pthread_t> tid((size_t)threads);
    unsigned char* lds = (unsigned char*)aligned_alloc(512, (lds_bytes + 511) & ~(size_t)511);
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 256 * 1024);
    emu_block blk;
    blk.lds = lds;
    blk.lds_bytes = lds_bytes;
    blk.block_id = block;
    std::memset(lds, 0xff, lds_bytes);             /* NaN pattern: reads of unwritten LDS show up */
    pthread_barrier_init(&blk.bar, nullptr, (unsigned)threads);
    for (int w = 0; w < waves; w++) pthread_barrier_init(&wv[(size_t)w].bar, nullptr, 64);
    for (int t = 0; t < threads; t++) {
        args[(size_t)t].ctx = emu_lane_ctx{t & 63, t >> 6, &wv[(size_t)(t >> 6)], &blk};
        args[(size_t)t].P = &P;
        args[(size_t)t].kind = kind;
        args[(size_t)t].key = key;
        args[(size_t)t].n_units = n_units;
        args[(size_t)t].PP = PP;
        args[(size_t)t].Q = Q;
        args[(size_t)t].V = V;
        pthread_create(&tid[(size_t)t], &attr, lane_main, &args[(size_t)t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(tid[(size_t)t], nullptr);
    for (int w = 0; w < waves; w++) pthread_barrier_destroy(&wv[(size_t)w].bar);
    pthread_barrier_destroy(&blk.bar);
    pthread_attr_destroy(&attr);
    free(lds);
}

/* a whole grid, workgroup after workgroup */
void launch(const aacg_kparams& P, int kind, unsigned key, int grid, int waves, size_t lds_bytes, int n_units = 0, const aacg_parse_params* PP = nullptr,
            const aacg_couple_params* Q = nullptr, const aacg_rv_args* V = nullptr)
{
    for (int b = 0; b < grid; b++) run_block(P, kind, key, b, waves, lds_bytes, n_units, PP, Q, V);
}

size_t run_lds_bytes(unsigned key)
{
    const bool quant = (key & AACG_RK_QUANT) != 0;
    if (key & AACG_RK_EX) return quant ? AACG_LDS_BYTES_QUANT_EX : AACG_LDS_BYTES_F32_EX;
    return quant ? AACG_LDS_BYTES_QUANT : AACG_LDS_BYTES_F32;
}

aacg_tables g_tab;
int g_tab_index = -1;
std::string g_err;

}  // namespace

extern "C" {

const char* emu_last_error() { return g_err.c_str(); }
void emu_set_staged(int on) { g_staged = on; }
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

/* full path: plan + "launch".  overlap_pool: [max_streams][max_channels][AACG_OV_BUFFERS][1024]; parity: [max_streams*max_channels] (0..AACG_OV_BUFFERS-1), updated. */
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

/* pns_mode == AACG_PNS_SPEC: batches with AACG_UNIT_HAS_PNS units take the optional-stage routes;
 * cce != NULL: AACG_CCE_SPEC.  The route is the engine's: aacg_pick_route (aacg_routes.cpp), executed here as launch_run
 * (aacg_engine.hip) executes it. */
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
    if (ph.any_pns && (pns_mode != AACG_PNS_SPEC || input_kind != AACG_INPUT_QUANT_I16)) { g_err = "PNS unit in a batch without AACG_PNS_SPEC"; return AACG_ERR_UNSUPPORTED; }
    aacg_route R = aacg_pick_route(input_kind, g_out_kind, (g_unfused ? AACG_DEBUG_ROUTE_UNFUSED_COUPLING : 0) | (g_rv ? 0 : AACG_DEBUG_ROUTE_RECOMPUTE), false, ph, false);
    if (g_staged && R.has_run && (R.run_key & AACG_RK_EX)) {     /* test switch: the optional stages as a launch of their own even where the engine runs them inside */
        R.stage = AACG_STAGE_SPECTRAL_EX; R.stage_quant = input_kind == AACG_INPUT_QUANT_I16;
        R.run_key = ph.needs_scratch ? AACG_RK_DD : 0; R.rv = false;
    }
    aacg_kparams P;
    std::memset(&P, 0, sizeof P);
    P.units = ph.units.data(); P.runs = ph.runs.data(); P.coeffs = coeffs; P.meta = meta; P.pcm = pcm;
    P.overlap = overlap_pool; P.tab = &g_tab; P.flip = 0; P.n_runs = (int32_t)ph.runs.size();
    P.tns = ph.any_tns ? ph.tns.data() : nullptr;
    std::vector<float> scratch(ph.needs_scratch ? ph.runs.size() * AACG_SLOT_FLOATS : 1, 0.0f);
    P.scratch = ph.needs_scratch ? scratch.data() : nullptr;
    /* the plan's TNS transition matrices, made once by a kernel of their own (aacg_tns_matrices): what the engine does when it
     * creates the plan; they ride in `scratch` for the launches that run filters (aacg_set_tns_m) */
    std::vector<double> tns_m(ph.any_tns ? ph.tns.size() * AACG_TNS_M_DOUBLES : 1, std::numeric_limits<double>::quiet_NaN());
    if (ph.any_tns) {
        aacg_kparams T = P;
        aacg_set_tns_m(&T, tns_m.data());
        launch(T, 10, 0, (int)((ph.tns.size() + AACG_WG_WAVES - 1) / AACG_WG_WAVES), AACG_WG_WAVES, 64, (int)ph.tns.size());
    }
    std::vector<float> spec;
    static aacg_pns_tables pns_tab;
    aacg_build_pns_tables(sample_index, &pns_tab);
    std::vector<float> side((size_t)ph.side_blocks * 1024u + 1, 0.0f);
    const int unit_blocks = (int)((n_units + AACG_WG_WAVES - 1) / AACG_WG_WAVES);
    auto couple = [&](int point) {
        for (uint32_t r = 0; r < ph.couple_rounds; r++) {
            const uint32_t first = ph.couple_first[(size_t)point * ph.couple_rounds + r], last = ph.couple_first[(size_t)point * ph.couple_rounds + r + 1];
            if (last <= first) continue;
            aacg_couple_params Q;
            Q.jobs = ph.couple_jobs.data() + first; Q.n_jobs = (int32_t)(last - first); Q.units = ph.units.data(); Q.meta = meta; Q.tab = &g_tab;
            Q.gains = ph.gains.data(); Q.spec = spec.data(); Q.side = side.data(); Q.pcm = pcm; Q.reserved = 0;
            launch(P, point == AACG_CCE_AFTER_IMDCT ? 9 : 8, 0, (Q.n_jobs + 3) / 4, 4, 64, 0, nullptr, &Q);
        }
    };
    if (R.rv) {
        /* plain batches with a chain longer than a run: every run 16 frames, a rendezvous between consecutive runs
         * (imdct_run_body<..., RV>); block order forward or — g_rv == 2 — reversed */
        static unsigned long long epoch = 1000;
        std::vector<unsigned long long> rv_state((size_t)ph.n_links_rv * AACG_RV_STATE_WORDS + 1, 0x5a5a5a5a5a5a5a5aull);
        std::vector<float> rv_data((size_t)ph.n_links_rv * AACG_RV_DATA_FLOATS + 1, std::numeric_limits<float>::quiet_NaN());
        std::vector<aacg_run> runs = ph.runs_rv;
        std::vector<aacg_rv_link> links = ph.links_rv;
        if (g_rv == 2) { std::reverse(runs.begin(), runs.end()); std::reverse(links.begin(), links.end()); }
        aacg_kparams K = P;
        K.runs = runs.data(); K.n_runs = (int32_t)runs.size(); K.scratch = nullptr;
        if (R.run_key & AACG_RK_EX) { K.pns = &pns_tab; if (ph.any_tns) aacg_set_tns_m(&K, tns_m.data()); }      /* optional stages inside the run kernel */
        aacg_rv_args V;
        std::memset(&V, 0, sizeof V);
        V.links = links.data(); V.state = rv_state.data(); V.data = rv_data.data(); V.epoch = ++epoch;
        launch(K, 1, R.run_key, (int)runs.size(), AACG_WG_WAVES, run_lds_bytes(R.run_key), 0, nullptr, nullptr, &V);
    } else {
        if (R.stage == AACG_STAGE_DEPENDENT_COUPLING) {
            spec.assign((size_t)ph.coef_blocks * 1024u, 0.0f);
            aacg_kparams Q = P;
            Q.spec_out = spec.data(); Q.pns = &pns_tab; Q.tns = nullptr;
            if (R.stage_quant) launch(Q, 3, 0, unit_blocks, AACG_WG_WAVES, (AACG_SPX_TAB_FLOATS + AACG_WG_WAVES * AACG_SPX_WAVE_FLOATS) * 4, (int)n_units);
            else std::memcpy(spec.data(), coeffs, spec.size() * sizeof(float));
            couple(AACG_CCE_BEFORE_TNS);
            if (ph.any_tns) {
                Q.coeffs = spec.data(); Q.meta = nullptr; Q.tns = ph.tns.data(); aacg_set_tns_m(&Q, tns_m.data());
                launch(Q, 4, 0, unit_blocks, AACG_WG_WAVES, AACG_WG_WAVES * AACG_SPX_WAVE_FLOATS * 4, (int)n_units);
            }
            couple(AACG_CCE_AFTER_TNS);
            P.coeffs = spec.data(); P.meta = nullptr; P.tns = nullptr;
        } else if (R.stage == AACG_STAGE_SPECTRAL_EX) {
            spec.assign((size_t)ph.coef_blocks * 1024u, 0.0f);
            aacg_kparams Q = P;
            Q.spec_out = spec.data(); Q.pns = &pns_tab; if (ph.any_tns) aacg_set_tns_m(&Q, tns_m.data());
            launch(Q, R.stage_quant ? 3 : 4, 0, unit_blocks, AACG_WG_WAVES, ((R.stage_quant ? AACG_SPX_TAB_FLOATS : 0) + AACG_WG_WAVES * AACG_SPX_WAVE_FLOATS) * 4, (int)n_units);
            P.coeffs = spec.data(); P.meta = nullptr; P.tns = nullptr;
        } else if (R.has_run && (R.run_key & AACG_RK_EX)) { P.pns = &pns_tab; if (ph.any_tns) aacg_set_tns_m(&P, tns_m.data()); }
        auto side_pass = [&]() {
            aacg_kparams C = P;
            C.runs = ph.cce_runs.data(); C.n_runs = (int32_t)ph.cce_runs.size(); C.pcm = side.data(); C.scratch = nullptr;
            launch(C, 1, R.side_key, (int)ph.cce_runs.size(), AACG_WG_WAVES, run_lds_bytes(R.side_key));
        };
        if (R.has_side && R.side_first) side_pass();
        if (R.has_run) {
            if (R.run_key & AACG_RK_CPL) aacg_set_cpl(&P, ph.couple_jobs.data() + ph.fused_first, ph.gains.data(), side.data());
            launch(P, 1, R.run_key, (int)ph.runs.size(), AACG_WG_WAVES, run_lds_bytes(R.run_key));
        }
        if (R.has_side && !R.side_first) side_pass();
        if (R.couple_pcm) couple(AACG_CCE_AFTER_IMDCT);
    }
    for (auto& c : ph.chains)
        for (int k = 0; k < c.n_ch; k++) { uint8_t& b = parity[(size_t)c.stream * (size_t)max_channels + c.channel + k]; b = (uint8_t)((b + 1) % AACG_OV_BUFFERS); }
    return AACG_OK;
}

/* aacg_decode_pipelined (aacg_engine.hip) for n_launches consecutive launches of ONE plan: launch j reads the coefficient
 * blocks at coeffs[j] / meta[j] and writes pcm[j]; the chains of neighbouring launches meet in the cross-launch cells.  The
 * emulator runs one workgroup at a time, in an order the engine's ordering rules allow: only neighbouring launches overlap
 * (launch j + 2 starts after launch j is complete).  order: 0 = launch after launch; 1 = within every pair (j, j + 1) the LATER
 * launch's workgroups first (every consumer leaves its first half, every producer finishes a frame of the next launch);
 * >= 2: the workgroups of all launches the engine's ordering rules allow to be in flight, interleaved at random (seed = order).
 * xl_cells: [max_streams][max_channels][3] records of 4 x u64 (aacg_xl_cell), xl_head: like the overlap pool; both kept by the
 * caller so that a sequence can be continued by a later call (first_epoch_in: 0 = its input state is complete; else the epoch
 * the previous call returned in *last_epoch, i.e. that call's last launch is "still in flight"). */
int emu_decode_pipelined(int input_kind, int sample_index, int max_streams, int max_channels,
                         const aacg_unit_desc* units, uint32_t n_units, int n_launches, const void* const* coeffs, const aacg_band_meta* const* meta,
                         float* const* pcm, size_t n_pcm_floats, float* overlap_pool, uint8_t* parity,
                         void* xl_cells, float* xl_head, int order, unsigned long long first_epoch_in, unsigned long long* last_epoch, int streams)
{
    if (g_tab_index != sample_index) { int rc = aacg_build_tables(sample_index, &g_tab, nullptr); if (rc) return rc; g_tab_index = sample_index; }
    aacg_plan_host ph;
    int rc = aacg_plan_build(units, n_units, sample_index, max_streams, max_channels, parity, &ph, &g_err);
    if (rc) return rc;
    if (ph.pcm_floats > n_pcm_floats) { g_err = "pcm buffer too small"; return AACG_ERR_CAPACITY; }
    const aacg_route R = aacg_pick_route(input_kind, AACG_OUTPUT_F32, 0, false, ph, true);
    if (!R.overlappable) { g_err = "not a plain batch"; return AACG_ERR_UNSUPPORTED; }
    static unsigned long long epoch = 5000;
    const int NS = streams > 0 ? streams : aacg_pipeline_streams(ph, R.run_key);          /* streams the sequence takes in turn (0: the engine's choice) */
    const size_t cells = (size_t)ph.n_links_rv;
    std::vector<unsigned long long> rv_state(AACG_PIPE_STREAMS * cells * AACG_RV_STATE_WORDS + 1, 0x5a5a5a5a5a5a5a5aull);
    std::vector<float> rv_data(AACG_PIPE_STREAMS * cells * AACG_RV_DATA_FLOATS + 1, std::numeric_limits<float>::quiet_NaN());
    std::vector<aacg_kparams> P((size_t)n_launches);
    std::vector<aacg_rv_args> V((size_t)n_launches);
    for (int j = 0; j < n_launches; j++) {
        if (ph.zero_fill) std::memset(pcm[j], 0, n_pcm_floats * 4);
        std::memset(&P[(size_t)j], 0, sizeof(aacg_kparams));
        aacg_kparams& p = P[(size_t)j];
        p.units = ph.units.data(); p.runs = ph.runs_rv.data(); p.coeffs = coeffs[j]; p.meta = meta ? meta[j] : nullptr; p.pcm = pcm[j];
        p.overlap = overlap_pool; p.tab = &g_tab; p.flip = j % AACG_OV_BUFFERS; p.n_runs = (int32_t)ph.runs_rv.size();
        aacg_rv_args& v = V[(size_t)j];
        std::memset(&v, 0, sizeof v);
        v.links = ph.links_rv.data();
        const size_t set = (size_t)aacg_pipeline_order((uint64_t)j, NS).stream;      /* launches in flight together never share a set of in-launch cells */
        v.state = rv_state.data() + set * cells * AACG_RV_STATE_WORDS;
        v.data = rv_data.data() + set * cells * AACG_RV_DATA_FLOATS;
        v.epoch = ++epoch;
        v.xl_cells = (aacg_xl_cell*)xl_cells; v.xl_head = xl_head;
        v.epoch_in = j ? V[(size_t)j - 1].epoch : first_epoch_in;
    }
    const int B = (int)ph.runs_rv.size();
    std::vector<std::pair<int, int>> sched;               /* (launch, block) */
    if (order == 0) {
        for (int j = 0; j < n_launches; j++) for (int b = 0; b < B; b++) sched.emplace_back(j, b);
    } else if (order == 1) {
        /* as late as the streams allow: of every NS consecutive launches the last one first */
        for (int j = 0; j < n_launches; j += NS)
            for (int k = NS - 1; k >= 0; k--)
                if (j + k < n_launches) for (int b = 0; b < B; b++) sched.emplace_back(j + k, b);
    } else {
        /* the engine's ordering rules, exactly (aacg_pipeline_order, aacg_routes.cpp): launch j goes to stream j mod
         * NS, so it starts after the launch NS before it is complete, and it does not exist before
         * the launches the host waits for are complete.  Among the launches those rules allow to run, the next workgroup is drawn at random, in each launch's
         * own shuffled block order */
        uint32_t rng = (uint32_t)order * 2654435761u + 12345u;
        auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return rng; };
        std::vector<std::vector<int>> left((size_t)n_launches);
        for (int j = 0; j < n_launches; j++) {
            for (int b = 0; b < B; b++) left[(size_t)j].push_back(b);
            for (int b = B - 1; b > 0; b--) std::swap(left[(size_t)j][(size_t)b], left[(size_t)j][next() % (uint32_t)(b + 1)]);
        }
        auto done = [&](int j) { return j < 0 || left[(size_t)j].empty(); };
        size_t remaining = (size_t)n_launches * (size_t)B;
        while (remaining) {
            std::vector<int> ready;
            for (int j = 0; j < n_launches; j++) {
                const aacg_pipe_order o = aacg_pipeline_order((uint64_t)j, NS);
                if (left[(size_t)j].empty() || !done(j - NS)) continue;
                bool known = true;                          /* the host enqueues it only after everything up to complete_upto is complete */
                for (int m = 0; m <= (int)o.complete_upto && known; m++) known = done(m);
                if (!known) continue;
                ready.push_back(j);
            }
            const int j = ready[next() % (uint32_t)ready.size()];
            sched.emplace_back(j, left[(size_t)j].back());
            left[(size_t)j].pop_back();
            remaining--;
        }
    }
    for (auto& jb : sched)
        run_block(P[(size_t)jb.first], 1, R.run_key, jb.second, AACG_WG_WAVES, run_lds_bytes(R.run_key), 0, nullptr, nullptr, &V[(size_t)jb.first]);
    for (auto& c : ph.chains)
        for (int k = 0; k < c.n_ch; k++) { uint8_t& b = parity[(size_t)c.stream * (size_t)max_channels + c.channel + k]; b = (uint8_t)((b + n_launches) % AACG_OV_BUFFERS); }
    if (last_epoch) *last_epoch = n_launches ? V[(size_t)n_launches - 1].epoch : first_epoch_in;
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
    launch(P, 2, 0, (int)((n_units + AACG_WG_WAVES - 1) / AACG_WG_WAVES), AACG_WG_WAVES,
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
    launch(none, 7, 0, (int)((n_frames + PP.wg_threads - 1) / PP.wg_threads), (int)PP.wg_threads / 64,
           fixed + PP.arena_bytes, 0, &PP);
    return AACG_OK;
}

}  // extern "C"
