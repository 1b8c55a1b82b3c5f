/*
 * aacg_routes.cpp — the ONE route decision (aacg_routes.h): which launches a planned batch takes.  Plain C++: the engine
 * (aacg_engine.hip: launch_run executes the descriptor, aacg_plan_kernels prints it) and the lane emulator of tests/emu share
 * this function, so there is no second copy of the conditions anywhere.
 * What every route computes: reference src/decoder.js:218-248 (process) + src/filter_bank.js:88-204.
 */
#include "aacg_host.h"
#include "aacg_routes.h"

aacg_route aacg_pick_route(int input_kind, int output_kind, int debug_route, bool tracing, const aacg_plan_host& h, bool pipelined)
{
    aacg_route r;
    r.stage = AACG_STAGE_NONE; r.stage_quant = false; r.has_run = false; r.run_key = 0; r.rv = false;
    r.has_side = false; r.side_key = 0; r.side_first = false; r.couple_pcm = false; r.overlappable = false;
    const bool i16 = output_kind == AACG_OUTPUT_I16;
    bool quant = input_kind == AACG_INPUT_QUANT_I16;
    const bool stages = h.any_tns || (quant && h.any_pns);
    const unsigned nt = (h.wide_frames && !tracing) ? AACG_RK_NT : 0u;     /* batches of multichannel frames: non-temporal loads of the spectra */
    /* plain batches — no optional stage, no coupling element: with a chain longer than a run, or launched through the
     * pipeline, they take the rendezvous cut of their chains (AACG_DEBUG_ROUTE_RECOMPUTE: long chains the old way; serial launches only) */
    const bool plain = !h.any_cce && !stages && !h.runs_rv.empty();
    if (plain && (pipelined || (h.long_chains && !(debug_route & AACG_DEBUG_ROUTE_RECOMPUTE)))) {
        r.has_run = true;
        r.run_key = AACG_RK_RV | nt | (i16 ? AACG_RK_I16 : 0u) | (quant ? AACG_RK_QUANT : 0u);
        r.rv = true;
        r.overlappable = true;
        return r;
    }
    /* the optional stages inside the run kernel, same cut: one launch where a chain longer than a run took a staged route, and
     * launches through the pipeline overlap */
    if (stages && !i16 && !h.any_cce && !h.runs_rv.empty() && (pipelined || (h.long_chains && !(debug_route & AACG_DEBUG_ROUTE_RECOMPUTE)))) {
        r.has_run = true;
        r.run_key = AACG_RK_EX | AACG_RK_RV | (quant ? AACG_RK_QUANT : 0u);
        r.rv = true;
        r.overlappable = true;
        return r;
    }
    unsigned key = 0;
    if (h.any_cce_dependent) {
        /* coupling in the spectral domain: every unit's spectrum as f32 first, the coupling passes and the TNS filters on that */
        r.stage = AACG_STAGE_DEPENDENT_COUPLING; r.stage_quant = quant;
        quant = false;
    } else if (stages && !i16 && !h.any_cce && !h.needs_scratch) {
        key |= AACG_RK_EX;                                  /* optional stages inside the run kernel: one launch */
    } else if (stages) {
        r.stage = AACG_STAGE_SPECTRAL_EX; r.stage_quant = quant;    /* int16 PCM, coupling elements or double-duty runs: the stages as a launch of their own */
        quant = false;
    }
    const bool ex = (key & AACG_RK_EX) != 0;
    const unsigned q = quant ? AACG_RK_QUANT : 0u;
    const bool fused = h.fused_independent && !ex && !i16 && !(debug_route & AACG_DEBUG_ROUTE_UNFUSED_COUPLING);
    r.has_side = h.any_cce && !h.cce_runs.empty();
    r.side_key = q;                                         /* the plain kernel: the coupling elements' chains are ordinary single channels */
    r.has_run = !h.runs.empty();
    if (fused) {
        /* independent coupling in the targets' epilogues: the coupling elements go first (into the side buffer) */
        r.side_first = true;
        r.run_key = AACG_RK_CPL | nt | q;
        return r;
    }
    if (!ex) key |= (i16 ? AACG_RK_I16 : 0u) | (h.needs_scratch ? AACG_RK_DD : nt);
    r.run_key = key | q;
    r.couple_pcm = h.any_cce;
    return r;
}

std::string aacg_run_kernel_name(unsigned key)
{
    std::string s = std::string("aacg_imdct_run_") + ((key & AACG_RK_QUANT) ? "quant" : "f32");
    if (key & AACG_RK_EX)  s += "_ex";
    if (key & AACG_RK_DD)  s += "_dd";
    if (key & AACG_RK_CPL) s += "_cpl";
    if (key & AACG_RK_RV)  s += "_rv";
    if (key & AACG_RK_I16) s += "_i16";
    if (key & AACG_RK_NT)  s += "_nt";
    return s;
}

std::string aacg_route_names(const aacg_route& r, bool any_tns)
{
    std::string s;
    auto add = [&](const std::string& k) { if (!s.empty()) s += " + "; s += k; };
    if (r.stage == AACG_STAGE_DEPENDENT_COUPLING) {
        add(r.stage_quant ? "aacg_spectral_ex_quant" : "copy");
        add("aacg_couple_spec");
        if (any_tns) add("aacg_spectral_ex_f32");
    } else if (r.stage == AACG_STAGE_SPECTRAL_EX) add(r.stage_quant ? "aacg_spectral_ex_quant" : "aacg_spectral_ex_f32");
    if (r.has_side && r.side_first) add(aacg_run_kernel_name(r.side_key) + " (coupling elements)");
    if (r.has_run) add(aacg_run_kernel_name(r.run_key));
    if (r.has_side && !r.side_first) add(aacg_run_kernel_name(r.side_key) + " (coupling elements)");
    if (r.couple_pcm) add("aacg_couple_pcm");
    return s;
}

aacg_pipe_order aacg_pipeline_order(uint64_t n, int streams)
{
    const uint64_t S = (uint64_t)(streams < 1 ? 1 : (streams > AACG_PIPE_STREAMS ? AACG_PIPE_STREAMS : streams));
    const uint64_t depth = AACG_PIPE_DEPTH(S);
    static_assert(AACG_PIPE_DEPTH(1) / AACG_PIPE_MARK < AACG_PIPE_RING && AACG_PIPE_DEPTH(AACG_PIPE_STREAMS) >= AACG_PIPE_MARK, "the round waited for is a marked one whose events are still kept");
    static_assert(AACG_PIPE_STREAMS * (AACG_PIPE_MARK + AACG_PIPE_DEPTH(AACG_PIPE_STREAMS)) - AACG_PIPE_STREAMS <= AACG_OV_BUFFERS - 1 &&
                  2 * (AACG_PIPE_MARK + AACG_PIPE_DEPTH(2)) - 2 <= AACG_OV_BUFFERS - 1 && (AACG_PIPE_MARK + AACG_PIPE_DEPTH(1)) - 1 <= AACG_OV_BUFFERS - 1,
                  "a launch could start before one whose overlap buffers it reuses has finished");
    aacg_pipe_order o;
    const uint64_t round = n / S, pos = n % S;
    o.stream = (int)pos;
    o.marked = round % AACG_PIPE_MARK == 0;
    const uint64_t check = round - round % AACG_PIPE_MARK;            /* the most recent round that began with a wait */
    o.sync_round = (pos == 0 && round == check && check >= depth) ? (int64_t)(check - depth) : -1;
    o.complete_upto = check >= depth ? (int64_t)((check - depth) * S + S - 1) : -1;
    return o;
}

int aacg_pipeline_streams(const aacg_plan_host& h, unsigned run_key)
{
    if (h.runs_rv.size() > 512) return 2;                                   /* several rounds of workgroups per launch */
    if (run_key & AACG_RK_I16) return 2;                                    /* int16 PCM */
    if ((run_key & AACG_RK_EX) && (run_key & AACG_RK_QUANT)) return 2;      /* optional stages on the int16 seam */
    if (2u * h.short_units > h.units.size()) return 2;                      /* mostly frames of eight short windows */
    return AACG_PIPE_STREAMS;
}
