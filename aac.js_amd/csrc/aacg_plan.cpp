/*
 * aacg_plan.cpp — batch planner: turns the host's list of parsed elements ("units", the
 * `elements` array of decoder.js:218 for many frames and streams) into workgroup runs.
 *
 * A chain is the sequence of frames of one element (same stream, same first channel, same
 * width) inside the batch; the only cross-frame dependency of the path is the overlap
 * buffer of each of its channels (filter_bank.js:38-41).  A chain is cut into runs: the
 * first holds up to AACG_RUN_W frames and takes its incoming tail from the overlap state;
 * every later run holds up to AACG_RUN_W - 1 frames and recomputes the tail of the frame
 * before it from that frame's spectrum (no inter-workgroup communication).
 */
#include "aacg_host.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>

namespace {

struct stream_state {
    bool     seen = false;
    uint32_t cur_off = 0;      /* pcm_offset of the frame being assembled */
    uint32_t frame = 0;        /* frame ordinal inside the batch */
    uint32_t mask = 0;         /* channels covered in the current frame */
    uint16_t n_out = 0;
    bool     holes = false;    /* some frame leaves a channel unwritten */
    bool     aligned = true;   /* every pcm_offset is a multiple of 4 floats */
    uint32_t n_chains = 0;
    /* the stream's first chains by channel: spares the per-unit map lookup (map nodes do not move) */
    struct open_chain* chain[4] = {nullptr, nullptr, nullptr, nullptr};
    uint16_t chain_channel[4] = {0, 0, 0, 0};
};

struct open_chain {
    std::vector<int32_t> units;
    uint32_t last_frame;
    uint8_t  n_ch;
    bool     is_cce = false;   /* an independently switched coupling element: filterbank output to the side buffer */
};

int fail(std::string* err, int code, const char* fmt, long a = 0, long b = 0, long c = 0)
{
    if (err) {
        char buf[256];
        std::snprintf(buf, sizeof buf, fmt, a, b, c);
        *err = buf;
    }
    return code;
}

}  // namespace

/* TNS_MAX_BANDS_1024 / _128 (tns.js:65-66; ISO/IEC 14496-3 Table 4.138) by sampleIndex */
static const uint8_t kTnsMaxBandsLong[13]  = {31, 31, 34, 40, 42, 51, 46, 46, 42, 42, 42, 39, 39};
static const uint8_t kTnsMaxBandsShort[13] = {9, 9, 10, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14};

int aacg_tns_prepare(int sample_index, const aacg_chan_info* info, const aacg_tns_info* in, aacg_dev_tns* out)
{
    std::memset(out, 0, sizeof *out);
    const bool is_short = info->window_sequence == AACG_EIGHT_SHORT_SEQUENCE;
    int swb[64];
    const int swb_count = aacg_swb_offsets(sample_index, is_short ? 0 : 1, swb);
    /* tns.js:106 intends min(maxBands, maxSFB); the short-window table is the documented deviation of SPEC mode */
    const int max_bands = is_short ? kTnsMaxBandsShort[sample_index] : kTnsMaxBandsLong[sample_index];
    const int mmm = std::min<int>(max_bands, info->max_sfb);
    const int n_win = is_short ? 8 : 1;
    for (int w = 0; w < n_win; w++) {
        int bottom = swb_count;                                      /* tns.js:112 */
        const int nf = in->n_filt[w];
        if (nf > (is_short ? 1 : 3)) return AACG_ERR_INVALID_ARG;
        for (int f = 0; f < nf; f++) {
            const aacg_tns_filter& tf = in->filt[is_short ? w : f];
            const int slot = is_short ? w : f;
            const int top = bottom;                                  /* tns.js:121-123 */
            bottom = std::max(0, top - (int)tf.length);
            const int order = tf.order;
            if (order == 0) continue;
            /* AAC-LC limits: 12 for long windows, 7 for short ones (3-bit field, tns.js:47) */
            if (order > (is_short ? 8 : AACG_TNS_MAX_ORDER)) return AACG_ERR_UNSUPPORTED;
            float lpc[AACG_TNS_MAX_ORDER];
            for (int i = 0; i < order; i++) {                        /* tns.js:128-140, Float32Array stores */
                const float r = -tf.coef[i];
                lpc[i] = r;
                for (int j = 0, len = (i + 1) >> 1; j < len; j++) {
                    const float fj = lpc[j], b = lpc[i - 1 - j];
                    lpc[j] = (float)((double)fj + (double)r * (double)b);
                    lpc[i - 1 - j] = (float)((double)b + (double)r * (double)fj);
                }
            }
            int start = swb[std::min(bottom, mmm)];                  /* tns.js:142-152 */
            const int end = swb[std::min(top, mmm)];
            const int size = end - start;
            if (size <= 0) continue;
            int inc = 1;
            if (tf.direction) { inc = -1; start = end - 1; }
            out->start[slot] = start + w * 128;
            out->size[slot] = size;
            out->inc[slot] = inc;
            out->order[slot] = order;
            for (int i = 0; i < order; i++) out->lpc[slot][i] = lpc[i];
        }
    }
    return AACG_OK;
}

/* the window fields of one channel as the kernels rely on them (ics.js:279-314); 0 or an error text */
static const char* check_chan(const aacg_chan_info& ci, int n_long, int n_short)
{
    if (ci.window_sequence > 3 || ci.window_shape > 1 || ci.window_shape_prev > 1) return "bad window fields";
    if (ci.window_sequence == AACG_EIGHT_SHORT_SEQUENCE) {
        int sum = 0;
        if (ci.group_count < 1 || ci.group_count > 8) return "group_count";
        for (int g = 0; g < ci.group_count; g++) sum += ci.group_len[g];
        if (sum != 8) return "group lengths do not sum to 8";
        if (ci.max_sfb > n_short) return "max_sfb (short)";
    } else {
        if (ci.group_count != 1 || ci.group_len[0] != 1) return "long window needs one group of one";
        if (ci.max_sfb > n_long) return "max_sfb (long)";
    }
    if ((int)ci.group_count * (int)ci.max_sfb > AACG_MAX_SECTIONS) return "more than 120 bands";
    return nullptr;
}

/* group-of-window map: 4 bits per window (ics.js:288-296 grouping) */
static uint32_t group_map(const aacg_chan_info& ci)
{
    uint32_t gmap = 0;
    if (ci.window_sequence == AACG_EIGHT_SHORT_SEQUENCE) {
        int w = 0;
        for (int g = 0; g < ci.group_count; g++)
            for (int k = 0; k < ci.group_len[g] && w < 8; k++, w++) gmap |= (uint32_t)g << (4 * w);
    }
    return gmap;
}


/* Every run table's per-wave copies of where its units' spectra lie (aacg_run.wave_*): after the tables are made, and again
 * whenever unit records are rewritten with other block offsets. */
void aacg_plan_fill_run_waves(aacg_plan_host* h)
{
    for (std::vector<aacg_run>* tab : {&h->runs, &h->runs_rv, &h->cce_runs})
        for (aacg_run& r : *tab) {
            r.wave_nch = 0;
            for (int w = 0; w < AACG_RUN_W; w++) {
                const int32_t ui = aacg_run_wave_unit(r, w);
                const aacg_unit_desc& d = h->units[ui >= 0 ? (size_t)ui : 0].d;
                r.wave_unit[w] = ui >= 0 ? ui : 0;
                r.wave_coef[w] = d.coef_offset;
                r.wave_meta[w] = d.meta_offset;
                r.wave_nch |= (uint32_t)(d.n_ch & 3) << (2 * w);
            }
        }
}

/* A kept plan takes the next batch's unit records: same streams, frames, elements and PCM positions (that is what the run
 * tables were built from), new window info, flags and block offsets.  What the host pays per batch is this loop and a
 * copy of the records instead of aacg_plan_build (27 ns per unit) — for callers that keep spectra on the device and
 * parse on the host.  Plans with TNS records, noise bands or coupling elements are built per batch. */
int aacg_plan_refresh_host(aacg_plan_host* h, const aacg_unit_desc* units, uint32_t n_units, int sample_index, bool tns_spec, std::string* err)
{
    int swb[64];
    const int n_long = aacg_swb_offsets(sample_index, 1, swb), n_short = aacg_swb_offsets(sample_index, 0, swb);
    if (n_units != h->units.size()) return fail(err, AACG_ERR_LAYOUT_CHANGE, "the plan was built for %ld units, the batch has %ld", (long)h->units.size(), (long)n_units);
    if (h->any_tns || h->any_pns || h->any_cce) return fail(err, AACG_ERR_UNSUPPORTED, "plans with TNS records, noise bands or coupling elements are built per batch");
    uint32_t coef_blocks = 0, meta_blocks = 0;
    for (uint32_t i = 0; i < n_units; i++) {
        const aacg_unit_desc& u = units[i];
        const aacg_unit_desc& was = h->units[i].d;
        if (u.stream != was.stream || u.pcm_offset != was.pcm_offset || u.channel != was.channel || u.n_out_ch != was.n_out_ch || u.n_ch != was.n_ch)
            return fail(err, AACG_ERR_LAYOUT_CHANGE, "unit %ld: stream / PCM position / channels differ from the plan's", i);
        if (u.flags & (AACG_UNIT_CCE | AACG_UNIT_HAS_PNS))
            return fail(err, AACG_ERR_LAYOUT_CHANGE, "unit %ld: a coupling element or noise bands: the route changes, plan anew", i);
        /* an AACG_TNS_SPEC engine runs the filters of a frame that carries TNS side info: such a batch needs a plan with TNS
         * records (aacg_plan_create_tns) — clearing the flag below would decode it without them, wrong PCM and no error */
        for (int c = 0; tns_spec && c < u.n_ch; c++)
            if (u.ch[c].flags & AACG_CHAN_TNS_PRESENT)
                return fail(err, AACG_ERR_LAYOUT_CHANGE, "unit %ld: TNS side info on an AACG_TNS_SPEC engine: the route changes, plan anew with the TNS records", i);
        for (int c = 0; c < u.n_ch; c++)
            if (const char* why = check_chan(u.ch[c], n_long, n_short)) { if (err) *err = why; return AACG_ERR_INVALID_ARG; }
        if (u.coef_offset > UINT32_MAX - 2u || u.meta_offset > UINT32_MAX - 2u)
            return fail(err, AACG_ERR_INVALID_ARG, "unit %ld: coefficient / meta offset out of range", i);
        coef_blocks = std::max(coef_blocks, u.coef_offset + u.n_ch);
        meta_blocks = std::max(meta_blocks, u.meta_offset + u.n_ch);
    }
    bool moved = false;
    h->short_units = 0;
    for (uint32_t i = 0; i < n_units; i++) {               /* nothing is touched before the whole batch has passed */
        aacg_dev_unit& du = h->units[i];
        if (units[i].ch[0].window_sequence == AACG_EIGHT_SHORT_SEQUENCE || (units[i].n_ch == 2 && units[i].ch[1].window_sequence == AACG_EIGHT_SHORT_SEQUENCE)) h->short_units++;
        moved = moved || du.d.coef_offset != units[i].coef_offset || du.d.meta_offset != units[i].meta_offset;
        du.d = units[i];
        for (int c = 0; c < 2; c++) {
            du.gmap[c] = c < du.d.n_ch ? group_map(du.d.ch[c]) : 0;
            du.d.ch[c].flags &= (uint8_t)~AACG_CHAN_TNS_PRESENT;                  /* no TNS records in such a plan */
        }
    }
    h->coef_blocks = coef_blocks; h->meta_blocks = meta_blocks;
    /* the run tables carry copies of the block offsets (aacg_run.wave_coef / wave_meta): a batch laid out otherwise takes new ones */
    h->runs_moved = moved;
    if (moved) aacg_plan_fill_run_waves(h);
    return AACG_OK;
}

int aacg_plan_build(const aacg_unit_desc* units, uint32_t n_units, int sample_index,
                    int max_streams, int max_channels, const uint8_t* parity,
                    aacg_plan_host* out, std::string* err,
                    const aacg_tns_info* tns, uint32_t n_tns,
                    const aacg_cce_info* cce, uint32_t n_cce)
{
    int swb[64];
    const int n_long = aacg_swb_offsets(sample_index, 1, swb);
    const int n_short = aacg_swb_offsets(sample_index, 0, swb);
    if (!n_long) return fail(err, AACG_ERR_INVALID_ARG, "sample_index %ld out of range", sample_index);

    *out = aacg_plan_host();
    out->units.reserve(n_units);
    std::vector<stream_state> st((size_t)max_streams);
    std::map<uint64_t, open_chain> open;                     /* key: stream << 16 | channel */

    auto close_frame = [&](stream_state& s) {
        if (s.seen && s.mask != ((1u << s.n_out) - 1u)) { out->zero_fill = true; s.holes = true; }
    };

    uint32_t n_wide = 0;                                    /* units of frames with more than two channels */
    std::vector<uint32_t> frame_of(n_units, 0);             /* frame ordinal of every unit inside its stream's batch */
    for (uint32_t i = 0; i < n_units; i++) {
        const aacg_unit_desc& u = units[i];
        if (u.flags & AACG_UNIT_HAS_PNS) out->any_pns = true;
        if (u.n_out_ch > 2) n_wide++;
        if (u.ch[0].window_sequence == AACG_EIGHT_SHORT_SEQUENCE || (u.n_ch == 2 && u.ch[1].window_sequence == AACG_EIGHT_SHORT_SEQUENCE)) out->short_units++;
        if (u.n_ch < 1 || u.n_ch > 2) return fail(err, AACG_ERR_INVALID_ARG, "unit %ld: n_ch %ld", i, u.n_ch);
        if ((int)u.stream >= max_streams) return fail(err, AACG_ERR_CAPACITY, "unit %ld: stream %ld >= max_streams", i, u.stream);
        const bool is_cce = (u.flags & AACG_UNIT_CCE) != 0;
        if (u.n_out_ch < 1 || u.n_out_ch > max_channels || (!is_cce && u.channel + u.n_ch > u.n_out_ch))
            return fail(err, AACG_ERR_INVALID_ARG, "unit %ld: channel %ld does not fit %ld output channels", i, u.channel, u.n_out_ch);
        if (is_cce) {
            /* cce.js:25-31: one channel of its own, beyond the output channels; the reference parses and ignores it */
            if (!cce) return fail(err, AACG_ERR_UNSUPPORTED, "unit %ld is a coupling channel element: aac.js never applies them (AACG_CCE_SPEC engines do)", i);
            if (u.n_ch != 1 || u.reserved1 >= n_cce || u.channel < u.n_out_ch || (int)u.channel >= max_channels)
                return fail(err, AACG_ERR_INVALID_ARG, "unit %ld: a coupling element is one channel at a stream channel beyond the output channels", i);
            const aacg_cce_info& ci = cce[u.reserved1];
            if (ci.coupling_point > AACG_CCE_AFTER_IMDCT || ci.n_targets > AACG_CCE_MAX_TARGETS)
                return fail(err, AACG_ERR_INVALID_ARG, "unit %ld: malformed aacg_cce_info", i);
            for (int t = 0; t < ci.n_targets; t++)
                if (ci.target[t].channel >= u.n_out_ch || ci.target[t].gain_list >= AACG_CCE_MAX_TARGETS)
                    return fail(err, AACG_ERR_INVALID_ARG, "unit %ld: coupling target %ld out of range", i, t);
            out->any_cce = true;
            if (ci.coupling_point != AACG_CCE_AFTER_IMDCT) out->any_cce_dependent = true;
        }
        for (int c = 0; c < u.n_ch; c++)
            if (const char* why = check_chan(u.ch[c], n_long, n_short)) {
                if (err) { char buf[160]; std::snprintf(buf, sizeof buf, "unit %ld ch %d: %s", (long)i, c, why); *err = buf; }
                return AACG_ERR_INVALID_ARG;
            }

        stream_state& s = st[u.stream];
        if (!s.seen) { s.seen = true; s.cur_off = u.pcm_offset; s.frame = 0; s.mask = 0; s.n_out = u.n_out_ch; }
        else if (u.pcm_offset != s.cur_off) { close_frame(s); s.cur_off = u.pcm_offset; s.frame++; s.mask = 0; }
        if (u.n_out_ch != s.n_out) return fail(err, AACG_ERR_LAYOUT_CHANGE, "unit %ld: stream %ld changes channel count inside a batch", i, u.stream);
        if (u.pcm_offset & 3u) s.aligned = false;
        frame_of[i] = s.frame;
        const bool independent = is_cce && cce[u.reserved1].coupling_point == AACG_CCE_AFTER_IMDCT;
        if (!is_cce) {
            const uint32_t bits = ((1u << u.n_ch) - 1u) << u.channel;
            if (s.mask & bits) return fail(err, AACG_ERR_INVALID_ARG, "unit %ld: channel %ld written twice in one frame", i, u.channel);
            s.mask |= bits;
        }

        const uint64_t key = ((uint64_t)u.stream << 16) | u.channel;
        open_chain* found = nullptr;
        const bool chained = !is_cce || independent;           /* dependent coupling: spectrum only, no filterbank, no state */
        const uint32_t cached = chained ? std::min<uint32_t>(s.n_chains, 4u) : 0u;
        for (uint32_t k = 0; k < cached; k++) if (s.chain_channel[k] == u.channel) { found = s.chain[k]; break; }
        if (chained && !found && s.n_chains > 4u) { auto it = open.find(key); if (it != open.end()) found = &it->second; }
        if (!chained) {
        } else if (!found) {
            if (s.frame != 0) return fail(err, AACG_ERR_LAYOUT_CHANGE, "unit %ld: element at channel %ld appears mid-batch", i, u.channel);
            open_chain oc; oc.last_frame = 0; oc.n_ch = u.n_ch; oc.is_cce = is_cce; oc.units.reserve(16); oc.units.push_back((int32_t)i);
            auto at = open.emplace(key, std::move(oc)).first;
            if (s.n_chains < 4u) { s.chain[s.n_chains] = &at->second; s.chain_channel[s.n_chains] = u.channel; }
            s.n_chains++;
        } else {
            open_chain& oc = *found;
            if (oc.n_ch != u.n_ch || oc.last_frame + 1 != s.frame)
                return fail(err, AACG_ERR_LAYOUT_CHANGE, "unit %ld: element layout of stream %ld changes inside a batch", i, u.stream);
            oc.last_frame = s.frame;
            oc.units.push_back((int32_t)i);
        }

        /* device copy with the group-of-window map: 4 bits per window (ics.js:288-296 grouping) */
        aacg_dev_unit du;
        std::memset(&du, 0, sizeof du);
        du.d = u;
        if (independent) {                                     /* its filterbank output goes to the side buffer, planar */
            du.d.pcm_offset = out->side_blocks++ * 1024u;
            du.d.n_out_ch = 1;
            du.d.channel = 0;
        }
        for (int c = 0; c < 2; c++) {
            du.gmap[c] = c < u.n_ch ? group_map(u.ch[c]) : 0;
            if (!tns) du.d.ch[c].flags &= (uint8_t)~AACG_CHAN_TNS_PRESENT;       /* no records: nothing to apply */
            if (tns && c < u.n_ch && (u.ch[c].flags & AACG_CHAN_TNS_PRESENT)) {
                const uint32_t ti = u.tns_offset + (uint32_t)c;
                if (ti >= n_tns) return fail(err, AACG_ERR_INVALID_ARG, "unit %ld ch %ld: tns_offset outside the TNS array", i, c);
                if (out->tns.size() < n_tns) out->tns.resize(n_tns);
                int trc = aacg_tns_prepare(sample_index, &u.ch[c], &tns[ti], &out->tns[ti]);
                if (trc) return fail(err, trc, "unit %ld ch %ld: TNS filter order > 12 or too many filters", i, c);
                out->any_tns = true;
            }
        }
        out->units.push_back(du);

        /* extents in 64 bits: an offset near UINT32_MAX (the records come straight from a caller's byte array) must not
         * wrap to a small extent and pass the bounds checks */
        if (u.coef_offset > UINT32_MAX - 2u || u.meta_offset > UINT32_MAX - 2u || (uint64_t)u.pcm_offset + 1024u * (uint64_t)u.n_out_ch > UINT32_MAX)
            return fail(err, AACG_ERR_INVALID_ARG, "unit %ld: coefficient / meta / pcm offset out of range", i, 0);
        out->coef_blocks = std::max(out->coef_blocks, u.coef_offset + u.n_ch);
        out->meta_blocks = std::max(out->meta_blocks, u.meta_offset + u.n_ch);
        out->pcm_floats = std::max(out->pcm_floats, (size_t)((uint64_t)u.pcm_offset + 1024u * (uint64_t)u.n_out_ch));
    }

    out->wide_frames = 2u * n_wide >= n_units && n_wide > 0;
    std::vector<uint32_t> couple_target;                     /* target unit of couple_jobs[j] */
    if (out->any_cce) {
        /* coupling jobs: every (coupling element, target) pair, by coupling point and by round — the r-th coupling element
         * of a frame is in round r, so that the jobs of one launch never add to the same channel (decoder.js:411-431 walks
         * the elements in order; additions commute up to rounding only within the tolerance, so the order is kept) */
        out->gains.resize((size_t)n_cce * AACG_CCE_MAX_TARGETS * AACG_MAX_SECTIONS);
        for (uint32_t c = 0; c < n_cce; c++) std::memcpy(&out->gains[(size_t)c * AACG_CCE_MAX_TARGETS * AACG_MAX_SECTIONS], cce[c].gain, sizeof cce[c].gain);
        std::map<uint64_t, std::vector<uint32_t>> frames;      /* (stream, frame) -> its units */
        for (uint32_t i = 0; i < n_units; i++) frames[((uint64_t)units[i].stream << 32) | frame_of[i]].push_back(i);
        struct keyed { uint32_t key; uint32_t target; aacg_couple_job job; };
        std::vector<keyed> jobs;
        for (auto& fr : frames) {
            uint32_t round = 0;
            for (uint32_t i : fr.second) {
                const aacg_unit_desc& u = units[i];
                if (!(u.flags & AACG_UNIT_CCE)) continue;
                const aacg_cce_info& ci = cce[u.reserved1];
                for (int t = 0; t < ci.n_targets; t++) {
                    const uint32_t tch = ci.target[t].channel;
                    /* one coupling element's jobs share a launch, and a launch must not add to one channel twice (a
                     * read-modify-write race, and the reference's order of additions would be lost): a coupling element
                     * that names a channel twice is refused (cce.js:56-66 lists distinct targets) */
                    for (int t2 = 0; t2 < t; t2++)
                        if (ci.target[t2].channel == tch) return fail(err, AACG_ERR_INVALID_ARG, "unit %ld: coupling element lists target channel %ld twice", i, tch);
                    int32_t target = -1;
                    for (uint32_t j : fr.second)
                        if (!(units[j].flags & AACG_UNIT_CCE) && tch >= units[j].channel && tch < (uint32_t)units[j].channel + units[j].n_ch) target = (int32_t)j;
                    if (target < 0) return fail(err, AACG_ERR_INVALID_ARG, "unit %ld: coupling target channel %ld is not in the frame", i, tch);
                    const aacg_unit_desc& tu = units[target];
                    keyed k;
                    std::memset(&k, 0, sizeof k);
                    k.key = ci.coupling_point * 4096u + round;
                    k.target = (uint32_t)target;
                    k.job.cce_unit = i;
                    k.job.gain_off = (uint32_t)(((size_t)u.reserved1 * AACG_CCE_MAX_TARGETS + ci.target[t].gain_list) * AACG_MAX_SECTIONS);
                    if (ci.coupling_point == AACG_CCE_AFTER_IMDCT) {
                        k.job.src = out->units[i].d.pcm_offset / 1024u;
                        k.job.dst = tu.pcm_offset + tch;
                        k.job.stride = tu.n_out_ch;
                    } else {
                        k.job.src = u.coef_offset;
                        k.job.dst = tu.coef_offset + (tch - tu.channel);
                    }
                    jobs.push_back(k);
                }
                round++;
                if (round >= 4096u) return fail(err, AACG_ERR_CAPACITY, "too many coupling elements in one frame");
            }
            out->couple_rounds = std::max(out->couple_rounds, round);
        }
        std::stable_sort(jobs.begin(), jobs.end(), [](const keyed& a, const keyed& b) { return a.key < b.key; });
        /* couple_first[point * rounds + round] = first job of that launch; one entry more closes the last */
        out->couple_first.assign((size_t)3 * out->couple_rounds + 1, (uint32_t)jobs.size());
        for (size_t j = jobs.size(); j-- > 0;) out->couple_first[(size_t)(jobs[j].key / 4096u) * out->couple_rounds + jobs[j].key % 4096u] = (uint32_t)j;
        for (size_t k = out->couple_first.size() - 1; k-- > 0;) out->couple_first[k] = std::min(out->couple_first[k], out->couple_first[k + 1]);
        for (auto& k : jobs) { out->couple_jobs.push_back(k.job); couple_target.push_back(k.target); }
    }
    for (auto& s : st) close_frame(s);
    /* every chain must reach its stream's last frame, or a later batch would chain onto a stale tail */
    for (auto& kv : open)
        if (kv.second.last_frame != st[(size_t)(kv.first >> 16)].frame)
            return fail(err, AACG_ERR_LAYOUT_CHANGE, "stream %ld: element at channel %ld ends before the batch does",
                        (long)(kv.first >> 16), (long)(kv.first & 0xffff));

    /* chains -> runs, generated chain by chain */
    std::vector<aacg_run> gen, cce_gen;
    std::vector<aacg_run> gen_rv;
    std::vector<aacg_rv_link> gen_rv_links;
    bool long_chain = false;
    for (auto& kv : open) {
        const open_chain& oc = kv.second;
        aacg_chain ch;
        ch.stream = (uint32_t)(kv.first >> 16);
        ch.channel = (uint16_t)(kv.first & 0xffff);
        ch.n_ch = oc.n_ch;
        ch.first_run = (uint32_t)gen.size();
        for (int c = 0; c < 2; c++)
            ch.parity[c] = (parity && c < oc.n_ch) ? parity[(size_t)ch.stream * (size_t)max_channels + ch.channel + c] : 0;
        const size_t n = oc.units.size();
        /* The first run takes 16 frames.  A later run recomputes the tail of the frame before it: with up to 15
         * frames a wave of its own does that, a full run of 16 gives its first wave double duty (one IMDCT more in
         * series).  Use as few double-duty runs as it takes to reach the minimum number of runs. */
        size_t n_full = 0;
        if (n > AACG_RUN_W && !oc.is_cce) {                /* coupling elements' runs: their own launch of the plain kernel */
            const size_t rem = n - AACG_RUN_W, later = (rem + AACG_RUN_W - 1) / AACG_RUN_W;
            n_full = rem > later * (AACG_RUN_W - 1) ? rem - later * (AACG_RUN_W - 1) : 0;
        }
        for (size_t pos = 0; pos < n;) {
            aacg_run r;
            size_t cap = AACG_RUN_W;
            if (pos) { if (n_full) n_full--; else cap = AACG_RUN_W - 1; }
            r.pred_unit = pos ? oc.units[pos - 1] : -1;
            r.n_units = (int32_t)std::min<size_t>(cap, n - pos);
            for (int k = 0; k < AACG_RUN_W; k++) r.unit[k] = k < r.n_units ? oc.units[pos + k] : -1;
            for (int c = 0; c < 2; c++) {
                const uint32_t chn = ch.channel + (c < oc.n_ch ? c : 0);
                r.ov0[c] = aacg_ov_offset(max_channels, ch.stream, chn, 0);
                r.rot[c] = ch.parity[c < oc.n_ch ? c : 0];
            }
            if (pos && r.n_units == AACG_RUN_W) out->needs_scratch = true;
            pos += (size_t)r.n_units;
            r.is_last = pos >= n ? 1 : 0;
            (oc.is_cce ? cce_gen : gen).push_back(r);
        }
        ch.n_runs = oc.is_cce ? 0 : (uint32_t)gen.size() - ch.first_run;
        out->chains.push_back(ch);
        /* the same chain for the 16-wave kernels with a rendezvous between its runs instead of a recomputed frame */
        if (!oc.is_cce) {
            if (n > AACG_RUN_W) long_chain = true;
            int32_t link = -1;
            for (size_t pos = 0; pos < n; pos += AACG_RUN_W) {
                aacg_run r;
                r.pred_unit = -1;
                r.n_units = (int32_t)std::min<size_t>(AACG_RUN_W, n - pos);
                for (int k = 0; k < AACG_RUN_W; k++) r.unit[k] = k < r.n_units ? oc.units[pos + k] : -1;
                for (int c = 0; c < 2; c++) {
                    const uint32_t chn = ch.channel + (c < oc.n_ch ? c : 0);
                    r.ov0[c] = aacg_ov_offset(max_channels, ch.stream, chn, 0);
                    r.rot[c] = ch.parity[c < oc.n_ch ? c : 0];
                }
                const bool more = pos + AACG_RUN_W < n;
                r.is_last = more ? 0 : 1;
                aacg_rv_link lk;
                lk.link_in = link;
                link = more ? (int32_t)out->n_links_rv++ : -1;
                lk.link_out = link;
                lk.succ_unit = more ? oc.units[pos + AACG_RUN_W] : -1;
                lk.reserved = 0;
                gen_rv.push_back(r);
                gen_rv_links.push_back(lk);
            }
        }
    }

    /* Independent coupling (cce.js:121-128) where the target's PCM is formed instead of a read-modify-write pass over the
     * interleaved PCM: the jobs regrouped by target unit, in the order of the frame's coupling elements (the reference adds
     * them in that order, decoder.js:411-431).  Plans with double-duty runs keep the separate pass (their kernels have no
     * coupling epilogue). */
    if (out->any_cce && !out->needs_scratch && out->couple_rounds) {
        const uint32_t rounds = out->couple_rounds;
        const uint32_t first = out->couple_first[(size_t)AACG_CCE_AFTER_IMDCT * rounds], last = out->couple_first[(size_t)AACG_CCE_AFTER_IMDCT * rounds + rounds];
        if (last > first) {
            struct by_unit { uint32_t unit, order; aacg_couple_job job; };
            std::vector<by_unit> list;
            for (uint32_t j = first; j < last; j++) {
                const aacg_unit_desc& tu = units[couple_target[j]];
                by_unit b; b.unit = couple_target[j]; b.order = j; b.job = out->couple_jobs[j];
                b.job.dst = out->couple_jobs[j].dst - (tu.pcm_offset + tu.channel); b.job.stride = 0;     /* channel 0 / 1 of the target unit */
                list.push_back(b);
            }
            if (list.size() == (size_t)(last - first)) {
                std::stable_sort(list.begin(), list.end(), [](const by_unit& a, const by_unit& b) { return a.unit != b.unit ? a.unit < b.unit : a.order < b.order; });
                for (size_t j = 0; j < list.size(); j++) {
                    aacg_dev_unit& du = out->units[list[j].unit];
                    if (du.cpl_n == 0) du.cpl_first = (uint32_t)j;
                    du.cpl_n++;
                    out->fused_jobs.push_back(list[j].job);
                }
                /* they travel behind the launch-ordered jobs in the same array (couple_first only indexes the part in front) */
                out->fused_first = (uint32_t)out->couple_jobs.size();
                out->couple_jobs.insert(out->couple_jobs.end(), out->fused_jobs.begin(), out->fused_jobs.end());
                out->fused_independent = true;
            }
        }
    }

    /* XCD-aware block order: the dispatcher puts block b on XCD b % 8 (observed, speed only), and a
     * run re-reads the last frame of the run before it; keep neighbours on one XCD's L2. */
    const size_t R = gen.size();
    out->runs.resize(R);
    size_t i = 0;
    for (size_t x = 0; x < 8 && x < R; x++) {
        const size_t cnt = (R - 1 - x) / 8 + 1;
        for (size_t s = 0; s < cnt; s++) out->runs[s * 8 + x] = gen[i++];
    }
    /* the rendezvous cut of the same chains (every run 16 frames, links between consecutive runs): the route of plain batches
     * with a chain longer than a run, and of every plain batch launched through aacg_decode_pipelined */
    out->long_chains = long_chain;
    {
        const size_t RR = gen_rv.size();
        out->runs_rv.resize(RR);
        out->links_rv.resize(RR);
        i = 0;
        for (size_t x = 0; x < 8 && x < RR; x++) {
            const size_t cnt = (RR - 1 - x) / 8 + 1;
            for (size_t s = 0; s < cnt; s++) { out->runs_rv[s * 8 + x] = gen_rv[i]; out->links_rv[s * 8 + x] = gen_rv_links[i]; i++; }
        }
    }
    out->cce_runs = cce_gen;
    aacg_plan_fill_run_waves(out);
    /* chain.first_run refers to generation order; the engine only needs counts, keep as is */
    return AACG_OK;
}
